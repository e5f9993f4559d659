// Launch arguments shared by the forward convolution kernels (conv_igemm.hip, conv_u.hip).
#pragma once
#include "common.h"

namespace cl4 {

struct ConvArgs {
  const float* x;
  const float* wp;
  const float* scale;
  const float* shift;
  const float* pre;
  const float* post;
  float* y;
  int n, cin, hin, win, cout, stride, pad, dil, hout, wout;
  int in_ctot, in_coff, out_ctot, out_coff;
  int act1, act2;
  float slope1, slope2;
  int K;        // cin*KS*KS
  int npix;     // n*hout*wout — or n*hw_pad for the bf16 launches of a "vector gather" plan (see hw_pad)
  int hw_pad;   // logical pixels per image: pixel p = image p / hw_pad, position p % hw_pad (positions >= hout*wout are padding).
                // = hout*wout except for 1x1 plans whose activation loader fetches 4 pixels per lane (hw_pad = roundup(hout*wout, 4),
                // so that a 4-pixel group never straddles two images)
  int xcd_remap;      // 1: XCD-aware logical tile order (bf16x3 kernel)
  float neg1, neg2;   // branch-free activations: act(v) = v > 0 ? v : v * neg  (none: 1, relu: 0, leaky: slope)
  float* ws;    // split-K partial slabs [splits][cout][npix] (v2 kernel only)
  int kchunk;   // k range per split (multiple of 32)
  // bf16 kernels (1-D launches): logical tile = tile_base + (workgroup % ntile_l), M index fastest over gx tiles; the split-K
  // slabs cover the pixels [ws_p0, ws_p0 + ws_npix) only ("main + split tail" plans split just the last partial round)
  int gx, ntile_l, tile_base, ws_p0, ws_npix;
  // Experiments only (tools/probes/probe_timeline.py): when set, every workgroup of a bf16 launch records s_memtime stamps and its
  // XCC / CU ids into timeline[blockIdx * 8 ..]: [0] start, [1] first K-tile ready, [2] K loop done, [3] accumulators parked,
  // [4] epilogue done (MFMA wave 0), [5] loader wave 0 done, [6] (xcc << 8) | cu, [7] logical tile.
  unsigned long long* timeline;
  // "Planes" operands (conv_s.hip): a tensor stored as pre-split bf16 planes, [n][c/8][plane: hi, lo][h*w][8 x bf16] — one 16-byte
  // chunk = 8 consecutive channels of one pixel of one plane = one MFMA operand fragment row. xs: the input in that format (x is
  // then unused), ys: an additional / alternative output in that format (y may be null), post_s: the residual operand in that
  // format (instead of post). Channel slices use in_ctot / in_coff / out_ctot / out_coff in units of channels (multiples of 8).
  const void* xs;
  void* ys;
  const void* post_s;
  // fp16x3 launches (r05, precision code 5): the planes operand holds fp16 hi / lo of x * s and the packed weights fp16 hi / lo of
  // w * 2^12; *xs_scale = s (a power of two written by cl4_to_planes_f16), and the accumulators are multiplied by 1 / (2^12 * s)
  // before the epilogue. Null for every bf16 launch.
  const float* xs_scale;
  // Experiments only (cl4_conv_debug_stagger, tools/probes/probe_b.py): first-round workgroups on odd CUs start this many REFCLK ticks (10 ns)
  // late, so that neighbouring CUs run their K loops and their epilogues in anti-phase. 0 in every product launch.
  int dbg_stagger;
  // r05: per-channel partial sums of the OUTPUT for the train-mode BatchNorm behind the convolution (cl4_conv2d_fwd_planes_stats): every
  // workgroup of an unsplit planes launch adds up the rows of its parked tile — stats[(channel * stats_np + pixel tile) * 2 + {0, 1}] =
  // {sum, sum of squares} in double, the layout the BatchNorm kernels' channel_sums folds. Null otherwise.
  double* stats;
  int stats_np;
};

// {sum, sum of squares} of every row of a parked fp32 tile [rows][BNT] (LDS) -> a.stats; RPW rows per wave starting at wave * RPW, row r of the
// tile = output channel m0 + r; columns past the tensor's last pixel hold zeros (their operands were zero-filled). Called by all waves.
// A lane owns a PART of one row (64 / RPW lanes per row) and walks its 16-byte chunks with a per-row skew (lanes of different rows on different
// banks); the parts are combined by one or two shuffles. (The first version reduced one row at a time across the lanes — 12 dependent
// double-precision shuffle steps per row, ~9 us per 256 x 144 tile: the convolutions lost more than the statistics pass had cost.)
template <int BNT, int RPW>
__device__ __forceinline__ void tile_row_stats(const ConvArgs& a, const float* __restrict__ tile, int m0, int pixel_tile, int wave, int lane) {
  typedef float f32x4s __attribute__((ext_vector_type(4)));
  constexpr int Q = BNT / 4, LPR = 64 / RPW, QS = Q / LPR;      // chunks per row, lanes per row, chunks per lane
  static_assert(BNT % 4 == 0 && 64 % RPW == 0 && Q % LPR == 0, "tile_row_stats: row parts");
  const int r = lane % RPW, part = lane / RPW;
  const int row = wave * RPW + r, co = m0 + row;
  const f32x4s* src = reinterpret_cast<const f32x4s*>(tile + (long)row * BNT) + part * QS;
  float s = 0.f, s2 = 0.f;
#pragma unroll 4
  for (int j = 0; j < QS; ++j) {
    int q = j + r;
    q = q >= QS ? q - QS * (q / QS) : q;
    const f32x4s v = src[q];
    s += (v[0] + v[1]) + (v[2] + v[3]);
    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
#pragma unroll
  for (int o = RPW; o < 64; o <<= 1) {
    s += __shfl_xor(s, o);
    s2 += __shfl_xor(s2, o);
  }
  if (part == 0 && co < a.cout) {
    a.stats[((long)co * a.stats_np + pixel_tile) * 2 + 0] = (double)s;
    a.stats[((long)co * a.stats_np + pixel_tile) * 2 + 1] = (double)s2;
  }
}

// conv_u.hip: the unified-wave bf16x3 kernel (every wave loads, splits and multiplies). One launch of `grid` workgroups of 256
// threads; ks = 1 | 3, split = K-split slabs (a.ws) instead of the fused epilogue.
int launch_conv_u(const ConvArgs& a, int ks, bool split, unsigned grid, hipStream_t st);

// conv_s.hip: the planes-input bf16x3 kernel (activations arrive pre-split: no conversion in the K loop). Any stride / pad / dilation,
// ks = 1 | 3, cin % 32 == 0. Same launch conventions as launch_conv_u.
// bn = pixels per tile: 128, or 144 (one round of the 33 x 33 maps, see conv_s.hip).
// planes = bf16 planes per operand: 2 (bf16x3) or 3 (bf16x6; 144-pixel tiles only).
int launch_conv_s(const ConvArgs& a, int ks, bool split, unsigned grid, hipStream_t st, int bn = 128, int planes = 2);

// conv_p.hip: the persistent, software-pipelined planes kernel for short-K 1x1 layers with planes output (and planes residual): one
// 512-thread workgroup per CU walks the 128 x 128 tiles and overlaps tile t's register-direct epilogue with tile t+1's K loop.
bool conv_p_eligible(const ConvArgs& a, int ks);
int launch_conv_p(const ConvArgs& a, unsigned ntile, hipStream_t st);


// conv_b.hip: the LDS-DMA planes kernel (operands straight into LDS, three LDS stages, one workgroup per CU) for launches whose K loop
// dominates. rows = output channels per tile: 256 (8 waves) or 128 (4 waves, or 8 waves of 32 rows each: launches with too few 256-row
// tiles to fill the chip); 144 pixels per tile. Unsplit; `grid` = cdiv(cout, rows) * cdiv(npix, 144) with a.gx = cdiv(cout, rows).
bool conv_b_eligible(const ConvArgs& a, int ks, int rows);
// planes = 2, or 3 (bf16x6 operands; rows 128 / 8 waves only).
// f16: fp16x3 operands (two fp16 planes, a.xs_scale set) instead of bf16x3; 8 waves, two planes only.
int launch_conv_b(const ConvArgs& a, int ks, int rows, int waves, int planes, unsigned grid, hipStream_t st, bool f16 = false, bool split = false);

}  // namespace cl4
