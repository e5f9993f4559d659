// Dense convolution for NCHW fp32 activations as an implicit GEMM on the gfx950 fp32 MFMA
// pipe (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 64 FLOP/clk/SIMD).
//
// GEMM view:  Y[co][p] = sum_k  Wp[k][co] * Xg[k][p]
//   M = cout (weights, pre-packed [K][cout] so global loads run along co),
//   N = pixels p = (n, oy, ox)   (NCHW keeps them contiguous per channel -> coalesced),
//   K = cin*kh*kw, Xg gathered on the fly (im2col never materialised).
// Both LDS tiles are k-major with the M/N index contiguous, which is exactly the operand
// map of the f32 MFMA (lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]): every ds_read_b32
// touches 32 consecutive floats per half-wave and is bank-conflict free without padding.
// The accumulator keeps the pixel on the lane and 16 output channels in registers, so each
// epilogue store writes 32 consecutive pixels of one channel (128-B segments).
//
// Epilogue (fused; replaces the separate InPlaceABN / add / leaky-ReLU passes of the
// reference, modules/residual.py:84-97, modules/deeplab.py:54-70):
//   y = act2( act1( (acc + pre) * scale[co] + shift[co] ) + post )
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include "common.h"
#include "conv_args.h"
#include "planes_fmt.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;       // K-tile of the forward kernel
constexpr int NTHREADS = 256;

using cl4::ConvArgs;


// Fused epilogue shared by the forward kernels: y = act2( act1( (acc + pre) * scale[co] + shift[co] ) + post ).
// The four optional operands are tested ONCE per 16-element accumulator block (wave-uniform branches) and each block's
// loads are issued together before any arithmetic — per-element `if (a.pre)` made hipcc emit 64 tiny basic blocks with a
// vmcnt(0) after every load, which serialised the epilogue of the small-K layers.
template <int TM, int TN>
__device__ __forceinline__ void fused_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], int co_base, int p_base, int lr,
                                               int lh, int HWo) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int p = p_base + j * 32 + lr;
    if (p >= a.npix) continue;
    const int n = p / a.hw_pad;
    const int hw = p - n * a.hw_pad;
    if (hw >= HWo) continue;
    const long ybase = ((long)n * a.out_ctot + a.out_coff) * HWo + hw;
    const long rbase = (long)n * a.cout * HWo + hw;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int cc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cc[r] = min(co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, a.cout - 1);
      float v[16], sc[16], sh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
      if (a.pre) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a.pre[rbase + (long)cc[r] * HWo];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t[r];
      }
      if (a.scale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = a.scale[cc[r]];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 1.f;
      }
      if (a.shift) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[r] = a.shift[cc[r]];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sh[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = v[r] * sc[r] + sh[r];
        v[r] = t > 0.f ? t : t * a.neg1;
      }
      if (a.post) {
        float t[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = a.post[rbase + (long)cc[r] * HWo];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < a.cout) a.y[ybase + (long)co * HWo] = v[r] > 0.f ? v[r] : v[r] * a.neg2;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KS>
__global__ __launch_bounds__(NTHREADS) void conv_igemm_f32(ConvArgs a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 MFMA tile");
  constexpr int A_PER = (BM + 63) / 64;  // floats per lane per k-row of the A tile
  constexpr int B_PER = BN / 64;
  constexpr int ROWS = BK / 4;           // k-rows staged by each wave
  constexpr int KHW = KS * KS;

  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // Workgroup -> tile. blockIdx.x walks the M tiles fastest so that consecutive workgroups
  // (dealt round-robin over the 8 XCDs) share the same pixel tile through L2 / Infinity Cache.
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  const int HWo = a.hout * a.wout;
  const int HWi = a.hin * a.win;

  // Decode this lane's B_PER pixels once: they stay fixed over the whole K loop.
  long xbase[B_PER];
  int iy0[B_PER], ix0[B_PER];
  bool pvalid[B_PER];
#pragma unroll
  for (int q = 0; q < B_PER; ++q) {
    int p = n0 + lane + 64 * q;
    pvalid[q] = p < a.npix;
    int pc = pvalid[q] ? p : 0;
    int n = pc / HWo;
    int hw = pc - n * HWo;
    int oy = hw / a.wout;
    int ox = hw - oy * a.wout;
    iy0[q] = oy * a.stride - a.pad;
    ix0[q] = ox * a.stride - a.pad;
    xbase[q] = ((long)n * a.in_ctot + a.in_coff) * HWi;
  }

  float ra[ROWS][A_PER], rb[ROWS][B_PER];

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int k = kt * BK + wave * ROWS + r;  // wave-uniform
      const bool kval = k < a.K;
      const int kc = kval ? k : 0;
#pragma unroll
      for (int q = 0; q < A_PER; ++q) {
        int ml = lane + 64 * q;
        int m = m0 + ml;
        bool ok = kval && (ml < BM) && (m < a.cout);
        ra[r][q] = ok ? a.wp[(long)kc * a.cout + m] : 0.f;
      }
      int ci, dy, dx;
      if (KS == 1) {
        ci = kc; dy = 0; dx = 0;
      } else {
        ci = kc / KHW;
        int t = kc - ci * KHW;
        int ky = t / KS;
        dy = ky * a.dil;
        dx = (t - ky * KS) * a.dil;
      }
      const long coff = (long)ci * HWi;
#pragma unroll
      for (int q = 0; q < B_PER; ++q) {
        int iy = iy0[q] + dy, ix = ix0[q] + dx;
        bool ok = kval && pvalid[q] && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
        rb[r][q] = ok ? a.x[xbase[q] + coff + (long)iy * a.win + ix] : 0.f;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int kr = wave * ROWS + r;
#pragma unroll
      for (int q = 0; q < A_PER; ++q) {
        int ml = lane + 64 * q;
        if (ml < BM) As[buf][kr][ml] = ra[r][q];
      }
#pragma unroll
      for (int q = 0; q < B_PER; ++q) Bs[buf][kr][lane + 64 * q] = rb[r][q];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);  // global loads in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[buf][kk + lh][wm * WTM + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][kk + lh][wn * WTN + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  fused_epilogue<TM, TN>(a, acc, m0 + wm * WTM, n0 + wn * WTN, lr, lh, HWo);
}


// ------------------------------------------------------------------------------------
// v2 forward kernel: same GEMM view and LDS image as above, but both operands are staged by LDS-DMA
// (global_load_lds: global -> LDS with no VGPR round trip and no ds_write), BK = 32, two LDS buffers.
//   * weights: one global_load_lds_dwordx4 moves two (BM=128) / four (BM=64) whole k-rows;
//   * activations: global_load_lds_dword with a PER-LANE source address = the im2col gather; lanes that fall
//     into the zero padding (or past K / past the last pixel) read from a small zero page instead, so EXEC
//     stays full and the LDS image needs no separate clearing;
//   * the DMA of tile t+1 is issued before the 64 MFMAs of tile t and retired by the single
//     vmcnt(0)+barrier at the end of the iteration.
// Requires cout % 4 == 0 (16-byte weight rows); other shapes use the register-staged kernel above.
// ------------------------------------------------------------------------------------
constexpr int BK2 = 32;

#define CL4_GLDS(gptr, lptr, bytes)                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(lptr), bytes, 0, 0)

// Wave specialisation: a workgroup is 8 waves = 4 MFMA waves (one per SIMD, each owning a quarter of the output
// tile) + 4 DMA waves (one per SIMD) that only issue the LDS-DMA of the next K-tile. An LDS-DMA wave-instruction
// costs the issuing wave ~60-180 cycles (MI355X_MICROARCH.md), i.e. the 20 of them per K-tile rival the tile's MFMA
// time when they sit in the MFMA wave's own in-order stream; in a partner wave they overlap it instead. The two
// roles meet at one s_barrier per K-tile: DMA waves arrive after `vmcnt(0)` (tile t+1 landed), MFMA waves after
// consuming tile t.
constexpr int V2_THREADS = 512;

template <int BM, int BN, int KS, bool SPLIT>
__global__ __launch_bounds__(V2_THREADS, 2) void conv_igemm_f32_v2(ConvArgs a, const float* __restrict__ zero_page) {
  constexpr int WAVES_M = 2, WAVES_N = 2;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KHW = KS * KS;
  constexpr int A_ROWS_PER_INSTR = 256 / BM;             // k-rows covered by one dwordx4 wave-instruction
  constexpr int A_INSTR = BK2 / A_ROWS_PER_INSTR / 4;    // per wave
  constexpr int B_ROWS = BK2 / 4;                        // k-rows per wave
  constexpr int B_PER = BN / 64;                         // dword instructions per k-row

  __shared__ __attribute__((aligned(16))) float lds[2][BK2 * (BM + BN)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_role = wave8 >= 4;
  const int wave = wave8 & 3;            // index within the role
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int HWo = a.hout * a.wout;
  const int HWi = a.hin * a.win;
  // split-K: this workgroup reduces k in [kbeg, kend) (kchunk is a multiple of BK2)
  const int kbeg = SPLIT ? blockIdx.z * a.kchunk : 0;
  const int kend = SPLIT ? min(a.K, kbeg + a.kchunk) : a.K;

  const float* xptr[B_PER];
  int iy0[B_PER], ix0[B_PER];
  bool pvalid[B_PER];
#pragma unroll
  for (int q = 0; q < B_PER; ++q) {
    int p = n0 + lane + 64 * q;
    pvalid[q] = p < a.npix;
    int pc = pvalid[q] ? p : 0;
    int n = pc / HWo;
    int hw = pc - n * HWo;
    int oy = hw / a.wout;
    int ox = hw - oy * a.wout;
    iy0[q] = oy * a.stride - a.pad;
    ix0[q] = ox * a.stride - a.pad;
    xptr[q] = a.x + ((long)n * a.in_ctot + a.in_coff) * HWi + (long)iy0[q] * a.win + ix0[q];
  }
  // weight source: lane -> (row within the instruction, 4 consecutive output channels)
  const int a_row = lane / (BM / 4);
  const int a_m = (lane % (BM / 4)) * 4;
  const bool a_mvalid = (m0 + a_m) < a.cout;
  const float* wsrc = a.wp + m0 + a_m;

  auto issue_tile = [&](int kt, int buf) {
    float* As = lds[buf];
    float* Bs = lds[buf] + BK2 * BM;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const int r0 = (wave * A_INSTR + i) * A_ROWS_PER_INSTR;   // wave-uniform first row of this instruction
      const int k = kbeg + kt * BK2 + r0 + a_row;
      const float* g = (a_mvalid && k < kend) ? wsrc + (long)k * a.cout : zero_page;
      CL4_GLDS(g, As + r0 * BM, 16);
    }
#pragma unroll
    for (int r = 0; r < B_ROWS; ++r) {
      const int kr = wave * B_ROWS + r;
      const int k = kbeg + kt * BK2 + kr;     // wave-uniform
      const bool kval = k < kend;
      const int kc = kval ? k : 0;
      int ci, dy, dx;
      if (KS == 1) {
        ci = kc; dy = 0; dx = 0;
      } else {
        ci = kc / KHW;
        int t = kc - ci * KHW;
        int ky = t / KS;
        dy = ky * a.dil;
        dx = (t - ky * KS) * a.dil;
      }
      const long koff = (long)ci * HWi + (long)dy * a.win + dx;
#pragma unroll
      for (int q = 0; q < B_PER; ++q) {
        bool ok = kval && pvalid[q];
        if (KS != 1 || a.pad != 0) {
          int iy = iy0[q] + dy, ix = ix0[q] + dx;
          ok = ok && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
        }
        const float* g = ok ? xptr[q] + koff : zero_page;
        CL4_GLDS(g, Bs + kr * BN + 64 * q, 4);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg + BK2 - 1) / BK2;
  const int lr = lane & 31, lh = lane >> 5;
  if (dma_role) {
    // ---- DMA waves: stream the K-tiles into the two LDS buffers, one tile ahead of the MFMA waves -------------
    issue_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): this wave's DMA has landed
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) issue_tile(kt + 1, (kt & 1) ^ 1);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  // ---- MFMA waves -------------------------------------------------------------------------------------------
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const float* As = lds[kt & 1] + wm * WTM + lr;
    const float* Bs = lds[kt & 1] + BK2 * BM + wn * WTN + lr;
    // operands of k-step kk+2 are fetched from LDS while the MFMAs of k-step kk run (two register sets)
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = As[lh * BM + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = Bs[lh * BN + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK2; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < BK2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nxt][i] = As[(kk + 2 + lh) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nxt][j] = Bs[(kk + 2 + lh) * BN + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int p = n0 + wn * WTN + j * 32 + lr;
    if (!SPLIT || p >= a.npix) continue;
    {
      // raw partial sums -> workspace [split][cout][npix]; the epilogue runs in conv_splitk_epilogue
      float* wsp = a.ws + ((long)blockIdx.z * a.cout) * a.npix + p;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < a.cout) wsp[(long)co * a.npix] = acc[i][j][r];
        }
    }
  }
  if (!SPLIT) fused_epilogue<TM, TN>(a, acc, m0 + wm * WTM, n0 + wn * WTN, lr, lh, HWo);
}

// Split-K second pass: sum the partial slabs in a fixed order (bitwise reproducible) and apply the fused epilogue.
// Split-K reduction + fused epilogue over the slab region: grid = (pixel chunks of 256, cout) — one channel row segment per
// workgroup, lane = pixel (coalesced slab reads, residual reads and output writes), one division per thread; the slabs are
// summed in a fixed order (z = 0 .. splits-1) so the result does not depend on scheduling.
__global__ void conv_splitk_epilogue(ConvArgs a, int splits) {
  const int co = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ws_npix) return;
  const long total = (long)a.cout * a.ws_npix;
  const float* w = a.ws + (long)co * a.ws_npix + i;
  float v = 0.f;
  int z = 0;
  for (; z + 4 <= splits; z += 4) {              // four independent loads in flight, same summation order
    const float t0 = w[(long)z * total], t1 = w[(long)(z + 1) * total], t2 = w[(long)(z + 2) * total], t3 = w[(long)(z + 3) * total];
    v += t0; v += t1; v += t2; v += t3;
  }
  for (; z < splits; ++z) v += w[(long)z * total];
  const int HWo = a.hout * a.wout;
  const int p = a.ws_p0 + i;
  const int n = p / a.hw_pad;
  const int hw = p - n * a.hw_pad;
  if (hw >= HWo) return;
  const long r = ((long)n * a.cout + co) * HWo + hw;
  if (a.pre) v += a.pre[r];
  const float sc = a.scale ? a.scale[co] : 1.f;
  const float sh = a.shift ? a.shift[co] : 0.f;
  v = v * sc + sh;
  v = v > 0.f ? v : v * a.neg1;
  if (a.post) v += a.post[r];
  v = v > 0.f ? v : v * a.neg2;
  a.y[((long)n * a.out_ctot + a.out_coff + co) * HWo + hw] = v;
}


// ------------------------------------------------------------------------------------
// v3 forward kernel — "bf16x3": fp32-equivalent products on the bf16 MFMA pipe (16x the fp32 MFMA rate).
// Every fp32 operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the kernel accumulates
//   hi_w*hi_x + hi_w*lo_x + lo_w*hi_x   (fp32 accumulators, v_mfma_f32_32x32x16_bf16),
// dropping only lo_w*lo_x (~2^-18 relative): measured 4.4e-6 relative error per layer vs 4e-7 for the exact fp32
// MFMA — far inside the 1e-3 parity budget — at 3/16 of the MFMA cycles.
//   * weights are split once, offline, into [K/8][plane(hi,lo)][cout][8 x bf16] (cl4_pack_conv_weight_bf16x3): an
//     MFMA A-fragment is one 16-byte chunk, and a K-tile is staged by LDS-DMA (dwordx4, lane-linear along cout);
//   * activations stay fp32 NCHW in HBM. The 4 loader waves gather 8 consecutive k for one pixel per item into
//     registers (coalesced along pixels), split them, and ds_write_b128 the hi / lo fragments — this is also the
//     [k][pixel] -> [pixel][k] transpose the bf16 MFMA operand needs; the 4 MFMA waves only ds_read_b128 + MFMA.
//   * same tiles (128x128 / 64x64, BK 32), split-K and fused epilogue as the fp32 kernels.
// ------------------------------------------------------------------------------------
#ifndef CL4_DBG
#define CL4_DBG 0   // experiments only: 1 = loader issues no loads in the loop, 2 = no convert/LDS stores, 4 = no MFMA
#endif
#ifndef CL4_TL2
#define CL4_TL2 0   // experiments only (tools/probes/probe_phases.py): per-K-tile phase times of loader wave 0 / MFMA wave 0, summed per workgroup
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TAP = true (cin % 32 == 0): K is ordered tap-major (k = tap*cin + ci, the weights are packed in the same order), so a
// 32-deep K-tile holds ONE filter tap and 32 consecutive input channels: padding validity and the pixel address are
// computed once per tile and lane, and each gather is a buffer_load with the lane's byte offset in a VGPR and the
// channel offset in an SGPR (out-of-image lanes carry an out-of-range offset and read 0 through the descriptor's
// bounds check). TAP = false keeps the generic ci-major decode (stem 7x7 with cin = 3).
// P = 2: "bf16x3" (x = hi + lo, three products). P = 3: "bf16x6" (x = hi + mid + lo, 24 mantissa bits; the six products
// hh, hm, mh, hl, lh, mm keep every term down to 2^-24: fp32-level accuracy at 2/16 of the fp32 MFMA cycles). P = 3 is
// built for 64x64 tiles only (48 KB of LDS for the two stages).
// LW / NS: loader waves and register sets per loader wave. 4 / 2 is the shape of all bf16x3 launches and of the 64x64 bf16x6
// tile (two workgroups per CU keep enough bytes in flight). The 128x128 bf16x6 tile owns a whole CU (96 KB LDS); alone,
// 4 loader waves x 2 sets keep only 80 KB in flight (latency-bound at ~150 TFLOP/s, r01i probe), and a wave cannot have more
// than 63 vector-memory instructions outstanding — so that tile runs 8 loader waves x 4 sets (768 threads, 11 loads per
// lane and K-tile, 160 KB in flight).
// PERS (non-split only): a PERSISTENT workgroup walks several output tiles. Its K-tile stream is continuous across tiles —
// the loaders' register sets already hold the first K-tiles of the next tile while the MFMA waves drain the accumulators of
// the previous one — so the pipeline is filled once per workgroup instead of once per tile and no workgroup is re-dispatched
// per tile (short-K 1x1 layers spent more time on those two than in their eight K-tiles, tools/probes/probe_epilogue.py).
// VEC (1x1, stride 1, cin % 32 == 0, unsplit 128x128 launches): the activation loader fetches FOUR CONSECUTIVE PIXELS of one
// channel per lane and load (buffer_load_dwordx4) instead of one pixel of eight channels with eight dword loads. The CU's
// vector-memory front end retires a 256-byte dword wave-load in the same ~16 cycles as a 1-KB dwordx4 one, and with 2 x 4
// loader waves x 20 wave-loads per K-tile it — not L2, HBM or the MFMA pipe — set the K-tile period (tools/probes/probe_timeline.py:
// 2.1-2.5 us per K-tile and workgroup for 0.64 us of MFMA work). 4 dwordx4 + 4 weight loads per lane and K-tile instead of
// 16 + 4. The GEMM column order inside a tile is permuted (column = (pixel % 4) * 32 + pixel / 4) so that the loader's
// ds_writes and the MFMA waves' ds_read_b128 stay bank-conflict free; the wide epilogue undoes it when it streams rows out.
template <int BM, int BN, int KS, bool SPLIT, bool TAP, int P = 2, int LW = 4, int NS = 2, bool PERS = false, bool VEC = false>
__global__ __launch_bounds__(256 + 64 * LW, LW == 4 ? 4 : 3) void conv_igemm_bf16x3(ConvArgs a, const float* __restrict__ zero_page) {
  static_assert(!VEC || (KS == 1 && TAP && P == 2 && BN == 128 && LW == 4 && !SPLIT && !PERS), "vector gather: unsplit 1x1 128-pixel tiles");
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KHW = KS * KS;
  constexpr int KG = BK2 / 8;               // 4 k-groups of 8 per K-tile
  constexpr int NITEM = BN * KG / (LW * 64);   // (pixel, k-group) items per loader lane
  static_assert(NITEM == 1 || NITEM == 2, "loader mapping");
  static_assert(NS % 2 == 0, "the LDS stage of a register set must be static");
  static_assert(!(PERS && SPLIT), "persistent tiles are unsplit");
  static_assert(P == 2 || P == 3, "two (bf16x3) or three (bf16x6) planes");   // 128x128x3 planes = 96 KB LDS: one workgroup per CU
  constexpr int A_INSTR = KG * P * BM / 64 / LW;   // dwordx4 loads per loader wave and K-tile
  constexpr int A_CHUNKS = KG * P * BM;     // 16-byte chunks of the weight tile
  constexpr int B_CHUNKS = KG * P * BN;

  __shared__ u32x4 lds[2][A_CHUNKS + B_CHUNKS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = loader ? wave8 - 4 : wave8;         // index within the role: 0..LW-1 loaders, 0..3 MFMA waves
  const int wm = (wave & 3) >> 1, wn = wave & 1;
  // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs in dispatch order, so ids that
  // are congruent mod 8 share an L2. Remap so that each XCD walks a CONTIGUOUS range of logical tiles with the M index
  // fastest: the M tiles that re-read one pixel tile (and neighbouring pixel tiles that share halo rows) meet in one L2.
  const unsigned total = gridDim.x;            // 1-D launch: ntile_l tiles x splits
  const unsigned id = blockIdx.x;
  unsigned long long* tl = a.timeline ? a.timeline + (size_t)id * 8 : nullptr;
  auto stamp = [&](int slot) {
    if (tl && lane == 0 && (wave8 == 0 || (slot == 5 && wave8 == 4))) tl[slot] = __builtin_amdgcn_s_memtime();
  };
  if (tl && tid == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    tl[6] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
  }
  stamp(0);
  const unsigned q8 = total >> 3, r8 = total & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const unsigned bz = PERS ? 0u : lid / (unsigned)a.ntile_l;
  // PERS: this workgroup is the (id >> 3)-th of the wgx workgroups on its XCD and walks the XCD's contiguous logical range
  // [t_first - (id >> 3), ...) with stride wgx; otherwise one tile.
  unsigned t_first = lid - bz * (unsigned)a.ntile_l, t_stride = 1u;
  int cnt = 1;
  if (PERS) {
    const unsigned nt = (unsigned)a.ntile_l;
    if (a.xcd_remap) {
      const unsigned tq = nt >> 3, tr = nt & 7;
      const unsigned x_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, x_len = tq + (xcd < tr ? 1u : 0u);
      const unsigned wgx = q8 + (xcd < r8 ? 1u : 0u), j = id >> 3;
      t_first = x_start + j;
      t_stride = wgx;
      cnt = j < x_len ? (int)((x_len - j + wgx - 1) / wgx) : 0;
    } else {
      t_first = id;
      t_stride = total;
      cnt = id < nt ? (int)((nt - id + total - 1) / total) : 0;
    }
    if (cnt == 0) return;                      // whole workgroup, before any barrier
  }
  int m0, n0;
  auto tile_origin = [&](int ti) {             // logical tile ti of this workgroup -> (m0, n0)
    const unsigned tile_ = (unsigned)a.tile_base + t_first + (unsigned)ti * t_stride;
    const unsigned by_ = tile_ / (unsigned)a.gx, bx_ = tile_ - by_ * (unsigned)a.gx;
    m0 = bx_ * BM;
    n0 = by_ * BN;
  };
  tile_origin(0);
  const int HWo = a.hout * a.wout;
  const int HWi = a.hin * a.win;
  const int kbeg = SPLIT ? bz * a.kchunk : 0;
  const int kend = SPLIT ? min(a.K, kbeg + a.kchunk) : a.K;
  const int nk = (kend - kbeg + BK2 - 1) / BK2;
  const int TOT = PERS ? cnt * nk : nk;        // K-tiles this workgroup streams
  const int lr = lane & 31, lh = lane >> 5;

  // Wide epilogue (unsplit 8-wave launches): the MFMA waves park their accumulators in LDS (the K-tile stages are free by
  // then) and ALL eight waves — the loaders have nothing else to do — stream the tile out row by row: lane = pixel, so a wave
  // instruction covers 256 contiguous bytes of one channel row (the register epilogue wrote two 128-byte runs per instruction),
  // twice as many waves keep residual loads in flight and each issues a batch of 8 rows before touching the data. The
  // arithmetic per element is exactly fused_epilogue's.
  constexpr bool WIDE = !SPLIT && !PERS;       // LW = 8 (12 waves): waves 0-7 stream, the other loaders only join the barrier
  auto wide_epilogue = [&]() {
    const float* stage = reinterpret_cast<const float*>(&lds[0][0]);       // [BM][BN] fp32, <= the two K-tile stages
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(3);
    if (LW > 4 && wave8 >= 8) return;
    constexpr int RPW = BM / 8, PPL = BN / 64, RB = 8;   // rows per batch: 16 measured slower (registers), 4 no better
    long ybase[PPL], rbase[PPL];
    bool pv[PPL];
    int col[PPL];                                  // column of the parked tile that holds this lane's pixel
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      const int pt = lane + 64 * q;                // pixel within the tile
      col[q] = VEC ? (pt & 3) * 32 + (pt >> 2) : pt;
      const int p = n0 + pt;
      const int n_ = p / a.hw_pad;
      const int hw_ = p - n_ * a.hw_pad;
      pv[q] = p < a.npix && hw_ < HWo;
      const int n = pv[q] ? n_ : 0;
      const int hw = pv[q] ? hw_ : 0;
      ybase[q] = ((long)n * a.out_ctot + a.out_coff) * HWo + hw;
      rbase[q] = (long)n * a.cout * HWo + hw;
    }
#pragma unroll 1
    for (int r0 = 0; r0 < RPW; r0 += RB) {
      const int row0 = wave8 * RPW + r0;
      float v[RB][PPL], t[RB][PPL];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr)
#pragma unroll
        for (int q = 0; q < PPL; ++q) v[rr][q] = stage[(row0 + rr) * BN + col[q]];
      if (a.pre) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) t[rr][q] = a.pre[rbase[q] + (long)min(m0 + row0 + rr, a.cout - 1) * HWo];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) v[rr][q] += t[rr][q];
      }
      if (a.post) {                              // issue the residual loads before the arithmetic that does not need them
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) t[rr][q] = a.post[rbase[q] + (long)min(m0 + row0 + rr, a.cout - 1) * HWo];
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int cc = min(m0 + row0 + rr, a.cout - 1);
        const float sc = a.scale ? a.scale[cc] : 1.f;
        const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
          const float u = v[rr][q] * sc + sh;
          v[rr][q] = u > 0.f ? u : u * a.neg1;
        }
      }
      if (a.post) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) v[rr][q] += t[rr][q];
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int co = m0 + row0 + rr;
        if (co >= a.cout) continue;                // wave-uniform
#pragma unroll
        for (int q = 0; q < PPL; ++q)
          if (pv[q]) a.y[ybase[q] + (long)co * HWo] = v[rr][q] > 0.f ? v[rr][q] : v[rr][q] * a.neg2;
      }
    }
    if (tl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(wave8 == 4 ? 5 : 4);
    }
  };

  if (loader) {
    // item -> (pixel, k-group): BN=128: pixel = lane + 64*(wave&1), k-groups {2*(wave>>1), +1}; BN=64: pixel = lane, kg = wave
    const int pseg = BN == 128 ? (wave & 1) : 0;
    const int kg0 = BN == 128 ? NITEM * (wave >> 1) : wave;
    const int pl = lane + 64 * pseg;
    // per-tile loader state (recomputed when the issue cursor enters a new tile of a persistent workgroup)
    bool pvalid, tile_ok = true;
    int iy0, ix0, lane_base;
    const float* xptr;
    auto set_tile = [&](int ti) {
      tile_ok = ti < cnt;                      // past the last tile: every load goes out of range (zeros, no traffic)
      tile_origin(tile_ok ? ti : 0);
      const int p = VEC ? n0 + 4 * (lane & 31) : n0 + pl;          // VEC: first pixel of this lane's 4-pixel group
      const int n_ = p / a.hw_pad;
      const int hw_ = p - n_ * a.hw_pad;
      pvalid = tile_ok && p < a.npix && hw_ < HWo;
      const int n = pvalid ? n_ : 0;
      const int hw = pvalid ? hw_ : 0;
      const int oy = hw / a.wout;
      const int ox = hw - oy * a.wout;
      iy0 = oy * a.stride - a.pad;
      ix0 = ox * a.stride - a.pad;
      xptr = a.x + ((long)n * a.in_ctot + a.in_coff) * HWi + (long)iy0 * a.win + ix0;
      lane_base = (int)((((long)n * a.in_ctot + a.in_coff + (VEC ? 4 * (lane >> 5) : 0)) * HWi + (long)iy0 * a.win + ix0) * 4);
    };
    set_tile(0);
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(a.wp);   // [K/8][2][cout] chunks of 8 bf16
    const int KGtot = (a.K + 7) / 8;

    // Two static register sets (tile t lives in set t & 1): the gathers of tile t+3 are issued right after tile t+1 has
    // been converted, i.e. a whole K-tile of MFMA time before they are needed — with a single set the loader was a
    // serial load -> wait(full memory latency) -> convert chain.
    float xr[NS][NITEM][8];
    // Branch-free gather: the per-lane base pointer is fixed, the tap offset is wave-uniform (scalar), and lanes whose
    // tap falls into the padding / past K / past the last pixel read the zero page instead (a pointer select, no
    // branch). Every lane therefore ALWAYS issues exactly 8*NITEM loads per K-tile — the counted vmcnt waits below
    // rely on that number — and all of them are in flight together.
    const unsigned x_bytes = (unsigned)((long)a.n * a.in_ctot * HWi * 4);   // < 2^31, checked on the host for TAP
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, TAP ? x_bytes : 0u, 0x00020000);
    auto gload = [&](float (&xr)[NITEM][8], int kt) {
      if constexpr (VEC) {
        // lane = (4-pixel group g = lane % 32, channel half h = lane / 32) of k-group `wave`: channels ci0 + 8*wave + 4*h + j,
        // j = 0..3, four consecutive pixels each (16 bytes; 4-byte aligned is all a buffer load needs). xr holds [channel j][pixel].
        const int T = (kbeg >> 5) + kt;                     // wave-uniform K-tile index = 32-channel block
        const bool ok = pvalid && kt < nk;
        const unsigned voff = ok ? (unsigned)lane_base : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int soff = __builtin_amdgcn_readfirstlane((T * 32 + wave * 8 + j) * HWi * 4);
          // (element access on the builtin's result converted to an ext_vector type is miscompiled by this hipcc: bit_cast first)
          const f32x4v v4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
#pragma unroll
          for (int q = 0; q < 4; ++q) xr[(j * 4 + q) >> 3][(j * 4 + q) & 7] = v4[q];
        }
        return;
      }
      if constexpr (TAP) {
        // K order (shared with cl4_pack_conv_weight_bf16x3): k = (cb*KHW + tap)*32 + cil — all KHW taps of a 32-channel
        // block are adjacent K-tiles, so the (shifted) re-reads of those channels hit L1/L2 instead of streaming the whole
        // input once per tap (FETCH_SIZE showed ~4x the algorithmic bytes with the plain tap-major order).
        const int T = (kbeg >> 5) + kt;                     // wave-uniform K-tile index
        const int cb = T / KHW;
        const int tap = T - cb * KHW;
        const int ci0 = cb * 32;
        const int ky = tap / KS;
        const int dy = ky * a.dil, dx = (tap - ky * KS) * a.dil;
        const int iy = iy0 + dy, ix = ix0 + dx;
        const bool ok = pvalid && kt < nk && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
        const unsigned voff = ok ? (unsigned)(lane_base + (dy * a.win + dx) * 4) : 0x80000000u;
#pragma unroll
        for (int it = 0; it < NITEM; ++it)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            // readfirstlane: the value is wave-uniform by construction, but unless the compiler can prove it, it keeps
            // the offset in a VGPR and wraps every load in a waterfall loop (~14 instructions per load)
            const int soff = __builtin_amdgcn_readfirstlane((ci0 + (kg0 + it) * 8 + j) * HWi * 4);
            xr[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
          }
        return;
      }
#pragma unroll
      for (int it = 0; it < NITEM; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kbeg + kt * BK2 + (kg0 + it) * 8 + j;   // wave-uniform
          const bool kval = k < kend;
          const int kc = kval ? k : 0;
          int ci, dy, dx;
          if (KS == 1) { ci = kc; dy = 0; dx = 0; }
          else {
            ci = kc / KHW;
            int t = kc - ci * KHW;
            int ky = t / KS;
            dy = ky * a.dil;
            dx = (t - ky * KS) * a.dil;
          }
          bool ok = kval && pvalid;
          if (KS != 1 || a.pad != 0) {
            const int iy = iy0 + dy, ix = ix0 + dx;
            ok = ok && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
          }
          const float* g = ok ? xptr + ((long)ci * HWi + (long)dy * a.win + dx) : zero_page;
          xr[it][j] = *g;
        }
      }
    };
    auto cvt_store = [&](const float (&xr)[NITEM][8], int buf) {
      u32x4* Bs = lds[buf] + A_CHUNKS;
      if constexpr (VEC) {
        // pixel q of the group lands in column q*32 + g; this lane owns channels 4h..4h+3 of the k-group = bytes 8h..8h+7 of the
        // 16-byte chunk. Consecutive lanes write consecutive chunks: conflict free.
        const int g = lane & 31, h = lane >> 5;
        unsigned long long* B8 = reinterpret_cast<unsigned long long*>(Bs);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = xr[(j * 4 + q) >> 3][(j * 4 + q) & 7];
            hi[j] = (__bf16)v;
            lo[j] = (__bf16)(v - (float)hi[j]);
          }
          B8[(((wave * P + 0) * BN + q * 32 + g) << 1) + h] = __builtin_bit_cast(unsigned long long, hi);
          B8[(((wave * P + 1) * BN + q * 32 + g) << 1) + h] = __builtin_bit_cast(unsigned long long, lo);
        }
        return;
      }
#pragma unroll
      for (int it = 0; it < NITEM; ++it) {
        bf16x8 hi, mid, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          hi[j] = (__bf16)xr[it][j];
          const float r1 = xr[it][j] - (float)hi[j];
          if (P == 3) {
            mid[j] = (__bf16)r1;
            lo[j] = (__bf16)(r1 - (float)mid[j]);
          } else {
            lo[j] = (__bf16)r1;
          }
        }
        Bs[((kg0 + it) * P + 0) * BN + pl] = __builtin_bit_cast(u32x4, hi);
        if (P == 3) Bs[((kg0 + it) * P + 1) * BN + pl] = __builtin_bit_cast(u32x4, mid);
        Bs[((kg0 + it) * P + P - 1) * BN + pl] = __builtin_bit_cast(u32x4, lo);
      }
    };
    // Weight tile through registers as well (buffer_load_dwordx4, A_PER_LANE 16-byte chunks per lane and K-tile): with
    // LDS-DMA the tile had to be issued and awaited inside the same iteration (its landing zone is the LDS stage itself),
    // which exposed one full L2 round trip per K-tile; two register sets keep two tiles of BOTH operands in flight.
    constexpr int A_PER_LANE = A_INSTR;
    const unsigned w_bytes = (unsigned)((long)KGtot * P * a.cout * 16);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, w_bytes, 0x00020000);
    u32x4 ar[NS][A_PER_LANE];
    auto loadA = [&](u32x4 (&ar)[A_PER_LANE], int kt) {
#pragma unroll
      for (int i = 0; i < A_PER_LANE; ++i) {
        const int ii = wave * A_PER_LANE + i;            // wave-uniform chunk row: (kg, plane, 64-wide m segment)
        const int seg = ii % (BM / 64);
        const int kp = ii / (BM / 64);                   // kg*P + plane
        const int m = m0 + seg * 64 + lane;
        const int kgg = (kbeg + kt * BK2) / 8 + kp / P;
        const bool kok = tile_ok && kgg < KGtot && (kgg * 8) < kend;             // scalar
        const unsigned voff = (kok && m < a.cout) ? (unsigned)m * 16u : 0x80000000u;
        const int soff = __builtin_amdgcn_readfirstlane(kok ? (kgg * P + kp % P) * a.cout * 16 : 0);
        ar[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, soff, 0);
      }
    };
    auto storeA = [&](const u32x4 (&ar)[A_PER_LANE], int buf) {
      u32x4* As = lds[buf];
#pragma unroll
      for (int i = 0; i < A_PER_LANE; ++i) {
        const int ii = wave * A_PER_LANE + i;
        As[(ii / (BM / 64)) * BM + (ii % (BM / 64)) * 64 + lane] = ar[i];
      }
    };
    constexpr int TILE_LOADS = (VEC ? 4 : 8 * NITEM) + A_PER_LANE;   // vector-memory instructions per lane and K-tile
    constexpr int PF_PER_LANE = (BM * (BN / 32) + 64 * LW - 1) / (64 * LW);     // residual-prefetch touches per loader lane
    // One loader half-iteration while the MFMA waves consume tile t: retire the loads of tile t+1 (issued two
    // half-iterations ago; only tile t+2's stay in flight), split / store it into the other LDS stage, refill the register
    // set with tile t+3. The sequence is deliberately free of conditionals: tiles past the end of K are "loaded" through
    // out-of-range buffer offsets (no memory traffic, zeros) and stored into the idle LDS stage. With any data-dependent
    // skip in the loop the compiler's own s_waitcnt insertion must assume the shortest path at every join and drains
    // the queue to vmcnt(0) before each conversion — which serialised the loader on one memory round trip per K-tile.
    // Issue cursor: the next K-tile to load is (ti_i, kt_i). Non-persistent: kt_i simply runs past nk (loads out of range).
    int ti_i = 0, kt_i = 0;
    float pf[PF_PER_LANE];
    auto prefetch_residual = [&]() {
      // Pull the residual tile of the fused epilogue towards L2: the MFMA waves read it in four dependent batches after the
      // K loop, and with the lines still in HBM each batch paid a full memory round trip (the epilogue of a K = 256 layer took
      // as long as its K loop). One dword per 32-pixel row segment touches every line of the tile but the last partial one.
      // The loads are younger than the sets in flight, so the counted waits only become (harmlessly) stricter; the destination
      // registers stay live until the loop is over.
      if (!SPLIT && (a.post || a.pre) && tile_ok) {
        const float* r = a.post ? a.post : a.pre;
#pragma unroll
        for (int i = 0; i < PF_PER_LANE; ++i) {
          const int item = (wave * 64 + lane) * PF_PER_LANE + i;   // (row, segment)
          const int co = min(m0 + min(item / (BN / 32), BM - 1), a.cout - 1);
          const int pp = min(n0 + (item % (BN / 32)) * 32, a.npix - 1);
          const int pn = min(pp / a.hw_pad, a.n - 1);
          const float* g = r + ((long)pn * a.cout + co) * HWo + min(pp - pn * a.hw_pad, HWo - 1);
          asm volatile("global_load_dword %0, %1, off" : "=v"(pf[i]) : "v"(g) : "memory");
        }
      }
    };
    auto issue = [&](float (&cur)[NITEM][8], u32x4 (&acur)[A_PER_LANE]) {
      if (PERS && kt_i == nk) {                // uniform: the cursor enters the next tile
        kt_i = 0;
        ++ti_i;
        set_tile(ti_i);
        prefetch_residual();
      }
      gload(cur, kt_i);
      loadA(acur, kt_i);
      ++kt_i;
    };
#if CL4_TL2
    unsigned long long tl2[5] = {0, 0, 0, 0, 0};   // wait for loads, convert + LDS stores, issue of the next loads, LDS drain, barrier
#define CL4_T(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tl2[i] += now_ - tprev_; tprev_ = now_; }
#else
#define CL4_T(i)
#endif
    auto half = [&](float (&cur)[NITEM][8], u32x4 (&acur)[A_PER_LANE], int t) {
      const int nb = (t + 1) & 1;
#if CL4_TL2
      unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
      wait_vmcnt<(NS - 1) * TILE_LOADS>();
      CL4_T(0)
#if !(CL4_DBG & 2)
      cvt_store(cur, nb);
      storeA(acur, nb);
#endif
      CL4_T(1)
#if !(CL4_DBG & 1)
      issue(cur, acur);
#else
      ++kt_i;
#endif
      CL4_T(2)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CL4_T(3)
      __builtin_amdgcn_s_barrier();
      CL4_T(4)
    };

#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) issue(xr[s_], ar[s_]);
    wait_vmcnt<(NS - 1) * TILE_LOADS>();
    cvt_store(xr[0], 0);
    storeA(ar[0], 0);
    issue(xr[0], ar[0]);
    if (ti_i == 0) prefetch_residual();        // first tile (later tiles: when the issue cursor reaches them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < TOT; t += NS) {     // 1 + NS*ceil(TOT/NS) barriers; the MFMA waves pad theirs to the same count
#pragma unroll
      for (int j_ = 0; j_ < NS; ++j_) half(xr[(j_ + 1) % NS], ar[(j_ + 1) % NS], t + j_);   // tile t+j+1 lives in set (j+1) % NS
    }
    if (!SPLIT && (a.post || a.pre)) {
      wait_vmcnt<0>();
#pragma unroll
      for (int i = 0; i < PF_PER_LANE; ++i) asm volatile("" ::"v"(pf[i]));
    }
#if CL4_TL2
    if (a.timeline && wave8 == 4 && lane == 0) {
      unsigned long long* o = a.timeline + (size_t)gridDim.x * 8 + (size_t)id * 16;
      for (int i = 0; i < 5; ++i) o[i] = tl2[i];
      o[5] = (unsigned long long)TOT;
    }
#endif
    if constexpr (WIDE) wide_epilogue();
    return;
  }

  // ---- MFMA waves ---------------------------------------------------------------------------------------------
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __builtin_amdgcn_s_barrier();
  stamp(1);
  int gk = 0;                                  // K-tiles consumed so far (its parity is the LDS stage)
  for (int ti = 0; ti < (PERS ? cnt : 1); ++ti) {
  if (PERS && ti > 0) {
    tile_origin(ti);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // (Measured and dropped, r02: software-pipelining the fragment reads of the MFMA waves — every ds_read_b128 issued one product
  // group ahead of its MFMAs, rotated loop with exact counted lgkmcnt waits, 120 VGPRs — changed nothing: 125.5 -> 129.5 us on
  // the res4 conv3 shape, 185 -> 199 us on the 3x3 one. The MFMA waves are not the pacing role; they wait at the barrier.)
#if CL4_TL2
  unsigned long long tm2[2] = {0, 0};          // fragment reads + MFMAs, barrier
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#define CL4_TM(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tm2[i] += now_ - tprev_; tprev_ = now_; }
#else
#define CL4_TM(i)
#endif
  for (int kt = 0; kt < nk; ++kt, ++gk) {
    const u32x4* As = lds[gk & 1] + wm * WTM + lr;
    const u32x4* Bs = lds[gk & 1] + A_CHUNKS + wn * WTN + lr;
#pragma unroll
    for (int s2 = 0; s2 < KG / 2; ++s2) {
      const int kg = 2 * s2 + lh;
      bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 0) * BM + i * 32]);
        if (P == 3) am[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 1) * BM + i * 32]);
        al[i] = __builtin_bit_cast(bf16x8, As[(kg * P + P - 1) * BM + i * 32]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 0) * BN + j * 32]);
        if (P == 3) bm[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 1) * BN + j * 32]);
        bl[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + P - 1) * BN + j * 32]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#if CL4_DBG & 4
          acc[i][j][0] += (float)al[i][0] * (float)bh[j][0] + (float)ah[i][1] * (float)bl[j][1];
#else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          if (P == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#endif
        }
    }
    // all of this wave's LDS reads of the stage have returned before the loaders are told it may be overwritten
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CL4_TM(0)
    __builtin_amdgcn_s_barrier();
    CL4_TM(1)
  }
#if CL4_TL2
  if (a.timeline && wave8 == 0 && lane == 0) {
    unsigned long long* o = a.timeline + (size_t)gridDim.x * 8 + (size_t)id * 16 + 8;
    o[0] = tm2[0]; o[1] = tm2[1];
  }
#endif
  // persistent: drain this tile while the loaders (blocked at their next barrier, their register sets already filled with
  // the next tile's first K-tiles) wait
  if (PERS) fused_epilogue<TM, TN>(a, acc, m0 + wm * WTM, n0 + wn * WTN, lr, lh, HWo);
  }
  for (int e_ = TOT % NS; e_ != 0 && e_ < NS; ++e_) __builtin_amdgcn_s_barrier();   // pairs with the loaders' padded half-iterations
  stamp(2);
  if (PERS) return;

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int p = n0 + wn * WTN + j * 32 + lr;
    if (!SPLIT || p >= a.npix) continue;
    {
      float* wsp = a.ws + ((long)bz * a.cout) * a.ws_npix + (p - a.ws_p0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < a.cout) wsp[(long)co * a.ws_npix] = acc[i][j][r];
        }
    }
  }
  if constexpr (WIDE) {
    float* stage = reinterpret_cast<float*>(&lds[0][0]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WTN + j * 32 + lr] = acc[i][j][r];
    wide_epilogue();
    return;
  }
  if (!SPLIT) fused_epilogue<TM, TN>(a, acc, m0 + wm * WTM, n0 + wn * WTN, lr, lh, HWo);
}

// w [cout][K] fp32 -> wq [ceil(K/8)][plane: hi, (mid,) lo][cout][8] bf16 (zero padded along K); planes = 2 or 3.
__global__ void pack_weight_bf16x3_kernel(const float* __restrict__ w, bf16x8* __restrict__ wq, int cout, int K, int cin,
                                          int khw, int tap_major, int planes) {
  const int KGtot = (K + 7) / 8;
  const long total = (long)KGtot * cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = i % cout;
    const int kg = i / cout;
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kg * 8 + j;                      // position in the kernel's K order
      int ksrc = k;                                  // position in PyTorch's [cin][kh*kw] order
      if (tap_major && k < K) {                      // k = (cb*khw + tap)*32 + cil  ->  ci = cb*32 + cil
        const int t32 = k >> 5, cil = k & 31;
        const int cb = t32 / khw, tap = t32 - cb * khw;
        ksrc = (cb * 32 + cil) * khw + tap;
      }
      const float v = k < K ? w[(long)m * K + ksrc] : 0.f;
      hi[j] = (__bf16)v;
      const float r1 = v - (float)hi[j];
      mid[j] = (__bf16)r1;
      lo[j] = planes == 3 ? (__bf16)(r1 - (float)mid[j]) : mid[j];
    }
    wq[((long)kg * planes + 0) * cout + m] = hi;
    if (planes == 3) wq[((long)kg * planes + 1) * cout + m] = mid;
    wq[((long)kg * planes + planes - 1) * cout + m] = lo;
  }
}

// ------------------------------------------------------------------------------------
// Weight gradient: dW[co][j] += sum_p dY[co][p] * Xg[p][j],  j = (ci, ky, kx).
// GEMM with M = cout, N = cin*kh*kw, K = pixels. Both operands are pixel-contiguous in
// NCHW, so loads run along K; the LDS tiles are stored [row][BKP+1] (the +1 pad makes the
// k-strided MFMA operand reads conflict free). Split-K over pixel chunks (grid.z) with
// fp32 atomics: each atomic wave-instruction adds 32 consecutive floats of one dW row.
// ------------------------------------------------------------------------------------
constexpr int BKP = 32;  // pixel (K) tile of the wgrad kernel

struct WgradArgs {
  const float* x;
  const float* dy;
  float* dw;
  int n, cin, hin, win, cout, stride, pad, dil, hout, wout;
  int in_ctot, in_coff, out_ctot, out_coff;
  int Nj;       // cin*KS*KS
  int npix;     // n*hout*wout
  int chunk;    // pixels per grid.z slice (multiple of BKP)
  int precision;  // 0: exact fp32 MFMA, 1: bf16x3 split products
  int hw_pad;   // VEC kernels: hout*wout rounded up to a multiple of 4; their K axis is the pixel space padded per image, n*hw_pad long
};

template <int BM, int BN, int KS>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_f32(WgradArgs a) {
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KHW = KS * KS;
  constexpr int RA = BM * BKP / NTHREADS;  // elements per thread of the A tile
  constexpr int RB = BN * BKP / NTHREADS;
  constexpr int LD = BKP + 1;

  __shared__ float As[2][BM * LD];
  __shared__ float Bs[2][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int j0 = blockIdx.y * BN;
  const int pbeg = blockIdx.z * a.chunk;
  const int pend = min(a.npix, pbeg + a.chunk);
  if (pbeg >= pend) return;

  const int HWo = a.hout * a.wout;
  const int HWi = a.hin * a.win;
  const int pl = lane & 31;      // pixel within the K tile
  const int rsub = lane >> 5;    // which of the two rows this half-wave stages

  // The B rows (ci,ky,kx) this thread stages are fixed for the whole kernel: decode once.
  long b_coff[RB];
  int b_dy[RB], b_dx[RB];
  bool b_ok[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int row = wave * (BN / 4) + 2 * r + rsub;
    int j = j0 + row;
    b_ok[r] = j < a.Nj;
    int jc = b_ok[r] ? j : 0;
    int ci = jc / KHW;
    int t = jc - ci * KHW;
    int ky = t / KS;
    b_dy[r] = ky * a.dil;
    b_dx[r] = (t - ky * KS) * a.dil;
    b_coff[r] = (long)(a.in_coff + ci) * HWi;
  }

  float ra[RA], rb[RB];
  auto load_tile = [&](int kt) {
    int p = pbeg + kt * BKP + pl;
    bool pv = p < pend;
    int pc = pv ? p : 0;
    int n = pc / HWo;
    int hw = pc - n * HWo;
    int oy = hw / a.wout;
    int ox = hw - oy * a.wout;
    int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    const long dybase = ((long)n * a.out_ctot + a.out_coff) * HWo + hw;
    const long xb = (long)n * a.in_ctot * HWi;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      int row = wave * (BM / 4) + 2 * r + rsub;
      int co = m0 + row;
      ra[r] = (pv && co < a.cout) ? a.dy[dybase + (long)co * HWo] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      int iy = iy0 + b_dy[r], ix = ix0 + b_dx[r];
      bool ok = pv && b_ok[r] && (unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win;
      rb[r] = ok ? a.x[xb + b_coff[r] + (long)iy * a.win + ix] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int r = 0; r < RA; ++r) As[buf][(wave * (BM / 4) + 2 * r + rsub) * LD + pl] = ra[r];
#pragma unroll
    for (int r = 0; r < RB; ++r) Bs[buf][(wave * (BN / 4) + 2 * r + rsub) * LD + pl] = rb[r];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (pend - pbeg + BKP - 1) / BKP;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int lr = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BKP; kk += 2) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[buf][(wm * WTM + i * 32 + lr) * LD + kk + lh];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][(wn * WTN + j * 32 + lr) * LD + kk + lh];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jj = j0 + wn * WTN + j * 32 + lr;
    if (jj >= a.Nj) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= a.cout) continue;
        atomicAdd(&a.dw[(long)co * a.Nj + jj], acc[i][j][r]);
      }
  }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int K) {
  // w [cout][K] -> wp [K][cout]; 32x32 LDS transpose so both sides stay coalesced.
  __shared__ float t[32][33];
  int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {
    int c = c0 + r, k = k0 + tx;
    t[r][tx] = (c < cout && k < K) ? w[(long)c * K + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int k = k0 + r, c = c0 + tx;
    if (k < K && c < cout) wp[(long)k * cout + c] = t[tx][r];
  }
}

__global__ void bn_fold_kernel(const float* g, const float* b, const float* m, const float* v, float eps,
                               float* scale, float* shift, int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  float s = (g ? g[i] : 1.f) / sqrtf(v[i] + eps);
  scale[i] = s;
  shift[i] = (b ? b[i] : 0.f) - m[i] * s;
}

int check_desc(const cl4_conv_desc* d) {
  CL4_REQUIRE(d != nullptr, "conv: null descriptor");
  CL4_REQUIRE(d->n > 0 && d->cin > 0 && d->cout > 0 && d->hin > 0 && d->win > 0, "conv: non-positive dims");
  CL4_REQUIRE(d->kh == d->kw && (d->kh == 1 || d->kh == 3 || d->kh == 7), "conv: kernel %dx%d unsupported (1,3,7)", d->kh, d->kw);
  CL4_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "conv: bad stride/dil/pad");
  int ho = (d->hin + 2 * d->pad - d->dil * (d->kh - 1) - 1) / d->stride + 1;
  int wo = (d->win + 2 * d->pad - d->dil * (d->kw - 1) - 1) / d->stride + 1;
  CL4_REQUIRE(ho == d->hout && wo == d->wout, "conv: hout/wout %dx%d do not match geometry %dx%d", d->hout, d->wout, ho, wo);
  CL4_REQUIRE(d->in_coff >= 0 && d->in_coff + d->cin <= d->in_ctot, "conv: input channel slice out of range");
  CL4_REQUIRE(d->out_coff >= 0 && d->out_coff + d->cout <= d->out_ctot, "conv: output channel slice out of range");
  CL4_REQUIRE((long)d->n * d->hout * d->wout < 2147483647L, "conv: pixel count overflows int32");
  CL4_REQUIRE((long)d->cin * d->kh * d->kw * (long)d->cout < 2147483647L, "conv: weight count overflows int32");
  return CL4_OK;
}

// 16 zero floats the LDS-DMA lanes of the v2 kernel read instead of the convolution's zero padding.
__device__ __attribute__((aligned(64))) float g_zero_page[16] = {0};

unsigned long long* g_timeline = nullptr;     // experiments only, see ConvArgs::timeline

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Tile / split-K plan of one forward launch. The fp32 MFMA pipe is slow enough (64 cycles per 32x32x2) that the
// kernel is latency- rather than bandwidth-limited: what matters is having >= ~3 workgroups per CU in flight.
// 128x128 tiles are used when the grid is large enough on its own; otherwise 64x64 tiles (4x the workgroups),
// and if that is still short (the 33x33 maps at small batch) K is split across grid.z.
struct FwdPlan {
  int impl;    // 1: register-staged kernel, 2: LDS-DMA kernel
  int tile;    // 128 or 64 (v2)
  int splits;
  int kchunk;
  int tail;    // bf16 kernels: > 1 = "main + split tail": whole rounds of tiles run unsplit with the fused epilogue, only the
               // last partial round is split `tail` ways (kchunk is then the tail's K range); 1 = "main + small-tile tail": the
               // last partial round of a 128x128 plan runs as 64x64 tiles (4x the workgroups, a quarter of the work each — for
               // short-K layers where splitting K further is pointless); 0 = plain
  int bn = 0;  // pixels per tile where they differ from `tile` (planes-input kernel: 144, see conv_s.hip); 0 = same as `tile`
  int pix() const { return bn ? bn : tile; }
};

// Measured plans (cl4_conv2d_fwd_set_plan, filled by the host-side autotuner during warm-up) override the heuristic.
struct PlanKey {
  int cout, npix, K, ks, stride, dil, hin;
  bool operator<(const PlanKey& o) const {
    return std::tie(cout, npix, K, ks, stride, dil, hin) < std::tie(o.cout, o.npix, o.K, o.ks, o.stride, o.dil, o.hin);
  }
};
std::map<PlanKey, FwdPlan>& plan_table() { static std::map<PlanKey, FwdPlan> t; return t; }
std::mutex& plan_mutex() { static std::mutex m; return m; }

// precision codes of the planes-input launches: 3 = two planes (bf16x3), 4 = three planes (bf16x6, 144-pixel tiles only),
// 5 = two fp16 planes (fp16x3, r05: the LDS-DMA kernel only — plan impl 4 / 6, unsplit, fp32 output)
static inline bool planes_code(int precision) { return precision >= 3 && precision <= 5; }

FwdPlan make_plan(int impl, int tile, int splits, int cout, int K, int precision = 0) {
  FwdPlan p{1, 128, 1, 0, 0};
  if (impl == 3 && precision == 3) {                                         // persistent pipelined planes kernel (conv_p.hip): unsplit 128 x 128 tiles
    p.impl = 3; p.tile = 128; p.splits = 1; p.kchunk = cl4::cdiv(K, BK2) * BK2;
    return p;
  }
  if ((impl >= 4 && impl <= 6 && precision == 3) || (impl == 6 && precision == 4) || ((impl == 4 || impl == 6) && precision == 5)) {                            // LDS-DMA planes kernel (conv_b.hip): 256 x 144 (4) or 128 x 144 tiles (5: four waves, 6: eight)
    p.impl = impl; p.tile = 128; p.bn = 144; p.splits = 1; p.kchunk = cl4::cdiv(K, BK2) * BK2;
    if (impl == 6 && precision == 3 && splits > 1) {       // r05: K-split slabs on the eight-wave tile (two planes): launches that cannot fill the chip
      const int ktiles = cl4::cdiv(K, BK2);
      const int sp = splits > ktiles ? ktiles : splits;
      p.kchunk = cl4::cdiv(ktiles, sp) * BK2;
      p.splits = cl4::cdiv(K, p.kchunk);
    }
    return p;
  }
  if (impl != 2 || cout % 4 != 0 || cout < 64) return p;
  p.impl = 2;
  p.tile = ((tile == 128 && cout >= 128) || planes_code(precision)) ? 128 : 64;      // planes-input kernel: 128 output channels per tile
  if ((precision == 3 && tile == 144) || precision == 4) p.bn = 144;                   // ... x 128 or 144 pixels (three planes: 144 only)
  else if (precision == 3 && tile == 64) p.bn = 64;                                    // ... or 64 pixels (three workgroups per CU: short-K, HBM-bound layers)
  const int ktiles = cl4::cdiv(K, BK2);
  int s = splits < 0 ? -splits : (splits < 1 ? 1 : splits);
  if (s > ktiles) s = ktiles;
  const int per = cl4::cdiv(ktiles, s);
  p.kchunk = per * BK2;
  const int actual = cl4::cdiv(K, p.kchunk);
  if (splits == -1 && precision != 0 && !planes_code(precision) && p.tile == 128) { p.tail = 1; p.splits = 1; }     // main + SMALL-TILE tail (64x64, unsplit)
  else if (splits < 0 && precision != 0 && actual > 1) { p.tail = actual; p.splits = 1; }
  else p.splits = actual;
  return p;
}

// Heuristic default. The fp32 MFMA pipe is slow (64 cycles per 32x32x2), so these kernels are limited by workgroup
// quantisation, not bandwidth: W workgroups dealt over 256 CUs finish in ceil(W/256) rounds. Pick the tile (128 or 64)
// and the K split that minimise  rounds * (k-tiles per workgroup + fixed cost) * tile area  + slab traffic.
FwdPlan plan_fwd(const PlanKey& key) {
  static const int impl = env_int("CL4_CONV_IMPL", 2);
  static const int force_tile = env_int("CL4_CONV_TILE", 0);
  static const int force_split = env_int("CL4_CONV_SPLIT", 0);
  {
    std::lock_guard<std::mutex> g(plan_mutex());
    auto it = plan_table().find(key);
    if (it != plan_table().end()) return it->second;
  }
  const int cout = key.cout, npix = key.npix, K = key.K, precision = key.hin % 8;
  if (impl != 2 || cout % 4 != 0 || cout < 64) return FwdPlan{1, 128, 1, 0, 0};
  if (force_tile || force_split) return make_plan(2, force_tile ? force_tile : 128, force_split ? force_split : 1, cout, K, precision);
  const int ktiles = cl4::cdiv(K, BK2);
  double best = 1e30;
  FwdPlan bestp = make_plan(2, 64, 1, cout, K, precision);
  for (int tile = 64; tile <= 128; tile += 64) {
    if (planes_code(precision) && tile != 128) continue;
    if (tile == 128 && !planes_code(precision) && (cout < 128 || (precision == 2 && ktiles < 64))) continue;   // bf16x6 at 128x128: 96 KB LDS, one WG per CU — long K only
    const long wgs = (long)cl4::cdiv(cout, tile) * cl4::cdiv(npix, tile);
    const double t_tile = tile == 128 ? 2.4 : 0.65;   // us per K-tile of one workgroup at ~70 % MFMA issue
    const double t_fixed = tile == 128 ? 5.0 : 2.5;   // prologue + epilogue
    for (int s = 1; s <= 32 && s <= ktiles / 2 + (ktiles < 2); ++s) {
      const FwdPlan p = make_plan(2, tile, s, cout, K, precision);
      if (p.splits != s && s != 1) continue;
      const long rounds = (wgs * p.splits + 255) / 256;
      double t = rounds * ((double)(p.kchunk / BK2) * t_tile + t_fixed);
      if (p.splits > 1) t += (double)p.splits * cout * npix * 8.0 / 4.0e6 + 3.0;   // slab write+read at ~4 TB/s, + launch
      if (t < best) { best = t; bestp = p; }
    }
  }
  return bestp;
}

// "Main + split tail" geometry of a bf16 plan: the first `main` logical tiles (whole rounds of the resident-workgroup
// slots, a multiple of the M-tile count so that the tail starts on a pixel-tile boundary) run unsplit; returns main (0 =
// the plan degenerates to a plain launch) and the first tail pixel.
struct TailGeom { int gx, ntiles, main, p0; };
TailGeom tail_geometry(const FwdPlan& pl, int cout, int npix, int precision) {
  TailGeom g;
  g.gx = cl4::cdiv(cout, pl.tile);
  g.ntiles = g.gx * cl4::cdiv(npix, pl.pix());
  g.main = 0;
  g.p0 = 0;
  if (pl.tail > 0) {
    int slots = ((precision == 2 && pl.tile == 128) || precision == 4) ? 256 : (pl.bn == 64 ? 768 : 512);     // resident workgroups: one / two / three per CU
    if (g.ntiles < slots + slots / 16) slots = 256;
    int m = (g.ntiles / slots) * slots;
    m -= m % g.gx;
    if (m > 0 && m < g.ntiles) { g.main = m; g.p0 = (m / g.gx) * pl.pix(); }
  }
  return g;
}

// Does this launch use the 4-pixels-per-lane activation loader (VEC)? 1x1, stride 1, same input and output map, 32-channel
// K-tiles, bf16x3, 128-pixel tiles. CL4_CONV_VEC=0 switches it off (experiments).
bool vec_gather_plan(const ConvArgs& a, int ks, int precision, int tile) {
  static const int enabled = env_int("CL4_CONV_VEC", 1);
  return enabled && ks == 1 && precision == 1 && tile == 128 && a.stride == 1 && a.pad == 0 && a.cin % 32 == 0 && a.hin == a.hout &&
         a.win == a.wout && (long)a.n * a.in_ctot * a.hin * a.win * 4 < 0x7FFFFFFFL;
}

// Split-K second pass of the planes-input kernel: like conv_splitk_epilogue, but one thread owns 8 consecutive output channels of one
// pixel, so that it can emit the planes chunks (a.ys) and read a planes residual (a.post_s); fp32 rows (a.y) are written as well when
// requested. grid = (pixel chunks of 256, ceil(cout / 8)).
__global__ void conv_splitk_epilogue_s(ConvArgs a, int splits) {
  const int co0 = blockIdx.y * 8;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ws_npix) return;
  const long total = (long)a.cout * a.ws_npix;
  float v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float* w = a.ws + (long)min(co0 + r, a.cout - 1) * a.ws_npix + i;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += w[(long)z * total];     // fixed order: reproducible
    v[r] = s;
  }
  const int HWo = a.hout * a.wout;
  const int p = a.ws_p0 + i;
  const int n = p / HWo;
  const int hw = p - n * HWo;
  const long rbase = (long)n * a.cout * HWo + hw;
  float t[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int cc = min(co0 + r, a.cout - 1);
    if (a.pre) v[r] += a.pre[rbase + (long)cc * HWo];
    const float u = v[r] * (a.scale ? a.scale[cc] : 1.f) + (a.shift ? a.shift[cc] : 0.f);
    v[r] = u > 0.f ? u : u * a.neg1;
    t[r] = a.post ? a.post[rbase + (long)cc * HWo] : 0.f;
  }
  if (a.post_s) {
    const u32x4* ps = reinterpret_cast<const u32x4*>(a.post_s) + (((long)n * (a.cout >> 3) + (co0 >> 3)) * 2) * HWo + hw;
    const u32x4 hi = ps[0], lo = ps[HWo];
#pragma unroll
    for (int q = 0; q < 4; ++q) plfmt::join2(hi[q], lo[q], t[2 * q], t[2 * q + 1]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    v[r] += t[r];
    v[r] = v[r] > 0.f ? v[r] : v[r] * a.neg2;
  }
  if (a.y) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (co0 + r < a.cout) a.y[((long)n * a.out_ctot + a.out_coff + co0 + r) * HWo + hw] = v[r];
  }
  if (a.ys) {
    u32x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // the same split as conv_s.hip::split8 (planes_fmt.h)
      unsigned h_, l_;
      plfmt::split2(v[2 * q], v[2 * q + 1], h_, l_);
      hi[q] = h_;
      lo[q] = l_;
    }
    u32x4* ys = reinterpret_cast<u32x4*>(a.ys) + (((long)n * (a.out_ctot >> 3) + ((a.out_coff + co0) >> 3)) * 2) * HWo + hw;
    ys[0] = hi;
    ys[HWo] = lo;
  }
}

// Planes-input launch (precision code 3, conv_s.hip): same plan machinery as the bf16 launches — unsplit, K-split, or whole rounds
// unsplit + split tail — on 128x128 tiles only.
template <int KS>
int launch_planes(ConvArgs& a, hipStream_t st, void* workspace, size_t workspace_bytes, int pcode) {
  if constexpr (KS == 7) {
    cl4::set_error("conv2d_fwd_planes: 7x7 kernels are not built for planes input");
    return CL4_EINVAL;
  } else {
    FwdPlan pl = plan_fwd(PlanKey{a.cout, a.npix, a.K, KS, a.stride, a.dil, a.hin * 8 + pcode});
    if (pcode == 5) {
      // fp16x3: the LDS-DMA kernel, unsplit. Without a pinned plan: 256-row tiles when they fill the chip, else eight waves on 128 rows
      if (pl.impl != 4 && pl.impl != 6) pl = make_plan((a.cout % 256 == 0 && (a.cout / 256) * cl4::cdiv(a.npix, 144) >= 200) ? 4 : 6, 144, 1, a.cout, a.K, 5);
      const int rows = pl.impl == 4 ? 256 : 128;
      if (!cl4::conv_b_eligible(a, KS, rows)) { cl4::set_error("conv2d_fwd_planes_f16: shape not eligible for the LDS-DMA kernel (cin %% 32, cout %% 8, cout >= %d)", rows); return CL4_EINVAL; }
      a.gx = cl4::cdiv(a.cout, rows);
      a.ntile_l = a.gx * cl4::cdiv(a.npix, 144);
      a.tile_base = 0; a.ws = nullptr; a.ws_p0 = 0; a.ws_npix = a.npix; a.kchunk = pl.kchunk;
      return cl4::launch_conv_b(a, KS, rows, 8, 2, (unsigned)a.ntile_l, st, true);
    }
    if (pl.impl < 2 || pl.impl > 6 || pl.tile != 128) { cl4::set_error("conv2d_fwd_planes: no 128x128 plan for this shape"); return CL4_EINVAL; }
    if (pl.impl == 3) {
      if (pcode == 3 && !a.stats && cl4::conv_p_eligible(a, KS)) {
        a.gx = cl4::cdiv(a.cout, 128);
        a.ntile_l = a.gx * cl4::cdiv(a.npix, 128);
        a.tile_base = 0; a.ws = nullptr; a.ws_p0 = 0; a.ws_npix = a.npix; a.kchunk = pl.kchunk;
        return cl4::launch_conv_p(a, (unsigned)a.ntile_l, st);
      }
      // a plan pinned for this shape by a call with other operands (fp32 output, fp32 residual, ...): the plain kernel, unsplit
    }
    if (pl.impl >= 4 && pl.impl <= 6 && (pcode == 3 || pl.impl == 6) && cl4::conv_b_eligible(a, KS, pl.impl == 4 ? 256 : 128)) {
      const int rows = pl.impl == 4 ? 256 : 128;
      a.gx = cl4::cdiv(a.cout, rows);
      a.ntile_l = a.gx * cl4::cdiv(a.npix, 144);
      a.tile_base = 0; a.ws = nullptr; a.ws_p0 = 0; a.ws_npix = a.npix; a.kchunk = pl.kchunk;
      if (pl.impl == 6 && pcode == 3 && pl.splits > 1) {    // K-split slabs + the planes split-K epilogue (as the conv_s plans below)
        const size_t need = (size_t)pl.splits * a.cout * a.ws_npix * sizeof(float);
        if (!workspace || workspace_bytes < need) { cl4::set_error("conv2d_fwd_planes: workspace too small for split-K"); return CL4_ENOSPC; }
        a.ws = (float*)workspace;
        const int rc = cl4::launch_conv_b(a, KS, 128, 8, 2, (unsigned)a.ntile_l * (unsigned)pl.splits, st, false, true);
        if (rc != CL4_OK) return rc;
        hipLaunchKernelGGL(conv_splitk_epilogue_s, dim3(cl4::cdiv(a.ws_npix, 256), cl4::cdiv(a.cout, 8)), dim3(256), 0, st, a, pl.splits);
        return cl4::launch_status();
      }
      return cl4::launch_conv_b(a, KS, rows, pl.impl == 5 ? 4 : 8, pcode == 4 ? 3 : 2, (unsigned)a.ntile_l, st);
    }
    const TailGeom tg = tail_geometry(pl, a.cout, a.npix, pcode);
    const bool hybrid = tg.main > 0 && pl.tail > 1;
    const int nsplit = hybrid ? pl.tail : pl.splits;
    a.gx = tg.gx;
    a.kchunk = pl.kchunk;
    a.ws = nullptr;
    a.ws_p0 = hybrid ? tg.p0 : 0;
    a.ws_npix = a.npix - a.ws_p0;
    if (nsplit > 1) {
      const size_t need = (size_t)nsplit * a.cout * a.ws_npix * sizeof(float);
      if (!workspace || workspace_bytes < need) { cl4::set_error("conv2d_fwd_planes: workspace too small for split-K"); return CL4_ENOSPC; }
      a.ws = (float*)workspace;
    }
    auto go = [&](int base, int ntile, int sp) {
      a.tile_base = base;
      a.ntile_l = ntile;
      return cl4::launch_conv_s(a, KS, sp > 1, (unsigned)ntile * (unsigned)(sp > 1 ? sp : 1), st, pl.pix(), pcode == 4 ? 3 : 2);
    };
    int rc;
    if (hybrid) {
      if ((rc = go(0, tg.main, 1)) != CL4_OK) return rc;
      if ((rc = go(tg.main, tg.ntiles - tg.main, nsplit)) != CL4_OK) return rc;
    } else if ((rc = go(0, tg.ntiles, nsplit)) != CL4_OK) return rc;
    if (nsplit > 1)
      hipLaunchKernelGGL(conv_splitk_epilogue_s, dim3(cl4::cdiv(a.ws_npix, 256), cl4::cdiv(a.cout, 8)), dim3(256), 0, st, a, nsplit);
    return cl4::launch_status();
  }
}

// Does this launch run on the unified-wave kernel (conv_u.hip)? bf16x3, 128x128 plan, 1x1 / 3x3 stride-1 "same" convolution with
// 32-channel K-tiles. Like the VEC plans it walks a pixel index space padded per image to a multiple of 4. CL4_CONV_U=0: off.
bool u_plan(const ConvArgs& a, int ks, int precision, int tile) {
  static const int enabled = env_int("CL4_CONV_U", 1);
  return enabled && precision == 1 && tile == 128 && (ks == 1 || ks == 3) && a.stride == 1 && a.pad == a.dil * (ks / 2) &&
         a.cin % 32 == 0 && a.hin == a.hout && a.win == a.wout && (long)a.n * a.in_ctot * a.hin * a.win * 4 < 0x7FFFFFFFL;
}

template <int KS>
int launch_fwd(ConvArgs& a, hipStream_t st, void* workspace, size_t workspace_bytes, int precision) {
  if (planes_code(precision)) return launch_planes<KS>(a, st, workspace, workspace_bytes, precision);
  const FwdPlan pl = plan_fwd(PlanKey{a.cout, a.npix, a.K, KS, a.stride, a.dil, a.hin * 8 + precision});
  if (precision == 1 || precision == 2) {
    static float* zp3 = nullptr;
    if (!zp3 && hipGetSymbolAddress((void**)&zp3, HIP_SYMBOL(g_zero_page)) != hipSuccess) return (int)hipGetLastError();
    const float* z = zp3;
    const bool tapmaj = a.cin % 32 == 0;   // must match cl4_pack_conv_weight_bf16x3 / _bf16x6
    if (tapmaj && (long)a.n * a.in_ctot * a.hin * a.win * 4 >= 0x7FFFFFFFL) {
      cl4::set_error("conv2d_fwd(bf16 split): input tensor >= 2 GiB is not addressable by the 32-bit buffer offsets");
      return CL4_EINVAL;
    }
    // "vector gather" plans (see the VEC template parameter): logical pixel index space padded per image to a multiple of 4
    const bool upl = u_plan(a, KS, precision, pl.tile);
    const bool vec = !upl && vec_gather_plan(a, KS, precision, pl.tile);
    if (vec || upl) {
      a.hw_pad = (a.hout * a.wout + 3) & ~3;
      a.npix = a.n * a.hw_pad;
    }
    const TailGeom tg = tail_geometry(pl, a.cout, a.npix, precision);
    const bool hybrid = tg.main > 0;
    static const int persist = env_int("CL4_CONV_PERSIST", 0);   // measured: a loss at 128x128 (epilogue spills), +5 % at 64x64
    const int nsplit = hybrid ? (pl.tail == 1 ? 1 : pl.tail) : pl.splits;
    a.gx = tg.gx;
    a.kchunk = pl.kchunk;
    a.ws = nullptr;
    a.ws_p0 = hybrid ? tg.p0 : 0;
    a.ws_npix = a.npix - a.ws_p0;
    if (nsplit > 1) {
      const size_t need = (size_t)nsplit * a.cout * a.ws_npix * sizeof(float);
      if (!workspace || workspace_bytes < need) { cl4::set_error("conv2d_fwd: workspace too small for split-K"); return CL4_ENOSPC; }
      a.ws = (float*)workspace;
    }
    // one launch of `ntile` logical tiles starting at `base`, unsplit (fused epilogue) or split `sp` ways (slabs)
    auto go = [&](int base, int ntile, int sp, int tile) {
      a.tile_base = base;
      a.ntile_l = ntile;
      a.gx = cl4::cdiv(a.cout, tile);
      const dim3 g((unsigned)ntile * (unsigned)(sp > 1 ? sp : 1));
      const bool split = sp > 1;
      if (upl && tile == 128) {
        if (cl4::launch_conv_u(a, KS, split, g.x, st) != CL4_OK) return;
      } else if (precision == 2) {
        if (tile == 128) {
          if (split) { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, true, true, 3, 8, 4>), g, dim3(768), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, true, false, 3, 8, 4>), g, dim3(768), 0, st, a, z); }
          else { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, true, 3, 8, 4>), g, dim3(768), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, false, 3, 8, 4>), g, dim3(768), 0, st, a, z); }
        } else {
          if (split) { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, true, true, 3>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, true, false, 3>), g, dim3(V2_THREADS), 0, st, a, z); }
          else { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, true, 3>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, false, 3>), g, dim3(V2_THREADS), 0, st, a, z); }
        }
      } else if (!split && persist && ntile > 512 && !vec) {
        // persistent workgroups: 512 resident slots (two per CU) walk ntile / 512 tiles each
        const dim3 gp(512);
        if (tile == 128) { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, true, 2, 4, 2, true>), gp, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, false, 2, 4, 2, true>), gp, dim3(V2_THREADS), 0, st, a, z); }
        else { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, true, 2, 4, 2, true>), gp, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, false, 2, 4, 2, true>), gp, dim3(V2_THREADS), 0, st, a, z); }
      } else {
        if (tile == 128) {
          if (split) { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, true, true>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, true, false>), g, dim3(V2_THREADS), 0, st, a, z); }
          else if (vec) { if constexpr (KS == 1) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, 1, false, true, 2, 4, 2, false, true>), g, dim3(V2_THREADS), 0, st, a, z); }
          else { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, true>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<128, 128, KS, false, false>), g, dim3(V2_THREADS), 0, st, a, z); }
        } else {
          if (split) { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, true, true>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, true, false>), g, dim3(V2_THREADS), 0, st, a, z); }
          else { if (tapmaj) hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, true>), g, dim3(V2_THREADS), 0, st, a, z); else hipLaunchKernelGGL((conv_igemm_bf16x3<64, 64, KS, false, false>), g, dim3(V2_THREADS), 0, st, a, z); }
        }
      }
    };
    if (hybrid && pl.tail == 1) {               // small-tile tail: the 64x64 tiles that cover the pixels from p0 on
      go(0, tg.main, 1, pl.tile);
      const int gx64 = cl4::cdiv(a.cout, 64);
      const int base64 = (tg.p0 / 64) * gx64;
      go(base64, gx64 * cl4::cdiv(a.npix, 64) - base64, 1, 64);
    } else if (hybrid) {
      go(0, tg.main, 1, pl.tile);
      go(tg.main, tg.ntiles - tg.main, nsplit, pl.tile);
    } else {
      go(0, tg.ntiles, nsplit, pl.tile);
    }
    if (nsplit > 1) {
      const long total = (long)a.cout * a.ws_npix;
      hipLaunchKernelGGL(conv_splitk_epilogue, dim3(cl4::cdiv(a.ws_npix, 256), a.cout), dim3(256), 0, st, a, nsplit);
    }
    return cl4::launch_status();
  }
  if (pl.impl == 2) {
    static float* zp = nullptr;   // one device per process (one process per GPU): resolve the symbol once
    if (!zp && hipGetSymbolAddress((void**)&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess) return (int)hipGetLastError();
    a.kchunk = pl.kchunk;
    a.ws = nullptr;
    if (pl.splits > 1) {
      const size_t need = (size_t)pl.splits * a.cout * a.npix * sizeof(float);
      if (!workspace || workspace_bytes < need) { cl4::set_error("conv2d_fwd: workspace too small for split-K"); return CL4_ENOSPC; }
      a.ws = (float*)workspace;
    }
    const dim3 g(cl4::cdiv(a.cout, pl.tile), cl4::cdiv(a.npix, pl.tile), pl.splits);
    const float* z = zp;
    if (pl.tile == 128) {
      if (pl.splits > 1) hipLaunchKernelGGL((conv_igemm_f32_v2<128, 128, KS, true>), g, dim3(V2_THREADS), 0, st, a, z);
      else hipLaunchKernelGGL((conv_igemm_f32_v2<128, 128, KS, false>), g, dim3(V2_THREADS), 0, st, a, z);
    } else {
      if (pl.splits > 1) hipLaunchKernelGGL((conv_igemm_f32_v2<64, 64, KS, true>), g, dim3(V2_THREADS), 0, st, a, z);
      else hipLaunchKernelGGL((conv_igemm_f32_v2<64, 64, KS, false>), g, dim3(V2_THREADS), 0, st, a, z);
    }
    if (pl.splits > 1) {
      const long total = (long)a.cout * a.npix;
      hipLaunchKernelGGL(conv_splitk_epilogue, dim3(cl4::cdiv(a.ws_npix, 256), a.cout), dim3(256), 0, st, a, pl.splits);
    }
    return cl4::launch_status();
  }
  const int np = cl4::cdiv(a.npix, 128);
  if (a.cout > 64) {
    dim3 g(cl4::cdiv(a.cout, 128), np);
    hipLaunchKernelGGL((conv_igemm_f32<128, 128, 2, 2, KS>), g, dim3(NTHREADS), 0, st, a);
  } else if (a.cout > 32) {
    dim3 g(1, np);
    hipLaunchKernelGGL((conv_igemm_f32<64, 128, 2, 2, KS>), g, dim3(NTHREADS), 0, st, a);
  } else {
    dim3 g(1, np);
    hipLaunchKernelGGL((conv_igemm_f32<32, 128, 1, 4, KS>), g, dim3(NTHREADS), 0, st, a);
  }
  return cl4::launch_status();
}

// bf16 split-product weight gradient: same staging as conv_wgrad_f32 (lane = pixel along K, rows fixed per thread), but the
// tiles are split into P bf16 planes on their way into LDS — P = 2: hi/lo, three v_mfma_f32_32x32x16_bf16 per fragment pair
// ("bf16x3", ~4e-6 relative, like the forward kernel); P = 3: hi/mid/lo, six products ("bf16x6", fp32-level accuracy). LDS
// image per plane: [row][4 k-groups of 8 bf16], 64-byte rows, the 16-byte k-group slots XOR-swizzled with (row >> 2) & 3 so
// that the ds_read_b128 fragment reads (32 rows x one k-group) are bank-conflict free without padding (2P planes x 2 stages
// of a 128-row tile = 64 KB for P = 2, 96 KB for P = 3).
// Wave specialisation like the forward kernel: waves 0-3 are loaders (global loads -> bf16 split -> LDS) with TWO register
// sets — the loads of K-tile t+2 are issued before tile t+1 (issued one phase earlier) is converted and stored, so every
// load has a whole phase to land — while waves 4-7 only read LDS fragments and issue MFMAs; one workgroup barrier per
// K-tile. (The earlier single-role kernel waited the full L2/HBM latency every K-tile: 75-120 TFLOP/s.) All loads are
// unconditional (out-of-range offsets past the end), which keeps the compiler's vmcnt accounting exact.
// VEC (stride-1 "same" convolutions, 1x1 / 3x3): the loader waves were bound by the CU's vector-memory front end — one wave-load
// per ~16 cycles whatever its width, 32 dword loads per thread and K-tile = ~2,000 cycles against 770 cycles of MFMA work (the r02g
// roofline leg: 144 TFLOP/s, 0.17 of the bf16x3 ceiling). K = pixels are contiguous in both operands (dy rows; x rows shifted by
// the tap, because win == wout makes the flattened pixel index linear across row ends), so a lane now fetches FOUR consecutive
// pixels of one row with one buffer_load_dwordx4: 4 + 4 loads per thread and K-tile. The K axis becomes the pixel space padded per
// image to a multiple of 4 (no group straddles two images); elements past the image's last pixel and taps outside the image are
// zeroed in registers from per-pixel tap masks; a group whose first element lies in front of the tensor (image 0, channel 0, the
// taps above / left of the first pixels) would be bounds-checked away as a whole, so its offset is clamped to 0 and the elements are
// shifted back in registers. Lanes hold 4 consecutive k of one row: bf16 pairs are packed without the DPP neighbour exchange and
// stored with ds_write_b64 into the same swizzled LDS image; the MFMA waves are unchanged.
template <int BM, int BN, int KS, int P, bool VEC = false>
__global__ __launch_bounds__(2 * NTHREADS) void conv_wgrad_bf16(WgradArgs a) {
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KHW = KS * KS;
  constexpr int RA = BM * BKP / NTHREADS;
  constexpr int RB = BN * BKP / NTHREADS;
  static_assert(BKP == 32, "one K tile = 4 k-groups of 8 pixels");
  static_assert(P == 2 || P == 3, "two (bf16x3) or three (bf16x6) planes");

  __shared__ __attribute__((aligned(16))) unsigned short Ash[2][P][BM * BKP];   // [stage][plane][row*32 + swizzled k]
  __shared__ __attribute__((aligned(16))) unsigned short Bsh[2][P][BN * BKP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 < 4;               // waves 0-3 stage the tiles, waves 4-7 run the MFMAs (wave-uniform roles)
  const int wave = wave8 & 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM;
  const int j0 = blockIdx.y * BN;
  const int pbeg = blockIdx.z * a.chunk;
  const int pend = min(VEC ? a.n * a.hw_pad : a.npix, pbeg + a.chunk);
  if (pbeg >= pend) return;

  const int HWo = a.hout * a.wout;
  const int HWi = a.hin * a.win;
  if constexpr (VEC) {
    if (wave8 < 4) {
      static_assert(KS == 1 || KS == 3, "VEC loader: 1x1 / 3x3");
      constexpr int RV = BM * 8 / NTHREADS, RW = BN * 8 / NTHREADS;      // x4 loads per thread and K-tile (A rows, B rows)
      constexpr unsigned OOBV = 0x80000000u;
      const int wv = wave8, kq = lane & 7, rl = lane >> 3;
      const int row0 = wv * 8 + rl;                                     // rows of this thread: row0 + 32 r
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)((long)a.n * a.in_ctot * HWi * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (unsigned)((long)a.n * a.out_ctot * HWo * 4), 0x00020000);
      int b_off[RW], b_tap[RW];                  // per B row (ci, tap): element offset relative to the pixel's own position, tap index (or 31: row out of range)
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int j = j0 + row0 + 32 * r;
        const bool ok = j < a.Nj;
        const int jc = ok ? j : 0;
        const int ci = jc / KHW, t = jc - ci * KHW, ky = t / KS, kx = t - ky * KS;
        b_off[r] = (a.in_coff + ci) * HWi + (ky * a.dil - a.pad) * a.win + (kx * a.dil - a.pad);
        b_tap[r] = ok ? t : 31;
      }
      const int lds_idx = row0 * BKP + (((kq >> 1) ^ ((row0 >> 2) & 3)) << 3) + ((kq & 1) << 2);   // (row0 + 32 r) >> 2 & 3 == row0 >> 2 & 3
      typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
      typedef float f32x2w __attribute__((ext_vector_type(2)));
      typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
      auto pk2 = [](float lo, float hi) { f32x2w f = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2v)); };
      auto load_tile = [&](int kt, f32x4v (&ra)[RV], f32x4v (&rb)[RW], unsigned (&em)[5]) {
        const int p0 = pbeg + kt * BKP + kq * 4;
        const bool gv = p0 < pend;               // pend is a multiple of 4
        const int pc = gv ? p0 : 0;
        const int img = pc / a.hw_pad, hw0 = pc - img * a.hw_pad;
        const int oy0 = hw0 / a.wout, ox0 = hw0 - oy0 * a.wout;
        // per element: tap mask (bit t: tap t of that pixel reads inside the image; 0 for the padding pixels of the image)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ox = ox0 + e, oy = oy0;
          if (ox >= a.wout) { ox -= a.wout; oy += 1; }     // wout >= 4 is required by the launcher: one wrap at most
          unsigned m = 0;
          if (KS == 1) m = 1u;
          else {
            unsigned cols = 0, rows = 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
              cols |= ((unsigned)(ox - a.pad + k * a.dil) < (unsigned)a.win ? 1u : 0u) << k;
              rows |= ((unsigned)(oy - a.pad + k * a.dil) < (unsigned)a.hin ? 1u : 0u) << k;
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) m |= ((rows >> k) & 1u) ? cols << (k * KS) : 0u;
          }
          em[e] = (gv && hw0 + e < HWo) ? m : 0u;
        }
        const unsigned gm = 0u - (unsigned)gv;
        const unsigned a_v = (unsigned)(((img * a.out_ctot + a.out_coff + m0 + row0) * HWo + hw0) * 4);
#pragma unroll
        for (int r = 0; r < RV; ++r) {
          const unsigned m = gm & (0u - (unsigned)(m0 + row0 + 32 * r < a.cout));
          ra[r] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rdy, (a_v & m) | (OOBV & ~m), r * 32 * HWo * 4, 0));
        }
        const int xb = img * a.in_ctot * HWi + hw0;
        unsigned shp = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int off = xb + b_off[r];         // element index of the group's first source pixel; negative only in front of the tensor
          const int offc = max(off, 0);
          const unsigned m = gm & (0u - (unsigned)(b_tap[r] != 31));
          rb[r] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rx, ((unsigned)(offc * 4) & m) | (OOBV & ~m), 0, 0));
          shp |= (unsigned)min(offc - off, 3) << (2 * r);   // 0, or how many leading elements lie in front of the tensor (undone at store time; those elements are masked)
        }
        em[4] = shp;
      };
      auto store_tile = [&](int buf, f32x4v (&ra)[RV], f32x4v (&rb)[RW], const unsigned (&em)[5]) {
        auto put = [&](unsigned short* base, int rows_total, int idx, const float (&f)[4]) {
          const unsigned h0 = pk2(f[0], f[1]), h1 = pk2(f[2], f[3]);
          const float r0 = f[0] - __builtin_bit_cast(float, h0 << 16), r1 = f[1] - __builtin_bit_cast(float, h0 & 0xFFFF0000u);
          const float r2 = f[2] - __builtin_bit_cast(float, h1 << 16), r3 = f[3] - __builtin_bit_cast(float, h1 & 0xFFFF0000u);
          *reinterpret_cast<u32x2w*>(base + idx) = u32x2w{h0, h1};
          if (P == 2) {
            *reinterpret_cast<u32x2w*>(base + (P - 1) * rows_total * BKP + idx) = u32x2w{pk2(r0, r1), pk2(r2, r3)};
          } else {
            const unsigned q0 = pk2(r0, r1), q1 = pk2(r2, r3);
            *reinterpret_cast<u32x2w*>(base + rows_total * BKP + idx) = u32x2w{q0, q1};
            const float s0 = r0 - __builtin_bit_cast(float, q0 << 16), s1 = r1 - __builtin_bit_cast(float, q0 & 0xFFFF0000u);
            const float s2 = r2 - __builtin_bit_cast(float, q1 << 16), s3 = r3 - __builtin_bit_cast(float, q1 & 0xFFFF0000u);
            *reinterpret_cast<u32x2w*>(base + 2 * rows_total * BKP + idx) = u32x2w{pk2(s0, s1), pk2(s2, s3)};
          }
        };
        // hipcc narrows some of these b128 loads to their first dword although all four elements are read (FINDINGS.md finding 15; here:
        // every load of register set 0). An empty asm that "modifies" the whole vector at its point of use keeps all elements demanded
        // without forcing a wait where the load is issued.
#pragma unroll
        for (int r = 0; r < RV; ++r) asm volatile("" : "+v"(ra[r]));
#pragma unroll
        for (int r = 0; r < RW; ++r) asm volatile("" : "+v"(rb[r]));
#pragma unroll
        for (int r = 0; r < RV; ++r) {
          float f[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = em[e] ? ra[r][e] : 0.f;   // "the pixel exists" = any tap bit (the centre tap always reads inside)
          put(&Ash[buf][0][0], BM, lds_idx + 32 * r * BKP, f);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          f32x4v v = rb[r];
          const unsigned sh = (em[4] >> (2 * r)) & 3u;
          v[3] = sh == 0 ? v[3] : sh == 1 ? v[2] : sh == 2 ? v[1] : v[0];
          v[2] = sh == 0 ? v[2] : sh == 1 ? v[1] : v[0];
          v[1] = sh == 0 ? v[1] : v[0];
          float f[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = ((em[e] >> (b_tap[r] & 31)) & 1u) ? v[e] : 0.f;
          put(&Bsh[buf][0][0], BN, lds_idx + 32 * r * BKP, f);
        }
      };
      const int nk = (pend - pbeg + BKP - 1) / BKP;
      f32x4v ra0[RV], rb0[RW], ra1[RV], rb1[RW];     // float vectors: element access on an unsigned-vector view of the b128 builtin's result read element 0 four times (hipcc, ROCm 7.2)
      unsigned em0[5], em1[5];
      load_tile(0, ra0, rb0, em0);
      load_tile(1, ra1, rb1, em1);
      store_tile(0, ra0, rb0, em0);
      __syncthreads();
      for (int kt = 0; kt < nk; kt += 2) {        // tile kt is in LDS stage 0, tile kt+1 in register set 1
        load_tile(kt + 2, ra0, rb0, em0);
        store_tile(1, ra1, rb1, em1);
        __syncthreads();
        load_tile(kt + 3, ra1, rb1, em1);
        store_tile(0, ra0, rb0, em0);
        __syncthreads();
      }
      return;
    }
  }
  const int pl = lane & 31;      // pixel within the K tile
  const int rsub = lane >> 5;    // which of the two rows this half-wave stages

  // Per staged B row (ci, tap): element offset relative to the pixel's (iy0, ix0) corner and the tap's bit in the
  // per-pixel validity mask (computed once per K tile and lane for all KS*KS taps).
  int b_off[RB];                               // byte offset of the row's (ci, tap) relative to the pixel's corner
  typedef typename std::conditional<(KHW <= 25), unsigned, unsigned long long>::type mask_t;   // one bit per filter tap
  constexpr bool SMALL_MASK = KHW <= 25;       // 32-bit tap mask: bit 31 is never set and doubles as "row out of range"
  int b_sh[RB];                                // SMALL_MASK: 31 - tap, so that (vmask << b_sh) has the tap's bit as its sign
  mask_t b_bit[SMALL_MASK ? 1 : RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    int row = wave * (BN / 4) + 2 * r + rsub;
    int j = j0 + row;
    const bool ok = j < a.Nj;
    int jc = ok ? j : 0;
    int ci = jc / KHW;
    int t = jc - ci * KHW;
    int ky = t / KS;
    b_off[r] = ((a.in_coff + ci) * HWi + ky * a.dil * a.win + (t - ky * KS) * a.dil) * 4;
    b_sh[r] = ok ? 31 - t : 0;
    if (!SMALL_MASK) b_bit[r] = ok ? (mask_t)1 << t : (mask_t)0;
  }
  // 32-bit buffer addressing (host guarantees < 2 GiB tensors): masked elements read 0 through an out-of-range offset
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)((long)a.n * a.in_ctot * HWi * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (unsigned)((long)a.n * a.out_ctot * HWo * 4), 0x00020000);
  const bool m_interior = m0 + BM <= a.cout;   // uniform

  auto load_tile = [&](int kt, float (&ra)[RA], float (&rb)[RB]) {
    int p = pbeg + kt * BKP + pl;
    bool pv = p < pend;                        // false for every lane of a tile past the end: all offsets out of range
    int pc = pv ? p : 0;
    int n = pc / HWo;
    int hw = pc - n * HWo;
    int oy = hw / a.wout;
    int ox = hw - oy * a.wout;
    int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    const int dybase = (n * a.out_ctot + a.out_coff) * HWo + hw;
    const int xbase = n * a.in_ctot * HWi + iy0 * a.win + ix0;
    mask_t vmask = 0;                         // bit t: tap t of this pixel reads inside the image
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const bool in = (unsigned)(iy0 + ky * a.dil) < (unsigned)a.hin && (unsigned)(ix0 + kx * a.dil) < (unsigned)a.win;
        vmask |= in ? (mask_t)1 << (ky * KS + kx) : (mask_t)0;
      }
    vmask = pv ? vmask : (mask_t)0;
    // Branch-free offset selects (a ?: makes hipcc split a load into divergent paths that it then drains with vmcnt(0)).
    // A rows: row r of this lane is 2r rows further down -> the same vector offset plus a SCALAR offset r * 2*HWo*4.
    const unsigned pvm = 0u - (unsigned)pv;
    const unsigned a_v = (unsigned)(dybase + (m0 + wave * (BM / 4) + rsub) * HWo) * 4u;
    const int a2 = 2 * HWo * 4;
    if (m_interior) {                          // uniform
      const unsigned voff = (a_v & pvm) | (OOB & ~pvm);
#pragma unroll
      for (int r = 0; r < RA; ++r) ra[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, voff, r * a2, 0));
    } else {
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const int co = m0 + wave * (BM / 4) + 2 * r + rsub;
        const unsigned m = pvm & (0u - (unsigned)(co < a.cout));
        ra[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, ((a_v + r * a2) & m) | (OOB & ~m), 0, 0));
      }
    }
    const unsigned x4 = (unsigned)xbase * 4u;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      unsigned m;
      if (SMALL_MASK) m = (unsigned)((int)((unsigned)vmask << b_sh[r]) >> 31);
      else m = 0u - (unsigned)((vmask & b_bit[SMALL_MASK ? 0 : r]) != 0);
      rb[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ((x4 + (unsigned)b_off[r]) & m) | (OOB & ~m), 0, 0));
    }
  };
  // element (row, k = pl) -> bf16 index row*32 + ((k>>3) ^ ((row>>2)&3))*8 + (k&7). Neighbouring lanes hold neighbouring
  // k of the same row, and a lane stores whole bf16 PAIRS: it pulls the neighbour's fp32 value through DPP and converts both
  // with one v_cvt_pk_bf16_f32 — even lanes (k, k+1) of the hi plane, odd lanes (k-1, k) of the lo plane (of the residuals),
  // one ds_write_b32 per element; for P = 3 the even lanes also store the mid pair. Rows of one lane are 2r apart, so
  // (row >> 2) & 3 == (r >> 1) & 3 is known at compile time: four per-lane base indices, everything else is immediate.
  const bool odd = pl & 1;
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  auto pk = [](float lo, float hi) { f32x2v f = {lo, hi}; return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2)); };
  auto next_of = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x101, 0xf, 0xf, true)); };
  auto prev_of = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, v), 0x111, 0xf, 0xf, true)); };
  auto store_rows = [&](unsigned short* base, auto rows_c, const auto& vals) {
    constexpr int ROWS = decltype(rows_c)::value;
    constexpr int NV = sizeof(vals) / sizeof(float);     // compile-time trip count: the values stay in registers
    int lane_idx[4];
#pragma unroll
    for (int sw = 0; sw < 4; ++sw)
      lane_idx[sw] = (odd ? (P - 1) * ROWS * BKP : 0) + (wave * (ROWS / 4) + rsub) * BKP + (((pl >> 3) ^ sw) << 3) + (pl & 6);
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const float v = vals[r];
      const unsigned hi2 = pk(v, next_of(v));                                  // low half: bf16(v) of this lane
      const float r1 = v - __builtin_bit_cast(float, hi2 << 16);
      const int idx = lane_idx[(r >> 1) & 3] + 2 * r * BKP;
      if (P == 2) {
        const unsigned lo2 = pk(prev_of(r1), r1);
        *reinterpret_cast<unsigned*>(base + idx) = odd ? lo2 : hi2;
      } else {
        const unsigned mid2 = pk(r1, next_of(r1));
        const float r2 = r1 - __builtin_bit_cast(float, mid2 << 16);
        const unsigned lo2 = pk(prev_of(r2), r2);
        *reinterpret_cast<unsigned*>(base + idx) = odd ? lo2 : hi2;
        if (!odd) *reinterpret_cast<unsigned*>(base + idx + ROWS * BKP) = mid2;
      }
    }
  };
  auto store_tile = [&](int buf, const float (&ra)[RA], const float (&rb)[RB]) {
    store_rows(&Ash[buf][0][0], std::integral_constant<int, BM>{}, ra);
    store_rows(&Bsh[buf][0][0], std::integral_constant<int, BN>{}, rb);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  auto compute = [&](int buf) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int kg = 2 * s2 + lh;
      bf16x8 af[TM][P], bfr[TN][P];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + lr;
        const int o = row * BKP + ((kg ^ ((row >> 2) & 3)) << 3);
#pragma unroll
        for (int q = 0; q < P; ++q) af[i][q] = *reinterpret_cast<const bf16x8*>(&Ash[buf][q][o]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + lr;
        const int o = row * BKP + ((kg ^ ((row >> 2) & 3)) << 3);
#pragma unroll
        for (int q = 0; q < P; ++q) bfr[j][q] = *reinterpret_cast<const bf16x8*>(&Bsh[buf][q][o]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest products first; planes: 0 = hi, P-1 = lo, (1 = mid for P = 3)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][P - 1], bfr[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][P - 1], acc[i][j], 0, 0, 0);
          if (P == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nk = (pend - pbeg + BKP - 1) / BKP;
  if (loader) {
    float ra0[RA], rb0[RB], ra1[RA], rb1[RB];
    load_tile(0, ra0, rb0);
    load_tile(1, ra1, rb1);
    store_tile(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {        // tile kt is in LDS stage 0, tile kt+1 in register set 1
      load_tile(kt + 2, ra0, rb0);
      store_tile(1, ra1, rb1);
      __syncthreads();
      load_tile(kt + 3, ra1, rb1);
      store_tile(0, ra0, rb0);
      __syncthreads();
    }
    return;
  }
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    compute(0);
    __syncthreads();
    if (kt + 1 < nk) compute(1);                // uniform
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jj = j0 + wn * WTN + j * 32 + lr;
    if (jj >= a.Nj) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= a.cout) continue;
        atomicAdd(&a.dw[(long)co * a.Nj + jj], acc[i][j][r]);
      }
  }
}

template <int KS>
int launch_wgrad(WgradArgs& a, hipStream_t st) {
  const bool small_m = a.cout <= 64;
  const int BMv = small_m ? 64 : 128;
  const int gm = cl4::cdiv(a.cout, BMv), gn = cl4::cdiv(a.Nj, 128);
  // Split K (pixels) so that the grid fills the chip in whole rounds: `slots` workgroups are resident at once (2 per CU for
  // the 64 KB kernels, 1 per CU for the 96 KB bf16x6 tile); among the split counts near 512 / (gm*gn) (r01-r03: 1024) pick the one whose last
  // round is fullest (fewest splits on ties: every split adds one atomic pass over dW). Each slice is a multiple of BKP.
  // bf16x3 MFMA path: 32-bit buffer offsets, so both tensors must stay below 2 GiB (always true on the phase-2 path)
  const long lim = (1L << 31) - 4096;
  // bf16x6 (precision 2) pays off for k >= 3 (114 vs 88 TFLOP/s on the ASPP 3x3); for 1x1 the exact fp32 kernel is as fast
  const bool split3 = (a.precision == 1 || (a.precision == 2 && KS > 1)) && (long)a.n * a.in_ctot * a.hin * a.win * 4 < lim && (long)a.n * a.out_ctot * a.hout * a.wout * 4 < lim;
  static const int vec_enabled = env_int("CL4_WGRAD_VEC", 1);
  a.hw_pad = (a.hout * a.wout + 3) & ~3;
  const bool vec = vec_enabled && split3 && (KS == 1 || KS == 3) && a.stride == 1 && a.hin == a.hout && a.win == a.wout && a.wout >= 4 &&
                   a.pad == a.dil * (KS / 2);
  const int kpix = vec ? a.n * a.hw_pad : a.npix;
  const int tiles = cl4::cdiv(kpix, BKP);
  const int slots = (a.precision == 2 && KS > 1 && !small_m) ? 256 : 512;
  // r04: one round of the 512 slots, not two (tools/probes/probe_wgrad_split.py: 3x3 256 -> 256 at 33 x 33 x 16 132.7 -> 110.4 us, 1x1 1280 -> 256
  // 83.9 -> 65.4 us, every other shape of the steps within 2 %; 256 is no better, 2048 is 10-30 % worse: each split is one more 64 KB
  // atomic pass over the tile of dW)
  static const int target_wgs = env_int("CL4_WGRAD_TARGET_WGS", 512);
  int want = cl4::cdiv(target_wgs, gm * gn);
  if (want < 1) want = 1;
  if (want > tiles) want = tiles;
  int best_s = want;
  double best_eff = 0.0;
  for (int sp = cl4::cdiv(want, 2); sp <= 2 * want + 1 && sp <= tiles; ++sp) {
    const int per = cl4::cdiv(tiles, sp);
    const long wgs = (long)gm * gn * cl4::cdiv(tiles, per);
    const long rounds = (wgs + slots - 1) / slots;
    const double eff = (double)((long)gm * gn * tiles) / (double)(rounds * slots * per);     // useful K-tiles per slot-K-tile
    if (eff > best_eff + 0.02) { best_eff = eff; best_s = sp; }
  }
  const int tiles_per = cl4::cdiv(tiles, best_s);
  a.chunk = tiles_per * BKP;
  const int gz = cl4::cdiv(kpix, a.chunk);
  dim3 g(gm, gn, gz);
  if constexpr (KS == 1 || KS == 3) {
    if (vec) {
      if (a.precision == 2 && small_m) hipLaunchKernelGGL((conv_wgrad_bf16<64, 128, KS, 3, true>), g, dim3(2 * NTHREADS), 0, st, a);
      else if (a.precision == 2) hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, KS, 3, true>), g, dim3(2 * NTHREADS), 0, st, a);
      else if (small_m) hipLaunchKernelGGL((conv_wgrad_bf16<64, 128, KS, 2, true>), g, dim3(2 * NTHREADS), 0, st, a);
      else hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, KS, 2, true>), g, dim3(2 * NTHREADS), 0, st, a);
      return cl4::launch_status();
    }
  }
  if (split3 && a.precision == 2 && small_m)
    hipLaunchKernelGGL((conv_wgrad_bf16<64, 128, KS, 3>), g, dim3(2 * NTHREADS), 0, st, a);
  else if (split3 && a.precision == 2)
    hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, KS, 3>), g, dim3(2 * NTHREADS), 0, st, a);
  else if (split3 && small_m)
    hipLaunchKernelGGL((conv_wgrad_bf16<64, 128, KS, 2>), g, dim3(2 * NTHREADS), 0, st, a);
  else if (split3)
    hipLaunchKernelGGL((conv_wgrad_bf16<128, 128, KS, 2>), g, dim3(2 * NTHREADS), 0, st, a);
  else if (small_m)
    hipLaunchKernelGGL((conv_wgrad_f32<64, 128, KS>), g, dim3(NTHREADS), 0, st, a);
  else
    hipLaunchKernelGGL((conv_wgrad_f32<128, 128, KS>), g, dim3(NTHREADS), 0, st, a);
  return cl4::launch_status();
}

}  // namespace

extern "C" int cl4_pack_conv_weight(const float* w, float* wp, int cout, int cin, int kh, int kw, void* stream) {
  CL4_REQUIRE(w && wp && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_conv_weight: bad arguments");
  int K = cin * kh * kw;
  dim3 g(cl4::cdiv(K, 32), cl4::cdiv(cout, 32));
  hipLaunchKernelGGL(pack_weight_kernel, g, dim3(256), 0, cl4::as_stream(stream), w, wp, cout, K);
  return cl4::launch_status();
}

extern "C" int cl4_pack_conv_weight_bf16x3(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream) {
  CL4_REQUIRE(w && wq && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_conv_weight_bf16x3: bad arguments");
  const int K = cin * kh * kw;
  const long total = (long)((K + 7) / 8) * cout;
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), w,
                     (bf16x8*)wq, cout, K, cin, kh * kw, (cin % 32 == 0) ? 1 : 0, 2);
  return cl4::launch_status();
}

extern "C" int cl4_pack_conv_weight_bf16x6(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream) {
  CL4_REQUIRE(w && wq && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_conv_weight_bf16x6: bad arguments");
  const int K = cin * kh * kw;
  const long total = (long)((K + 7) / 8) * cout;
  hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), w,
                     (bf16x8*)wq, cout, K, cin, kh * kw, (cin % 32 == 0) ? 1 : 0, 3);
  return cl4::launch_status();
}

extern "C" int cl4_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           float* scale, float* shift, int c, void* stream) {
  CL4_REQUIRE(mean && var && scale && shift && c > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cl4::cdiv(c, 256)), dim3(256), 0, cl4::as_stream(stream), gamma, beta, mean,
                     var, eps, scale, shift, c);
  return cl4::launch_status();
}

extern "C" size_t cl4_conv2d_fwd_workspace_bytes(const cl4_conv_desc* d) {
  if (!d || d->cout <= 0 || d->n <= 0) return 0;
  const int K = d->cin * d->kh * d->kw;
  int npix = d->n * d->hout * d->wout;
  const FwdPlan pl = plan_fwd(PlanKey{d->cout, npix, K, d->kh, d->stride, d->dil, d->hin * 8 + d->precision});
  ConvArgs va;      // the fields vec_gather_plan looks at
  va.n = d->n; va.cin = d->cin; va.hin = d->hin; va.win = d->win; va.hout = d->hout; va.wout = d->wout; va.stride = d->stride;
  va.pad = d->pad; va.in_ctot = d->in_ctot; va.dil = d->dil;
  if (pl.impl == 2 && (vec_gather_plan(va, d->kh, d->precision, pl.tile) || u_plan(va, d->kh, d->precision, pl.tile)))
    npix = d->n * ((d->hout * d->wout + 3) & ~3);
  if (planes_code(d->precision) && pl.impl == 6 && d->precision == 3 && pl.splits > 1)                        // r05: K-split slabs of the LDS-DMA kernel
    return (size_t)pl.splits * d->cout * npix * sizeof(float);
  if (planes_code(d->precision) && (pl.impl != 2 || pl.tile != 128)) return 0;      // impl 3 ... 6 unsplit (persistent / LDS-DMA kernels): no slabs
  if (d->precision != 0 && pl.impl == 2 && pl.tail > 0) {
    const TailGeom tg = tail_geometry(pl, d->cout, npix, d->precision);
    return (tg.main > 0 && pl.tail > 1) ? (size_t)pl.tail * d->cout * (npix - tg.p0) * sizeof(float) : 0;   // small-tile tail / degenerate: no slabs
  }
  return pl.splits > 1 ? (size_t)pl.splits * d->cout * npix * sizeof(float) : 0;
}

// Pin the launch plan of one convolution shape (impl 1 = register-staged, 2 = LDS-DMA; tile 64|128; K splits).
// impl = 0 removes the entry. Used by the host-side autotuner, which times the candidates during warm-up.
extern "C" int cl4_conv2d_fwd_set_plan(const cl4_conv_desc* d, int impl, int tile, int splits) {
  int rc = check_desc(d);
  if (rc) return rc;
  const int K = d->cin * d->kh * d->kw;
  const PlanKey key{d->cout, d->n * d->hout * d->wout, K, d->kh, d->stride, d->dil, d->hin * 8 + d->precision};
  std::lock_guard<std::mutex> g(plan_mutex());
  if (impl == 0) { plan_table().erase(key); return CL4_OK; }
  CL4_REQUIRE(((impl == 1 || impl == 2) && d->precision != 5) || (impl >= 3 && impl <= 6 && d->precision == 3) || (impl == 6 && d->precision == 4) ||
                  ((impl == 4 || impl == 6) && d->precision == 5),
              "set_plan: impl must be 0, 1 or 2 (3 ... 6: planes-input launches only; three planes: 6 only; fp16x3: 4 or 6)");
  CL4_REQUIRE(tile == 64 || tile == 128 || (tile == 144 && planes_code(d->precision)), "set_plan: tile must be 64 or 128 (144: planes-input launches only)");
  plan_table()[key] = make_plan(impl, tile, splits, d->cout, K, d->precision);
  return CL4_OK;
}

extern "C" int cl4_conv2d_fwd_clear_plans(void) {
  std::lock_guard<std::mutex> g(plan_mutex());
  plan_table().clear();
  return CL4_OK;
}

static int conv2d_fwd_common(const cl4_conv_desc* d, const float* x, const void* xs, const float* wp, const float* scale,
                             const float* shift, const float* pre, const float* post, const void* post_s, float* y, void* ys,
                             void* workspace, size_t workspace_bytes, void* stream, const float* xs_scale = nullptr, double* stats = nullptr,
                             int stats_np = 0) {
  ConvArgs a;
  a.xs_scale = xs_scale;
  a.dbg_stagger = 0;
  a.stats = stats;
  a.stats_np = stats_np;
  a.x = x; a.wp = wp; a.scale = scale; a.shift = shift; a.pre = pre; a.post = post; a.y = y;
  a.n = d->n; a.cin = d->cin; a.hin = d->hin; a.win = d->win; a.cout = d->cout;
  a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.hout = d->hout; a.wout = d->wout;
  a.in_ctot = d->in_ctot; a.in_coff = d->in_coff; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.K = d->cin * d->kh * d->kw;
  a.npix = d->n * d->hout * d->wout;
  a.hw_pad = d->hout * d->wout;
  a.ws = nullptr; a.kchunk = 0;
  a.gx = 1; a.ntile_l = 1; a.tile_base = 0; a.ws_p0 = 0; a.ws_npix = a.npix;
  static const int xcd_default = env_int("CL4_XCD_REMAP", 1);
  a.xcd_remap = xcd_default;
  a.timeline = g_timeline;
  a.xs = xs; a.ys = ys; a.post_s = post_s;
  a.neg1 = d->act1 == CL4_ACT_NONE ? 1.f : (d->act1 == CL4_ACT_RELU ? 0.f : d->slope1);
  a.neg2 = d->act2 == CL4_ACT_NONE ? 1.f : (d->act2 == CL4_ACT_RELU ? 0.f : d->slope2);
  hipStream_t st = cl4::as_stream(stream);
  switch (d->kh) {
    case 1: return launch_fwd<1>(a, st, workspace, workspace_bytes, d->precision);
    case 3: return launch_fwd<3>(a, st, workspace, workspace_bytes, d->precision);
    default: return launch_fwd<7>(a, st, workspace, workspace_bytes, d->precision);
  }
}

extern "C" int cl4_conv2d_fwd(const cl4_conv_desc* d, const float* x, const float* wp, const float* scale,
                              const float* shift, const float* pre, const float* post, float* y, void* workspace,
                              size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(x && wp && y, "conv2d_fwd: null tensor");
  CL4_REQUIRE(d->precision >= 0 && d->precision <= 2, "conv2d_fwd: precision %d unknown (0 fp32, 1 bf16x3, 2 bf16x6)", d->precision);
  CL4_REQUIRE(d->precision == 0 || (d->cout % 4 == 0 && d->cout >= 64), "conv2d_fwd: bf16x3 / bf16x6 need cout %% 4 == 0 and cout >= 64");
  return conv2d_fwd_common(d, x, nullptr, wp, scale, shift, pre, post, nullptr, y, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int cl4_conv2d_fwd_planes(const cl4_conv_desc* d, const void* xs, const float* wp, const float* scale, const float* shift,
                                     const float* pre, const float* post, const void* post_s, float* y, void* ys, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(xs && wp && (y || ys), "conv2d_fwd_planes: null tensor");
  CL4_REQUIRE(d->precision == 3 || d->precision == 4, "conv2d_fwd_planes: the descriptor's precision must be 3 (bf16x3 on a two-plane input) or 4 (bf16x6 on a three-plane input); fp16x3: cl4_conv2d_fwd_planes_f16");
  CL4_REQUIRE(d->precision == 3 || (!ys && !post_s), "conv2d_fwd_planes: a three-plane (bf16x6) launch writes fp32 only and takes no planes residual");
  CL4_REQUIRE(d->kh == 1 || d->kh == 3, "conv2d_fwd_planes: 1x1 and 3x3 kernels only");
  CL4_REQUIRE(d->cin % 32 == 0 && d->in_ctot % 8 == 0 && d->in_coff % 8 == 0, "conv2d_fwd_planes: cin %% 32 == 0 and 8-aligned input channel slices");
  CL4_REQUIRE(d->cout % 4 == 0 && d->cout >= 64, "conv2d_fwd_planes: cout %% 4 == 0 and cout >= 64");
  CL4_REQUIRE(!ys || (d->cout % 8 == 0 && d->out_ctot % 8 == 0 && d->out_coff % 8 == 0), "conv2d_fwd_planes: a planes output needs 8-aligned output channels");
  CL4_REQUIRE(!post_s || d->cout % 8 == 0, "conv2d_fwd_planes: a planes residual needs cout %% 8 == 0");
  CL4_REQUIRE(!(post && post_s), "conv2d_fwd_planes: one residual operand (post or post_s)");
  CL4_REQUIRE((long)d->n * d->in_ctot * d->hin * d->win * (d->precision == 4 ? 6 : 4) < 0x7FFFFFFFL, "conv2d_fwd_planes: input >= 2 GiB is not addressable by the 32-bit buffer offsets");
  return conv2d_fwd_common(d, nullptr, xs, wp, scale, shift, pre, post, post_s, y, ys, workspace, workspace_bytes, stream);
}

// fp16x3 (r05): the same convolution group on an fp16 two-plane input (cl4_to_planes_f16: fp16 hi / lo of x * *xs_scale) and
// cl4_pack_conv_weight_f16x3 weights (fp16 hi / lo of w * 2^12): three MFMA products per MAC at bf16x6's accuracy. fp32 output.
extern "C" int cl4_conv2d_fwd_planes_f16(const cl4_conv_desc* d, const void* xs, const float* xs_scale, const float* wp, const float* scale,
                                         const float* shift, const float* pre, const float* post, float* y, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(xs && xs_scale && wp && y, "conv2d_fwd_planes_f16: null tensor");
  CL4_REQUIRE(d->precision == 5, "conv2d_fwd_planes_f16: the descriptor's precision must be 5 (fp16x3 on a two-plane fp16 input)");
  CL4_REQUIRE(d->kh == 1 || d->kh == 3, "conv2d_fwd_planes_f16: 1x1 and 3x3 kernels only");
  CL4_REQUIRE(d->cin % 32 == 0 && d->in_ctot % 8 == 0 && d->in_coff % 8 == 0, "conv2d_fwd_planes_f16: cin %% 32 == 0 and 8-aligned input channel slices");
  CL4_REQUIRE(d->cout % 8 == 0 && d->cout >= 128, "conv2d_fwd_planes_f16: cout %% 8 == 0 and cout >= 128");
  CL4_REQUIRE((long)d->n * d->in_ctot * d->hin * d->win * 4 < 0x7FFFFFFFL, "conv2d_fwd_planes_f16: input >= 2 GiB is not addressable by the 32-bit buffer offsets");
  return conv2d_fwd_common(d, nullptr, xs, wp, scale, shift, pre, post, nullptr, y, nullptr, workspace, workspace_bytes, stream, xs_scale);
}

// r05: a plain planes convolution (no affine, activation or residual: the differentiated networks' conv in front of a train-mode ABN) that
// also leaves the per-channel partial sums of its output for that ABN. stats_parts(d) = partial slots per channel the CURRENT launch plan of
// this shape produces (its pixel tiles), 0 when the plan cannot (K-split slabs, main + tail plans): the caller then runs the separate pass.
static int stats_parts_of(const cl4_conv_desc* d) {
  if (!(d->precision == 3 || d->precision == 5) || !(d->kh == 1 || d->kh == 3)) return 0;
  const int K = d->cin * d->kh * d->kw, npix = d->n * d->hout * d->wout;
  const FwdPlan pl = plan_fwd(PlanKey{d->cout, npix, K, d->kh, d->stride, d->dil, d->hin * 8 + d->precision});
  if (d->precision == 5) return cl4::cdiv(npix, 144);                           // always the LDS-DMA kernel, unsplit
  if (pl.impl >= 4 && pl.impl <= 6) return pl.splits > 1 ? 0 : cl4::cdiv(npix, 144);
  if (pl.impl == 3) return cl4::cdiv(npix, 128);                               // (runs on the plain planes kernel when statistics are asked for)
  if (pl.impl == 2 && pl.tile == 128 && pl.splits == 1 && pl.tail == 0) return cl4::cdiv(npix, pl.pix());
  return 0;
}
extern "C" int cl4_conv2d_fwd_planes_stats_parts(const cl4_conv_desc* d) {
  if (check_desc(d)) return 0;
  return stats_parts_of(d);
}
extern "C" int cl4_conv2d_fwd_planes_stats(const cl4_conv_desc* d, const void* xs, const float* xs_scale, const float* wp, float* y, double* stats,
                                           int stats_np, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(xs && wp && y && stats, "conv2d_fwd_planes_stats: null tensor");
  CL4_REQUIRE((d->precision == 3 && !xs_scale) || (d->precision == 5 && xs_scale), "conv2d_fwd_planes_stats: precision 3 (the build's planes) or 5 (fp16 planes with their scale)");
  CL4_REQUIRE(d->act1 == CL4_ACT_NONE && d->act2 == CL4_ACT_NONE, "conv2d_fwd_planes_stats: a plain convolution (the statistics are those of the accumulators)");
  CL4_REQUIRE(d->kh == 1 || d->kh == 3, "conv2d_fwd_planes_stats: 1x1 and 3x3 kernels only");
  CL4_REQUIRE(d->cin % 32 == 0 && d->in_ctot % 8 == 0 && d->in_coff % 8 == 0, "conv2d_fwd_planes_stats: cin %% 32 == 0 and 8-aligned input channel slices");
  CL4_REQUIRE(d->cout % 8 == 0 && d->cout >= (d->precision == 5 ? 128 : 64), "conv2d_fwd_planes_stats: cout %% 8 == 0 and cout >= 64 (128 for fp16 planes with a scale)");
  CL4_REQUIRE((long)d->n * d->in_ctot * d->hin * d->win * 4 < 0x7FFFFFFFL, "conv2d_fwd_planes_stats: input >= 2 GiB is not addressable by the 32-bit buffer offsets");
  const int np = stats_parts_of(d);
  CL4_REQUIRE(np > 0 && np == stats_np, "conv2d_fwd_planes_stats: stats_np %d != cl4_conv2d_fwd_planes_stats_parts() = %d (0: this shape's plan cannot produce them)", stats_np, np);
  return conv2d_fwd_common(d, nullptr, xs, wp, nullptr, nullptr, nullptr, nullptr, nullptr, y, nullptr, workspace, workspace_bytes, stream, xs_scale, stats, np);
}

// Experiments only (not part of the drop-in boundary): device buffer of 8 x uint64 per workgroup that the bf16 forward kernels
// fill with phase time stamps, or NULL to switch the instrumentation off (the default).
extern "C" int cl4_conv_debug_timeline(void* buf) {
  g_timeline = (unsigned long long*)buf;
  return CL4_OK;
}

extern "C" int cl4_conv2d_wgrad(const cl4_conv_desc* d, const float* x, const float* dy, float* dw, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dw, "conv2d_wgrad: null tensor");
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.n = d->n; a.cin = d->cin; a.hin = d->hin; a.win = d->win; a.cout = d->cout;
  a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.hout = d->hout; a.wout = d->wout;
  a.in_ctot = d->in_ctot; a.in_coff = d->in_coff; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
  CL4_REQUIRE(d->precision >= 0 && d->precision <= 2, "conv2d_wgrad: precision %d unknown (0 fp32, 1 bf16x3, 2 bf16x6)", d->precision);
  a.precision = d->precision;
  a.Nj = d->cin * d->kh * d->kw;
  a.npix = d->n * d->hout * d->wout;
  a.chunk = 0;
  hipStream_t st = cl4::as_stream(stream);
  switch (d->kh) {
    case 1: return launch_wgrad<1>(a, st);
    case 3: return launch_wgrad<3>(a, st);
    default: return launch_wgrad<7>(a, st);
  }
}
