// Fused identity bottleneck on planes ("F" kernel, r04): conv1 1x1 (C4 -> CM) + ABN + leaky, conv2 3x3 (CM -> CM, stride 1, dilation 1) +
// ABN + leaky, conv3 1x1 (CM -> C4) + ABN + residual add + leaky of ONE ResidualBlock (modules/residual.py:84-97) in ONE launch. The
// high-resolution stages of the ResNet (res2: 256 channels at 129 x 129, res3: 512 at 65 x 65) are HBM-bound as three launches: every
// block reads the trunk twice (conv1, residual), writes it once and moves both CM-channel intermediates through HBM as well — 2.2 GB per
// res2 block at 32 images for 74 GFLOP (FINDINGS.md findings 4, 20, 25, 31). Here a workgroup owns a TH x TW tile
// of output pixels of one image:
//
//   phase 1  y1[CM][halo]  = leaky(bn1(W1 . x[C4][halo]))      halo = (TH + 2) x (TW + 2) pixels; y1 = 0 outside the image (conv2's zero
//                                                               padding). The x chunks go global -> registers in MFMA B-fragment layout
//                                                               (a planes chunk IS a fragment row): no LDS traffic for the big operand.
//   phase 2  y2[CM][tile]  = leaky(bn2(W2 * y1))                the nine taps are nine LDS offsets into the halo image of y1
//   phase 3  out[C4][tile] = leaky(bn3(W3 . y2) + x)            32 output channels per step; residual chunks re-read (L2 / Infinity
//                                                               Cache: this workgroup fetched them a few microseconds ago)
//
// so the trunk is read once (+ halo overlap, served by L2) and written once, and y1 / y2 never leave the CU: they live in LDS in the
// planes format (bf16 hi / lo chunks of 8 channels per pixel), which is exactly what the next phase's MFMA B fragments read.
// All weights stream through one two-stage LDS ring as a single sequence of (8 * CM)-chunk tiles across the three phases (K-tiles of
// W1, then of W2 in the packed order (channel block, tap), then 32-row slabs of W3), requested two tiles ahead.
//
// No accumulator parking: the rows of every weight tile are permuted on their way into LDS so that MFMA row block b of a 32-channel
// pair holds channels 8q + 4b + r in row 4q + r; lane (pixel c = lane & 15, q = lane >> 4) then owns channels 8q .. 8q + 7 of pixel c
// across the two row blocks of the pair = one planes chunk, which it splits and stores itself (to LDS for y1 / y2, to HBM for out).
//
// Arithmetic: the products, their order (K-tiles ascending; lo*hi, hi*lo, hi*hi per tile) and the epilogue expressions are those of
// conv_s_bf16x3, so a fused block returns what the three planes launches return (tests/test_conv_f_gpu.py compares bit for bit).
#include "conv_args.h"

#include "planes_epilogue.h"

#ifndef CL4_F_DBG
#define CL4_F_DBG 0   // experiments only (tools/probes/probe_f.py): 1 = no x loads in phase 1, 2 = no residual loads, 4 = no output stores, 8 = no MFMAs,
                      // 16 = no weight loads
#endif

namespace cl4 {

struct FusedArgs {
  const void* xs;      // input planes [n][C4/8][hi, lo][h*w][8 x bf16]
  void* ys;            // output planes, same shape; must not alias xs (neighbouring tiles read each other's halo)
  float* y;            // optional fp32 NCHW copy of the output
  const void* w1;      // packed bf16x3 weights (cl4_pack_conv_weight_bf16x3): [K/8][hi, lo][cout][8]
  const void* w2;
  const void* w3;
  const void* wp;      // PROJ only: the projection shortcut's weight (cin = CM -> C4), packed like w3
  const float* aff;    // [sc1 | sh1 | sc2 | sh2] (CM each), [sc3 | sh3] (C4 each) (, PROJ: [scp | shp] (C4 each)): folded eval-mode ABN
  float neg1, neg2, neg3, negout;   // branch-free activations act(v) = v > 0 ? v : v * neg of bn1, bn2, bn3 (identity: 1) and of the sum
  int n, h, w, tiles_x, tiles_y, xcd_remap;
  int n_f32;           // the fp32 copy is written for images [0, n_f32) only (the others' stores are dropped by the descriptor)
};

}  // namespace cl4

namespace {

using cl4::FusedArgs;

typedef float f32x4a __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4a mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4a c) {
#if CL4_F_DBG & 8
  c[0] += (float)al[0] * (float)bh[0] + (float)ah[1] * (float)bl[1];
  return c;
#endif
  return plfmt::mfma16_x3(ah, al, bh, bl, c);      // the two-plane arithmetic of this build (planes_fmt.h)
}

// RS = row split of phases 1 and 2: the NW waves form NW / RS column groups x RS row groups; a wave multiplies CM / RS of the CM rows of W1 /
// W2 with the column blocks of its column group (RS = 2: the 8-wave workgroup of the 128-channel res3 blocks — as 8 column groups its
// 11 / 15 column blocks would leave 31 % of the matrix pipe's lanes idle, and 128 rows x 4 blocks do not fit one wave's accumulators).
// The two waves of a column group load the same x chunks in phase 1 (the second request hits the CU's vector cache).
// PROJ = the stage's FIRST block (modules/residual.py:79-82: projection shortcut): the input has CM channels (res2: the pooled stem output),
// conv1 is CM -> CM, and the residual operand is bn_p(Wp . x) — a second 32-row slab GEMM per step of phase 3 (K = CM, the B fragments are
// the tile's own x chunks, fetched once and held in registers), rounded through the planes split like the stored shortcut of the
// three-launch path. The weight stream of phase 3 then alternates slabs of W3 and Wp.
// Every 16-byte buffer load lands in a FLOAT vector. hipcc (ROCm 7.2) mishandles the builtin's own <4 x i32> result in this kernel: in the
// projection variant it reused the FIRST register of a loaded weight chunk as an address temporary between the load and the ds_write_b128
// that stores the chunk (v_lshl_add_u32 v30, ... ; ds_write_b128 v171, v[30:33]), so every weight row reached LDS with a wrong first dword —
// the same family as FINDINGS.md finding 15 (element 0 of a raw_buffer_load_b128 result). A real bitcast to <4 x float> at the load keeps
// the four registers together (conv_u.hip's workaround); the values go back to integer vectors only where they are consumed.
typedef float ldv_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ldv_t bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(ldv_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
// ... and where a loaded chunk is consumed, an empty asm takes all four registers as an in/out operand: the bitcast alone did not stop the
// allocator (same ISA), this does (the address temporaries moved to another register).
__device__ __forceinline__ u32x4 as_u32x4(const ldv_t& v) {
  u32x4 t = __builtin_bit_cast(u32x4, v);
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ bf16x8 as_bf16x8(const ldv_t& v) { return __builtin_bit_cast(bf16x8, as_u32x4(v)); }

template <int CM, int NW, int RS, int TH, int TW, bool F32OUT, bool PROJ>
__global__ __launch_bounds__(NW * 64, 2) void bottleneck_fused_kernel(FusedArgs a) {
  constexpr int NT = NW * 64;
  constexpr int CIN = PROJ ? CM : 4 * CM;                  // channels of the block's input
  constexpr int NWC = NW / RS;                             // column groups
  constexpr int C4 = 4 * CM;
  constexpr int HWD = TW + 2;                              // halo width
  constexpr int HP = (TH + 2) * HWD;                       // halo pixels
  constexpr int NP = TH * TW;                              // output pixels
  constexpr int CB1 = (HP + 15) / 16, CB2 = (NP + 15) / 16;   // 16-pixel column blocks of phase 1 / phases 2, 3
  constexpr int NJ1 = (CB1 + NWC - 1) / NWC, NJ2 = (CB2 + NWC - 1) / NWC;   // ... per wave in phases 1, 2 (block index = wcol + NWC * j)
  constexpr int NJ3 = (CB2 + NW - 1) / NW;                 // ... in phase 3 (32 output channels per step: no row split; block = wave + NW * j)
  constexpr int HPS = (HP + 7) & ~7;                       // chunk stride of one (channel group, plane) row of y1: a multiple of 8 keeps the
                                                           // four k-groups of a fragment read 128 B-aligned against each other (LDS banks)
  constexpr int NPS = CB2 * 16;                            // ... of y2
  static_assert(NPS <= HPS, "y2 reuses y1's LDS");
  constexpr int RB = CM / 16 / RS, RP = CM / 32 / RS;      // MFMA row blocks / 32-channel pairs of conv1, conv2 PER WAVE
  static_assert(NW % RS == 0 && (CM / 32) % RS == 0, "row split");
  constexpr int KT1 = CIN / 32, KT2 = 9 * CM / 32, KT3 = CM / 32, NM3 = C4 / 32;
  constexpr int T3 = PROJ ? 2 * NM3 : NM3;                 // weight tiles of phase 3
  constexpr int G = KT1 + KT2 + T3;                        // weight tiles of the whole kernel
  constexpr int WCH = 8 * CM;                              // chunks per weight tile (32 k x CM rows, or 32 rows x CM k: two planes)
  constexpr int LW = WCH / NT;                             // weight chunks per thread and tile
  static_assert(WCH % NT == 0 && LW >= 1, "weight staging");
  constexpr int LB = NJ1 * 2;                              // x chunks per thread and K-tile of phase 1
  constexpr int NR = NJ3 * 2;                              // residual chunks per thread and step of phase 3
  constexpr int NS = NJ3 * (F32OUT ? 10 : 2);              // stores per thread and step of phase 3

  __shared__ u32x4 ybuf[(CM / 8) * 2 * HPS];               // y1, then y2: [channel group][hi, lo][pixel] chunks
  __shared__ u32x4 wst[2][WCH];                            // weight ring
  constexpr int NAFF = 4 * CM + 2 * C4 + (PROJ ? 2 * C4 : 0);
  __shared__ float aff[NAFF];
#ifdef CL4_F_PAD
  __shared__ float lds_pad[CL4_F_PAD];                     // experiments: force one workgroup per CU
  if (a.n < 0) lds_pad[threadIdx.x] = 1.f;
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wcol = RS == 1 ? wave : wave % NWC, wrow = RS == 1 ? 0 : wave / NWC;      // wave-uniform

  // ---- which tile ------------------------------------------------------------------------------------------------------------------
  const unsigned total = gridDim.x, id = blockIdx.x;
  const unsigned q8 = total >> 3, r8 = total & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const int tpi = a.tiles_x * a.tiles_y;
  const int img = lid / tpi, tl = lid - img * tpi;
  const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;
  const int H = a.h, W = a.w, HW = H * W;

  // (fp16x3 build, planes_fmt.h: the operand scales 2^-16 ride in the folded-ABN SCALES and the epilogues work in the format's scaled domain —
  // scale and shift x 16, residuals joined and outputs split without the extra multiplications — exact: powers of two)
  for (int i = tid; i < NAFF; i += NT) {
    const bool is_scale = i < CM || (i >= 2 * CM && i < 3 * CM) || (i >= 4 * CM && i < 4 * CM + C4) || (i >= 4 * CM + 2 * C4 && i < 4 * CM + 3 * C4);
    aff[i] = a.aff[i] * (is_scale ? plfmt::ACC * plfmt::XSCALE : plfmt::XSCALE);
  }

  const unsigned act_bytes = (unsigned)((long)a.n * C4 * HW * 4);      // < 2^31: checked on the host
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, (unsigned)((long)a.n * CIN * HW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.ys, 0, act_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (F32OUT && img < a.n_f32) ? act_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, (unsigned)(CIN * CM * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, (unsigned)(9 * CM * CM * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w3rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w3, 0, (unsigned)(CM * C4 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wprsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, PROJ ? (unsigned)(CIN * C4 * 4) : 0u, 0x00020000);

  // ---- weight stream: tile g -> registers (set g & 1) -> LDS stage g & 1 -------------------------------------------------------------
  // K-tiles of W1 / W2 (rows = CM): chunk e = r * CM + s, r = 2 * k-group + plane, LDS slot s holds weight row pair * 32 + q * 8 + b * 4 + i
  // for s = pair * 32 + b * 16 + q * 4 + i (the row permutation of the header). Slabs of W3 (32 rows x CM k): chunk e = rr * 32 + s,
  // rr = 8 * K-tile + 2 * k-group + plane, slot s = b * 16 + q * 4 + i holds row q * 8 + b * 4 + i of the slab.
  auto unperm = [](int s) { return (s & ~31) | (((s >> 2) & 3) << 3) | (((s >> 4) & 1) << 2) | (s & 3); };
  unsigned w12_off[LW], w3_off[LW];                        // per-thread byte offsets inside a tile of W1 / W2 resp. a slab of W3
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int e = tid + NT * i;
    const int r = e / CM, s = e - r * CM;
    w12_off[i] = (unsigned)((r * CM + unperm(s)) * 16);
    const int rr = e >> 5, s3 = e & 31;
    const int kt = rr >> 3, r3 = rr & 7;
    w3_off[i] = (unsigned)((((kt * 8 + r3) * C4) + unperm(s3)) * 16);
  }
  auto wissue = [&](ldv_t (&wr)[LW], int g) {
#if CL4_F_DBG & 16
    g = G;
#endif
    // uniform: which weight, which tile
    if (g < KT1) {
      const unsigned so = (unsigned)g * (unsigned)(WCH * 16);
#pragma unroll
      for (int i = 0; i < LW; ++i) wr[i] = bload16(w1rsrc, w12_off[i], so);
    } else if (g < KT1 + KT2) {
      const unsigned so = (unsigned)(g - KT1) * (unsigned)(WCH * 16);
#pragma unroll
      for (int i = 0; i < LW; ++i) wr[i] = bload16(w2rsrc, w12_off[i], so);
    } else {
      const int t3 = g - KT1 - KT2;
      const unsigned so = g < G ? (unsigned)(PROJ ? t3 >> 1 : t3) * 32u * 16u : 0u;
      const unsigned keep = (unsigned)(-(int)(g < G));
      if (PROJ && (t3 & 1)) {
#pragma unroll
        for (int i = 0; i < LW; ++i) wr[i] = bload16(wprsrc, (w3_off[i] & keep) | (OOB & ~keep), so);
      } else {
#pragma unroll
        for (int i = 0; i < LW; ++i) wr[i] = bload16(w3rsrc, (w3_off[i] & keep) | (OOB & ~keep), so);
      }
    }
  };
  auto wstore = [&](const ldv_t (&wr)[LW], int g) {
    u32x4* st = wst[g & 1];
#pragma unroll
    for (int i = 0; i < LW; ++i) st[tid + NT * i] = as_u32x4(wr[i]);
  };

  // ---- phase-1 geometry: this lane's halo pixel in each of the wave's column blocks -------------------------------------------------------
  unsigned xoff[NJ1];                                      // byte offset of (k-group q, plane 0, pixel) inside an image's 32-channel block, or OOB
  bool inimg[NJ1];
  int hp1[NJ1];
#pragma unroll
  for (int j = 0; j < NJ1; ++j) {
    const int hp = (wcol + NWC * j) * 16 + c16;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    hp1[j] = hp;
    inimg[j] = hp < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    xoff[j] = inimg[j] ? (unsigned)((q * 2 * HW + iy * W + ix) * 16) : OOB;
  }
  const unsigned img_base = (unsigned)img * (unsigned)(CIN / 8) * 2u * (unsigned)HW * 16u;    // byte offset of the image in the INPUT planes tensor
  const unsigned oimg_base = (unsigned)img * (unsigned)(C4 / 8) * 2u * (unsigned)HW * 16u;    // ... in the output planes tensor
  auto bissue = [&](ldv_t (&bq)[NJ1][2], int kt) {
#if CL4_F_DBG & 1
    kt = KT1;
#endif
    const unsigned keep = (unsigned)(-(int)(kt < KT1));
    const unsigned so = img_base + (unsigned)(kt < KT1 ? kt : 0) * 8u * (unsigned)HW * 16u;   // 4 channel groups x 2 planes per K-tile
#pragma unroll
    for (int j = 0; j < NJ1; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        bq[j][p] = bload16(xrsrc, ((xoff[j] + (unsigned)(p * HW * 16)) & keep) | (OOB & ~keep), so);
  };

  // ===================================================================================================================================
  // phase 1: y1 = leaky(bn1(W1 . x)) on the halo
  // ===================================================================================================================================
  ldv_t wr[2][LW];
  ldv_t bq[3][NJ1][2];                                     // x chunks of K-tiles g, g + 1, g + 2 (mod 3)
  // issue stream of phase 1: W0 B0 | W1 B1 | W2 B2 | then iteration g adds W(g+3) B(g+3)
  wissue(wr[0], 0);
  bissue(bq[0], 0);
  wissue(wr[1], 1);
  bissue(bq[1], 1);
  wait_vmcnt<LW + 2 * LB>();                               // W0 has landed (B0, W1, B1 may be in flight)
  wstore(wr[0], 0);
  wissue(wr[0], 2);
  bissue(bq[2], 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4a acc1[RB][NJ1];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NJ1; ++j) acc1[i][j] = f32x4a{0.f, 0.f, 0.f, 0.f};

  // iteration g of phase 1 (K-tile g of conv1). In flight at its start, oldest first: W(g+1) B(g+1) | W(g+2) B(g+2) behind B(g). It needs W(g)
  // in LDS (stored by the previous iteration), B(g) in registers and W(g+1) in registers (to store it): everything but the last
  // LB + LW + LB operations.
  auto step1 = [&](ldv_t (&w_next)[LW], ldv_t (&b_cur)[NJ1][2], int g) {
    wait_vmcnt<LW + 2 * LB>();
    const u32x4* As = wst[g & 1];
    bf16x8 ah[RB], al[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, As[(q * 2 + 0) * CM + (wrow * RB + i) * 16 + c16]);
      al[i] = __builtin_bit_cast(bf16x8, As[(q * 2 + 1) * CM + (wrow * RB + i) * 16 + c16]);
    }
    wstore(w_next, g + 1);
    wissue(w_next, g + 3);
#pragma unroll
    for (int j = 0; j < NJ1; ++j) {
      if (wcol + NWC * j < CB1) {                          // wave-uniform
        // (128-channel variant: 256 registers without the liveness marker, a spill with it; its fragments feed MFMAs directly)
        const bf16x8 bh = CM == 128 ? __builtin_bit_cast(bf16x8, b_cur[j][0]) : as_bf16x8(b_cur[j][0]);
        const bf16x8 bl = CM == 128 ? __builtin_bit_cast(bf16x8, b_cur[j][1]) : as_bf16x8(b_cur[j][1]);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc1[i][j] = mfma3(ah[i], al[i], bh, bl, acc1[i][j]);
      }
    }
    bissue(b_cur, g + 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  static_assert(KT1 % 2 == 0, "the weight ring's parity continues into phase 2");
#pragma unroll
  for (int g = 0; g < KT1; ++g) step1(wr[(g + 1) & 1], bq[g % 3], g);      // fully unrolled: the register sets are compile-time indices
  wait_vmcnt<0>();                                         // W(KT1 + 1), W(KT1 + 2) are in registers, the dummy x loads are done

  // epilogue 1: scale / shift / leaky, zero outside the image, split, into LDS
#pragma unroll
  for (int pr = 0; pr < RP; ++pr) {
    const int cg = 4 * (wrow * RP + pr) + q;
    float sc[8], sh[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { sc[r] = aff[cg * 8 + r]; sh[r] = aff[CM + cg * 8 + r]; }
#pragma unroll
    for (int j = 0; j < NJ1; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc1[2 * pr][j][r]; v[4 + r] = acc1[2 * pr + 1][j][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float u = v[r] * sc[r] + sh[r];
        const float t = u > 0.f ? u : u * a.neg1;
        v[r] = inimg[j] ? t : 0.f;
      }
      u32x4 hi, lo;
      split8s(v, hi, lo);
      if (hp1[j] < HP) {
        ybuf[(cg * 2 + 0) * HPS + hp1[j]] = hi;
        ybuf[(cg * 2 + 1) * HPS + hp1[j]] = lo;
      }
    }
  }

  // ---- geometry of phases 2 and 3: this lane's output pixel in each of the wave's column blocks ---------------------------------------
  int h0[NJ2];                                             // phase 2: chunk index of the pixel's top-left tap inside a y1 row
  int pcol2[NJ2];                                          // ... and the pixel's slot in a y2 row
#pragma unroll
  for (int j = 0; j < NJ2; ++j) {
    const int p = (wcol + NWC * j) * 16 + c16;
    const int oy = p / TW, ox = p - oy * TW;
    pcol2[j] = (wcol + NWC * j) < CB2 ? p : 0;
    h0[j] = p < NP ? oy * HWD + ox : 0;
  }
  unsigned ooff[NJ3];                                      // phase 3: byte offset of (channel group q, plane 0, pixel) inside a 32-channel block, or OOB
  unsigned foff[NJ3];                                      // ... of (channel 8 q, pixel) in the fp32 tensor
  int pcol[NJ3];
#pragma unroll
  for (int j = 0; j < NJ3; ++j) {
    const int p = (wave + NW * j) * 16 + c16;
    const int oy = p / TW, ox = p - oy * TW;
    const bool valid = (wave + NW * j) < CB2 && p < NP && y0 + oy < H && x0 + ox < W;
    pcol[j] = (wave + NW * j) < CB2 ? p : 0;
    ooff[j] = valid ? (unsigned)((q * 2 * HW + (y0 + oy) * W + x0 + ox) * 16) : OOB;
    foff[j] = valid ? (unsigned)((q * 8 * HW + (y0 + oy) * W + x0 + ox) * 4) : OOB;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // y1 is complete

  // ===================================================================================================================================
  // phase 2: y2 = leaky(bn2(W2 * y1)); K-tile t2 = (channel block, tap) in the packed order
  // ===================================================================================================================================
  f32x4a acc2[RB][NJ2];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NJ2; ++j) acc2[i][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
  // iteration g: W(g) in LDS, W(g+1) and W(g+2) requested; in flight at most W(g+2) when W(g+1) is needed
  // B fragments (y1, static in LDS for the whole phase) are read ONE ITERATION AHEAD: their LDS latency runs under the current tile's MFMAs,
  // only the weight fragments (which need the ring's barrier) are read at the head of an iteration.
  auto bread2 = [&](bf16x8 (&bh)[NJ2], bf16x8 (&bl)[NJ2], int g) {
    const int t2 = (g < KT1 + KT2 ? g : KT1 + KT2 - 1) - KT1;
    const int cb = t2 / 9, tap = t2 - cb * 9;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int toff = __builtin_amdgcn_readfirstlane(ky * HWD + kx);
    const u32x4* Bs = ybuf + ((cb * 4 + q) * 2) * HPS + toff;
#pragma unroll
    for (int j = 0; j < NJ2; ++j) {
      bh[j] = __builtin_bit_cast(bf16x8, Bs[h0[j]]);
      bl[j] = __builtin_bit_cast(bf16x8, Bs[HPS + h0[j]]);
    }
  };
  auto step2 = [&](ldv_t (&w_next)[LW], bf16x8 (&bh)[NJ2], bf16x8 (&bl)[NJ2], bf16x8 (&bh_n)[NJ2], bf16x8 (&bl_n)[NJ2], int g) {
    wait_vmcnt<LW>();
    const u32x4* As = wst[g & 1];
    bf16x8 ah[RB], al[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, As[(q * 2 + 0) * CM + (wrow * RB + i) * 16 + c16]);
      al[i] = __builtin_bit_cast(bf16x8, As[(q * 2 + 1) * CM + (wrow * RB + i) * 16 + c16]);
    }
    bread2(bh_n, bl_n, g + 1);
    wstore(w_next, g + 1);
    wissue(w_next, g + 3);
#pragma unroll
    for (int j = 0; j < NJ2; ++j)
      if (wcol + NWC * j < CB2) {
#pragma unroll
        for (int i = 0; i < RB; ++i) acc2[i][j] = mfma3(ah[i], al[i], bh[j], bl[j], acc2[i][j]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  static_assert(KT2 % 2 == 0 && KT1 % 2 == 0, "phase 2 is unrolled by two and starts on an even tile");
  {
    bf16x8 bh0[NJ2], bl0[NJ2], bh1[NJ2], bl1[NJ2];
    bread2(bh0, bl0, KT1);
    for (int g = KT1; g < KT1 + KT2; g += 2) {
      step2(wr[1], bh0, bl0, bh1, bl1, g);
      step2(wr[0], bh1, bl1, bh0, bl0, g + 1);
    }
  }
  wait_vmcnt<0>();
  // the residual chunks of the first two steps of phase 3 travel under the epilogue of phase 2 (the geometry is known since phase 1)
  ldv_t rq[2][NJ3][2];                                     // residual chunks of steps m, m + 1
  auto rissue = [&](ldv_t (&r)[NJ3][2], int m) {
#if CL4_F_DBG & 2
    m = NM3;
#endif
    const unsigned keep = (unsigned)(-(int)(m < NM3));
    const unsigned so = img_base + (unsigned)(m < NM3 ? m : 0) * 8u * (unsigned)HW * 16u;
#pragma unroll
    for (int j = 0; j < NJ3; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        r[j][p] = bload16(xrsrc, ((ooff[j] + (unsigned)(p * HW * 16)) & keep) | (OOB & ~keep), so);
  };
  constexpr int KTP = PROJ ? CIN / 32 : 1;
  ldv_t xr[KTP][NJ3][2];                                   // PROJ: the tile's own x chunks = B fragments of the shortcut GEMM, all K-tiles
#if !(CL4_F_DBG & 1024)
  if constexpr (PROJ) {
#pragma unroll
    for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
      for (int j = 0; j < NJ3; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          xr[kt][j][p] = bload16(xrsrc, ooff[j] == OOB ? OOB : ooff[j] + (unsigned)(p * HW * 16), img_base + (unsigned)kt * 8u * (unsigned)HW * 16u);
  } else
#endif
  if constexpr (!PROJ) {
    rissue(rq[0], 0);
    rissue(rq[1], 1);
  }
  // every wave has left the K loop (barrier above): y1 is dead, y2 takes its place
#pragma unroll
  for (int pr = 0; pr < RP; ++pr) {
    const int cg = 4 * (wrow * RP + pr) + q;
    float sc[8], sh[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { sc[r] = aff[2 * CM + cg * 8 + r]; sh[r] = aff[3 * CM + cg * 8 + r]; }
#if CL4_F_DBG & 128
#pragma unroll
    for (int r = 0; r < 8; ++r) { sc[r] = 1.f; sh[r] = 0.5f; }
#endif
#pragma unroll
    for (int j = 0; j < NJ2; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc2[2 * pr][j][r]; v[4 + r] = acc2[2 * pr + 1][j][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float u = v[r] * sc[r] + sh[r];
        v[r] = u > 0.f ? u : u * a.neg2;
      }
      u32x4 hi, lo;
      split8s(v, hi, lo);
      if (wcol + NWC * j < CB2) {
        ybuf[(cg * 2 + 0) * NPS + pcol2[j]] = hi;
        ybuf[(cg * 2 + 1) * NPS + pcol2[j]] = lo;
      }
    }
  }

  // ===================================================================================================================================
  // phase 3: out = leaky(bn3(W3 . y2) + x), 32 output channels (one slab of W3) per step
  // ===================================================================================================================================
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // y2 is complete
#if CL4_F_DBG & 1024
  if constexpr (PROJ) {
#pragma unroll
    for (int kt = 0; kt < KTP; ++kt)
#pragma unroll
      for (int j = 0; j < NJ3; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          xr[kt][j][p] = bload16(xrsrc, ooff[j] == OOB ? OOB : ooff[j] + (unsigned)(p * HW * 16), img_base + (unsigned)kt * 8u * (unsigned)HW * 16u);
  }
#endif
  // step m (weight tile g = KT1 + KT2 + m). Issue order of a step: W(g+3), residual R(m+2), stores S(m) — the loads in FRONT of the stores,
  // because vmcnt retires in order: a wait for a load also waits for every older store's acknowledgement (measured: with the stores
  // first the steps stalled on them, 102 of 428 us). At the start of step m >= 2 everything up to R(m) has to be there; S(m-2) and the whole
  // of step m-1 may be in flight: 2 NS + LW + NR (capped by the 6-bit counter). Step 1 has no S(-1) in front of it, step 0 sees only R(0) R(1).
  constexpr int WAIT3 = (2 * NS + LW + NR) < 63 ? (2 * NS + LW + NR) : 63;
  // one 32-row slab (weight tile in `As`: 32 rows x CM k) against B fragments handed out by `bfrag(kt, j, plane)`
  auto slab3 = [&](const u32x4* As, f32x4a (&acc)[2][NJ3], auto bfrag) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < NJ3; ++j) acc[b][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT3; ++kt) {
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ah[b] = __builtin_bit_cast(bf16x8, As[(kt * 8 + q * 2 + 0) * 32 + b * 16 + c16]);
        al[b] = __builtin_bit_cast(bf16x8, As[(kt * 8 + q * 2 + 1) * 32 + b * 16 + c16]);
#if CL4_F_DBG & 256
        ah[b] = __builtin_bit_cast(bf16x8, u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u});
        al[b] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#endif
      }
#pragma unroll
      for (int j = 0; j < NJ3; ++j)
        if (wave + NW * j < CB2) {
#if CL4_F_DBG & 512
          const bf16x8 bh = __builtin_bit_cast(bf16x8, u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}), bl = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#else
          const bf16x8 bh = bfrag(kt, j, 0), bl = bfrag(kt, j, 1);
#endif
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b][j] = mfma3(ah[b], al[b], bh, bl, acc[b][j]);
        }
    }
  };
  auto y2frag = [&](int kt, int j, int pl) { return __builtin_bit_cast(bf16x8, ybuf[((kt * 4 + q) * 2 + pl) * NPS + pcol[j]]); };
  // identity blocks: the y2 fragments of a lane are the same for all NM3 steps — read them once (KT3 x NJ3 x 2 chunks: 48 registers at
  // 64 channels). The projection variant keeps reading them per step (its registers hold the x chunks of the shortcut GEMM).
  constexpr bool HOIST_Y2 = !PROJ && CM == 64;
  bf16x8 y2r[HOIST_Y2 ? KT3 : 1][HOIST_Y2 ? NJ3 : 1][2];
  if constexpr (HOIST_Y2) {
#pragma unroll
    for (int kt = 0; kt < KT3; ++kt)
#pragma unroll
      for (int j = 0; j < NJ3; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) y2r[kt][j][pl] = y2frag(kt, j, pl);
  }
  auto y2reg = [&](int kt, int j, int pl) { return y2r[HOIST_Y2 ? kt : 0][HOIST_Y2 ? j : 0][pl]; };
  // epilogue of step m: channels 32 m + 8 q .. + 7 of this lane's pixels = bn3(acc) [+ res] -> leaky -> chunks -> stores.
  // `between()` runs after the last use of `res` and before the stores (the residual loads of a later step go there).
  auto epi3 = [&](int m, const f32x4a (&acc3)[2][NJ3], const float (&res)[NJ3][8], auto between) {
    const int cg = 4 * m + q;
    float sc[8], sh[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { sc[r] = aff[4 * CM + cg * 8 + r]; sh[r] = aff[4 * CM + C4 + cg * 8 + r]; }
    const unsigned so = oimg_base + (unsigned)m * 8u * (unsigned)HW * 16u;
    const unsigned fso = ((unsigned)img * (unsigned)C4 + 32u * (unsigned)m) * (unsigned)HW * 4u;
    float v[NJ3][8];
    u32x4 ohi[NJ3], olo[NJ3];
#pragma unroll
    for (int j = 0; j < NJ3; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[j][r] = acc3[0][j][r]; v[j][4 + r] = acc3[1][j][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float u = v[j][r] * sc[r] + sh[r];
        float t = u > 0.f ? u : u * a.neg3;
        t += res[j][r];
        v[j][r] = t > 0.f ? t : t * a.negout;
      }
      split8s(v[j], ohi[j], olo[j]);
      if constexpr (F32OUT) {                              // the fp32 copy leaves the scaled domain (before the store fence below)
#pragma unroll
        for (int r = 0; r < 8; ++r) v[j][r] *= plfmt::XINV;
      }
    }
    // Every chunk of the step is complete BEFORE the first store is issued, and nothing is computed between the stores: a
    // buffer_store_dwordx4 whose soffset is an SGPR gets no wait state from hipcc (ROCm 7.2) in front of a VALU write of its data
    // registers — the ISA manual exempts that form — but on gfx950 under load the store then picks the NEW value up (first dword of the chunk =
    // whatever the next instruction wrote there: FINDINGS.md finding 37; found as 4-pixel groups of wrong channels 8 k, 8 k + 1 in
    // the workgroups of the second round only). Here the stores also keep their whole offset in the VGPR (soffset 0: the form
    // the compiler does guard), and the scheduler may not move anything across the fence in front of them.
    unsigned ahi[NJ3], alo[NJ3], af32[NJ3];               // ... and every store address
#pragma unroll
    for (int j = 0; j < NJ3; ++j) {
#if CL4_F_DBG & 4
      const unsigned skeep = v[j][0] == 123.456f ? ~0u : 0u;     // keeps the arithmetic alive, drops (practically) every store
#else
      const unsigned skeep = ~0u;
#endif
      const unsigned o0 = (ooff[j] & skeep) | (OOB & ~skeep);
      ahi[j] = o0 == OOB ? OOB : o0 + so;
      alo[j] = o0 == OOB ? OOB : o0 + so + (unsigned)(HW * 16);
      af32[j] = (foff[j] & skeep) | (OOB & ~skeep);
    }
    between();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ3; ++j) {
      if constexpr (F32OUT) {                              // dword stores: no multi-cycle data read, the uniform part may stay in the SGPR offset
#pragma unroll
        for (int r = 0; r < 8; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j][r]), frsrc, af32[j], fso + (unsigned)(r * HW * 4), 0);
      }
      __builtin_amdgcn_raw_buffer_store_b128(ohi[j], yrsrc, ahi[j], 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(olo[j], yrsrc, alo[j], 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(NM3 % 2 == 0 && (KT1 + KT2) % 2 == 0, "phase 3 is unrolled by two and starts on an even tile");
  if constexpr (!PROJ) {
    auto step3 = [&](ldv_t (&w_next)[LW], ldv_t (&r_cur)[NJ3][2], int m, int which) {
      if (which == 0) wait_vmcnt<NR>(); else if (which == 1) wait_vmcnt<LW + NR + NS>(); else wait_vmcnt<WAIT3>();
      const int g = KT1 + KT2 + m;
      f32x4a acc3[2][NJ3];
      if constexpr (HOIST_Y2) slab3(wst[g & 1], acc3, y2reg); else slab3(wst[g & 1], acc3, y2frag);
      wstore(w_next, g + 1);
      wissue(w_next, g + 3);
      float res[NJ3][8];
#pragma unroll
      for (int j = 0; j < NJ3; ++j) join8s(as_u32x4(r_cur[j][0]), as_u32x4(r_cur[j][1]), res[j]);
      epi3(m, acc3, res, [&] { rissue(r_cur, m + 2); });    // into the registers the sums above have just consumed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    step3(wr[1], rq[0], 0, 0);
    step3(wr[0], rq[1], 1, 1);
    for (int m = 2; m < NM3; m += 2) {
      step3(wr[1], rq[0], m, 2);
      step3(wr[0], rq[1], m + 1, 2);
    }
  } else {
    // two ring tiles per step: the slab of W3 (against y2), then the slab of Wp (against the tile's x chunks). Issue stream per step: W | W S(m);
    // a tile is needed two iterations after its request, with at most W + S(m-1) issued behind it.
    constexpr int WAIT3P = (LW + NS) < 63 ? (LW + NS) : 63;
    auto xfrag = [&](int kt, int j, int pl) { return as_bf16x8(xr[kt < KTP ? kt : 0][j][pl]); };
    for (int m = 0; m < NM3; ++m) {
      const int g = KT1 + KT2 + 2 * m;                     // even: W3's slab sits in stage 0, Wp's in stage 1
      f32x4a acc3[2][NJ3], accp[2][NJ3];
      wait_vmcnt<WAIT3P>();
      slab3(wst[0], acc3, y2frag);
      wstore(wr[1], g + 1);
      wissue(wr[1], g + 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      wait_vmcnt<WAIT3P>();
#if CL4_F_DBG & 32
      slab3(wst[1], accp, y2frag);
#else
      slab3(wst[1], accp, xfrag);
#endif
      wstore(wr[0], g + 2);
      wissue(wr[0], g + 4);
      // the shortcut bn_p(Wp . x), rounded through the planes split exactly like the stored shortcut of the three-launch path
      const int cg = 4 * m + q;
      float res[NJ3][8];
#pragma unroll
      for (int j = 0; j < NJ3; ++j) {
        float pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = accp[0][j][r]; pv[4 + r] = accp[1][j][r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) pv[r] = pv[r] * aff[4 * CM + 2 * C4 + cg * 8 + r] + aff[4 * CM + 3 * C4 + cg * 8 + r];
        u32x4 ph, pl;
        split8s(pv, ph, pl);
        join8s(ph, pl, res[j]);
      }
#if CL4_F_DBG & 2048
#pragma unroll
      for (int j = 0; j < NJ3; ++j) {
        join8s(ybuf[(((m & 1) * 4 + q) * 2 + 0) * NPS + pcol[j]], ybuf[(((m & 1) * 4 + q) * 2 + 1) * NPS + pcol[j]], res[j]);
#pragma unroll
        for (int b = 0; b < 2; ++b) acc3[b][j] = f32x4a{0.f, 0.f, 0.f, 0.f};
      }
#endif
#if CL4_F_DBG & 64
      epi3(m, accp, res, [] {});
#else
      epi3(m, acc3, res, [] {});
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
}

}  // namespace

static long g_f_launches = 0;
extern "C" long cl4_conv_debug_f_launches(void) { return g_f_launches; }

namespace {

// tile of output pixels per workgroup, by middle width: res2 (cm 64, 129 x 129 maps) 13 x 13, two 4-wave workgroups per CU; res3 (cm 128,
// 65 x 65 maps) one 8-wave workgroup per CU (y1 alone is 116 KB)
constexpr int F_TH64 = 13, F_TW64 = 13, F_TH128 = 11, F_TW128 = 13;

}  // namespace

extern "C" int cl4_bottleneck_fused_tile(int cm, int* th, int* tw) {
  if (cm == 64) { *th = F_TH64; *tw = F_TW64; return CL4_OK; }
  if (cm == 128) { *th = F_TH128; *tw = F_TW128; return CL4_OK; }
  return CL4_EINVAL;
}

static int fused_launch(const void* xs, void* ys, float* y, int n_f32, const void* w1, const void* w2, const void* w3, const void* wp,
                        const float* aff, int n, int cm, int h, int w, float slope1, float slope2, float slope_out, void* stream) {
  CL4_REQUIRE(xs && ys && w1 && w2 && w3 && aff, "bottleneck_fused: null operand");
  CL4_REQUIRE(xs != ys, "bottleneck_fused: the output must not alias the input (tiles read their neighbours' halo)");
  CL4_REQUIRE(cm == 64 || (cm == 128 && !wp), "bottleneck_fused: middle width %d not built (64; 128 without projection shortcut)", cm);
  CL4_REQUIRE(n > 0 && h > 0 && w > 0, "bottleneck_fused: bad shape");
  CL4_REQUIRE((long)n * 4 * cm * h * w * 4 < (1l << 31), "bottleneck_fused: tensor of %ld bytes exceeds the 2 GiB buffer-descriptor range",
              (long)n * 4 * cm * h * w * 4);
  cl4::FusedArgs a;
  a.xs = xs; a.ys = ys; a.y = y; a.w1 = w1; a.w2 = w2; a.w3 = w3; a.wp = wp; a.aff = aff;
  a.neg1 = slope1; a.neg2 = slope2; a.neg3 = 1.f; a.negout = slope_out;
  a.n = n; a.h = h; a.w = w;
  a.n_f32 = y ? (n_f32 < 0 || n_f32 > n ? n : n_f32) : 0;
  const int th = cm == 64 ? F_TH64 : F_TH128, tw = cm == 64 ? F_TW64 : F_TW128;
  a.tiles_x = cl4::cdiv(w, tw);
  a.tiles_y = cl4::cdiv(h, th);
  a.xcd_remap = 1;
  const dim3 grid((unsigned)(n * a.tiles_x * a.tiles_y));
  hipStream_t st = cl4::as_stream(stream);
  ++g_f_launches;
  if (cm == 64 && wp) {
    if (y) hipLaunchKernelGGL((bottleneck_fused_kernel<64, 4, 1, F_TH64, F_TW64, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bottleneck_fused_kernel<64, 4, 1, F_TH64, F_TW64, false, true>), grid, dim3(256), 0, st, a);
  } else if (cm == 64) {
    if (y) hipLaunchKernelGGL((bottleneck_fused_kernel<64, 4, 1, F_TH64, F_TW64, true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((bottleneck_fused_kernel<64, 4, 1, F_TH64, F_TW64, false, false>), grid, dim3(256), 0, st, a);
  } else {
    if (y) hipLaunchKernelGGL((bottleneck_fused_kernel<128, 8, 2, F_TH128, F_TW128, true, false>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((bottleneck_fused_kernel<128, 8, 2, F_TH128, F_TW128, false, false>), grid, dim3(512), 0, st, a);
  }
  return cl4::launch_status();
}

extern "C" int cl4_bottleneck_fused(const void* xs, void* ys, float* y, int n_f32, const void* w1, const void* w2, const void* w3, const float* aff,
                                    int n, int cm, int h, int w, float slope1, float slope2, float slope_out, void* stream) {
  return fused_launch(xs, ys, y, n_f32, w1, w2, w3, nullptr, aff, n, cm, h, w, slope1, slope2, slope_out, stream);
}

extern "C" int cl4_bottleneck_fused_proj(const void* xs, void* ys, float* y, int n_f32, const void* w1, const void* w2, const void* w3, const void* wp,
                                         const float* aff, int n, int cm, int h, int w, float slope1, float slope2, float slope_out, void* stream) {
  CL4_REQUIRE(wp, "bottleneck_fused_proj: null projection weight");
  return fused_launch(xs, ys, y, n_f32, w1, w2, w3, wp, aff, n, cm, h, w, slope1, slope2, slope_out, stream);
}
