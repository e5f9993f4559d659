// Persistent, software-pipelined planes kernel for the short-K 1x1 layers whose time is their epilogue ("P" kernel, r03):
// the bottleneck-closing convolutions 256 -> 1024 (res4), 512 -> 2048 (res5), 128 -> 512 (res3) with a planes residual and a planes
// output. In conv_s such a workgroup spends 15k cycles in its K loop (8 K-tiles), then 21-26k cycles streaming 128 KB of residual and
// output — strictly one after the other, and the two workgroups of a CU do it in lock-step. Delaying half of them (r02e, r03:
// profiles/r03_conv_s_stagger.txt) changes nothing: each workgroup's phases are LATENCY-bound chains (memory round trips of a 4-wave
// workgroup), so what has to overlap is one workgroup's own epilogue with its own next K loop. That needs the finished accumulators
// to stay out of the way of the next tile: here they stay in REGISTERS.
//
//   * one 512-thread workgroup per CU (8 waves = 2 row halves x 4 column quarters of a 128 x 128 tile, 64 x 32 per wave: 32 accumulator
//     registers), persistent: workgroup w walks the tiles w, w + G, w + 2G, ...
//   * when tile t's K loop ends, its accumulators are copied to `prev` (32 registers) and tile t+1's loads are already on their way;
//     the epilogue of tile t runs in four chunks (one 16-row block each) inside K-steps 0..3 of tile t+1, straight from registers:
//     lane (q = l >> 4, p = l & 15) of an MFMA block holds channels 4q..4q+3 of pixel p, i.e. one HALF of a planes chunk (8 channels of
//     a pixel, 16 bytes) — it loads / stores 8 bytes per plane, and the two lanes of a chunk make full 16-byte chunks, 16 pixels = 256
//     contiguous bytes per (group, plane). No LDS parking, no barrier for the epilogue; LDS holds the two K-tile stages only.
//   * vector-memory accounting (vmcnt is in order): per K-step a thread issues 4 operand loads; during the pipelined steps also 4 residual
//     loads (two steps ahead of their use) and 4 stores. The wait at the start of step s is the number of operations YOUNGER than the
//     operand loads it needs: 4, 12, 12, 8, 12, 8, 4, 4, ... (derivation next to the K loop).
//
// Arithmetic, operand formats, LDS image and MFMA fragment mapping are those of conv_s.hip (16x16x32, lo*hi + hi*lo + hi*hi).
// Constraints (checked by the launcher): 1x1, stride 1, pad 0, cin % 32 == 0, K >= 128, cout % 128 == 0, planes output only, planes
// residual or none, no `pre`, no K split.
#include <type_traits>
#include "conv_args.h"
#include "planes_fmt.h"

namespace {

using cl4::ConvArgs;

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, P = 2, KG = 4;
constexpr int A_CHUNKS = KG * P * BM, B_CHUNKS = KG * P * BN, STG = A_CHUNKS + B_CHUNKS;
constexpr unsigned OOB = 0x80000000u;
constexpr int THREADS = 512;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  f32x2v t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float sub_f32(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct TileCtx {            // what the epilogue of a finished tile needs (kept while the next tile is multiplied)
  int m0;
  unsigned pix[2];          // chunk index (pn * c8 * 2 * HWo + phw) of this lane's two pixel columns, or OOB when the pixel does not exist
};

template <bool RES>
__global__ __launch_bounds__(THREADS, 2) void conv_p_bf16x3(ConvArgs a) {
  __shared__ u32x4 lds[2 * STG];                           // 64 KB: two K-tile stages
  __shared__ float affine[2][2 * BM];                      // folded-ABN scale / shift of the tile being multiplied and of the one being written

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;                 // 64-row half, 32-column quarter
  const int l15 = lane & 15, q = lane >> 4;
  const int HW = a.hout * a.wout;                          // 1x1 / stride 1: input and output maps coincide
  const int nk = a.K >> 5;
  const int c8in = a.in_ctot >> 3, c8out = a.out_ctot >> 3, c8res = a.cout >> 3;

  // XCD-aware walk: consecutive logical tiles (M fastest) share their pixel operand through one XCD's L2
  const unsigned G = gridDim.x, id = blockIdx.x;
  const unsigned q8 = G >> 3, r8 = G & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const int ntiles = a.ntile_l;

  // staging role: thread = (row / column tid & 127, chunk-row pair cr = tid >> 7): chunk rows cr and cr + 4 of the 8 (2 * k-group + plane)
  const int col = tid & 127, cr = tid >> 7;
  const int plane = cr & 1, kg0 = cr >> 1;
  const unsigned x_bytes = (unsigned)((long)a.n * a.in_ctot * HW * 4);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, x_bytes, 0x00020000);
  const unsigned w_bytes = (unsigned)((long)(a.K / 8) * P * a.cout * 16);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, w_bytes, 0x00020000);
  const unsigned y_bytes = (unsigned)((long)a.n * a.out_ctot * HW * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.ys, 0, y_bytes, 0x00020000);
  const unsigned r_bytes = RES ? (unsigned)((long)a.n * a.cout * HW * 4) : 0u;
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.post_s : a.xs), 0, r_bytes, 0x00020000);

  // per-tile state of the tile being multiplied
  int m0 = 0, n0 = 0;
  unsigned a_voff = OOB, b_voff = OOB;                     // byte offsets of this thread's weight row / pixel column (k-group and plane added per load)
  auto begin_tile = [&](int tile) {
    const unsigned by_ = (unsigned)tile / (unsigned)a.gx, bx_ = (unsigned)tile - by_ * (unsigned)a.gx;
    m0 = bx_ * BM;
    n0 = by_ * BN;
    a_voff = (unsigned)(m0 + col) * 16u;                   // cout % 128 == 0: every row exists
    const int p = n0 + col;
    if (p < a.npix) {
      const int pn = p / HW, phw = p - pn * HW;
      b_voff = (unsigned)((((long)pn * c8in + (a.in_coff >> 3)) * 2 + plane) * HW + phw) * 16u;
    } else b_voff = OOB;
  };
  auto load_tile = [&](u32x4 (&ar)[2], u32x4 (&br)[2], int kt) {
    const unsigned keep = (unsigned)(-(int)(kt < nk));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kgg = kt * 4 + kg0 + 2 * i;
      const int soff = __builtin_amdgcn_readfirstlane((kgg * P + plane) * a.cout * 16);
      ar[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (a_voff & keep) | (OOB & ~keep), soff, 0);
    }
    const unsigned bk = keep & (unsigned)(-(int)(b_voff != OOB));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned co = (unsigned)__builtin_amdgcn_readfirstlane((kt * 4 + kg0 + 2 * i) * 2 * HW * 16);
      br[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ((b_voff + co) & bk) | (OOB & ~bk), 0, 0);
    }
  };
  auto storeAB = [&](const u32x4 (&ar)[2], const u32x4 (&br)[2], int stage) {
    u32x4* As = lds + stage * STG;
    u32x4* Bs = As + A_CHUNKS;
#pragma unroll
    for (int i = 0; i < 2; ++i) As[(2 * (kg0 + 2 * i) + plane) * BM + col] = ar[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) Bs[(2 * (kg0 + 2 * i) + plane) * BN + col] = br[i];
  };

  f32x4v acc[4][2], prev[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  };
  bf16x8 bh16[2], bl16[2];
  auto mfma_half = [&](int stage, int s2) {                // row blocks 2*s2, 2*s2+1 x both column blocks; B fragments read by half 0
    const u32x4* As = lds + stage * STG + wm * 64 + l15;
    const u32x4* Bs = lds + stage * STG + A_CHUNKS + wn * 32 + l15;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, As[(q * P + 0) * BM + (2 * s2 + i) * 16]);
      al[i] = __builtin_bit_cast(bf16x8, As[(q * P + 1) * BM + (2 * s2 + i) * 16]);
    }
    if (s2 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh16[j] = __builtin_bit_cast(bf16x8, Bs[(q * P + 0) * BN + j * 16]);
        bl16[j] = __builtin_bit_cast(bf16x8, Bs[(q * P + 1) * BN + j * 16]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4v& c = acc[2 * s2 + i][j];
        c = plfmt::mfma16_x3(ah[i], al[i], bh16[j], bl16[j], c);
      }
  };

  // ---- epilogue of the PREVIOUS tile, chunk c = row block c of the wave's 64 x 32 part: channels m0 + wm*64 + c*16 + 4q .. +3 of the
  // pixels n0 + wn*32 + j*16 + l15 (j = 0, 1). Half-chunk address: ((pix + (g*2 + plane) * HW) * 16 + (q & 1) * 8 bytes, g = channel group.
  TileCtx pv;
  pv.m0 = 0; pv.pix[0] = pv.pix[1] = OOB;
  int pv_par = 0;                                          // which half of `affine` belongs to the previous tile
  unsigned long long rres[2][2][2];                        // residual half-chunks: [buffer c & 1][j][plane]. 64-bit scalars, not 2-vectors: element 1 of a
                                                           // vector made from the b64 load builtin's result reads element 0 with this hipcc (FINDINGS.md finding 15, third form)
  auto ctx_of_current = [&]() {
    TileCtx t;
    t.m0 = m0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = n0 + wn * 32 + j * 16 + l15;
      if (p < a.npix) {
        const int pn = p / HW, phw = p - pn * HW;
        t.pix[j] = (unsigned)phw | ((unsigned)pn << 20);                              // phw < 2^20 pixels per map, pn < 2^12 images
      } else t.pix[j] = OOB;
    }
    return t;
  };
  auto half_off = [&](unsigned pix, int c8tot, int coff8, int g, int pl) -> unsigned {   // byte offset of this lane's half chunk
    const unsigned pn = pix >> 20, phw = pix & 0xFFFFFu;
    return (unsigned)(((((long)pn * c8tot + coff8 + g) * 2 + pl) * HW + phw) * 16) + (unsigned)(q & 1) * 8u;
  };
  auto issue_res = [&](auto cc) {                          // residual loads of chunk c into buffer c & 1
    constexpr int c = decltype(cc)::value;
    if constexpr (RES) {
      const int g = ((pv.m0 + wm * 64 + c * 16) >> 3) + (q >> 1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const unsigned off = pv.pix[j] != OOB ? half_off(pv.pix[j], c8res, 0, g, pl) : OOB;
          rres[c & 1][j][pl] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(rrsrc, off, 0, 0));
        }
    }
  };
  auto finish_chunk = [&](auto cc) {                       // arithmetic + 4 stores of chunk c
    constexpr int c = decltype(cc)::value;
    const int chl = wm * 64 + c * 16 + q * 4;              // first of this lane's 4 channels inside the tile
    const int g = ((pv.m0 + wm * 64 + c * 16) >> 3) + (q >> 1);
    const float* aff = affine[pv_par];
    float sc[4], sh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { sc[r] = aff[chl + r]; sh[r] = aff[BM + chl + r]; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u = (prev[c][j][r] * plfmt::ACC) * sc[r] + sh[r];      // (x 1 in a bf16 build; the fp16x3 operand scales otherwise)
        v[r] = u > 0.f ? u : u * a.neg1;
      }
      if constexpr (RES) {
        const unsigned h0 = (unsigned)rres[c & 1][j][0], h1 = (unsigned)(rres[c & 1][j][0] >> 32);
        const unsigned l0 = (unsigned)rres[c & 1][j][1], l1 = (unsigned)(rres[c & 1][j][1] >> 32);
        float t0, t1, t2, t3;
        plfmt::join2s(h0, l0, t0, t1);
        plfmt::join2s(h1, l1, t2, t3);
        v[0] += t0; v[1] += t1; v[2] += t2; v[3] += t3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.neg2;
      u32x2 hi, lo;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned h, l;
        plfmt::split2s(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h;
        lo[i] = l;
      }
      const bool ok = pv.pix[j] != OOB;
      const unsigned o0 = ok ? half_off(pv.pix[j], c8out, a.out_coff >> 3, g, 0) : OOB;
      const unsigned o1 = ok ? half_off(pv.pix[j], c8out, a.out_coff >> 3, g, 1) : OOB;
      __builtin_amdgcn_raw_buffer_store_b64(hi, yrsrc, o0, 0, 0);     // out-of-range offsets are dropped by the descriptor
      __builtin_amdgcn_raw_buffer_store_b64(lo, yrsrc, o1, 0, 0);
    }
  };

  // ---- K loop of one tile. WITH_PREV: K-steps 0..3 carry the four epilogue chunks of the previous tile ------------------------------------
  // vmcnt per thread, in issue order, with a previous tile pending:
  //   before the loop: R0 R1 (4 + 4 residual loads) | t0 t1 (4 + 4) | wait(4) -> R0, R1, t0 landed | t2 (4)
  //   step 0: wait(4)  [t1]           then t3 (4), chunk 0 stores S0 (4), R2 (4)  -> in flight: t2 t3 S0 R2
  //   step 1: wait(12) [t2]           then t4, S1, R3                           -> t3 S0 R2 t4 S1 R3
  //   step 2: wait(12) [t3 and R2]    then t5, S2                               -> t4 S1 R3 t5 S2
  //   step 3: wait(8)  [t4 and R3]    then t6, S3                               -> t5 S2 t6 S3
  //   step 4: wait(12) [t5]           then t7                                   -> S2 t6 S3 t7
  //   step 5: wait(8)  [t6]           then t8                                   -> S3 t7 t8
  //   step 6..: wait(4)                                                          (steady state: one tile of operand loads in flight)
  //   (a chunk's stores come BEFORE the loads that refill its residual buffer: chunk c and chunk c + 2 share one)
  // without a previous tile (first tile of the workgroup): wait(4) everywhere. A store-less launch (RES = false) issues no R loads:
  // the counts are upper bounds then (waiting for fewer younger operations than exist is always safe, never too few).
  u32x4 ar[2][2], br[2][2];
  using NONE = std::integral_constant<int, -1>;
  auto step = [&](auto waitc, u32x4 (&a1)[2], u32x4 (&b1)[2], int t, int stage, auto chunkc, auto rnextc) {
    wait_vmcnt<decltype(waitc)::value>();                  // (the compiler adds its own, exact waits where a loaded register is consumed; these keep the prefetch distance)
    mfma_half(stage, 0);
    storeAB(a1, b1, stage ^ 1);
    load_tile(a1, b1, t + 3);
    if constexpr (decltype(chunkc)::value >= 0) finish_chunk(chunkc);       // consumes residual buffer chunk & 1 ...
    if constexpr (decltype(rnextc)::value >= 0) issue_res(rnextc);         // ... which the loads of chunk + 2 then refill
    mfma_half(stage, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  using C3 = std::integral_constant<int, 3>;
  using W4 = std::integral_constant<int, 4>;
  using W8 = std::integral_constant<int, 8>;
  using W12 = std::integral_constant<int, 12>;
  using WS1 = std::integral_constant<int, RES ? 12 : 8>;     // step 1: t3 + S0 (+ R2) younger than t2
  using WS2 = std::integral_constant<int, 12>;               // step 2: t4 + S1 + R3 younger than R2 (RES) / S0 + t4 + S1 younger than t3
  using WS3 = std::integral_constant<int, RES ? 8 : 12>;     // step 3: t5 + S2 younger than R3 (RES) / S1 + t5 + S2 younger than t4
  using R2 = std::integral_constant<int, RES ? 2 : -1>;
  using R3 = std::integral_constant<int, RES ? 3 : -1>;

  bool have_prev = false;
  int par = 0;
  for (int tile = (int)lid; tile < ntiles; tile += (int)G) {
    begin_tile(tile);
    // this tile's channel affine (read in ITS epilogue, i.e. during the next tile): requested now, put into LDS after the K loop
    float aff_reg = 0.f;
    if (tid < 256) {
      const float* src = tid < 128 ? a.scale : a.shift;
      aff_reg = src ? src[m0 + (tid & 127)] : (tid < 128 ? 1.f : 0.f);
    }
    if (have_prev) { issue_res(C0{}); issue_res(C1{}); }
    load_tile(ar[0], br[0], 0);
    load_tile(ar[1], br[1], 1);
    wait_vmcnt<4>();
    storeAB(ar[0], br[0], 0);
    load_tile(ar[0], br[0], 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    zero_acc();
    int t = 0;
    if (have_prev) {                                       // nk >= 4 (launcher): the four chunks fit the first four K-steps
      step(W4{}, ar[1], br[1], 0, 0, C0{}, R2{});
      step(WS1{}, ar[0], br[0], 1, 1, C1{}, R3{});
      step(WS2{}, ar[1], br[1], 2, 0, C2{}, NONE{});
      step(WS3{}, ar[0], br[0], 3, 1, C3{}, NONE{});
      t = 4;
      if (t < nk) { step(W12{}, ar[1], br[1], 4, 0, NONE{}, NONE{}); ++t; }
      if (t < nk) { step(W8{}, ar[0], br[0], 5, 1, NONE{}, NONE{}); ++t; }
    }
    for (; t < nk; t += 2) {                               // t is even here (0 or 6; 4 / 5 only when the tile is finished)
      step(W4{}, ar[1], br[1], t, 0, NONE{}, NONE{});
      if (t + 1 < nk) step(W4{}, ar[0], br[0], t + 1, 1, NONE{}, NONE{});
    }
    wait_vmcnt<0>();                                       // out-of-range tail loads and the last stores: a clean count for the next tile
    if (tid < 256) affine[par][tid] = aff_reg * plfmt::XSCALE;      // the epilogue works in the planes format's scaled domain (planes_fmt.h)
    // hand the finished tile over: accumulators -> prev, its coordinates -> pv
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) prev[i][j] = acc[i][j];
    pv = ctx_of_current();
    pv_par = par;
    par ^= 1;
    have_prev = true;
  }
  if (have_prev) {                                         // drain: the last tile's epilogue, nothing left to overlap it with
    __syncthreads();                                       // the last tile's affine, written just above by other threads
    issue_res(C0{}); issue_res(C1{});
    wait_vmcnt<0>();
    finish_chunk(C0{}); finish_chunk(C1{});
    issue_res(C2{}); issue_res(C3{});
    wait_vmcnt<0>();
    finish_chunk(C2{}); finish_chunk(C3{});
  }
}

}  // namespace

namespace cl4 {

bool conv_p_eligible(const ConvArgs& a, int ks) {
  return ks == 1 && a.stride == 1 && a.pad == 0 && a.hin == a.hout && a.win == a.wout && a.cin % 32 == 0 && a.K >= 128 &&
         a.cout % 128 == 0 && a.ys && !a.y && !a.pre && !a.post && a.xs && (a.hout * a.wout) < (1 << 20) && a.n < (1 << 12) &&
         (long)a.n * a.out_ctot * a.hout * a.wout * 4 < 0x7FFFFFFFL && (long)a.n * a.cout * a.hout * a.wout * 4 < 0x7FFFFFFFL;
}

// One persistent launch over all `ntile` logical 128 x 128 tiles (a.gx, a.ntile_l set by the caller); at most one workgroup per CU.
int launch_conv_p(const ConvArgs& a, unsigned ntile, hipStream_t st) {
  const dim3 g(ntile < 256u ? ntile : 256u), b(THREADS);
  if (a.post_s) hipLaunchKernelGGL((conv_p_bf16x3<true>), g, b, 0, st, a);
  else hipLaunchKernelGGL((conv_p_bf16x3<false>), g, b, 0, st, a);
  return launch_status();
}

}  // namespace cl4
