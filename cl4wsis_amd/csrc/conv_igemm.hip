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
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;       // K-tile of the forward kernel
constexpr int NTHREADS = 256;

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
  int npix;     // n*hout*wout
};

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

  // ---- fused epilogue -------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int p = n0 + wn * WTN + j * 32 + lr;
    if (p >= a.npix) continue;
    const int n = p / HWo;
    const int hw = p - n * HWo;
    const long ybase = ((long)n * a.out_ctot + a.out_coff) * HWo + hw;
    const long rbase = (long)n * a.cout * HWo + hw;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co >= a.cout) continue;
        float v = acc[i][j][r];
        if (a.pre) v += a.pre[rbase + (long)co * HWo];
        const float sc = a.scale ? a.scale[co] : 1.f;
        const float sh = a.shift ? a.shift[co] : 0.f;
        v = cl4::apply_act(v * sc + sh, a.act1, a.slope1);
        if (a.post) v += a.post[rbase + (long)co * HWo];
        v = cl4::apply_act(v, a.act2, a.slope2);
        a.y[ybase + (long)co * HWo] = v;
      }
    }
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

template <int KS>
int launch_fwd(const ConvArgs& a, hipStream_t st) {
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

template <int KS>
int launch_wgrad(WgradArgs& a, hipStream_t st) {
  const bool small_m = a.cout <= 64;
  const int BMv = small_m ? 64 : 128;
  const int gm = cl4::cdiv(a.cout, BMv), gn = cl4::cdiv(a.Nj, 128);
  // Split K so that the grid has >= ~4 workgroups per CU, each slice a multiple of BKP.
  int tiles = cl4::cdiv(a.npix, BKP);
  int want = cl4::cdiv(1024, gm * gn);
  if (want < 1) want = 1;
  if (want > tiles) want = tiles;
  int tiles_per = cl4::cdiv(tiles, want);
  a.chunk = tiles_per * BKP;
  int gz = cl4::cdiv(a.npix, a.chunk);
  dim3 g(gm, gn, gz);
  if (small_m)
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

extern "C" int cl4_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           float* scale, float* shift, int c, void* stream) {
  CL4_REQUIRE(mean && var && scale && shift && c > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cl4::cdiv(c, 256)), dim3(256), 0, cl4::as_stream(stream), gamma, beta, mean,
                     var, eps, scale, shift, c);
  return cl4::launch_status();
}

extern "C" int cl4_conv2d_fwd(const cl4_conv_desc* d, const float* x, const float* wp, const float* scale,
                              const float* shift, const float* pre, const float* post, float* y, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  CL4_REQUIRE(x && wp && y, "conv2d_fwd: null tensor");
  CL4_REQUIRE(d->precision == 0, "conv2d_fwd: precision %d not built", d->precision);
  ConvArgs a;
  a.x = x; a.wp = wp; a.scale = scale; a.shift = shift; a.pre = pre; a.post = post; a.y = y;
  a.n = d->n; a.cin = d->cin; a.hin = d->hin; a.win = d->win; a.cout = d->cout;
  a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.hout = d->hout; a.wout = d->wout;
  a.in_ctot = d->in_ctot; a.in_coff = d->in_coff; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
  a.act1 = d->act1; a.act2 = d->act2; a.slope1 = d->slope1; a.slope2 = d->slope2;
  a.K = d->cin * d->kh * d->kw;
  a.npix = d->n * d->hout * d->wout;
  hipStream_t st = cl4::as_stream(stream);
  switch (d->kh) {
    case 1: return launch_fwd<1>(a, st);
    case 3: return launch_fwd<3>(a, st);
    default: return launch_fwd<7>(a, st);
  }
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
