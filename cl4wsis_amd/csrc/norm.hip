// Train-mode nn.BatchNorm2d (+ReLU) of the Panoptic-DeepLab instance decoder / heads
// (modules/panoptic_deeplab.py:26-28,44-46,74-76,108-110,121-123,161-163) — forward statistics,
// normalise+activation, and the backward reductions / input gradient. HBM-bound: every pass reads
// the activation once with the pixel index on the lane; per-channel sums are kept in fp64 so the
// single-pass variance is as accurate as PyTorch's two-pass one.
#include "common.h"
#include "planes_fmt.h"

namespace {

constexpr int SPLIT = 32;  // workgroups per channel for the statistics / backward reductions

struct BnShape {
  int n, c, hw, ctot, coff;  // x is a channel slice [coff, coff+c) of a tensor with ctot channels
};

__device__ __forceinline__ void block_reduce2(double& a, double& b) {
  __shared__ double sa[4], sb[4];
  a = cl4::wave_sum_d(a);
  b = cl4::wave_sum_d(b);
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
  __syncthreads();
  a = sa[0] + sa[1] + sa[2] + sa[3];
  b = sb[0] + sb[1] + sb[2] + sb[3];
}

// Reduction grids (r05): (c, nsplit) with nsplit <= SPLIT chosen so that a thread owns >= 8 elements where the tensor allows it — at 16 images
// of 33 x 33 the fixed c x 32 grid left TWO elements per thread under a double-precision block reduction (res4: 2.0 TB/s, 8 ms per phase-1
// step in these two kernels) — and four independent loads per thread and iteration instead of one. Consumers still read SPLIT partials per
// channel: the entries no workgroup owns are written as zeros by the workgroups that exist.
static int reduction_split(int per_c, int c) {
  static const bool fixed = [] { const char* e = getenv("CL4_BN_SPLIT_FIXED"); return e && atoi(e) != 0; }();      // A/B: the r01-r04 grid
  if (fixed) return SPLIT;
  int sp = per_c / 2048;
  if (sp > SPLIT) sp = SPLIT;
  if (sp < 1) sp = 1;
  while (sp < SPLIT && (long)c * sp < 1024) sp *= 2;      // >= ~4 workgroups per CU
  return sp < SPLIT ? sp : SPLIT;
}

// grid (c, nsplit): partial[(c*SPLIT + s)*2 + {0,1}] = {sum x, sum x^2} over this block's share.
__global__ void bn_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, BnShape s) {
  const int c = blockIdx.x, sp = blockIdx.y;
  const int per_c = s.n * s.hw;                // checked on the host: < 2^31
  const int step = gridDim.y * blockDim.x;
  double a = 0.0, b = 0.0;
  const float* xc = x + (long)(s.coff + c) * s.hw;
  const long img = (long)s.ctot * s.hw;
  for (long i = sp * blockDim.x + threadIdx.x; i < per_c; i += 4L * step) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long j = i + (long)k * step;
      const int n = j < per_c ? (int)j / s.hw : 0, p = (int)j - n * s.hw;
      v[k] = j < per_c ? xc[n * img + p] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a += (double)v[k];
      b += (double)v[k] * (double)v[k];
    }
  }
  block_reduce2(a, b);
  if (threadIdx.x == 0) {
    partial[((long)c * SPLIT + sp) * 2 + 0] = a;
    partial[((long)c * SPLIT + sp) * 2 + 1] = b;
    for (int z = sp + gridDim.y; z < SPLIT; z += gridDim.y) {
      partial[((long)c * SPLIT + z) * 2 + 0] = 0.0;
      partial[((long)c * SPLIT + z) * 2 + 1] = 0.0;
    }
  }
}

// {sum a, sum b} of channel c over its SPLIT partials, computed by every wave for itself: lanes 0 .. SPLIT-1 load one partial each,
// a butterfly adds them (r03: the per-channel "final" kernels between the reduction and the elementwise pass are gone — 36 launches
// of ~6 us per step on the critical path; every workgroup of the elementwise pass redoes this 32-term sum, in one fixed order)
__device__ __forceinline__ void channel_sums(const double* __restrict__ partial, int c, double& a, double& b, int nparts = SPLIT) {
  const int lane = threadIdx.x & 63;
  a = 0.0;
  b = 0.0;
  for (int i = lane; i < nparts; i += 64) {              // nparts = SPLIT (the reduction kernels of this file), or what a producer wrote
    a += partial[((long)c * nparts + i) * 2 + 0];
    b += partial[((long)c * nparts + i) * 2 + 1];
  }
  a = cl4::wave_sum_d(a);
  b = cl4::wave_sum_d(b);
}

// mean / biased var; running stats updated with the unbiased variance (PyTorch semantics, momentum m).
__global__ void bn_stats_final_kernel(const double* __restrict__ partial, float* __restrict__ mean, float* __restrict__ var,
                                      float* running_mean, float* running_var, float momentum, long count, int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < SPLIT; ++s) {
    a += partial[((long)i * SPLIT + s) * 2 + 0];
    b += partial[((long)i * SPLIT + s) * 2 + 1];
  }
  double m = a / (double)count;
  double v = b / (double)count - m * m;
  if (v < 0.0) v = 0.0;
  mean[i] = (float)m;
  var[i] = (float)v;
  if (running_mean) {
    double unb = count > 1 ? v * (double)count / (double)(count - 1) : v;
    running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)m;
    running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unb;
  }
}

// bn_stats_final_kernel for partials that came from somewhere else (r05: the producing convolution's epilogue, cl4_conv2d_fwd_planes_stats:
// `nparts` = its pixel tiles, up to ~1,850 per channel): one wave per channel folds them (channel_sums) and finalises.
__global__ __launch_bounds__(256) void bn_stats_from_partials_kernel(const double* __restrict__ partial, int nparts, float* __restrict__ mean,
                                                                     float* __restrict__ var, float* running_mean, float* running_var,
                                                                     float momentum, long count, int c) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  double a, b;
  channel_sums(partial, ch, a, b, nparts);
  if ((threadIdx.x & 63) != 0) return;
  double m = a / (double)count;
  double v = b / (double)count - m * m;
  if (v < 0.0) v = 0.0;
  mean[ch] = (float)m;
  var[ch] = (float)v;
  if (running_mean) {
    double unb = count > 1 ? v * (double)count / (double)(count - 1) : v;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
  }
}

// Cross-rank (InPlaceABNSync) form: the SPLIT partials are folded to sums[c][2] doubles, which the host all-reduces over the
// ranks before the statistics are finalised with the global element count.
__global__ void bn_fold_partials_kernel(const double* __restrict__ partial, double* __restrict__ sums, int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < SPLIT; ++s) {
    a += partial[((long)i * SPLIT + s) * 2 + 0];
    b += partial[((long)i * SPLIT + s) * 2 + 1];
  }
  sums[2 * i] = a;
  sums[2 * i + 1] = b;
}

__global__ void bn_stats_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ var,
                                          float* running_mean, float* running_var, float momentum, double count, int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  double m = sums[2 * i] / count;
  double v = sums[2 * i + 1] / count - m * m;
  if (v < 0.0) v = 0.0;
  mean[i] = (float)m;
  var[i] = (float)v;
  if (running_mean) {
    double unb = count > 1.0 ? v * count / (count - 1.0) : v;
    running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)m;
    running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unb;
  }
}

__global__ void bn_bwd_sums_to_float_kernel(const double* __restrict__ sums, float* __restrict__ sum_dz, float* __restrict__ sum_dz_xh,
                                            int c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  sum_dz[i] = (float)sums[2 * i];
  sum_dz_xh[i] = (float)sums[2 * i + 1];
}

// y = act((x - mean) * rstd * gamma + beta); optional inverted-dropout mask (uint8 keep flags, scale 1/(1-p)).
// grid = (n*c planes, chunks of hw): per-plane constants are computed once per workgroup, no per-element divisions.
__global__ void bn_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int act, float slope, BnShape s, int y_ctot,
                                int y_coff, const uint8_t* __restrict__ keep, float keep_scale) {
  const int plane = blockIdx.x;                // n * c + c  (x: up to 2^31 planes; y: chunks of hw)
  const int n = plane / s.c, c = plane - n * s.c;
  const float rstd = 1.f / sqrtf(var[c] + eps);
  const float mu = mean[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  const float* xp = x + ((long)n * s.ctot + s.coff + c) * s.hw;
  float* yp = y + ((long)n * y_ctot + y_coff + c) * s.hw;
  const uint8_t* kp = keep ? keep + (long)plane * s.hw : nullptr;
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < s.hw; p += gridDim.y * blockDim.x) {
    float v = (xp[p] - mu) * rstd * g + bb;
    v = cl4::apply_act(v, act, slope);
    if (kp) v = kp[p] ? v * keep_scale : 0.f;
    yp[p] = v;
  }
}

// bn_apply_kernel for a contiguous [n][c][hw] tensor with c % 8 == 0 that ALSO writes the two-plane twin of y (planes_fmt.h) — the operand
// the next convolution multiplies (r05: the fp16x3 route of the differentiated networks reads it instead of converting y in a pass of its
// own). One thread per (image, 8-channel group, pixel): 8 strided dword loads and stores (coalesced along pixels), two 16-byte chunk stores.
// The per-element expression is bn_apply_kernel's, so y is bit-identical to the plain launch.
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
// `post` / act2 (r05): y = act2(act(bn(x)) + post) — the residual add + activation of modules/residual.py:91-97 in the same pass (the bn3
// output of a bottleneck is then never written). `ys` may be null (no twin wanted).
__global__ void bn_apply_planes_kernel(const float* __restrict__ x, float* __restrict__ y, u32x4n* __restrict__ ys, const float* __restrict__ mean,
                                       const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float eps, int act, float slope, int n, int c, int hw, const float* __restrict__ post, int act2, float slope2) {
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g = (int)(r - (unsigned)im * (unsigned)c8);
    const long base = ((long)im * c + g * 8) * hw + px;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = g * 8 + j;
      const float rstd = 1.f / sqrtf(var[ch] + eps);
      const float mu = mean[ch], gm = gamma ? gamma[ch] : 1.f, bb = beta ? beta[ch] : 0.f;
      v[j] = cl4::apply_act((x[base + (long)j * hw] - mu) * rstd * gm + bb, act, slope);
    }
    if (post) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = cl4::apply_act(v[j] + post[base + (long)j * hw], act2, slope2);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) y[base + (long)j * hw] = v[j];
    if (!ys) continue;
    u32x4n hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h_, l_;
      plfmt::split2(v[2 * q], v[2 * q + 1], h_, l_);
      hi[q] = h_;
      lo[q] = l_;
    }
    u32x4n* dst = ys + (((long)im * c8 + g) * 2) * hw + px;
    dst[0] = hi;
    dst[hw] = lo;
  }
}

// bn_apply_kernel with the statistics finalised in place: mean / var of the plane's channel from the partial sums; the first
// workgroup of image 0 also publishes them (mean, var: the backward pass reads them) and updates the running statistics.
__global__ void bn_apply_fused_kernel(const float* __restrict__ x, float* __restrict__ y, const double* __restrict__ partial,
                                      float* __restrict__ mean, float* __restrict__ var, float* running_mean, float* running_var,
                                      float momentum, long count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float eps, int act, float slope, BnShape s, int y_ctot, int y_coff,
                                      const uint8_t* __restrict__ keep, float keep_scale, int nparts) {
  const int plane = blockIdx.x;
  const int n = plane / s.c, c = plane - n * s.c;
  double a, b;
  channel_sums(partial, c, a, b, nparts);
  const double md = a / (double)count;
  double vd = b / (double)count - md * md;
  if (vd < 0.0) vd = 0.0;
  const float mu = (float)md, vf = (float)vd;
  if (n == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    mean[c] = mu;
    var[c] = vf;
    if (running_mean) {
      const double unb = count > 1 ? vd * (double)count / (double)(count - 1) : vd;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  const float rstd = 1.f / sqrtf(vf + eps);
  const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  const float* xp = x + ((long)n * s.ctot + s.coff + c) * s.hw;
  float* yp = y + ((long)n * y_ctot + y_coff + c) * s.hw;
  const uint8_t* kp = keep ? keep + (long)plane * s.hw : nullptr;
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < s.hw; p += gridDim.y * blockDim.x) {
    float v = (xp[p] - mu) * rstd * g + bb;
    v = cl4::apply_act(v, act, slope);
    if (kp) v = kp[p] ? v * keep_scale : 0.f;
    yp[p] = v;
  }
}

// Derivative of the fused activation at pre-activation z (ReLU: [z > 0]; leaky: slope where z <= 0).
__device__ __forceinline__ float act_grad(float d, float z, int act, float slope) {
  if (act == CL4_ACT_RELU) return z > 0.f ? d : 0.f;
  if (act == CL4_ACT_LEAKY) return z > 0.f ? d : d * slope;
  return d;
}

// Backward reductions. dz = dy * keepscale * act'(z) with z recomputed from x.
//   partial = {sum dz, sum dz * xhat}
__global__ void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, double* __restrict__ partial,
                                      const float* __restrict__ mean, const float* __restrict__ var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                      float slope, BnShape s, int dy_ctot, int dy_coff, const uint8_t* __restrict__ keep,
                                      float keep_scale, float* __restrict__ maxima = nullptr) {
  const int c = blockIdx.x, sp = blockIdx.y;
  const int per_c = s.n * s.hw;
  const int step = gridDim.y * blockDim.x;       // (c, nsplit) grid: reduction_split
  const float mu = mean[c], rstd = 1.f / sqrtf(var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  double a = 0.0, b = 0.0;
  float md = 0.f, mx = 0.f;          // max |dz|, max |xhat| of this workgroup's share (r05: the bound behind cl4_bn_sync_bwd_apply_planes)
  const float* xc = x + (long)(s.coff + c) * s.hw;
  const float* dc = dy + (long)(dy_coff + c) * s.hw;
  const uint8_t* kc = keep ? keep + (long)c * s.hw : nullptr;
  const long ximg = (long)s.ctot * s.hw, dimg = (long)dy_ctot * s.hw, kimg = (long)s.c * s.hw;
  for (long i = sp * blockDim.x + threadIdx.x; i < per_c; i += 4L * step) {
    float xv[4], dv[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                // four independent (x, dy) loads in flight per thread
      const long j = i + (long)k * step;
      ok[k] = j < per_c;
      const int n = ok[k] ? (int)j / s.hw : 0, p = ok[k] ? (int)j - n * s.hw : 0;
      xv[k] = ok[k] ? xc[n * ximg + p] : 0.f;
      dv[k] = ok[k] ? dc[n * dimg + p] : 0.f;
      if (kc && ok[k]) dv[k] = kc[n * kimg + p] ? dv[k] * keep_scale : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!ok[k]) continue;
      const float xh = (xv[k] - mu) * rstd;
      const float d = act_grad(dv[k], xh * g + bb, act, slope);
      a += (double)d;
      b += (double)d * (double)xh;
      md = fmaxf(md, fabsf(d));
      mx = fmaxf(mx, fabsf(xh));
    }
  }
  block_reduce2(a, b);
  if (threadIdx.x == 0) {
    partial[((long)c * SPLIT + sp) * 2 + 0] = a;
    partial[((long)c * SPLIT + sp) * 2 + 1] = b;
    for (int z = sp + gridDim.y; z < SPLIT; z += gridDim.y) {
      partial[((long)c * SPLIT + z) * 2 + 0] = 0.0;
      partial[((long)c * SPLIT + z) * 2 + 1] = 0.0;
    }
  }
  if (maxima) {                      // (uniform branch; block_reduce2's shared arrays are not reused)
    __shared__ float sm[8];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { md = fmaxf(md, __shfl_xor(md, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = md; sm[4 + (threadIdx.x >> 6)] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
      maxima[((long)c * SPLIT + sp) * 2 + 0] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
      maxima[((long)c * SPLIT + sp) * 2 + 1] = fmaxf(fmaxf(sm[4], sm[5]), fmaxf(sm[6], sm[7]));
      for (int z = sp + gridDim.y; z < SPLIT; z += gridDim.y) {
        maxima[((long)c * SPLIT + z) * 2 + 0] = 0.f;
        maxima[((long)c * SPLIT + z) * 2 + 1] = 0.f;
      }
    }
  }
}

// `sign_src` (r05; InPlaceABN: the kernels' gamma is |weight| + eps, mapillary/inplace_abn believed behaviour, see cl4_sign_axpy): dgamma
// receives d(weight) = (weight > 0 ? 1 : -1) * d(gamma) directly — no scratch gradient, no zero fill, no cl4_sign_axpy launch per layer.
__device__ __forceinline__ float gamma_sign(const float* __restrict__ sign_src, int c) { return (sign_src && !(sign_src[c] > 0.f)) ? -1.f : 1.f; }

__global__ void bn_bwd_final_kernel(const double* __restrict__ partial, float* __restrict__ sum_dz, float* __restrict__ sum_dz_xh,
                                    float* dgamma, float* dbeta, int c, const float* __restrict__ sign_src = nullptr,
                                    double* __restrict__ sums_out = nullptr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < SPLIT; ++s) {
    a += partial[((long)i * SPLIT + s) * 2 + 0];
    b += partial[((long)i * SPLIT + s) * 2 + 1];
  }
  if (sums_out) {                    // r05: the packed vector of the cross-rank exchange (bn_fold_partials_kernel's job, one launch less)
    sums_out[2 * i] = a;
    sums_out[2 * i + 1] = b;
  }
  sum_dz[i] = (float)a;
  sum_dz_xh[i] = (float)b;
  if (dgamma) dgamma[i] += gamma_sign(sign_src, i) * (float)b;
  if (dbeta) dbeta[i] += (float)a;
}

// bn_bwd_apply_kernel with the two channel sums taken from the partials in place; the first workgroup of image 0 adds them into
// dgamma / dbeta.
__global__ void bn_bwd_apply_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                          const float* __restrict__ mean, const float* __restrict__ var,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const double* __restrict__ partial, float* dgamma, float* dbeta, float eps, int act,
                                          float slope, float invn, BnShape s, int dy_ctot, int dy_coff,
                                          const uint8_t* __restrict__ keep, float keep_scale, const float* __restrict__ sign_src = nullptr) {
  const int plane = blockIdx.x;
  const int n = plane / s.c, c = plane - n * s.c;
  double a, b;
  channel_sums(partial, c, a, b);
  if (n == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (dgamma) dgamma[c] += gamma_sign(sign_src, c) * (float)b;
    if (dbeta) dbeta[c] += (float)a;
  }
  const float rstd = 1.f / sqrtf(var[c] + eps);
  const float mu = mean[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  const float sa = (float)a * invn, sb = (float)b * invn;
  const float* xp = x + ((long)n * s.ctot + s.coff + c) * s.hw;
  float* dxp = dx + ((long)n * s.ctot + s.coff + c) * s.hw;
  const float* dp = dy + ((long)n * dy_ctot + dy_coff + c) * s.hw;
  const uint8_t* kp = keep ? keep + (long)plane * s.hw : nullptr;
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < s.hw; p += gridDim.y * blockDim.x) {
    float xh = (xp[p] - mu) * rstd;
    float d = dp[p];
    if (kp) d = kp[p] ? d * keep_scale : 0.f;
    d = act_grad(d, xh * g + bb, act, slope);
    dxp[p] = g * rstd * (d - sa - xh * sb);
  }
}

// dx = gamma * rstd * (dz - sum_dz/N - xhat * sum_dz_xh/N)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                    const float* __restrict__ mean, const float* __restrict__ var,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sum_dz, const float* __restrict__ sum_dz_xh, float eps, int act,
                                    float slope, float invn, BnShape s, int dy_ctot, int dy_coff,
                                    const uint8_t* __restrict__ keep, float keep_scale) {
  const int plane = blockIdx.x;                // grid = (n*c planes, chunks of hw), like bn_apply_kernel
  const int n = plane / s.c, c = plane - n * s.c;
  const float rstd = 1.f / sqrtf(var[c] + eps);
  const float mu = mean[c], g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
  const float sa = sum_dz[c] * invn, sb = sum_dz_xh[c] * invn;
  const float* xp = x + ((long)n * s.ctot + s.coff + c) * s.hw;
  float* dxp = dx + ((long)n * s.ctot + s.coff + c) * s.hw;
  const float* dp = dy + ((long)n * dy_ctot + dy_coff + c) * s.hw;
  const uint8_t* kp = keep ? keep + (long)plane * s.hw : nullptr;
  for (int p = blockIdx.y * blockDim.x + threadIdx.x; p < s.hw; p += gridDim.y * blockDim.x) {
    float xh = (xp[p] - mu) * rstd;
    float d = dp[p];
    if (kp) d = kp[p] ? d * keep_scale : 0.f;
    d = act_grad(d, xh * g + bb, act, slope);
    dxp[p] = g * rstd * (d - sa - xh * sb);
  }
}

// r05: the input gradient with its fp16 planes twin — what the fp16x3 input-gradient convolution of the layer in front reads instead of a
// max |dx| pass + a conversion pass over the fp32 tensor (12 bytes per element of traffic less, 4 more here). The twin needs its power-of-two
// scale BEFORE dx exists, so the scale comes from a bound: |dx| <= |gamma| rstd (max |dz| + |mean dz| + max |xhat| |mean dz xhat|) per channel,
// with the two maxima from the reduction pass that already reads x and dy. s = 2^(15 - e), B = f 2^e: B s in [2^14, 2^15) < 65504, so nothing
// saturates; the bound is within ~2x of max |dx| (the mean terms are small beside max |dz|), i.e. the same headroom as the data-derived
// 2^(14 - e) of cl4_to_planes_f16. Two steps, neither serial: one WAVE per channel folds that channel's SPLIT maxima, converts its two sums
// to float (bn_bwd_sums_to_float_kernel's job) and leaves the channel's bound in bound[c]; every workgroup of the apply kernel then folds the
// c bounds itself (8 KB at most, from L2), like to_planes_f16_kernel folds its partial maxima.
// coef[c][8] = {mean, rstd, gamma, beta, mean dz, mean dz xhat, -, -}: the apply kernel's per-channel constants as two 16-byte loads.
// One-rank form (`partial` instead of `sums`): the channel's two sums come straight from the reduction workgroups' partials (channel_sums, the
// fixed order the fused apply kernels use) and the wave also adds them into dgamma / dbeta — no fold, no final launch.
__global__ __launch_bounds__(256) void bn_bwd_prep_kernel(const double* __restrict__ sums, const float* __restrict__ maxima, const float* __restrict__ mean,
                                                          const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float invn, int c, float* __restrict__ coef, float* __restrict__ bound,
                                                          const double* __restrict__ partial = nullptr, float* dgamma = nullptr, float* dbeta = nullptr,
                                                          const float* __restrict__ sign_src = nullptr) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ch >= c) return;
  double sa_ = 0.0, sb_ = 0.0;
  if (partial) {
    channel_sums(partial, ch, sa_, sb_);
    if (lane == 0) {
      if (dgamma) dgamma[ch] += gamma_sign(sign_src, ch) * (float)sb_;
      if (dbeta) dbeta[ch] += (float)sa_;
    }
  }
  float md = 0.f, mx = 0.f;
  if (lane < SPLIT) {
    md = maxima[((long)ch * SPLIT + lane) * 2 + 0];
    mx = maxima[((long)ch * SPLIT + lane) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { md = fmaxf(md, __shfl_xor(md, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
  if (lane == 0) {
    const float a = partial ? (float)sa_ : (float)sums[2 * ch], b = partial ? (float)sb_ : (float)sums[2 * ch + 1];
    const float rstd = 1.f / sqrtf(var[ch] + eps), g = gamma ? gamma[ch] : 1.f;
    float* q = coef + (long)ch * 8;
    q[0] = mean[ch]; q[1] = rstd; q[2] = g; q[3] = beta ? beta[ch] : 0.f;
    q[4] = a * invn; q[5] = b * invn; q[6] = 0.f; q[7] = 0.f;
    bound[ch] = fabsf(g) * rstd * (md + fabsf(a) * invn + mx * fabsf(b) * invn);
  }
}

__device__ __forceinline__ float scale_from_bounds(const float* __restrict__ bound, int c) {      // the same value in every workgroup
  __shared__ float red[4];
  float bmax = 0.f;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) bmax = fmaxf(bmax, bound[ch]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bmax;
  __syncthreads();
  const float b = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 1.0001f;        // (rounding of the bound's own arithmetic)
  float sc = 1.f;
  if (b > 0.f && b < 3.0e38f) {
    int e;
    frexpf(b, &e);
    sc = ldexpf(1.f, min(max(15 - e, -100), 100));
  }
  return sc;
}

// One thread per (image, 8-channel group, pixel), like bn_apply_planes_kernel; per element the expression of bn_bwd_apply_kernel (dx is
// bit-identical to the plain launch).
__global__ void bn_bwd_apply_planes_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, u32x4n* __restrict__ dxs,
                                           const float* __restrict__ coef, int act, float slope, int n, int c, int hw, int dy_ctot, int dy_coff,
                                           const float* __restrict__ bound, float* __restrict__ slot) {
  typedef float f32x4n __attribute__((ext_vector_type(4)));
  const float sc = scale_from_bounds(bound, c);
  if (blockIdx.x == 0 && threadIdx.x == 0) slot[0] = sc;
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g8 = (int)(r - (unsigned)im * (unsigned)c8);
    const long base = ((long)im * c + g8 * 8) * hw + px;
    const long dbase = ((long)im * dy_ctot + dy_coff + g8 * 8) * hw + px;
    float xv[8], dv[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xv[j] = x[base + (long)j * hw]; dv[j] = dy[dbase + (long)j * hw]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4n* q = (const f32x4n*)(coef + (long)(g8 * 8 + j) * 8);
      const f32x4n k0 = q[0], k1 = q[1];                       // {mean, rstd, gamma, beta}, {mean dz, mean dz xhat, -, -}
      const float xh = (xv[j] - k0[0]) * k0[1];
      const float d = act_grad(dv[j], xh * k0[2] + k0[3], act, slope);
      v[j] = k0[2] * k0[1] * (d - k1[0] - xh * k1[1]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[base + (long)j * hw] = v[j];
    u32x4n hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = plfmt::sat_f16(v[2 * q] * sc), b = plfmt::sat_f16(v[2 * q + 1] * sc);
      const unsigned hp = plfmt::pk_f16(a, b);
      const plfmt::f16x2 h = __builtin_bit_cast(plfmt::f16x2, hp);
      const float ha = (float)h[0], hb = (float)h[1];       // (plain subtractions: finding 48)
      hi[q] = hp;
      lo[q] = plfmt::pk_f16(a - ha, b - hb);
    }
    u32x4n* dst = dxs + (((long)im * c8 + g8) * 2) * hw + px;
    dst[0] = hi;
    dst[hw] = lo;
  }
}

// ReLU backward for layers without BN (ASPPPooling conv -> ReLU): dx = dy * [y > 0].
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// Per-channel sum over (n, hw) of dy -> bias gradient (accumulated). grid (c, slices): the heads have 2-15 channels, so
// one workgroup per channel left 240 CUs idle on a 4 M element reduction; every slice sums in double and adds its part
// with one fp32 atomic (like the split-K weight gradients).
__global__ void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int n, int c, int hw, int ctot, int coff) {
  __shared__ double part[4];
  const int ch = blockIdx.x;
  double a = 0.0;
  const long per_c = (long)n * hw;
  for (long i = blockIdx.y * (long)blockDim.x + threadIdx.x; i < per_c; i += (long)gridDim.y * blockDim.x) {
    int nn = i / hw;
    int p = i - (long)nn * hw;
    a += (double)dy[((long)nn * ctot + coff + ch) * hw + p];
  }
  a = cl4::wave_sum_d(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&db[ch], (float)(part[0] + part[1] + part[2] + part[3]));
}

__device__ __forceinline__ uint64_t splitmix64_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Counter-based keep mask for nn.Dropout(p): element i of the stream keyed by `seed` is mix(mix(seed) + G * (i + 1)).
// The seed is hashed BEFORE the element counter is added: with z = seed + G * (i + 1) and a per-step seed advance that is a
// multiple of G, step t+1's mask would be step t's mask shifted by one element (masks of successive iterations must be
// independent draws, like nn.Dropout's).
__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, long n, uint64_t seed, const uint64_t* seed_dev, float p) {
  if (seed_dev) seed = *seed_dev;   // device-resident seed: lets a captured HIP graph draw a new mask on every replay
  const uint64_t key = splitmix64_mix(seed + 0x9E3779B97F4A7C15ull);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint64_t z = splitmix64_mix(key + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
    float u = (float)(z >> 40) * (1.f / 16777216.f);
    keep[i] = u >= p ? 1 : 0;
  }
}

// Uniform integer draws in [0, modulus): splitmix64 of (seed, element index) — RandropLoss' random old classes.
__global__ void random_labels_kernel(int* __restrict__ out, long n, uint64_t seed, const uint64_t* __restrict__ seed_dev, int modulus) {
  if (seed_dev) seed = *seed_dev * 0xD1342543DE82EF95ull + 7919;      // device-resident seed (graph replays): one counter value per step
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = (int)((z >> 33) % (uint64_t)modulus);
  }
}

int check_shape(const BnShape& s) {
  CL4_REQUIRE(s.n > 0 && s.c > 0 && s.hw > 0, "bn: non-positive dims");
  CL4_REQUIRE((long)s.n * s.hw < 2147483647L && (long)s.n * s.c < 2147483647L, "bn: n*hw / n*c overflow int32");
  CL4_REQUIRE(s.coff >= 0 && s.coff + s.c <= s.ctot, "bn: channel slice out of range");
  return CL4_OK;
}

}  // namespace

extern "C" size_t cl4_bn_workspace_bytes(int c) { return (size_t)c * SPLIT * 2 * sizeof(double); }

extern "C" int cl4_bn_train_stats(const float* x, int n, int c, int hw, int ctot, int coff, float* mean, float* var,
                                  float* running_mean, float* running_var, float momentum, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, ctot, coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && mean && var && workspace, "bn_train_stats: null pointer");
  if (workspace_bytes < cl4_bn_workspace_bytes(c)) { cl4::set_error("bn_train_stats: workspace too small"); return CL4_ENOSPC; }
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, (double*)workspace, s);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, (const double*)workspace, mean, var,
                     running_mean, running_var, momentum, (long)n * hw, c);
  return cl4::launch_status();
}

// ... and their cross-rank form: the same partials folded into the packed vector sums[c][2] that the ranks exchange (cl4_bn_sync_stats_local
// without its pass over the tensor); cl4_bn_sync_stats_final follows the all-reduce.
__global__ __launch_bounds__(256) void bn_fold_partials_n_kernel(const double* __restrict__ partial, int nparts, double* __restrict__ sums, int c) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  double a, b;
  channel_sums(partial, ch, a, b, nparts);
  if ((threadIdx.x & 63) == 0) {
    sums[2 * ch] = a;
    sums[2 * ch + 1] = b;
  }
}
extern "C" int cl4_bn_sync_stats_from_partials(const double* partial, int nparts, double* sums, int c, void* stream) {
  CL4_REQUIRE(partial && sums && nparts > 0 && c > 0, "bn_sync_stats_from_partials: bad arguments");
  hipLaunchKernelGGL(bn_fold_partials_n_kernel, dim3(cl4::cdiv(c, 4)), dim3(256), 0, cl4::as_stream(stream), partial, nparts, sums, c);
  return cl4::launch_status();
}

extern "C" int cl4_bn_stats_from_partials(const double* partial, int nparts, long count, int c, float* mean, float* var, float* running_mean,
                                          float* running_var, float momentum, void* stream) {
  CL4_REQUIRE(partial && mean && var && nparts > 0 && count > 0 && c > 0, "bn_stats_from_partials: bad arguments");
  hipLaunchKernelGGL(bn_stats_from_partials_kernel, dim3(cl4::cdiv(c, 4)), dim3(256), 0, cl4::as_stream(stream), partial, nparts, mean, var, running_mean,
                     running_var, momentum, count, c);
  return cl4::launch_status();
}

extern "C" int cl4_bn_apply(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                            const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act,
                            float slope, const uint8_t* keep, float keep_scale, void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && y && mean && var, "bn_apply: null pointer");
  CL4_REQUIRE(y_coff >= 0 && y_coff + c <= y_ctot, "bn_apply: output slice out of range");
  long total = (long)n * c * hw;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, cl4::as_stream(stream), x, y, mean,
                     var, gamma, beta, eps, act, slope, s, y_ctot, y_coff, keep, keep_scale);
  return cl4::launch_status();
}

extern "C" int cl4_bn_apply_planes(const float* x, float* y, void* ys, int n, int c, int hw, const float* mean, const float* var,
                                   const float* gamma, const float* beta, float eps, int act, float slope, const float* post, int act2,
                                   float slope2, void* stream) {
  CL4_REQUIRE(x && y && mean && var, "bn_apply_planes: null pointer");
  CL4_REQUIRE(n > 0 && c > 0 && hw > 0 && c % 8 == 0, "bn_apply_planes: c %% 8 == 0 required");
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(bn_apply_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, (u32x4n*)ys, mean, var,
                     gamma, beta, eps, act, slope, n, c, hw, post, act2, slope2);
  return cl4::launch_status();
}

extern "C" int cl4_bn_train_bwd(const float* x, const float* dy, float* dx, int n, int c, int hw, int x_ctot, int x_coff,
                                int dy_ctot, int dy_coff, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, int act, const uint8_t* keep, float keep_scale,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dx && mean && var && workspace, "bn_train_bwd: null pointer");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_train_bwd: dy slice out of range");
  size_t need = cl4_bn_workspace_bytes(c) + (size_t)2 * c * sizeof(float);
  if (workspace_bytes < need) { cl4::set_error("bn_train_bwd: workspace too small"); return CL4_ENOSPC; }
  double* partial = (double*)workspace;
  float* sums = (float*)((char*)workspace + cl4_bn_workspace_bytes(c));
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, dy, partial, mean, var, gamma, beta, eps, act,
                     0.f, s, dy_ctot, dy_coff, keep, keep_scale);
  static const bool fused = [] { const char* e = getenv("CL4_BN_FUSED_FINAL"); return !e || atoi(e) != 0; }();
  if (fused) {
    hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, st, x, dy, dx, mean,
                       var, gamma, beta, (const double*)partial, dgamma, dbeta, eps, act, 0.f, 1.f / (float)((long)n * hw), s, dy_ctot, dy_coff,
                       keep, keep_scale);
    return cl4::launch_status();
  }
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, (const double*)partial, sums, sums + c,
                     dgamma, dbeta, c);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, st, x, dy, dx, mean, var, gamma,
                     beta, sums, sums + c, eps, act, 0.f, 1.f / (float)((long)n * hw), s, dy_ctot, dy_coff, keep, keep_scale);
  return cl4::launch_status();
}

// Train-mode BatchNorm forward in two launches (r03): partial sums, then normalise + activation with the statistics finalised inside
// the elementwise pass (cl4_bn_train_stats + cl4_bn_apply are three). mean / var are outputs (the backward pass reads them).
extern "C" int cl4_bn_train_fwd(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                                float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                const float* gamma, const float* beta, float eps, int act, float slope, const uint8_t* keep,
                                float keep_scale, void* workspace, size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && y && mean && var && workspace, "bn_train_fwd: null pointer");
  CL4_REQUIRE(y_coff >= 0 && y_coff + c <= y_ctot, "bn_train_fwd: output slice out of range");
  if (workspace_bytes < cl4_bn_workspace_bytes(c)) { cl4::set_error("bn_train_fwd: workspace too small"); return CL4_ENOSPC; }
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, (double*)workspace, s);
  hipLaunchKernelGGL(bn_apply_fused_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, st, x, y,
                     (const double*)workspace, mean, var, running_mean, running_var, momentum, (long)n * hw, gamma, beta, eps, act, slope, s,
                     y_ctot, y_coff, keep, keep_scale, SPLIT);
  return cl4::launch_status();
}

// The same with the partial sums already made by the kernel that PRODUCED x (cl4_dwconv5_fwd_stats: `nparts` pairs {sum, sum of squares}
// per channel, partial[(c * nparts + i) * 2 + {0, 1}]): one launch, x is read once instead of twice.
extern "C" int cl4_bn_train_fwd_partials(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                                         float* mean, float* var, float* running_mean, float* running_var, float momentum,
                                         const float* gamma, const float* beta, float eps, int act, float slope, const uint8_t* keep,
                                         float keep_scale, const double* partial, int nparts, void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && y && mean && var && partial && nparts > 0, "bn_train_fwd_partials: null pointer");
  CL4_REQUIRE(y_coff >= 0 && y_coff + c <= y_ctot, "bn_train_fwd_partials: output slice out of range");
  hipLaunchKernelGGL(bn_apply_fused_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, cl4::as_stream(stream),
                     x, y, partial, mean, var, running_mean, running_var, momentum, (long)n * hw, gamma, beta, eps, act, slope, s, y_ctot, y_coff,
                     keep, keep_scale, nparts);
  return cl4::launch_status();
}

// ---- two-phase (cross-rank) forms: InPlaceABNSync in training mode (segmentation_module.py:17-29) ----------------------
extern "C" int cl4_bn_sync_stats_local(const float* x, int n, int c, int hw, int ctot, int coff, double* sums, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, ctot, coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && sums && workspace, "bn_sync_stats_local: null pointer");
  if (workspace_bytes < cl4_bn_workspace_bytes(c)) { cl4::set_error("bn_sync_stats_local: workspace too small"); return CL4_ENOSPC; }
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, (double*)workspace, s);
  hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, (const double*)workspace, sums, c);
  return cl4::launch_status();
}

extern "C" int cl4_bn_sync_stats_final(const double* sums, double count, float* mean, float* var, float* running_mean,
                                       float* running_var, float momentum, int c, void* stream) {
  CL4_REQUIRE(sums && mean && var && c > 0 && count > 0.0, "bn_sync_stats_final: bad arguments");
  hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, cl4::as_stream(stream), sums, mean, var,
                     running_mean, running_var, momentum, count, c);
  return cl4::launch_status();
}

extern "C" int cl4_bn_sync_bwd_local(const float* x, const float* dy, int n, int c, int hw, int x_ctot, int x_coff, int dy_ctot,
                                     int dy_coff, const float* mean, const float* var, const float* gamma, const float* beta,
                                     float eps, int act, float slope, const uint8_t* keep, float keep_scale, double* sums,
                                     float* dgamma, float* dbeta, const float* gamma_sign_src, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && mean && var && sums && workspace, "bn_sync_bwd_local: null pointer");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_sync_bwd_local: dy slice out of range");
  size_t need = cl4_bn_workspace_bytes(c) + (size_t)2 * c * sizeof(float);
  if (workspace_bytes < need) { cl4::set_error("bn_sync_bwd_local: workspace too small"); return CL4_ENOSPC; }
  double* partial = (double*)workspace;
  float* fsum = (float*)((char*)workspace + cl4_bn_workspace_bytes(c));
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, dy, partial, mean, var, gamma, beta, eps, act,
                     slope, s, dy_ctot, dy_coff, keep, keep_scale);
  // the affine gradients are this rank's share (the gradient all-reduce averages them with everything else); the same kernel folds the partials
  // into the exchange's packed vector
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, (const double*)partial, fsum, fsum + c, dgamma,
                     dbeta, c, gamma_sign_src, sums);
  return cl4::launch_status();
}

extern "C" int cl4_bn_sync_bwd_apply(const float* x, const float* dy, float* dx, int n, int c, int hw, int x_ctot, int x_coff,
                                     int dy_ctot, int dy_coff, const float* mean, const float* var, const float* gamma,
                                     const float* beta, float eps, int act, float slope, const uint8_t* keep, float keep_scale,
                                     const double* sums, double count, void* workspace, size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, x_ctot, x_coff};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dx && mean && var && sums && workspace && count > 0.0, "bn_sync_bwd_apply: bad arguments");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_sync_bwd_apply: dy slice out of range");
  if (workspace_bytes < (size_t)2 * c * sizeof(float)) { cl4::set_error("bn_sync_bwd_apply: workspace too small"); return CL4_ENOSPC; }
  float* fsum = (float*)workspace;
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_bwd_sums_to_float_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, sums, fsum, fsum + c, c);
  long total = (long)n * c * hw;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, st, x, dy, dx, mean, var, gamma,
                     beta, fsum, fsum + c, eps, act, slope, (float)(1.0 / count), s, dy_ctot, dy_coff, keep, keep_scale);
  return cl4::launch_status();
}

// r05: cl4_bn_sync_bwd_local that also leaves max |dz| / max |xhat| per (channel, reduction workgroup) in `maxima` (2 * c * 32 floats:
// cl4_bn_bwd_maxima_floats), and cl4_bn_sync_bwd_apply that also writes dx's fp16 planes twin `dxs` (cl4_planes_bytes(n, c, hw, 2)) scaled by
// the power of two it leaves in slot[0] (see bn_bwd_bound_kernel). x and dx whole tensors, c % 8 == 0; no Dropout mask.
extern "C" size_t cl4_bn_bwd_maxima_floats(int c) { return (size_t)c * SPLIT * 2; }

extern "C" int cl4_bn_sync_bwd_local_max(const float* x, const float* dy, int n, int c, int hw, int dy_ctot, int dy_coff, const float* mean,
                                         const float* var, const float* gamma, const float* beta, float eps, int act, float slope, double* sums,
                                         float* dgamma, float* dbeta, const float* gamma_sign_src, float* maxima, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, c, 0};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && mean && var && sums && workspace && maxima, "bn_sync_bwd_local_max: null pointer");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_sync_bwd_local_max: dy slice out of range");
  size_t need = cl4_bn_workspace_bytes(c) + (size_t)2 * c * sizeof(float);
  if (workspace_bytes < need) { cl4::set_error("bn_sync_bwd_local_max: workspace too small"); return CL4_ENOSPC; }
  double* partial = (double*)workspace;
  float* fsum = (float*)((char*)workspace + cl4_bn_workspace_bytes(c));
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, dy, partial, mean, var, gamma, beta, eps, act,
                     slope, s, dy_ctot, dy_coff, (const uint8_t*)nullptr, 1.f, maxima);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cl4::cdiv(c, 64)), dim3(64), 0, st, (const double*)partial, fsum, fsum + c, dgamma,
                     dbeta, c, gamma_sign_src, sums);
  return cl4::launch_status();
}

// The whole backward of a ONE-RANK train-mode ABN (no exchange between the reduction and the elementwise pass) in two launches, three with
// the planes twin: reduction (+ maxima) -> [per-channel constants, bound, dgamma / dbeta] -> dx (+ twin). The two-phase pair above is
// 5 launches (7 with the InPlaceABN sign rule as separate passes) of which 3-5 are a few microseconds of work each: 109 layers per phase-1
// iteration. workspace: cl4_bn_train_bwd_act_workspace_bytes(c).
extern "C" size_t cl4_bn_train_bwd_act_workspace_bytes(int c) {
  return cl4_bn_workspace_bytes(c) + (size_t)c * SPLIT * 2 * sizeof(float) + (size_t)9 * c * sizeof(float);
}

extern "C" int cl4_bn_train_bwd_act(const float* x, const float* dy, float* dx, void* dxs, float* slot, int n, int c, int hw, int dy_ctot,
                                    int dy_coff, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act,
                                    float slope, float* dgamma, float* dbeta, const float* gamma_sign_src, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  BnShape s{n, c, hw, c, 0};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dx && mean && var && workspace, "bn_train_bwd_act: null pointer");
  CL4_REQUIRE((dxs == nullptr) == (slot == nullptr), "bn_train_bwd_act: dxs and slot come together");
  CL4_REQUIRE(!dxs || c % 8 == 0, "bn_train_bwd_act: c %% 8 == 0 for the planes twin");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_train_bwd_act: dy slice out of range");
  if (workspace_bytes < cl4_bn_train_bwd_act_workspace_bytes(c)) { cl4::set_error("bn_train_bwd_act: workspace too small"); return CL4_ENOSPC; }
  double* partial = (double*)workspace;
  float* maxima = (float*)((char*)workspace + cl4_bn_workspace_bytes(c));
  float* coef = maxima + (size_t)c * SPLIT * 2;
  hipStream_t st = cl4::as_stream(stream);
  const float invn = (float)(1.0 / (double)((long)n * hw));       // (as cl4_bn_sync_bwd_apply computes it from the global count)
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(c, reduction_split(n * hw, c)), dim3(256), 0, st, x, dy, partial, mean, var, gamma, beta, eps, act, slope, s, dy_ctot,
                     dy_coff, (const uint8_t*)nullptr, 1.f, dxs ? maxima : (float*)nullptr);
  if (!dxs) {
    hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, dim3(n * c, cl4::cdiv(hw, 1024) < 64 ? cl4::cdiv(hw, 1024) : 64), dim3(256), 0, st, x, dy, dx, mean,
                       var, gamma, beta, (const double*)partial, dgamma, dbeta, eps, act, slope, invn, s, dy_ctot, dy_coff, (const uint8_t*)nullptr, 1.f,
                       gamma_sign_src);
    return cl4::launch_status();
  }
  hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(cl4::cdiv(c, 4)), dim3(256), 0, st, (const double*)nullptr, (const float*)maxima, mean, var, gamma, beta, eps,
                     invn, c, coef, coef + 8 * (long)c, (const double*)partial, dgamma, dbeta, gamma_sign_src);
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, st, x, dy, dx, (u32x4n*)dxs, (const float*)coef, act, slope,
                     n, c, hw, dy_ctot, dy_coff, (const float*)(coef + 8 * (long)c), slot);
  return cl4::launch_status();
}

extern "C" int cl4_bn_sync_bwd_apply_planes(const float* x, const float* dy, float* dx, void* dxs, float* slot, int n, int c, int hw, int dy_ctot,
                                            int dy_coff, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                                            int act, float slope, const double* sums, double count, const float* maxima, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  BnShape s{n, c, hw, c, 0};
  int rc = check_shape(s);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dx && dxs && slot && mean && var && sums && maxima && workspace && count > 0.0, "bn_sync_bwd_apply_planes: bad arguments");
  CL4_REQUIRE(c % 8 == 0, "bn_sync_bwd_apply_planes: c %% 8 == 0 (the planes format's channel groups)");
  CL4_REQUIRE(dy_coff >= 0 && dy_coff + c <= dy_ctot, "bn_sync_bwd_apply_planes: dy slice out of range");
  if (workspace_bytes < (size_t)9 * c * sizeof(float)) { cl4::set_error("bn_sync_bwd_apply_planes: workspace too small"); return CL4_ENOSPC; }
  float* coef = (float*)workspace;            // (hipMalloc'ed by the caller: 16-byte aligned)
  hipStream_t st = cl4::as_stream(stream);
  const float invn = (float)(1.0 / count);
  hipLaunchKernelGGL(bn_bwd_prep_kernel, dim3(cl4::cdiv(c, 4)), dim3(256), 0, st, sums, maxima, mean, var, gamma, beta, eps, invn, c, coef, coef + 8 * (long)c);
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(bn_bwd_apply_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, st, x, dy, dx, (u32x4n*)dxs, (const float*)coef, act, slope,
                     n, c, hw, dy_ctot, dy_coff, (const float*)(coef + 8 * (long)c), slot);
  return cl4::launch_status();
}

extern "C" int cl4_relu_bwd(const float* y, const float* dy, float* dx, long n, void* stream) {
  CL4_REQUIRE(y && dy && dx && n > 0, "relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), y, dy, dx, n);
  return cl4::launch_status();
}

extern "C" int cl4_bias_grad(const float* dy, float* db, int n, int c, int hw, int ctot, int coff, void* stream) {
  CL4_REQUIRE(dy && db && n > 0 && c > 0 && hw > 0 && coff >= 0 && coff + c <= ctot, "bias_grad: bad arguments");
  const long per_c = (long)n * hw;
  int slices = (int)((per_c + 4095) / 4096);          // >= 16 elements per thread
  const int want = (1024 + c - 1) / c;                // ~4 workgroups per CU over all channels
  if (slices > want) slices = want;
  if (slices < 1) slices = 1;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(c, slices), dim3(256), 0, cl4::as_stream(stream), dy, db, n, c, hw, ctot, coff);
  return cl4::launch_status();
}

extern "C" int cl4_dropout_mask(uint8_t* keep, long n, uint64_t seed, float p, void* stream) {
  CL4_REQUIRE(keep && n > 0 && p >= 0.f && p < 1.f, "dropout_mask: bad arguments");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), keep, n, seed,
                     (const uint64_t*)nullptr, p);
  return cl4::launch_status();
}

extern "C" int cl4_random_labels(int* out, long n, uint64_t seed, int modulus, void* stream) {
  CL4_REQUIRE(out && n > 0 && modulus > 0, "random_labels: bad arguments");
  hipLaunchKernelGGL(random_labels_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), out, n, seed,
                     (const uint64_t*)nullptr, modulus);
  return cl4::launch_status();
}

__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { *c += inc; }

__global__ void abs_eps_kernel(const float* __restrict__ w, float* __restrict__ out, int n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fabsf(w[i]) + eps;
}
__global__ void sign_axpy_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ w, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += w[i] > 0.f ? x[i] : -x[i];
}

// InPlaceABN's effective scale and its chain rule (mapillary/inplace_abn, absent third-party package, believed behaviour:
// the fused kernels compute with gamma = |weight| + eps and return d(weight) = (weight > 0 ? 1 : -1) * d(gamma)).
extern "C" int cl4_abs_eps(const float* w, float* out, int n, float eps, void* stream) {
  CL4_REQUIRE(w && out && n > 0, "abs_eps: bad arguments");
  hipLaunchKernelGGL(abs_eps_kernel, dim3(cl4::cdiv(n, 256)), dim3(256), 0, cl4::as_stream(stream), w, out, n, eps);
  return cl4::launch_status();
}
// r05: cl4_abs_eps of every InPlaceABN layer of a network in one launch (109 launches of ~5 us per phase-1 iteration otherwise); the job
// table is laid out like cl4_pack_job's (include/cl4wsis.h: cl4_abs_job), one item per channel.
struct AbsJob {
  const float* w;
  float* out;
  long first;
  int n;
  float eps;
};
__global__ __launch_bounds__(256) void abs_eps_batched_kernel(const AbsJob* __restrict__ jobs, int njobs, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].first <= i) lo = mid; else hi = mid - 1;
    }
    const AbsJob jb = jobs[lo];
    const long r = i - jb.first;
    jb.out[r] = fabsf(jb.w[r]) + jb.eps;
  }
}
extern "C" int cl4_abs_eps_batched(const cl4_abs_job* jobs, int njobs, long total_items, void* stream) {
  CL4_REQUIRE(jobs && njobs > 0 && total_items > 0, "abs_eps_batched: bad arguments");
  static_assert(sizeof(AbsJob) == sizeof(cl4_abs_job) && sizeof(AbsJob) == 32, "cl4_abs_job layout (include/cl4wsis.h)");
  hipLaunchKernelGGL(abs_eps_batched_kernel, dim3(cl4::stream_grid(total_items, 256)), dim3(256), 0, cl4::as_stream(stream), (const AbsJob*)jobs, njobs,
                     total_items);
  return cl4::launch_status();
}
extern "C" int cl4_sign_axpy(float* y, const float* x, const float* w, int n, void* stream) {
  CL4_REQUIRE(y && x && w && n > 0, "sign_axpy: bad arguments");
  hipLaunchKernelGGL(sign_axpy_kernel, dim3(cl4::cdiv(n, 256)), dim3(256), 0, cl4::as_stream(stream), y, x, w, n);
  return cl4::launch_status();
}

// The same draws from a seed that lives in device memory and is advanced by the launch (like cl4_dropout_mask_dev): a captured
// phase-1 step draws fresh RandDrop labels at every replay (utils/wss_loss.py:237 draws them with Tensor.random_ per call).
extern "C" int cl4_random_labels_dev(int* out, long n, uint64_t* seed_dev, uint64_t advance, int modulus, void* stream) {
  CL4_REQUIRE(out && seed_dev && n > 0 && modulus > 0, "random_labels_dev: bad arguments");
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(random_labels_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, st, out, n, (uint64_t)0, (const uint64_t*)seed_dev, modulus);
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, st, seed_dev, advance);
  return cl4::launch_status();
}

extern "C" int cl4_dropout_mask_dev(uint8_t* keep, long n, uint64_t* seed_dev, uint64_t advance, float p, void* stream) {
  CL4_REQUIRE(keep && seed_dev && n > 0 && p >= 0.f && p < 1.f, "dropout_mask_dev: bad arguments");
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, st, keep, n, (uint64_t)0,
                     (const uint64_t*)seed_dev, p);
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, st, seed_dev, advance);
  return cl4::launch_status();
}
