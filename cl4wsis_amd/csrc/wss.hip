// Weakly-supervised loss stack of phase 1 / step 0. Everything here works on the
// stride-16 class maps ([b, 21, 33, 33] at crop 513) or streams a full-resolution logit tensor once, so the kernels are
// latency / HBM bound: one launch per stage instead of the reference's ~35 small kernels per PAMR call, no
// [b, c, 40, h, w] intermediates, reductions in fixed order (double accumulators) so results are run-to-run identical.
//   * PAMR (wss/modules.py:17-152): affinity from the image, then num_iter propagation steps with the class planes
//     ping-ponged in LDS;
//   * nGWP + focal CAM classification loss and its gradient (utils/wss_loss.py:7-53);
//   * pseudo_gtmask / binarize (wss/single_stage.py:18-40, utils/wss_loss.py:188-190), label mask fused;
//   * balanced_mask_loss_ce value + gradient (wss/single_stage.py:43-80);
//   * BCEWithLogitsLossWithIgnoreIndex(reduction='none')(...).mean() value + gradient (utils/loss.py:62-85).
#include "common.h"

namespace {

constexpr int MAX_DIL = 8;

struct Dil {
  int n;
  int d[MAX_DIL];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Block-wide sum of one double per thread (256 threads), result valid in thread 0.
__device__ __forceinline__ double block_sum_d(double v) {
  __shared__ double red[4];
  v = cl4::wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_f(float v) {
  __shared__ float red[4];
  v = cl4::wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// PAMR
// ---------------------------------------------------------------------------------------------------------------
// Neighbour order of LocalAffinity (wss/modules.py:31-40): the 3x3 ring without the centre, row-major.
__constant__ int RING_DY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
__constant__ int RING_DX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

// One thread per pixel: aff[b, t, y, x] = softmax_t( -mean_rgb |x_c - x_n(t)| / (1e-8 + 0.1 * std_c) ), t over
// 8 * ndil neighbours; std_c = unbiased std of the 9 * ndil replicate-padded samples (centre counted ndil times).
__global__ __launch_bounds__(256) void pamr_affinity_kernel(const float* __restrict__ img, float* __restrict__ aff, int b,
                                                            int h, int w, Dil dil) {
  const int hw = h * w;
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)b * hw) return;
  const int n = (int)(p / hw), q = (int)(p % hw), y = q / w, x = q % w;
  const int nt = 8 * dil.n;
  float logit[8 * MAX_DIL];
  for (int t = 0; t < nt; ++t) logit[t] = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float* pl = img + ((long)n * 3 + c) * hw;
    const float ctr = pl[q];
    float sum = 0.f;
    for (int k = 0; k < dil.n; ++k) {
      const int d = dil.d[k];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) sum += pl[clampi(y + dy * d, 0, h - 1) * w + clampi(x + dx * d, 0, w - 1)];
    }
    const int cnt = 9 * dil.n;
    const float mean = sum / (float)cnt;
    float ss = 0.f;
    for (int k = 0; k < dil.n; ++k) {
      const int d = dil.d[k];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const float v = pl[clampi(y + dy * d, 0, h - 1) * w + clampi(x + dx * d, 0, w - 1)] - mean;
          ss += v * v;
        }
    }
    const float inv = 1.f / (1e-8f + 0.1f * sqrtf(ss / (float)(cnt - 1)));
    for (int k = 0; k < dil.n; ++k) {
      const int d = dil.d[k];
      for (int r = 0; r < 8; ++r) {
        const float v = pl[clampi(y + RING_DY[r] * d, 0, h - 1) * w + clampi(x + RING_DX[r] * d, 0, w - 1)];
        logit[k * 8 + r] -= fabsf(ctr - v) * inv;
      }
    }
  }
  float mx = -INFINITY;
  for (int t = 0; t < nt; ++t) {
    logit[t] *= (1.f / 3.f);
    mx = fmaxf(mx, logit[t]);
  }
  float den = 0.f;
  for (int t = 0; t < nt; ++t) {
    logit[t] = expf(logit[t] - mx);
    den += logit[t];
  }
  const float rden = 1.f / den;
  for (int t = 0; t < nt; ++t) aff[((long)n * nt + t) * hw + q] = logit[t] * rden;
}

// One propagation step for G class planes sharing the same affinities: dst[g][q] = sum_t aff[t][q] * src[g][nbr_t(q)].
template <int G>
__device__ __forceinline__ void pamr_step(const float* __restrict__ aff, const float* src, float* dst, int h, int w, int hw,
                                          int plane_stride, const Dil& dil) {
  for (int q = threadIdx.x; q < hw; q += blockDim.x) {
    const int y = q / w, x = q % w;
    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.f;
    for (int k = 0; k < dil.n; ++k) {
      const int d = dil.d[k];
      const int ym = clampi(y - d, 0, h - 1) * w, yp = clampi(y + d, 0, h - 1) * w, y0 = y * w;
      const int xm = clampi(x - d, 0, w - 1), xp = clampi(x + d, 0, w - 1);
      const int nb[8] = {ym + xm, ym + x, ym + xp, y0 + xm, y0 + xp, yp + xm, yp + x, yp + xp};
      const float* a = aff + (long)k * 8 * hw + q;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float av = a[(long)r * hw];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] += av * src[g * plane_stride + nb[r]];
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) dst[g * plane_stride + q] = acc[g];
  }
}

// grid (ceil(c / G), b): G class planes of one image ping-pong in LDS for all iterations; a single launch.
template <int G>
__global__ __launch_bounds__(256) void pamr_propagate_lds_kernel(const float* __restrict__ aff, const float* __restrict__ mask,
                                                                 float* __restrict__ out, int c, int h, int w, int num_iter,
                                                                 Dil dil) {
  extern __shared__ float planes[];  // [2][G][hw]
  const int hw = h * w, n = blockIdx.y, c0 = blockIdx.x * G;
  const float* a = aff + (long)n * 8 * dil.n * hw;
  for (int g = 0; g < G; ++g) {
    const int ch = c0 + g;
    for (int q = threadIdx.x; q < hw; q += blockDim.x) planes[g * hw + q] = ch < c ? mask[((long)n * c + ch) * hw + q] : 0.f;
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < num_iter; ++it) {
    pamr_step<G>(a, planes + cur * G * hw, planes + (cur ^ 1) * G * hw, h, w, hw, hw, dil);
    cur ^= 1;
    __syncthreads();
  }
  for (int g = 0; g < G; ++g) {
    const int ch = c0 + g;
    if (ch >= c) break;
    for (int q = threadIdx.x; q < hw; q += blockDim.x) out[((long)n * c + ch) * hw + q] = planes[(cur * G + g) * hw + q];
  }
}

// Large planes: one launch per iteration, planes ping-pong between two global buffers. grid (chunks, c, b).
__global__ __launch_bounds__(256) void pamr_propagate_global_kernel(const float* __restrict__ aff, const float* __restrict__ src,
                                                                    float* __restrict__ dst, int c, int h, int w, Dil dil) {
  const int hw = h * w, n = blockIdx.z, ch = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= hw) return;
  const float* a = aff + (long)n * 8 * dil.n * hw + q;
  const float* s = src + ((long)n * c + ch) * hw;
  const int y = q / w, x = q % w;
  float acc = 0.f;
  for (int k = 0; k < dil.n; ++k) {
    const int d = dil.d[k];
    const int ym = clampi(y - d, 0, h - 1) * w, yp = clampi(y + d, 0, h - 1) * w, y0 = y * w;
    const int xm = clampi(x - d, 0, w - 1), xp = clampi(x + d, 0, w - 1);
    const int nb[8] = {ym + xm, ym + x, ym + xp, y0 + xm, y0 + xp, yp + xm, yp + x, yp + xp};
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += a[(long)(k * 8 + r) * hw] * s[nb[r]];
  }
  dst[((long)n * c + ch) * hw + q] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
// nGWP + focal BCE
// ---------------------------------------------------------------------------------------------------------------
struct NgwpArgs {
  const float* x;       // [b, c, hw]
  const float* labels;  // [b, nl]
  int b, c, hw, nl, reduction_sum;
  float lam;
  double* sums;         // [b, c, 2]: sum x*m, sum m
  float* coef;          // [b, c, 2]: gA, gB (gradient coefficients on the two sums)
  float* ngwp;          // [b, c] or null
  float* loss;          // [1]
  float* dx;            // [b, c, hw] or null
};

// grid (c, b): per-class spatial sums of the channel softmax (softmax recomputed per class: c is ~21, hw ~1089).
__global__ __launch_bounds__(256) void ngwp_sums_kernel(NgwpArgs a) {
  const int k = blockIdx.x, n = blockIdx.y;
  const float* xb = a.x + (long)n * a.c * a.hw;
  double sxm = 0.0, sm = 0.0;
  for (int q = threadIdx.x; q < a.hw; q += blockDim.x) {
    float mx = -INFINITY;
    for (int j = 0; j < a.c; ++j) mx = fmaxf(mx, xb[(long)j * a.hw + q]);
    float den = 0.f;
    for (int j = 0; j < a.c; ++j) den += expf(xb[(long)j * a.hw + q] - mx);
    const float xk = xb[(long)k * a.hw + q];
    const float m = expf(xk - mx) / den;
    sxm += (double)(xk * m);
    sm += (double)m;
  }
  sxm = block_sum_d(sxm);
  sm = block_sum_d(sm);
  if (threadIdx.x == 0) {
    a.sums[((long)n * a.c + k) * 2 + 0] = sxm;
    a.sums[((long)n * a.c + k) * 2 + 1] = sm;
  }
}

// One workgroup: nGWP logits, BCE-with-logits against the labels of the last nl channels, gradient coefficients.
__global__ __launch_bounds__(256) void ngwp_final_kernel(NgwpArgs a) {
  const int total = a.b * a.c;
  double part = 0.0;
  const double scale = a.reduction_sum ? 1.0 / a.b : 1.0 / ((double)a.b * a.nl);
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int n = i / a.c, k = i % a.c;
    const double sxm = a.sums[(long)i * 2], sm = a.sums[(long)i * 2 + 1];
    const double u = sm / a.hw, om = 1.0 - u;
    const double lg = log((double)a.lam + u);
    const double y = sxm / (1.0 + sm) + om * om * om * lg;
    if (a.ngwp) a.ngwp[i] = (float)y;
    double g = 0.0;
    const int kl = k - (a.c - a.nl);
    if (kl >= 0) {
      const double t = a.labels[(long)n * a.nl + kl];
      part += fmax(y, 0.0) - y * t + log1p(exp(-fabs(y)));
      g = (1.0 / (1.0 + exp(-y)) - t) * scale;
    }
    const double fprime = -3.0 * om * om * lg + om * om * om / ((double)a.lam + u);
    a.coef[(long)i * 2 + 0] = (float)(g / (1.0 + sm));
    a.coef[(long)i * 2 + 1] = (float)(g * (-sxm / ((1.0 + sm) * (1.0 + sm)) + fprime / a.hw));
  }
  part = block_sum_d(part);
  if (threadIdx.x == 0) a.loss[0] = (float)(part * scale);
}

// dx[j] = m_j * (gA_j * (1 + x_j) + gB_j - sum_k m_k * (gA_k * x_k + gB_k)); one thread per pixel.
__global__ __launch_bounds__(256) void ngwp_grad_kernel(NgwpArgs a) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)a.b * a.hw) return;
  const int n = (int)(p / a.hw), q = (int)(p % a.hw);
  const float* xb = a.x + (long)n * a.c * a.hw + q;
  const float* cf = a.coef + (long)n * a.c * 2;
  float mx = -INFINITY;
  for (int j = 0; j < a.c; ++j) mx = fmaxf(mx, xb[(long)j * a.hw]);
  float den = 0.f, s = 0.f;
  for (int j = 0; j < a.c; ++j) {
    const float xj = xb[(long)j * a.hw];
    const float e = expf(xj - mx);
    den += e;
    s += e * (cf[j * 2] * xj + cf[j * 2 + 1]);
  }
  const float rden = 1.f / den;
  s *= rden;
  float* db = a.dx + (long)n * a.c * a.hw + q;
  for (int j = 0; j < a.c; ++j) {
    const float xj = xb[(long)j * a.hw];
    const float m = expf(xj - mx) * rden;
    db[(long)j * a.hw] = m * (cf[j * 2] * (1.f + xj) + cf[j * 2 + 1] - s);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pseudo_gtmask / binarize
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float label_scale(const float* labels, int n, int c, int k) {
  return (labels && k > 0) ? labels[(long)n * (c - 1) + (k - 1)] : 1.f;
}

// grid (c, b): thr[b, k] = max(max_hw(mask * label) * cutoff, cutoff_low)
__global__ __launch_bounds__(256) void pgt_threshold_kernel(const float* __restrict__ mask, const float* __restrict__ labels,
                                                            float* __restrict__ thr, int c, int hw, float cut_bkg, float cut_top,
                                                            float cut_low) {
  const int k = blockIdx.x, n = blockIdx.y;
  const float* pl = mask + ((long)n * c + k) * hw;
  const float ls = label_scale(labels, n, c, k);
  float mx = -INFINITY;
  for (int q = threadIdx.x; q < hw; q += blockDim.x) mx = fmaxf(mx, pl[q] * ls);
  mx = block_max_f(mx);
  if (threadIdx.x == 0) thr[(long)n * c + k] = fmaxf(mx * (k == 0 ? cut_bkg : cut_top), cut_low);
}

__global__ __launch_bounds__(256) void pgt_apply_kernel(const float* __restrict__ mask, const float* __restrict__ labels,
                                                        const float* __restrict__ thr, float* __restrict__ out, int b, int c, int hw) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)b * hw) return;
  const int n = (int)(p / hw), q = (int)(p % hw);
  const float* mb = mask + (long)n * c * hw + q;
  int cnt = 0;
  for (int k = 0; k < c; ++k) cnt += (mb[(long)k * hw] * label_scale(labels, n, c, k) > thr[(long)n * c + k]) ? 1 : 0;
  float* ob = out + (long)n * c * hw + q;
  for (int k = 0; k < c; ++k) {
    const bool on = mb[(long)k * hw] * label_scale(labels, n, c, k) > thr[(long)n * c + k];
    ob[(long)k * hw] = (on && cnt <= 1) ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(256) void binarize_kernel(const float* __restrict__ x, const float* __restrict__ labels,
                                                       float* __restrict__ out, int b, int c, int hw) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)b * hw) return;
  const int n = (int)(p / hw), q = (int)(p % hw);
  const float* xb = x + (long)n * c * hw + q;
  float mx = -INFINITY;
  for (int k = 0; k < c; ++k) mx = fmaxf(mx, xb[(long)k * hw] * label_scale(labels, n, c, k));
  float* ob = out + (long)n * c * hw + q;
  for (int k = 0; k < c; ++k) ob[(long)k * hw] = (xb[(long)k * hw] * label_scale(labels, n, c, k) >= mx) ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------
// balanced_mask_loss_ce
// ---------------------------------------------------------------------------------------------------------------
struct BalArgs {
  const float* mask;       // [b, c, hw] logits
  const float* pseudo_gt;  // [b, c, hw] 0/1
  const float* labels;     // [b, c-1]
  int b, c, hw, nchunk;
  float* counts;           // [b, c] pixels per class
  float* cw;               // [b, c] class weights, cw[b, c] ... followed by bw [b]
  float* bw;               // [b]
  double* partial;         // [b, nchunk]
  float* loss;             // [1]
  float* dmask;            // [b, c, hw] or null
};

__global__ __launch_bounds__(256) void bal_count_kernel(BalArgs a) {  // grid (c, b)
  const int k = blockIdx.x, n = blockIdx.y;
  const float* pl = a.pseudo_gt + ((long)n * a.c + k) * a.hw;
  double s = 0.0;
  for (int q = threadIdx.x; q < a.hw; q += blockDim.x) s += (double)pl[q];
  s = block_sum_d(s);
  if (threadIdx.x == 0) a.counts[(long)n * a.c + k] = (float)s;
}

__global__ void bal_weights_kernel(BalArgs a) {  // one thread per image
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.b) return;
  const float* cnt = a.counts + (long)n * a.c;
  float tot = 0.f;
  for (int k = 0; k < a.c; ++k) tot += cnt[k];
  float gt_num = 1.f, ps_num = 0.f;
  for (int k = 0; k < a.c; ++k) {
    a.cw[(long)n * a.c + k] = (tot - cnt[k]) / (1.f + tot);
    const float lab = k > 0 ? a.labels[(long)n * (a.c - 1) + k - 1] : 1.f;
    if (k > 0) gt_num += lab;
    ps_num += (cnt[k] * lab > 0.f) ? 1.f : 0.f;
  }
  a.bw[n] = (gt_num == ps_num) ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void bal_main_kernel(BalArgs a) {  // grid (nchunk, b)
  const int n = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (q < a.hw) {
    const float* gb = a.pseudo_gt + (long)n * a.c * a.hw + q;
    const float* xb = a.mask + (long)n * a.c * a.hw + q;
    // torch.argmax semantics: first index of the maximum; pixel ignored when the one-hot sum is < 1
    int cls = 0;
    float best = gb[0], tot = gb[0];
    for (int k = 1; k < a.c; ++k) {
      const float g = gb[(long)k * a.hw];
      tot += g;
      if (g > best) { best = g; cls = k; }
    }
    const bool valid = !(tot < 1.f);
    float pw = 0.f;
    for (int k = 0; k < a.c; ++k) pw += gb[(long)k * a.hw] * a.cw[(long)n * a.c + k];
    float mx = -INFINITY;
    for (int k = 0; k < a.c; ++k) mx = fmaxf(mx, xb[(long)k * a.hw]);
    float den = 0.f;
    for (int k = 0; k < a.c; ++k) den += expf(xb[(long)k * a.hw] - mx);
    const float lse = mx + logf(den);
    if (valid) v = (double)(pw * (lse - xb[(long)cls * a.hw]));
    if (a.dmask) {
      const float gscale = valid ? a.bw[n] * pw / ((float)a.hw * (float)a.b) : 0.f;
      float* db = a.dmask + (long)n * a.c * a.hw + q;
      for (int k = 0; k < a.c; ++k) {
        const float sm = expf(xb[(long)k * a.hw] - lse);
        db[(long)k * a.hw] = gscale * (sm - (k == cls ? 1.f : 0.f));
      }
    }
  }
  v = block_sum_d(v);
  if (threadIdx.x == 0) a.partial[(long)n * a.nchunk + blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void bal_final_kernel(BalArgs a) {  // one workgroup
  double s = 0.0;
  for (int n = threadIdx.x; n < a.b; n += blockDim.x) {
    double t = 0.0;
    for (int j = 0; j < a.nchunk; ++j) t += a.partial[(long)n * a.nchunk + j];
    s += (double)a.bw[n] * t / a.hw;
  }
  s = block_sum_d(s);
  if (threadIdx.x == 0) a.loss[0] = (float)(s / a.b);
}

// ---------------------------------------------------------------------------------------------------------------
// BCE-with-logits against one-hot labels with an ignore index (streams the logits once)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_ignore_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                         float* __restrict__ dx, double* __restrict__ partial, int b, int c,
                                                         long hw, int ignore_index) {
  const long N = (long)b * hw;
  const float gscale = 1.f / (float)N;
  double v = 0.0;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    const long n = p / hw, q = p % hw;
    const long long t = target[p];
    const bool live = t != ignore_index && t >= 0 && t < c;
    const float* xb = x + n * c * hw + q;
    float* db = dx ? dx + n * c * hw + q : nullptr;
    if (live) {
      float s = 0.f;
      for (int k = 0; k < c; ++k) {
        const float xv = xb[(long)k * hw];
        const float tk = k == t ? 1.f : 0.f;
        s += fmaxf(xv, 0.f) - xv * tk + log1pf(expf(-fabsf(xv)));
        if (db) db[(long)k * hw] = (1.f / (1.f + expf(-xv)) - tk) * gscale;
      }
      v += (double)s;
    } else if (db) {
      for (int k = 0; k < c; ++k) db[(long)k * hw] = 0.f;
    }
  }
  v = block_sum_d(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double scale,
                                                           float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// softmax over the channel axis of [b, c, hw] (train.py:372-373 `int_masks.softmax(dim=1)`).
__global__ __launch_bounds__(256) void softmax_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int b, int c, int hw) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)b * hw) return;
  const int n = (int)(p / hw), q = (int)(p % hw);
  const float* xb = x + (long)n * c * hw + q;
  float* yb = y + (long)n * c * hw + q;
  float mx = -INFINITY;
  for (int k = 0; k < c; ++k) mx = fmaxf(mx, xb[(long)k * hw]);
  float den = 0.f;
  for (int k = 0; k < c; ++k) den += expf(xb[(long)k * hw] - mx);
  for (int k = 0; k < c; ++k) yb[(long)k * hw] = expf(xb[(long)k * hw] - mx) / den;
}

bool make_dil(const int* dilations, int ndil, Dil* out) {
  if (!dilations || ndil < 1 || ndil > MAX_DIL) return false;
  out->n = ndil;
  for (int i = 0; i < ndil; ++i) {
    if (dilations[i] < 1) return false;
    out->d[i] = dilations[i];
  }
  return true;
}

constexpr int BCE_GRID = 2048;

}  // namespace

extern "C" size_t cl4_pamr_workspace_bytes(int b, int c, int h, int w, int ndil) {
  if (b <= 0 || c <= 0 || h <= 0 || w <= 0 || ndil <= 0) return 0;
  const size_t hw = (size_t)h * w;
  // affinities, plus one ping-pong buffer for planes too large for LDS
  return ((size_t)b * 8 * ndil * hw + (size_t)b * c * hw) * sizeof(float);
}

extern "C" int cl4_pamr(const float* img, const float* mask, float* out, int b, int c, int h, int w, int num_iter,
                        const int* dilations, int ndil, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(img && mask && out && workspace, "pamr: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0 && num_iter >= 0, "pamr: bad dims");
  CL4_REQUIRE(mask != out, "pamr: in-place not supported");
  Dil dil;
  CL4_REQUIRE(make_dil(dilations, ndil, &dil), "pamr: 1..%d dilations >= 1 expected", MAX_DIL);
  if (workspace_bytes < cl4_pamr_workspace_bytes(b, c, h, w, ndil)) {
    cl4::set_error("pamr: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  const int hw = h * w;
  float* aff = (float*)workspace;
  float* tmp = aff + (size_t)b * 8 * ndil * hw;
  hipLaunchKernelGGL(pamr_affinity_kernel, dim3(cl4::cdiv((long)b * hw, 256)), dim3(256), 0, st, img, aff, b, h, w, dil);
  if (num_iter == 0) {
    hipMemcpyAsync(out, mask, (size_t)b * c * hw * sizeof(float), hipMemcpyDeviceToDevice, st);
    return cl4::launch_status();
  }
  const size_t lds4 = (size_t)2 * 4 * hw * sizeof(float), lds1 = (size_t)2 * hw * sizeof(float);
  if (lds4 <= 64 * 1024 && c >= 4) {
    hipLaunchKernelGGL(pamr_propagate_lds_kernel<4>, dim3(cl4::cdiv(c, 4), b), dim3(256), lds4, st, aff, mask, out, c, h, w,
                       num_iter, dil);
  } else if (lds1 <= 64 * 1024) {
    hipLaunchKernelGGL(pamr_propagate_lds_kernel<1>, dim3(c, b), dim3(256), lds1, st, aff, mask, out, c, h, w, num_iter, dil);
  } else {
    // ping-pong so that the last iteration lands in `out`
    const float* src = mask;
    for (int it = 0; it < num_iter; ++it) {
      float* dst = ((num_iter - 1 - it) & 1) ? tmp : out;
      hipLaunchKernelGGL(pamr_propagate_global_kernel, dim3(cl4::cdiv(hw, 256), c, b), dim3(256), 0, st, aff, src, dst, c, h, w,
                         dil);
      src = dst;
    }
  }
  return cl4::launch_status();
}

extern "C" size_t cl4_ngwp_workspace_bytes(int b, int c) {
  if (b <= 0 || c <= 0) return 0;
  return (size_t)b * c * 2 * (sizeof(double) + sizeof(float));
}

extern "C" int cl4_ngwp_focal_bce(const float* x, const float* labels, int b, int c, int h, int w, int nl, int reduction_sum,
                                  float lam, float* ngwp, float* loss, float* dx, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  CL4_REQUIRE(x && labels && loss && workspace, "ngwp_focal_bce: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0 && nl > 0 && nl <= c, "ngwp_focal_bce: bad dims");
  if (workspace_bytes < cl4_ngwp_workspace_bytes(b, c)) {
    cl4::set_error("ngwp_focal_bce: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  NgwpArgs a;
  a.x = x; a.labels = labels; a.b = b; a.c = c; a.hw = h * w; a.nl = nl; a.reduction_sum = reduction_sum; a.lam = lam;
  a.sums = (double*)workspace;
  a.coef = (float*)(a.sums + (size_t)b * c * 2);
  a.ngwp = ngwp; a.loss = loss; a.dx = dx;
  hipLaunchKernelGGL(ngwp_sums_kernel, dim3(c, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(ngwp_final_kernel, dim3(1), dim3(256), 0, st, a);
  if (dx) hipLaunchKernelGGL(ngwp_grad_kernel, dim3(cl4::cdiv((long)b * a.hw, 256)), dim3(256), 0, st, a);
  return cl4::launch_status();
}

extern "C" int cl4_pseudo_gtmask(const float* mask, const float* labels, float* out, int b, int c, int h, int w, float cutoff_top,
                                 float cutoff_bkg, float cutoff_low, float* thresholds, void* stream) {
  CL4_REQUIRE(mask && out && thresholds, "pseudo_gtmask: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0, "pseudo_gtmask: bad dims");
  hipStream_t st = cl4::as_stream(stream);
  const int hw = h * w;
  hipLaunchKernelGGL(pgt_threshold_kernel, dim3(c, b), dim3(256), 0, st, mask, labels, thresholds, c, hw, cutoff_bkg, cutoff_top,
                     cutoff_low);
  hipLaunchKernelGGL(pgt_apply_kernel, dim3(cl4::cdiv((long)b * hw, 256)), dim3(256), 0, st, mask, labels, thresholds, out, b, c, hw);
  return cl4::launch_status();
}

extern "C" int cl4_binarize(const float* x, const float* labels, float* out, int b, int c, int h, int w, void* stream) {
  CL4_REQUIRE(x && out, "binarize: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0, "binarize: bad dims");
  const int hw = h * w;
  hipLaunchKernelGGL(binarize_kernel, dim3(cl4::cdiv((long)b * hw, 256)), dim3(256), 0, cl4::as_stream(stream), x, labels, out, b, c,
                     hw);
  return cl4::launch_status();
}

extern "C" size_t cl4_balanced_ce_workspace_bytes(int b, int c, int h, int w) {
  if (b <= 0 || c <= 0 || h <= 0 || w <= 0) return 0;
  const size_t nchunk = (size_t)cl4::cdiv((long)h * w, 256);
  return (size_t)b * nchunk * sizeof(double) + ((size_t)b * c * 2 + b) * sizeof(float);
}

extern "C" int cl4_balanced_mask_loss_ce(const float* mask, const float* pseudo_gt, const float* labels, int b, int c, int h, int w,
                                         float* loss, float* dmask, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(mask && pseudo_gt && labels && loss && workspace, "balanced_mask_loss_ce: null pointer");
  CL4_REQUIRE(b > 0 && c > 1 && h > 0 && w > 0, "balanced_mask_loss_ce: bad dims");
  if (workspace_bytes < cl4_balanced_ce_workspace_bytes(b, c, h, w)) {
    cl4::set_error("balanced_mask_loss_ce: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  BalArgs a;
  a.mask = mask; a.pseudo_gt = pseudo_gt; a.labels = labels; a.b = b; a.c = c; a.hw = h * w;
  a.nchunk = cl4::cdiv(a.hw, 256);
  a.partial = (double*)workspace;
  a.counts = (float*)(a.partial + (size_t)b * a.nchunk);
  a.cw = a.counts + (size_t)b * c;
  a.bw = a.cw + (size_t)b * c;
  a.loss = loss; a.dmask = dmask;
  hipLaunchKernelGGL(bal_count_kernel, dim3(c, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bal_weights_kernel, dim3(cl4::cdiv(b, 64)), dim3(64), 0, st, a);
  hipLaunchKernelGGL(bal_main_kernel, dim3(a.nchunk, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bal_final_kernel, dim3(1), dim3(256), 0, st, a);
  return cl4::launch_status();
}

extern "C" size_t cl4_bce_ignore_workspace_bytes(void) { return (size_t)BCE_GRID * sizeof(double); }

extern "C" int cl4_bce_ignore_mean(const float* x, const long long* target, int b, int c, int h, int w, int ignore_index, float* loss,
                                   float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && target && loss && workspace, "bce_ignore_mean: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0, "bce_ignore_mean: bad dims");
  if (workspace_bytes < cl4_bce_ignore_workspace_bytes()) {
    cl4::set_error("bce_ignore_mean: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  const long hw = (long)h * w, N = (long)b * hw;
  const int grid = cl4::stream_grid(N, 256);
  hipLaunchKernelGGL(bce_ignore_kernel, dim3(grid), dim3(256), 0, st, x, target, dx, (double*)workspace, b, c, hw, ignore_index);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, grid, 1.0 / (double)N, loss);
  return cl4::launch_status();
}

extern "C" int cl4_softmax_channels(const float* x, float* y, int b, int c, int h, int w, void* stream) {
  CL4_REQUIRE(x && y && b > 0 && c > 0 && h > 0 && w > 0, "softmax_channels: bad arguments");
  hipLaunchKernelGGL(softmax_channels_kernel, dim3(cl4::cdiv((long)b * h * w, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, b, c,
                     h * w);
  return cl4::launch_status();
}
