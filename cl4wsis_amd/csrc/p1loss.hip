// Phase-1 training-path losses, value and gradient, on the stride-16 class maps:
//   * PeakGenerator in training mode + its classification loss (wss/modules.py:354-423, train.py:334-336,344-345):
//     PAM (ReLU, keep x >= alpha * max_hw x) -> 1x1 conv nc->nc -> nGWP logit -> BCE-with-logits;
//   * FLAC flip / rot90 consistency loss (train.py:314-333);
//   * localisation prior l_loc (train.py:346-354), feature distillation lde (train.py:362-363);
//   * segmentation loss against the blended pseudo map l_seg (train.py:374-409).
// All maps are [b, c, 33, 33]-sized, so every stage is one or two small launches with fixed-order double reductions.
#include "common.h"

namespace {

constexpr int NC_MAX = 24;  // new classes handled by the PeakGenerator kernels (5 on VOC 15-5, 10 on 10-10, 20 on coco-voc)

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
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

// ---------------------------------------------------------------------------------------------------------------
// PeakGenerator (training)
// ---------------------------------------------------------------------------------------------------------------
struct PgArgs {
  const float* x;        // [b, ctot, hw]; the generator reads the last nc channels
  const float* w;        // [nc, nc]
  const float* bias;     // [nc]
  const float* labels;   // [b, nc]
  int b, ctot, hw, nc;
  float alpha;
  float* peak;           // [b, nc]
  double* sums;          // [b, nc, 2]
  float* coef;           // [b, nc, 2]
  float* logit;          // [b, nc]
  float* loss;           // [1]
  float* dx;             // [b, ctot, hw] or null (old-class channels are zero-filled)
  const float* dlogit;   // [b, nc] upstream gradient of the logits, or null = the built-in mean BCE against `labels`
  float* dw;             // [nc, nc] accumulated
  float* db;             // [nc] accumulated
};

// z_k = PAM(x)_k, u_j = sum_k w[j][k] z_k + bias_j, m = softmax_j(u) for one pixel.
__device__ __forceinline__ void pg_pixel(const PgArgs& a, int n, int q, const float* wsm, float* z, float* u, float* m) {
  const float* xb = a.x + ((long)n * a.ctot + (a.ctot - a.nc)) * a.hw + q;
  for (int k = 0; k < a.nc; ++k) {
    const float r = fmaxf(xb[(long)k * a.hw], 0.f);
    z[k] = r < a.peak[n * a.nc + k] * a.alpha ? 0.f : r;
  }
  float mx = -INFINITY;
  for (int j = 0; j < a.nc; ++j) {
    float s = wsm[a.nc * a.nc + j];
    for (int k = 0; k < a.nc; ++k) s += wsm[j * a.nc + k] * z[k];
    u[j] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int j = 0; j < a.nc; ++j) { m[j] = expf(u[j] - mx); den += m[j]; }
  const float rd = 1.f / den;
  for (int j = 0; j < a.nc; ++j) m[j] *= rd;
}

__global__ __launch_bounds__(256) void pg_peak_kernel(PgArgs a) {   // grid (nc, b)
  const int k = blockIdx.x, n = blockIdx.y;
  const float* pl = a.x + ((long)n * a.ctot + (a.ctot - a.nc) + k) * a.hw;
  float mx = 0.f;                                                   // max of relu(x)
  for (int q = threadIdx.x; q < a.hw; q += blockDim.x) mx = fmaxf(mx, pl[q]);
  mx = block_max_f(mx);
  if (threadIdx.x == 0) a.peak[n * a.nc + k] = mx;
}

__global__ __launch_bounds__(256) void pg_sums_kernel(PgArgs a) {   // grid (b)
  __shared__ float wsm[NC_MAX * NC_MAX + NC_MAX];
  for (int i = threadIdx.x; i < a.nc * a.nc + a.nc; i += blockDim.x) wsm[i] = i < a.nc * a.nc ? a.w[i] : a.bias[i - a.nc * a.nc];
  __syncthreads();
  const int n = blockIdx.x;
  double sxm[NC_MAX], sm[NC_MAX];
  for (int j = 0; j < NC_MAX; ++j) sxm[j] = sm[j] = 0.0;
  for (int q = threadIdx.x; q < a.hw; q += blockDim.x) {
    float z[NC_MAX], u[NC_MAX], m[NC_MAX];
    pg_pixel(a, n, q, wsm, z, u, m);
    for (int j = 0; j < a.nc; ++j) { sxm[j] += (double)(u[j] * m[j]); sm[j] += (double)m[j]; }
  }
  for (int j = 0; j < a.nc; ++j) {
    const double s1 = block_sum_d(sxm[j]), s2 = block_sum_d(sm[j]);
    if (threadIdx.x == 0) { a.sums[((long)n * a.nc + j) * 2] = s1; a.sums[((long)n * a.nc + j) * 2 + 1] = s2; }
  }
}

__global__ __launch_bounds__(256) void pg_final_kernel(PgArgs a) {  // one workgroup
  const int total = a.b * a.nc;
  double part = 0.0;
  const double scale = 1.0 / total;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const double sxm = a.sums[(long)i * 2], sm = a.sums[(long)i * 2 + 1];
    const double y = sxm / (1.0 + sm);
    const double t = a.labels ? a.labels[i] : 0.0;
    a.logit[i] = (float)y;
    part += fmax(y, 0.0) - y * t + log1p(exp(-fabs(y)));
    const double g = a.dlogit ? (double)a.dlogit[i] : (1.0 / (1.0 + exp(-y)) - t) * scale;
    a.coef[(long)i * 2] = (float)(g / (1.0 + sm));
    a.coef[(long)i * 2 + 1] = (float)(-g * sxm / ((1.0 + sm) * (1.0 + sm)));
  }
  part = block_sum_d(part);
  if (threadIdx.x == 0 && a.loss) a.loss[0] = (float)(part * scale);
}

__global__ __launch_bounds__(256) void pg_grad_kernel(PgArgs a) {   // grid (chunks, b)
  __shared__ float wsm[NC_MAX * NC_MAX + NC_MAX];
  for (int i = threadIdx.x; i < a.nc * a.nc + a.nc; i += blockDim.x) wsm[i] = i < a.nc * a.nc ? a.w[i] : a.bias[i - a.nc * a.nc];
  __syncthreads();
  const int n = blockIdx.y;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  float z[NC_MAX], du[NC_MAX];
  for (int j = 0; j < NC_MAX; ++j) { z[j] = 0.f; du[j] = 0.f; }
  if (q < a.hw) {
    float u[NC_MAX], m[NC_MAX];
    pg_pixel(a, n, q, wsm, z, u, m);
    const float* cf = a.coef + (long)n * a.nc * 2;
    float s = 0.f;
    for (int j = 0; j < a.nc; ++j) s += m[j] * (cf[2 * j] * u[j] + cf[2 * j + 1]);
    for (int j = 0; j < a.nc; ++j) du[j] = m[j] * (cf[2 * j] * (1.f + u[j]) + cf[2 * j + 1] - s);
    if (a.dx) {
      float* db_ = a.dx + (long)n * a.ctot * a.hw + q;
      for (int k = 0; k < a.ctot - a.nc; ++k) db_[(long)k * a.hw] = 0.f;
      for (int k = 0; k < a.nc; ++k) {
        float dz = 0.f;
        for (int j = 0; j < a.nc; ++j) dz += wsm[j * a.nc + k] * du[j];
        db_[(long)(a.ctot - a.nc + k) * a.hw] = z[k] != 0.f ? dz : 0.f;   // PAM passes the gradient where it kept x (> 0)
      }
    }
  }
  for (int j = 0; j < a.nc; ++j) {
    for (int k = 0; k < a.nc; ++k) {
      const double s = block_sum_d((double)(du[j] * z[k]));
      if (threadIdx.x == 0 && s != 0.0) atomicAdd(&a.dw[j * a.nc + k], (float)s);
    }
    const double s = block_sum_d((double)du[j]);
    if (threadIdx.x == 0 && s != 0.0) atomicAdd(&a.db[j], (float)s);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// FLAC
// ---------------------------------------------------------------------------------------------------------------
// Position in the rot90(k)-rotated map of the value that sits at (r, c) of the original square map.
__device__ __forceinline__ int rot_pos(int r, int c, int h, int k) {
  if (k == 1) return (h - 1 - c) * h + r;
  if (k == 2) return (h - 1 - r) * h + (h - 1 - c);
  if (k == 3) return c * h + (h - 1 - r);
  return r * h + c;
}

__global__ __launch_bounds__(256) void flac_kernel(const float* __restrict__ A, float* __restrict__ dA, double* __restrict__ partial,
                                                   int bs, int c, int old, int h, int k) {
  const int hw = h * h;
  const long N = (long)bs * hw;
  const int nn = c - old;
  const float inv_n = 1.f / (float)nn;
  const float gscale = 2.f / (3.f * (float)N) * inv_n;
  double v = 0.0;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    const int n = (int)(p / hw), q = (int)(p % hw), r = q / h, cc = q % h;
    const int qf = r * h + (h - 1 - cc);
    const int qr = rot_pos(r, cc, h, k);
    const float* po = A + ((long)n * c + old) * hw + q;
    const float* pf = A + ((long)(bs + n) * c + old) * hw + qf;
    const float* pr = A + ((long)(2 * bs + n) * c + old) * hw + qr;
    float so = 0.f, sf = 0.f, sr = 0.f;
    for (int j = 0; j < nn; ++j) { so += po[(long)j * hw]; sf += pf[(long)j * hw]; sr += pr[(long)j * hw]; }
    const float ao = sigmoidf(so * inv_n), af = sigmoidf(sf * inv_n), ar = sigmoidf(sr * inv_n);
    const float t = fmaxf(fmaxf(ao, af), ar);
    v += (double)((ao - t) * (ao - t)) + (double)((af - t) * (af - t)) + (double)((ar - t) * (ar - t));
    if (dA) {
      const float go = gscale * (ao - t) * ao * (1.f - ao), gf = gscale * (af - t) * af * (1.f - af), gr = gscale * (ar - t) * ar * (1.f - ar);
      float* d0 = dA + (long)n * c * hw + q;
      float* d1 = dA + (long)(bs + n) * c * hw + qf;
      float* d2 = dA + (long)(2 * bs + n) * c * hw + qr;
      for (int j = 0; j < c; ++j) {
        const bool nw = j >= old;
        d0[(long)j * hw] = nw ? go : 0.f;
        d1[(long)j * hw] = nw ? gf : 0.f;
        d2[(long)j * hw] = nw ? gr : 0.f;
      }
    }
  }
  v = block_sum_d(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double scale, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = block_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// ---------------------------------------------------------------------------------------------------------------
// l_loc, lde
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lloc_kernel(const float* __restrict__ x, const float* __restrict__ told, const float* __restrict__ labels,
                                                   float* __restrict__ dx, double* __restrict__ partial, int b, int ctot, int cold,
                                                   int hw) {
  const long N = (long)b * cold * hw;
  const float gs = 1.f / (float)N;
  double v = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % hw);
    const long t2 = i / hw;
    const int k = (int)(t2 % cold), n = (int)(t2 / cold);
    float t = sigmoidf(told[i]);
    if (labels && k > 0) t *= labels[(long)n * (ctot - 1) + k - 1];
    const long xo = ((long)n * ctot + k) * hw + q;
    const float xv = x[xo];
    v += (double)bce_logits(xv, t);
    if (dx) dx[xo] = (sigmoidf(xv) - t) * gs;
  }
  v = block_sum_d(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ dx,
                                                  double* __restrict__ partial, long n, float gscale) {
  double v = 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = x[i] - y[i];
    v += (double)(d * d);
    if (dx) dx[i] = gscale * d;
  }
  v = block_sum_d(v);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// l_seg
// ---------------------------------------------------------------------------------------------------------------
struct SegArgs {
  const float* seg;      // [b, tot, hw] logits of the new model
  const float* old_seg;  // [b, cold, hw] logits of the old model
  const float* masks;    // [b, tot, hw] PseudoLabeler logits (detached)
  const float* labels;   // [b, tot-1]
  int b, tot, cold, hw, nchunk;
  float alpha, icarl_bkg, weight;
  double* partial;       // [b, nchunk, 1 + cnew]: per-chunk BCE sum and new-class pixel mass
  float* bw;             // [b + 1]: batch weights, then weight / (sum bw + 1e-5)
  float* loss;
  float* dseg;           // [b, tot, hw] or null
};

// lx = alpha * binarize(p * lab) + (1 - alpha) * p * lab with p = softmax(masks): the blended pseudo map of one pixel.
__device__ __forceinline__ void seg_target(const SegArgs& a, int n, int q, float* tgt) {
  const float* mb = a.masks + (long)n * a.tot * a.hw + q;
  float mx = -INFINITY;
  for (int k = 0; k < a.tot; ++k) mx = fmaxf(mx, mb[(long)k * a.hw]);
  float den = 0.f;
  for (int k = 0; k < a.tot; ++k) den += expf(mb[(long)k * a.hw] - mx);
  float pmax = -INFINITY;
  for (int k = 0; k < a.tot; ++k) {
    float p = expf(mb[(long)k * a.hw] - mx) / den;
    if (k > 0) p *= a.labels[(long)n * (a.tot - 1) + k - 1];
    tgt[k] = p;
    pmax = fmaxf(pmax, p);
  }
  for (int k = 0; k < a.tot; ++k) tgt[k] = a.alpha * (tgt[k] >= pmax ? 1.f : 0.f) + (1.f - a.alpha) * tgt[k];
  const float* ob = a.old_seg + (long)n * a.cold * a.hw + q;
  const float lx0 = tgt[0];
  for (int k = 0; k < a.cold; ++k) tgt[k] = sigmoidf(ob[(long)k * a.hw]);
  tgt[0] = a.icarl_bkg == -1.f ? fminf(tgt[0], lx0) : (1.f - a.icarl_bkg) * tgt[0] + a.icarl_bkg * lx0;
}

constexpr int TOT_MAX = 96;

__global__ __launch_bounds__(256) void seg_sums_kernel(SegArgs a) {   // grid (nchunk, b)
  const int n = blockIdx.y, q = blockIdx.x * blockDim.x + threadIdx.x;
  const int cnew = a.tot - a.cold;
  float tgt[TOT_MAX];
  double l = 0.0;
  const bool live = q < a.hw;
  if (live) {
    seg_target(a, n, q, tgt);
    const float* sb = a.seg + (long)n * a.tot * a.hw + q;
    float s = 0.f;
    for (int k = 0; k < a.tot; ++k) s += bce_logits(sb[(long)k * a.hw], tgt[k]);
    l = (double)s;
  }
  double* out = a.partial + ((long)n * a.nchunk + blockIdx.x) * (1 + cnew);
  const double ls = block_sum_d(l);
  if (threadIdx.x == 0) out[0] = ls;
  for (int k = 0; k < cnew; ++k) {
    const double ps = block_sum_d(live ? (double)tgt[a.cold + k] : 0.0);
    if (threadIdx.x == 0) out[1 + k] = ps;
  }
}

__global__ __launch_bounds__(256) void seg_final_kernel(SegArgs a) {  // one workgroup
  const int cnew = a.tot - a.cold;
  __shared__ double lsum[256];
  __shared__ float bws[256];
  double num = 0.0, den = 0.0;
  for (int n0 = 0; n0 < a.b; n0 += 256) {
    const int n = n0 + threadIdx.x;
    if (n < a.b) {
      double l = 0.0;
      bool match = true;
      for (int k = 0; k < cnew; ++k) {
        double px = 0.0;
        for (int j = 0; j < a.nchunk; ++j) px += a.partial[((long)n * a.nchunk + j) * (1 + cnew) + 1 + k];
        match = match && ((px > 0.0) == (a.labels[(long)n * (a.tot - 1) + a.cold - 1 + k] != 0.f));
      }
      for (int j = 0; j < a.nchunk; ++j) l += a.partial[((long)n * a.nchunk + j) * (1 + cnew)];
      const float w = match ? 1.f : 0.f;
      a.bw[n] = w;
      lsum[threadIdx.x] = (double)w * l / a.hw;
      bws[threadIdx.x] = w;
    } else {
      lsum[threadIdx.x] = 0.0;
      bws[threadIdx.x] = 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int i = 0; i < 256; ++i) { num += lsum[i]; den += (double)bws[i]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double sc = (double)a.weight / (den + 1e-5);
    a.loss[0] = (float)(num * sc);
    a.bw[a.b] = (float)sc;
  }
}

__global__ __launch_bounds__(256) void seg_grad_kernel(SegArgs a) {   // grid (nchunk, b)
  const int n = blockIdx.y, q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.hw) return;
  float tgt[TOT_MAX];
  seg_target(a, n, q, tgt);
  const float gs = a.bw[n] * a.bw[a.b] / (float)a.hw;
  const float* sb = a.seg + (long)n * a.tot * a.hw + q;
  float* db = a.dseg + (long)n * a.tot * a.hw + q;
  for (int k = 0; k < a.tot; ++k) db[(long)k * a.hw] = gs * (sigmoidf(sb[(long)k * a.hw]) - tgt[k]);
}

// RandropLoss.forward — utils/wss_loss.py:227-248: at pixels where the largest new-class reference probability
// (sigmoid(masks) * labels) exceeds 0.5, a randomly drawn OLD class (drawn by the caller; the reference draws with the CPU
// generator) is pushed as a positive: -log sigmoid(x[neg]) averaged over those pixels. partial[2*blk] = loss sum,
// partial[2*blk+1] = pixel count; the gradient is written unscaled and normalised by randrop_scale_kernel.
__global__ __launch_bounds__(256) void randrop_kernel(const float* __restrict__ x, const float* __restrict__ masks,
                                                      const float* __restrict__ labels, const int* __restrict__ neg,
                                                      const float* __restrict__ mask_labels, float* __restrict__ dx,
                                                      double* __restrict__ partial, int b, int tot, int cold, int hw) {
  const long N = (long)b * hw;
  double v = 0.0, cnt = 0.0;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    const int n = (int)(p / hw), q = (int)(p % hw);
    const float* mb = masks + (long)n * tot * hw + q;
    float mx = -INFINITY;
    for (int k = cold; k < tot; ++k) mx = fmaxf(mx, sigmoidf(mb[(long)k * hw]) * labels[(long)n * (tot - 1) + k - 1]);
    int t = neg[p];
    bool live = mx > 0.5f && t >= 0 && t < tot;
    // --no_mask: a drawn class whose image-level label is 0 is ignored (class 0 = background is always kept)
    if (live && mask_labels && t > 0 && mask_labels[(long)n * (tot - 1) + t - 1] == 0.f) live = false;
    const float* xb = x + (long)n * tot * hw + q;
    if (dx) {
      float* db = dx + (long)n * tot * hw + q;
      for (int k = 0; k < tot; ++k) db[(long)k * hw] = (live && k == t) ? sigmoidf(xb[(long)k * hw]) - 1.f : 0.f;
    }
    if (live) { v += (double)bce_logits(xb[(long)t * hw], 1.f); cnt += 1.0; }
  }
  v = block_sum_d(v);
  cnt = block_sum_d(cnt);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = v; partial[2 * blockIdx.x + 1] = cnt; }
}

__global__ __launch_bounds__(256) void randrop_final_kernel(const double* __restrict__ partial, int n, float* __restrict__ loss,
                                                            float* __restrict__ inv_count) {
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  s = block_sum_d(s);
  c = block_sum_d(c);
  if (threadIdx.x == 0) {
    loss[0] = c > 0.0 ? (float)(s / c) : 0.f;
    inv_count[0] = c > 0.0 ? (float)(1.0 / c) : 0.f;
  }
}

__global__ void scale_by_dev_kernel(float* __restrict__ x, long n, const float* __restrict__ s) {
  const float f = *s;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= f;
}

constexpr int RED_GRID = 1024;

}  // namespace

extern "C" size_t cl4_p1loss_workspace_bytes(int b, int tot, int h, int w) {
  if (b <= 0 || tot <= 0 || h <= 0 || w <= 0) return 0;
  const size_t nchunk = (size_t)cl4::cdiv((long)h * w, 256);
  size_t seg = (size_t)b * nchunk * (1 + tot) * sizeof(double) + ((size_t)b + 1) * sizeof(float);
  size_t pg = (size_t)b * NC_MAX * (2 * sizeof(double) + 3 * sizeof(float));
  size_t red = (size_t)RED_GRID * 2 * sizeof(double) + 64;
  size_t m = seg > pg ? seg : pg;
  return (m > red ? m : red) + 256;
}

extern "C" int cl4_peakgen_train_loss(const float* x, const float* weight, const float* bias, const float* labels, int b, int ctot,
                                      int h, int w, int nc, float alpha, float* logit, float* loss, float* dx, float* dweight,
                                      float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && weight && bias && labels && logit && loss && workspace, "peakgen_train_loss: null pointer");
  CL4_REQUIRE(b > 0 && h > 0 && w > 0 && nc >= 1 && nc <= NC_MAX && nc <= ctot, "peakgen_train_loss: bad dims (1 <= nc <= %d)", NC_MAX);
  CL4_REQUIRE((dx == nullptr) == (dweight == nullptr) && (dx == nullptr) == (dbias == nullptr),
              "peakgen_train_loss: dx / dweight / dbias must come together");
  if (workspace_bytes < cl4_p1loss_workspace_bytes(b, ctot, h, w)) {
    cl4::set_error("peakgen_train_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  PgArgs a;
  a.x = x; a.w = weight; a.bias = bias; a.labels = labels; a.b = b; a.ctot = ctot; a.hw = h * w; a.nc = nc; a.alpha = alpha;
  a.sums = (double*)workspace;
  a.coef = (float*)(a.sums + (size_t)b * nc * 2);
  a.peak = a.coef + (size_t)b * nc * 2;
  a.logit = logit; a.loss = loss; a.dx = dx; a.dw = dweight; a.db = dbias; a.dlogit = nullptr;
  hipLaunchKernelGGL(pg_peak_kernel, dim3(nc, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(pg_sums_kernel, dim3(b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(pg_final_kernel, dim3(1), dim3(256), 0, st, a);
  if (dx) hipLaunchKernelGGL(pg_grad_kernel, dim3(cl4::cdiv(a.hw, 256), b), dim3(256), 0, st, a);
  return cl4::launch_status();
}

// PeakGenerator.forward in training mode as a differentiable map (wss/modules.py:390-423): logit [b, nc] = nGWP(conv(PAM(x))).
// dlogit == null: forward only (logit written). dlogit != null: vector-Jacobian product — dx [b, ctot, h, w] written,
// dweight [nc, nc] / dbias [nc] accumulated — for an arbitrary upstream gradient (the fused cl4_peakgen_train_loss hard-wires the
// mean BCE of train.py:344-345).
extern "C" int cl4_peakgen_vjp(const float* x, const float* weight, const float* bias, const float* dlogit, int b, int ctot, int h,
                               int w, int nc, float alpha, float* logit, float* dx, float* dweight, float* dbias, void* workspace,
                               size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && weight && bias && logit && workspace, "peakgen_vjp: null pointer");
  CL4_REQUIRE(b > 0 && h > 0 && w > 0 && nc >= 1 && nc <= NC_MAX && nc <= ctot, "peakgen_vjp: bad dims (1 <= nc <= %d)", NC_MAX);
  CL4_REQUIRE((dlogit == nullptr) == (dx == nullptr) && (dx == nullptr) == (dweight == nullptr) && (dx == nullptr) == (dbias == nullptr),
              "peakgen_vjp: dlogit / dx / dweight / dbias must come together");
  if (workspace_bytes < cl4_p1loss_workspace_bytes(b, ctot, h, w)) {
    cl4::set_error("peakgen_vjp: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  PgArgs a;
  a.x = x; a.w = weight; a.bias = bias; a.labels = nullptr; a.b = b; a.ctot = ctot; a.hw = h * w; a.nc = nc; a.alpha = alpha;
  a.sums = (double*)workspace;
  a.coef = (float*)(a.sums + (size_t)b * nc * 2);
  a.peak = a.coef + (size_t)b * nc * 2;
  a.logit = logit; a.loss = nullptr; a.dx = dx; a.dw = dweight; a.db = dbias; a.dlogit = dlogit;
  hipLaunchKernelGGL(pg_peak_kernel, dim3(nc, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(pg_sums_kernel, dim3(b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(pg_final_kernel, dim3(1), dim3(256), 0, st, a);
  if (dx) hipLaunchKernelGGL(pg_grad_kernel, dim3(cl4::cdiv(a.hw, 256), b), dim3(256), 0, st, a);
  return cl4::launch_status();
}

extern "C" int cl4_flac_loss(const float* A, int bs, int c, int old_classes, int h, int w, int angle, float* loss, float* dA,
                             void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(A && loss && workspace, "flac_loss: null pointer");
  CL4_REQUIRE(bs > 0 && c > old_classes && old_classes >= 0 && h > 0 && h == w, "flac_loss: square maps and new classes expected");
  CL4_REQUIRE(angle == 90 || angle == 180 || angle == 270, "flac_loss: angle must be 90, 180 or 270");
  if (workspace_bytes < (size_t)RED_GRID * sizeof(double)) {
    cl4::set_error("flac_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  const long N = (long)bs * h * h;
  int grid = cl4::stream_grid(N, 256);
  if (grid > RED_GRID) grid = RED_GRID;
  hipLaunchKernelGGL(flac_kernel, dim3(grid), dim3(256), 0, st, A, dA, (double*)workspace, bs, c, old_classes, h, angle / 90);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, grid, 1.0 / (3.0 * (double)N), loss);
  return cl4::launch_status();
}

extern "C" int cl4_lloc_loss(const float* x, const float* old_seg, const float* labels, int b, int ctot, int cold, int h, int w,
                             float* loss, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && old_seg && loss && workspace, "lloc_loss: null pointer");
  CL4_REQUIRE(b > 0 && cold > 0 && cold <= ctot && h > 0 && w > 0, "lloc_loss: bad dims");
  if (workspace_bytes < (size_t)RED_GRID * sizeof(double)) {
    cl4::set_error("lloc_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  const long N = (long)b * cold * h * w;
  int grid = cl4::stream_grid(N, 256);
  if (grid > RED_GRID) grid = RED_GRID;
  hipLaunchKernelGGL(lloc_kernel, dim3(grid), dim3(256), 0, st, x, old_seg, labels, dx, (double*)workspace, b, ctot, cold, h * w);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, grid, 1.0 / (double)N, loss);
  return cl4::launch_status();
}

extern "C" int cl4_mse_loss(const float* x, const float* y, long n, float grad_scale, float* loss, float* dx, void* workspace,
                            size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && y && loss && workspace && n > 0, "mse_loss: bad arguments");
  if (workspace_bytes < (size_t)RED_GRID * sizeof(double)) {
    cl4::set_error("mse_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  int grid = cl4::stream_grid(n, 256);
  if (grid > RED_GRID) grid = RED_GRID;
  hipLaunchKernelGGL(mse_kernel, dim3(grid), dim3(256), 0, st, x, y, dx, (double*)workspace, n, grad_scale * 2.f / (float)n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, grid, 1.0 / (double)n, loss);
  return cl4::launch_status();
}

extern "C" int cl4_p1_seg_loss(const float* seg, const float* old_seg, const float* masks, const float* labels, int b, int tot,
                               int cold, int h, int w, float alpha, float icarl_bkg, float weight, float* loss, float* dseg,
                               void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(seg && old_seg && masks && labels && loss && workspace, "p1_seg_loss: null pointer");
  CL4_REQUIRE(b > 0 && cold > 0 && cold < tot && tot <= TOT_MAX && h > 0 && w > 0, "p1_seg_loss: bad dims (tot <= %d)", TOT_MAX);
  if (workspace_bytes < cl4_p1loss_workspace_bytes(b, tot, h, w)) {
    cl4::set_error("p1_seg_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  SegArgs a;
  a.seg = seg; a.old_seg = old_seg; a.masks = masks; a.labels = labels; a.b = b; a.tot = tot; a.cold = cold; a.hw = h * w;
  a.nchunk = cl4::cdiv(a.hw, 256);
  a.alpha = alpha; a.icarl_bkg = icarl_bkg; a.weight = weight;
  a.partial = (double*)workspace;
  a.bw = (float*)(a.partial + (size_t)b * a.nchunk * (1 + (tot - cold)));
  a.loss = loss; a.dseg = dseg;
  hipLaunchKernelGGL(seg_sums_kernel, dim3(a.nchunk, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(seg_final_kernel, dim3(1), dim3(256), 0, st, a);
  if (dseg) hipLaunchKernelGGL(seg_grad_kernel, dim3(a.nchunk, b), dim3(256), 0, st, a);
  return cl4::launch_status();
}

extern "C" int cl4_randrop_loss(const float* x, const float* masks, const float* labels, const int* neg_labels,
                                const float* mask_labels, int b, int tot, int cold, int h, int w, float* loss, float* dx,
                                void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(x && masks && labels && neg_labels && loss && workspace, "randrop_loss: null pointer");
  CL4_REQUIRE(b > 0 && cold > 0 && cold < tot && h > 0 && w > 0, "randrop_loss: bad dims");
  if (workspace_bytes < (size_t)RED_GRID * 2 * sizeof(double) + 64) {
    cl4::set_error("randrop_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  const long N = (long)b * h * w;
  int grid = cl4::stream_grid(N, 256);
  if (grid > RED_GRID) grid = RED_GRID;
  double* partial = (double*)workspace;
  float* inv = (float*)(partial + (size_t)RED_GRID * 2);
  hipLaunchKernelGGL(randrop_kernel, dim3(grid), dim3(256), 0, st, x, masks, labels, neg_labels, mask_labels, dx, partial, b, tot, cold,
                     h * w);
  hipLaunchKernelGGL(randrop_final_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, grid, loss, inv);
  if (dx) hipLaunchKernelGGL(scale_by_dev_kernel, dim3(cl4::stream_grid((long)b * tot * h * w, 256)), dim3(256), 0, st, dx,
                             (long)b * tot * h * w, (const float*)inv);
  return cl4::launch_status();
}
