// Label-stage kernels, part 1: CAM normalisation + smoothing, peak extraction, the flip-TTA semantic
// pseudo ground truth, and connected-component labelling with per-component statistics.
// All HBM/LDS-bound integer/float work; results feed the bit-exact index outputs (argmax maps, peak
// coordinates), so every formula follows the reference's operation order.
#include "label_common.h"

namespace {

using cl4::Bilin;
using cl4::bilin_half_pixel;
using cl4::bilin_sample;

// ---------------------------------------------------------------------------------------------
// PeakGenerator (eval) + cam_normalize + 3x3 smoothing — wss/modules.py:354-368,390-434; wss/utils.py:28-32.
// One workgroup per image; everything lives in LDS (cnew*h*w floats twice).
// ---------------------------------------------------------------------------------------------
__device__ float block_max(float v, float* red) {
  v = cl4::wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void cam_smooth_kernel(const float* __restrict__ int_masks, int ctot, int h, int w,
                                                         const float* __restrict__ pg_w, const float* __restrict__ pg_b,
                                                         const float* __restrict__ label, int cnew, float alpha,
                                                         float* __restrict__ cam) {
  extern __shared__ float lds[];
  __shared__ float red[4];
  const int hw = h * w;
  float* xs = lds;              // [cnew][hw] PAM output
  float* ys = lds + cnew * hw;  // [cnew][hw] normalised cam
  const int b = blockIdx.x;
  const float* src = int_masks + ((long)b * ctot + (ctot - cnew)) * hw;
  for (int i = threadIdx.x; i < cnew * hw; i += blockDim.x) xs[i] = fmaxf(src[i], 0.f);
  __syncthreads();
  // PAM: zero everything below alpha * per-channel max
  for (int c = 0; c < cnew; ++c) {
    float m = 0.f;  // relu output >= 0; AdaptiveMaxPool2d(1)
    for (int i = threadIdx.x; i < hw; i += blockDim.x) m = fmaxf(m, xs[c * hw + i]);
    m = block_max(m, red);
    const float thr = m * alpha;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) {
      float v = xs[c * hw + i];
      xs[c * hw + i] = (v < thr) ? 0.f : v;
    }
  }
  __syncthreads();
  // extra_conv4 (1x1, cnew -> cnew, bias) then relu * label
  for (int i = threadIdx.x; i < cnew * hw; i += blockDim.x) {
    int c = i / hw, p = i - c * hw;
    float acc = 0.f;
    for (int k = 0; k < cnew; ++k) acc += pg_w[c * cnew + k] * xs[k * hw + p];
    acc += pg_b[c];
    ys[i] = fmaxf(acc, 0.f) * label[b * cnew + c];
  }
  __syncthreads();
  // cam /= (max_hw + 1e-5)
  for (int c = 0; c < cnew; ++c) {
    float m = -INFINITY;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) m = fmaxf(m, ys[c * hw + i]);
    m = block_max(m, red);
    const float den = m + 1e-5f;
    for (int i = threadIdx.x; i < hw; i += blockDim.x) ys[c * hw + i] = ys[c * hw + i] / den;
  }
  __syncthreads();
  // 3x3 average, zero padding, divisor 9
  for (int i = threadIdx.x; i < cnew * hw; i += blockDim.x) {
    int c = i / hw, p = i - c * hw;
    int y = p / w, x = p - y * w;
    float s = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int yy = y + dy, xx = x + dx;
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) s += ys[c * hw + yy * w + xx];
      }
    cam[((long)b * cnew + c) * hw + p] = s / 9.f;
  }
}

// The same computation for maps that do not fit LDS (coco-voc: 20 new classes at 56x56): four small launches through a
// global scratch [b, cnew, hw] + per-plane maxima; identical arithmetic and summation order.
__global__ __launch_bounds__(256) void cam_peak_kernel(const float* __restrict__ int_masks, int ctot, int hw, int cnew,
                                                       float* __restrict__ peak) {   // grid (cnew, b)
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* src = int_masks + ((long)b * ctot + (ctot - cnew) + c) * hw;
  float m = 0.f;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) m = fmaxf(m, src[i]);
  m = block_max(m, red);
  if (threadIdx.x == 0) peak[b * cnew + c] = m;
}

__global__ __launch_bounds__(256) void cam_conv_kernel(const float* __restrict__ int_masks, int ctot, int hw, const float* __restrict__ pg_w,
                                                       const float* __restrict__ pg_b, const float* __restrict__ label, int cnew,
                                                       float alpha, const float* __restrict__ peak, float* __restrict__ ys, int b) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= (long)b * hw) return;
  const int n = (int)(p / hw), q = (int)(p % hw);
  const float* src = int_masks + ((long)n * ctot + (ctot - cnew)) * hw + q;
  for (int c = 0; c < cnew; ++c) {
    float acc = 0.f;
    for (int k = 0; k < cnew; ++k) {
      const float v = fmaxf(src[(long)k * hw], 0.f);
      const float z = (v < peak[n * cnew + k] * alpha) ? 0.f : v;
      acc += pg_w[c * cnew + k] * z;
    }
    acc += pg_b[c];
    ys[((long)n * cnew + c) * hw + q] = fmaxf(acc, 0.f) * label[n * cnew + c];
  }
}

__global__ __launch_bounds__(256) void cam_den_kernel(const float* __restrict__ ys, int hw, int cnew, float* __restrict__ den) {  // grid (cnew, b)
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* src = ys + ((long)b * cnew + c) * hw;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) m = fmaxf(m, src[i]);
  m = block_max(m, red);
  if (threadIdx.x == 0) den[b * cnew + c] = m + 1e-5f;
}

__global__ __launch_bounds__(256) void cam_avg_kernel(const float* __restrict__ ys, const float* __restrict__ den, int h, int w, long planes,
                                                      float* __restrict__ cam) {
  const int hw = h * w;
  const long total = planes * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pl = i / hw;
    const int p = (int)(i - pl * hw), y = p / w, x = p - y * w;
    const float d = den ? den[pl] : 1.f;
    float s = 0.f;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int yy = y + dy, xx = x + dx;
        if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) s += ys[pl * hw + yy * w + xx] / d;
      }
    cam[i] = s / 9.f;
  }
}

// ---------------------------------------------------------------------------------------------
// peak_extract — wss/utils.py:3-25, fused with the bilinear upsample of train.py:431:
// the full-resolution CAM is never written to HBM. Tile = 32x32 outputs + halo R=(kernel-1)/2.
// ---------------------------------------------------------------------------------------------
constexpr int PT = 32;
constexpr int PMAXR = 7;                      // kernel 15
constexpr int PTS = PT + 2 * PMAXR;           // 46
constexpr int PK_MAX = 64;                    // K <= 64 (25 in the reference)

struct PeakCand {
  float v;     // 0 = empty slot
  int idx;
};

// key = (value bits << 32) | (0xFFFFFFFF - idx): max key = largest value, then smallest flat index (positive values only,
// so the float bit pattern orders like the value)
__device__ __forceinline__ unsigned long long peak_key(float v, int idx) {
  return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)idx);
}

// workgroup-wide max of a 64-bit key (256 threads); every thread returns the winner
__device__ __forceinline__ unsigned long long block_max_key(unsigned long long key, unsigned long long* best) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(key, o, 64);
    key = other > key ? other : key;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) best[threadIdx.x >> 6] = key;
  __syncthreads();
  key = best[0];
  for (int q = 1; q < 4; ++q) key = best[q] > key ? best[q] : key;
  return key;
}

// One workgroup per 32x32 output tile of one (b, c): upsample + 15x15 max-pool NMS in LDS, then the tile's OWN top-K
// (value descending, ties by ascending flat index) is written to its K slots. A tile can contribute at most K entries to the
// global top-K, so the second pass over tiles*K slots is exact whatever the number of plateau ties (the earlier global
// candidate list with a fixed capacity dropped real peaks at 513x513, where near-zero plateaus yield > 8192 candidates).
__global__ __launch_bounds__(256) void peak_nms_kernel(const float* __restrict__ cam, int h, int w, int H, int W, int R, int K,
                                                       const float* __restrict__ label, PeakCand* __restrict__ cand) {
  extern __shared__ float lds[];
  float* lo = lds;                 // [h*w]
  float* val = lds + h * w;        // [PTS][PTS]
  float* rmax = val + PTS * PTS;   // [PTS][PT]
  __shared__ unsigned long long keys[PT * PT];
  __shared__ unsigned long long best[4];
  const int tiles_x = (W + PT - 1) / PT;
  const int ty0 = (blockIdx.x / tiles_x) * PT, tx0 = (blockIdx.x % tiles_x) * PT;
  const int c = blockIdx.y, b = blockIdx.z, C = gridDim.y;
  PeakCand* out = cand + (((long)(b * C + c)) * gridDim.x + blockIdx.x) * K;
  if (label[b * C + c] == 0.f) {   // unlabelled class: its CAM is identically zero, no positive peak
    for (int k = threadIdx.x; k < K; k += blockDim.x) out[k] = PeakCand{0.f, 0};
    return;
  }
  const float* plane = cam + ((long)b * C + c) * h * w;
  for (int i = threadIdx.x; i < h * w; i += blockDim.x) lo[i] = plane[i];
  __syncthreads();
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int TS = PT + 2 * R;
  for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) {
    int ty = i / TS, tx = i - ty * TS;
    int Y = ty0 - R + ty, X = tx0 - R + tx;
    float v = -INFINITY;  // max_pool2d pads with -inf
    if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W) {
      Bilin by = bilin_half_pixel(Y, sy, h), bx = bilin_half_pixel(X, sx, w);
      v = bilin_sample(lo, w, by, bx);
    }
    val[ty * PTS + tx] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TS * PT; i += blockDim.x) {
    int ty = i / PT, x = i - ty * PT;
    float m = -INFINITY;
    for (int d = 0; d <= 2 * R; ++d) m = fmaxf(m, val[ty * PTS + x + d]);
    rmax[ty * PT + x] = m;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PT * PT; i += blockDim.x) {
    int y = i / PT, x = i - y * PT;
    int Y = ty0 + y, X = tx0 + x;
    unsigned long long key = 0ull;
    if (Y < H && X < W) {
      float m = -INFINITY;
      for (int d = 0; d <= 2 * R; ++d) m = fmaxf(m, rmax[(y + d) * PT + x]);
      float v = val[(y + R) * PTS + x + R];
      if (v == m && v > 0.f) key = peak_key(v, Y * W + X);
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    unsigned long long key = 0ull;
    for (int i = threadIdx.x; i < PT * PT; i += blockDim.x) key = keys[i] > key ? keys[i] : key;
    key = block_max_key(key, best);
    if (key == 0ull) {             // uniform: the tile has fewer than K candidates
      for (int q = k + threadIdx.x; q < K; q += blockDim.x) out[q] = PeakCand{0.f, 0};
      return;
    }
    const int idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
    const int ly = idx / W - ty0, lx = idx % W - tx0;
    if (threadIdx.x == 0) {
      out[k] = PeakCand{__uint_as_float((unsigned)(key >> 32)), idx};
      keys[ly * PT + lx] = 0ull;   // remove the winner
    }
    __syncthreads();
  }
}

// top-K of the n = tiles*K slots of one (b, c): value descending, ties by ascending flat index.
// LDSK > 0 (r03): the candidates' keys are loaded into LDS once (LDSK of them at most) and the K selection passes run there — the
// global-memory form below re-read the list twice per pass from 80 workgroups (b * c), a chain of 50 dependent sweeps: 0.25 ms per step.
template <int LDSK>
__global__ __launch_bounds__(256) void peak_topk_kernel(PeakCand* __restrict__ cand, int n, int W,
                                                        int K, float* __restrict__ conf, int* __restrict__ ys,
                                                        int* __restrict__ xs) {
  __shared__ unsigned long long best[4];
  const int bc = blockIdx.x;
  PeakCand* cl = cand + (long)bc * n;
  if constexpr (LDSK > 0) {
    __shared__ unsigned long long keys[LDSK];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const float v = cl[i].v;
      keys[i] = v > 0.f ? peak_key(v, cl[i].idx) : 0ull;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
      unsigned long long key = 0ull;
      for (int i = threadIdx.x; i < n; i += blockDim.x) key = keys[i] > key ? keys[i] : key;
      key = block_max_key(key, best);
      float v = 0.f;
      int idx = 0;
      if (key != 0ull) {
        v = __uint_as_float((unsigned)(key >> 32));
        idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
        const unsigned low = (unsigned)(key & 0xFFFFFFFFu);
        for (int i = threadIdx.x; i < n; i += blockDim.x)
          if (keys[i] != 0ull && (unsigned)(keys[i] & 0xFFFFFFFFu) == low) keys[i] = 0ull;     // remove the winner (every copy of that pixel)
      }
      if (threadIdx.x == 0) {
        conf[(long)bc * K + k] = v;
        ys[(long)bc * K + k] = idx / W;
        xs[(long)bc * K + k] = idx % W;
      }
      __syncthreads();
    }
    return;
  }
  for (int k = 0; k < K; ++k) {
    unsigned long long key = 0ull;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      float v = cl[i].v;
      if (v > 0.f) {
        unsigned long long kk = peak_key(v, cl[i].idx);
        key = kk > key ? kk : key;
      }
    }
    key = block_max_key(key, best);
    float v = 0.f;
    int idx = 0;
    if (key != 0ull) {
      v = __uint_as_float((unsigned)(key >> 32));
      idx = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
      for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (cl[i].idx == idx && cl[i].v > 0.f) cl[i].v = -1.f;  // remove the winner
    }
    if (threadIdx.x == 0) {
      conf[(long)bc * K + k] = v;
      ys[(long)bc * K + k] = idx / W;
      xs[(long)bc * K + k] = idx % W;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Semantic pseudo ground truth — train.py:263,438-449. One workgroup per 16 output rows of one image.
// ---------------------------------------------------------------------------------------------
constexpr int SEG_ROWS = 16;   // output rows per workgroup (halved until the low-resolution band fits LDS)
constexpr int SEG_MAXC = 96;   // 21 on VOC, 81 on coco-voc
constexpr int SEG_REGC = 24;   // class counts up to this keep the per-pixel logits in registers (VOC); beyond: recomputed per pass

// No per-thread arrays with run-time indices here. The r02 kernel kept `float v[SEG_MAXC], mask[SEG_MAXC]` per thread = 784 bytes of
// scratch (private segment) per lane, ~410 MB for a full chip of waves. Removing them is a register / scratch optimisation (0.66 -> 0.26 ms,
// and no kernel of the step path uses private-segment scratch any more: tools/probes/kernel_resources.py). It was NOT the cause of the r02
// single-stream graph fault — that was the runtime's hipMemsetAsync graph nodes (FINDINGS.md finding 23, cl4::fill_bytes_async): the fault
// came back with these arrays already gone and disappears with kernel fills alone. REGC > 0:
// fully unrolled loops over REGC classes with compile-time indices (registers); REGC == 0: the logits are re-sampled from LDS in each
// of the three passes (max, sum of exponentials, arg-max) — the same expressions in the same order, so the same bits.
template <int REGC>
__global__ __launch_bounds__(256) void seg_pseudo_gt_kernel(const float* __restrict__ seg2b, const float* __restrict__ l1h,
                                                            int bs, int C, int h, int w, int H, int W, int old_classes,
                                                            int no_mask, int band, int seg_rows, uint8_t* __restrict__ seg_gt,
                                                            float* __restrict__ old_w, float* __restrict__ prob_win) {
  extern __shared__ float lds[];  // [C][band][w] flip-TTA mean of the logits, then [C] class mask
  const int b = blockIdx.y;
  const int Y0 = blockIdx.x * seg_rows;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int ylo = bilin_half_pixel(Y0, sy, h).i0;
  const long plane = (long)h * w;
  for (int i = threadIdx.x; i < C * band * w; i += blockDim.x) {
    int x = i % w;
    int t = i / w;
    int r = t % band;
    int c = t / band;
    int y = min(ylo + r, h - 1);
    float a = seg2b[((long)b * C + c) * plane + y * w + x];
    float f = seg2b[((long)(bs + b) * C + c) * plane + y * w + (w - 1 - x)];
    lds[i] = (a + f) / 2.f;
  }
  float* mask = lds + C * band * w;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = 1.f;
    if (no_mask) { if (c >= 1) m = l1h[b * (C - 1) + c - 1]; }
    else if (c >= old_classes) m = l1h[b * (C - 1) + c - 1];
    mask[c] = m;
  }
  __syncthreads();
  const int rows = min(seg_rows, H - Y0);
  for (int i = threadIdx.x; i < rows * W; i += blockDim.x) {
    int yy = i / W, X = i - yy * W;
    int Y = Y0 + yy;
    Bilin by = bilin_half_pixel(Y, sy, h), bx = bilin_half_pixel(X, sx, w);
    by.i0 -= ylo; by.i1 -= ylo;  // band-relative rows
    float mx = -INFINITY;
    float sum = 0.f;
    int arg = 0;
    float best = -INFINITY;
    if constexpr (REGC > 0) {
      float v[REGC];
#pragma unroll
      for (int c = 0; c < REGC; ++c)
        if (c < C) {
          v[c] = bilin_sample(lds + c * band * w, w, by, bx);
          mx = fmaxf(mx, v[c]);
        }
#pragma unroll
      for (int c = 0; c < REGC; ++c)
        if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
#pragma unroll
      for (int c = 0; c < REGC; ++c)
        if (c < C) {
          float p = (v[c] / sum) * mask[c];
          if (p > best) { best = p; arg = c; }
        }
    } else {
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, bilin_sample(lds + c * band * w, w, by, bx));
      for (int c = 0; c < C; ++c) sum += expf(bilin_sample(lds + c * band * w, w, by, bx) - mx);
      for (int c = 0; c < C; ++c) {
        float p = (expf(bilin_sample(lds + c * band * w, w, by, bx) - mx) / sum) * mask[c];
        if (p > best) { best = p; arg = c; }
      }
    }
    long o = ((long)b * H + Y) * W + X;
    old_w[o] = (arg < old_classes && arg != 0) ? 1.f : 0.f;
    seg_gt[o] = (uint8_t)(arg < old_classes ? 0 : arg);
    prob_win[o] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// Connected-component labelling (union-find with atomicMin) on a uint8 class map: pixels are
// connected iff they are neighbours (4- or 8-way) and carry the same non-zero value.
// label = smallest flat index (b*H*W + y*W + x) of the component, -1 for background.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int* L, int a) {
  int p = L[a];
  while (p != a) { a = p; p = L[a]; }
  return a;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }  // a > b: link a under b
    int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}

__global__ void ccl_init_kernel(const uint8_t* __restrict__ cls, int* __restrict__ L, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    L[i] = cls[i] ? (int)i : -1;
}

__global__ void ccl_merge_kernel(const uint8_t* __restrict__ cls, int* __restrict__ L, int H, int W, long total, int conn8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    uint8_t v = cls[i];
    if (!v) continue;
    int x = i % W;
    int y = (i / W) % H;
    if (x > 0 && cls[i - 1] == v) uf_union(L, (int)i, (int)i - 1);
    if (y > 0) {
      if (cls[i - W] == v) uf_union(L, (int)i, (int)i - W);
      if (conn8) {
        if (x > 0 && cls[i - W - 1] == v) uf_union(L, (int)i, (int)i - W - 1);
        if (x < W - 1 && cls[i - W + 1] == v) uf_union(L, (int)i, (int)i - W + 1);
      }
    }
  }
}

__global__ void ccl_compress_kernel(int* __restrict__ L, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int l = L[i];
    if (l >= 0) L[i] = uf_find(L, (int)i);
  }
}

// area / sum_x / sum_y per component root. Each thread walks 8 consecutive pixels of a row and flushes one
// set of atomics per run of equal roots (cuts same-address atomic traffic ~8x on large components).
constexpr int RUN = 8;
__global__ void cc_stats_kernel(const int* __restrict__ L, int H, int W, long total_runs, int runs_per_row,
                                unsigned* __restrict__ area, unsigned* __restrict__ sumx, unsigned* __restrict__ sumy) {
  for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < total_runs; r += (long)gridDim.x * blockDim.x) {
    int xr = r % runs_per_row;
    long row = r / runs_per_row;  // b*H + y
    int y = row % H;
    int x0 = xr * RUN;
    const int* lp = L + row * W;
    int cur = -1;
    unsigned a = 0, sx = 0;
    for (int x = x0; x < min(x0 + RUN, W); ++x) {
      int l = lp[x];
      if (l != cur) {
        if (cur >= 0) { atomicAdd(&area[cur], a); atomicAdd(&sumx[cur], sx); atomicAdd(&sumy[cur], a * (unsigned)y); }
        cur = l; a = 0; sx = 0;
      }
      if (l >= 0) { ++a; sx += (unsigned)x; }
    }
    if (cur >= 0) { atomicAdd(&area[cur], a); atomicAdd(&sumx[cur], sx); atomicAdd(&sumy[cur], a * (unsigned)y); }
  }
}


// Validation (train.py:695-745): softmax over the (optionally flip-averaged) full-resolution logits, arg-max (first maximum,
// like torch.max), and the confusion-matrix update of metrics/stream_metrics.py:72-79 — hist[true * n + pred] += 1 for
// labels in [0, n). grid-stride over pixels; per-workgroup LDS histogram when n*n fits, else global atomics.
__global__ __launch_bounds__(256) void semseg_hist_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                          unsigned long long* __restrict__ hist, unsigned char* __restrict__ pred,
                                                          int nb, int c, int H, int W, int flip, int n_classes, int zero_below) {
  extern __shared__ unsigned int lh[];
  const int nn = n_classes * n_classes;
  const bool use_lds = nn <= 4096;
  if (use_lds) {
    for (int i = threadIdx.x; i < nn; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
  const long HW = (long)H * W;
  const int nimg = flip ? 1 : nb;
  const long N = (long)nimg * HW;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const long q = p - (long)n * HW;
    const int y = (int)(q / W), x = (int)(q - (long)y * W);
    const float* a = logits + (long)n * c * HW + q;
    const float* b = flip ? logits + (long)c * HW + (long)y * W + (W - 1 - x) : nullptr;
    float mx = -INFINITY;
    for (int k = 0; k < c; ++k) {
      const float v = flip ? (a[(long)k * HW] + b[(long)k * HW]) / 2.f : a[(long)k * HW];
      mx = fmaxf(mx, v);
    }
    float den = 0.f;
    for (int k = 0; k < c; ++k) {
      const float v = flip ? (a[(long)k * HW] + b[(long)k * HW]) / 2.f : a[(long)k * HW];
      den += expf(v - mx);
    }
    float best = -1.f;
    int arg = 0;
    for (int k = 0; k < c; ++k) {
      const float v = flip ? (a[(long)k * HW] + b[(long)k * HW]) / 2.f : a[(long)k * HW];
      const float pr = expf(v - mx) / den;
      if (pr > best) { best = pr; arg = k; }
    }
    if (pred) pred[p] = (unsigned char)arg;
    long long t = labels[p];
    if (t >= 0 && t < zero_below) t = 0;            // eval_pseudolabeler: labels[labels < old_classes] = 0
    if (t >= 0 && t < n_classes) {
      if (use_lds) atomicAdd(&lh[(int)t * n_classes + arg], 1u);
      else atomicAdd(&hist[(long)t * n_classes + arg], 1ull);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += blockDim.x)
      if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
  }
}


// get_ins_map's semantic part (dataset/utils.py:817-836): per-sample softmax, flip average of the PROBABILITIES,
// optional image-level label mask (--val_clean), arg-max (first maximum) and the winning probability.
__global__ __launch_bounds__(256) void val_seg_kernel(const float* __restrict__ logits, const float* __restrict__ cls_label,
                                                      unsigned char* __restrict__ seg_map, float* __restrict__ prob_win, int c, int H,
                                                      int W, int flip) {
  const long HW = (long)H * W;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < HW; q += (long)gridDim.x * blockDim.x) {
    const int y = (int)(q / W), x = (int)(q - (long)y * W);
    const float* a = logits + q;
    const float* b = logits + (long)c * HW + (long)y * W + (W - 1 - x);
    float ma = -INFINITY, mb = -INFINITY;
    for (int k = 0; k < c; ++k) { ma = fmaxf(ma, a[(long)k * HW]); if (flip) mb = fmaxf(mb, b[(long)k * HW]); }
    float da = 0.f, db = 0.f;
    for (int k = 0; k < c; ++k) { da += expf(a[(long)k * HW] - ma); if (flip) db += expf(b[(long)k * HW] - mb); }
    float best = -1.f;
    int arg = 0;
    for (int k = 0; k < c; ++k) {
      float pr = expf(a[(long)k * HW] - ma) / da;
      if (flip) pr = (pr + expf(b[(long)k * HW] - mb) / db) / 2.f;
      if (cls_label && k > 0) pr *= cls_label[k - 1];
      if (pr > best) { best = pr; arg = k; }
    }
    seg_map[q] = (unsigned char)arg;
    prob_win[q] = best;
  }
}

}  // namespace

extern "C" size_t cl4_cam_smooth_workspace_bytes(int b, int cnew, int h, int w) {
  if (b <= 0 || cnew <= 0 || h <= 0 || w <= 0) return 0;
  if ((size_t)2 * cnew * h * w * sizeof(float) <= 64 * 1024) return 0;          // everything stays in LDS
  return ((size_t)b * cnew * h * w + (size_t)2 * b * cnew) * sizeof(float);
}

extern "C" int cl4_cam_smooth(const float* int_masks, int b, int ctot, int h, int w, const float* pg_w, const float* pg_b,
                              const float* label_new, int cnew, float alpha, float* cam, void* workspace, size_t workspace_bytes,
                              void* stream) {
  CL4_REQUIRE(int_masks && pg_w && pg_b && label_new && cam, "cam_smooth: null pointer");
  CL4_REQUIRE(b > 0 && cnew > 0 && cnew <= ctot && h > 0 && w > 0, "cam_smooth: bad dims");
  hipStream_t st = cl4::as_stream(stream);
  size_t lds = (size_t)2 * cnew * h * w * sizeof(float);
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(cam_smooth_kernel, dim3(b), dim3(256), lds, st, int_masks, ctot, h, w, pg_w, pg_b, label_new, cnew, alpha, cam);
    return cl4::launch_status();
  }
  const size_t need = cl4_cam_smooth_workspace_bytes(b, cnew, h, w);
  if (!workspace || workspace_bytes < need) { cl4::set_error("cam_smooth: workspace too small (%zu bytes needed)", need); return CL4_ENOSPC; }
  const int hw = h * w;
  float* ys = (float*)workspace;
  float* peak = ys + (size_t)b * cnew * hw;
  float* den = peak + (size_t)b * cnew;
  hipLaunchKernelGGL(cam_peak_kernel, dim3(cnew, b), dim3(256), 0, st, int_masks, ctot, hw, cnew, peak);
  hipLaunchKernelGGL(cam_conv_kernel, dim3(cl4::cdiv((long)b * hw, 256)), dim3(256), 0, st, int_masks, ctot, hw, pg_w, pg_b, label_new, cnew,
                     alpha, (const float*)peak, ys, b);
  hipLaunchKernelGGL(cam_den_kernel, dim3(cnew, b), dim3(256), 0, st, (const float*)ys, hw, cnew, den);
  hipLaunchKernelGGL(cam_avg_kernel, dim3(cl4::stream_grid((long)b * cnew * hw, 256)), dim3(256), 0, st, (const float*)ys, (const float*)den,
                     h, w, (long)b * cnew, cam);
  return cl4::launch_status();
}

// PeakGenerator.forward in eval mode without the caller's `smoothing` (wss/modules.py:390-434): the normalised CAM of the
// new classes, cam [b, cnew, h, w] = relu(conv(PAM(x))) * label / (max_hw + 1e-5). Same kernels and summation order as the
// large-map path of cl4_cam_smooth; workspace = cl4_cam_smooth_workspace_bytes' large-map size (b*cnew*hw + 2*b*cnew floats).
extern "C" size_t cl4_cam_normalize_workspace_bytes(int b, int cnew, int h, int w) {
  if (b <= 0 || cnew <= 0 || h <= 0 || w <= 0) return 0;
  return ((size_t)b * cnew * h * w + (size_t)2 * b * cnew) * sizeof(float);
}

namespace {
__global__ __launch_bounds__(256) void cam_divide_kernel(const float* __restrict__ ys, const float* __restrict__ den, int hw, long total,
                                                         float* __restrict__ cam) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) cam[i] = ys[i] / den[i / hw];
}
}  // namespace

extern "C" int cl4_cam_normalize(const float* int_masks, int b, int ctot, int h, int w, const float* pg_w, const float* pg_b,
                                 const float* label_new, int cnew, float alpha, float* cam, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  CL4_REQUIRE(int_masks && pg_w && pg_b && label_new && cam && workspace, "cam_normalize: null pointer");
  CL4_REQUIRE(b > 0 && cnew > 0 && cnew <= ctot && h > 0 && w > 0, "cam_normalize: bad dims");
  if (workspace_bytes < cl4_cam_normalize_workspace_bytes(b, cnew, h, w)) { cl4::set_error("cam_normalize: workspace too small"); return CL4_ENOSPC; }
  hipStream_t st = cl4::as_stream(stream);
  const int hw = h * w;
  float* ys = (float*)workspace;
  float* peak = ys + (size_t)b * cnew * hw;
  float* den = peak + (size_t)b * cnew;
  hipLaunchKernelGGL(cam_peak_kernel, dim3(cnew, b), dim3(256), 0, st, int_masks, ctot, hw, cnew, peak);
  hipLaunchKernelGGL(cam_conv_kernel, dim3(cl4::cdiv((long)b * hw, 256)), dim3(256), 0, st, int_masks, ctot, hw, pg_w, pg_b, label_new, cnew,
                     alpha, (const float*)peak, ys, b);
  hipLaunchKernelGGL(cam_den_kernel, dim3(cnew, b), dim3(256), 0, st, (const float*)ys, hw, cnew, den);
  const long total = (long)b * cnew * hw;
  hipLaunchKernelGGL(cam_divide_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, st, (const float*)ys, (const float*)den, hw, total, cam);
  return cl4::launch_status();
}

// `smoothing` (wss/utils.py:28-32): 3x3 average, stride 1, zero padding, divisor always 9, over `planes` maps of h x w.
extern "C" int cl4_smoothing3x3(const float* x, long planes, int h, int w, float* out, void* stream) {
  CL4_REQUIRE(x && out && x != out && planes > 0 && h > 0 && w > 0, "smoothing3x3: bad arguments (not in place)");
  hipLaunchKernelGGL(cam_avg_kernel, dim3(cl4::stream_grid(planes * h * w, 256)), dim3(256), 0, cl4::as_stream(stream), x,
                     (const float*)nullptr, h, w, planes, out);
  return cl4::launch_status();
}

extern "C" size_t cl4_peak_workspace_bytes(int b, int c, int H, int W, int K) {
  return (size_t)b * c * cl4::cdiv(H, PT) * cl4::cdiv(W, PT) * K * sizeof(PeakCand);
}

extern "C" int cl4_peak_extract(const float* cam, const float* label_new, int b, int c, int h, int w, int H, int W, int kernel,
                                int K, float* conf, int* ys, int* xs, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(cam && label_new && conf && ys && xs && workspace, "peak_extract: null pointer");
  CL4_REQUIRE(b > 0 && c > 0 && h > 0 && w > 0 && H > 0 && W > 0 && K > 0 && K <= PK_MAX, "peak_extract: bad dims (K <= 64)");
  CL4_REQUIRE((long)H * W < 2147483647L, "peak_extract: flat index overflows int32");
  CL4_REQUIRE(kernel % 2 == 1 && kernel >= 1 && (kernel - 1) / 2 <= PMAXR, "peak_extract: kernel %d unsupported (odd, <= 15)", kernel);
  if (workspace_bytes < cl4_peak_workspace_bytes(b, c, H, W, K)) { cl4::set_error("peak_extract: workspace too small"); return CL4_ENOSPC; }
  size_t lds = ((size_t)h * w + PTS * PTS + PTS * PT) * sizeof(float);
  CL4_REQUIRE(lds <= 48 * 1024, "peak_extract: low-res map %dx%d does not fit LDS", h, w);
  hipStream_t st = cl4::as_stream(stream);
  PeakCand* cand = (PeakCand*)workspace;
  const int tiles = cl4::cdiv(H, PT) * cl4::cdiv(W, PT);
  dim3 g(tiles, c, b);
  hipLaunchKernelGGL(peak_nms_kernel, g, dim3(256), lds, st, cam, h, w, H, W, (kernel - 1) / 2, K, label_new, cand);
  if (tiles * K <= 7680) hipLaunchKernelGGL(peak_topk_kernel<7680>, dim3(b * c), dim3(256), 0, st, cand, tiles * K, W, K, conf, ys, xs);      // 60 KB of keys (513 x 513, K = 25: 7225)
  else hipLaunchKernelGGL(peak_topk_kernel<0>, dim3(b * c), dim3(256), 0, st, cand, tiles * K, W, K, conf, ys, xs);
  return cl4::launch_status();
}

extern "C" int cl4_seg_pseudo_gt(const float* seg2b, const float* l1h, int bs, int C, int h, int w, int H, int W,
                                 int old_classes, int no_mask, uint8_t* seg_gt, float* old_w, float* prob_win, void* stream) {
  CL4_REQUIRE(seg2b && l1h && seg_gt && old_w && prob_win, "seg_pseudo_gt: null pointer");
  CL4_REQUIRE(bs > 0 && C > 1 && C <= SEG_MAXC && h > 0 && w > 0 && H > 0 && W > 0, "seg_pseudo_gt: bad dims (C <= %d)", SEG_MAXC);
  CL4_REQUIRE(old_classes >= 1 && old_classes <= C, "seg_pseudo_gt: old_classes out of range");
  int rows = SEG_ROWS, band;
  size_t lds;
  for (;;) {
    band = (int)((float)rows * (float)h / (float)H) + 3;
    if (band > h) band = h;
    lds = ((size_t)C * band * w + C) * sizeof(float);
    if (lds <= 64 * 1024 || rows == 1) break;
    rows /= 2;
  }
  CL4_REQUIRE(lds <= 64 * 1024, "seg_pseudo_gt: low-res band does not fit LDS");
  dim3 g(cl4::cdiv(H, rows), bs);
  if (C <= SEG_REGC)
    hipLaunchKernelGGL(seg_pseudo_gt_kernel<SEG_REGC>, g, dim3(256), lds, cl4::as_stream(stream), seg2b, l1h, bs, C, h, w, H, W,
                       old_classes, no_mask, band, rows, seg_gt, old_w, prob_win);
  else
    hipLaunchKernelGGL(seg_pseudo_gt_kernel<0>, g, dim3(256), lds, cl4::as_stream(stream), seg2b, l1h, bs, C, h, w, H, W,
                       old_classes, no_mask, band, rows, seg_gt, old_w, prob_win);
  return cl4::launch_status();
}

extern "C" int cl4_ccl(const uint8_t* cls_map, int b, int H, int W, int connectivity, int* labels, void* stream) {
  CL4_REQUIRE(cls_map && labels && b > 0 && H > 0 && W > 0, "ccl: bad arguments");
  CL4_REQUIRE(connectivity == 4 || connectivity == 8, "ccl: connectivity must be 4 or 8");
  long total = (long)b * H * W;
  CL4_REQUIRE(total < 2147483647L, "ccl: index overflow");
  hipStream_t st = cl4::as_stream(stream);
  int g = cl4::stream_grid(total, 256);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(g), dim3(256), 0, st, cls_map, labels, total);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(g), dim3(256), 0, st, cls_map, labels, H, W, total, connectivity == 8);
  hipLaunchKernelGGL(ccl_compress_kernel, dim3(g), dim3(256), 0, st, labels, total);
  return cl4::launch_status();
}

// area/sumx/sumy: uint32 [b*H*W], indexed by component root; zeroed here.
extern "C" int cl4_cc_stats(const int* labels, int b, int H, int W, unsigned* area, unsigned* sumx, unsigned* sumy, void* stream) {
  CL4_REQUIRE(labels && area && sumx && sumy && b > 0 && H > 0 && W > 0, "cc_stats: bad arguments");
  hipStream_t st = cl4::as_stream(stream);
  size_t bytes = (size_t)b * H * W * sizeof(unsigned);
  cl4::fill_bytes_async(area, 0, bytes, st);
  cl4::fill_bytes_async(sumx, 0, bytes, st);
  cl4::fill_bytes_async(sumy, 0, bytes, st);
  int rpr = cl4::cdiv(W, RUN);
  long runs = (long)b * H * rpr;
  hipLaunchKernelGGL(cc_stats_kernel, dim3(cl4::stream_grid(runs, 256)), dim3(256), 0, st, labels, H, W, runs, rpr, area, sumx, sumy);
  return cl4::launch_status();
}

extern "C" int cl4_semseg_hist(const float* logits, const long long* labels, unsigned long long* hist, unsigned char* pred, int nb, int c,
                               int H, int W, int flip, int n_classes, int zero_below, void* stream) {
  CL4_REQUIRE(logits && labels && hist, "semseg_hist: null pointer");
  CL4_REQUIRE(nb > 0 && c > 0 && H > 0 && W > 0 && n_classes > 0 && c <= n_classes, "semseg_hist: bad dims");
  CL4_REQUIRE(!flip || nb == 2, "semseg_hist: flip averaging expects the batch [image, flipped image]");
  const long N = (long)(flip ? 1 : nb) * H * W;
  const int nn = n_classes * n_classes;
  hipLaunchKernelGGL(semseg_hist_kernel, dim3(cl4::stream_grid(N, 256)), dim3(256), nn <= 4096 ? (size_t)nn * 4 : 0, cl4::as_stream(stream),
                     logits, labels, hist, pred, nb, c, H, W, flip, n_classes, zero_below);
  return cl4::launch_status();
}

extern "C" int cl4_val_seg(const float* logits, const float* cls_label, unsigned char* seg_map, float* prob_win, int nb, int c, int H,
                           int W, int flip, void* stream) {
  CL4_REQUIRE(logits && seg_map && prob_win && c > 0 && c < 256 && H > 0 && W > 0, "val_seg: bad arguments");
  CL4_REQUIRE(nb == (flip ? 2 : 1), "val_seg: batch must be 1, or 2 = [image, flipped image] with flip");
  hipLaunchKernelGGL(val_seg_kernel, dim3(cl4::stream_grid((long)H * W, 256)), dim3(256), 0, cl4::as_stream(stream), logits, cls_label,
                     seg_map, prob_win, c, H, W, flip);
  return cl4::launch_status();
}
