// HBM-bound spatial kernels of the backbone / heads: max-pool, box average pool, global mean,
// replicate pad, bilinear resize (forward + deterministic gather backward), W-flip, axpy.
// All tensors NCHW fp32; "planes" = n*c. One lane per output element with the fastest index
// on the lane (coalesced), grid-stride loops sized by cl4::stream_grid.
#include <cstdlib>
#include "common.h"

namespace {

// models/resnet.py:63 nn.MaxPool2d(3, stride=2, padding=1) (implicit -inf padding).
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int hin, int win,
                                    int hout, int wout) {
  // grid = (planes, chunks of the output plane): one 32-bit division per element
  const long pl = blockIdx.x;
  const int per = hout * wout;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < per; i += gridDim.y * blockDim.x) {
    const int oy = i / wout, ox = i - oy * wout;
    const float* xp = x + pl * hin * win;
    float m = -INFINITY;
    int iy0 = oy * 2 - 1, ix0 = ox * 2 - 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      int iy = iy0 + dy;
      if ((unsigned)iy >= (unsigned)hin) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        int ix = ix0 + dx;
        if ((unsigned)ix >= (unsigned)win) continue;
        m = fmaxf(m, xp[iy * win + ix]);
      }
    }
    y[pl * per + i] = m;
  }
}

// The same pooling with two wave-loads per output row instead of nine (FINDINGS.md finding 22: a CU retires one wave-load per ~16 cycles
// whatever its width, and thread-per-output issues 9 stride-2 dword loads). Workgroup = (plane, band of MPB output rows), thread =
// output column: per input row ONE unaligned 16-byte buffer load fetches columns 2ox-1 .. 2ox+2 (neighbouring lanes overlap by half;
// the lines are in L1), its horizontal max is kept, and the vertical window slides: output row oy = max(rows 2oy-1, 2oy, 2oy+1), the
// last of which is the first of row oy+1. Column 0 has no left neighbour: it loads from column 0 and shifts.
constexpr int MPB = 16, MPK = 4;
__global__ __launch_bounds__(1024) void maxpool3x3s2_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long planes,
                                                                 int hin, int win, int hout, int wout) {
  typedef float f32x4p __attribute__((ext_vector_type(4)));
  const long pl = blockIdx.x;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (unsigned)(planes * hin * win * 4), 0x00020000);
  // a workgroup takes MPK bands at once, (band, column) flattened over its lanes: map widths are 2^k + 1, and one band per
  // workgroup would spend a whole wave on the last column
  for (int idx = threadIdx.x; idx < MPK * wout; idx += blockDim.x) {
    const int bk = idx / wout, ox = idx - bk * wout;
    const int oy_a = (blockIdx.y * MPK + bk) * MPB, oy_b = min(hout, oy_a + MPB);
    if (oy_a >= hout) continue;
    const int c0 = 2 * ox - 1;                   // first column of the window
    const int cl = max(c0, 0);                   // first column loaded
    const bool v0 = c0 >= 0, v1 = c0 + 1 < win, v2 = c0 + 2 < win;
    auto rowmax = [&](int iy) {
      if ((unsigned)iy >= (unsigned)hin) return -INFINITY;          // uniform
      const f32x4p v = __builtin_bit_cast(f32x4p, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)((pl * hin + iy) * (long)win + cl) * 4u, 0, 0));
      const float e0 = v0 ? v[0] : -INFINITY;                      // column c0 (absent for ox = 0)
      const float e1 = v0 ? v[1] : v[0], e2 = v0 ? v[2] : v[1];    // columns c0+1, c0+2
      return fmaxf(e0, fmaxf(v1 ? e1 : -INFINITY, v2 ? e2 : -INFINITY));
    };
    float top = rowmax(2 * oy_a - 1);
    for (int oy = oy_a; oy < oy_b; ++oy) {
      const float mid = rowmax(2 * oy), bot = rowmax(2 * oy + 1);
      y[(pl * hout + oy) * (long)wout + ox] = fmaxf(top, fmaxf(mid, bot));
      top = bot;
    }
  }
}

// modules/deeplab.py:87 avg_pool2d(x, (kh,kw), stride=1): one workgroup per plane, the plane is staged
// in LDS once and every output window is summed from LDS (planes here are <= 65x65).
__global__ void avgpool_s1_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int kh, int kw) {
  extern __shared__ float tile[];
  const long pl = blockIdx.x;
  const float* xp = x + pl * h * w;
  for (int i = threadIdx.x; i < h * w; i += blockDim.x) tile[i] = xp[i];
  __syncthreads();
  const int ho = h - kh + 1, wo = w - kw + 1;
  const float inv = 1.f / (float)(kh * kw);
  // one wave per output window: lanes stride over the window, wave-sum at the end
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int o = wave; o < ho * wo; o += nw) {
    int oy = o / wo, ox = o - oy * wo;
    float s = 0.f;
    for (int t = lane; t < kh * kw; t += 64) {
      int dy = t / kw, dx = t - dy * kw;
      s += tile[(oy + dy) * w + ox + dx];
    }
    s = cl4::wave_sum(s);
    if (lane == 0) y[pl * ho * wo + o] = s * inv;
  }
}

// The same pooling when the window is almost the whole plane (DeepLabV3's eval image pooling: a 32 x 32 window on a 33 x 33 map,
// four outputs): one WAVE per plane, no LDS, no barrier — every lane walks the plane with coalesced loads and keeps one partial sum per
// output window. The one-workgroup-per-plane kernel above did load -> barrier -> sum for 4 KB at a time: 1.5 TB/s on the 65,536 planes
// of a 32-image res5 map (r03: 0.29 -> 0.12 ms per step for the two launches).
template <int NOUT>
__global__ __launch_bounds__(256) void avgpool_s1_wave_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int h, int w,
                                                              int kh, int kw) {
  const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;                                // wave-uniform
  const int lane = threadIdx.x & 63;
  const int ho = h - kh + 1, wo = w - kw + 1;
  const float* xp = x + pl * h * w;
  float s[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s[o] = 0.f;
  for (int i = lane; i < h * w; i += 64) {
    const float v = xp[i];
    const int yy = i / w, xx = i - yy * w;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int oy = o / wo, ox = o - oy * wo;             // o < ho * wo <= NOUT checked by the launcher
      const bool in = o < ho * wo && yy >= oy && yy < oy + kh && xx >= ox && xx < ox + kw;
      s[o] += in ? v : 0.f;
    }
  }
  const float inv = 1.f / (float)(kh * kw);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    const float t = cl4::wave_sum(s[o]);
    if (lane == 0 && o < ho * wo) y[pl * ho * wo + o] = t * inv;
  }
}

// F.pad(mode="replicate") — modules/deeplab.py:88.
__global__ void replicate_pad_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int hin, int win,
                                     int pad_t, int pad_l, int hout, int wout) {
  long total = planes * hout * wout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ox = i % wout;
    long t = i / wout;
    int oy = t % hout;
    long pl = t / hout;
    int iy = min(max(oy - pad_t, 0), hin - 1), ix = min(max(ox - pad_l, 0), win - 1);
    y[i] = x[pl * hin * win + iy * win + ix];
  }
}

// Global mean over H*W per plane (AdaptiveAvgPool2d(1), modules/panoptic_deeplab.py:84; deeplab.py:75).
// one wave per plane (maps of a few thousand pixels: no LDS, no barrier; r03)
__global__ __launch_bounds__(256) void global_mean_wave_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int hw) {
  const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const int lane = threadIdx.x & 63;
  const float* xp = x + pl * hw;
  float s = 0.f;
  for (int i = lane; i < hw; i += 64) s += xp[i];
  s = cl4::wave_sum(s);
  if (lane == 0) y[pl] = s / (float)hw;
}

__global__ void global_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int hw) {
  __shared__ float part[4];
  const long pl = blockIdx.x;
  const float* xp = x + pl * hw;
  float s = 0.f;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) s += xp[i];
  s = cl4::wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) y[pl] = (part[0] + part[1] + part[2] + part[3]) / (float)hw;
}

// PyTorch's bilinear source index (upsample_bilinear2d): align_corners=True: src = dst*(in-1)/(out-1);
// align_corners=False: src = max(fma(scale, dst+0.5, -0.5), 0) — the FMA form is what the reference's CPU kernel
// evaluates (see label_common.h). Returns i0, i1 and the weight of i1.
__device__ __forceinline__ void src_index(int dst, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  float src;
  if (align) {
    src = scale * (float)dst;
  } else {
    src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

__host__ __device__ inline float resize_scale(int in_size, int out_size, int align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  return (float)in_size / (float)out_size;
}

struct ResizeArgs {
  int n, c, hin, win, hout, wout, align;
  int in_ctot, in_coff, out_ctot, out_coff;
  float sy, sx;
};

// grid = (n*c planes, chunks of the output plane): one 32-bit division per element instead of four 64-bit ones
__global__ void resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, ResizeArgs a) {
  const int plane = blockIdx.x;
  const int n = plane / a.c, c = plane - n * a.c;
  const int per = a.hout * a.wout;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < per; i += gridDim.y * blockDim.x) {
    const int oy = i / a.wout, ox = i - oy * a.wout;
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, a.sy, a.align, a.hin, y0, y1, ly);
    src_index(ox, a.sx, a.align, a.win, x0, x1, lx);
    const float* xp = x + ((long)n * a.in_ctot + a.in_coff + c) * a.hin * a.win;
    float hy = 1.f - ly, hx = 1.f - lx;
    float t0 = fmaf(xp[y0 * a.win + x0], hx, xp[y0 * a.win + x1] * lx);
    float t1 = fmaf(xp[y1 * a.win + x0], hx, xp[y1 * a.win + x1] * lx);
    float v = fmaf(t0, hy, t1 * ly);
    y[(((long)n * a.out_ctot + a.out_coff + c) * a.hout + oy) * a.wout + ox] = v;
  }
}

// Backward of the resize as a deterministic gather: input pixel (iy,ix) sums the output pixels whose
// forward footprint touched it. The candidate output range is bounded from the scale and each
// candidate recomputes the forward's own (i0,i1,l1), so forward and backward can never disagree.
__global__ void resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, ResizeArgs a, int accumulate) {
  const int plane = blockIdx.x;                // grid = (n*c planes, chunks of the input plane)
  const int n = plane / a.c, c = plane - n * a.c;
  const int per = a.hin * a.win;
  const float ry = a.sy > 0.f ? 1.f / a.sy : 0.f, rx = a.sx > 0.f ? 1.f / a.sx : 0.f;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < per; i += gridDim.y * blockDim.x) {
    const int iy = i / a.win, ix = i - iy * a.win;
    // outputs whose source coordinate falls in (iy-1, iy+1). align_corners: src = dst*s -> dst in ((iy-1)/s, (iy+1)/s);
    // half-pixel: src = (dst+0.5)*s - 0.5 -> dst in ((iy-0.5)/s - 0.5, (iy+1.5)/s - 0.5). One conservative range covers both
    // (the +1.5 matters for large upsampling factors: 33 -> 513 spans 15.5 outputs per input); the clamped borders take
    // everything beyond them. Every candidate re-derives the forward's own (i0, i1, l1), so extra candidates contribute 0.
    int oy_lo = max(0, (int)floorf(((float)iy - 1.f) * ry) - 1), oy_hi = min(a.hout - 1, (int)ceilf(((float)iy + 1.5f) * ry) + 1);
    int ox_lo = max(0, (int)floorf(((float)ix - 1.f) * rx) - 1), ox_hi = min(a.wout - 1, (int)ceilf(((float)ix + 1.5f) * rx) + 1);
    if (iy == a.hin - 1) oy_hi = a.hout - 1;
    if (ix == a.win - 1) ox_hi = a.wout - 1;
    if (iy == 0) oy_lo = 0;
    if (ix == 0) ox_lo = 0;
    if (a.sy == 0.f) { oy_lo = 0; oy_hi = a.hout - 1; }
    if (a.sx == 0.f) { ox_lo = 0; ox_hi = a.wout - 1; }
    const float* dyp = dy + ((long)n * a.out_ctot + a.out_coff + c) * a.hout * a.wout;
    float acc = 0.f;
    constexpr int WXN = 12;                      // column weights once per input pixel, not once per candidate row (the decoder's
    const int nx = ox_hi - ox_lo + 1;            // x2 upsamplings have <= 9 candidates per axis; larger factors take the plain loop)
    if (nx <= WXN) {
      float wxs[WXN];
#pragma unroll
      for (int t = 0; t < WXN; ++t) {
        int x0, x1; float lx;
        src_index(min(ox_lo + t, a.wout - 1), a.sx, a.align, a.win, x0, x1, lx);
        wxs[t] = t < nx ? (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f) : 0.f;
      }
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1; float ly;
        src_index(oy, a.sy, a.align, a.hin, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        const float* row = dyp + oy * a.wout + ox_lo;
#pragma unroll
        for (int t = 0; t < WXN; ++t)            // same products in the same order as the plain loop (zero weights were skipped there)
          if (wxs[t] != 0.f) acc += wy * wxs[t] * row[t];
      }
    } else
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      src_index(oy, a.sy, a.align, a.hin, y0, y1, ly);
      float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        src_index(ox, a.sx, a.align, a.win, x0, x1, lx);
        float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        acc += wy * wx * dyp[oy * a.wout + ox];
      }
    }
    long o = (((long)n * a.in_ctot + a.in_coff + c) * a.hin + iy) * a.win + ix;
    dx[o] = accumulate ? dx[o] + acc : acc;
  }
}

// The same gather for UPSAMPLING (hout >= hin, wout >= win), restructured so that the gradient is read coalesced and once (plus the
// overlap of neighbouring row bands). resize_bwd_kernel's thread = input pixel walks a (2/s + 3)^2 window of dy with a lane stride of
// 1/s floats: 81 dword loads per thread at 129 -> 513, each wave-load spread over 1 KB — 0.6-0.7 ms for the 320 centre-map planes of
// the bench step, ten times their HBM time (profiles/r02g_kernel_stats_bench.csv). Here a workgroup = (plane, band of RBR input
// rows) makes two separable passes: (1) vertical: thread = output column, walks the band's candidate output rows top to bottom
// (lanes read consecutive floats of one row); the forward's first source row y0(oy) never decreases and steps by at most 1 when
// upsampling, so two running sums (rows y0 and y0 + 1) are enough and each finished row is stored once into an LDS image
// t[row][ox]; (2) horizontal: thread = input pixel of the band sums its candidate columns of t. Every candidate re-derives the
// forward's own (i0, i1, l1), like the plain kernel, so forward and backward cannot disagree; only the order of the fp32 sums differs.
// The workgroup has one thread per output column (whole waves, at most 1024): 513 columns on 256 threads would walk the band three
// times, the third time for a single column.
constexpr int RBR = 16;
__global__ __launch_bounds__(1024) void resize_bwd_rows_kernel(const float* __restrict__ dy, float* __restrict__ dx, ResizeArgs a,
                                                              int accumulate) {
  extern __shared__ float tbuf[];                // [RBR][wout]
  const int plane = blockIdx.x;
  const int n = plane / a.c, c = plane - n * a.c;
  const int iy_a = blockIdx.y * RBR, iy_b = min(a.hin, iy_a + RBR);
  const int rows = iy_b - iy_a;
  const float ry = 1.f / a.sy, rx = 1.f / a.sx;  // the launcher routes sy, sx > 0 here
  int oy_lo = max(0, (int)floorf(((float)iy_a - 1.f) * ry) - 1), oy_hi = min(a.hout - 1, (int)ceilf(((float)(iy_b - 1) + 1.5f) * ry) + 1);
  if (iy_b == a.hin) oy_hi = a.hout - 1;
  if (iy_a == 0) oy_lo = 0;
  const float* dyp = dy + ((long)n * a.out_ctot + a.out_coff + c) * a.hout * a.wout;
  for (int i = threadIdx.x; i < rows * a.wout; i += blockDim.x) tbuf[i] = 0.f;
  __syncthreads();
  // Map sizes are 2^k + 1 (33, 65, 129, 513): a wave for the one column past the last multiple of 64 would walk the band for a
  // single lane. Up to 15 such columns are left to pass 1b below (thread = (band row, left-over column), a plain vertical gather).
  const int rem = a.wout & 63;
  const int wmain = (rem && rem < 16 && a.wout >= 64) ? a.wout - rem : a.wout;     // the launcher sizes the workgroup for wmain
  for (int ox = threadIdx.x; ox < wmain; ox += blockDim.x) {
    const float* col = dyp + ox;
    int cur = -2;                                // y0 of the rows being summed: acc0 belongs to input row cur, acc1 to cur + 1
    float acc0 = 0.f, acc1 = 0.f;
    auto put = [&](int row, float v) {
      if (row >= iy_a && row < iy_b) tbuf[(row - iy_a) * a.wout + ox] = v;
    };
    constexpr int UN = 8;                        // rows requested together: the walk is a chain of dependent sums, the loads are not
    for (int oyb = oy_lo; oyb <= oy_hi; oyb += UN) {
      float gv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) gv[u] = col[(long)min(oyb + u, oy_hi) * a.wout];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int oy = oyb + u;
        if (oy > oy_hi) break;                   // uniform
        int y0, y1; float ly;
        src_index(oy, a.sy, a.align, a.hin, y0, y1, ly);
        const float g = gv[u];
        if (y0 != cur) {                         // uniform: oy is
          if (y0 == cur + 1) { put(cur, acc0); acc0 = acc1; }
          else { put(cur, acc0); put(cur + 1, acc1); acc0 = 0.f; }   // first row of the walk (or a jump: not when upsampling)
          acc1 = 0.f;
          cur = y0;
        }
        acc0 += (1.f - ly) * g;
        if (y1 != y0) acc1 += ly * g; else acc0 += ly * g;             // clamped last row: both taps are row y0
      }
    }
    put(cur, acc0);
    put(cur + 1, acc1);
  }
  for (int i = threadIdx.x; i < rows * (a.wout - wmain); i += blockDim.x) {
    const int r = i / (a.wout - wmain), ox = wmain + (i - r * (a.wout - wmain));
    const int iy = iy_a + r;
    int lo = max(0, (int)floorf(((float)iy - 1.f) * ry) - 1), hi = min(a.hout - 1, (int)ceilf(((float)iy + 1.5f) * ry) + 1);
    if (iy == a.hin - 1) hi = a.hout - 1;
    if (iy == 0) lo = 0;
    float acc = 0.f;
    for (int oy = lo; oy <= hi; ++oy) {
      int y0, y1; float ly;
      src_index(oy, a.sy, a.align, a.hin, y0, y1, ly);
      const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
      if (wy != 0.f) acc += wy * dyp[(long)oy * a.wout + ox];
    }
    tbuf[r * a.wout + ox] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * a.win; i += blockDim.x) {
    const int r = i / a.win, ix = i - r * a.win;
    int ox_lo = max(0, (int)floorf(((float)ix - 1.f) * rx) - 1), ox_hi = min(a.wout - 1, (int)ceilf(((float)ix + 1.5f) * rx) + 1);
    if (ix == a.win - 1) ox_hi = a.wout - 1;
    if (ix == 0) ox_lo = 0;
    const float* t = tbuf + r * a.wout;
    float acc = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1; float lx;
      src_index(ox, a.sx, a.align, a.win, x0, x1, lx);
      const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      acc += wx * t[ox];
    }
    const long o = (((long)n * a.in_ctot + a.in_coff + c) * a.hin + iy_a + r) * a.win + ix;
    dx[o] = accumulate ? dx[o] + acc : acc;
  }
}

// Upsampling FROM a 1x1 map (ASPP image pooling, panoptic_deeplab.py:80-98: the pooled vector is broadcast back over the map): every
// output pixel reads the one input pixel with weight 1, so its gradient is the plane's sum. The gather above gave this to ONE thread
// per plane (0.29 ms for 4096 planes of 33x33); here a workgroup reduces the plane.
__global__ void resize_bwd_1x1_kernel(const float* __restrict__ dy, float* __restrict__ dx, ResizeArgs a, int accumulate) {
  __shared__ float part[4];
  const int plane = blockIdx.x;
  const int n = plane / a.c, c = plane - n * a.c;
  const int hw = a.hout * a.wout;
  const float* dyp = dy + ((long)n * a.out_ctot + a.out_coff + c) * hw;
  float s = 0.f;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) s += dyp[i];
  s = cl4::wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    const long o = (long)n * a.in_ctot + a.in_coff + c;
    dx[o] = accumulate ? dx[o] + t : t;
  }
}

// y[r][w-1-x] = x[r][x] (torch.flip(-1), train.py:261).
__global__ void flip_w_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int w) {
  long total = rows * w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xx = i % w;
    long r = i / w;
    y[i] = x[r * w + (w - 1 - xx)];
  }
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] += alpha * x[i];
}

// Copy a channel slice [n, c, hw] between tensors with different channel totals.
__global__ void copy_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c, long hw,
                                     int in_ctot, int in_coff, int out_ctot, int out_coff) {
  long total = (long)n * c * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long p = i % hw;
    long t = i / hw;
    int cc = t % c;
    int nn = t / c;
    y[((long)nn * out_ctot + out_coff + cc) * hw + p] = x[((long)nn * in_ctot + in_coff + cc) * hw + p];
  }
}

}  // namespace

extern "C" int cl4_maxpool3x3s2(const float* x, float* y, long planes, int hin, int win, int hout, int wout, void* stream) {
  CL4_REQUIRE(x && y && planes > 0, "maxpool: bad arguments");
  CL4_REQUIRE(hout == (hin + 2 - 3) / 2 + 1 && wout == (win + 2 - 3) / 2 + 1, "maxpool: output size mismatch");
  static const int rows_kernel = [] { const char* v = getenv("CL4_MAXPOOL_ROWS"); return v ? atoi(v) : 1; }();   // 0: thread per output (experiments)
  if (rows_kernel && planes * hin * win * 4 < (1L << 31) && planes <= 0x7fffffffL) {      // 32-bit buffer offsets
    const int threads = MPK * wout >= 1024 ? 1024 : 64 * cl4::cdiv(MPK * wout, 64);
    hipLaunchKernelGGL(maxpool3x3s2_rows_kernel, dim3((unsigned)planes, cl4::cdiv(hout, MPB * MPK)), dim3(threads), 0, cl4::as_stream(stream),
                       x, y, planes, hin, win, hout, wout);
    return cl4::launch_status();
  }
  const int chunks = cl4::cdiv((long)hout * wout, 1024) < 64 ? cl4::cdiv((long)hout * wout, 1024) : 64;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)planes, chunks), dim3(256), 0, cl4::as_stream(stream), x, y,
                     planes, hin, win, hout, wout);
  return cl4::launch_status();
}

extern "C" int cl4_avgpool_s1(const float* x, float* y, long planes, int h, int w, int kh, int kw, void* stream) {
  CL4_REQUIRE(x && y && planes > 0 && kh >= 1 && kw >= 1 && kh <= h && kw <= w, "avgpool_s1: bad arguments");
  if ((h - kh + 1) * (w - kw + 1) <= 4 && (long)h * w <= 8192) {
    hipLaunchKernelGGL(avgpool_s1_wave_kernel<4>, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, cl4::as_stream(stream), x, y, planes, h, w, kh, kw);
    return cl4::launch_status();
  }
  CL4_REQUIRE((size_t)h * w * 4 <= 64 * 1024, "avgpool_s1: plane %dx%d does not fit the LDS staging", h, w);
  hipLaunchKernelGGL(avgpool_s1_kernel, dim3((unsigned)planes), dim3(256), (size_t)h * w * 4, cl4::as_stream(stream), x, y, h,
                     w, kh, kw);
  return cl4::launch_status();
}

extern "C" int cl4_replicate_pad(const float* x, float* y, long planes, int hin, int win, int pad_t, int pad_l, int hout,
                                 int wout, void* stream) {
  CL4_REQUIRE(x && y && planes > 0 && pad_t >= 0 && pad_l >= 0 && hout >= hin && wout >= win, "replicate_pad: bad arguments");
  long total = planes * hout * wout;
  hipLaunchKernelGGL(replicate_pad_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, y,
                     planes, hin, win, pad_t, pad_l, hout, wout);
  return cl4::launch_status();
}

// Sum over H*W per plane with fp64 partial sums, one wave per plane (r04: the pooling branch of the ASPP backward sums the gradient of the
// projection's pre-activation per plane BEFORE the 1x1 input-gradient product instead of after it — see heads.py, FINDINGS.md finding 41).
__global__ __launch_bounds__(256) void plane_sum_d_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int hw) {
  const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const float* xp = x + pl * hw;
  double s = 0.0;
  for (int i = threadIdx.x & 63; i < hw; i += 64) s += (double)xp[i];
  s = cl4::wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) y[pl] = (float)s;
}

extern "C" int cl4_plane_sum(const float* x, float* y, long planes, int hw, void* stream) {
  CL4_REQUIRE(x && y && planes > 0 && hw > 0, "plane_sum: bad arguments");
  hipLaunchKernelGGL(plane_sum_d_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, cl4::as_stream(stream), x, y, planes, hw);
  return cl4::launch_status();
}

extern "C" int cl4_global_mean(const float* x, float* y, long planes, int hw, void* stream) {
  CL4_REQUIRE(x && y && planes > 0 && hw > 0, "global_mean: bad arguments");
  if (hw <= 8192) {
    hipLaunchKernelGGL(global_mean_wave_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, cl4::as_stream(stream), x, y, planes, hw);
    return cl4::launch_status();
  }
  hipLaunchKernelGGL(global_mean_kernel, dim3((unsigned)planes), dim3(256), 0, cl4::as_stream(stream), x, y, hw);
  return cl4::launch_status();
}

static int fill_resize(ResizeArgs& a, int n, int c, int hin, int win, int hout, int wout, int align, int in_ctot,
                       int in_coff, int out_ctot, int out_coff) {
  CL4_REQUIRE(n > 0 && c > 0 && hin > 0 && win > 0 && hout > 0 && wout > 0, "resize: non-positive dims");
  CL4_REQUIRE(in_coff >= 0 && in_coff + c <= in_ctot && out_coff >= 0 && out_coff + c <= out_ctot, "resize: channel slice out of range");
  a.n = n; a.c = c; a.hin = hin; a.win = win; a.hout = hout; a.wout = wout; a.align = align;
  a.in_ctot = in_ctot; a.in_coff = in_coff; a.out_ctot = out_ctot; a.out_coff = out_coff;
  a.sy = resize_scale(hin, hout, align);
  a.sx = resize_scale(win, wout, align);
  return CL4_OK;
}

extern "C" int cl4_resize_bilinear_fwd(const float* x, float* y, int n, int c, int hin, int win, int hout, int wout,
                                       int align_corners, int in_ctot, int in_coff, int out_ctot, int out_coff, void* stream) {
  CL4_REQUIRE(x && y, "resize_fwd: null tensor");
  ResizeArgs a;
  int rc = fill_resize(a, n, c, hin, win, hout, wout, align_corners, in_ctot, in_coff, out_ctot, out_coff);
  if (rc) return rc;
  long total = (long)n * c * hout * wout;
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(a.n * a.c, cl4::cdiv((long)a.hout * a.wout, 1024) < 64 ? cl4::cdiv((long)a.hout * a.wout, 1024) : 64), dim3(256), 0, cl4::as_stream(stream), x, y, a);
  return cl4::launch_status();
}

extern "C" int cl4_resize_bilinear_bwd(const float* dy, float* dx, int n, int c, int hin, int win, int hout, int wout,
                                       int align_corners, int in_ctot, int in_coff, int out_ctot, int out_coff,
                                       int accumulate, void* stream) {
  CL4_REQUIRE(dy && dx, "resize_bwd: null tensor");
  ResizeArgs a;
  int rc = fill_resize(a, n, c, hin, win, hout, wout, align_corners, in_ctot, in_coff, out_ctot, out_coff);
  if (rc) return rc;
  long total = (long)n * c * hin * win;
  static const int rows_kernel = [] { const char* v = getenv("CL4_RESIZE_BWD_ROWS"); return v ? atoi(v) : 1; }();   // 0: the plain gather (experiments)
  if (rows_kernel && a.hin == 1 && a.win == 1) {
    hipLaunchKernelGGL(resize_bwd_1x1_kernel, dim3(a.n * a.c), dim3(256), 0, cl4::as_stream(stream), dy, dx, a, accumulate);
    return cl4::launch_status();
  }
  if (rows_kernel && a.hout >= a.hin && a.wout >= a.win && a.sy > 0.f && a.sx > 0.f && a.sy <= 1.f && a.sx <= 1.f && (size_t)RBR * a.wout * 4 <= (48u << 10)) {
    const int rem = a.wout & 63;
    const int wmain = (rem && rem < 16 && a.wout >= 64) ? a.wout - rem : a.wout;       // see the kernel: left-over columns take pass 1b
    const int threads = wmain >= 1024 ? 1024 : 64 * cl4::cdiv(wmain, 64);
    hipLaunchKernelGGL(resize_bwd_rows_kernel, dim3(a.n * a.c, cl4::cdiv(a.hin, RBR)), dim3(threads), (size_t)RBR * a.wout * sizeof(float),
                       cl4::as_stream(stream), dy, dx, a, accumulate);
    return cl4::launch_status();
  }
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(a.n * a.c, cl4::cdiv((long)a.hin * a.win, 1024) < 64 ? cl4::cdiv((long)a.hin * a.win, 1024) : 64), dim3(256), 0, cl4::as_stream(stream), dy, dx, a,
                     accumulate);
  return cl4::launch_status();
}

extern "C" int cl4_flip_w(const float* x, float* y, long rows, int w, void* stream) {
  CL4_REQUIRE(x && y && rows > 0 && w > 0 && x != y, "flip_w: bad arguments");
  hipLaunchKernelGGL(flip_w_kernel, dim3(cl4::stream_grid(rows * w, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, rows, w);
  return cl4::launch_status();
}

extern "C" int cl4_axpy(float* y, const float* x, float alpha, long n, void* stream) {
  CL4_REQUIRE(x && y && n > 0, "axpy: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), y, x, alpha, n);
  return cl4::launch_status();
}

extern "C" int cl4_copy_channels(const float* x, float* y, int n, int c, long hw, int in_ctot, int in_coff, int out_ctot,
                                 int out_coff, void* stream) {
  CL4_REQUIRE(x && y && n > 0 && c > 0 && hw > 0, "copy_channels: bad arguments");
  CL4_REQUIRE(in_coff >= 0 && in_coff + c <= in_ctot && out_coff >= 0 && out_coff + c <= out_ctot, "copy_channels: slice out of range");
  long total = (long)n * c * hw;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, n,
                     c, hw, in_ctot, in_coff, out_ctot, out_coff);
  return cl4::launch_status();
}
