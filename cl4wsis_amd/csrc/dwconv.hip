// 5x5 depthwise convolution (stride 1, pad 2) of the Panoptic-DeepLab fuse / head blocks
// (modules/panoptic_deeplab.py:32-47, basic_conv with groups=in_planes) — forward, input gradient
// and weight gradient. Per-channel K=25: HBM/LDS-bound, no MFMA. A workgroup owns a 32x32 output
// tile of one (n, c) plane; the 36x36 halo tile is staged through LDS once and every lane reads its
// 5x5 window from LDS (row stride 37 floats: conflict-free for the 32-lane row reads).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int T = 32;        // output tile edge
constexpr int R = 2;         // halo
constexpr int TS = T + 2 * R;  // 36
constexpr int LDT = TS + 1;    // 37

struct DwArgs {
  int n, c, h, w;
  int x_ctot, x_coff, y_ctot, y_coff;
  int flip;  // 1: correlate with the 180-degree rotated kernel (input gradient)
};

__device__ __forceinline__ void stage_tile(float (*tile)[LDT], const float* __restrict__ xp, int h, int w, int y0, int x0) {
  for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) {
    int ty = i / TS, tx = i - ty * TS;
    int iy = y0 - R + ty, ix = x0 - R + tx;
    tile[ty][tx] = ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? xp[(long)iy * w + ix] : 0.f;
  }
}

__global__ __launch_bounds__(256) void dwconv5_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                      float* __restrict__ y, DwArgs a) {
  __shared__ float tile[TS][LDT];
  __shared__ float wk[25];
  const int tiles_x = (a.w + T - 1) / T;
  const int ty0 = (blockIdx.x / tiles_x) * T, tx0 = (blockIdx.x % tiles_x) * T;
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + ((long)n * a.x_ctot + a.x_coff + c) * a.h * a.w;
  if (threadIdx.x < 25) wk[threadIdx.x] = wgt[c * 25 + (a.flip ? 24 - threadIdx.x : threadIdx.x)];
  stage_tile(tile, xp, a.h, a.w, ty0, tx0);
  __syncthreads();
  const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;  // 8 rows per pass
  float* yp = y + ((long)n * a.y_ctot + a.y_coff + c) * a.h * a.w;
#pragma unroll
  for (int pass = 0; pass < T / 8; ++pass) {
    int ly = ly0 + pass * 8;
    int oy = ty0 + ly, ox = tx0 + lx;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) acc = fmaf(wk[ky * 5 + kx], tile[ly + ky][lx + kx], acc);
    if (oy < a.h && ox < a.w) yp[(long)oy * a.w + ox] = acc;
  }
}


// Streaming form of the same convolution: no LDS, no workgroup barrier. A wave owns a strip of 60 output columns (lanes
// 2..61; the two lanes on either side only carry the halo) and slides down a chunk of rows: per input row one coalesced
// load per lane and four cross-lane shifts give the row's 5-wide window, which is kept for the five output rows it
// contributes to. Loads are issued eight rows ahead, so a CU keeps ~64 KB in flight (the tile kernel above did one
// load -> barrier -> compute -> store round trip per workgroup and ran at 2 TB/s of plane traffic). Every output
// accumulates its 25 taps in the same (ky, kx) order as the tile kernel: bit-identical results.
constexpr int SW = 60;       // output columns per wave
constexpr int SROWS = 32;    // output rows per wave (upper bound; the launch uses equal chunks)

// STATS (r03): the wave also sums its outputs and their squares in fp64 and writes the pair to
// stats[(c * nparts + n * units_pad + unit) * 2 + {0, 1}] (nparts = n_images * units_pad) — the partial sums train-mode BatchNorm needs
// of this tensor (cl4_bn_train_fwd_partials), so that the statistics pass does not read it back.
template <bool STATS>
__global__ __launch_bounds__(256) void dwconv5_strip_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                            float* __restrict__ y, DwArgs a, int strips_x, int srows,
                                                            double* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int plane = blockIdx.x;                // n * c + c
  const int n = plane / a.c, c = plane - n * a.c;
  const int unit = blockIdx.y * 4 + wave;      // (row chunk, strip)
  const int chunk = unit / strips_x, strip = unit - chunk * strips_x;
  const int r0 = chunk * srows;
  double s1 = 0.0, s2 = 0.0;
  const long spos = STATS ? ((long)c * ((long)a.n * gridDim.y * 4) + (long)n * gridDim.y * 4 + unit) * 2 : 0;
  if (r0 >= a.h) {                             // wave-uniform, no barrier in this kernel
    if (STATS && lane == 0) { stats[spos] = 0.0; stats[spos + 1] = 0.0; }
    return;
  }
  const int r1 = min(a.h, r0 + srows);
  const int xs = strip * SW - 2 + lane;        // this lane's column
  const bool xin = (unsigned)xs < (unsigned)a.w;
  const bool xout = lane >= 2 && lane < 2 + SW && xin;
  const float* xp = x + ((long)n * a.x_ctot + a.x_coff + c) * a.h * a.w;
  float* yp = y + ((long)n * a.y_ctot + a.y_coff + c) * a.h * a.w;
  float wr[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) wr[t] = wgt[c * 25 + (a.flip ? 24 - t : t)];   // wave-uniform (scalar loads)
  auto load_row = [&](int iy) { return (xin && (unsigned)iy < (unsigned)a.h) ? xp[(long)iy * a.w + xs] : 0.f; };
  float win[5][5];                             // win[ky][kx]: rows oy-2 .. oy+2 of the current output row
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) win[ky][kx] = 0.f;
  constexpr int AHEAD = 8;
  float q[AHEAD];
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) q[i] = load_row(r0 - 2 + i);
  // input rows r0-2 .. r1+1; output row oy is complete once input row oy+2 has been pushed
  for (int iy = r0 - 2; iy < r1 + 2; iy += AHEAD) {
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
      const float v = q[i];
      q[i] = load_row(iy + i + AHEAD);         // refill the slot for AHEAD rows later (out-of-image rows load nothing)
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) win[ky][kx] = win[ky + 1][kx];
      win[4][2] = v;
      win[4][1] = __shfl_up(v, 1, 64);
      win[4][0] = __shfl_up(v, 2, 64);
      win[4][3] = __shfl_down(v, 1, 64);
      win[4][4] = __shfl_down(v, 2, 64);
      const int oy = iy + i - 2;
      if (oy >= r0 && oy < r1) {               // wave-uniform
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) acc = fmaf(wr[ky * 5 + kx], win[ky][kx], acc);
        if (xout) yp[(long)oy * a.w + xs] = acc;
        if (STATS && xout) { s1 += (double)acc; s2 += (double)acc * (double)acc; }
      }
    }
  }
  if (STATS) {
    s1 = cl4::wave_sum_d(s1);
    s2 = cl4::wave_sum_d(s2);
    if (lane == 0) { stats[spos] = s1; stats[spos + 1] = s2; }
  }
}

// dw[c][ky][kx] += sum_{n,y,x} dy[n,c,y,x] * x[n,c,y+ky-2,x+kx-2]. grid (tile groups, c, n): a workgroup walks its tiles
// of one (n, c) plane with 25 per-lane partial sums and pays the 25 wave reductions + atomics once at the end (one
// reduction per tile made the reductions, not the 25 FMAs per pixel, the cost of this kernel).
__global__ __launch_bounds__(256) void dwconv5_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, DwArgs a) {
  __shared__ float tile[TS][LDT];
  __shared__ float red[4][25];
  const int tiles_x = (a.w + T - 1) / T;
  const int ntiles = tiles_x * ((a.h + T - 1) / T);
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + ((long)n * a.x_ctot + a.x_coff + c) * a.h * a.w;
  const float* dp = dy + ((long)n * a.y_ctot + a.y_coff + c) * a.h * a.w;
  const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;
  float acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = 0.f;
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int ty0 = (tl / tiles_x) * T, tx0 = (tl % tiles_x) * T;
    __syncthreads();                       // the previous tile has been consumed
    stage_tile(tile, xp, a.h, a.w, ty0, tx0);
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < T / 8; ++pass) {
      int ly = ly0 + pass * 8;
      int oy = ty0 + ly, ox = tx0 + lx;
      float g = (oy < a.h && ox < a.w) ? dp[(long)oy * a.w + ox] : 0.f;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] = fmaf(g, tile[ly + ky][lx + kx], acc[ky * 5 + kx]);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    float s = cl4::wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < 25) {
    float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&dw[c * 25 + threadIdx.x], s);
  }
}


// Streaming weight gradient: the same strips / row chunks / sliding window as dwconv5_strip_kernel; each lane adds
// dy[oy][x] * window to 25 partial sums, the four waves of a workgroup are reduced through LDS and one atomic per tap and
// workgroup goes to dw.
__global__ __launch_bounds__(256) void dwconv5_wgrad_strip_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dw, DwArgs a, int strips_x, int srows) {
  __shared__ float red[4][25];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int plane = blockIdx.x;
  const int n = plane / a.c, c = plane - n * a.c;
  const int unit = blockIdx.y * 4 + wave;
  const int chunk = unit / strips_x, strip = unit - chunk * strips_x;
  const int r0 = chunk * srows;
  const int r1 = min(a.h, r0 + srows);         // r0 >= h: empty range, the wave only joins the reduction
  const int xs = strip * SW - 2 + lane;
  const bool xin = (unsigned)xs < (unsigned)a.w;
  const bool xout = lane >= 2 && lane < 2 + SW && xin;
  const float* xp = x + ((long)n * a.x_ctot + a.x_coff + c) * a.h * a.w;
  const float* dp = dy + ((long)n * a.y_ctot + a.y_coff + c) * a.h * a.w;
  float acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = 0.f;
  if (r0 < a.h) {
    auto load_row = [&](int iy) { return (xin && (unsigned)iy < (unsigned)a.h) ? xp[(long)iy * a.w + xs] : 0.f; };
    auto load_dy = [&](int oy) { return (xout && oy >= r0 && oy < r1) ? dp[(long)oy * a.w + xs] : 0.f; };
    float win[5][5];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) win[ky][kx] = 0.f;
    constexpr int AHEAD = 8;
    float q[AHEAD], gq[AHEAD];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) { q[i] = load_row(r0 - 2 + i); gq[i] = load_dy(r0 - 4 + i); }
    for (int iy = r0 - 2; iy < r1 + 2; iy += AHEAD) {
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) {
        const float v = q[i], g = gq[i];         // input row iy+i completes the window of output row iy+i-2
        q[i] = load_row(iy + i + AHEAD);
        gq[i] = load_dy(iy + i - 2 + AHEAD);
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) win[ky][kx] = win[ky + 1][kx];
        win[4][2] = v;
        win[4][1] = __shfl_up(v, 1, 64);
        win[4][0] = __shfl_up(v, 2, 64);
        win[4][3] = __shfl_down(v, 1, 64);
        win[4][4] = __shfl_down(v, 2, 64);
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] = fmaf(g, win[ky][kx], acc[ky * 5 + kx]);   // g = 0 outside the chunk
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const float sm = cl4::wave_sum(acc[t]);
    if (lane == 0) red[wave][t] = sm;
  }
  __syncthreads();
  if (threadIdx.x < 25) {
    const float sm = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&dw[c * 25 + threadIdx.x], sm);
  }
}

int check(const DwArgs& a) {
  CL4_REQUIRE(a.n > 0 && a.c > 0 && a.h > 0 && a.w > 0, "dwconv5: non-positive dims");
  CL4_REQUIRE(a.n <= 65535 && a.c <= 65535, "dwconv5: n/c exceed grid limits");
  CL4_REQUIRE(a.x_coff >= 0 && a.x_coff + a.c <= a.x_ctot && a.y_coff >= 0 && a.y_coff + a.c <= a.y_ctot,
              "dwconv5: channel slice out of range");
  return CL4_OK;
}

}  // namespace

extern "C" int cl4_dwconv5_fwd(const float* x, const float* w, float* y, int n, int c, int h, int wd, int x_ctot, int x_coff,
                               int y_ctot, int y_coff, int flip, void* stream) {
  DwArgs a{n, c, h, wd, x_ctot, x_coff, y_ctot, y_coff, flip};
  int rc = check(a);
  if (rc) return rc;
  CL4_REQUIRE(x && w && y, "dwconv5_fwd: null pointer");
  static const int tile_form = getenv("CL4_DWCONV_TILE") ? atoi(getenv("CL4_DWCONV_TILE")) : 0;   // 1: the LDS tile kernel
  if (tile_form) {
    dim3 g(cl4::cdiv(h, T) * cl4::cdiv(wd, T), c, n);
    hipLaunchKernelGGL(dwconv5_kernel, g, dim3(256), 0, cl4::as_stream(stream), x, w, y, a);
  } else {
    const int strips_x = cl4::cdiv(wd, SW);
    const int srows = cl4::cdiv(h, cl4::cdiv(h, SROWS));      // equal chunks of <= SROWS rows (129 rows: 5 x 26, not 4 x 32 + 1)
    const int units = strips_x * cl4::cdiv(h, srows);
    dim3 g((unsigned)n * c, cl4::cdiv(units, 4));
    hipLaunchKernelGGL(dwconv5_strip_kernel<false>, g, dim3(256), 0, cl4::as_stream(stream), x, w, y, a, strips_x, srows, (double*)nullptr);
  }
  return cl4::launch_status();
}

// Partial sums per channel cl4_dwconv5_fwd_stats writes for an n x (h x w) launch (the `nparts` of cl4_bn_train_fwd_partials).
extern "C" int cl4_dwconv5_stats_parts(int n, int h, int wd) {
  const int strips_x = cl4::cdiv(wd, SW);
  const int srows = cl4::cdiv(h, cl4::cdiv(h, SROWS));
  const int units = strips_x * cl4::cdiv(h, srows);
  return n * cl4::cdiv(units, 4) * 4;
}

// cl4_dwconv5_fwd (flip = 0) that also leaves the per-channel partial sums {sum y, sum y^2} of its output in `stats`
// (c * cl4_dwconv5_stats_parts(n, h, w) pairs of doubles): the depthwise 5x5 convolutions of the trainable instance decoder / heads are
// followed by a train-mode BatchNorm (modules/panoptic_deeplab.py:26-28,44-46), whose statistics pass then need not read y again.
extern "C" int cl4_dwconv5_fwd_stats(const float* x, const float* w, float* y, int n, int c, int h, int wd, int x_ctot, int x_coff,
                                     int y_ctot, int y_coff, double* stats, void* stream) {
  DwArgs a{n, c, h, wd, x_ctot, x_coff, y_ctot, y_coff, 0};
  int rc = check(a);
  if (rc) return rc;
  CL4_REQUIRE(x && w && y && stats, "dwconv5_fwd_stats: null pointer");
  const int strips_x = cl4::cdiv(wd, SW);
  const int srows = cl4::cdiv(h, cl4::cdiv(h, SROWS));
  const int units = strips_x * cl4::cdiv(h, srows);
  dim3 g((unsigned)n * c, cl4::cdiv(units, 4));
  hipLaunchKernelGGL(dwconv5_strip_kernel<true>, g, dim3(256), 0, cl4::as_stream(stream), x, w, y, a, strips_x, srows, stats);
  return cl4::launch_status();
}

extern "C" int cl4_dwconv5_wgrad(const float* x, const float* dy, float* dw, int n, int c, int h, int wd, int x_ctot,
                                 int x_coff, int dy_ctot, int dy_coff, void* stream) {
  DwArgs a{n, c, h, wd, x_ctot, x_coff, dy_ctot, dy_coff, 0};
  int rc = check(a);
  if (rc) return rc;
  CL4_REQUIRE(x && dy && dw, "dwconv5_wgrad: null pointer");
  static const int tile_form = getenv("CL4_DWCONV_TILE") ? atoi(getenv("CL4_DWCONV_TILE")) : 0;   // 1: the LDS tile kernel
  if (tile_form) {
    const int ntiles = cl4::cdiv(h, T) * cl4::cdiv(wd, T);
    int groups = cl4::cdiv(2048, (long)c * n);           // ~8 workgroups per CU in total
    if (groups > ntiles) groups = ntiles;
    if (groups < 1) groups = 1;
    dim3 g(groups, c, n);
    hipLaunchKernelGGL(dwconv5_wgrad_kernel, g, dim3(256), 0, cl4::as_stream(stream), x, dy, dw, a);
  } else {
    const int strips_x = cl4::cdiv(wd, SW);
    const int srows = cl4::cdiv(h, cl4::cdiv(h, SROWS));
    const int units = strips_x * cl4::cdiv(h, srows);
    dim3 g((unsigned)n * c, cl4::cdiv(units, 4));
    hipLaunchKernelGGL(dwconv5_wgrad_strip_kernel, g, dim3(256), 0, cl4::as_stream(stream), x, dy, dw, a, strips_x, srows);
  }
  return cl4::launch_status();
}
