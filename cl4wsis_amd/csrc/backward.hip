// Building blocks of the train-mode forward / backward of the ResNet body, DeepLabV3 head and PseudoLabeler (phase 1 /
// step 0; SURVEY.md §8 A1-A4, A8 with gradients). All HBM-bound streaming kernels:
//   * zero_insert        : dy of a strided conv spread onto the stride-1 grid, so that the input gradient is the ordinary
//                          forward kernel run with transposed / flipped weights (models/resnet.py strided 3x3 / 1x1);
//   * maxpool3x3s2_bwd   : nn.MaxPool2d(3, 2, 1) backward as a deterministic gather (first maximum wins, like ATen);
//   * add_act / act_bwd  : residual add + in-place leaky-ReLU of modules/residual.py:91-93 and its derivative;
//   * transpose_flip     : w[co][ci][ky][kx] -> wt[ci][co][k-1-ky][k-1-kx], the weight of the input-gradient conv.
#include "common.h"
#include "planes_fmt.h"

namespace {

__global__ void zero_insert_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int h, int w, int stride, int H,
                                   int W) {
  const long total = planes * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = i % W;
    const long t = i / W;
    const int Y = t % H;
    const long pl = t / H;
    float v = 0.f;
    if (Y % stride == 0 && X % stride == 0) {
      const int yy = Y / stride, xx = X / stride;
      if (yy < h && xx < w) v = x[(pl * h + yy) * w + xx];
    }
    y[i] = v;
  }
}

// argmax of the 3x3 stride-2 pad-1 window (oy, ox): first maximum in row-major order (ATen: `val > maxval`).
__device__ __forceinline__ int window_argmax(const float* __restrict__ xp, int oy, int ox, int hin, int win) {
  float m = -INFINITY;
  int arg = -1;
  const int iy0 = oy * 2 - 1, ix0 = ox * 2 - 1;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = iy0 + dy;
    if ((unsigned)iy >= (unsigned)hin) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ix0 + dx;
      if ((unsigned)ix >= (unsigned)win) continue;
      const float v = xp[iy * win + ix];
      if (v > m || arg < 0) { m = v; arg = iy * win + ix; }
    }
  }
  return arg;
}

__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long planes,
                                        int hin, int win, int hout, int wout) {
  const long pl = blockIdx.x;                  // grid = (planes, chunks of the input plane)
  const int per = hin * win;
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < per; i += gridDim.y * blockDim.x) {
    const int iy = i / win, ix = i - iy * win;
    const float* xp = x + pl * hin * win;
    const float* dp = dy + pl * hout * wout;
    // windows containing (iy, ix): 2*o - 1 <= i <= 2*o + 1  ->  o in [ceil((i-1)/2), floor((i+1)/2)]
    const int oy_lo = iy / 2, oy_hi = min(hout - 1, (iy + 1) / 2);
    const int ox_lo = ix / 2, ox_hi = min(wout - 1, (ix + 1) / 2);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox)
        if (window_argmax(xp, oy, ox, hin, win) == iy * win + ix) acc += dp[oy * wout + ox];
    dx[pl * per + i] = acc;
  }
}

// The same gather with every window's argmax computed ONCE (r05): a workgroup owns MP_ROWS input rows of a plane, first leaves the argmax of the
// <= MP_ROWS / 2 + 1 window rows that touch them in LDS (9 loads per window = 2.25 per input element), then every input element looks its <= 4
// windows up there. maxpool3x3s2_bwd_kernel recomputes up to four 9-element argmaxes per input element: 1.56 ms for the stem of a 16-image
// phase-1 step (68 M elements, ~25 cached loads each); this one is bound by its 0.6 GB of traffic. Same comparisons, same accumulation order:
// bit-identical.
constexpr int MP_ROWS = 16;
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_tile_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                                    int hin, int win, int hout, int wout) {
  extern __shared__ int warg[];                // [MP_ROWS / 2 + 1][wout]
  const long pl = blockIdx.x;
  const int r0 = blockIdx.y * MP_ROWS, r1 = min(r0 + MP_ROWS, hin);
  const float* xp = x + pl * hin * win;
  const float* dp = dy + pl * hout * wout;
  const int oy0 = r0 / 2, oy1 = min(hout - 1, r1 / 2);            // window rows touching input rows r0 .. r1 - 1: iy / 2 .. (iy + 1) / 2
  const int nwin = (oy1 - oy0 + 1) * wout;
  for (int t = threadIdx.x; t < nwin; t += blockDim.x) {
    const int oy = oy0 + t / wout, ox = t - (t / wout) * wout;
    warg[t] = window_argmax(xp, oy, ox, hin, win);
  }
  __syncthreads();
  const int nel = (r1 - r0) * win;
  for (int i = threadIdx.x; i < nel; i += blockDim.x) {
    const int iy = r0 + i / win, ix = i - (i / win) * win;
    const int oy_lo = iy / 2, oy_hi = min(hout - 1, (iy + 1) / 2);
    const int ox_lo = ix / 2, ox_hi = min(wout - 1, (ix + 1) / 2);
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox)
        if (warg[(oy - oy0) * wout + ox] == iy * win + ix) acc += dp[oy * wout + ox];
    dx[pl * hin * win + (long)iy * win + ix] = acc;
  }
}

__global__ void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n, int act,
                               float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = cl4::apply_act(a[i] + b[i], act, slope);
}

// add_act for contiguous [n][c][hw] tensors with c % 8 == 0 that also writes the two-plane twin of y (r05, see bn_apply_planes_kernel)
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
__global__ void add_act_planes_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, u32x4b* __restrict__ ys,
                                      int n, int c, int hw, int act, float slope) {
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
    for (int j = 0; j < 8; ++j) v[j] = cl4::apply_act(a[base + (long)j * hw] + b[base + (long)j * hw], act, slope);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[base + (long)j * hw] = v[j];
    u32x4b hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h_, l_;
      plfmt::split2(v[2 * q], v[2 * q + 1], h_, l_);
      hi[q] = h_;
      lo[q] = l_;
    }
    u32x4b* dst = ys + (((long)im * c8 + g) * 2) * hw + px;
    dst[0] = hi;
    dst[hw] = lo;
  }
}

// dx = dy * act'(pre-activation); for ReLU / leaky-ReLU the sign of the output y equals the sign of the pre-activation.
__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long n, int act,
                               float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = dy[i];
    float r = d;
    if (act == CL4_ACT_RELU) r = y[i] > 0.f ? d : 0.f;
    else if (act == CL4_ACT_LEAKY) r = y[i] > 0.f ? d : d * slope;
    dx[i] = r;
  }
}

// y[pl][:] += scale * v[pl]: the broadcast add of the image-pooling branch (modules/deeplab.py:64-67) and, with
// scale = 1/hw, the backward of the global mean.
__global__ void add_per_plane_kernel(float* __restrict__ y, const float* __restrict__ v, long planes, int hw, float scale) {
  const long total = planes * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    y[i] += scale * v[i / hw];
}

// nn.Dropout2d (models/wide_resnet.py:150-155, train mode of mod6 / mod7): a whole (image, channel) plane is dropped or scaled by
// 1 / (1 - p); `keep` is one byte per plane. The backward is the same operation on the gradient.
__global__ void dropout2d_kernel(const float* __restrict__ x, const unsigned char* __restrict__ keep, float* __restrict__ y, long planes, int hw,
                                 float scale) {
  const long total = planes * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    y[i] = keep[i / hw] ? x[i] * scale : 0.f;
}

// torch.rot90(x, k, dims=(-2,-1)) on square planes: out[i][j] = x[j][h-1-i] (k=1), x[h-1-i][h-1-j] (k=2), x[h-1-j][i] (k=3).
__global__ void rot90_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int h, int k) {
  const long hw = (long)h * h, total = planes * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pl = i / hw;
    const int q = (int)(i - pl * hw), r = q / h, c = q % h;
    int sr = r, sc = c;
    if (k == 1) { sr = c; sc = h - 1 - r; }
    else if (k == 2) { sr = h - 1 - r; sc = h - 1 - c; }
    else if (k == 3) { sr = h - 1 - c; sc = r; }
    y[i] = x[pl * hw + (long)sr * h + sc];
  }
}

// y[n][c][:] = x[n][c][:] * scale[c] + shift[c] (utils/utils.py:23-40 `denorm`: t.mul_(s).add_(m), no contraction).
__global__ void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int c, int hw,
                                      const float* __restrict__ scale, const float* __restrict__ shift) {
  const long total = planes * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)((i / hw) % c);
    const float t = x[i] * scale[ch];
    y[i] = t + shift[ch];
  }
}

__global__ void transpose_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin, int k) {
  const long total = (long)cout * cin * k * k;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kx = i % k;
    long t = i / k;
    const int ky = t % k;
    t /= k;
    const int ci = t % cin;
    const int co = t / cin;
    wt[(((long)ci * cout + co) * k + (k - 1 - ky)) * k + (k - 1 - kx)] = w[i];
  }
}

}  // namespace

extern "C" int cl4_zero_insert(const float* x, float* y, long planes, int h, int w, int stride, int H, int W, void* stream) {
  CL4_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && stride >= 1, "zero_insert: bad arguments");
  CL4_REQUIRE(H >= (h - 1) * stride + 1 && W >= (w - 1) * stride + 1, "zero_insert: output %dx%d too small", H, W);
  const long total = planes * H * W;
  hipLaunchKernelGGL(zero_insert_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, planes, h,
                     w, stride, H, W);
  return cl4::launch_status();
}

extern "C" int cl4_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, long planes, int hin, int win, int hout, int wout,
                                    void* stream) {
  CL4_REQUIRE(x && dy && dx && planes > 0, "maxpool_bwd: bad arguments");
  CL4_REQUIRE(hout == (hin + 2 - 3) / 2 + 1 && wout == (win + 2 - 3) / 2 + 1, "maxpool_bwd: output size mismatch");
  static const bool tiled = [] { const char* e = getenv("CL4_MAXPOOL_BWD_TILE"); return !e || atoi(e) != 0; }();
  const size_t lds = (size_t)(MP_ROWS / 2 + 1) * wout * sizeof(int);
  if (tiled && lds <= 48 * 1024 && cl4::cdiv(hin, MP_ROWS) <= 65535) {
    hipLaunchKernelGGL(maxpool3x3s2_bwd_tile_kernel, dim3((unsigned)planes, cl4::cdiv(hin, MP_ROWS)), dim3(256), lds, cl4::as_stream(stream), x, dy, dx,
                       hin, win, hout, wout);
    return cl4::launch_status();
  }
  const int chunks = cl4::cdiv((long)hin * win, 1024) < 64 ? cl4::cdiv((long)hin * win, 1024) : 64;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)planes, chunks), dim3(256), 0, cl4::as_stream(stream), x, dy, dx,
                     planes, hin, win, hout, wout);
  return cl4::launch_status();
}

extern "C" int cl4_add_act(const float* a, const float* b, float* y, long n, int act, float slope, void* stream) {
  CL4_REQUIRE(a && b && y && n > 0, "add_act: bad arguments");
  hipLaunchKernelGGL(add_act_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), a, b, y, n, act, slope);
  return cl4::launch_status();
}

extern "C" int cl4_add_act_planes(const float* a, const float* b, float* y, void* ys, int n, int c, int hw, int act, float slope, void* stream) {
  CL4_REQUIRE(a && b && y && ys && n > 0 && c > 0 && hw > 0 && c % 8 == 0, "add_act_planes: bad arguments (c %% 8 == 0)");
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(add_act_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), a, b, y, (u32x4b*)ys, n, c, hw, act, slope);
  return cl4::launch_status();
}

extern "C" int cl4_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float slope, void* stream) {
  CL4_REQUIRE(y && dy && dx && n > 0, "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), y, dy, dx, n, act, slope);
  return cl4::launch_status();
}

extern "C" int cl4_add_per_plane(float* y, const float* v, long planes, int hw, float scale, void* stream) {
  CL4_REQUIRE(y && v && planes > 0 && hw > 0, "add_per_plane: bad arguments");
  hipLaunchKernelGGL(add_per_plane_kernel, dim3(cl4::stream_grid(planes * hw, 256)), dim3(256), 0, cl4::as_stream(stream), y, v,
                     planes, hw, scale);
  return cl4::launch_status();
}

extern "C" int cl4_dropout2d(const float* x, const unsigned char* keep, float* y, long planes, int hw, float p, void* stream) {
  CL4_REQUIRE(x && keep && y && planes > 0 && hw > 0 && p >= 0.f && p < 1.f, "dropout2d: bad arguments");
  hipLaunchKernelGGL(dropout2d_kernel, dim3(cl4::stream_grid(planes * hw, 256)), dim3(256), 0, cl4::as_stream(stream), x, keep, y, planes, hw,
                     1.f / (1.f - p));
  return cl4::launch_status();
}

extern "C" int cl4_rot90(const float* x, float* y, long planes, int h, int w, int k, void* stream) {
  CL4_REQUIRE(x && y && x != y && planes > 0 && h > 0 && h == w, "rot90: square planes expected");
  k = ((k % 4) + 4) % 4;
  hipLaunchKernelGGL(rot90_kernel, dim3(cl4::stream_grid(planes * h * h, 256)), dim3(256), 0, cl4::as_stream(stream), x, y, planes, h, k);
  return cl4::launch_status();
}

extern "C" int cl4_channel_affine(const float* x, float* y, int n, int c, int hw, const float* scale, const float* shift, void* stream) {
  CL4_REQUIRE(x && y && scale && shift && n > 0 && c > 0 && hw > 0, "channel_affine: bad arguments");
  const long planes = (long)n * c;
  hipLaunchKernelGGL(channel_affine_kernel, dim3(cl4::stream_grid(planes * hw, 256)), dim3(256), 0, cl4::as_stream(stream), x, y,
                     planes, c, hw, scale, shift);
  return cl4::launch_status();
}

extern "C" int cl4_transpose_flip_weight(const float* w, float* wt, int cout, int cin, int k, void* stream) {
  CL4_REQUIRE(w && wt && w != wt && cout > 0 && cin > 0 && k > 0, "transpose_flip_weight: bad arguments");
  const long total = (long)cout * cin * k * k;
  hipLaunchKernelGGL(transpose_flip_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), w, wt, cout,
                     cin, k);
  return cl4::launch_status();
}
