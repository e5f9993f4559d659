// Phase-2 loss stack and optimiser (HBM-bound):
//   * label blending + the four weighted centre / offset losses of train.py:503-522 with
//     utils/loss.py:228-275 (Weighted_L1_Loss / Weighted_MSELoss), fused with the bilinear upsample
//     (align_corners=False) of the low-resolution network outputs (train.py:433-434,503-504), forward
//     value and the gradient with respect to the low-resolution centre / offset maps;
//   * Adam (torch.optim.Adam semantics, train.py:181-182) over flat parameter / gradient / state buffers.
#include "label_common.h"

extern "C" int cl4_resize_bilinear_bwd(const float* dy, float* dx, int n, int c, int hin, int win, int hout, int wout,
                                       int align_corners, int in_ctot, int in_coff, int out_ctot, int out_coff,
                                       int accumulate, void* stream);

namespace {

using cl4::Bilin;
using cl4::bilin_half_pixel;
using cl4::bilin_sample;

struct LossArgs {
  int b, H, W, hc, wc, cold, cnew;
  const float *center, *offset, *old_center, *old_offset, *old_w;
  const float *p_center, *p_offset, *p_weight, *r_center, *r_offset, *r_weight;
  float center_weight, offset_weight;
  double* acc;       // [0..3] loss sums, [4] n1 = #{old_w > 0}, [5] n2 = #{max(pw, wr) > 0}
  float* g_center;   // [b, cold+cnew, H, W] full-resolution gradient (or null)
  float* g_offset;   // [b, 2, H, W]
  float* losses;     // [4]
};

__device__ void block_accumulate(double* vals, int n, double* acc) {
  __shared__ double red[4][8];
  for (int k = 0; k < n; ++k) {
    double v = cl4::wave_sum_d(vals[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < n) {
    double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (s != 0.0) atomicAdd(&acc[threadIdx.x], s);
  }
}

__global__ __launch_bounds__(256) void loss_count_kernel(LossArgs a) {
  const long N = (long)a.b * a.H * a.W;
  double v[2] = {0.0, 0.0};
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    float w1 = a.old_w[p];
    float w2 = a.p_weight[p];
    if (a.r_weight) w2 = fmaxf(w2, a.r_weight[p]);
    v[0] += w1 > 0.f ? 1.0 : 0.0;
    v[1] += w2 > 0.f ? 1.0 : 0.0;
  }
  block_accumulate(v, 2, a.acc + 4);
}

__global__ __launch_bounds__(256) void loss_main_kernel(LossArgs a) {
  const long N = (long)a.b * a.H * a.W, HW = (long)a.H * a.W;
  const float sy = (float)a.hc / (float)a.H, sx = (float)a.wc / (float)a.W;
  const long lo_plane = (long)a.hc * a.wc;
  const double n1 = a.acc[4], n2 = a.acc[5];
  // d(loss_i)/d(sum_i) — 0.5 * weight / count (utils/loss.py: loss.sum() / (weight > 0).sum(); 0 when empty)
  const float k_c1 = n1 > 0 ? (float)(0.5 * a.center_weight / n1) : 0.f;
  const float k_o1 = n1 > 0 ? (float)(0.5 * a.offset_weight / n1) : 0.f;
  const float k_c2 = n2 > 0 ? (float)(0.5 * a.center_weight / n2) : 0.f;
  const float k_o2 = n2 > 0 ? (float)(0.5 * a.offset_weight / n2) : 0.f;
  const int ctot = a.cold + a.cnew;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    const int b = p / HW;
    const long q = p - (long)b * HW;
    const float w1 = a.old_w[p];
    const float pw = a.p_weight[p];
    const float wr = a.r_weight ? a.r_weight[p] : 0.f;
    const float w2 = a.r_weight ? fmaxf(pw, wr) : pw;
    if (w1 == 0.f && w2 == 0.f) {
      if (a.g_center) {
        for (int c = 0; c < ctot; ++c) a.g_center[((long)b * ctot + c) * HW + q] = 0.f;
        a.g_offset[((long)b * 2) * HW + q] = 0.f;
        a.g_offset[((long)b * 2 + 1) * HW + q] = 0.f;
      }
      continue;
    }
    const int Y = q / a.W, X = q - (long)Y * a.W;
    const Bilin by = bilin_half_pixel(Y, sy, a.hc), bx = bilin_half_pixel(X, sx, a.wc);
    // old classes: distil the frozen model's centre map under the old-class foreground weight
    for (int c = 0; c < a.cold; ++c) {
      float g = 0.f;
      if (w1 != 0.f) {
        float u = bilin_sample(a.center + ((long)b * ctot + c) * lo_plane, a.wc, by, bx);
        float t = bilin_sample(a.old_center + ((long)b * a.cold + c) * lo_plane, a.wc, by, bx);
        float d = u - t;
        v[0] += (double)(d * d * w1);
        g = 2.f * d * w1 * k_c1;
      }
      if (a.g_center) a.g_center[((long)b * ctot + c) * HW + q] = g;
    }
    // new classes: blended pseudo / refined centre target (train.py:509,511)
    for (int c = 0; c < a.cnew; ++c) {
      float g = 0.f;
      if (w2 != 0.f) {
        float u = bilin_sample(a.center + ((long)b * ctot + a.cold + c) * lo_plane, a.wc, by, bx);
        float t = a.p_center[((long)b * a.cnew + c) * HW + q];
        if (a.r_center) t = pw * t + (1.f - pw) * a.r_center[((long)b * a.cnew + c) * HW + q];
        float d = u - t;
        v[2] += (double)(d * d * w2);
        g = 2.f * d * w2 * k_c2;
      }
      if (a.g_center) a.g_center[((long)b * ctot + a.cold + c) * HW + q] = g;
    }
    const float pws = fmaxf(w1, pw);  // pseudo_weight_sum, train.py:506
    for (int o = 0; o < 2; ++o) {
      float u = bilin_sample(a.offset + ((long)b * 2 + o) * lo_plane, a.wc, by, bx);
      float g = 0.f;
      if (w1 != 0.f) {
        float t = bilin_sample(a.old_offset + ((long)b * 2 + o) * lo_plane, a.wc, by, bx);
        float d = u - t;
        v[1] += (double)(fabsf(d) * w1);
        g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w1 * k_o1;
      }
      if (w2 != 0.f) {
        float t = a.p_offset[((long)b * 2 + o) * HW + q];
        if (a.r_offset) t = pws * t + (1.f - pws) * a.r_offset[((long)b * 2 + o) * HW + q];
        float d = u - t;
        v[3] += (double)(fabsf(d) * w2);
        g += (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w2 * k_o2;
      }
      if (a.g_offset) a.g_offset[((long)b * 2 + o) * HW + q] = g;
    }
  }
  block_accumulate(v, 4, a.acc);
}

__global__ void loss_final_kernel(LossArgs a) {
  const double n1 = a.acc[4], n2 = a.acc[5];
  a.losses[0] = n1 > 0 ? (float)(0.5 * a.center_weight * a.acc[0] / n1) : 0.f;
  a.losses[1] = n1 > 0 ? (float)(0.5 * a.offset_weight * a.acc[1] / n1) : 0.f;
  a.losses[2] = n2 > 0 ? (float)(0.5 * a.center_weight * a.acc[2] / n2) : 0.f;
  a.losses[3] = n2 > 0 ? (float)(0.5 * a.offset_weight * a.acc[3] / n2) : 0.f;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float beta1, float beta2, float eps, float wd, float step_size, float bc2_sqrt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

}  // namespace

extern "C" size_t cl4_p2_loss_workspace_bytes(int b, int H, int W, int cold, int cnew, int with_grad) {
  size_t bytes = 256;  // accumulators
  if (with_grad) bytes += (size_t)b * (cold + cnew + 2) * H * W * sizeof(float);
  return bytes;
}

extern "C" int cl4_p2_loss(const cl4_p2_loss_io* d, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(d && workspace, "p2_loss: null pointer");
  CL4_REQUIRE(d->b > 0 && d->H > 0 && d->W > 0 && d->hc > 0 && d->wc > 0 && d->cold >= 0 && d->cnew > 0, "p2_loss: bad dims");
  CL4_REQUIRE(d->center && d->offset && d->old_w && d->p_center && d->p_offset && d->p_weight && d->losses, "p2_loss: null tensor");
  CL4_REQUIRE(d->cold == 0 || (d->old_center && d->old_offset), "p2_loss: old-model maps missing");
  CL4_REQUIRE((d->r_center != nullptr) == (d->r_offset != nullptr) && (d->r_center != nullptr) == (d->r_weight != nullptr),
              "p2_loss: refined maps must be all present or all absent");
  const int with_grad = d->d_center != nullptr;
  CL4_REQUIRE(with_grad == (d->d_offset != nullptr), "p2_loss: d_center / d_offset must come together");
  if (workspace_bytes < cl4_p2_loss_workspace_bytes(d->b, d->H, d->W, d->cold, d->cnew, with_grad)) {
    cl4::set_error("p2_loss: workspace too small");
    return CL4_ENOSPC;
  }
  hipStream_t st = cl4::as_stream(stream);
  LossArgs a;
  a.b = d->b; a.H = d->H; a.W = d->W; a.hc = d->hc; a.wc = d->wc; a.cold = d->cold; a.cnew = d->cnew;
  a.center = d->center; a.offset = d->offset; a.old_center = d->old_center; a.old_offset = d->old_offset; a.old_w = d->old_w;
  a.p_center = d->p_center; a.p_offset = d->p_offset; a.p_weight = d->p_weight;
  a.r_center = d->r_center; a.r_offset = d->r_offset; a.r_weight = d->r_weight;
  a.center_weight = d->center_weight; a.offset_weight = d->offset_weight;
  a.acc = (double*)workspace;
  a.losses = d->losses;
  const long N = (long)d->b * d->H * d->W;
  const int ctot = d->cold + d->cnew;
  a.g_center = with_grad ? (float*)((char*)workspace + 256) : nullptr;
  a.g_offset = with_grad ? a.g_center + (size_t)d->b * ctot * d->H * d->W : nullptr;
  cl4::fill_bytes_async(a.acc, 0, 64, st);
  const int g = cl4::stream_grid(N, 256);
  hipLaunchKernelGGL(loss_count_kernel, dim3(g), dim3(256), 0, st, a);
  hipLaunchKernelGGL(loss_main_kernel, dim3(g), dim3(256), 0, st, a);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, st, a);
  int rc = cl4::launch_status();
  if (rc || !with_grad) return rc;
  rc = cl4_resize_bilinear_bwd(a.g_center, d->d_center, d->b, ctot, d->hc, d->wc, d->H, d->W, 0, ctot, 0, ctot, 0, 0, stream);
  if (rc) return rc;
  return cl4_resize_bilinear_bwd(a.g_offset, d->d_offset, d->b, 2, d->hc, d->wc, d->H, d->W, 0, 2, 0, 2, 0, 0, stream);
}

extern "C" int cl4_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, void* stream) {
  CL4_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), p, g, m, v, n, beta1,
                     beta2, eps, weight_decay, step_size, bc2_sqrt);
  return cl4::launch_status();
}

// Graph-friendly Adam: learning rate and step count live in device memory, bias corrections are computed on the device,
// and the step counter is advanced by the launch itself — nothing in the kernel arguments changes between replays.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long n, const float* __restrict__ lr_dev, float beta1, float beta2, float eps, float wd,
                                const int* __restrict__ step_dev) {
  const int step = *step_dev + 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)*lr_dev / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}
__global__ void int_add_kernel(int* c, int inc) { *c += inc; }

extern "C" int cl4_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float beta1,
                                 float beta2, float eps, float weight_decay, int* step_dev, int advance, void* stream) {
  CL4_REQUIRE(p && g && m && v && lr_dev && step_dev && n > 0, "adam_step_dev: bad arguments");
  hipStream_t st = cl4::as_stream(stream);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, st, p, g, m, v, n, lr_dev, beta1, beta2, eps,
                     weight_decay, (const int*)step_dev);
  if (advance) hipLaunchKernelGGL(int_add_kernel, dim3(1), dim3(1), 0, st, step_dev, advance);
  return cl4::launch_status();
}

// torch.optim.SGD(momentum, nesterov, dampening 0) over a flat segment — train.py:180 (phase 1 / step 0 with --optim sgd):
//   g += wd * p;  buf = momentum * buf + g  (buf starts at 0, which equals torch's "buf = g" first step);
//   p -= lr * (nesterov ? g + momentum * buf : buf).  lr_dev != NULL reads the learning rate from device memory.
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr,
                           const float* __restrict__ lr_dev, float momentum, float wd, int nesterov) {
  if (lr_dev) lr = *lr_dev;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float step = gi;
    if (momentum != 0.f) {
      const float b = buf[i] * momentum + gi;
      buf[i] = b;
      step = nesterov ? gi + momentum * b : b;
    }
    p[i] = pi - lr * step;
  }
}

extern "C" int cl4_sgd_step(float* p, const float* g, float* buf, long n, float lr, const float* lr_dev, float momentum,
                            float weight_decay, int nesterov, void* stream) {
  CL4_REQUIRE(p && g && n > 0 && (buf || momentum == 0.f), "sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(cl4::stream_grid(n, 256)), dim3(256), 0, cl4::as_stream(stream), p, g, buf, n, lr, lr_dev,
                     momentum, weight_decay, nesterov);
  return cl4::launch_status();
}

extern "C" int cl4_fill_zero(void* p, size_t bytes, void* stream) {
  CL4_REQUIRE(p && bytes > 0, "fill_zero: bad arguments");
  hipError_t e = cl4::fill_bytes_async(p, 0, bytes, cl4::as_stream(stream));
  return e == hipSuccess ? CL4_OK : (int)e;
}
