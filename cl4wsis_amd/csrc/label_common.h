// Device helpers shared by the label-stage kernels (label.hip, refine.hip, loss.hip).
#pragma once
#include "common.h"

namespace cl4 {

// PyTorch upsample_bilinear2d(align_corners=False) source coordinates.
struct Bilin {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Bilin bilin_half_pixel(int dst, float scale, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Bilin r;
  r.i0 = (int)src;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// value of the upsampled plane at (by, bx); `plane` is the low-resolution [h][w] map.
__device__ __forceinline__ float bilin_sample(const float* __restrict__ plane, int w, const Bilin& by, const Bilin& bx) {
  const float* r0 = plane + by.i0 * w;
  const float* r1 = plane + by.i1 * w;
  return by.l0 * (bx.l0 * r0[bx.i0] + bx.l1 * r0[bx.i1]) + by.l1 * (bx.l0 * r1[bx.i0] + bx.l1 * r1[bx.i1]);
}

// order-preserving float -> uint32 map (for packed atomicMax on signed floats)
__device__ __forceinline__ unsigned ordered_bits(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}

}  // namespace cl4
