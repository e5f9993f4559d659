// Device helpers shared by the label-stage kernels (label.hip, refine.hip, loss.hip).
#pragma once
#include "common.h"

namespace cl4 {

// PyTorch upsample_bilinear2d(align_corners=False) source coordinates.
struct Bilin {
  int i0, i1;
  float l0, l1;
};

// The arithmetic below reproduces PyTorch's CPU kernel bit for bit (verified against F.interpolate on the AVX2/AVX-512
// builds, where the compiler contracts the index and interpolation expressions into FMAs):
//   src = fma(scale, dst + 0.5, -0.5);  row(y) = fma(v[y][x0], lx0, v[y][x1] * lx1);  out = fma(row0, ly0, row1 * ly1)
// Bit-exactness matters because the label stage compares these values with == (max-pool NMS, tie plateaus at
// clamped borders) and thresholds them.
__device__ __forceinline__ Bilin bilin_half_pixel(int dst, float scale, int in_size) {
  float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
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
  const float t0 = fmaf(r0[bx.i0], bx.l0, r0[bx.i1] * bx.l1);
  const float t1 = fmaf(r1[bx.i0], bx.l0, r1[bx.i1] * bx.l1);
  return fmaf(t0, by.l0, t1 * by.l1);
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
