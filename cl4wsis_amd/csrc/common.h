// Shared helpers for the gfx950 kernels of libcl4wsis_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/cl4wsis.h"

namespace cl4 {

void set_error(const char* fmt, ...);

#define CL4_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      cl4::set_error(__VA_ARGS__);            \
      return CL4_EINVAL;                      \
    }                                         \
  } while (0)

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CL4_OK : (int)e;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// memset(p, value, bytes) as a kernel launch of this library (api.hip). Every fill of the step path goes through it instead of
// hipMemsetAsync, so that a captured step consists of kernel nodes only (CL4_MEMSET_KERNEL=0 restores hipMemsetAsync: experiments).
hipError_t fill_bytes_async(void* p, int value, size_t bytes, hipStream_t st);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Grid size for a grid-stride elementwise / streaming kernel: enough workgroups to
// fill 256 CUs x 8 blocks, never more than the work needs.
static inline int stream_grid(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == CL4_ACT_LEAKY) return v > 0.f ? v : v * slope;
  if (act == CL4_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// 64-lane wave reductions (DPP/shuffle based).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace cl4
