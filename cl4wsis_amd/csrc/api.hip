// Library-level entry points: ABI version and the per-thread error string.
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"

namespace cl4 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cl4

namespace {
// bytes [0, head) and [head + 16 * body, bytes) one per thread, the 16-byte aligned middle as uint4 stores
__global__ void fill_bytes_kernel(unsigned char* __restrict__ p, unsigned v4, size_t head, size_t body, size_t bytes) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4* mid = reinterpret_cast<uint4*>(p + head);
  const uint4 v = make_uint4(v4, v4, v4, v4);
  for (size_t i = t; i < body; i += stride) mid[i] = v;
  const size_t tail0 = head + body * 16;
  const size_t edge = head + (bytes - tail0);
  for (size_t i = t; i < edge; i += stride) p[i < head ? i : tail0 + (i - head)] = (unsigned char)v4;
}
}  // namespace

namespace cl4 {
hipError_t fill_bytes_async(void* p, int value, size_t bytes, hipStream_t st) {
  static const bool own = [] { const char* e = getenv("CL4_MEMSET_KERNEL"); return !e || atoi(e) != 0; }();
  if (!own) return hipMemsetAsync(p, value, bytes, st);
  if (bytes == 0) return hipSuccess;
  const size_t addr = reinterpret_cast<size_t>(p);
  size_t head = (16 - (addr & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t body = (bytes - head) / 16;
  const unsigned b = (unsigned)value & 0xFFu;
  const unsigned v4 = b | (b << 8) | (b << 16) | (b << 24);
  const size_t work = body > 32 ? body : 32;
  hipLaunchKernelGGL(fill_bytes_kernel, dim3(stream_grid((long)work, 256)), dim3(256), 0, st, (unsigned char*)p, v4, head, body, bytes);
  return hipGetLastError();
}
}  // namespace cl4

namespace {
// one thread writes the 100 MHz wall clock (s_memrealtime): a time stamp INSIDE a stream / a captured graph
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
}  // namespace

// Measurement aid (tools/chain_gantt.py): records the device's constant-rate clock (10 ns ticks) into `slot` when
// the stream reaches this point. It is a kernel node like any other, so it can sit at the head and the tail of every chain of a
// captured step and shows which chains really ran side by side — evidence a serialising profiler cannot give.
extern "C" int cl4_stamp(void* slot, void* stream) {
  CL4_REQUIRE(slot, "stamp: null pointer");
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, cl4::as_stream(stream), (unsigned long long*)slot);
  return cl4::launch_status();
}

extern "C" int cl4_abi_version(void) { return 1; }
extern "C" const char* cl4_last_error(void) { return cl4::g_err; }
