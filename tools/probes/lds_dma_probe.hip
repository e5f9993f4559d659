// Probe (r03): what `buffer_load_dwordx4 ... offen lds` does on gfx950 with (a) an out-of-range offset, (b) an LDS destination beyond
// 64 KB, (c) lanes masked off by EXEC. Build: hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__global__ __launch_bounds__(256) void k(const u32x4* x, u32x4* y, unsigned bytes, int mode) {
  __shared__ u32x4 lds[9 * 1024];                        // 144 KB
  for (int i = threadIdx.x; i < 9 * 1024; i += 256) lds[i] = u32x4{0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, bytes, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned first = (unsigned)(size_t)(&lds[0]);
  const unsigned base = first + (mode == 1 ? 8 * 1024 * 16 : 0) + wave * 1024;      // mode 1: destination at 128 KB + ...
  unsigned voff = threadIdx.x * 16u;
  if (mode == 2 && (lane & 1)) voff = 0x80000000u;       // odd lanes out of range
  if (mode == 3) {
    if (lane < 16) dma16(__builtin_amdgcn_readfirstlane(base), voff, r, 0);          // EXEC = lanes 0..15
  } else
    dma16(__builtin_amdgcn_readfirstlane(base), voff, r, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  y[threadIdx.x] = lds[(mode == 1 ? 8 * 1024 : 0) + threadIdx.x];
}
int main() {
  std::vector<unsigned> h(256 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i + 1;
  u32x4 *x, *y;
  hipMalloc(&x, 4096); hipMalloc(&y, 4096);
  hipMemcpy(x, h.data(), 4096, hipMemcpyHostToDevice);
  int bad = 0;
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, 4096u, mode);
    std::vector<unsigned> o(256 * 4);
    hipMemcpy(o.data(), y, 4096, hipMemcpyDeviceToHost);
    int ok = 0, zero = 0, untouched = 0, other = 0;
    for (int t = 0; t < 256; ++t) {
      const unsigned v = o[t * 4], e = (unsigned)t * 4 + 1;
      if (v == e && o[t * 4 + 3] == e + 3) ++ok; else if (v == 0) ++zero; else if (v == 0xAAAAAAAAu) ++untouched; else ++other;
    }
    printf("mode %d: copied %d zero %d untouched %d other %d\n", mode, ok, zero, untouched, other);
    if (mode <= 1 && ok != 256) bad = 1;
    if (mode == 2 && !(ok == 128 && zero == 128)) bad = 1;
    if (mode == 3 && !(ok == 64 && untouched == 192)) bad = 1;
  }
  printf(bad ? "UNEXPECTED\n" : "as expected: OOB lanes write zeros, destinations beyond 64 KB work, masked lanes write nothing\n");
  return bad;
}
