// Weight gradient of a dilated 3x3 convolution as ONE K-major GEMM on the LDS-DMA planes kernel (conv_b.hip), r03.
//
//   dW[co][ci][tap] = sum over input pixels q of  dy[co][q - s(tap)] * x[ci][q]        (s(tap) = the tap's pixel offset; zero outside dy)
//
// The contraction runs over PIXELS, so both operands are repacked with 8 consecutive pixels per 16-byte chunk ("k-planes": chunk
// (q / 8, plane, row) at ((q / 8) * 2 + plane) * R + row — at once the packed-weight format [K-tile][k-group][plane][row] and the planes
// format [c / 8][plane][pixel] of a tensor whose "channels" are the pixels q and whose "pixels" are the rows):
//   A = x   : R = cin rows                     (the frozen res5 features: packed once per step, shared by the ASPP branches)
//   B = dy  : R = 9 * cout rows (tap, co), shifted by the tap and zeroed where the shifted pixel falls outside dy
// and  Y[ci][(tap, co)] = A . B^T  is a 1x1 "convolution" with K = n * hin * win (padded to 32), 2048 output "channels" and 2304 "pixels":
// 16 x 16 tiles of 128 x 144 = one workgroup per CU, K loop of 545 K-tiles at the kernel's full rate, no atomics (the role-split
// conv_wgrad_bf16 kernel it replaces for these layers: 0.68 ms per ASPP branch at 0.29 of the bf16x3 ceiling). A last pass adds
// Y into dW[co][ci][tap]. Same arithmetic as conv_wgrad_bf16<..., P = 2>: hi / lo bf16 split of both operands, lo*hi + hi*lo + hi*hi.
#include "planes_epilogue.h"

namespace {

// k-planes packing of src[n][ctot][hs][ws] (channel slice coff .. coff + c): dst chunk ((q / 8) * 2 + plane) * R + tap * c + ch holds
// src[n][ch][yq - oy(tap)][xq - ox(tap)] for the 8 pixels q = (n, yq, xq) of the q-grid (hq x wq per image), zero outside src and for
// q >= n * hq * wq. grid = (kpad / 64, cdiv(c, 64), taps), 256 threads: a 64 x 64 tile through LDS (reads coalesced along q, writes
// along rows).
__global__ __launch_bounds__(256) void kpack_kernel(const float* __restrict__ src, u32x4* __restrict__ dst, int n, int ctot, int coff, int c,
                                                    int hs, int ws, int hq, int wq, int ks, int dil, int pad, int kq, int kpad, int R) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int oy = (tap / ks) * dil - pad, ox = (tap % ks) * dil - pad;
  const int HWq = hq * wq;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int idx = it * 256 + tid;
    const int cl = idx >> 6, ql = idx & 63;
    const int q = q0 + ql, ch = c0 + cl;
    float v = 0.f;
    if (q < kq && ch < c) {
      const int im = q / HWq, r = q - im * HWq;
      const int yq = r / wq, xq = r - yq * wq;
      const int ys = yq - oy, xs = xq - ox;
      if ((unsigned)ys < (unsigned)hs && (unsigned)xs < (unsigned)ws) v = src[(((long)im * ctot + coff + ch) * hs + ys) * ws + xs];
    }
    tile[cl][ql] = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = it * 256 + tid;
    const int cl = idx & 63, qc = idx >> 6;                // 8 chunks of 8 pixels per tile row
    if (c0 + cl >= c || q0 + qc * 8 >= kpad) continue;     // kpad is a multiple of 32: the last tile of 64 may hold 32 pixels only
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[cl][qc * 8 + j];
    u32x4 hi, lo;
    split8_bf16(v, hi, lo);      // k-planes are bf16 in every build (the GEMM flavour of conv_b multiplies bf16)
    const long chunk = ((long)(q0 / 8 + qc) * 2) * R + (long)tap * c + c0 + cl;
    dst[chunk] = hi;
    dst[chunk + R] = lo;
  }
}

// dw[co][ci][tap] += y[ci][tap * cout + co]. grid = (cin / 16, cout / 16), 256 threads; a 16 x 16 x taps block through LDS so that the
// reads run along co and the writes along (ci, tap).
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_scatter_kernel(const float* __restrict__ y, float* __restrict__ dw, int cin, int cout) {
  __shared__ float t[16][16 * TAPS + 1];
  const int tid = threadIdx.x;
  const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 16;
  const int cil = tid >> 4, col = tid & 15;
  const long R = (long)TAPS * cout;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) t[col][cil * TAPS + tap] = y[(ci0 + cil) * R + (long)tap * cout + co0 + col];
  __syncthreads();
  for (int idx = tid; idx < 16 * 16 * TAPS; idx += 256) {
    const int c = idx / (16 * TAPS), rem = idx - c * (16 * TAPS);
    dw[((long)(co0 + c) * cin + ci0) * TAPS + rem] += t[c][rem];
  }
}

}  // namespace

extern "C" size_t cl4_kpack_bytes(int rows, int kq) { return (size_t)((kq + 31) / 32 * 32) * rows * 4; }

extern "C" int cl4_kpack(const float* src, void* dst, int n, int ctot, int coff, int c, int hs, int ws, int hq, int wq, int ks, int dil, int pad,
                         void* stream) {
  CL4_REQUIRE(src && dst, "kpack: null tensor");
  CL4_REQUIRE(n > 0 && c > 0 && coff >= 0 && coff + c <= ctot && hs > 0 && ws > 0 && hq > 0 && wq > 0 && (ks == 1 || ks == 3), "kpack: bad shape");
  const long kq = (long)n * hq * wq;
  const long kpad = (kq + 31) / 32 * 32;
  const long R = (long)ks * ks * c;
  CL4_REQUIRE(kpad * R * 4 < (1L << 31), "kpack: packed operand of %ld x %ld elements exceeds the 2 GiB buffer range of the GEMM kernel", kpad, R);
  const dim3 g((unsigned)((kpad + 63) / 64), (unsigned)((c + 63) / 64), (unsigned)(ks * ks));
  hipLaunchKernelGGL(kpack_kernel, g, dim3(256), 0, cl4::as_stream(stream), src, (u32x4*)dst, n, ctot, coff, c, hs, ws, hq, wq, ks, dil, pad, (int)kq, (int)kpad, (int)R);
  return cl4::launch_status();
}

extern "C" int cl4_wgrad_scatter(const float* y, float* dw, int cin, int cout, int taps, void* stream) {
  CL4_REQUIRE(y && dw, "wgrad_scatter: null tensor");
  CL4_REQUIRE(cin % 16 == 0 && cout % 16 == 0 && (taps == 9 || taps == 1), "wgrad_scatter: cin, cout multiples of 16 and 1 or 9 taps expected");
  const dim3 g(cin / 16, cout / 16);
  if (taps == 9) hipLaunchKernelGGL((wgrad_scatter_kernel<9>), g, dim3(256), 0, cl4::as_stream(stream), y, dw, cin, cout);
  else hipLaunchKernelGGL((wgrad_scatter_kernel<1>), g, dim3(256), 0, cl4::as_stream(stream), y, dw, cin, cout);
  return cl4::launch_status();
}
