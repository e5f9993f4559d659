// Feasibility probe (r05): split-fp16 operands (x = h + l, fp16 each) with three MFMA products hh + hl + lh against the library's
// bf16x3 (hi / lo bf16, three products) and bf16x6 (three planes, six products), all with fp32 accumulation, on a 16 x 16 x K GEMM
// per wave against float64. Also: do fp16 SUBNORMAL operands survive v_mfma_f32_16x16x32_f16 under hipcc's default mode?
// build: hipcc --offload-arch=gfx950 -O3 -o f16x3_probe f16x3_probe.hip ; run: ./f16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// mode: 0 bf16x3, 1 bf16x6, 2 fp16x3, 3 fp16 hi only (one product)
template <int MODE>
__global__ void gemm16(const float* A, const float* B, float* D, int K, float sa, float sb) {   // A [16][K], B [16][K] (row = n), D [16][16]
  const int lane = threadIdx.x, l15 = lane & 15, kg = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = A[l15 * K + k0 + kg * 8 + i] * sa; b[i] = B[l15 * K + k0 + kg * 8 + i] * sb; }
    if constexpr (MODE <= 1) {
      b8 ah, am, al, bh, bm, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (__bf16)a[i]; float r = a[i] - (float)ah[i]; am[i] = (__bf16)r; al[i] = (__bf16)(r - (float)am[i]);
        bh[i] = (__bf16)b[i]; r = b[i] - (float)bh[i]; bm[i] = (__bf16)r; bl[i] = (__bf16)(r - (float)bm[i]);
      }
      if constexpr (MODE == 0) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
      }
    } else {
      h8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)a[i]; al[i] = (_Float16)(a[i] - (float)ah[i]);
        bh[i] = (_Float16)b[i]; bl[i] = (_Float16)(b[i] - (float)bh[i]);
      }
      if constexpr (MODE == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  const float inv = 1.f / (sa * sb);
  for (int r = 0; r < 4; ++r) D[(kg * 4 + r) * 16 + l15] = acc[r] * inv;
}

// throughput: NW waves per workgroup, each issuing `iters` x 12 MFMAs on register operands
template <bool F16>
__global__ void rate(float* out, int iters) {
  f32x4 acc[4] = {};
  h8 ha = {}, hb = {}; b8 ba = {}, bb = {};
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.001f + i); hb[i] = (_Float16)(1.f + i); ba[i] = (__bf16)(threadIdx.x * 0.001f + i); bb[i] = (__bf16)(1.f + i); }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      if constexpr (F16) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j & 3], 0, 0, 0);
      else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[j & 3], 0, 0, 0);
    }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (s == 1.2345f) out[threadIdx.x] = s;
}

static double rel_err(const std::vector<float>& D, const std::vector<double>& R) {
  double num = 0, den = 0;
  for (size_t i = 0; i < D.size(); ++i) { num += (D[i] - R[i]) * (D[i] - R[i]); den += R[i] * R[i]; }
  return std::sqrt(num / den);
}

int main() {
  const int K = 2304;
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  float *dA, *dB, *dD;
  hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, 16 * K * 4); hipMalloc(&dD, 256 * 4);
  struct Case { const char* name; float wstd, xscale; bool leaky; } cases[] = {
      {"weights N(0,0.03), activations leaky(N(0,1))", 0.03f, 1.f, true},
      {"weights N(0,0.03), activations N(0,1)*1e-3 (small gradients)", 0.03f, 1e-3f, false},
      {"weights N(0,0.03), activations N(0,1)*1e-6 (tiny gradients)", 0.03f, 1e-6f, false},
      {"weights N(0,0.005), activations leaky(N(0,1))*30 (large trunk)", 0.005f, 30.f, true}};
  for (auto& c : cases) {
    std::vector<float> A(16 * K), B(16 * K), D(256);
    for (auto& v : A) v = nd(g) * c.wstd;
    for (auto& v : B) { float t = nd(g); if (c.leaky && t < 0) t *= 0.01f; v = t * c.xscale; }
    std::vector<double> R(256, 0.0);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)B[n * K + k]; R[m * 16 + n] = s; }
    std::vector<float> F(256);   // plain fp32 accumulation in k order (what an exact-fp32 kernel gives)
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[m * K + k], B[n * K + k], s); F[m * 16 + n] = s; }
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("== %s (K = %d)\n   fp32 fma chain on the host      %.3e\n", c.name, K, rel_err(F, R));
    auto run = [&](const char* nm, auto kern, float sa, float sb) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, sa, sb);
      hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
      printf("   %-30s %.3e\n", nm, rel_err(D, R));
    };
    run("bf16x3", gemm16<0>, 1.f, 1.f);
    run("bf16x6", gemm16<1>, 1.f, 1.f);
    run("fp16x1 (hi only)", gemm16<3>, 1.f, 1.f);
    run("fp16x3 unscaled", gemm16<2>, 1.f, 1.f);
    run("fp16x3 w*2^12", gemm16<2>, 4096.f, 1.f);
    run("fp16x3 w*2^12 x*2^4", gemm16<2>, 4096.f, 16.f);
    run("fp16x3 w*2^12 x*2^10", gemm16<2>, 4096.f, 1024.f);
    run("fp16x3 w*2^12 x*2^20", gemm16<2>, 4096.f, 1048576.f);
  }
  {   // subnormal operands: a = 2^-20 (fp16 subnormal), b = 1 -> K * 2^-20 if honoured, 0 if flushed
    std::vector<float> A(16 * 32, 9.5367431640625e-07f), B(16 * 32, 1.f), D(256);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm16<3>, dim3(1), dim3(64), 0, 0, dA, dB, dD, 32, 1.f, 1.f);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    printf("== fp16 subnormal A operand (2^-20) x 1.0 over K = 32: got %.6e, exact %.6e -> %s\n", D[0], 32 * 9.5367431640625e-07, D[0] == 0.f ? "FLUSHED" : "honoured");
  }
  {   // issue rate of the two MFMA forms, 4 waves per CU on every CU
    float* o; hipMalloc(&o, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f16 = 0; f16 < 2; ++f16) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (f16) hipLaunchKernelGGL(rate<true>, dim3(256), dim3(256), 0, 0, o, iters);
        else hipLaunchKernelGGL(rate<false>, dim3(256), dim3(256), 0, 0, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("== %s 16x16x32 MFMA rate: %.0f TFLOP/s (1024 waves x %d x 12 MFMAs in %.3f ms)\n", f16 ? "f16 " : "bf16", 1024.0 * iters * 12 * 16384 / ms / 1e9, iters, ms);
    }
  }
  return 0;
}
