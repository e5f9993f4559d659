// The two-plane ("planes") activation format and the arithmetic of the kernels that multiply it, selected at build time (r05):
//   CL4_F16_PLANES = 1 (default): fp16x3 — a chunk holds fp16 hi / lo of 16 x (x = (h + l) / 16 to 22 significant bits), packed weights hold
//     fp16 hi / lo of 4096 w (cl4_pack_conv_weight_f16x3), products hh + hl + lh on v_mfma_*_f16, accumulators x 2^-16. Same bytes, same
//     MFMA rate as bf16x3, ~10x closer to fp32 per layer (FINDINGS.md finding 47);
//   CL4_F16_PLANES = 0: bf16x3 — bf16 hi / lo of x (16-17 bits), cl4_pack_conv_weight_bf16x3 weights: the r02-r04 format, bit for bit.
// Three-plane (bf16x6) operands and the fp32-input kernels (conv_u, conv_igemm) are bf16 in either build.
#pragma once
#include <hip/hip_runtime.h>

#ifndef CL4_F16_PLANES
#define CL4_F16_PLANES 1
#endif

namespace plfmt {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr bool F16 = CL4_F16_PLANES != 0;
constexpr float WSCALE = F16 ? 4096.f : 1.f;      // packed weights of a two-plane launch
constexpr float XSCALE = F16 ? 16.f : 1.f;        // two-plane activations
constexpr float XINV = 1.f / XSCALE;
constexpr float ACC = 1.f / (WSCALE * XSCALE);    // what the accumulators of a two-plane launch are multiplied by (exact: powers of two)
constexpr float F16_MAX = 65504.f;

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {       // {bf16(lo), bf16(hi)}, round to nearest even
  f32x2 t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {        // {fp16(lo), fp16(hi)}, round to nearest even (callers keep |x| <= 65504)
  f32x2 t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
}
// one v_med3_f32 (fminf(fmaxf()) costs a canonicalising v_max in front of it; a NaN comes out as -65504 either way it is clamped)
__device__ __forceinline__ float sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, -F16_MAX, F16_MAX); }
__device__ __forceinline__ float sub_asm(float a, float b) {            // one v_sub_f32 (keeps hipcc's SLP pass from forming v_pk_add_f32)
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// one pair of values -> the hi and lo dwords of the format (values UNSCALED: the format's scale is applied here)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  if constexpr (F16) {
    a = sat_f16(a * XSCALE); b = sat_f16(b * XSCALE);     // saturating: |x| >= 4094 is stored as +-4094 (the remainder a - hi is then < 1 ulp: no second clamp)
    hi = pk_f16(a, b);
    const f16x2 h = __builtin_bit_cast(f16x2, hi);
    const float ha = (float)h[0], hb = (float)h[1];     // (plain subtractions: finding 48)
    lo = pk_f16(a - ha, b - hb);
  } else {
    hi = pk_bf16(a, b);
    lo = pk_bf16(sub_asm(a, __builtin_bit_cast(float, hi << 16)), sub_asm(b, __builtin_bit_cast(float, hi & 0xFFFF0000u)));
  }
}
// The epilogues work in the format's SCALED domain (r05): with the folded-ABN scale / shift multiplied by XSCALE when they are staged, the
// value that reaches the split already is XSCALE x (leaky-ReLU / ReLU are positively homogeneous), and a planes residual is added as
// hi + lo without the division — the same result bit for bit (powers of two commute with rounding), four multiplications per pair fewer.
__device__ __forceinline__ void split2s(float a, float b, unsigned& hi, unsigned& lo) {      // a, b = XSCALE x the values
  if constexpr (F16) {
    a = sat_f16(a); b = sat_f16(b);
    hi = pk_f16(a, b);
    const f16x2 h = __builtin_bit_cast(f16x2, hi);
    const float ha = (float)h[0], hb = (float)h[1];
    lo = pk_f16(a - ha, b - hb);
  } else split2(a, b, hi, lo);
}
__device__ __forceinline__ void join2s(unsigned hi, unsigned lo, float& a, float& b) {       // XSCALE x the stored values
  if constexpr (F16) {
    const f16x2 h = __builtin_bit_cast(f16x2, hi), l = __builtin_bit_cast(f16x2, lo);
    a = (float)h[0] + (float)l[0];
    b = (float)h[1] + (float)l[1];
  } else {
    a = __builtin_bit_cast(float, hi << 16) + __builtin_bit_cast(float, lo << 16);
    b = __builtin_bit_cast(float, hi & 0xFFFF0000u) + __builtin_bit_cast(float, lo & 0xFFFF0000u);
  }
}
// ... and back: hi + lo, exact in fp32
__device__ __forceinline__ void join2(unsigned hi, unsigned lo, float& a, float& b) {
  if constexpr (F16) {
    const f16x2 h = __builtin_bit_cast(f16x2, hi), l = __builtin_bit_cast(f16x2, lo);
    a = ((float)h[0] + (float)l[0]) * XINV;
    b = ((float)h[1] + (float)l[1]) * XINV;
  } else {
    a = __builtin_bit_cast(float, hi << 16) + __builtin_bit_cast(float, lo << 16);
    b = __builtin_bit_cast(float, hi & 0xFFFF0000u) + __builtin_bit_cast(float, lo & 0xFFFF0000u);
  }
}

// the three products of a two-plane MAC, smallest first: lo * hi + hi * lo + hi * hi (fragments as they come out of LDS: 16 bytes each)
template <typename V>
__device__ __forceinline__ f32x4 mfma16_x3(const V& ah, const V& al, const V& bh, const V& bl, f32x4 c) {
  if constexpr (F16) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, ah), a1 = __builtin_bit_cast(f16x8, al), b0 = __builtin_bit_cast(f16x8, bh), b1 = __builtin_bit_cast(f16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
  } else {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, al), b0 = __builtin_bit_cast(bf16x8, bh), b1 = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c, 0, 0, 0);
  }
  return c;
}
template <typename V>
__device__ __forceinline__ f32x16 mfma32_x3(const V& ah, const V& al, const V& bh, const V& bl, f32x16 c) {
  if constexpr (F16) {
    const f16x8 a0 = __builtin_bit_cast(f16x8, ah), a1 = __builtin_bit_cast(f16x8, al), b0 = __builtin_bit_cast(f16x8, bh), b1 = __builtin_bit_cast(f16x8, bl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
  } else {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, al), b0 = __builtin_bit_cast(bf16x8, bh), b1 = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
  }
  return c;
}

}  // namespace plfmt
