// Shared by the planes-input forward kernels (conv_s.hip, conv_b.hip): the bf16 hi / lo split of the planes format and the fused
// epilogue of one thread over the accumulator tile parked in LDS.
#pragma once
#include "conv_args.h"
#include "planes_fmt.h"

namespace {

using cl4::ConvArgs;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef CL4_S_NO_AFFINE_LDS
#define CL4_S_NO_AFFINE_LDS 0   // experiments only: 1 = read scale / shift from global memory inside the epilogue loop (the r02e kernel)
#endif
#ifndef CL4_EPI_NT
#define CL4_EPI_NT 0  // experiments only: 1 = non-temporal loads of a planes residual, 2 = non-temporal stores of the planes output
#endif
#ifndef CL4_S_DBG
#define CL4_S_DBG 0   // experiments only: 1 = no loads in the loop, 2 = no LDS stores, 4 = no MFMA
#endif

constexpr int BM = 128, BKS = 32, KG = 4;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {   // {bf16(lo), bf16(hi)} in one dword (round to nearest even)
  f32x2v t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ float sub_f32(float a, float b) {          // one v_sub_f32 (keeps hipcc's SLP pass from forming v_pk_add_f32)
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// split 8 fp32 values into the hi / lo chunks of the planes format (planes_fmt.h: fp16 hi / lo of 16 x, or — CL4_F16_PLANES=0 — bf16 hi / lo of x)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    plfmt::split2(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}
// scaled-domain forms (planes_fmt.h): v = XSCALE x the values
__device__ __forceinline__ void split8s(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    plfmt::split2s(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}
__device__ __forceinline__ void join8s(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) plfmt::join2s(hi[i], lo[i], v[2 * i], v[2 * i + 1]);
}
// the bf16 hi / lo split whatever the build's planes format is: the k-planes operands of the weight-gradient GEMM (wgrad_gemm.hip) are bf16
__device__ __forceinline__ void split8_bf16(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = pack_bf16(v[2 * i], v[2 * i + 1]);
    lo[i] = pack_bf16(sub_f32(v[2 * i], __builtin_bit_cast(float, hi[i] << 16)), sub_f32(v[2 * i + 1], __builtin_bit_cast(float, hi[i] & 0xFFFF0000u)));
  }
}
__device__ __forceinline__ void join8(const u32x4& hi, const u32x4& lo, float (&v)[8]) {     // hi + lo, exact in fp32
#pragma unroll
  for (int i = 0; i < 4; ++i) plfmt::join2(hi[i], lo[i], v[2 * i], v[2 * i + 1]);
}

// ---- fp16x3 (r05): x = h + l with h = fp16(x * s), l = fp16(x * s - h): 22 significant bits for three MFMA products (hh + hl + lh), the
// accuracy of bf16x6's six (tools/probes/f16x3_probe.hip: 5.3e-7 against 7.8e-7 relative L2 on a K = 2304 dot product; bf16x3 4.6e-6) — provided
// the operands sit inside fp16's range, hence the power-of-two scales: 2^12 on weights (|w| < 16), 2^4 on activations of O(1) networks,
// and 2^(14 - exponent of max |x|) on gradients (cl4_to_planes_f16, dynamic). fp16 subnormals are honoured by the MFMA (same probe).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_WSCALE = 4096.f, F16_XSCALE = 16.f, F16_MAX = 65504.f;
__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {    // {fp16(lo), fp16(hi)} in one dword (round to nearest even; callers keep |x| <= 65504)
  f32x2v t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
}
__device__ __forceinline__ void split8_f16(const float (&v)[8], float s, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = plfmt::sat_f16(v[2 * i] * s), b = plfmt::sat_f16(v[2 * i + 1] * s);      // saturating
    const unsigned hp = pack_f16(a, b);
    const f16x2 h = __builtin_bit_cast(f16x2, hp);
    const float ha = (float)h[0], hb = (float)h[1];
    hi[i] = hp;
    lo[i] = pack_f16(a - ha, b - hb);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- fused epilogue of one thread, r03: NG 8-channel groups g0, g0 + gstep, ... of the tile's 16, for ONE pixel column of the tile.
// y = act2( act1( (acc + pre) * scale + shift ) + post ); written as fp32 NCHW rows (a.y) and / or as planes chunks (a.ys).
// Two halves, called on either side of the accumulator parking: `issue` requests EVERY residual operand of the thread (16 x 16 bytes
// for a planes residual, 64 dwords for an fp32 one) in one burst — one memory round trip per tile instead of one per group, and it
// runs under the parking and the barrier; `finish` is straight-line code (no loop-carried wait): LDS reads, arithmetic and the 16 / 64
// stores of the thread stream out with counted waits on the loads only. The r02 loop (one group per iteration: load, wait, store)
// made every iteration's load wait behind the previous iteration's stores (vmcnt is in order): res4 conv3 at n = 32 112.8 -> 103.5 us,
// with fp32 output 125 -> 108 us (profiles/r03_conv_s_epilogue.txt).
// RES: 0 no residual, 1 fp32 (a.post), 2 planes (a.post_s). Separate instantiations on purpose: a merged body (r02e) came out wrong.
template <int RES, int NG, int BNT>
struct SEpilogue {
  u32x4 rh[NG], rl[NG];      // RES == 2
  float rf[NG][8];           // RES == 1
  __device__ __forceinline__ void issue(const ConvArgs& a, int m0, int g0, int gstep, bool pvalid, int pn, int phw, int HWo) {
    if constexpr (RES == 2) {
#pragma unroll
      for (int it = 0; it < NG; ++it) {
        const int co0 = m0 + (g0 + gstep * it) * 8;
        if (pvalid && co0 < a.cout) {
          const u32x4* ps = reinterpret_cast<const u32x4*>(a.post_s) + (((long)pn * (a.cout >> 3) + (co0 >> 3)) * 2) * HWo + phw;
#if CL4_EPI_NT & 1
          rh[it] = __builtin_nontemporal_load(ps);
          rl[it] = __builtin_nontemporal_load(ps + HWo);
#else
          rh[it] = ps[0];
          rl[it] = ps[HWo];
#endif
        }
      }
    } else if constexpr (RES == 1) {
      const long rbase = (long)pn * a.cout * HWo + phw;
#pragma unroll
      for (int it = 0; it < NG; ++it) {
        const int co0 = m0 + (g0 + gstep * it) * 8;
        if (pvalid && co0 < a.cout) {
#pragma unroll
          for (int r = 0; r < 8; ++r) rf[it][r] = a.post[rbase + (long)min(co0 + r, a.cout - 1) * HWo];
        }
      }
    }
  }
  __device__ __forceinline__ void finish(const ConvArgs& a, const float* stage, const float* affine, int m0, int col, int g0, int gstep,
                                         bool pvalid, int pn, int phw, int HWo) {
    const long ybase = ((long)pn * a.out_ctot + a.out_coff) * HWo + phw;
    const long rbase = (long)pn * a.cout * HWo + phw;
    const int oc8tot = a.out_ctot >> 3;
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int g8 = g0 + gstep * it;
      const int co0 = m0 + g8 * 8;
      if (co0 < a.cout && pvalid) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = stage[(g8 * 8 + r) * BNT + col];
        if (a.pre) {
          float t[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) t[r] = a.pre[rbase + (long)min(co0 + r, a.cout - 1) * HWo];
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += t[r];
        }
        // from here on in the planes format's SCALED domain (planes_fmt.h): `affine` holds XSCALE x (scale, shift), both activations are
        // positively homogeneous, a planes residual comes in as hi + lo, and the value that reaches the split already is XSCALE x the output
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#if !CL4_S_NO_AFFINE_LDS
          const float sc = affine[g8 * 8 + r], sh = affine[BM + g8 * 8 + r];
#else
          const int cc = min(co0 + r, a.cout - 1);
          const float sc = (a.scale ? a.scale[cc] : 1.f) * plfmt::XSCALE;
          const float sh = (a.shift ? a.shift[cc] : 0.f) * plfmt::XSCALE;
#endif
          const float u = v[r] * sc + sh;
          v[r] = u > 0.f ? u : u * a.neg1;
        }
        if constexpr (RES == 2) {
          float tq[8];
          join8s(rh[it], rl[it], tq);
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += tq[r];
        } else if constexpr (RES == 1) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += rf[it][r] * plfmt::XSCALE;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * a.neg2;
        if (a.y) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (co0 + r < a.cout) a.y[ybase + (long)(co0 + r) * HWo] = v[r] * plfmt::XINV;
        }
        if (a.ys) {                                        // cout % 8 == 0 whenever ys is requested (checked on the host)
          u32x4 hi, lo;
          split8s(v, hi, lo);
          u32x4* ys = reinterpret_cast<u32x4*>(a.ys) + (((long)pn * oc8tot + ((a.out_coff + co0) >> 3)) * 2) * HWo + phw;
#if CL4_EPI_NT & 2
          __builtin_nontemporal_store(hi, ys);
          __builtin_nontemporal_store(lo, ys + HWo);
#else
          ys[0] = hi;
          ys[HWo] = lo;
#endif
        }
      }
    }
  }
};

}  // namespace
