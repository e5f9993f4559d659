// Unified-wave bf16x3 forward convolution ("U" kernel): the same arithmetic, packed-weight format, K order, LDS image, tile
// mapping, split-K slabs and fused epilogue as conv_igemm_bf16x3<128, 128, KS, SPLIT, true, 2> in conv_igemm.hip — but without
// the loader / MFMA role split. tools/probes/probe_phases.py (profiles/r02c_phase_probe_*.txt) showed where that kernel's time goes:
// its four loader waves never wait for memory; they are bound by their own instruction stream — vector-memory issue (the CU's
// texture-address unit retires ONE wave-load per ~16 cycles whatever its width, so 64 dword gathers + 16 dwordx4 weight loads per
// K-tile cost 1280 cycles against 768 cycles of MFMA work) and the convert / ds_write burst — while the four MFMA waves sit at
// the barrier for half of every K-tile. Here
//   * a workgroup is 4 waves (256 threads, two workgroups per CU, <= 256 VGPRs each), every wave owning a 64x64 quarter of the
//     128x128 tile AND a quarter of the staging work of every K-tile: 4 weight chunks per lane (dwordx4) and the activations of
//     ONE 8-channel k-group — each lane fetches 4 consecutive pixels of 4 channels with four 16-byte buffer loads (a 3x3 tap is a
//     uniform shift of the flat pixel index for stride-1 "same" convolutions; pixels whose tap falls outside the image are zeroed
//     from a per-lane 36-bit validity mask computed once) — 32 wave-loads per K-tile instead of 80;
//   * the loads of K-tile t+3 are issued while tile t is being multiplied (two register sets per operand), the split / ds_write
//     of tile t+1 sits between the two MFMA groups of tile t, and the hardware overlaps the VALU / LDS / VMEM work of one wave with
//     the MFMAs of the other three resident on its SIMD pair;
//   * every address the activation loads form goes through the buffer descriptor's bounds check (channel offset in the VGPR
//     offset, no scalar offset): nothing is ever read outside the tensor.
// Eligible launches (launch_fwd in conv_igemm.hip): bf16x3, 128x128 plan, 1x1 or 3x3, stride 1, pad = dil * (k / 2), cin % 32 == 0.
#include "conv_args.h"

namespace {

using cl4::ConvArgs;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef CL4_U_DBG
#define CL4_U_DBG 0   // experiments only: 1 = no loads in the loop, 2 = no convert / LDS stores, 4 = no MFMA
#endif

constexpr int BM = 128, BN = 128, BKU = 32, P = 2, KG = 4;
constexpr int A_CHUNKS = KG * P * BM;      // 16-byte chunks of the weight tile of one stage
constexpr int B_CHUNKS = KG * P * BN;
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {   // {bf16(lo), bf16(hi)} in one dword (round to nearest even)
  f32x2v t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}

// a - b as ONE v_sub_f32: hipcc's SLP pass pairs plain subtractions into v_pk_add_f32, which issues badly beside MFMAs
__device__ __forceinline__ float sub_f32(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KS, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_u_bf16x3(ConvArgs a) {
  constexpr int TM = 2, TN = 2, WTM = 64, WTN = 64;
  constexpr int KHW = KS * KS;
  __shared__ u32x4 lds[2][A_CHUNKS + B_CHUNKS];            // 64 KB: two K-tile stages; reused as the parked [128][128] fp32 tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // workgroup -> logical tile: XCD-aware order, M fastest (identical to conv_igemm_bf16x3)
  const unsigned total = gridDim.x, id = blockIdx.x;
  unsigned long long* tl = a.timeline ? a.timeline + (size_t)id * 8 : nullptr;
  auto stamp = [&](int slot) {
    if (tl && tid == 0) tl[slot] = __builtin_amdgcn_s_memtime();
  };
  if (tl && tid == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    tl[6] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    tl[7] = __builtin_amdgcn_s_memrealtime();              // 100 MHz: with [5] at the end, the shader clock the workgroup ran at
  }
  stamp(0);
  const unsigned q8 = total >> 3, r8 = total & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const unsigned bz = lid / (unsigned)a.ntile_l;
  const unsigned tile_ = (unsigned)a.tile_base + (lid - bz * (unsigned)a.ntile_l);
  const unsigned by_ = tile_ / (unsigned)a.gx, bx_ = tile_ - by_ * (unsigned)a.gx;
  const int m0 = bx_ * BM, n0 = by_ * BN;
  const int HW = a.hout * a.wout;                          // == hin * win for the eligible launches
  const int kbeg = SPLIT ? bz * a.kchunk : 0;
  const int kend = SPLIT ? min(a.K, kbeg + a.kchunk) : a.K;
  const int nk = (kend - kbeg + BKU - 1) / BKU;
  const int T0 = kbeg >> 5;                                // first K-tile = 32-channel block (x tap)

  // ---- activation loader state: this wave stages k-group `wave` of every K-tile; lane = (4-pixel group g, channel half h) ------------
  const int g = lane >> 1, h = lane & 1;
  const int p0 = n0 + 4 * g;
  const bool lane_ok = p0 < a.npix;                        // hw_pad % 4 == 0: a 4-pixel group never straddles two images
  const int n_ = lane_ok ? p0 / a.hw_pad : 0;
  const int hw0 = lane_ok ? p0 - n_ * a.hw_pad : 0;
  const unsigned lane_base = (unsigned)((((long)n_ * a.in_ctot + a.in_coff + 8 * wave + 4 * h) * HW + hw0) * 4);
  unsigned m_lo = lane_ok ? 0xFFFFFFFFu : 0u, m_hi = lane_ok ? 0xFu : 0u;   // validity of (tap, pixel q): bit tap * 4 + q
  if (KS == 3) {
    m_lo = 0; m_hi = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int hwq = hw0 + q;
      const bool vq = lane_ok && hwq < HW;
      const int oy = hwq / a.wout, ox = hwq - oy * a.wout;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = (tap / 3 - 1) * a.dil, dx = (tap % 3 - 1) * a.dil;
        const unsigned ok = (vq && (unsigned)(oy + dy) < (unsigned)a.hin && (unsigned)(ox + dx) < (unsigned)a.win) ? 1u : 0u;
        if (tap < 8) m_lo |= ok << (tap * 4 + q); else m_hi |= ok << q;
      }
    }
  }
  // A tap with a negative shift can put the START of a lane's 16-byte load up to 12 bytes in front of the tensor while its later
  // pixels are valid (first channel of the first image only: a handful of lanes per launch). The bounds check then zeroes the whole
  // load, so such a load is clamped to offset 0 and its dwords are moved up by sh_tab(tap) = 1..3 places when the tile is converted
  // (`special`: wave-uniform and almost always false — the fix-up is a cold branch outside the scheduled K-tile body).
  unsigned sh_tab = 0;
  if (KS == 3) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int st = (int)(lane_base + (unsigned)((((tap / 3 - 1) * a.win + (tap % 3 - 1)) * a.dil) * 4));
      const unsigned mt = tap < 8 ? (m_lo >> (tap * 4)) & 15u : m_hi;
      if (mt != 0u && st < 0) sh_tab |= (((unsigned)(-st)) >> 2) << (2 * tap);
    }
  }
  const bool special = KS == 3 && __builtin_amdgcn_readfirstlane((int)(__ballot(sh_tab != 0u) != 0ull)) != 0;
  const unsigned x_bytes = (unsigned)((long)a.n * a.in_ctot * HW * 4);       // < 2^31, checked on the host
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  const int KGtot = a.K / 8;
  const unsigned w_bytes = (unsigned)((long)KGtot * P * a.cout * 16);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, w_bytes, 0x00020000);
  const int am = (wave & 1) * 64 + lane;                   // weight chunk row of this lane; plane = wave >> 1, k-group = i
  const unsigned a_voff = (m0 + am) < a.cout ? (unsigned)(m0 + am) * 16u : OOB;

  // per-lane tap mask of K-tile kt (KS == 3), 4 bits
  auto tap_mask = [&](int kt) -> unsigned {
    if (KS == 1) return m_hi;
    const int T = T0 + kt;
    const int tap = T - (T / KHW) * KHW;
    return tap < 8 ? (m_lo >> (tap * 4)) & 15u : m_hi;
  };
  auto loadB = [&](f32x4v (&xr)[4], int kt) {
    const int T = T0 + kt;                                 // wave-uniform
    int ci0, dyw = 0;
    if (KS == 1) { ci0 = T * 32; }
    else {
      const int cb = T / KHW, tap = T - cb * KHW;
      ci0 = cb * 32;
      dyw = ((tap / 3 - 1) * a.win + (tap % 3 - 1)) * a.dil;
    }
    // (bitwise select, not `ok ? off : OOB`: hipcc turns the ternary in front of a buffer load into divergent duplicated loads)
    const unsigned keep = (unsigned)(-(int)(tap_mask(kt) != 0u)) & (unsigned)(-(int)(kt < nk));
    const unsigned vo = lane_base + (unsigned)(dyw * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned co = (unsigned)__builtin_amdgcn_readfirstlane((ci0 + j) * HW * 4);
      unsigned off = vo + co;                              // whole address in the checked offset: out-of-tensor dwords read 0
      if (KS == 3 && j == 0) off = (unsigned)max((int)off, 0);   // see sh_tab (tensors are < 2^31 bytes: negative = in front of the tensor)
      const unsigned voff = (off & keep) | (OOB & ~keep);
      xr[j] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
    }
  };
  auto loadA = [&](u32x4 (&ar)[4], int kt) {
    const bool kok = kt < nk;                              // uniform
    const unsigned kkeep = (unsigned)(-(int)kok);
    const unsigned voff = (a_voff & kkeep) | (OOB & ~kkeep);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kgg = (kbeg >> 3) + kt * 4 + i;
      const int soff = __builtin_amdgcn_readfirstlane((kgg * P + (wave >> 1)) * a.cout * 16);   // (no select: past K the lane offset is out of range)
      ar[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, soff, 0);
    }
  };
  auto storeA = [&](const u32x4 (&ar)[4], int stage) {
    u32x4* As = lds[stage];
#pragma unroll
    for (int i = 0; i < 4; ++i) As[(2 * i + (wave >> 1)) * BM + am] = ar[i];
  };
  // split the 4 channels x 4 pixels of this lane into bf16 hi / lo and store them: pixel q of group g is GEMM column q*32 + g, this
  // lane owns bytes 8h..8h+7 of the 16-byte chunk (channels 4h..4h+3 of the k-group) -> consecutive lanes write consecutive 8 bytes
  auto cvtB = [&](const f32x4v (&xr)[4], int kt, int stage) {
    unsigned long long* B8 = reinterpret_cast<unsigned long long*>(lds[stage] + A_CHUNKS);
    const unsigned mb = KS == 3 ? tap_mask(kt) : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = xr[j][q];
      if (KS == 3) {
        const unsigned keep = (unsigned)(-(int)((mb >> q) & 1u));
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[j]) & keep);
      }
      // one v_cvt_pk_bf16_f32 per pair; the halves of the packed word ARE the fp32 values of hi (<< 16 / & 0xffff0000)
      u32x2 hi, lo;
      hi[0] = pack_bf16(v[0], v[1]); hi[1] = pack_bf16(v[2], v[3]);
      lo[0] = pack_bf16(sub_f32(v[0], __builtin_bit_cast(float, hi[0] << 16)), sub_f32(v[1], __builtin_bit_cast(float, hi[0] & 0xFFFF0000u)));
      lo[1] = pack_bf16(sub_f32(v[2], __builtin_bit_cast(float, hi[1] << 16)), sub_f32(v[3], __builtin_bit_cast(float, hi[1] & 0xFFFF0000u)));
      B8[((wave * P + 0) * BN + q * 32) * 2 + lane] = __builtin_bit_cast(unsigned long long, hi);
      B8[((wave * P + 1) * BN + q * 32) * 2 + lane] = __builtin_bit_cast(unsigned long long, lo);
    }
  };

  auto fix_clamped = [&](f32x4v (&xr)[4], int kt) {         // cold: undo the clamp of loadB (first 32-channel block only)
    const int T = T0 + kt;
    const int cb = T / KHW, tap = T - cb * KHW;
    if (cb != 0) return;
    const unsigned sh = (sh_tab >> (2 * tap)) & 3u;
    const f32x4v L = xr[0];
    xr[0][3] = sh == 0 ? L[3] : (sh == 1 ? L[2] : (sh == 2 ? L[1] : L[0]));
    xr[0][2] = sh == 0 ? L[2] : (sh == 1 ? L[1] : L[0]);
    xr[0][1] = sh == 0 ? L[1] : L[0];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mfma_half = [&](int stage, int s2) {
    const u32x4* As = lds[stage] + wm * WTM + lr;
    const u32x4* Bs = lds[stage] + A_CHUNKS + wn * WTN + lr;
    const int kg = 2 * s2 + lh;
    bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 0) * BM + i * 32]);
      al[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 1) * BM + i * 32]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 0) * BN + j * 32]);
      bl[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 1) * BN + j * 32]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#if CL4_U_DBG & 4
        acc[i][j][0] += (float)al[i][0] * (float)bh[j][0] + (float)ah[i][1] * (float)bl[j][1];
#else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#endif
      }
  };

  // Two register sets per operand: tile t+1 lives in set (t+1) & 1 while tile t+2 is in flight in the other one. Every wave issues
  // exactly 8 loads per K-tile (tiles past the end of K go out of range: zeros, no traffic), so the counted waits are static.
  u32x4 ar[2][4];
  f32x4v xb[2][4];
  loadA(ar[0], 0); loadB(xb[0], 0);
  loadA(ar[1], 1); loadB(xb[1], 1);
  wait_vmcnt<8>();
  if (special) fix_clamped(xb[0], 0);
  storeA(ar[0], 0); cvtB(xb[0], 0, 0);
  loadA(ar[0], 2); loadB(xb[0], 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(1);

  // one K-tile: multiply tile t out of `stage` while tile t+1 (register set a1 / x1) is split into the other stage and tile t+3 is
  // requested. Straight-line code (no branch: the scheduler is free to interleave the two MFMA groups with the staging work); the
  // padding step of an odd K-tile count multiplies a zero tile.
  auto step = [&](u32x4 (&a1)[4], f32x4v (&x1)[4], int t, int stage) {
    wait_vmcnt<8>();                                       // tile t+1 has landed (requested a whole K-tile ago); tile t+2's 8 loads stay in flight
    if (__builtin_expect(special, 0)) fix_clamped(x1, t + 1);
    mfma_half(stage, 0);
#if !(CL4_U_DBG & 2)
    storeA(a1, stage ^ 1);
    cvtB(x1, t + 1, stage ^ 1);
#endif
#if !(CL4_U_DBG & 1)
    loadA(a1, t + 3);
    loadB(x1, t + 3);
#endif
    mfma_half(stage, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    step(ar[1], xb[1], t, 0);
    step(ar[0], xb[0], t + 1, 1);
  }
  stamp(2);

  // ---- wide epilogue: park the accumulators in LDS (both stages are free), then every wave streams 32 channel rows out: lane = pixel,
  // 256 contiguous bytes per wave store, 8 rows of residual loads in flight before any arithmetic (as conv_igemm_bf16x3) --------------
  wait_vmcnt<0>();                                         // the padding loads of tiles nk+1, nk+2 (registers are dead, keep the count exact)
  float* stage = reinterpret_cast<float*>(&lds[0][0]);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        stage[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WTN + j * 32 + lr] = acc[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(3);
  {
    constexpr int RPW = BM / 4, PPL = BN / 64, RB = 8;
    long ybase[PPL], rbase[PPL];
    bool pv[PPL];
    int col[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      const int pt = lane + 64 * q;                        // pixel within the tile
      col[q] = (pt & 3) * 32 + (pt >> 2);
      const int p = n0 + pt;
      const int pn = p / a.hw_pad;
      const int phw = p - pn * a.hw_pad;
      pv[q] = p < a.npix && phw < HW;
      const int n = pv[q] ? pn : 0;
      const int hw = pv[q] ? phw : 0;
      ybase[q] = ((long)n * a.out_ctot + a.out_coff) * HW + hw;
      rbase[q] = (long)n * a.cout * HW + hw;
    }
    if (SPLIT) {                                           // K-split launch: raw partial sums into this split's slab, channel rows
      float* wsb = a.ws + ((long)bz * a.cout) * a.ws_npix - a.ws_p0 + n0;
#pragma unroll 4
      for (int rr = 0; rr < RPW; ++rr) {
        const int co = m0 + wave * RPW + rr;
        if (co >= a.cout) break;                           // wave-uniform
#pragma unroll
        for (int q = 0; q < PPL; ++q)
          if (n0 + lane + 64 * q < a.npix) wsb[(long)co * a.ws_npix + lane + 64 * q] = stage[(wave * RPW + rr) * BN + col[q]];
      }
      return;
    }
#pragma unroll 1
    for (int r0 = 0; r0 < RPW; r0 += RB) {
      const int row0 = wave * RPW + r0;
      float v[RB][PPL], t[RB][PPL];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr)
#pragma unroll
        for (int q = 0; q < PPL; ++q) v[rr][q] = stage[(row0 + rr) * BN + col[q]];
      if (a.pre) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) t[rr][q] = a.pre[rbase[q] + (long)min(m0 + row0 + rr, a.cout - 1) * HW];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) v[rr][q] += t[rr][q];
      }
      if (a.post) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) t[rr][q] = a.post[rbase[q] + (long)min(m0 + row0 + rr, a.cout - 1) * HW];
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int cc = min(m0 + row0 + rr, a.cout - 1);
        const float sc = a.scale ? a.scale[cc] : 1.f;
        const float sh = a.shift ? a.shift[cc] : 0.f;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
          const float u = v[rr][q] * sc + sh;
          v[rr][q] = u > 0.f ? u : u * a.neg1;
        }
      }
      if (a.post) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
          for (int q = 0; q < PPL; ++q) v[rr][q] += t[rr][q];
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int co = m0 + row0 + rr;
        if (co >= a.cout) continue;                        // wave-uniform
#pragma unroll
        for (int q = 0; q < PPL; ++q)
          if (pv[q]) a.y[ybase[q] + (long)co * HW] = v[rr][q] > 0.f ? v[rr][q] : v[rr][q] * a.neg2;
      }
    }
  }
  if (tl) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(4);
    if (tid == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

static long g_u_launches = 0;
extern "C" long cl4_conv_debug_u_launches(void) { return g_u_launches; }

namespace cl4 {

int launch_conv_u(const ConvArgs& a, int ks, bool split, unsigned grid, hipStream_t st) {
  const dim3 g(grid), b(256);
  ++g_u_launches;
  if (ks == 1) {
    if (split) hipLaunchKernelGGL((conv_u_bf16x3<1, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((conv_u_bf16x3<1, false>), g, b, 0, st, a);
  } else if (ks == 3) {
    if (split) hipLaunchKernelGGL((conv_u_bf16x3<3, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((conv_u_bf16x3<3, false>), g, b, 0, st, a);
  } else {
    set_error("conv_u: kernel size %d not built", ks);
    return CL4_EINVAL;
  }
  return launch_status();
}

}  // namespace cl4
