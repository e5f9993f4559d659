// Big-tile planes-input bf16x3 forward convolution ("B" kernel, r03): 256 output channels x 144 pixels per workgroup, operands brought
// into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), one 512-thread workgroup per CU.
//
// Why (profiles/r03_conv_s_attribution.txt): on the long-K layers conv_s's K loop with its operand traffic REMOVED runs at 610-620 TFLOP/s
// (the MFMA pipe is ~100 % busy at the clock the chip holds), and with the MFMAs removed the operand traffic alone takes ~the same time —
// 68 KB per K-tile for the two resident 128 x 144 workgroups of a CU, through 72 `buffer_load_dwordx4` into registers and 72
// `ds_write_b128` (13 cycles each on the register -> LDS path) — and the two only half overlap (sum 1.95, measured 1.39 us per K-tile
// pair). This kernel moves fewer bytes per MFMA and moves them without the register round trip:
//   * 256 x 144 tile: 50 KB of operands per K-tile for the same 120 MFMAs per SIMD (0.74 x), 8 waves = 4 row blocks of 64 channels x 2
//     column parts (64 + 16 and 64 pixels: 5 and 4 MFMA column blocks), the wave-level code of conv_s (16x16x32, lo*hi + hi*lo + hi*hi);
//   * LDS-DMA: a wave-instruction copies 64 x 16 bytes from per-lane source addresses (a weight row / a pixel's tap — out-of-range lanes
//     write zeros: padding and tile overhang need no masking) to 1 KB of consecutive LDS; the LDS image of conv_s ([k-group, plane][row])
//     is exactly that. 50 pieces per K-tile: 4 weight + 2 pixel pieces per wave, and one piece each for waves 0 / 1 (the 16 extra pixel
//     columns, kept in their own [8][16] block). No staging registers, no ds_write;
//   * three LDS stages (3 x 50 KB) and a barrier in the MIDDLE of each K-tile's MFMAs: iteration t multiplies row blocks 0, 1 of tile t
//     (fragments read during the previous iteration), reads the fragments of row blocks 2, 3, waits for tile t+1 (tile t+2 stays in
//     flight), barrier, requests tile t+3 into the stage tile t just left, multiplies row blocks 2, 3 while reading tile t+1's first
//     fragments — the MFMA pipe has queued work on both sides of the barrier, and every tile is requested two K-tiles ahead;
//   * the epilogue is conv_s's (planes_epilogue.h), run by the two 256-thread halves of the workgroup on the two 128-row halves of the
//     parked 256 x 144 fp32 tile (147 KB, over the three stages).
// One workgroup per CU means prologue and epilogue are not hidden by a neighbour: the kernel is for launches whose K loop dominates
// (the tuner decides per shape; plan impl 4). Unsplit launches only. Results are bit-identical to conv_s (same products in the same order).
#include "planes_epilogue.h"

namespace {

#ifndef CL4_B_DBG
#define CL4_B_DBG 0   // experiments only: 1 = no LDS-DMA in the K loop, 4 = no MFMA, 8 = no K loop at all (prologue + epilogue), 16 = no epilogue traffic
#endif

constexpr int BNB = 144;

// one LDS-DMA piece: 64 lanes x 16 bytes -> LDS bytes [lds_addr, lds_addr + 1024); source = descriptor base + voff (per lane,
// bounds-checked: out of range writes zeros) + soff (wave-uniform). hipcc does not count it: waits are written by hand below.
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// BMB = output channels per tile, NW = waves: <256, 8> = 256 channels, 64 rows per wave, 50 KB per stage; <128, 4> = 128 channels,
// 64 rows per wave, 34 KB per stage; <128, 8> = 128 channels, 32 rows per wave (two waves per SIMD again: one wave's LDS-DMA issue and
// fragment reads run under the other's MFMAs) — the launches with fewer than 256 tiles of 256 rows (16 images of a 33 x 33 map at 256
// output channels), where conv_s has one workgroup per CU and nothing to hide its operand traffic behind.
// PB = bf16 planes per operand: 2 (bf16x3: three products) or 3 (bf16x6: hi, mid, lo planes and six products — the arithmetic of the
// differentiated convolutions; <128, 8> only: 51 KB per stage, 120 MFMAs per SIMD and K-tile).
// GEMM = true: the same kernel under another name for cl4_gemm_kplanes (wgrad_gemm.hip: the weight gradient as a K-major GEMM), so that
// kernel traces keep the forward-convolution family and the weight-gradient launches apart.
// NSTG = LDS stages (K-tiles in flight + 1): 3; 4 where they fit (the 128-row tile with two planes: 136 KB) — a tile is then requested
// three K-tiles ahead, which is what the K-major GEMM of the weight gradient needs (both its operands miss L2 on every K-tile).
// F16 (r05): the two planes hold fp16 hi / lo (planes_epilogue.h, fp16x3) — the same K loop with v_mfma_f32_16x16x32_f16, and the
// accumulators scaled by 1 / (2^12 * *a.xs_scale) on their way into the parked tile. fp32 output (no planes residual / output yet).
// SPLITK (r05, <128, 8> two planes): the grid is ntile_l tiles x nsplit K ranges (as conv_s's SPLIT); a workgroup multiplies its range of the
// tile's compacted K-tiles and leaves the raw fp32 partial tile in its slab of a.ws ([split][cout][ws_npix]); conv_splitk_epilogue_s sums the
// slabs and runs the fused epilogue. For launches that cannot fill the chip with whole-K tiles (2-4 images of a 33 x 33 map = 62 tiles of
// 72 K-tiles): conv_s's split kernel needs ~2 us per K-tile there (one workgroup per CU, nothing to hide its operand loads behind), this
// kernel's three LDS-DMA stages ~0.5.
template <int KS, int BMB, int NW, int PB, bool GEMM = false, int NSTG = 3, bool SPLITK = false>
__global__ __launch_bounds__(64 * NW, NSTG == 2 ? 4 : (NW == 8 ? 2 : 1)) void conv_b_bf16x3(ConvArgs a) {
  // NSTG == 2 (r05 experiment, <128, 8> only): 70 KB of stages / 74 KB of parked tile -> TWO workgroups per CU, so that one's epilogue
  // (HBM streaming) can run under the other's K loop (MFMA + L2 -> LDS)
  constexpr int KHW = KS * KS;
  constexpr int R = KG * PB;                               // chunk rows (k-group, plane) of an operand tile: 8 or 12
  constexpr int B_CH = R * 128;                            // [k-group * PB + plane][128 pixel columns]
  constexpr int E_CH = R * 16;                             // [k-group * PB + plane][16 extra pixel columns]
  constexpr int NE = R / 4;                                // extra pieces per K-tile (4 chunk rows x 16 columns each): waves 0 .. NE-1
  static_assert(PB == 2 || (PB == 3 && BMB == 128 && NW == 8), "planes");
  constexpr bool F16 = plfmt::F16 && PB == 2 && !GEMM;     // the two-plane format of this build (planes_fmt.h); the k-planes GEMM and three planes are bf16
  constexpr int WR = 2 * BMB / NW;                         // rows per wave: 64 or 32
  constexpr int NRB = WR / 32;                             // MFMA row blocks per half of a K-tile's work: 2 or 1
  constexpr int RB64 = BMB / 64;                           // 64-row groups of the tile (= weight pieces per chunk row)
  constexpr int A_CH = R * BMB;                            // [k-group * PB + plane][BMB rows]
  constexpr int STGB = A_CH + B_CH + E_CH;                 // 3200 chunks = 50 KB (256 rows), 2176 chunks = 34 KB (128 rows)
  constexpr int NAP = R * RB64 / NW;                       // weight pieces per wave and K-tile: 4, or 2 / 3 (<128, 8>)
  constexpr int NBP = 2 * R / NW;                          // pixel pieces per wave and K-tile
  constexpr int NPW = NAP + NBP;                           // pieces per wave and K-tile (one more for waves 0 and 1)
  constexpr int NHALF = BMB / 128;                         // 128-row halves of the tile, each finished by TH threads
  constexpr int TH = 64 * NW / NHALF;                      // 256, or 512 (<128, 8>)
  static_assert((BMB == 256 && NW == 8) || (BMB == 128 && (NW == 4 || NW == 8)), "tile");
  static_assert(NSTG == 2 || NSTG == 3 || NSTG == 4, "stages");
  constexpr int PARKC = BMB * BNB / 4;                     // chunks of the parked [BMB][144] fp32 tile
  __shared__ u32x4 lds[NSTG * STGB > PARKC ? NSTG * STGB : PARKC];   // the K-tile stages; reused as the parked tile
  __shared__ float affine[NHALF][2 * BM];                  // scale / shift of the tile's 128-channel halves

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                 // WR-row block, column part
  const int nbj = wn == 0 ? 5 : 4;                         // wave-uniform

  const unsigned total = gridDim.x, id = blockIdx.x;
  const unsigned q8 = total >> 3, r8 = total & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const unsigned bz = SPLITK ? lid / (unsigned)a.ntile_l : 0u;            // K range of this workgroup
  const unsigned tile_ = (unsigned)a.tile_base + (lid - bz * (unsigned)a.ntile_l);
  unsigned by_ = tile_ / (unsigned)a.gx, bx_ = tile_ - by_ * (unsigned)a.gx;
  if constexpr (GEMM) {
    // xcd_remap == 2 (cl4_gemm_kplanes): both operands are large matrices, so an XCD's workgroups should share rows AND columns:
    // XCD x takes the row tiles of group x >> 1 (of 4) and the column tiles of group x & 1 (of 2) — for 16 x 16 tiles each L2 streams
    // 1/4 of A and 1/2 of B instead of all of A and 1/8 of B (0.93 instead of 1.30 GB through the fabric per launch)
    if (a.xcd_remap == 2) {
      const unsigned rt = (unsigned)a.gx, pt = total / rt, rg = rt >> 2, pg = pt >> 1;
      const unsigned w = id >> 3;
      bx_ = (xcd >> 1) * rg + w % rg;
      by_ = (xcd & 1) * pg + w / rg;
    }
  }
  const int m0 = bx_ * BMB, n0 = by_ * BNB;
  const int HWo = a.hout * a.wout, HWi = a.hin * a.win;

  // ---- surviving taps of this tile (as conv_s.hip: taps that read padding for every pixel of the tile are dropped) ----
  unsigned long long vlist = 0;
  int nv = 1;
  if constexpr (KS == 3) {
    const int pl = min(n0 + BNB, a.npix) - 1;
    const int im0 = n0 / HWo, im1 = pl / HWo;
    int ymin = (n0 - im0 * HWo) / a.wout, ymax = (pl - im1 * HWo) / a.wout;
    if (im0 != im1) { ymin = 0; ymax = a.hout - 1; }
    nv = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int lo = ymin * a.stride - a.pad + ky * a.dil, hi = ymax * a.stride - a.pad + ky * a.dil;
      if (hi < 0 || lo >= a.hin) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xlo = -a.pad + kx * a.dil, xhi = (a.wout - 1) * a.stride - a.pad + kx * a.dil;
        if (xhi < 0 || xlo >= a.win) continue;
        vlist |= (unsigned long long)(ky * 3 + kx) << (4 * nv);
        ++nv;
      }
    }
    nv = __builtin_amdgcn_readfirstlane(nv);
    vlist = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(vlist >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)vlist);
  }
  int u0 = 0;                                              // first compacted K-tile of this workgroup
#if CL4_B_DBG & 8
  const int nk = 0;
#else
  int nk = (a.cin >> 5) * nv;
  if constexpr (SPLITK) {
    const int nsplit = (a.K + a.kchunk - 1) / a.kchunk;
    const int chunk_c = (nk + nsplit - 1) / nsplit;
    u0 = (int)bz * chunk_c;
    nk = max(0, min(nk - u0, chunk_c));
  }
#endif
  if (a.dbg_stagger != 0) {                                // experiments only (shader cycles): > 0: odd CUs of the first round start late;
    bool late = false;                                     // < 0: the second workgroup of every CU (first round) starts late
    if (a.dbg_stagger > 0 && blockIdx.x < 256u) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      late = (hw >> 8) & 1u;
    } else if (a.dbg_stagger < 0) late = blockIdx.x >= 256u && blockIdx.x < 512u;
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)(a.dbg_stagger > 0 ? a.dbg_stagger : -a.dbg_stagger);
      while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
  }

  // folded-ABN scale / shift of the tile's 256 channels -> LDS, before any LDS-DMA is in flight (hipcc counts ITS loads only)
  if (tid < 256 * NHALF) {
    const int h = tid >> 8, t = tid & 255;
    const int ch = min(m0 + h * BM + (t & 127), a.cout - 1);
    const float* src = t < 128 ? a.scale : a.shift;
    affine[h][t] = (src ? src[ch] : (t < 128 ? 1.f : 0.f)) * plfmt::XSCALE;      // the epilogue works in the format's scaled domain (planes_fmt.h)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- this lane's pixel (column tid & 127 of the tile: the column it requests AND the column it finishes in the epilogue) ----
  struct Pix { bool valid; int pn, phw, iy0, ix0; };
  auto decode = [&](int pcol) {
    Pix q;
    const int p = n0 + pcol;
    q.valid = p < a.npix;
    q.pn = q.valid ? p / HWo : 0;
    q.phw = q.valid ? p - q.pn * HWo : 0;
    const int oy = q.phw / a.wout, ox = q.phw - oy * a.wout;
    q.iy0 = oy * a.stride - a.pad;
    q.ix0 = ox * a.stride - a.pad;
    return q;
  };
  const int col = tid & 127;
  const Pix px = decode(col);
  const int ecol = 128 + (tid & 15);
  const Pix ex = decode(ecol);
  const int c8tot = a.in_ctot >> 3;
  const unsigned cbase = (unsigned)(((long)px.pn * c8tot + (a.in_coff >> 3)) * PB * HWi);     // chunk index of (image, slice's first group, plane 0, pixel 0)
  const unsigned ebase = (unsigned)(((long)ex.pn * c8tot + (a.in_coff >> 3)) * PB * HWi);
  const unsigned x_bytes = (unsigned)((long)a.n * a.in_ctot * HWi * 2 * PB);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, x_bytes, 0x00020000);
  const unsigned w_bytes = (unsigned)((long)(a.K / 8) * PB * a.cout * 16);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, w_bytes, 0x00020000);

  // ---- LDS-DMA roles (wave-uniform): weight pieces = rows (wave % RB64) * 64 + lane, chunk rows (wave / RB64) * NAP + i, i < NAP;
  //      pixel pieces = column (wave & 1) * 64 + lane, chunk rows (wave >> 1) * NBP + q, q < NBP; extra piece (waves 0, 1) = chunk row
  //      wave * 4 + (lane >> 4), column 128 + (lane & 15) ----
  const unsigned lds0 = (unsigned)(size_t)(&lds[0]);
  const int arow = m0 + (wave % RB64) * 64 + lane;
  const unsigned a_voff = arow < a.cout ? (unsigned)arow * 16u : OOB;
  const unsigned a_lds = (unsigned)(((wave / RB64) * NAP * BMB + (wave % RB64) * 64) * 16);
  const unsigned b_lds = (unsigned)((A_CH + (wave >> 1) * NBP * 128 + (wave & 1) * 64) * 16);
  const unsigned e_lds = (unsigned)((A_CH + B_CH + wave * 64) * 16);
  const int erow_dma = wave * 4 + (lane >> 4);             // waves 0 .. NE-1 only
  const unsigned pkeep = (unsigned)(-(int)px.valid), ekeep = (unsigned)(-(int)ex.valid);

  // A request = the weight part (4 pieces) and the pixel part (NBP pieces, and the extra piece of waves 0 / 1) of one K-tile, each with
  // its own cursor over the K-tiles: the K loop issues the two parts in different halves of an iteration (CL4_B_SPREAD), so that the
  // chip's vector-memory path sees an even stream instead of 50 pieces in one burst per K-tile. `prep_*` forms the addresses (K-tile
  // >= nk: every lane out of range -> zeros); `piece_*` issues one piece, between MFMA groups.
  struct Cursor { int cb, j; };
  Cursor ca{nv > 0 ? u0 / nv : 0, nv > 0 ? u0 % nv : 0}, cbt = ca;
  auto advance = [&](Cursor& c, int& cb, int& tap) {
    cb = c.cb;
    tap = KS == 3 ? (int)((vlist >> (4 * c.j)) & 15ull) : 0;
    if (++c.j >= nv) { c.j = 0; ++c.cb; }
  };
  struct Req { unsigned avo, T8, a_stage, bvo, bso, evoff, b_stage; } rq;
  auto prep_a = [&](int kt, unsigned stage_bytes) {      // weights: packed [K-tile T = cb * KHW + tap][k-group][plane][cout] chunks
    int cb, tap;
    advance(ca, cb, tap);
    const unsigned live = (unsigned)(-(int)(kt < nk));
    rq.a_stage = stage_bytes;
    rq.avo = (a_voff & live) | (OOB & ~live);
    rq.T8 = (unsigned)((cb * KHW + tap) * R + (wave / RB64) * NAP);
  };
  auto prep_b = [&](int kt, unsigned stage_bytes) {
    int cb, tap;
    advance(cbt, cb, tap);
    const unsigned live = (unsigned)(-(int)(kt < nk));
    int dy = 0, dx = 0;
    if (KS == 3) {
      const int ky = tap / 3, kx = tap - ky * 3;
      dy = ky * a.dil; dx = kx * a.dil;
    }
    rq.b_stage = stage_bytes;
    {   // pixels: this lane's tap of its pixel in plane 0 of the block's first k-group (bounds-checked per-lane offset); the piece's
        // chunk row is a wave-uniform offset on top
      const int iy = px.iy0 + dy, ix = px.ix0 + dx;
      const unsigned keep = pkeep & live & (unsigned)(-(int)((unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win));
      const unsigned vo = (cbase + (unsigned)(iy * a.win + ix)) * 16u;
      rq.bvo = (vo & keep) | (OOB & ~keep);
      rq.bso = (unsigned)(cb * R + (wave >> 1) * NBP) * (unsigned)HWi * 16u;
    }
    {   // the extra piece of waves 0, 1
      const int iy = ex.iy0 + dy, ix = ex.ix0 + dx;
      const unsigned keep = ekeep & live & (unsigned)(-(int)((unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win));
      const unsigned vo = (ebase + (unsigned)(iy * a.win + ix)) * 16u + (unsigned)((cb * R + erow_dma) * HWi) * 16u;
      rq.evoff = (vo & keep) | (OOB & ~keep);
    }
  };
  auto piece_a = [&](int i) { dma16(lds0 + rq.a_stage + a_lds + (unsigned)(i * BMB * 16), rq.avo, wrsrc, (rq.T8 + (unsigned)i) * (unsigned)a.cout * 16u); };
  auto piece_b = [&](int q) { dma16(lds0 + rq.b_stage + b_lds + (unsigned)(q * 128 * 16), rq.bvo, xrsrc, rq.bso + (unsigned)(q * HWi) * 16u); };
  auto piece_e = [&]() { if (wave < NE) dma16(lds0 + rq.b_stage + e_lds, rq.evoff, xrsrc, 0); };
  auto request_a = [&](int kt, unsigned stage_bytes) {
    prep_a(kt, stage_bytes);
#pragma unroll
    for (int i = 0; i < NAP; ++i) piece_a(i);
  };
  auto request_b = [&](int kt, unsigned stage_bytes) {
    prep_b(kt, stage_bytes);
#pragma unroll
    for (int q = 0; q < NBP; ++q) piece_b(q);
    piece_e();
  };
  // all but one K-tile's pieces of this wave have landed (NPW pieces per K-tile, one more for waves 0 and 1)
  auto wait_all_but_one_tile = [&]() {                    // ... but NSTG - 2 tiles: tile t+1 has landed, tiles t+2 .. t+NSTG-1 may be in flight
    if (wave < NE) wait_vmcnt<(NSTG - 2) * (NPW + 1)>();
    else wait_vmcnt<(NSTG - 2) * NPW>();
  };

  typedef float accv_t __attribute__((ext_vector_type(4)));
  accv_t acc[2 * NRB][5];
#pragma unroll
  for (int i = 0; i < 2 * NRB; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = accv_t{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, kg = lane >> 4;
  // fragment addresses inside a stage (chunks): A row block i of this wave, plane pl: (kg * 2 + pl) * BMB + wm * WR + i * 16 + l15
  const int a_frag = kg * PB * BMB + wm * WR + l15;
  const int b_frag = A_CH + kg * PB * 128 + wn * 64 + l15; // column block j < 4: + j * 16; plane: + 128
  const int e_frag = A_CH + B_CH + kg * PB * 16 + l15;     // column block 4 (wave part 0): plane: + 16
  auto read_a = [&](bf16x8 (&af)[PB][NRB], const u32x4* st, int i0) {
#pragma unroll
    for (int pl = 0; pl < PB; ++pl)
#pragma unroll
      for (int i = 0; i < NRB; ++i) af[pl][i] = __builtin_bit_cast(bf16x8, st[a_frag + pl * BMB + (i0 + i) * 16]);
  };
  auto read_b = [&](bf16x8 (&bf)[PB], const u32x4* st, int j) {
#pragma unroll
    for (int pl = 0; pl < PB; ++pl)
      bf[pl] = __builtin_bit_cast(bf16x8, j < 4 ? st[b_frag + pl * 128 + j * 16] : st[e_frag + pl * 16]);
  };
  // products, smallest first (as conv_s.hip): two planes lo*hi, hi*lo, hi*hi; three planes (hi = 0, mid = 1, lo = 2) lo*hi, hi*lo,
  // mid*mid, mid*hi, hi*mid, hi*hi
  auto mma = [&](accv_t& c, const bf16x8 (&af)[PB][NRB], int i, const bf16x8 (&bf)[PB]) {
#if CL4_B_DBG & 4
    c[0] += (float)af[PB - 1][i][0] * (float)bf[0][0] + (float)af[0][i][1] * (float)bf[PB - 1][1];
#else
    if constexpr (F16) {
      c = plfmt::mfma16_x3(af[0][i], af[1][i], bf[0], bf[1], c);
    } else if constexpr (PB == 2) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bf[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[0], c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][i], bf[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bf[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bf[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bf[0], c, 0, 0, 0);
    }
#endif
  };

  // ---- prologue: tiles 0 .. NSTG-2 and the weight part of tile NSTG-1 requested; tile 0 awaited; its first fragments read ----
#pragma unroll
  for (int k = 0; k < NSTG - 1; ++k) {
    request_a(k, (unsigned)(k * STGB * 16));
    request_b(k, (unsigned)(k * STGB * 16));
  }
  request_a(NSTG - 1, (unsigned)((NSTG - 1) * STGB * 16));
  if constexpr (NSTG == 2) {                               // two stages: tile 1 goes out whole (the loop requests tile t+2 in one piece too)
    request_b(1, (unsigned)(STGB * 16));
    if (wave < NE) wait_vmcnt<NPW + 1>();
    else wait_vmcnt<NPW>();
  } else {
    if (wave < NE) wait_vmcnt<(NSTG - 2) * (NPW + 1) + NAP>();
    else wait_vmcnt<(NSTG - 2) * NPW + NAP>();
  }
  __builtin_amdgcn_s_barrier();
  bf16x8 a01[PB][NRB], a23[PB][NRB], bfr[5][PB];      // fragments of the first / second half of the wave's row blocks
  read_a(a01, lds, 0);
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (j < nbj) read_b(bfr[j], lds, j);

  int ring = 0;                                            // stage of tile t (tile k lives in stage k % NSTG)
  for (int t = 0; t < nk; ++t) {
    const unsigned s_cur = (unsigned)ring * STGB;
    const unsigned s_nxt = (unsigned)(ring + 1 == NSTG ? 0 : ring + 1) * STGB;
    const unsigned s_last = (unsigned)(ring == 0 ? NSTG - 1 : ring - 1) * STGB;     // stage of tile t + NSTG - 1 (= of tile t - 1)
    const u32x4* cur = lds + s_cur;
    const u32x4* nxt = lds + s_nxt;
    // first half: the first row blocks of tile t (fragments already in registers) while the fragments of the second half are read and the
    // pixel part of tile t+NSTG-1 is requested (its stage was left by every wave before the previous barrier; its weight part is on its way)
    read_a(a23, cur, NRB);
#if !(CL4_B_DBG & 1)
    if constexpr (NSTG > 2) prep_b(t + NSTG - 1, s_last * 16);
#endif
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < nbj) {
#pragma unroll
        for (int i = 0; i < NRB; ++i) mma(acc[i][j], a01, i, bfr[j]);
      }
#if !(CL4_B_DBG & 1)
      if constexpr (NSTG == 2) {
        // two stages: nothing to request here (tile t+1 went out during the second half of iteration t-1)
      } else if (j == 0) {
#pragma unroll
        for (int q = 0; q < NBP / 2; ++q) piece_b(q);
      } else if (j == 1) {
#pragma unroll
        for (int q = NBP / 2; q < NBP; ++q) piece_b(q);
      } else if (j == 2) piece_e();
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave no longer reads stage `cur`
    wait_all_but_one_tile();                               // tile t+1 has landed
    __builtin_amdgcn_s_barrier();
#if !(CL4_B_DBG & 1)
    prep_a(t + NSTG, s_cur * 16);                          // the weight part of tile t+NSTG goes into the stage every wave has just left
    if constexpr (NSTG == 2) prep_b(t + 2, s_cur * 16);    // two stages: ... and its pixel part with it
#endif
    // second half: the remaining row blocks, column block by column block; block j's B registers are refilled from tile t+1 as soon as
    // they are free, and the pieces of the request are spread between the blocks (an LDS-DMA issue costs 60+ cycles of this wave's stream)
    read_a(a01, nxt, 0);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j < nbj) {
#pragma unroll
        for (int i = 0; i < NRB; ++i) mma(acc[NRB + i][j], a23, i, bfr[j]);
        read_b(bfr[j], nxt, j);
      }
#if !(CL4_B_DBG & 1)
      if (j < NAP) piece_a(j);
      if constexpr (NSTG == 2) {
        if (j < NBP) piece_b(j);
        if (j == 4) piece_e();
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    ring = ring + 1 == NSTG ? 0 : ring + 1;
  }

  // ---- epilogue: the two 256-thread halves of the workgroup finish the two 128-row halves of the tile ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-filled requests past the last K-tile have landed too
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int hsel = tid / TH, t2 = tid % TH;                 // 128-row half, thread within it
  constexpr int GST = TH / 128;                            // channel groups: g0, g0 + GST, ... (16 per half)
  constexpr int NGE = 16 / GST;
  const int g0 = t2 >> 7;
  const int erow = (t2 >> 4) & 15;
  const bool xvalid = ex.valid && t2 < 256;                // the 16 x 16 (column, group) extra items of a half: its first 256 threads
  const int m0h = m0 + hsel * BM;
  SEpilogue<2, NGE, BNB> ep2;
  SEpilogue<1, NGE, BNB> ep1;
  SEpilogue<2, 1, BNB> ex2;
  SEpilogue<1, 1, BNB> ex1;
#if CL4_B_DBG & 16
  {   // no epilogue traffic: one conditional store keeps the accumulators alive
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NRB; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 1.2345e33f && a.y) a.y[tid] = sum;
    return;
  }
#endif
  const int res_kind = SPLITK ? 0 : (a.post_s ? 2 : (a.post ? 1 : 0));
  if (res_kind == 2) {
    ep2.issue(a, m0h, g0, GST, px.valid, px.pn, px.phw, HWo);
    ex2.issue(a, m0h, erow, 0, xvalid, ex.pn, ex.phw, HWo);
  } else if (res_kind == 1) {
    ep1.issue(a, m0h, g0, GST, px.valid, px.pn, px.phw, HWo);
    ex1.issue(a, m0h, erow, 0, xvalid, ex.pn, ex.phw, HWo);
  }
  float* park = reinterpret_cast<float*>(&lds[0]);
  {
    const int l4 = lane >> 4;
    if constexpr (F16) {                                   // undo the operand scales: exact (powers of two); xs_scale: a dynamically scaled operand
      const float accm = a.xs_scale ? 1.f / (plfmt::WSCALE * a.xs_scale[0]) : plfmt::ACC;
#pragma unroll
      for (int i = 0; i < 2 * NRB; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] *= accm;
    }
#pragma unroll
    for (int i = 0; i < 2 * NRB; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        if (j < nbj) {
          const int cbase_ = j < 4 ? wn * 64 + j * 16 : 128;
#pragma unroll
          for (int r = 0; r < 4; ++r) park[(wm * WR + i * 16 + l4 * 4 + r) * BNB + cbase_ + l15] = acc[i][j][r];
        }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (SPLITK) {                                  // raw partial sums into this K range's slab: channel rows, lane = pixel
    constexpr int RPW = BMB / NW;
    float* wsb = a.ws + ((long)bz * a.cout) * a.ws_npix - a.ws_p0 + n0;
#pragma unroll 4
    for (int rr = 0; rr < RPW; ++rr) {
      const int co = m0 + wave * RPW + rr;
      if (co >= a.cout) break;                             // wave-uniform
#pragma unroll
      for (int q = 0; q < (BNB + 63) / 64; ++q)
        if (lane + 64 * q < BNB && n0 + lane + 64 * q < a.npix) wsb[(long)co * a.ws_npix + lane + 64 * q] = park[(wave * RPW + rr) * BNB + lane + 64 * q];
    }
    return;
  }
  if (a.stats) cl4::tile_row_stats<BNB, BMB / NW>(a, park, m0, (int)by_, wave, lane);     // (kernel argument: uniform)
  const float* stage = park + hsel * BM * BNB;
  const float* aff = affine[hsel];
  if (res_kind == 2) {
    ep2.finish(a, stage, aff, m0h, col, g0, GST, px.valid, px.pn, px.phw, HWo);
    ex2.finish(a, stage, aff, m0h, ecol, erow, 0, xvalid, ex.pn, ex.phw, HWo);
  } else if (res_kind == 1) {
    ep1.finish(a, stage, aff, m0h, col, g0, GST, px.valid, px.pn, px.phw, HWo);
    ex1.finish(a, stage, aff, m0h, ecol, erow, 0, xvalid, ex.pn, ex.phw, HWo);
  } else {
    SEpilogue<0, NGE, BNB> ep0;
    SEpilogue<0, 1, BNB> ex0;
    ep0.finish(a, stage, aff, m0h, col, g0, GST, px.valid, px.pn, px.phw, HWo);
    ex0.finish(a, stage, aff, m0h, ecol, erow, 0, xvalid, ex.pn, ex.phw, HWo);
  }
}

}  // namespace

static long g_b_launches = 0;
extern "C" long cl4_conv_debug_b_launches(void) { return g_b_launches; }

namespace cl4 {

// conv_b takes what conv_s takes (planes input, two planes, 1x1 / 3x3, cin % 32 == 0), unsplit, with at least one full 256-row tile
bool conv_b_eligible(const ConvArgs& a, int ks, int rows) {
  return (ks == 1 || ks == 3) && (rows == 128 || rows == 256) && a.cin % 32 == 0 && a.cout >= rows && a.cout % 8 == 0 && a.xs != nullptr;
}

// waves: 8 (rows 256 or 128) or 4 (rows 128); planes: 2, or 3 (bf16x6; rows 128 / 8 waves only)
static int g_b_stagger = 0;
}  // namespace cl4
extern "C" void cl4_conv_debug_stagger(int refclk_ticks) { cl4::g_b_stagger = refclk_ticks; }
namespace cl4 {

int launch_conv_b(const ConvArgs& a_, int ks, int rows, int waves, int planes, unsigned grid, hipStream_t st, bool f16, bool split) {
  ConvArgs a = a_;
  a.dbg_stagger = g_b_stagger;
  const dim3 g(grid), b(64 * waves);
  if (split) {                                             // K-split slabs: the 128-row / eight-wave tile on two planes only
    if (planes != 2 || rows != 128 || waves != 8 || !a.ws || f16) { set_error("conv_b: K-split launches are built for 128 rows / 8 waves / two planes"); return CL4_EINVAL; }
    ++g_b_launches;
    static const int stagesS = [] { const char* e = getenv("CL4_CONV_B_STAGES"); return e ? atoi(e) : 3; }();
    if (ks == 1 && stagesS == 4) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, false, 4, true>), g, b, 0, st, a);
    else if (ks == 3 && stagesS == 4) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 8, 2, false, 4, true>), g, b, 0, st, a);
    else if (ks == 1) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, false, 3, true>), g, b, 0, st, a);
    else if (ks == 3) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 8, 2, false, 3, true>), g, b, 0, st, a);
    else { set_error("conv_b: kernel size %d not built", ks); return CL4_EINVAL; }
    return launch_status();
  }
  if (f16 && (!plfmt::F16 || !a.xs_scale || planes != 2)) {
    set_error("conv_b: a dynamically scaled fp16x3 launch needs an fp16 build (CL4_F16_PLANES), the operand scale and two planes");
    return CL4_EINVAL;
  }
  static const int stagesN = [] { const char* e = getenv("CL4_CONV_B_STAGES"); return e ? atoi(e) : 3; }();   // experiments: four / two LDS stages on the 128-row / 8-wave tile
  const bool stages4 = stagesN == 4;
  ++g_b_launches;
  if (planes == 3) {
    if (ks == 1 && rows == 128 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 3>), g, b, 0, st, a);
    else if (ks == 3 && rows == 128 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 8, 3>), g, b, 0, st, a);
    else { set_error("conv_b: three planes are built for 128 rows / 8 waves only"); return CL4_EINVAL; }
    return launch_status();
  }
  if (ks == 1 && rows == 256 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<1, 256, 8, 2>), g, b, 0, st, a);
  else if (ks == 3 && rows == 256 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<3, 256, 8, 2>), g, b, 0, st, a);
  else if (ks == 1 && rows == 128 && waves == 4) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 4, 2>), g, b, 0, st, a);
  else if (ks == 3 && rows == 128 && waves == 4) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 4, 2>), g, b, 0, st, a);
  else if (ks == 1 && rows == 128 && waves == 8 && stagesN == 2) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, false, 2>), g, b, 0, st, a);
  else if (ks == 1 && rows == 128 && waves == 8 && stages4) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, false, 4>), g, b, 0, st, a);
  else if (ks == 3 && rows == 128 && waves == 8 && stages4) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 8, 2, false, 4>), g, b, 0, st, a);
  else if (ks == 1 && rows == 128 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2>), g, b, 0, st, a);
  else if (ks == 3 && rows == 128 && waves == 8) hipLaunchKernelGGL((conv_b_bf16x3<3, 128, 8, 2>), g, b, 0, st, a);
  else { set_error("conv_b: kernel size %d / %d rows / %d waves not built", ks, rows, waves); return CL4_EINVAL; }
  return launch_status();
}

}  // namespace cl4

// Y[m][p] = sum_k A[m][k] * B[p][k] (fp32 out [M][N]): A, B in the k-planes format of wgrad_gemm.hip (chunks of 8 consecutive k:
// ((k / 8) * 2 + plane) * rows + row), kpad % 32 == 0. One 128 x 144 tile per workgroup, eight waves, unsplit.
extern "C" int cl4_gemm_kplanes(const void* a_kplanes, const void* b_kplanes, float* y, int M, int N, int kpad, void* stream) {
  CL4_REQUIRE(a_kplanes && b_kplanes && y, "gemm_kplanes: null operand");
  CL4_REQUIRE(M >= 128 && M % 8 == 0 && N > 0 && kpad > 0 && kpad % 32 == 0, "gemm_kplanes: M >= 128, M %% 8 == 0, kpad %% 32 == 0 expected (M %d, N %d, kpad %d)", M, N, kpad);
  CL4_REQUIRE((long)kpad * M * 4 < (1L << 31) && (long)kpad * N * 4 < (1L << 31), "gemm_kplanes: operand exceeds the 2 GiB buffer range");
  cl4::ConvArgs a{};
  a.xs = b_kplanes; a.wp = (const float*)a_kplanes; a.y = y;
  a.n = 1; a.cin = kpad; a.hin = 1; a.win = N; a.cout = M; a.stride = 1; a.pad = 0; a.dil = 1; a.hout = 1; a.wout = N;
  a.in_ctot = kpad; a.in_coff = 0; a.out_ctot = M; a.out_coff = 0;
  a.K = kpad; a.npix = N; a.hw_pad = N; a.xcd_remap = 1; a.neg1 = 1.f; a.neg2 = 1.f;
  a.gx = cl4::cdiv(M, 128); a.ntile_l = a.gx * cl4::cdiv(N, 144); a.tile_base = 0; a.ws_npix = N; a.kchunk = kpad;
  static const int map2d = [] { const char* e = getenv("CL4_GEMM_XCD_2D"); return e ? atoi(e) : 1; }();
  if (map2d && a.gx % 4 == 0 && cl4::cdiv(N, 144) % 2 == 0) a.xcd_remap = 2;      // (then the tile count is a multiple of 8 as well)
  static const int stages = [] { const char* e = getenv("CL4_GEMM_STAGES"); return e ? atoi(e) : 4; }();
  if (stages == 4) hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, true, 4>), dim3((unsigned)a.ntile_l), dim3(512), 0, cl4::as_stream(stream), a);
  else hipLaunchKernelGGL((conv_b_bf16x3<1, 128, 8, 2, true, 3>), dim3((unsigned)a.ntile_l), dim3(512), 0, cl4::as_stream(stream), a);
  return cl4::launch_status();
}

