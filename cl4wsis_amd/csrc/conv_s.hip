// Planes-input bf16x3 forward convolution ("S" kernel). Same arithmetic as conv_igemm_bf16x3 / conv_u_bf16x3 (three
// v_mfma bf16 products per fp32-equivalent MAC, fp32 accumulation, fused epilogue), but the activations arrive ALREADY split:
//
//   planes format of a tensor [n][c][h][w] (c % 8 == 0):   [n][c/8][plane: hi, lo][h*w][8 x bf16]
//   hi = bf16(x), lo = bf16(x - hi); one 16-byte chunk = 8 consecutive channels of one pixel of one plane.
//
// tools/probes/probe_clock.py on the unified-wave kernel (profiles/r02c_u_clock.txt) showed what bounds a bf16x3 K-tile once loads are wide and
// every wave does everything: with the MFMAs REMOVED the loop still takes ~2200 cycles per K-tile — its ~95 VALU instructions per
// wave (three per activation element for the split, masks, addresses) plus LDS stores, against 1412 cycles for the MFMAs alone, and the
// shader clock falls from 2.34 to 1.8-1.9 GHz when both run together. A chunk of the planes format is exactly what one lane hands to
// ds_write_b128 / what an MFMA B-fragment row is, for ANY tap of ANY stride / dilation (a tap is a per-lane pixel offset; padding is an
// out-of-range buffer offset): the K loop of this kernel has no conversion, no masking and ~15 address VALU per K-tile. The split is
// paid once per activation by the PRODUCING kernel's epilogue (ys output below, or cl4_to_planes for tensors that other kernels
// write) instead of once per (M-tile x tap) by every consumer: 18 times less often for a 3x3 256 -> 256 layer.
//
// Structure (as conv_u.hip): 256-thread workgroups (4 waves, 64x64 each of a 128x128 tile, two workgroups per CU), 32-deep K-tiles =
// one tap x 32 channels, two LDS stages, two register sets per operand (tile t+3 requested while tile t is multiplied), one barrier per
// K-tile, XCD-aware tile order, K-split slabs, accumulators parked in LDS for a row-streaming fused epilogue that can emit fp32 NCHW
// and / or planes. MF selects the MFMA shape: 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16.
#include "conv_args.h"

#include "planes_epilogue.h"

namespace {

// BNT = pixels per tile: 128, or 144 (MF = 16 only, r03). Why 144: the 33 x 33 maps of res4 / res5 / DeepLabV3 give 273 tiles of 128
// pixels per 32 images and 137 per 16 — 1.07 and 0.54 rounds of the 512 resident workgroups for a 256-channel layer, i.e. a second
// round (or a second workgroup on 18 of the 256 CUs) for 7 % of the work. With 144 pixels there are 242 and 121 pixel tiles: one round
// at two workgroups per CU (n = 32), or one workgroup per CU / a whole-launch K split that fills both slots (n = 16).
// Wave layout for 144: two row halves x two column parts of 80 and 64 pixels (5 and 4 blocks of 16).
// PL = bf16 planes per operand: 2 = bf16x3 (hi, lo; three products), 3 = bf16x6 (hi, mid, lo; six products — the arithmetic of the
// differentiated convolutions, r03: the trainable ASPP of phase 2 reads a three-plane copy of the frozen res5 features). PL = 3 is built
// for 144-pixel tiles and the 16x16x32 MFMA only: 102 KB of LDS (two stages of 1536 + 1728 chunks), one workgroup per CU, 13 loads per
// thread and K-tile, 120 MFMAs per wave and K-tile.
template <int KS, bool SPLIT, int MF, int BNT, int PL>
__global__ __launch_bounds__(256, PL == 3 ? 1 : (BNT == 64 ? 3 : 2)) void conv_s_bf16x3(ConvArgs a) {
  static_assert(MF == 32 || MF == 16, "MFMA shape");
  static_assert(BNT == 128 || ((BNT == 144 || BNT == 64) && MF == 16), "tile width");
  static_assert(PL == 2 || (PL == 3 && MF == 16 && BNT == 144), "planes");
  constexpr int P = PL;
  constexpr int A_CHUNKS = KG * P * BM;
  constexpr int NCH = KG * P / 2;                          // chunk rows (k-group, plane) per staging thread of the weight tile
  constexpr bool WIDE = BNT == 144;
  // 64-pixel tiles (r03): 48 KB of LDS and ~150 registers, three workgroups per CU — for the short-K layers whose time is their
  // epilogue's HBM stream (256 -> 1024 + residual: 8 K-tiles, then 64 KB in and 64 KB out per 128-pixel tile): a third resident
  // workgroup in another phase keeps the memory pipe busy while the other two multiply. Pixel staging: thread = (column tid & 63,
  // quarter tid >> 6) moves 2 chunk rows; epilogue: 4 channel groups per thread.
  constexpr bool NARROW = BNT == 64;
  constexpr int NCHB = NARROW ? KG * P / 4 : NCH;          // chunk rows per staging thread of the pixel tile
  constexpr int WTM = 64;
  constexpr int KHW = KS * KS;
  constexpr int NB = MF == 32 ? 2 : 4;                     // MFMA row blocks per wave
  constexpr int NBJ = MF == 32 ? 2 : (WIDE ? 5 : (NARROW ? 2 : 4));   // MFMA column blocks per wave (the second column part of a wide tile has 4)
  constexpr int ACCV = MF == 32 ? 16 : 4;                  // accumulator registers per block
  constexpr int BCH = KG * P * BNT;                        // B chunks per stage
  constexpr int STG = A_CHUNKS + BCH;                      // chunks per stage
  constexpr int LDSC = (2 * STG * 16 > BM * BNT * 4) ? 2 * STG : BM * BNT / 4;
  constexpr int LPT = NCH + NCHB + (WIDE ? 1 : 0);         // vector-memory loads per thread and K-tile
  __shared__ u32x4 lds[LDSC];                              // two K-tile stages; reused as the parked [128][BNT] fp32 tile (64 / 72 KB)
#if !CL4_S_NO_AFFINE_LDS
  __shared__ float affine[2 * BM];                         // scale / shift of the tile's 128 output channels, fetched once at the start
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int cb0 = WIDE ? wn * 80 : (NARROW ? wn * 32 : wn * 64);   // first pixel column of this wave's part
  const int nbj = (WIDE && wn == 1) ? 4 : NBJ;             // wave-uniform

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
    tl[7] = __builtin_amdgcn_s_memrealtime();
  }
  stamp(0);
  const unsigned q8 = total >> 3, r8 = total & 7, xcd = id & 7;
  const unsigned lid = a.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3) : id;
  const unsigned bz = lid / (unsigned)a.ntile_l;
  const unsigned tile_ = (unsigned)a.tile_base + (lid - bz * (unsigned)a.ntile_l);
  const unsigned by_ = tile_ / (unsigned)a.gx, bx_ = tile_ - by_ * (unsigned)a.gx;
  const int m0 = bx_ * BM, n0 = by_ * BNT;
  const int HWo = a.hout * a.wout, HWi = a.hin * a.win;
  // ---- K-tiles of this workgroup. A K-tile = one tap x 32 channels. Taps whose input row (column) lies outside the image for EVERY
  // pixel of the tile are dropped (r03): a dilated 3x3 layer on a 33 x 33 map multiplies zeros for most of its outer taps — dilation
  // 18: the upper tap row is valid for output rows >= 18 only, so the tiles of rows 0..17 skip a third of their K loop (36 % of the
  // layer's K-tiles over all tiles, 24 % at dilation 12, 12 % at dilation 6). `vlist` = the surviving taps (4 bits each), nv of them;
  // the K loop walks the compacted index u = cb * nv + j (cb = block of 32 channels), and a K split divides THAT range.
  unsigned long long vlist = 0;
  int nv = 1;
  if constexpr (KS == 3) {
    const int pl = min(n0 + BNT, a.npix) - 1;             // last pixel of the tile (n0 < npix by construction of the grid)
    const int im0 = n0 / HWo, im1 = pl / HWo;
    int ymin = (n0 - im0 * HWo) / a.wout, ymax = (pl - im1 * HWo) / a.wout;
    if (im0 != im1) { ymin = 0; ymax = a.hout - 1; }      // the tile crosses an image boundary: every row may occur
    nv = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int lo = ymin * a.stride - a.pad + ky * a.dil, hi = ymax * a.stride - a.pad + ky * a.dil;   // input rows this tap touches
      if (hi < 0 || lo >= a.hin) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xlo = -a.pad + kx * a.dil, xhi = (a.wout - 1) * a.stride - a.pad + kx * a.dil;      // a tile holds whole rows: every column occurs
        if (xhi < 0 || xlo >= a.win) continue;
        vlist |= (unsigned long long)(ky * 3 + kx) << (4 * nv);
        ++nv;
      }
    }
    nv = __builtin_amdgcn_readfirstlane(nv);
    vlist = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(vlist >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)vlist);
  }
  const int nkc = (a.cin >> 5) * nv;                       // compacted K-tiles of this tile
  const int nsplit = SPLIT ? (a.K + a.kchunk - 1) / a.kchunk : 1;
  const int chunk_c = (nkc + nsplit - 1) / nsplit;
  const int u0 = SPLIT ? (int)bz * chunk_c : 0;
  const int nk = max(0, min(nkc - u0, chunk_c));
  // iterator over (channel block, tap) in load order: every K-tile is requested exactly once, in increasing order
  int it_cb = nv > 0 ? u0 / nv : 0, it_j = nv > 0 ? u0 - it_cb * nv : 0;
  auto next_tile = [&](int& cb, int& tap) {
    cb = it_cb;
    tap = KS == 3 ? (int)((vlist >> (4 * it_j)) & 15ull) : 0;
    if (++it_j >= nv) { it_j = 0; ++it_cb; }
  };
#if !CL4_S_NO_AFFINE_LDS
  // The epilogue multiplies every output by its channel's folded-ABN scale / shift. Read from global memory inside the epilogue loop
  // they were 16 dependent vector loads per 8-channel group (hipcc cannot use scalar loads: the pointers may alias the output);
  // one load per thread here, hidden under the whole K loop.
  if (!SPLIT) {
    const int ch = min(m0 + (tid & 127), a.cout - 1);
    const float* src = tid < 128 ? a.scale : a.shift;
    affine[tid] = (src ? src[ch] : (tid < 128 ? 1.f : 0.f)) * plfmt::XSCALE;      // the epilogue works in the format's scaled domain (planes_fmt.h)
  }
#endif

  // ---- staging roles: thread = (GEMM row / column tid & 127, plane tid >> 7); per K-tile it moves the 4 k-group chunks of that
  // plane for its weight row AND for its pixel: 8 x 16 bytes in, 8 x ds_write_b128 out. A wide tile has 16 more pixel columns = 128 more
  // chunks per K-tile: threads 0..127 move one each (column 128 + (tid & 15), chunk row tid >> 4 = 2 * k-group + plane); the other
  // half issues the same load out of range, so that every wave counts 9 loads per K-tile ------------------------------------------------
  const int col = tid & 127, half = wave >> 1;             // this thread moves the chunk rows r = half * NCH + i (r = k-group * P + plane) of weight row `col`; wave-uniform
  const int colB = NARROW ? (tid & 63) : col;              // ... and the chunk rows qB * NCHB + i of pixel column colB (also its epilogue column)
  const int qB = NARROW ? wave : half;
  const int c8tot = a.in_ctot >> 3;
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
  const Pix px = decode(colB);
  const bool pvalid = px.valid;
  const int pn = px.pn, phw = px.phw, iy0 = px.iy0, ix0 = px.ix0;
  // chunk index of (image, first channel group of the slice, plane, pixel 0) — the tap's pixel and the K-tile's channel groups are added per tile
  const unsigned cbase = (unsigned)(((long)pn * c8tot + (a.in_coff >> 3)) * P * HWi);    // plane and k-group are added per chunk row
  const unsigned x_bytes = (unsigned)((long)a.n * a.in_ctot * HWi * 2 * P);   // planes: 2 bytes per element and plane; < 2^31 checked on the host
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, x_bytes, 0x00020000);
  const int KGtot = a.K / 8;
  const unsigned w_bytes = (unsigned)((long)KGtot * P * a.cout * 16);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, w_bytes, 0x00020000);
  const unsigned a_voff = (m0 + col) < a.cout ? (unsigned)(m0 + col) * 16u : OOB;
  const unsigned pkeep = (unsigned)(-(int)pvalid);
  // the 16 extra columns of a wide tile (also this thread's extra epilogue item: column ecol, 8-channel group tid >> 4)
  const int ecol = 128 + (tid & 15), erow = tid >> 4;      // erow: chunk row k-group * P + plane (staging, tid < 64 * P) / channel group (epilogue)
  Pix ex = px;
  unsigned ebase = 0, ekeep = 0;
  if constexpr (WIDE) {
    ex = decode(ecol);
    ebase = (unsigned)((((long)ex.pn * c8tot + (a.in_coff >> 3)) * P + (erow % P)) * HWi);
    ekeep = (unsigned)(-(int)(ex.valid && tid < 64 * P));
  }

  auto tap_off = [&](int tap, int& dy, int& dx) {          // wave-uniform
    if (KS == 1) { dy = 0; dx = 0; }
    else {
      const int ky = tap / KS, kx = tap - ky * KS;
      dy = ky * a.dil; dx = kx * a.dil;
    }
  };
  auto loadB = [&](u32x4 (&br)[NCHB], int kt, int cb, int tap) {
    int dy, dx;
    tap_off(tap, dy, dx);
    const int cg0 = cb * 4;                                // first channel group (of 8) of this K-tile
    const int iy = iy0 + dy, ix = ix0 + dx;                // pad == 0 for every 1x1 layer of the networks; general for safety
    const unsigned keep = pkeep & (unsigned)(-(int)(kt < nk)) & (unsigned)(-(int)((unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win));
    const unsigned vo = (cbase + (unsigned)(iy * a.win + ix)) * 16u;
#pragma unroll
    for (int i = 0; i < NCHB; ++i) {
      const int r = qB * NCHB + i;                         // chunk row: k-group r / P, plane r % P
      const unsigned co = (unsigned)__builtin_amdgcn_readfirstlane(((cg0 + r / P) * P + r % P) * HWi * 16);
      const unsigned voff = ((vo + co) & keep) | (OOB & ~keep);   // whole address in the bounds-checked offset
      br[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0);
    }
  };
  auto loadX = [&](u32x4& bx, int kt, int cb, int tap) {   // wide tiles: this thread's extra pixel chunk (k-group erow >> 1, plane erow & 1)
    int dy, dx;
    tap_off(tap, dy, dx);
    const int iy = ex.iy0 + dy, ix = ex.ix0 + dx;
    const unsigned keep = ekeep & (unsigned)(-(int)(kt < nk)) & (unsigned)(-(int)((unsigned)iy < (unsigned)a.hin && (unsigned)ix < (unsigned)a.win));
    const unsigned vo = (ebase + (unsigned)(iy * a.win + ix)) * 16u + (unsigned)((cb * 4 + erow / P) * P * HWi * 16);
    bx = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (vo & keep) | (OOB & ~keep), 0, 0);
  };
  auto loadA = [&](u32x4 (&ar)[NCH], int kt, int cb, int tap) {
    const unsigned kkeep = (unsigned)(-(int)(kt < nk));
    const unsigned voff = (a_voff & kkeep) | (OOB & ~kkeep);
    const int T = cb * KHW + tap;                          // K-tile index in the packed weight (tap-major within a block of 32 channels)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int r = half * NCH + i;
      const int soff = __builtin_amdgcn_readfirstlane(((T * 4 + r / P) * P + r % P) * a.cout * 16);
      ar[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, soff, 0);
    }
  };
  auto load_tile = [&](u32x4 (&ar)[NCH], u32x4 (&br)[NCHB], u32x4& bxr, int kt) {
    int cb, tap;
    next_tile(cb, tap);
    loadA(ar, kt, cb, tap);
    loadB(br, kt, cb, tap);
    if constexpr (WIDE) loadX(bxr, kt, cb, tap);
  };
  auto storeAB = [&](const u32x4 (&ar)[NCH], const u32x4 (&br)[NCHB], const u32x4& bx, int stage) {
    u32x4* As = lds + stage * STG;
    u32x4* Bs = As + A_CHUNKS;
#pragma unroll
    for (int i = 0; i < NCH; ++i) As[(half * NCH + i) * BM + col] = ar[i];
#pragma unroll
    for (int i = 0; i < NCHB; ++i) Bs[(qB * NCHB + i) * BNT + colB] = br[i];
    if constexpr (WIDE)
      if (tid < 64 * P) Bs[erow * BNT + ecol] = bx;        // wave-uniform: waves 0, 1 (and 2 for three planes)
  };

  typedef float accv_t __attribute__((ext_vector_type(ACCV)));
  accv_t acc[NB][NBJ];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
#pragma unroll
      for (int r = 0; r < ACCV; ++r) acc[i][j][r] = 0.f;

  bf16x8 bfr[P][NBJ];                                      // MF = 16: the B fragments of the current K-tile (read by half 0, reused by half 1)
  // one half of a K-tile's MFMAs (MF = 32: 16 of the 32 k; MF = 16: two of the four row blocks)
  auto mfma_half = [&](int stage, int s2) {
    if constexpr (MF == 32) {
      const int lr = lane & 31, lh = lane >> 5;
      const u32x4* As = lds + stage * STG + wm * WTM + lr;
      const u32x4* Bs = lds + stage * STG + A_CHUNKS + cb0 + lr;
      const int kg = 2 * s2 + lh;
      bf16x8 ah[NB], al[NB], bh[NBJ], bl[NBJ];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 0) * BM + i * 32]);
        al[i] = __builtin_bit_cast(bf16x8, As[(kg * P + 1) * BM + i * 32]);
      }
#pragma unroll
      for (int j = 0; j < NBJ; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 0) * BNT + j * 32]);
        bl[j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + 1) * BNT + j * 32]);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
#if CL4_S_DBG & 4
          acc[i][j][0] += (float)al[i][0] * (float)bh[j][0] + (float)ah[i][1] * (float)bl[j][1];
#else
          acc[i][j] = plfmt::mfma32_x3(ah[i], al[i], bh[j], bl[j], acc[i][j]);
#endif
        }
    } else {
      // 16x16x32: lane l holds row / column l & 15 and k-group l >> 4 — one fragment spans the whole 32-deep K-tile. Half s2 = the row
      // blocks 2*s2, 2*s2+1 against all column blocks of the wave; the B fragments are read by half 0 and stay in registers for half 1
      // (16 fragment reads per K-tile, like the 32x32x16 path; 18 / 16 for the two column parts of a wide tile).
      // Products, smallest first: two planes lo*hi, hi*lo, hi*hi; three planes (hi = 0, mid = 1, lo = 2) lo*hi, hi*lo, mid*mid,
      // mid*hi, hi*mid, hi*hi (the dropped terms are <= 2^-26 relative).
      const int l15 = lane & 15, kg = lane >> 4;
      const u32x4* As = lds + stage * STG + wm * WTM + l15;
      const u32x4* Bs = lds + stage * STG + A_CHUNKS + cb0 + l15;
      bf16x8 afr[P][2];
#pragma unroll
      for (int pl = 0; pl < P; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) afr[pl][i] = __builtin_bit_cast(bf16x8, As[(kg * P + pl) * BM + (2 * s2 + i) * 16]);
      if (s2 == 0) {
#pragma unroll
        for (int pl = 0; pl < P; ++pl)
#pragma unroll
          for (int j = 0; j < NBJ; ++j)
            if (j < nbj) bfr[pl][j] = __builtin_bit_cast(bf16x8, Bs[(kg * P + pl) * BNT + j * 16]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBJ; ++j)
          if (j < nbj) {
            accv_t& c = acc[2 * s2 + i][j];
#if CL4_S_DBG & 4
            c[0] += (float)afr[P - 1][i][0] * (float)bfr[0][j][0] + (float)afr[0][i][1] * (float)bfr[P - 1][j][1];
#else
            if constexpr (P == 2) {
              c = plfmt::mfma16_x3(afr[0][i], afr[1][i], bfr[0][j], bfr[1][j], c);
            } else {
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[2][i], bfr[0][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0][i], bfr[2][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][i], bfr[1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[1][i], bfr[0][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0][i], bfr[1][j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0][i], bfr[0][j], c, 0, 0, 0);
            }
#endif
          }
    }
  };

  // (Measured and dropped: touching the residual tile's lines at the start of the K loop, as conv_igemm_bf16x3 does. On the short-K
  // residual layers — 256 -> 1024 at 33x33 — it moved 6,000 cycles from the epilogue into the prologue and nothing else: those
  // launches stream their residual and output at the HBM rate either way. profiles/r02d_planes_prefetch.txt)
  u32x4 ar[2][NCH], br[2][NCHB], bx[2];
  bx[0] = bx[1] = u32x4{0, 0, 0, 0};
  load_tile(ar[0], br[0], bx[0], 0);
  load_tile(ar[1], br[1], bx[1], 1);
  wait_vmcnt<LPT>();
  storeAB(ar[0], br[0], bx[0], 0);
  load_tile(ar[0], br[0], bx[0], 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(1);

  // one K-tile: multiply tile t out of `stage` while tile t+1 (register set a1 / b1) goes into the other stage and tile t+3 is requested
  auto step = [&](u32x4 (&a1)[NCH], u32x4 (&b1)[NCHB], u32x4& x1, int t, int stage) {
    wait_vmcnt<LPT>();                                     // tile t+1 has landed (requested a whole K-tile ago); tile t+2's loads stay in flight
    mfma_half(stage, 0);
#if !(CL4_S_DBG & 2)
    storeAB(a1, b1, x1, stage ^ 1);
#endif
#if !(CL4_S_DBG & 1)
    load_tile(a1, b1, x1, t + 3);
#endif
    mfma_half(stage, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  for (int t = 0; t < nk; t += 2) {
    step(ar[1], br[1], bx[1], t, 0);
    step(ar[0], br[0], bx[0], t + 1, 1);
  }
  stamp(2);

  // ---- residual operands requested (everything this thread will add, one burst), then the accumulators are parked: stage[row][col] fp32 ----
  wait_vmcnt<0>();
  constexpr int NGE = NARROW ? 4 : 8;                      // channel groups per thread: g0, g0 + gstep, ...
  const int g0 = NARROW ? wave : half, gstep = NARROW ? 4 : 2;
  SEpilogue<2, NGE, BNT> ep2;
  SEpilogue<1, NGE, BNT> ep1;
  SEpilogue<2, 1, BNT> ex2;                                // the extra (column, group) item of a wide tile
  SEpilogue<1, 1, BNT> ex1;
  const int res_kind = SPLIT ? 0 : (a.post_s ? 2 : (a.post ? 1 : 0));     // wave-uniform (kernel argument)
  if (res_kind == 2) {
    ep2.issue(a, m0, g0, gstep, pvalid, pn, phw, HWo);
    if constexpr (WIDE) ex2.issue(a, m0, erow, 0, ex.valid, ex.pn, ex.phw, HWo);
  } else if (res_kind == 1) {
    ep1.issue(a, m0, g0, gstep, pvalid, pn, phw, HWo);
    if constexpr (WIDE) ex1.issue(a, m0, erow, 0, ex.valid, ex.pn, ex.phw, HWo);
  }
  float* stage = reinterpret_cast<float*>(&lds[0]);
  // two-plane launches of an fp16x3 build: the operand scales leave the accumulators here (exact: powers of two; planes_fmt.h)
  const float accm = (P == 2 && plfmt::F16) ? (a.xs_scale ? 1.f / (plfmt::WSCALE * a.xs_scale[0]) : plfmt::ACC) : 1.f;
  if constexpr (MF == 32) {
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BNT + cb0 + j * 32 + lr] = acc[i][j][r] * accm;
  } else {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        if (j < nbj) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage[(wm * WTM + i * 16 + l4 * 4 + r) * BNT + cb0 + j * 16 + l15] = acc[i][j][r] * accm;
        }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(3);

  if (SPLIT) {                                             // raw partial sums into this split's slab, channel rows, lane = pixel
    constexpr int RPW = BM / 4;
    float* wsb = a.ws + ((long)bz * a.cout) * a.ws_npix - a.ws_p0 + n0;
#pragma unroll 4
    for (int rr = 0; rr < RPW; ++rr) {
      const int co = m0 + wave * RPW + rr;
      if (co >= a.cout) break;                             // wave-uniform
#pragma unroll
      for (int q = 0; q < (BNT + 63) / 64; ++q)
        if (lane + 64 * q < BNT && n0 + lane + 64 * q < a.npix) wsb[(long)co * a.ws_npix + lane + 64 * q] = stage[(wave * RPW + rr) * BNT + lane + 64 * q];
    }
    return;
  }
  if (a.stats) cl4::tile_row_stats<BNT, BM / 4>(a, stage, m0, (int)by_, wave, lane);     // r05 (kernel argument: uniform)

  {
#if !CL4_S_NO_AFFINE_LDS
    const float* aff = affine;
#else
    const float* aff = nullptr;
#endif
    if (res_kind == 2) {
      ep2.finish(a, stage, aff, m0, colB, g0, gstep, pvalid, pn, phw, HWo);
      if constexpr (WIDE) ex2.finish(a, stage, aff, m0, ecol, erow, 0, ex.valid, ex.pn, ex.phw, HWo);
    } else if (res_kind == 1) {
      ep1.finish(a, stage, aff, m0, colB, g0, gstep, pvalid, pn, phw, HWo);
      if constexpr (WIDE) ex1.finish(a, stage, aff, m0, ecol, erow, 0, ex.valid, ex.pn, ex.phw, HWo);
    } else {
      SEpilogue<0, NGE, BNT> ep0;
      ep0.finish(a, stage, aff, m0, colB, g0, gstep, pvalid, pn, phw, HWo);
      if constexpr (WIDE) { SEpilogue<0, 1, BNT> ex0; ex0.finish(a, stage, aff, m0, ecol, erow, 0, ex.valid, ex.pn, ex.phw, HWo); }
    }
  }
  if (tl) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(4);
    if (tid == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
  }
}

// fp32 NCHW (channel slice) -> planes. One thread per (image, channel group, pixel): 8 strided dword loads (coalesced along pixels),
// two 16-byte stores.
__global__ void to_planes_kernel(const float* __restrict__ x, u32x4* __restrict__ xs, int n, int ctot, int coff, int c, int hw) {
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g = (int)(r - (unsigned)im * (unsigned)c8);
    const float* src = x + ((long)im * ctot + coff + g * 8) * hw + px;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(long)j * hw];
    u32x4 hi, lo;
    split8(v, hi, lo);
    u32x4* dst = xs + (((long)im * c8 + g) * 2) * hw + px;
    dst[0] = hi;
    dst[hw] = lo;
  }
}

// planes -> fp32 NCHW (channel slice of the destination): hi + lo (exact in fp32)
__global__ void from_planes_kernel(const u32x4* __restrict__ xs, float* __restrict__ y, int n, int c, int hw, int out_ctot, int out_coff) {
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g = (int)(r - (unsigned)im * (unsigned)c8);
    const u32x4* src = xs + (((long)im * c8 + g) * 2) * hw + px;
    float v[8];
    join8(src[0], src[hw], v);
    float* dst = y + ((long)im * out_ctot + out_coff + g * 8) * hw + px;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[(long)j * hw] = v[j];
  }
}

// fp32 NCHW (channel slice) -> THREE planes [n][c/8][hi, mid, lo][hw][8 x bf16]: x = hi + mid + lo to 24 significant bits — the operand
// format of the bf16x6 launches (PL = 3).
__global__ void to_planes3_kernel(const float* __restrict__ x, u32x4* __restrict__ xs, int n, int ctot, int coff, int c, int hw) {
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g = (int)(r - (unsigned)im * (unsigned)c8);
    const float* src = x + ((long)im * ctot + coff + g * 8) * hw + px;
    float v[8], rest[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(long)j * hw];
    u32x4 hi, mid, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hi[q] = pack_bf16(v[2 * q], v[2 * q + 1]);
      rest[2 * q] = sub_f32(v[2 * q], __builtin_bit_cast(float, hi[q] << 16));
      rest[2 * q + 1] = sub_f32(v[2 * q + 1], __builtin_bit_cast(float, hi[q] & 0xFFFF0000u));
      mid[q] = pack_bf16(rest[2 * q], rest[2 * q + 1]);
      lo[q] = pack_bf16(sub_f32(rest[2 * q], __builtin_bit_cast(float, mid[q] << 16)), sub_f32(rest[2 * q + 1], __builtin_bit_cast(float, mid[q] & 0xFFFF0000u)));
    }
    u32x4* dst = xs + (((long)im * c8 + g) * 3) * hw + px;
    dst[0] = hi;
    dst[hw] = mid;
    dst[2 * (long)hw] = lo;
  }
}

// ---- fp16x3 operands (r05, planes_epilogue.h): fp32 NCHW (channel slice) -> two fp16 planes of x * s, same chunk layout as the bf16 planes.
// s is a power of two: `static_scale` (activations of O(1) networks: 2^4), or — dynamic, for gradients whose magnitude is anything —
// 2^(14 - e) with max |x| < 2^e. The maximum is a two-stage reduction without atomics or a cell that would have to be cleared (the launch
// must replay inside a HIP graph unchanged): absmax_kernel leaves one partial maximum per workgroup in slot[4 ..], every workgroup of the
// conversion kernel folds them itself. The kernel writes s to slot[0]: what the consuming convolution divides by.
constexpr int ABSMAX_PARTS = 512;
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, float* __restrict__ partial, int n, int ctot, int coff, int c, long hw) {
  __shared__ float red[4];
  const long per = (long)c * hw, total = (long)n * per;
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long im = i / per, r = i - im * per;
    m = fmaxf(m, fabsf(x[(im * ctot + coff) * hw + r]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void to_planes_f16_kernel(const float* __restrict__ x, u32x4* __restrict__ xs, int n, int ctot, int coff, int c, int hw,
                                                            float static_scale, float* __restrict__ slot, int nparts) {
  __shared__ float red[4];
  float s = static_scale;
  if (nparts > 0) {                                       // dynamic: fold the partial maxima (the same value in every workgroup)
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, slot[4 + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const float am = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    s = 1.f;
    if (am > 0.f && am < 3.0e38f) {
      int e;
      frexpf(am, &e);                                     // am = f * 2^e, f in [0.5, 1)
      s = ldexpf(1.f, min(max(14 - e, -100), 100));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) slot[0] = s;
  const int c8 = c >> 3;
  const long total = (long)n * c8 * hw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: the launchers check total < 2^31; the 64-bit divisions were ~250 of this loop's ~700 VALU instructions)
    const unsigned iu = (unsigned)i, r = iu / (unsigned)hw;
    const int px = (int)(iu - r * (unsigned)hw);
    const int im = (int)(r / (unsigned)c8), g = (int)(r - (unsigned)im * (unsigned)c8);
    const float* src = x + ((long)im * ctot + coff + g * 8) * hw + px;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(long)j * hw];
    u32x4 hi, lo;
    split8_f16(v, s, hi, lo);
    u32x4* dst = xs + (((long)im * c8 + g) * 2) * hw + px;
    dst[0] = hi;
    dst[hw] = lo;
  }
}

// fp16x3 flavour of the two-plane pack (r05): fp16 hi / lo of w * 2^12 (saturating at +-65504: |w| < 16), same layout
__global__ void pack_weight_f16x3_kernel(const float* __restrict__ w, u32x4* __restrict__ wq, int cout, int K, int khw) {
  const int KGtot = (K + 7) / 8;
  const long total = (long)KGtot * cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = i % cout;
    const int kg = i / cout;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kg * 8 + j;                      // tap-major K order (cin % 32 == 0 is required): k = (cb*khw + tap)*32 + cil
      const int t32 = k >> 5, cil = k & 31;
      const int cb = t32 / khw, tap = t32 - cb * khw;
      v[j] = k < K ? w[(long)m * K + (cb * 32 + cil) * khw + tap] : 0.f;
    }
    u32x4 hi, lo;
    split8_f16(v, F16_WSCALE, hi, lo);
    wq[((long)kg * 2 + 0) * cout + m] = hi;
    wq[((long)kg * 2 + 1) * cout + m] = lo;
  }
}

// r05: every fp16x3 weight pack of a training step in ONE launch. The differentiated networks re-derive the packed operand of each trainable
// convolution once per iteration — forward weight and the transposed / flipped input-gradient weight: ~320 launches of 14-20 us each in phase 1
// (5 ms of a 75 ms step, profiles/r05n_*), each too small to fill the chip. `jobs` (device memory, built once per set of layers): one entry per
// pack; a work item = one (8-deep K group, output row) pair of 16-byte chunks, items numbered across jobs (`first` = a job's first item), the
// job of an item found by binary search. transposed != 0: the pack of wt[ci][co][kh][kw] = w[co][ci][k-1-kh][k-1-kw] (cl4_transpose_flip_weight
// + cl4_pack_conv_weight_f16x3 in one pass: rows = cin, K runs over cout). Output bit-identical to the single-layer launches.
struct PackJob {
  const float* w;
  u32x4* wq;
  long first;
  int cout, cin, khw, transposed;
};

__global__ __launch_bounds__(256) void pack_weights_f16x3_batched_kernel(const PackJob* __restrict__ jobs, int njobs, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo_ = 0, hi_ = njobs - 1;
    while (lo_ < hi_) {                                  // last job with first <= i
      const int mid = (lo_ + hi_ + 1) >> 1;
      if (jobs[mid].first <= i) lo_ = mid; else hi_ = mid - 1;
    }
    const PackJob jb = jobs[lo_];
    const long r = i - jb.first;
    const int rows = jb.transposed ? jb.cin : jb.cout, red = jb.transposed ? jb.cout : jb.cin;      // output rows, reduction channels
    const int K = red * jb.khw;
    const int m = (int)(r % rows), kg = (int)(r / rows);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kg * 8 + j;                          // tap-major K order: k = (cb*khw + tap)*32 + cil
      const int t32 = k >> 5, cil = k & 31;
      const int cb = t32 / jb.khw, tap = t32 - cb * jb.khw;
      const int c = cb * 32 + cil;
      v[j] = jb.transposed ? jb.w[((long)c * jb.cin + m) * jb.khw + (jb.khw - 1 - tap)] : jb.w[((long)m * red + c) * jb.khw + tap];
    }
    u32x4 hi, lo;
    split8_f16(v, F16_WSCALE, hi, lo);
    jb.wq[((long)kg * 2 + 0) * rows + m] = hi;
    jb.wq[((long)kg * 2 + 1) * rows + m] = lo;
  }
}

}  // namespace

static long g_s_launches = 0;
extern "C" long cl4_conv_debug_s_launches(void) { return g_s_launches; }

extern "C" int cl4_to_planes(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, void* stream) {
  CL4_REQUIRE(x && xs, "to_planes: null tensor");
  CL4_REQUIRE(n > 0 && c > 0 && hw > 0 && c % 8 == 0 && coff >= 0 && coff + c <= ctot, "to_planes: bad channel slice (c %% 8 == 0 required)");
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(to_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, (u32x4*)xs, n, ctot, coff, c, hw);
  return cl4::launch_status();
}

extern "C" int cl4_to_planes3(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, void* stream) {
  CL4_REQUIRE(x && xs, "to_planes3: null tensor");
  CL4_REQUIRE(n > 0 && c > 0 && hw > 0 && c % 8 == 0 && coff >= 0 && coff + c <= ctot, "to_planes3: bad channel slice (c %% 8 == 0 required)");
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(to_planes3_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), x, (u32x4*)xs, n, ctot, coff, c, hw);
  return cl4::launch_status();
}

extern "C" int cl4_planes_format(void) { return plfmt::F16 ? 1 : 0; }      // 1: fp16x3 two-plane format / arithmetic (planes_fmt.h), 0: bf16x3
extern "C" size_t cl4_to_planes_f16_slot_floats(void) { return 4 + ABSMAX_PARTS; }

extern "C" int cl4_to_planes_f16(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, int dynamic, float* slot, void* stream) {
  CL4_REQUIRE(x && xs && slot, "to_planes_f16: null pointer");
  CL4_REQUIRE(n > 0 && c > 0 && hw > 0 && c % 8 == 0 && coff >= 0 && coff + c <= ctot, "to_planes_f16: bad channel slice (c %% 8 == 0 required)");
  hipStream_t st = cl4::as_stream(stream);
  int nparts = 0;
  if (dynamic) {
    const long elems = (long)n * c * hw;
    nparts = (int)std::min<long>(ABSMAX_PARTS, std::max<long>(1, cl4::cdiv((int)std::min<long>(elems, 1L << 30), 2048)));
    hipLaunchKernelGGL(absmax_kernel, dim3(nparts), dim3(256), 0, st, x, slot + 4, n, ctot, coff, c, (long)hw);
  }
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(to_planes_f16_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, st, x, (u32x4*)xs, n, ctot, coff, c, hw, F16_XSCALE, slot, nparts);
  return cl4::launch_status();
}

extern "C" int cl4_pack_conv_weight_f16x3(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream) {
  CL4_REQUIRE(w && wq && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_conv_weight_f16x3: bad arguments");
  CL4_REQUIRE(cin % 32 == 0, "pack_conv_weight_f16x3: cin %% 32 == 0 (the planes kernels' tap-major K order)");
  const int K = cin * kh * kw;
  const long total = (long)(K / 8) * cout;
  hipLaunchKernelGGL(pack_weight_f16x3_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), w, (u32x4*)wq, cout, K, kh * kw);
  return cl4::launch_status();
}

extern "C" int cl4_pack_conv_weights_f16x3_batched(const cl4_pack_job* jobs, int njobs, long total_items, void* stream) {
  CL4_REQUIRE(jobs && njobs > 0 && total_items > 0, "pack_conv_weights_f16x3_batched: bad arguments");
  static_assert(sizeof(PackJob) == sizeof(cl4_pack_job) && sizeof(PackJob) == 40, "cl4_pack_job layout (include/cl4wsis.h)");
  hipLaunchKernelGGL(pack_weights_f16x3_batched_kernel, dim3(cl4::stream_grid(total_items, 256)), dim3(256), 0, cl4::as_stream(stream), (const PackJob*)jobs, njobs,
                     total_items);
  return cl4::launch_status();
}

extern "C" int cl4_from_planes(const void* xs, float* y, int n, int c, int hw, int out_ctot, int out_coff, void* stream) {
  CL4_REQUIRE(xs && y, "from_planes: null tensor");
  CL4_REQUIRE(n > 0 && c > 0 && hw > 0 && c % 8 == 0 && out_coff >= 0 && out_coff + c <= out_ctot, "from_planes: bad channel slice");
  const long total = (long)n * (c >> 3) * hw;
  CL4_REQUIRE(total < (1L << 31), "planes pass: n * c / 8 * hw = %ld work items exceed the 32-bit index arithmetic of the kernel", total);
  hipLaunchKernelGGL(from_planes_kernel, dim3(cl4::stream_grid(total, 256)), dim3(256), 0, cl4::as_stream(stream), (const u32x4*)xs, y, n, c, hw, out_ctot, out_coff);
  return cl4::launch_status();
}

namespace cl4 {

int launch_conv_s(const ConvArgs& a, int ks, bool split, unsigned grid, hipStream_t st, int bn, int planes) {
  static const int mf = [] { const char* e = getenv("CL4_CONV_S_MFMA"); return e ? atoi(e) : 16; }();   // 16x16x32: 3-6 % faster on the long-K layers (higher sustained clock), equal elsewhere
  const dim3 g(grid), b(256);
  ++g_s_launches;
  if (bn != 128 && bn != 144 && bn != 64) { set_error("conv_s: tile width %d not built", bn); return CL4_EINVAL; }
  if (planes != 2 && !(planes == 3 && bn == 144)) { set_error("conv_s: %d planes at tile width %d not built", planes, bn); return CL4_EINVAL; }
#define CL4_S_GO(KS_, SP_)                                                                                   \
  do {                                                                                                       \
    if (planes == 3) hipLaunchKernelGGL((conv_s_bf16x3<KS_, SP_, 16, 144, 3>), g, b, 0, st, a);               \
    else if (bn == 144) hipLaunchKernelGGL((conv_s_bf16x3<KS_, SP_, 16, 144, 2>), g, b, 0, st, a);            \
    else if (bn == 64) hipLaunchKernelGGL((conv_s_bf16x3<KS_, SP_, 16, 64, 2>), g, b, 0, st, a);              \
    else if (mf == 16) hipLaunchKernelGGL((conv_s_bf16x3<KS_, SP_, 16, 128, 2>), g, b, 0, st, a);             \
    else hipLaunchKernelGGL((conv_s_bf16x3<KS_, SP_, 32, 128, 2>), g, b, 0, st, a);                           \
  } while (0)
  if (ks == 1) { if (split) CL4_S_GO(1, true); else CL4_S_GO(1, false); }
  else if (ks == 3) { if (split) CL4_S_GO(3, true); else CL4_S_GO(3, false); }
  else { set_error("conv_s: kernel size %d not built", ks); return CL4_EINVAL; }
#undef CL4_S_GO
  return launch_status();
}

}  // namespace cl4
