// Fused ResNet stem of the frozen networks: conv 7x7 / stride 2 / pad 3 (3 -> 64) + eval InPlaceABN (scale, shift, leaky-ReLU) +
// max-pool 3x3 / stride 2 / pad 1, one kernel (models/resnet.py:59-63: mod1.conv1 -> mod1.bn1 -> pool).
//
// Why: as two launches the stem wrote the 257x257x64 map (17 MB per image, 811 MB for the 48 images of a step) and the pooling read it
// back — 1.13 ms per step for 60 GFLOP, at the very start of the un-overlapped body forward (profiles/r02j). Fused, a workgroup turns a
// 23 x 71 x 3 input patch (20 KB) into a 4 x 16 x 64 pooled tile (16 KB, plus its planes twin): 7.5 MB of HBM traffic per image.
//
// Arithmetic: the bf16x3 products of the other frozen convolutions (hi/lo split of both operands, lo*hi + hi*lo + hi*hi on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation). K order: k = ((ky * 3 + ci) * 8 + kx), kx = 0..7 with a zero weight at kx = 7 — a
// k-group of 8 is then one run of 8 consecutive input columns of one row of one channel, i.e. the B fragment of a lane is four
// ds_read_b32 from the LDS copy of the patch, no per-element gather. 21 groups + 3 zero groups = 6 K-steps of 32 (192 vs 147 real k).
//
// Structure: persistent 512-thread workgroups (one per CU), tiles = (image, 4 x 16 pooled pixels); the packed weight image (48 KB of
// MFMA A fragments) is loaded into LDS once per workgroup; per tile: patch -> LDS (hi / lo bf16), 1,368 MFMAs over 19 pixel blocks (two 32-channel halves) of
// 16 conv pixels (9 x 33 conv pixels = the pooled tile's receptive field), conv + ABN values to LDS, 3x3/2 max over them, store as
// fp32 NCHW and / or planes. The next tile's patch is requested before the MFMA phase and lands under it.
#include "common.h"
#include "planes_fmt.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int PT_H = 4, PT_W = 16;                 // pooled tile
constexpr int CH = 2 * PT_H + 1, CW = 2 * PT_W + 1;   // conv pixels under it: 9 x 33
constexpr int NPX = CH * CW;                       // 297
constexpr int NBLK = (NPX + 15) / 16;              // 19 blocks of 16 conv pixels
constexpr int PR = 2 * CH + 5, PC = 2 * CW + 6;    // input patch rows / columns kept in LDS: 23 x 72 (column 71 only meets zero weights)
constexpr int PATCH = 3 * PR * PC;                 // 4968 elements per plane
constexpr int KSTEPS = 6, NGROUPS = 21;
constexpr int A_CHUNKS = KSTEPS * 4 * 2 * 64;      // 16-byte chunks of the packed weight image
constexpr int THREADS = 512;
constexpr int PER_THREAD = (PATCH + THREADS - 1) / THREADS;   // 10 patch elements per thread
constexpr int CO_STRIDE = NPX;                     // conv-out row stride (floats)

struct StemArgs {
  const float* x;
  const u32x4* wq;
  const float* scale;
  const float* shift;
  float* y;
  u32x4* ys;
  int n, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x;
  float neg;
};

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  f32x2v t = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
}

// [64][3][7][7] fp32 -> A-fragment image [ks 6][mb 4][plane hi, lo][lane 64][8 x bf16]: lane l of block mb holds output channel
// mb * 16 + (l & 15), k-group g = 4 * ks + (l >> 4) = (ky, ci) = (g / 3, g % 3), elements kx = 0..6 and a zero.
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, u32x4* __restrict__ wq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KSTEPS * 4 * 64) return;
  const int lane = i & 63, mb = (i >> 6) & 3, ks = i >> 8;
  const int co = mb * 16 + (lane & 15), g = 4 * ks + (lane >> 4);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (g < NGROUPS) {
    const int ky = g / 3, ci = g - 3 * ky;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) v[kx] = w[((co * 3 + ci) * 7 + ky) * 7 + kx];
  }
  u32x4 hi, lo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    hi[q] = pack_bf16(v[2 * q], v[2 * q + 1]);
    lo[q] = pack_bf16(v[2 * q] - __builtin_bit_cast(float, hi[q] << 16), v[2 * q + 1] - __builtin_bit_cast(float, hi[q] & 0xFFFF0000u));
  }
  wq[((ks * 4 + mb) * 2 + 0) * 64 + lane] = hi;
  wq[((ks * 4 + mb) * 2 + 1) * 64 + lane] = lo;
}

__global__ __launch_bounds__(THREADS, 1) void stem_fused_kernel(StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* As = reinterpret_cast<u32x4*>(smem);                                   // 49,152 bytes
  unsigned short* Ph = reinterpret_cast<unsigned short*>(smem + A_CHUNKS * 16);    // patch, hi plane: [ci][PR][PC] bf16
  unsigned short* Pl = Ph + PATCH;                                              // patch, lo plane
  float* Co = reinterpret_cast<float*>(smem + A_CHUNKS * 16 + ((2 * PATCH * 2 + 15) & ~15));   // conv + ABN values [64][NPX]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = a.n * a.tiles_y * a.tiles_x;

  for (int i = tid; i < A_CHUNKS; i += THREADS) As[i] = a.wq[i];

  // request the patch of `tile` into registers (zeros outside the image)
  float pre[PER_THREAD];
  auto request = [&](int tile) {
    const int img = tile / (a.tiles_y * a.tiles_x), t2 = tile - img * (a.tiles_y * a.tiles_x);
    const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
    const int iy0 = 2 * (2 * ty * PT_H - 1) - 3, ix0 = 2 * (2 * tx * PT_W - 1) - 3;
    const float* xi = a.x + (long)img * 3 * a.H * a.W;
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const int e = tid + j * THREADS;
      const int ci = e / (PR * PC), r2 = e - ci * (PR * PC);
      const int r = r2 / PC, c = r2 - r * PC;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = e < PATCH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      pre[j] = ok ? xi[((long)ci * a.H + iy) * a.W + ix] : 0.f;
    }
  };

  // per-lane constants of the MFMA phase
  const int l15 = lane & 15, q = lane >> 4;
  int goff[KSTEPS];                                   // element offset of this lane's k-group inside the patch, per K-step
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int g = min(4 * ks + q, NGROUPS - 1);       // zero groups (weights are zero) read group 20's data
    const int ky = g / 3, ci = g - 3 * ky;
    goff[ks] = (ci * PR + ky) * PC;
  }
  // work split: wave = (channel half mh, pixel-block phase nbq): 32 channels x the pixel blocks nbq, nbq + 4, ... (5 or 4 of the 19)
  constexpr int MYB = (NBLK + 3) / 4;
  const int mh = wave & 1, nbq = wave >> 1;
  int pbase[MYB];
#pragma unroll
  for (int b = 0; b < MYB; ++b) {
    const int px = min((nbq + 4 * b) * 16 + l15, NPX - 1);
    const int cy = px / CW, cx = px - cy * CW;
    pbase[b] = 2 * cy * PC + 2 * cx;
  }
  float sc[2][4], sf[2][4];                           // folded ABN of this lane's 8 output channels
#pragma unroll
  for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = (2 * mh + m2) * 16 + q * 4 + r;
      sc[m2][r] = a.scale[ch];
      sf[m2][r] = a.shift[ch];
    }

  int tile = blockIdx.x;
  if (tile < ntiles) request(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int img = tile / (a.tiles_y * a.tiles_x), t2 = tile - img * (a.tiles_y * a.tiles_x);
    const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
    // ---- patch -> LDS as bf16 hi / lo ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const int e = tid + j * THREADS;
      if (e < PATCH) {
        const float v = pre[j];
        const unsigned h = pack_bf16(v, 0.f) & 0xFFFFu;
        const float rest = v - __builtin_bit_cast(float, h << 16);
        Ph[e] = (unsigned short)h;
        Pl[e] = (unsigned short)(pack_bf16(rest, 0.f) & 0xFFFFu);
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) request(tile + gridDim.x);     // lands under the MFMA phase

    // ---- MFMA phase: acc[b][mb] = 16 channels x 16 conv pixels ------------------------------------------------------------------------
    f32x4v acc[MYB][2];
#pragma unroll
    for (int b = 0; b < MYB; ++b)
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) acc[b][m2] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2) {
        ah[m2] = __builtin_bit_cast(bf16x8, As[((ks * 4 + 2 * mh + m2) * 2 + 0) * 64 + lane]);
        al[m2] = __builtin_bit_cast(bf16x8, As[((ks * 4 + 2 * mh + m2) * 2 + 1) * 64 + lane]);
      }
#pragma unroll
      for (int b = 0; b < MYB; ++b) {
        if (nbq + 4 * b < NBLK) {                      // wave-uniform
          const unsigned* sh = reinterpret_cast<const unsigned*>(Ph + goff[ks] + pbase[b]);     // 4-byte aligned: PC and both offsets are even
          const unsigned* sl = reinterpret_cast<const unsigned*>(Pl + goff[ks] + pbase[b]);
          u32x4 h4, l4;
#pragma unroll
          for (int d = 0; d < 4; ++d) { h4[d] = sh[d]; l4[d] = sl[d]; }
          const bf16x8 bh = __builtin_bit_cast(bf16x8, h4), bl = __builtin_bit_cast(bf16x8, l4);
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
            acc[b][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m2], bh, acc[b][m2], 0, 0, 0);
            acc[b][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m2], bl, acc[b][m2], 0, 0, 0);
            acc[b][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m2], bh, acc[b][m2], 0, 0, 0);
          }
        }
      }
    }

    // ---- ABN + leaky-ReLU, conv pixels outside the 257 x 257 map become -inf (the pooling pads with -inf), -> LDS -----------------------
    const int cy0 = 2 * ty * PT_H - 1, cx0 = 2 * tx * PT_W - 1;
#pragma unroll
    for (int b = 0; b < MYB; ++b) {
      const int px = (nbq + 4 * b) * 16 + l15;
      if (nbq + 4 * b < NBLK && px < NPX) {
        const int cy = px / CW, cx = px - cy * CW;
        const bool inside = (unsigned)(cy0 + cy) < (unsigned)a.Hc && (unsigned)(cx0 + cx) < (unsigned)a.Wc;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ch = (2 * mh + m2) * 16 + q * 4 + r;
            const float u = acc[b][m2][r] * sc[m2][r] + sf[m2][r];
            Co[ch * CO_STRIDE + px] = inside ? (u > 0.f ? u : u * a.neg) : -INFINITY;
          }
      }
    }
    __syncthreads();

    // ---- 3x3 / stride 2 max: thread = (8-channel group, pooled pixel) -------------------------------------------------------------------
    {
      const int g8 = tid >> 6, pp = tid & 63;
      const int pr = pp >> 4, pc = pp & 15;
      const int py = ty * PT_H + pr, pxo = tx * PT_W + pc;
      if (py < a.Hp && pxo < a.Wp) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float* c = Co + (g8 * 8 + r) * CO_STRIDE + (2 * pr) * CW + 2 * pc;
          float m = c[0];
          m = fmaxf(m, c[1]); m = fmaxf(m, c[2]);
          m = fmaxf(m, c[CW]); m = fmaxf(m, c[CW + 1]); m = fmaxf(m, c[CW + 2]);
          m = fmaxf(m, c[2 * CW]); m = fmaxf(m, c[2 * CW + 1]); m = fmaxf(m, c[2 * CW + 2]);
          v[r] = m;
        }
        const long hw = (long)a.Hp * a.Wp, p = (long)py * a.Wp + pxo;
        if (a.y) {
#pragma unroll
          for (int r = 0; r < 8; ++r) a.y[((long)img * 64 + g8 * 8 + r) * hw + p] = v[r];
        }
        if (a.ys) {
          u32x4 hi, lo;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned h_, l_;
            plfmt::split2(v[2 * i], v[2 * i + 1], h_, l_);      // the planes format of this build (planes_fmt.h); the stem's own products stay bf16x3
            hi[i] = h_;
            lo[i] = l_;
          }
          u32x4* ys = a.ys + (((long)img * 8 + g8) * 2) * hw + p;
          ys[0] = hi;
          ys[hw] = lo;
        }
      }
    }
    // the next iteration's first barrier (after the patch store) orders these LDS reads before the next conv-value stores
  }
}

}  // namespace

extern "C" size_t cl4_stem_weight_bytes(void) { return (size_t)A_CHUNKS * 16; }

extern "C" int cl4_pack_stem_weight(const float* w, void* wq, void* stream) {
  CL4_REQUIRE(w && wq, "pack_stem_weight: null tensor");
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3(cl4::cdiv(KSTEPS * 4 * 64, 256)), dim3(256), 0, cl4::as_stream(stream), w, (u32x4*)wq);
  return cl4::launch_status();
}

extern "C" int cl4_stem_fused(const float* x, const void* wq, const float* scale, const float* shift, float slope, float* y, void* ys,
                              int n, int H, int W, void* stream) {
  CL4_REQUIRE(x && wq && scale && shift && (y || ys), "stem_fused: null tensor");
  CL4_REQUIRE(n > 0 && H >= 7 && W >= 7, "stem_fused: bad dims");
  StemArgs a;
  a.x = x; a.wq = (const u32x4*)wq; a.scale = scale; a.shift = shift; a.y = y; a.ys = (u32x4*)ys;
  a.n = n; a.H = H; a.W = W;
  a.Hc = (H + 6 - 7) / 2 + 1; a.Wc = (W + 6 - 7) / 2 + 1;          // conv 7x7 / 2 / pad 3
  a.Hp = (a.Hc + 2 - 3) / 2 + 1; a.Wp = (a.Wc + 2 - 3) / 2 + 1;    // pool 3x3 / 2 / pad 1
  a.tiles_y = cl4::cdiv(a.Hp, PT_H); a.tiles_x = cl4::cdiv(a.Wp, PT_W);
  a.neg = slope;
  const long ntiles = (long)n * a.tiles_y * a.tiles_x;
  CL4_REQUIRE(ntiles < (1L << 30), "stem_fused: too many tiles");
  const size_t lds = (size_t)A_CHUNKS * 16 + ((2 * PATCH * 2 + 15) & ~15) + (size_t)64 * CO_STRIDE * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { cl4::set_error("stem_fused: cannot reserve %zu bytes of LDS", lds); return (int)e; }
    attr_set = true;
  }
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL(stem_fused_kernel, dim3(grid), dim3(THREADS), lds, cl4::as_stream(stream), a);
  return cl4::launch_status();
}
