// Label-stage kernels, part 2: BESTIE pseudo-label generation (modules/utils.py:179-253) and
// refined-label generation (modules/utils.py:257-385, 463-632) entirely on the device.
//
// The reference runs these per image / class / contour on the host (OpenCV + numpy + dozens of
// .cpu()/.item() syncs). Here every contour of every image is processed in parallel:
//   8-CCL of seg_gt -> per-root statistics -> contour table (area >= 20)
//   pseudo labels : peak count per contour, paint weight/offset, gaussian splat list
//   refined labels: NMS centres (41x41, LDS tiled), offset-cluster centres (4-CCL of the weak-offset
//                   region), merge, rank-sort of the centre list, nearest-centre grouping, per-instance
//                   statistics with packed 64-bit atomics, finalise + splat + paint.
// All data-dependent counts stay in device memory; kernels run fixed grids that loop over them.
#include "label_common.h"

extern "C" int cl4_ccl(const uint8_t* cls_map, int b, int H, int W, int connectivity, int* labels, void* stream);
extern "C" int cl4_cc_stats(const int* labels, int b, int H, int W, unsigned* area, unsigned* sumx, unsigned* sumy, void* stream);

namespace {

using cl4::Bilin;
using cl4::bilin_half_pixel;
using cl4::bilin_sample;

constexpr int MIN_MASK = 20;       // MINIMUM_MASK_SIZE, modules/utils.py:14
constexpr int MAX_INST = 5;        // MAXIMUM_NUM_INST, modules/utils.py:15
constexpr int CT_CAP = 1 << 16;    // contour table capacity (per batch)
constexpr int LIST_CAP = 1 << 16;  // splat / centre / cluster-candidate list capacity

struct Contour {
  int root, b, cls, cx, cy;   // cls = new-class index (0..cnew-1); centroid truncated to int
  int ncen0, ncen, nmax;      // NMS centres, all centres, max assigned instance id
  unsigned wcnt, wsumx, wsumy;  // weak-offset pixels of this contour (for the background cluster)
  int cstart;
};
struct Splat { int b, c, cx, cy; };
struct Center { unsigned long long key; int cid, y, x, pad; };
struct Inst {
  unsigned cnt;
  double sump;
  unsigned long long amax;
  int py, px;
  float conf;
  int valid;
};
struct Counters { int n_contour, n_psplat, n_center, n_cand, overflow, n_nms, pad[2]; };

struct WS {
  int* L8; unsigned* area; unsigned* sumx; unsigned* sumy; unsigned* pcnt; int* cid_map;
  uint8_t* weak; int* L4; unsigned* area4; unsigned* sumx4; unsigned* sumy4; uint8_t* mark; uint8_t* ins_id;
  Counters* cnt; Contour* ct; Splat* psplat; Center* cen; Center* cen_sorted; Center* cand; Inst* inst;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t layout(WS* ws, char* base, long N) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
  char* p;
  p = take(N * 4); if (ws) ws->L8 = (int*)p;
  p = take(N * 4); if (ws) ws->area = (unsigned*)p;
  p = take(N * 4); if (ws) ws->sumx = (unsigned*)p;
  p = take(N * 4); if (ws) ws->sumy = (unsigned*)p;
  p = take(N * 4); if (ws) ws->pcnt = (unsigned*)p;
  p = take(N * 4); if (ws) ws->cid_map = (int*)p;
  p = take(N);     if (ws) ws->weak = (uint8_t*)p;
  p = take(N * 4); if (ws) ws->L4 = (int*)p;
  p = take(N * 4); if (ws) ws->area4 = (unsigned*)p;
  p = take(N * 4); if (ws) ws->sumx4 = (unsigned*)p;
  p = take(N * 4); if (ws) ws->sumy4 = (unsigned*)p;
  p = take(N);     if (ws) ws->mark = (uint8_t*)p;
  p = take(N);     if (ws) ws->ins_id = (uint8_t*)p;
  p = take(sizeof(Counters)); if (ws) ws->cnt = (Counters*)p;
  p = take(sizeof(Contour) * CT_CAP); if (ws) ws->ct = (Contour*)p;
  p = take(sizeof(Splat) * LIST_CAP); if (ws) ws->psplat = (Splat*)p;
  p = take(sizeof(Center) * LIST_CAP); if (ws) ws->cen = (Center*)p;
  p = take(sizeof(Center) * LIST_CAP); if (ws) ws->cen_sorted = (Center*)p;
  p = take(sizeof(Center) * LIST_CAP); if (ws) ws->cand = (Center*)p;
  p = take(sizeof(Inst) * (size_t)CT_CAP * (MAX_INST + 1)); if (ws) ws->inst = (Inst*)p;
  return off;
}

struct IO {
  int b, H, W, cnew, old_classes, hc, wc, center_ctot, K, sigma, nms_r, run_refine;
  float pseudo_thresh, refine_thresh, beta;
  const uint8_t* seg_gt; const float* prob_win; const float* label_new;
  const float* peak_conf; const int* peak_y; const int* peak_x;
  const float* center_lo; const float* offset_lo; const float* gauss;
  float* p_center; float* p_offset; float* p_weight;
  float* r_center; float* r_offset; float* r_weight;
  // validation mode (get_ins_map, dataset/utils.py:795-892): every class, no instance cap, instance table instead of maps
  int val, ignore, min_mask, v_cap;
  int* v_ins_map; float* v_table; int* v_count;
};

__device__ __forceinline__ float center_up(const IO& io, int b, int cls, int Y, int X) {
  const float sy = (float)io.hc / (float)io.H, sx = (float)io.wc / (float)io.W;
  Bilin by = bilin_half_pixel(Y, sy, io.hc), bx = bilin_half_pixel(X, sx, io.wc);
  const float* plane = io.center_lo + ((long)b * io.center_ctot + (io.old_classes - 1) + cls) * io.hc * io.wc;
  return bilin_sample(plane, io.wc, by, bx);
}
__device__ __forceinline__ void offset_up(const IO& io, int b, int Y, int X, float& oy, float& ox) {
  const float sy = (float)io.hc / (float)io.H, sx = (float)io.wc / (float)io.W;
  Bilin by = bilin_half_pixel(Y, sy, io.hc), bx = bilin_half_pixel(X, sx, io.wc);
  const float* plane = io.offset_lo + (long)b * 2 * io.hc * io.wc;
  oy = bilin_sample(plane, io.wc, by, bx);
  ox = bilin_sample(plane + (long)io.hc * io.wc, io.wc, by, bx);
}

// ---- pseudo labels -----------------------------------------------------------------------------
// train.py:452-458 + modules/utils.py:232-236: count, per contour, the accepted peaks lying on it.
__global__ void peak_count_kernel(IO io, WS ws) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int total = io.b * io.cnew * io.K;
  if (i >= total) return;
  int bc = i / io.K;
  int b = bc / io.cnew, c = bc - b * io.cnew;
  if (io.label_new[bc] == 0.f) return;
  if (io.peak_conf[i] < io.pseudo_thresh) return;  // scores are sorted descending: prefix rule of train.py:456-457
  int y = io.peak_y[i], x = io.peak_x[i];
  long p = ((long)b * io.H + y) * io.W + x;
  if (io.seg_gt[p] != (uint8_t)(io.old_classes + c)) return;
  atomicAdd(&ws.pcnt[ws.L8[p]], 1u);
}

// Per pixel: pseudo weight/offset (accept: area >= 20 and exactly one peak). Root pixels register the
// contour (area >= 20) for the refine stage and the gaussian splat of accepted components.
__global__ void pseudo_paint_kernel(IO io, WS ws) {
  const long N = (long)io.b * io.H * io.W, HW = (long)io.H * io.W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    float wgt = 0.f, oy = 0.f, ox = 0.f;
    int r = ws.L8[p];
    if (r >= 0) {
      unsigned a = ws.area[r];
      if (a >= (unsigned)io.min_mask) {
        int cx = (int)(ws.sumx[r] / a), cy = (int)(ws.sumy[r] / a);
        bool acc = ws.pcnt[r] == 1u;
        int b = p / HW;
        long q = p - (long)b * HW;
        int y = q / io.W, x = q - (long)y * io.W;
        if (acc) { wgt = 1.f; oy = (float)cy - (float)y; ox = (float)cx - (float)x; }
        if (r == (int)p) {
          int cls = (int)io.seg_gt[p] - io.old_classes;
          if (acc) {
            int s = atomicAdd(&ws.cnt->n_psplat, 1);
            if (s < LIST_CAP) ws.psplat[s] = Splat{b, cls, cx, cy}; else ws.cnt->overflow = 1;
          }
          int cid = atomicAdd(&ws.cnt->n_contour, 1);
          if (cid < CT_CAP) {
            Contour c{};
            c.root = r; c.b = b; c.cls = cls; c.cx = cx; c.cy = cy;
            ws.ct[cid] = c;
            ws.cid_map[r] = cid;
          } else {
            ws.cnt->overflow = 1;
          }
        }
      }
    }
    if (io.p_weight) {
      int b = p / HW;
      long q = p - (long)b * HW;
      io.p_weight[p] = wgt;
      io.p_offset[((long)b * 2) * HW + q] = oy;
      io.p_offset[((long)b * 2 + 1) * HW + q] = ox;
    }
  }
}

// center_map_gen (modules/utils.py:84-119): np.maximum paste of the clipped (6s+3)^2 gaussian. One workgroup
// per list entry (grid-stride over the device-side count). Values are positive -> int-ordered atomicMax.
__device__ void splat_one(float* center, int cnew, int H, int W, int sigma, const float* gauss, int b, int c, int cx, int cy) {
  if (cx < 0 || cy < 0 || cx >= W || cy >= H) return;
  const int size = 6 * sigma + 3, half = 3 * sigma + 1;
  float* plane = center + ((long)b * cnew + c) * H * W;
  for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
    int gy = i / size, gx = i - gy * size;
    int Y = cy - half + gy, X = cx - half + gx;
    if ((unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W)
      atomicMax((int*)&plane[(long)Y * W + X], __float_as_int(gauss[i]));
  }
}

__global__ void pseudo_splat_kernel(IO io, WS ws) {
  int n = min(ws.cnt->n_psplat, LIST_CAP);
  for (int e = blockIdx.x; e < n; e += gridDim.x) {
    Splat s = ws.psplat[e];
    splat_one(io.p_center, io.cnew, io.H, io.W, io.sigma, io.gauss, s.b, s.c, s.cx, s.cy);
  }
}

// ---- refined labels ----------------------------------------------------------------------------
// cluster_peaks (modules/utils.py:608-632): weak-offset region of every accepted contour.
__global__ void weak_map_kernel(IO io, WS ws) {
  const long N = (long)io.b * io.H * io.W, HW = (long)io.H * io.W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    uint8_t wv = 0;
    int r = ws.L8[p];
    if (r >= 0) {
      int cid = ws.cid_map[r];
      if (cid >= 0) {
        int b = p / HW;
        long q = p - (long)b * HW;
        int y = q / io.W, x = q - (long)y * io.W;
        float oy, ox;
        offset_up(io, b, y, x, oy, ox);
        float mag = sqrtf(ox * ox + oy * oy);
        if (mag < 2.5f) {
          wv = io.seg_gt[p];
          atomicAdd(&ws.ct[cid].wcnt, 1u);
          atomicAdd(&ws.ct[cid].wsumx, (unsigned)x);
          atomicAdd(&ws.ct[cid].wsumy, (unsigned)y);
        }
      }
    }
    ws.weak[p] = wv;
  }
}

// find_instance_center (modules/utils.py:463-502) restricted to each contour's own (masked) centre map:
// threshold, (2R+1)^2 max-pool NMS, keep pixels equal to their window maximum.
constexpr int NT = 32;
constexpr int NR_MAX = 20;
constexpr int NTS = NT + 2 * NR_MAX;  // 72

__global__ __launch_bounds__(256) void nms_center_kernel(IO io, WS ws) {
  __shared__ float val[NTS * NTS];
  __shared__ int root[NTS * NTS];
  __shared__ int any_cand;
  const int R = io.nms_r;
  const int TS = NT + 2 * R;
  const int tiles_x = (io.W + NT - 1) / NT;
  const int ty0 = (blockIdx.x / tiles_x) * NT, tx0 = (blockIdx.x % tiles_x) * NT;
  const int b = blockIdx.y;
  const long HW = (long)io.H * io.W;
  if (threadIdx.x == 0) any_cand = 0;
  __syncthreads();
  // value / component root of one pixel of the (tile + halo) window
  auto fetch = [&](int ty, int tx, bool interior) {
    int Y = ty0 - R + ty, X = tx0 - R + tx;
    float v = -1.f;
    int r = -1;
    if ((unsigned)Y < (unsigned)io.H && (unsigned)X < (unsigned)io.W) {
      long p = (long)b * HW + (long)Y * io.W + X;
      int rr = ws.L8[p];
      if (rr >= 0 && ws.cid_map[rr] >= 0) {
        r = rr;
        float cv = center_up(io, b, (int)io.seg_gt[p] - io.old_classes, Y, X);
        v = cv > io.refine_thresh ? cv : -1.f;  // F.threshold(x, thr, -1)
        if (interior && v > 0.f) any_cand = 1;
      }
    }
    val[ty * NTS + tx] = v;
    root[ty * NTS + tx] = r;
  };
  // the 32 x 32 interior first: a tile without a candidate (most tiles: the maps hold a few dozen centres) never reads its halo,
  // which is 4 of the 5 bilinear samples of the window at R = 20 (r03: 0.50 -> 0.12 ms per step)
  for (int i = threadIdx.x; i < NT * NT; i += blockDim.x) fetch(R + i / NT, R + i % NT, true);
  __syncthreads();
  if (!any_cand) return;
  for (int i = threadIdx.x; i < TS * TS; i += blockDim.x) {
    int ty = i / TS, tx = i - ty * TS;
    if (ty >= R && ty < R + NT && tx >= R && tx < R + NT) continue;
    fetch(ty, tx, false);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NT * NT; i += blockDim.x) {
    int y = i / NT, x = i - y * NT;
    int Y = ty0 + y, X = tx0 + x;
    if (Y >= io.H || X >= io.W) continue;
    float v = val[(y + R) * NTS + x + R];
    if (!(v > 0.f)) continue;
    int r = root[(y + R) * NTS + x + R];
    bool is_max = true;
    for (int dy = 0; dy <= 2 * R && is_max; ++dy)
      for (int dx = 0; dx <= 2 * R; ++dx) {
        int j = (y + dy) * NTS + x + dx;
        if (root[j] == r && val[j] > v) { is_max = false; break; }
      }
    if (!is_max) continue;
    int cid = ws.cid_map[r];
    int s = atomicAdd(&ws.cnt->n_center, 1);
    if (s < LIST_CAP) {
      long p = (long)b * HW + (long)Y * io.W + X;
      Center c;
      c.cid = cid; c.y = Y; c.x = X; c.pad = 0;
      c.key = ((unsigned long long)cid << 34) | (unsigned long long)(p + 1);  // kind 0, row-major order
      ws.cen[s] = c;
      atomicAdd(&ws.ct[cid].ncen0, 1);
      atomicAdd(&ws.ct[cid].ncen, 1);
    } else {
      ws.cnt->overflow = 1;
    }
  }
}

// Candidates from the 4-connected components of the weak-offset region whose area lies in (21-beta, 21+beta),
// plus (k = 0 of the reference loop) the background component of each contour.
__device__ void push_candidate(const IO& io, WS& ws, int cid, int cy, int cx, long order) {
  const Contour& ct = ws.ct[cid];
  if ((unsigned)cy >= (unsigned)io.H || (unsigned)cx >= (unsigned)io.W) return;
  long p = ((long)ct.b * io.H + cy) * io.W + cx;
  // ctr_hmp[0,0,cy,cx] > 0.05 on the contour-masked centre map
  if (ws.L8[p] != ct.root) return;
  if (!(center_up(io, ct.b, ct.cls, cy, cx) > 0.05f)) return;
  int s = atomicAdd(&ws.cnt->n_cand, 1);
  if (s >= LIST_CAP) { ws.cnt->overflow = 1; return; }
  Center c;
  c.cid = cid; c.y = cy; c.x = cx; c.pad = 0;
  c.key = ((unsigned long long)cid << 34) | (1ull << 33) | (unsigned long long)(order + 1);
  ws.cand[s] = c;
}

__global__ void cluster_candidate_kernel(IO io, WS ws) {
  const long N = (long)io.b * io.H * io.W;
  const float lo = 21.f - io.beta, hi = 21.f + io.beta;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    if (ws.L4[p] != (int)p) continue;  // component roots only (root = first pixel in raster order)
    unsigned a = ws.area4[p];
    if (!(lo < (float)a && (float)a < hi)) continue;
    int cid = ws.cid_map[ws.L8[p]];
    if (cid < 0) continue;
    push_candidate(io, ws, cid, (int)(ws.sumy4[p] / a), (int)(ws.sumx4[p] / a), p);
  }
}

__global__ void background_candidate_kernel(IO io, WS ws) {
  int n = min(ws.cnt->n_contour, CT_CAP);
  const float lo = 21.f - io.beta, hi = 21.f + io.beta;
  const long HW = (long)io.H * io.W;
  for (int cid = blockIdx.x * blockDim.x + threadIdx.x; cid < n; cid += gridDim.x * blockDim.x) {
    const Contour& ct = ws.ct[cid];
    long a = HW - (long)ct.wcnt;
    if (a <= 0 || !(lo < (float)a && (float)a < hi)) continue;
    // centroid of everything that is not weak: image sums minus the weak sums (exact integers)
    unsigned long long tx = (unsigned long long)io.H * ((unsigned long long)io.W * (io.W - 1) / 2) - ct.wsumx;
    unsigned long long ty = (unsigned long long)io.W * ((unsigned long long)io.H * (io.H - 1) / 2) - ct.wsumy;
    push_candidate(io, ws, cid, (int)(ty / (unsigned long long)a), (int)(tx / (unsigned long long)a), -1);
  }
}

// get_instance_segmentation merge rule (modules/utils.py:575-592): with no NMS centre every candidate is
// taken; otherwise only candidates farther than 100 px from every NMS centre. Accepted ones are marked 1.0.
__global__ void cluster_merge_kernel(IO io, WS ws) {
  int ncand = min(ws.cnt->n_cand, LIST_CAP);
  const int ncen = ws.cnt->n_nms;  // snapshot of the NMS centres taken before this launch (appends land behind it)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += gridDim.x * blockDim.x) {
    Center c = ws.cand[i];
    const Contour& ct = ws.ct[c.cid];
    bool take = true;
    if (ct.ncen0 > 0) {
      float dmin = INFINITY;
      for (int j = 0; j < ncen; ++j) {
        Center o = ws.cen[j];
        if (o.cid != c.cid) continue;
        float dy = (float)o.y - (float)c.y, dx = (float)o.x - (float)c.x;
        dmin = fminf(dmin, sqrtf(dy * dy + dx * dx));
      }
      take = dmin > 100.f;
    }
    if (!take) continue;
    int s = atomicAdd(&ws.cnt->n_center, 1);
    if (s >= LIST_CAP) { ws.cnt->overflow = 1; continue; }
    ws.cen[s] = c;
    atomicAdd(&ws.ct[c.cid].ncen, 1);
    ws.mark[((long)ct.b * io.H + c.y) * io.W + c.x] = 1;
  }
}

__global__ void snapshot_kernel(WS ws) { ws.cnt->n_nms = min(ws.cnt->n_center, LIST_CAP); }

// Rank sort of the centre list by key (contour, kind, order): O(n^2) compares, n is tens to a few thousand.
__global__ void center_sort_kernel(WS ws) {
  int n = min(ws.cnt->n_center, LIST_CAP);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    Center c = ws.cen[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += ws.cen[j].key < c.key ? 1 : 0;
    ws.cen_sorted[rank] = c;
  }
}
__global__ void center_range_kernel(WS ws) {
  int n = min(ws.cnt->n_center, LIST_CAP);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int cid = ws.cen_sorted[i].cid;
    if (i == 0 || ws.cen_sorted[i - 1].cid != cid) ws.ct[cid].cstart = i;
  }
}

// group_pixels (modules/utils.py:505-542) + per-instance statistics of refine_label_generation :327-345.
__global__ void group_kernel(IO io, WS ws) {
  const long N = (long)io.b * io.H * io.W, HW = (long)io.H * io.W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    int r = ws.L8[p];
    if (r < 0) continue;
    int cid = ws.cid_map[r];
    if (cid < 0) continue;
    const Contour& ct = ws.ct[cid];
    if (ct.ncen == 0) continue;
    int b = p / HW;
    long q = p - (long)b * HW;
    int y = q / io.W, x = q - (long)y * io.W;
    float oy, ox;
    offset_up(io, b, y, x, oy, ox);
    float ly = (float)y + oy, lx = (float)x + ox;
    float best = INFINITY;
    int id = 0;
    for (int k = 0; k < ct.ncen; ++k) {
      const Center& c = ws.cen_sorted[ct.cstart + k];
      float dy = (float)c.y - ly, dx = (float)c.x - lx;
      float d = sqrtf(dy * dy + dx * dx);
      if (d < best) { best = d; id = k + 1; }
    }
    ws.ins_id[p] = (uint8_t)min(id, 255);
    atomicMax(&ws.ct[cid].nmax, id);
    if (io.val) io.v_ins_map[p] = ct.cstart + id - 1;      // one table row per centre (sorted by contour)
    if (io.val || id <= MAX_INST) {
      Inst& in = io.val ? ws.inst[ct.cstart + id - 1] : ws.inst[(long)cid * (MAX_INST + 1) + id];
      atomicAdd(&in.cnt, 1u);
      atomicAdd(&in.sump, (double)io.prob_win[p]);
      float cv = ws.mark[p] ? 1.0f : center_up(io, b, ct.cls, y, x);
      unsigned long long key = ((unsigned long long)cl4::ordered_bits(cv) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)q);
      atomicMax(&in.amax, key);
    }
  }
}

// refine_label_generation :316-372 — one workgroup per contour: decide each instance's point / confidence,
// splat the gaussian into the refined centre map.
__global__ __launch_bounds__(256) void finalize_kernel(IO io, WS ws) {
  int n = min(ws.cnt->n_contour, CT_CAP);
  for (int cid = blockIdx.x; cid < n; cid += gridDim.x) {
    const Contour ct = ws.ct[cid];
    if (ct.ncen == 0 || ct.nmax > MAX_INST) continue;
    for (int id = 1; id <= ct.nmax; ++id) {
      Inst& in = ws.inst[(long)cid * (MAX_INST + 1) + id];
      if (in.cnt == 0) continue;
      unsigned long long key = in.amax;
      float center_score = cl4::from_ordered_bits((unsigned)(key >> 32));
      unsigned q = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu);
      int py = q / io.W, px = q - py * io.W;
      float seg_score = (float)(in.sump / (double)in.cnt);
      double conf;
      if (center_score < io.refine_thresh) { py = ct.cy; px = ct.cx; conf = (double)seg_score; }
      else conf = (double)center_score * (double)seg_score;
      conf = conf < 0.0 ? 0.0 : (conf > 1.0 ? 1.0 : conf);
      __syncthreads();
      if (threadIdx.x == 0) { in.py = py; in.px = px; in.conf = (float)conf; in.valid = 1; }
      splat_one(io.r_center, io.cnew, io.H, io.W, io.sigma, io.gauss, ct.b, ct.cls, px, py);
    }
  }
}

// Validation, --val_ignore off (modules/utils.py:599-602): a contour without any centre is one instance. A single centre
// at the contour's centroid makes group_pixels assign every pixel to id 1.
__global__ void fallback_center_kernel(IO io, WS ws) {
  int n = min(ws.cnt->n_contour, CT_CAP);
  for (int cid = blockIdx.x * blockDim.x + threadIdx.x; cid < n; cid += gridDim.x * blockDim.x) {
    Contour& ct = ws.ct[cid];
    if (ct.ncen != 0) continue;
    int s = atomicAdd(&ws.cnt->n_center, 1);
    if (s >= LIST_CAP) { ws.cnt->overflow = 1; continue; }
    Center c;
    c.cid = cid; c.y = ct.cy; c.x = ct.cx; c.pad = 0;
    c.key = ((unsigned long long)cid << 34) | 1ull;
    ws.cen[s] = c;
    ct.ncen = 1;
  }
}

// Validation scores (dataset/utils.py:866-880): seg_score = mean winning probability over the instance, center_score = the
// largest (contour-masked) centre value on it, replaced by seg_score for clustered centres (marked 1.0); one row per centre:
// {class, score, pixel count, contour id}; rows with count 0 (a centre that attracted no pixel) are dropped by the caller.
__global__ void val_finalize_kernel(IO io, WS ws) {
  int n = min(ws.cnt->n_center, min(LIST_CAP, io.v_cap));
  if (blockIdx.x == 0 && threadIdx.x == 0) *io.v_count = n;
  for (int gi = blockIdx.x * blockDim.x + threadIdx.x; gi < n; gi += gridDim.x * blockDim.x) {
    const Inst& in = ws.inst[gi];
    const Contour& ct = ws.ct[ws.cen_sorted[gi].cid];
    float score = 0.f;
    if (in.cnt > 0) {
      float center_score = cl4::from_ordered_bits((unsigned)(in.amax >> 32));
      const float seg_score = (float)(in.sump / (double)in.cnt);
      if (center_score >= 1.f) center_score = seg_score;
      score = (float)((double)center_score * (double)seg_score);
    }
    io.v_table[gi * 4 + 0] = (float)ct.cls;
    io.v_table[gi * 4 + 1] = score;
    io.v_table[gi * 4 + 2] = (float)in.cnt;
    io.v_table[gi * 4 + 3] = (float)ws.cen_sorted[gi].cid;
  }
}

__global__ void refine_paint_kernel(IO io, WS ws) {
  const long N = (long)io.b * io.H * io.W, HW = (long)io.H * io.W;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < N; p += (long)gridDim.x * blockDim.x) {
    float wgt = 0.f, oy = 0.f, ox = 0.f;
    int b = p / HW;
    long q = p - (long)b * HW;
    int r = ws.L8[p];
    if (r >= 0) {
      int cid = ws.cid_map[r];
      if (cid >= 0) {
        const Contour& ct = ws.ct[cid];
        int id = ws.ins_id[p];
        if (ct.ncen > 0 && ct.nmax <= MAX_INST && id >= 1 && id <= MAX_INST) {
          const Inst& in = ws.inst[(long)cid * (MAX_INST + 1) + id];
          if (in.valid) {
            int y = q / io.W, x = q - (long)y * io.W;
            wgt = in.conf;
            oy = (float)in.py - (float)y;
            ox = (float)in.px - (float)x;
          }
        }
      }
    }
    io.r_weight[p] = wgt;
    io.r_offset[((long)b * 2) * HW + q] = oy;
    io.r_offset[((long)b * 2 + 1) * HW + q] = ox;
  }
}

}  // namespace

extern "C" size_t cl4_label_workspace_bytes(int b, int H, int W) { return layout(nullptr, nullptr, (long)b * H * W); }

extern "C" int cl4_label_generation(const cl4_label_io* d, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(d && workspace, "label_generation: null pointer");
  CL4_REQUIRE(d->b > 0 && d->H > 0 && d->W > 0 && d->cnew > 0 && d->old_classes >= 1, "label_generation: bad dims");
  CL4_REQUIRE(d->seg_gt && d->prob_win && d->label_new && d->peak_conf && d->peak_y && d->peak_x && d->gauss,
              "label_generation: null input");
  CL4_REQUIRE(d->p_center && d->p_offset && d->p_weight, "label_generation: null pseudo output");
  CL4_REQUIRE(d->old_classes + d->cnew <= 255, "label_generation: class ids must fit uint8");
  CL4_REQUIRE(d->sigma >= 1, "label_generation: sigma");
  const long N = (long)d->b * d->H * d->W;
  CL4_REQUIRE(N < 2147483647L, "label_generation: index overflow");
  if (d->run_refine) {
    CL4_REQUIRE(d->center_lo && d->offset_lo && d->r_center && d->r_offset && d->r_weight, "label_generation: null refine tensor");
    CL4_REQUIRE(d->nms_kernel % 2 == 1 && (d->nms_kernel - 1) / 2 <= NR_MAX, "label_generation: --kernel %d unsupported (odd, <= 41)", d->nms_kernel);
    CL4_REQUIRE(d->center_ctot >= d->old_classes - 1 + d->cnew, "label_generation: centre map has too few channels");
  }
  if (workspace_bytes < cl4_label_workspace_bytes(d->b, d->H, d->W)) { cl4::set_error("label_generation: workspace too small"); return CL4_ENOSPC; }
  WS ws;
  layout(&ws, (char*)workspace, N);
  IO io;
  io.b = d->b; io.H = d->H; io.W = d->W; io.cnew = d->cnew; io.old_classes = d->old_classes; io.hc = d->hc; io.wc = d->wc;
  io.center_ctot = d->center_ctot; io.K = d->K; io.sigma = d->sigma; io.nms_r = (d->nms_kernel - 1) / 2; io.run_refine = d->run_refine;
  io.pseudo_thresh = d->pseudo_thresh; io.refine_thresh = d->refine_thresh; io.beta = d->beta;
  io.seg_gt = d->seg_gt; io.prob_win = d->prob_win; io.label_new = d->label_new;
  io.peak_conf = d->peak_conf; io.peak_y = d->peak_y; io.peak_x = d->peak_x;
  io.center_lo = d->center_lo; io.offset_lo = d->offset_lo; io.gauss = d->gauss;
  io.p_center = d->p_center; io.p_offset = d->p_offset; io.p_weight = d->p_weight;
  io.r_center = d->r_center; io.r_offset = d->r_offset; io.r_weight = d->r_weight;
  io.val = 0; io.ignore = 1; io.min_mask = MIN_MASK; io.v_cap = 0; io.v_ins_map = nullptr; io.v_table = nullptr; io.v_count = nullptr;

  hipStream_t st = cl4::as_stream(stream);
  const int g = cl4::stream_grid(N, 256);
  const size_t cmap = (size_t)d->b * d->cnew * d->H * d->W * sizeof(float);
  cl4::fill_bytes_async(ws.cnt, 0, sizeof(Counters), st);
  cl4::fill_bytes_async(ws.pcnt, 0, N * 4, st);
  cl4::fill_bytes_async(ws.cid_map, 0xFF, N * 4, st);
  cl4::fill_bytes_async(io.p_center, 0, cmap, st);
  int rc = cl4_ccl(io.seg_gt, d->b, d->H, d->W, 8, ws.L8, stream);
  if (rc) return rc;
  rc = cl4_cc_stats(ws.L8, d->b, d->H, d->W, ws.area, ws.sumx, ws.sumy, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(peak_count_kernel, dim3(cl4::cdiv((long)d->b * d->cnew * d->K, 256)), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(pseudo_paint_kernel, dim3(g), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(pseudo_splat_kernel, dim3(512), dim3(256), 0, st, io, ws);
  if (!d->run_refine) return cl4::launch_status();

  cl4::fill_bytes_async(io.r_center, 0, cmap, st);
  cl4::fill_bytes_async(ws.mark, 0, N, st);
  cl4::fill_bytes_async(ws.ins_id, 0, N, st);
  cl4::fill_bytes_async(ws.inst, 0, sizeof(Inst) * (size_t)CT_CAP * (MAX_INST + 1), st);
  hipLaunchKernelGGL(weak_map_kernel, dim3(g), dim3(256), 0, st, io, ws);
  rc = cl4_ccl(ws.weak, d->b, d->H, d->W, 4, ws.L4, stream);
  if (rc) return rc;
  rc = cl4_cc_stats(ws.L4, d->b, d->H, d->W, ws.area4, ws.sumx4, ws.sumy4, stream);
  if (rc) return rc;
  dim3 gn(cl4::cdiv(d->H, NT) * cl4::cdiv(d->W, NT), d->b);
  hipLaunchKernelGGL(nms_center_kernel, gn, dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(cluster_candidate_kernel, dim3(g), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(background_candidate_kernel, dim3(64), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, st, ws);
  hipLaunchKernelGGL(cluster_merge_kernel, dim3(64), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(center_sort_kernel, dim3(256), dim3(256), 0, st, ws);
  hipLaunchKernelGGL(center_range_kernel, dim3(64), dim3(256), 0, st, ws);
  hipLaunchKernelGGL(group_kernel, dim3(g), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(finalize_kernel, dim3(512), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(refine_paint_kernel, dim3(g), dim3(256), 0, st, io, ws);
  return cl4::launch_status();
}

// Device-side diagnostics of the last cl4_label_generation on this workspace: {n_contour, n_psplat, n_center,
// n_cand, overflow} copied to `out5` (device pointer, 5 ints). overflow != 0 means a table capacity was hit.
extern "C" int cl4_label_counters(const void* workspace, int b, int H, int W, int* out5, void* stream) {
  CL4_REQUIRE(workspace && out5, "label_counters: null pointer");
  WS ws;
  layout(&ws, (char*)workspace, (long)b * H * W);
  hipMemcpyAsync(out5, ws.cnt, 5 * sizeof(int), hipMemcpyDeviceToDevice, cl4::as_stream(stream));
  return cl4::launch_status();
}

static __global__ void counters_accumulate_kernel(const Counters* cnt, int* acc5) {
  const int v[5] = {cnt->n_contour, cnt->n_psplat, cnt->n_center, cnt->n_cand, cnt->overflow};
  for (int i = 0; i < 4; ++i) acc5[i] = max(acc5[i], v[i]);
  acc5[4] |= v[4];
}

// Sticky diagnostics for the training loop: acc5[0..3] = max over calls of {n_contour, n_psplat, n_center, n_cand},
// acc5[4] |= overflow. Enqueue-only (graph-capturable); the trainer reads acc5 at its logging sync and raises when a
// table capacity was hit, so truncated labels can never pass silently.
extern "C" int cl4_label_counters_accumulate(const void* workspace, int b, int H, int W, int* acc5, void* stream) {
  CL4_REQUIRE(workspace && acc5, "label_counters_accumulate: null pointer");
  WS ws;
  layout(&ws, (char*)workspace, (long)b * H * W);
  hipLaunchKernelGGL(counters_accumulate_kernel, dim3(1), dim3(1), 0, cl4::as_stream(stream), ws.cnt, acc5);
  return cl4::launch_status();
}

// Validation post-processing (dataset/utils.py:795-892 get_ins_map): one image, full-resolution maps, every class.
extern "C" int cl4_instance_map(const cl4_insmap_io* d, void* workspace, size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(d && workspace, "instance_map: null pointer");
  CL4_REQUIRE(d->H > 0 && d->W > 0 && d->classes > 0 && d->classes < 255, "instance_map: bad dims");
  CL4_REQUIRE(d->seg_map && d->prob_win && d->center && d->offset && d->ins_map && d->table && d->count && d->cap > 0,
              "instance_map: null tensor");
  CL4_REQUIRE(d->nms_kernel % 2 == 1 && (d->nms_kernel - 1) / 2 <= NR_MAX, "instance_map: --val_kernel %d unsupported (odd, <= 41)",
              d->nms_kernel);
  const long N = (long)d->H * d->W;
  CL4_REQUIRE(N < 2147483647L, "instance_map: index overflow");
  if (workspace_bytes < cl4_label_workspace_bytes(1, d->H, d->W)) { cl4::set_error("instance_map: workspace too small"); return CL4_ENOSPC; }
  WS ws;
  layout(&ws, (char*)workspace, N);
  IO io{};
  io.b = 1; io.H = d->H; io.W = d->W; io.cnew = d->classes; io.old_classes = 1; io.hc = d->H; io.wc = d->W;
  io.center_ctot = d->classes; io.K = 0; io.sigma = 1; io.nms_r = (d->nms_kernel - 1) / 2; io.run_refine = 1;
  io.pseudo_thresh = 2.f; io.refine_thresh = d->thresh; io.beta = d->beta;
  io.seg_gt = d->seg_map; io.prob_win = d->prob_win; io.center_lo = d->center; io.offset_lo = d->offset;
  io.val = 1; io.ignore = d->ignore; io.min_mask = d->min_mask; io.v_cap = d->cap;
  io.v_ins_map = d->ins_map; io.v_table = d->table; io.v_count = d->count;

  hipStream_t st = cl4::as_stream(stream);
  const int g = cl4::stream_grid(N, 256);
  cl4::fill_bytes_async(ws.cnt, 0, sizeof(Counters), st);
  cl4::fill_bytes_async(ws.pcnt, 0, N * 4, st);
  cl4::fill_bytes_async(ws.cid_map, 0xFF, N * 4, st);
  cl4::fill_bytes_async(d->ins_map, 0xFF, N * 4, st);
  int rc = cl4_ccl(io.seg_gt, 1, d->H, d->W, 8, ws.L8, stream);
  if (rc) return rc;
  rc = cl4_cc_stats(ws.L8, 1, d->H, d->W, ws.area, ws.sumx, ws.sumy, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(pseudo_paint_kernel, dim3(g), dim3(256), 0, st, io, ws);      // registers the contours
  cl4::fill_bytes_async(ws.mark, 0, N, st);
  cl4::fill_bytes_async(ws.ins_id, 0, N, st);
  cl4::fill_bytes_async(ws.inst, 0, sizeof(Inst) * (size_t)CT_CAP * (MAX_INST + 1), st);
  hipLaunchKernelGGL(weak_map_kernel, dim3(g), dim3(256), 0, st, io, ws);
  rc = cl4_ccl(ws.weak, 1, d->H, d->W, 4, ws.L4, stream);
  if (rc) return rc;
  rc = cl4_cc_stats(ws.L4, 1, d->H, d->W, ws.area4, ws.sumx4, ws.sumy4, stream);
  if (rc) return rc;
  dim3 gn(cl4::cdiv(d->H, NT) * cl4::cdiv(d->W, NT), 1);
  hipLaunchKernelGGL(nms_center_kernel, gn, dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(cluster_candidate_kernel, dim3(g), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(background_candidate_kernel, dim3(64), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, st, ws);
  hipLaunchKernelGGL(cluster_merge_kernel, dim3(64), dim3(256), 0, st, io, ws);
  if (!d->ignore) hipLaunchKernelGGL(fallback_center_kernel, dim3(64), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(center_sort_kernel, dim3(256), dim3(256), 0, st, ws);
  hipLaunchKernelGGL(center_range_kernel, dim3(64), dim3(256), 0, st, ws);
  hipLaunchKernelGGL(group_kernel, dim3(g), dim3(256), 0, st, io, ws);
  hipLaunchKernelGGL(val_finalize_kernel, dim3(64), dim3(256), 0, st, io, ws);
  return cl4::launch_status();
}
