// Input pipeline on the device: the per-sample pixel work of the reference's loader —
// RandomResizedCrop / Resize + RandomHorizontalFlip + ToTensor + Normalize (dataset/__init__.py:14-27 over
// dataset/transform.py:61-93,317-340,419-478,582-660), LabelTransform + instance gating (dataset/dataset.py:104-108,212-214)
// and label_generation (dataset/utils.py:299-350) — as batched gfx950 kernels over the decoded uint8 photos. The host keeps
// only JPEG decoding and the draw of the crop boxes; what crosses PCIe is the uint8 photo (<= 0.6 MB) instead of the float
// crop plus 22 target planes (23 MB per sample at 513^2).
//
// The reference's transforms run on PIL images, so the arithmetic reproduced here is Pillow's, bit for bit:
//  * BILINEAR: Resample.c two-pass convolution — per output position a window [xmin, xmin+n) and triangle weights computed
//    in double, normalised, quantised to 22-bit fixed point; horizontal pass into a uint8 temporary, then the vertical pass.
//  * NEAREST (labels): Geometry.c ImagingScaleAffine — source index = (int) of a double that is ACCUMULATED by repeated
//    addition (xo += scale), so the index tables are built by a sequential loop, one thread per (sample, axis).
// All kernels are HBM/latency-bound byte work: one thread per output element, coalesced along x.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Tables {      // per-sample integer tables inside the workspace
  int* bounds_h;     // [out_w][2]  (xmin, n)
  int* kk_h;         // [out_w][ksize]
  int* bounds_v;     // [out_h][2]
  int* kk_v;         // [out_h][ksize]
  int* nx;           // [out_w] nearest source column
  int* ny;           // [out_h] nearest source row
};

struct InArgs {
  const uint8_t* images;
  const uint8_t* labels;
  const cl4_sample_desc* descs;   // device copy
  int* tables;                    // b * table_ints
  uint8_t* temp;                  // [b][max_ch][out_w][3]
  int b, out_h, out_w, max_ch, ksize, table_ints;
  float mean[3], std[3];
  const uint8_t* lut;
  float* out_img;
  long long* out_seg;
  uint8_t* out_mask;
};

__device__ __forceinline__ Tables tables_of(const InArgs& a, int s) {
  int* p = a.tables + (long)s * a.table_ints;
  Tables t;
  t.bounds_h = p; p += 2 * a.out_w;
  t.kk_h = p;     p += a.ksize * a.out_w;
  t.bounds_v = p; p += 2 * a.out_h;
  t.kk_v = p;     p += a.ksize * a.out_h;
  t.nx = p;       p += a.out_w;
  t.ny = p;
  return t;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1), box = the whole crop.
__device__ void bilinear_coeffs(int in_size, int out_size, int ksize, int* bounds, int* kk, int tid, int nthreads) {
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  for (int xx = tid; xx < out_size; xx += nthreads) {
    const double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > ksize) xmax = ksize;        // cannot happen when ksize follows cl4_input_ksize; keeps the table in bounds
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      ww += v < 1.0 ? 1.0 - v : 0.0;
    }
    int* k = kk + (long)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      double w = 0.0;
      if (x < xmax) {
        double v = (x + xmin - center + 0.5) * ss;
        if (v < 0.0) v = -v;
        w = v < 1.0 ? 1.0 - v : 0.0;
        if (ww != 0.0) w /= ww;
      }
      k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

// Geometry.c ImagingScaleAffine: xo = a*0.5, then xo += a per output position; COORD(v) = v < 0 ? -1 : (int)v.
__device__ void nearest_table(int in_size, int out_size, int* tab) {
  const double a = (double)(in_size - 0) / out_size;
  double o = 0.0 + a * 0.5;
  for (int x = 0; x < out_size; ++x) {
    int xin = o < 0.0 ? -1 : (int)o;
    tab[x] = xin < 0 ? 0 : (xin >= in_size ? in_size - 1 : xin);   // Pillow leaves such a pixel unwritten; clamp instead
    o += a;
  }
}

__global__ void input_tables_kernel(InArgs a) {
  const int s = blockIdx.x;
  const cl4_sample_desc d = a.descs[s];
  const Tables t = tables_of(a, s);
  const int which = blockIdx.y;
  if (which == 0) bilinear_coeffs(d.cw, a.out_w, a.ksize, t.bounds_h, t.kk_h, threadIdx.x, blockDim.x);
  if (which == 1) bilinear_coeffs(d.ch, a.out_h, a.ksize, t.bounds_v, t.kk_v, threadIdx.x, blockDim.x);
  if (which == 2 && threadIdx.x == 0) nearest_table(d.cw, a.out_w, t.nx);
  if (which == 3 && threadIdx.x == 0) nearest_table(d.ch, a.out_h, t.ny);
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Horizontal pass: temp[s][y][x][c] for the ch rows of the crop. grid = (x-chunks, rows, samples).
__global__ void input_hpass_kernel(InArgs a) {
  const int s = blockIdx.z;
  const cl4_sample_desc d = a.descs[s];
  const int y = blockIdx.y;
  if (y >= d.ch) return;
  const Tables t = tables_of(a, s);
  const uint8_t* row = a.images + d.img_off + ((long)(d.ci + y) * d.W + d.cj) * 3;
  uint8_t* out = a.temp + (((long)s * a.max_ch + y) * a.out_w) * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.out_w * 3; i += gridDim.x * blockDim.x) {
    const int x = i / 3, c = i - 3 * x;
    const int xmin = t.bounds_h[2 * x], n = t.bounds_h[2 * x + 1];
    const int* k = t.kk_h + (long)x * a.ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    for (int j = 0; j < n; ++j) acc += (int)row[(xmin + j) * 3 + c] * k[j];
    out[i] = (uint8_t)clip8(acc);
  }
}

// Vertical pass + flip + ToTensor + Normalize: out_img[s][c][y][x'] float32.
__global__ void input_vpass_kernel(InArgs a) {
  const int s = blockIdx.z;
  const int y = blockIdx.y;
  const Tables t = tables_of(a, s);
  const int flip = a.descs[s].flip;
  const int ymin = t.bounds_v[2 * y], n = t.bounds_v[2 * y + 1];
  const int* k = t.kk_v + (long)y * a.ksize;
  const uint8_t* tmp = a.temp + ((long)s * a.max_ch + ymin) * a.out_w * 3;
  const long plane = (long)a.out_h * a.out_w;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < a.out_w; x += gridDim.x * blockDim.x) {
    int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int j = 0; j < n; ++j) {
      const uint8_t* p = tmp + ((long)j * a.out_w + x) * 3;
      acc[0] += (int)p[0] * k[j];
      acc[1] += (int)p[1] * k[j];
      acc[2] += (int)p[2] * k[j];
    }
    const int xo = flip ? a.out_w - 1 - x : x;
    float* o = a.out_img + (long)s * 3 * plane + (long)y * a.out_w + xo;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * plane] = ((float)clip8(acc[c]) / 255.0f - a.mean[c]) / a.std[c];
  }
}

// Labels: nearest gather of (seg, mask), LabelTransform LUT, `mask *= seg_map > 0`, flip.
__global__ void input_label_kernel(InArgs a) {
  const int s = blockIdx.z;
  const int y = blockIdx.y;
  const cl4_sample_desc d = a.descs[s];
  const Tables t = tables_of(a, s);
  const uint8_t* row = a.labels + d.lbl_off + ((long)(d.ci + t.ny[y]) * d.W + d.cj) * 2;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < a.out_w; x += gridDim.x * blockDim.x) {
    const uint8_t* p = row + (long)t.nx[x] * 2;
    const int seg = a.lut ? a.lut[p[0]] : p[0];
    const int xo = d.flip ? a.out_w - 1 - x : x;
    const long o = ((long)s * a.out_h + y) * a.out_w + xo;
    a.out_seg[o] = seg;
    if (a.out_mask) a.out_mask[o] = seg > 0 ? p[1] : 0;
  }
}

// ---- label_generation (dataset/utils.py:299-350) -----------------------------------------------------------------------
struct InstStat { unsigned long long count, sum_y, sum_x, first; };   // per (sample, instance id)

__global__ void inst_init_kernel(InstStat* st, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st[i] = InstStat{0ull, 0ull, 0ull, ~0ull};
}

// Per-workgroup LDS aggregation (an instance covers long runs of consecutive pixels, so per-pixel global atomics would all
// hit the same few addresses), then one global atomic per (workgroup, live id).
__global__ void inst_stats_kernel(const uint8_t* mask, InstStat* st, int H, int W) {
  __shared__ unsigned long long cnt[256], sy[256], sx[256], first[256];
  const int s = blockIdx.y;
  const long hw = (long)H * W;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) { cnt[i] = 0; sy[i] = 0; sx[i] = 0; first[i] = ~0ull; }
  __syncthreads();
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int id = mask[s * hw + i];
    if (id == 0 || id == 255) continue;
    atomicAdd(&cnt[id], 1ull);
    atomicAdd(&sy[id], (unsigned long long)(i / W));
    atomicAdd(&sx[id], (unsigned long long)(i % W));
    atomicMin(&first[id], (unsigned long long)i);
  }
  __syncthreads();
  for (int id = threadIdx.x; id < 256; id += blockDim.x) {
    if (cnt[id] == 0) continue;
    InstStat* q = st + (long)s * 256 + id;
    atomicAdd(&q->count, cnt[id]);
    atomicAdd(&q->sum_y, sy[id]);
    atomicAdd(&q->sum_x, sx[id]);
    atomicMin(&q->first, first[id]);
  }
}

// weight = 1 and offset = (float)(mean - (double)(float)coord) on instance pixels (the reference subtracts a float32
// coordinate grid from the float64 mean and stores into a float32 map).
__global__ void inst_paint_kernel(const uint8_t* mask, const InstStat* st, float* offset, float* weight, int H, int W) {
  const int s = blockIdx.y;
  const long hw = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int id = mask[s * hw + i];
    if (id == 0 || id == 255) continue;
    const InstStat q = st[(long)s * 256 + id];
    const double cy = (double)q.sum_y / (double)q.count, cx = (double)q.sum_x / (double)q.count;
    weight[s * hw + i] = 1.f;
    offset[(s * 2 + 0) * hw + i] = (float)(cy - (double)(float)(i / W));
    offset[(s * 2 + 1) * hw + i] = (float)(cx - (double)(float)(i % W));
  }
}

// center_map_gen (dataset/utils.py:217-252): np.maximum paste of the (6 sigma + 3)^2 gaussian at (int(cx), int(cy)) into the
// plane of the instance's class (class of its first pixel in row-major order, minus 1). One workgroup per (sample, id).
__global__ void inst_splat_kernel(const long long* seg, const InstStat* st, const float* gauss, float* center, int H, int W,
                                  int num_classes, int sigma) {
  const int s = blockIdx.y, id = blockIdx.x;
  const InstStat q = st[(long)s * 256 + id];
  if (id == 0 || id == 255 || q.count == 0) return;
  const long hw = (long)H * W;
  const int cls = (int)seg[s * hw + (long)q.first] - 1;
  if (cls < 0 || cls >= num_classes) return;          // the reference asserts cls < num_classes
  const int x = (int)((double)q.sum_x / (double)q.count), y = (int)((double)q.sum_y / (double)q.count);
  const int size = 6 * sigma + 3;
  const int ulx = x - 3 * sigma - 1, uly = y - 3 * sigma - 1;
  float* plane = center + ((long)s * num_classes + cls) * hw;
  for (int i = threadIdx.x; i < size * size; i += blockDim.x) {
    const int gy = i / size, gx = i - gy * size;
    const int Y = uly + gy, X = ulx + gx;
    if (Y < 0 || Y >= H || X < 0 || X >= W) continue;
    atomicMax((int*)&plane[(long)Y * W + X], __float_as_int(gauss[i]));   // non-negative floats order like their bits
  }
}

size_t table_ints(int out_h, int out_w, int ksize) { return (size_t)(out_w + out_h) * (2 + ksize + 1); }
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int cl4_input_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double scale = (double)((float)in_size) / out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil(scale) * 2 + 1;
}

extern "C" size_t cl4_input_workspace_bytes(int b, int out_h, int out_w, int max_ch, int ksize) {
  if (b <= 0 || out_h <= 0 || out_w <= 0 || max_ch <= 0 || ksize <= 0) return 0;
  return align256((size_t)b * sizeof(cl4_sample_desc)) + align256((size_t)b * table_ints(out_h, out_w, ksize) * sizeof(int)) +
         align256((size_t)b * max_ch * out_w * 3);
}

extern "C" int cl4_input_batch(const uint8_t* images, size_t images_bytes, const uint8_t* labels, size_t labels_bytes,
                               const cl4_sample_desc* descs, int b, int out_h, int out_w, const float* mean, const float* std,
                               const uint8_t* lut, float* out_img, long long* out_seg, uint8_t* out_mask, void* workspace,
                               size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(images && descs && out_img && mean && std && b > 0 && out_h > 0 && out_w > 0, "input_batch: bad arguments");
  CL4_REQUIRE((labels != nullptr) == (out_seg != nullptr) && (out_mask == nullptr || labels != nullptr),
              "input_batch: labels and out_seg go together (out_mask is optional)");
  int max_ch = 0, ksize = 0;
  for (int s = 0; s < b; ++s) {
    const cl4_sample_desc& d = descs[s];
    CL4_REQUIRE(d.H > 0 && d.W > 0 && d.ch > 0 && d.cw > 0 && d.ci >= 0 && d.cj >= 0 && d.ci + d.ch <= d.H && d.cj + d.cw <= d.W,
                "input_batch: crop box of sample %d outside its %dx%d image", s, d.H, d.W);
    CL4_REQUIRE(d.img_off >= 0 && (size_t)d.img_off + (size_t)d.H * d.W * 3 <= images_bytes,
                "input_batch: image %d exceeds the packed buffer", s);
    CL4_REQUIRE(!labels || (d.lbl_off >= 0 && (size_t)d.lbl_off + (size_t)d.H * d.W * 2 <= labels_bytes),
                "input_batch: label %d exceeds the packed buffer", s);
    if (d.ch > max_ch) max_ch = d.ch;
    const int kh = cl4_input_ksize(d.cw, out_w), kv = cl4_input_ksize(d.ch, out_h);
    if (kh > ksize) ksize = kh;
    if (kv > ksize) ksize = kv;
  }
  const size_t need = cl4_input_workspace_bytes(b, out_h, out_w, max_ch, ksize);
  CL4_REQUIRE(workspace && workspace_bytes >= need, "input_batch: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  hipStream_t st = cl4::as_stream(stream);
  uint8_t* ws = (uint8_t*)workspace;
  InArgs a;
  a.images = images; a.labels = labels;
  a.descs = (const cl4_sample_desc*)ws;
  ws += align256((size_t)b * sizeof(cl4_sample_desc));
  a.tables = (int*)ws;
  a.table_ints = (int)table_ints(out_h, out_w, ksize);
  ws += align256((size_t)b * a.table_ints * sizeof(int));
  a.temp = ws;
  a.b = b; a.out_h = out_h; a.out_w = out_w; a.max_ch = max_ch; a.ksize = ksize;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = std[c]; }
  a.lut = lut; a.out_img = out_img; a.out_seg = out_seg; a.out_mask = out_mask;
  hipError_t e = hipMemcpyAsync((void*)a.descs, descs, (size_t)b * sizeof(cl4_sample_desc), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(input_tables_kernel, dim3(b, labels ? 4 : 2), dim3(256), 0, st, a);
  const int xb = cl4::cdiv((long)out_w * 3, 256);
  hipLaunchKernelGGL(input_hpass_kernel, dim3(xb, max_ch, b), dim3(256), 0, st, a);
  hipLaunchKernelGGL(input_vpass_kernel, dim3(cl4::cdiv(out_w, 256), out_h, b), dim3(256), 0, st, a);
  if (labels) hipLaunchKernelGGL(input_label_kernel, dim3(cl4::cdiv(out_w, 256), out_h, b), dim3(256), 0, st, a);
  return cl4::launch_status();
}

extern "C" size_t cl4_target_generation_workspace_bytes(int b) { return b > 0 ? (size_t)b * 256 * sizeof(InstStat) : 0; }

extern "C" int cl4_target_generation(const long long* seg_map, const uint8_t* mask, int b, int H, int W, int num_classes, int sigma,
                                     const float* gauss, float* center, float* offset, float* weight, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  CL4_REQUIRE(seg_map && mask && gauss && center && offset && weight && b > 0 && H > 0 && W > 0 && num_classes > 0 && sigma > 0,
              "target_generation: bad arguments");
  CL4_REQUIRE(workspace && workspace_bytes >= cl4_target_generation_workspace_bytes(b), "target_generation: workspace too small");
  hipStream_t st = cl4::as_stream(stream);
  const long hw = (long)H * W;
  InstStat* stat = (InstStat*)workspace;
  hipError_t e = cl4::fill_bytes_async(center, 0, (size_t)b * num_classes * hw * sizeof(float), st);
  if (e == hipSuccess) e = cl4::fill_bytes_async(offset, 0, (size_t)b * 2 * hw * sizeof(float), st);
  if (e == hipSuccess) e = cl4::fill_bytes_async(weight, 0, (size_t)b * hw * sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(inst_init_kernel, dim3(cl4::cdiv((long)b * 256, 256)), dim3(256), 0, st, stat, b * 256);
  const int g = cl4::cdiv(hw, 256 * 4);
  hipLaunchKernelGGL(inst_stats_kernel, dim3(g, b), dim3(256), 0, st, mask, stat, H, W);
  hipLaunchKernelGGL(inst_paint_kernel, dim3(g, b), dim3(256), 0, st, mask, stat, offset, weight, H, W);
  hipLaunchKernelGGL(inst_splat_kernel, dim3(256, b), dim3(256), 0, st, seg_map, stat, gauss, center, H, W, num_classes, sigma);
  return cl4::launch_status();
}
