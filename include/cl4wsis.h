/*
 * cl4wsis.h — C ABI of libcl4wsis_hip.so, the MI355X (gfx950) native kernels
 * behind the CL4WSIS phase-2 training step.
 *
 * Boundary contract:
 *   - plain pointers + sizes only, no torch / C++ types;
 *   - every buffer (inputs, outputs, workspace) is owned by the caller
 *     (PyTorch's caching allocator in practice); the library never allocates,
 *     frees, or keeps a pointer after the call returns;
 *   - tensors are contiguous NCHW fp32 unless a parameter says otherwise;
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*);
 *     no implicit device synchronisation;
 *   - return value: 0 on success, a negative cl4_status on a bad argument,
 *     or the positive hipError_t of a failed launch. Nothing throws.
 *
 * Each entry point names the reference code (file:line under the upstream
 * CL4WSIS tree) whose arithmetic it replaces.
 */
#ifndef CL4WSIS_H
#define CL4WSIS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cl4_status {
  CL4_OK = 0,
  CL4_EINVAL = -1,  /* bad shape / null pointer / unsupported configuration */
  CL4_ENOSPC = -2   /* workspace too small */
} cl4_status;

enum { CL4_ACT_NONE = 0, CL4_ACT_LEAKY = 1, CL4_ACT_RELU = 2 };

int cl4_abi_version(void);
/* Human-readable description of the last CL4_EINVAL on this thread. */
const char* cl4_last_error(void);

/* ------------------------------------------------------------------------
 * Dense convolution as implicit GEMM on the fp32 MFMA pipe.
 *
 * Replaces: nn.Conv2d (+ InPlaceABN eval affine + leaky-ReLU, + residual add)
 *   modules/residual.py:67-97, models/resnet.py:59-63, modules/deeplab.py:54-70,
 *   modules/panoptic_deeplab.py:20-29,41,73,86,107,160, wss/modules.py:327-343,
 *   segmentation_module.py:107-113.
 *
 * y[n, out_coff+co, oy, ox] =
 *     act2( act1( (sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n, in_coff+ci, iy, ix] + pre) * scale[co]
 *                 + shift[co] ) + post )
 * with iy = oy*stride - pad + ky*dil (zero outside the image).
 *
 * `wp` is the weight PRE-PACKED by cl4_pack_conv_weight: [K = cin*kh*kw][cout].
 * scale/shift/pre/post may be NULL (1 / 0 / absent). `pre` and `post` are
 * [n, cout, hout, wout] contiguous tensors (their own channel count = cout).
 * x may be a channel slice of a wider tensor (in_ctot channels, starting at in_coff);
 * likewise y (out_ctot / out_coff) — this is how torch.cat is avoided.
 * ------------------------------------------------------------------------ */
typedef struct cl4_conv_desc {
  int32_t n, cin, hin, win;
  int32_t cout, kh, kw;
  int32_t stride, pad, dil;
  int32_t hout, wout;
  int32_t in_ctot, in_coff;
  int32_t out_ctot, out_coff;
  int32_t act1;
  float   slope1;
  int32_t act2;
  float   slope2;
  int32_t precision; /* 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1: bf16x3 split MFMA */
} cl4_conv_desc;

/* w: [cout, cin, kh, kw] (PyTorch layout) -> wp: [cin*kh*kw, cout]. */
int cl4_pack_conv_weight(const float* w, float* wp, int cout, int cin, int kh, int kw, void* stream);

int cl4_conv2d_fwd(const cl4_conv_desc* d, const float* x, const float* wp,
                   const float* scale, const float* shift,
                   const float* pre, const float* post, float* y, void* stream);

/* Weight gradient: dw[co,ci,ky,kx] += sum_{n,oy,ox} dy[n,co,oy,ox] * x[n,ci,iy,ix]
 * (autograd of nn.Conv2d, train.py:537). dw is in PyTorch layout [cout,cin,kh,kw]
 * and is ACCUMULATED into (caller zeroes it); split-K partial sums are combined
 * with fp32 atomics. `dy` may be a channel slice (out_ctot/out_coff). */
int cl4_conv2d_wgrad(const cl4_conv_desc* d, const float* x, const float* dy, float* dw, void* stream);

/* Fold BN running statistics into scale/shift:
 *   scale = gamma / sqrt(var + eps), shift = beta - mean*scale (+ bias*scale if bias).
 * Replaces the eval branch of InPlaceABN / nn.BatchNorm2d (segmentation_module.py:17-29). */
int cl4_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, float* scale, float* shift, int c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CL4WSIS_H */
