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
 * Dense convolution as implicit GEMM on the MFMA pipes; `precision` selects the arithmetic of the products:
 *   0  exact fp32 (v_mfma_f32_32x32x2_f32),
 *   1  "bf16x3": fp32 operands split hi + lo into bf16, three v_mfma_f32_32x32x16_bf16 products, fp32 accumulation
 *      (~4e-6 relative per layer) — the default of the frozen networks,
 *   2  "bf16x6": hi + mid + lo, six products, fp32-level accuracy — the default of every differentiated convolution.
 * Tensors are fp32 in HBM in every mode of cl4_conv2d_fwd; cl4_conv2d_fwd_planes (further down, descriptor precision 3) is the
 * bf16x3 arithmetic on activations that are kept pre-split ("planes") between two such layers.
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
  int32_t precision; /* 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1: bf16x3; 2: bf16x6 (three bf16 planes, six products:
                      * fp32-level accuracy on the bf16 MFMA pipe); 3: bf16x3 on a planes input (cl4_conv2d_fwd_planes only; also the
                      * key under which cl4_conv2d_fwd_set_plan / _workspace_bytes keep that entry point's launch plans) */
} cl4_conv_desc;

/* w: [cout, cin, kh, kw] (PyTorch layout) -> wp: [cin*kh*kw, cout]. */
int cl4_pack_conv_weight(const float* w, float* wp, int cout, int cin, int kh, int kw, void* stream);
/* precision 1 ("bf16x3"): w -> wq [ceil(K/8)][2 planes: hi, lo][cout][8 x bf16], i.e. ceil(K/8)*2*cout*16 bytes, where
 * hi = bf16(w), lo = bf16(w - hi); when cin % 32 == 0, K runs k = ((ci/32)*kh*kw + ky*kw+kx)*32 + ci%32 (32-channel
 * blocks, all taps of a block adjacent), otherwise in PyTorch's
 * order (k = ci*kh*kw + ky*kw + kx) otherwise. cl4_conv2d_fwd with desc.precision = 1 takes this buffer as `wp` and accumulates
 * hi*hi + hi*lo + lo*hi on the bf16 MFMA pipe in fp32 (~4e-6 relative error per layer; needs cout % 4 == 0, cout >= 64). */
int cl4_pack_conv_weight_bf16x3(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream);
/* precision 2 ("bf16x6"): x = hi + mid + lo (3 x bf16 = 24 mantissa bits); wq [ceil(K/8)][3 planes][cout][8 x bf16] =
 * ceil(K/8)*3*cout*16 bytes, same K order as bf16x3. The kernel issues hh, hm, mh, hl, lh, mm (drops terms below 2^-24). */
int cl4_pack_conv_weight_bf16x6(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream);

/* Bytes of scratch cl4_conv2d_fwd needs for this shape (0 unless the launch plan splits K: small grids on the 33x33
 * maps are split across grid.z into partial slabs [splits][cout][npix] that a second pass reduces in a fixed order). */
size_t cl4_conv2d_fwd_workspace_bytes(const cl4_conv_desc* d);
/* Pin the launch plan of this convolution shape: impl 1 = register-staged kernel, 2 = LDS-DMA kernel (tile 64|128,
 * `splits` K-slices); impl 0 forgets it (back to the built-in heuristic). splits < 0 (bf16x3 / bf16x6 kernels) selects
 * "main + split tail": the whole rounds of output tiles that fill the chip run unsplit with the fused epilogue and only
 * the last partial round is split |splits| ways (splits = -1 with tile 128: the last partial round runs as unsplit 64x64
 * tiles instead — for short-K layers). Results never depend on the plan beyond fp32 summation order. The host autotuner (cl4wsis_amd/ops.py) times the candidates once per shape in warm-up. */
int cl4_conv2d_fwd_set_plan(const cl4_conv_desc* d, int impl, int tile, int splits);
/* Forget every pinned plan (back to the heuristic); the autotuner's choices depend on timing, so tests reset them. */
int cl4_conv2d_fwd_clear_plans(void);
int cl4_conv2d_fwd(const cl4_conv_desc* d, const float* x, const float* wp,
                   const float* scale, const float* shift,
                   const float* pre, const float* post, float* y,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient: dw[co,ci,ky,kx] += sum_{n,oy,ox} dy[n,co,oy,ox] * x[n,ci,iy,ix]
 * (autograd of nn.Conv2d, train.py:537). dw is in PyTorch layout [cout,cin,kh,kw]
 * and is ACCUMULATED into (caller zeroes it); split-K partial sums are combined
 * with fp32 atomics. `dy` may be a channel slice (out_ctot/out_coff). desc.precision selects the arithmetic: 0 exact
 * fp32 MFMA (step 0 / phase 1: the whole network trains), 1 bf16x3 split products (phase 2's trainable branch). */
int cl4_conv2d_wgrad(const cl4_conv_desc* d, const float* x, const float* dy, float* dw, void* stream);

/* The same weight gradient (autograd of nn.Conv2d, train.py:537) for 3x3 stride-1 layers as ONE K-major GEMM on the LDS-DMA planes kernel
 * (csrc/wgrad_gemm.hip, r03): dW[co][ci][tap] = sum_q dy[co][q - s(tap)] * x[ci][q] over the input pixels q. Three calls:
 *   cl4_kpack           repack x (ks = 1: rows = cin) or dy (ks = 3: rows = 9 * cout, shifted per tap, zero outside dy) into "k-planes":
 *                       chunks of 8 consecutive pixels, bf16 hi / lo, ((q / 8) * 2 + plane) * rows + row; q runs over n x hq x wq
 *                       (the INPUT map), padded with zeros to a multiple of 32; cl4_kpack_bytes(rows, n * hq * wq) bytes
 *   cl4_gemm_kplanes    Y[m][p] = sum_k A[m][k] * B[p][k] (fp32 [M][N]), bf16x3 products (lo*hi + hi*lo + hi*hi), no atomics
 *   cl4_wgrad_scatter   dw[co][ci][tap] += Y[ci][tap * cout + co]
 * x's pack depends on x only: a caller with several layers on one input (the ASPP branches) packs it once. */
size_t cl4_kpack_bytes(int rows, int kq);
int cl4_kpack(const float* src, void* dst, int n, int ctot, int coff, int c, int hs, int ws, int hq, int wq, int ks, int dil, int pad,
              void* stream);
int cl4_gemm_kplanes(const void* a_kplanes, const void* b_kplanes, float* y, int M, int N, int kpad, void* stream);
int cl4_wgrad_scatter(const float* y, float* dw, int cin, int cout, int taps, void* stream);

/* Fold BN running statistics into scale/shift:
 *   scale = gamma / sqrt(var + eps), shift = beta - mean*scale (+ bias*scale if bias).
 * Replaces the eval branch of InPlaceABN / nn.BatchNorm2d (segmentation_module.py:17-29). */
int cl4_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, float* scale, float* shift, int c, void* stream);

/* ------------------------------------------------------------------------
 * Spatial / streaming kernels (HBM-bound). "planes" = n*c of an NCHW tensor.
 * ------------------------------------------------------------------------ */
/* The ResNet stem in eval mode, ONE kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64, no bias) -> InPlaceABN with running statistics
 * folded to scale / shift + leaky-ReLU(slope) -> max-pool 3x3 / stride 2 / pad 1 — models/resnet.py:59-63 (mod1.conv1, mod1.bn1,
 * pool1) as the frozen networks of phase 2 run it. bf16x3 arithmetic like cl4_conv2d_fwd precision 1. `wq`: the weight
 * [64][3][7][7] packed by cl4_pack_stem_weight (cl4_stem_weight_bytes() bytes). Outputs: `y` fp32 [n][64][hp][wp] and / or `ys` the
 * same tensor as planes (either may be NULL), hp = ((h - 1) / 2) / 2 + 1 for odd h (129 for 513). */
size_t cl4_stem_weight_bytes(void);
int cl4_pack_stem_weight(const float* w, void* wq, void* stream);
int cl4_stem_fused(const float* x, const void* wq, const float* scale, const float* shift, float slope, float* y, void* ys,
                   int n, int h, int w, void* stream);
/* nn.MaxPool2d(3, stride=2, padding=1) — models/resnet.py:63. */
int cl4_maxpool3x3s2(const float* x, float* y, long planes, int hin, int win, int hout, int wout, void* stream);
/* F.avg_pool2d(x, (kh,kw), stride=1) -> [planes, h-kh+1, w-kw+1] — modules/deeplab.py:87. */
int cl4_avgpool_s1(const float* x, float* y, long planes, int h, int w, int kh, int kw, void* stream);
/* F.pad(mode="replicate") — modules/deeplab.py:88 (pad_b/pad_r implied by hout/wout). */
int cl4_replicate_pad(const float* x, float* y, long planes, int hin, int win, int pad_t, int pad_l, int hout, int wout, void* stream);
/* mean over H*W per plane — AdaptiveAvgPool2d(1), modules/panoptic_deeplab.py:84; deeplab.py:75-76. */
int cl4_global_mean(const float* x, float* y, long planes, int hw, void* stream);
/* sum over H*W per plane, fp64 partial sums — the reduction of `x.repeat(1, 1, h, w)`'s backward in the image-pooling branch
   (modules/panoptic_deeplab.py:84-89 through autograd), taken on the projection's pre-activation gradient before the 1x1 input-gradient
   product instead of on its 1280-channel result (same sum, 256 instead of 256 x 1089 rounded terms per output). */
int cl4_plane_sum(const float* x, float* y, long planes, int hw, void* stream);
/* F.interpolate(mode="bilinear") in both align_corners conventions; x and y may be
 * channel slices. Backward is a deterministic gather (no atomics); accumulate!=0 adds into dx. */
int cl4_resize_bilinear_fwd(const float* x, float* y, int n, int c, int hin, int win, int hout, int wout,
                            int align_corners, int in_ctot, int in_coff, int out_ctot, int out_coff, void* stream);
int cl4_resize_bilinear_bwd(const float* dy, float* dx, int n, int c, int hin, int win, int hout, int wout,
                            int align_corners, int in_ctot, int in_coff, int out_ctot, int out_coff,
                            int accumulate, void* stream);
/* y[r, w-1-i] = x[r, i] — torch.flip(-1), train.py:261,263. */
int cl4_flip_w(const float* x, float* y, long rows, int w, void* stream);
int cl4_axpy(float* y, const float* x, float alpha, long n, void* stream);
int cl4_copy_channels(const float* x, float* y, int n, int c, long hw, int in_ctot, int in_coff, int out_ctot, int out_coff, void* stream);

/* ------------------------------------------------------------------------
 * Train-mode nn.BatchNorm2d (+ReLU, + nn.Dropout keep mask) of the instance decoder / heads —
 * modules/panoptic_deeplab.py:26-28,44-46,74-76,108-110,121-123,161-163.
 * ------------------------------------------------------------------------ */
size_t cl4_bn_workspace_bytes(int c);
/* batch mean / biased variance of a channel slice; running stats (may be NULL) are updated in place
 * with the unbiased variance and `momentum` (PyTorch semantics). */
int cl4_bn_train_stats(const float* x, int n, int c, int hw, int ctot, int coff, float* mean, float* var,
                       float* running_mean, float* running_var, float momentum,
                       void* workspace, size_t workspace_bytes, void* stream);
/* cl4_bn_train_stats + cl4_bn_apply in two launches instead of three (r03): the statistics are finalised inside the elementwise
 * pass; mean / var are outputs (the backward pass reads them), running stats (may be NULL) are updated in place. Same module:
 * train-mode nn.BatchNorm2d + ReLU (+ Dropout) of modules/panoptic_deeplab.py:26-28,44-46,74-76. */
int cl4_bn_train_fwd(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                     float* mean, float* var, float* running_mean, float* running_var, float momentum,
                     const float* gamma, const float* beta, float eps, int act, float slope, const uint8_t* keep,
                     float keep_scale, void* workspace, size_t workspace_bytes, void* stream);
/* y = dropout(act((x-mean)/sqrt(var+eps)*gamma+beta)); keep: uint8 [n,c,hw] or NULL. */
int cl4_bn_apply(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                 const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                 int act, float slope, const uint8_t* keep, float keep_scale, void* stream);
/* Backward of the above: dx (same layout as x); dgamma/dbeta (may be NULL) are accumulated into.
 * workspace_bytes >= cl4_bn_workspace_bytes(c) + 2*c*sizeof(float). */
int cl4_bn_train_bwd(const float* x, const float* dy, float* dx, int n, int c, int hw, int x_ctot, int x_coff,
                     int dy_ctot, int dy_coff, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, int act, const uint8_t* keep, float keep_scale,
                     float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int cl4_relu_bwd(const float* y, const float* dy, float* dx, long n, void* stream);
/* int32 uniform draws in [0, modulus) (counter-based, like the Dropout masks): the random old classes of RandropLoss
 * (utils/wss_loss.py:237; the reference draws them with the CPU generator, so the stream itself cannot be matched). */
int cl4_random_labels(int* out, long n, uint64_t seed, int modulus, void* stream);
/* ... from a seed in device memory (uint64 [1]) that the launch advances by `advance`: a HIP-graph replay of the phase-1 step draws fresh
 * labels every time (utils/wss_loss.py:237 draws them per call). */
int cl4_random_labels_dev(int* out, long n, uint64_t* seed_dev, uint64_t advance, int modulus, void* stream);
/* Two-phase forms for cross-rank batch statistics — InPlaceABNSync in training mode (segmentation_module.py:17-29;
 * SURVEY.md §8 A2): BN (eps 1e-5, momentum 0.1) + leaky-ReLU(slope) / identity, statistics over ALL ranks.
 *   forward : cl4_bn_sync_stats_local -> host all-reduce(SUM) of sums[c*2] doubles and of the element count ->
 *             cl4_bn_sync_stats_final -> cl4_bn_apply;
 *   backward: cl4_bn_sync_bwd_local (local {sum dz, sum dz*xhat}; accumulates this rank's dgamma / dbeta) ->
 *             all-reduce(SUM) -> cl4_bn_sync_bwd_apply with the global count.
 * With one rank the pair reproduces cl4_bn_train_stats / cl4_bn_train_bwd (and adds the leaky-ReLU derivative).
 * gamma_sign_src (nullable; InPlaceABN, whose kernels run with gamma = |weight| + eps — cl4_abs_eps): the layer's raw weight; dgamma then
 * receives d(weight) = (weight > 0 ? 1 : -1) * d(gamma) directly (what cl4_sign_axpy does as a separate pass). */
int cl4_bn_sync_stats_local(const float* x, int n, int c, int hw, int ctot, int coff, double* sums, void* workspace,
                            size_t workspace_bytes, void* stream);
int cl4_bn_sync_stats_final(const double* sums, double count, float* mean, float* var, float* running_mean, float* running_var,
                            float momentum, int c, void* stream);
int cl4_bn_sync_bwd_local(const float* x, const float* dy, int n, int c, int hw, int x_ctot, int x_coff, int dy_ctot, int dy_coff,
                          const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act,
                          float slope, const uint8_t* keep, float keep_scale, double* sums, float* dgamma, float* dbeta,
                          const float* gamma_sign_src, void* workspace, size_t workspace_bytes, void* stream);
int cl4_bn_sync_bwd_apply(const float* x, const float* dy, float* dx, int n, int c, int hw, int x_ctot, int x_coff, int dy_ctot,
                          int dy_coff, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                          int act, float slope, const uint8_t* keep, float keep_scale, const double* sums, double count,
                          void* workspace, size_t workspace_bytes, void* stream);
/* r05: the backward pair that also hands the input gradient to the NEXT convolution in its operand format (the fp16x3 input-gradient
 * convolution of the layer in front — modules/residual.py:84-97 differentiated — reads fp16 planes scaled by a power of two): _local_max also
 * leaves max |dz| and max |xhat| per (channel, reduction workgroup) in `maxima` (cl4_bn_bwd_maxima_floats(c) floats), _apply_planes derives the
 * scale from the bound |dx| <= |gamma| rstd (max |dz| + |mean dz| + max |xhat| |mean dz xhat|), writes it to slot[0], and writes dx (fp32,
 * bit-identical to cl4_bn_sync_bwd_apply) plus its planes twin `dxs` (cl4_planes_bytes(n, c, hw, 2) bytes). x / dx whole tensors, c % 8 == 0. */
size_t cl4_bn_bwd_maxima_floats(int c);
int cl4_bn_sync_bwd_local_max(const float* x, const float* dy, int n, int c, int hw, int dy_ctot, int dy_coff, const float* mean,
                              const float* var, const float* gamma, const float* beta, float eps, int act, float slope, double* sums,
                              float* dgamma, float* dbeta, const float* gamma_sign_src, float* maxima, void* workspace, size_t workspace_bytes,
                              void* stream);
int cl4_bn_sync_bwd_apply_planes(const float* x, const float* dy, float* dx, void* dxs, float* slot, int n, int c, int hw, int dy_ctot,
                                 int dy_coff, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act,
                                 float slope, const double* sums, double count, const float* maxima, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* The backward of a train-mode InPlaceABN on ONE rank (no exchange between reduction and elementwise pass: N = 1 runs of train.py:272-537)
 * in two launches — reduction, then dx with the channel sums folded in place and dgamma / dbeta accumulated (sign rule: gamma_sign_src) —
 * or three with the planes twin (dxs + slot non-null: see cl4_bn_sync_bwd_apply_planes). Results equal the two-phase pair's at world size 1. */
size_t cl4_bn_train_bwd_act_workspace_bytes(int c);
int cl4_bn_train_bwd_act(const float* x, const float* dy, float* dx, void* dxs, float* slot, int n, int c, int hw, int dy_ctot, int dy_coff,
                         const float* mean, const float* var, const float* gamma, const float* beta, float eps, int act, float slope,
                         float* dgamma, float* dbeta, const float* gamma_sign_src, void* workspace, size_t workspace_bytes, void* stream);
/* db[c] += sum_{n,hw} dy — bias gradient of the classifier convs. */
int cl4_bias_grad(const float* dy, float* db, int n, int c, int hw, int ctot, int coff, void* stream);
/* Counter-based keep mask for nn.Dropout(p) — modules/panoptic_deeplab.py:123. */
int cl4_dropout_mask(uint8_t* keep, long n, uint64_t seed, float p, void* stream);
/* Same, with the seed read from device memory and advanced by `advance` afterwards: a captured HIP graph of the step
 * then draws a fresh mask on every replay. */
int cl4_dropout_mask_dev(uint8_t* keep, long n, uint64_t* seed_dev, uint64_t advance, float p, void* stream);

/* ------------------------------------------------------------------------
 * 5x5 depthwise conv, stride 1, pad 2 — modules/panoptic_deeplab.py:39-40. w: [c,1,5,5].
 * flip=1 correlates with the rotated kernel (= input gradient). wgrad accumulates into dw.
 * ------------------------------------------------------------------------ */
int cl4_dwconv5_fwd(const float* x, const float* w, float* y, int n, int c, int h, int wd,
                    int x_ctot, int x_coff, int y_ctot, int y_coff, int flip, void* stream);
/* The same forward (flip = 0) that also writes the per-channel partial sums {sum y, sum y^2} of its output (fp64 pairs,
 * stats[(c * nparts + i) * 2 + {0, 1}], nparts = cl4_dwconv5_stats_parts(n, h, w)), and the train-mode BatchNorm forward that starts
 * from such partials (r03): the BatchNorm behind a depthwise convolution (modules/panoptic_deeplab.py:26-28,44-46) reads the
 * activation once instead of twice. */
int cl4_dwconv5_stats_parts(int n, int h, int w);
int cl4_dwconv5_fwd_stats(const float* x, const float* w, float* y, int n, int c, int h, int wd, int x_ctot, int x_coff,
                          int y_ctot, int y_coff, double* stats, void* stream);
int cl4_bn_train_fwd_partials(const float* x, float* y, int n, int c, int hw, int x_ctot, int x_coff, int y_ctot, int y_coff,
                              float* mean, float* var, float* running_mean, float* running_var, float momentum,
                              const float* gamma, const float* beta, float eps, int act, float slope, const uint8_t* keep,
                              float keep_scale, const double* partial, int nparts, void* stream);
int cl4_dwconv5_wgrad(const float* x, const float* dy, float* dw, int n, int c, int h, int wd,
                      int x_ctot, int x_coff, int dy_ctot, int dy_coff, void* stream);

/* ------------------------------------------------------------------------
 * Label stage of the phase-2 step (train.py:418-522). HBM/LDS-bound integer / float work.
 * ------------------------------------------------------------------------ */
/* PeakGenerator eval (PAM -> 1x1 conv -> relu*label -> /max) + 3x3 smoothing — wss/modules.py:354-368,390-434,
 * wss/utils.py:28-32. int_masks [b,ctot,h,w] (the last cnew channels are used), pg_w [cnew,cnew], pg_b [cnew],
 * label_new [b,cnew] -> cam [b,cnew,h,w] (new-class channels only; old-class channels are identically zero). */
size_t cl4_cam_smooth_workspace_bytes(int b, int cnew, int h, int w);   /* 0 when the maps fit LDS (VOC: 5 x 33x33) */
int cl4_cam_smooth(const float* int_masks, int b, int ctot, int h, int w, const float* pg_w, const float* pg_b,
                   const float* label_new, int cnew, float alpha, float* cam, void* workspace, size_t workspace_bytes,
                   void* stream);
/* PeakGenerator.forward in eval mode (wss/modules.py:390-434) WITHOUT the caller's smoothing: cam [b,cnew,h,w] =
 * relu(conv(PAM(x))) * label / (max_hw + 1e-5) — what `_, cam = peakgenerator(int_masks, l1h)` returns for the new classes
 * (train.py:427). cl4_smoothing3x3 is `smoothing` (wss/utils.py:28-32; train.py:428): 3x3 mean, zero pad, divisor 9. */
size_t cl4_cam_normalize_workspace_bytes(int b, int cnew, int h, int w);
int cl4_cam_normalize(const float* int_masks, int b, int ctot, int h, int w, const float* pg_w, const float* pg_b,
                      const float* label_new, int cnew, float alpha, float* cam, void* workspace, size_t workspace_bytes,
                      void* stream);
int cl4_smoothing3x3(const float* x, long planes, int h, int w, float* out, void* stream);
/* Bilinear upsample (align_corners=False) to HxW fused with peak_extract (wss/utils.py:3-25): kernel x kernel max-pool
 * NMS + top-K per (b,c). conf/ys/xs: [b,c,K]; slots beyond the positive peaks hold conf=0,y=x=0. Ties: lowest index. */
size_t cl4_peak_workspace_bytes(int b, int c, int H, int W, int K);
int cl4_peak_extract(const float* cam, const float* label_new, int b, int c, int h, int w, int H, int W, int kernel, int K,
                     float* conf, int* ys, int* xs, void* workspace, size_t workspace_bytes, void* stream);
/* train.py:263,438-449: flip-TTA mean of seg2b [2b,C,h,w] -> upsample -> softmax -> label mask -> argmax.
 * seg_gt uint8 [b,H,W] (old classes zeroed), old_w f32 [b,H,W], prob_win f32 [b,H,W] (masked prob of the argmax). */
int cl4_seg_pseudo_gt(const float* seg2b, const float* l1h, int bs, int C, int h, int w, int H, int W, int old_classes,
                      int no_mask, uint8_t* seg_gt, float* old_w, float* prob_win, void* stream);
/* Connected components of equal non-zero values (connectivity 4 or 8): labels int32 [b,H,W] = smallest flat index
 * (b*H*W + y*W + x) of the component, -1 on background. Replaces cv2.connectedComponentsWithStats
 * (modules/utils.py:224,307,627); cl4_cc_stats gives area / sum_x / sum_y at the root index. */
int cl4_ccl(const uint8_t* cls_map, int b, int H, int W, int connectivity, int* labels, void* stream);
int cl4_cc_stats(const int* labels, int b, int H, int W, unsigned* area, unsigned* sumx, unsigned* sumy, void* stream);

/* pseudo_label_generation (modules/utils.py:179-253) + refine_label_generation (:257-385) for the whole batch. */
typedef struct cl4_label_io {
  int32_t b, H, W;
  int32_t cnew, old_classes;      /* new classes; old classes INCLUDING background */
  int32_t hc, wc, center_ctot;    /* low-res size / channel count of center_lo */
  int32_t K, sigma, nms_kernel, run_refine;
  float pseudo_thresh, refine_thresh, beta;
  const uint8_t* seg_gt;          /* [b,H,W] */
  const float* prob_win;          /* [b,H,W] */
  const float* label_new;         /* [b,cnew] */
  const float* peak_conf;         /* [b,cnew,K] */
  const int32_t* peak_y;
  const int32_t* peak_x;
  const float* center_lo;         /* [b,center_ctot,hc,wc] new model; class c is channel old_classes-1+c */
  const float* offset_lo;         /* [b,2,hc,wc] */
  const float* gauss;             /* [(6*sigma+3)^2] float32 of modules/utils.py:49-59 */
  float* p_center;                /* [b,cnew,H,W] */
  float* p_offset;                /* [b,2,H,W] (channel 0 = y) */
  float* p_weight;                /* [b,1,H,W] */
  float* r_center;
  float* r_offset;
  float* r_weight;
} cl4_label_io;
size_t cl4_label_workspace_bytes(int b, int H, int W);
int cl4_label_generation(const cl4_label_io* d, void* workspace, size_t workspace_bytes, void* stream);
/* {n_contour, n_psplat, n_center, n_cand, overflow} of the last call on this workspace -> out5 (device). */
int cl4_label_counters(const void* workspace, int b, int H, int W, int* out5, void* stream);
/* Sticky form for the training loop (train.py:535 is the reference's only per-interval host check): acc5[0..3] = running
   max of the four counts, acc5[4] |= overflow. Enqueue-only, graph-capturable. */
int cl4_label_counters_accumulate(const void* workspace, int b, int H, int W, int* acc5, void* stream);

/* train.py:503-522 with utils/loss.py:228-275: blending + four weighted losses, fused with the bilinear upsample of
 * the low-resolution maps; optionally the gradient w.r.t. center / offset (low resolution). */
typedef struct cl4_p2_loss_io {
  int32_t b, H, W, hc, wc;
  int32_t cold, cnew;             /* old classes WITHOUT background (15), new classes (5) */
  const float* center;            /* [b,cold+cnew,hc,wc] */
  const float* offset;            /* [b,2,hc,wc] */
  const float* old_center;        /* [b,cold,hc,wc] */
  const float* old_offset;        /* [b,2,hc,wc] */
  const float* old_w;             /* [b,H,W] */
  const float* p_center; const float* p_offset; const float* p_weight;
  const float* r_center; const float* r_offset; const float* r_weight;   /* all NULL when run_refine is off */
  float center_weight, offset_weight;   /* 200, 0.01 (train.py:100-101) */
  float* losses;                  /* [4] device: center_old, offset_old, center_new, offset_new */
  float* d_center;                /* [b,cold+cnew,hc,wc] d(sum of the four)/d(center), or NULL */
  float* d_offset;
} cl4_p2_loss_io;
size_t cl4_p2_loss_workspace_bytes(int b, int H, int W, int cold, int cnew, int with_grad);
int cl4_p2_loss(const cl4_p2_loss_io* d, void* workspace, size_t workspace_bytes, void* stream);

/* torch.optim.Adam step (train.py:181-182,538) over flat buffers; `step` is the 1-based step count. */
int cl4_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, void* stream);
/* Graph-replayable form: learning rate and 0-based step counter in device memory (bias corrections computed on the
 * device); the counter is advanced by `advance` after the update. */
int cl4_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float beta1, float beta2,
                      float eps, float weight_decay, int* step_dev, int advance, void* stream);
/* torch.optim.SGD(momentum=0.9, nesterov=True) step over a flat segment (train.py:179-180, phase 1 / step 0). `buf` is the
 * momentum buffer (zero-initialised); lr_dev (device float, or NULL to use `lr`) makes the launch graph-replayable. */
int cl4_sgd_step(float* p, const float* g, float* buf, long n, float lr, const float* lr_dev, float momentum,
                 float weight_decay, int nesterov, void* stream);
int cl4_fill_zero(void* p, size_t bytes, void* stream);
/* Measurement aid: a one-thread kernel that writes the device's 100 MHz wall clock (10 ns ticks, uint64) into *slot when `stream`
 * reaches it; capturable like any kernel (per-chain time stamps inside the captured step, tools/chain_gantt.py). */
int cl4_stamp(void* slot, void* stream);

/* ------------------------------------------------------------------------
 * Weakly-supervised loss stack of phase 1 / step 0. Class maps are [b,c,h,w] fp32 at the
 * network stride (33x33 at crop 513); `labels` are the image-level one-hot labels [b,c-1] (l1h, no background).
 * ------------------------------------------------------------------------ */
/* PAMR.forward — wss/modules.py:133-152 with LocalAffinityAbs / LocalStDev / LocalAffinityCopy :47-119 (call site
 * train.py:378-381, num_iter=10, dilations [1,2,4,8,12]). img [b,3,h,w] is the denormalised image already resized to
 * the mask size (cl4_resize_bilinear_fwd, align_corners=1); mask/out [b,c,h,w], out != mask. */
size_t cl4_pamr_workspace_bytes(int b, int c, int h, int w, int ndil);
int cl4_pamr(const float* img, const float* mask, float* out, int b, int c, int h, int w, int num_iter,
             const int* dilations /* host */, int ndil, void* workspace, size_t workspace_bytes, void* stream);
/* bce_loss(outputs, labels, mode='ngwp', reduction) = BCE-with-logits(ngwp_focal(outputs)[:, -nl:], labels) —
 * utils/wss_loss.py:7-22,36-53 (train.py:340-342). labels [b,nl]; reduction_sum=0: 'mean', 1: 'sum' (sum over
 * classes, mean over b). Writes the scalar loss, optionally the nGWP logits [b,c] and d(loss)/d(x) [b,c,h,w]. */
size_t cl4_ngwp_workspace_bytes(int b, int c);
int cl4_ngwp_focal_bce(const float* x, const float* labels, int b, int c, int h, int w, int nl, int reduction_sum, float lam,
                       float* ngwp /* or NULL */, float* loss, float* dx /* or NULL */, void* workspace,
                       size_t workspace_bytes, void* stream);
/* pseudo_gtmask(mask * labels, ambiguous=True, cutoffs) — wss/single_stage.py:18-40 with the label masking of
 * train.py:381-382 fused (labels NULL = no masking). out [b,c,h,w] of 0/1; thresholds [b,c] scratch/out. */
int cl4_pseudo_gtmask(const float* mask, const float* labels, float* out, int b, int c, int h, int w, float cutoff_top,
                      float cutoff_bkg, float cutoff_low, float* thresholds, void* stream);
/* softmax over the channel axis (train.py:372-373). */
int cl4_softmax_channels(const float* x, float* y, int b, int c, int h, int w, void* stream);
/* binarize(x * labels) — utils/wss_loss.py:188-190 (train.py:387). */
int cl4_binarize(const float* x, const float* labels, float* out, int b, int c, int h, int w, void* stream);
/* balanced_mask_loss_ce(mask, pseudo_gt, gt_labels) — wss/single_stage.py:43-80 (train.py:411); mask and pseudo_gt
 * have the same spatial size. Scalar loss and optionally d(loss)/d(mask). */
size_t cl4_balanced_ce_workspace_bytes(int b, int c, int h, int w);
int cl4_balanced_mask_loss_ce(const float* mask, const float* pseudo_gt, const float* labels, int b, int c, int h, int w,
                              float* loss, float* dmask /* or NULL */, void* workspace, size_t workspace_bytes, void* stream);
/* BCEWithLogitsLossWithIgnoreIndex(reduction='none')(x, target).mean() — utils/loss.py:62-85 with train.py:285-288.
 * x [b,c,h,w], target int64 [b,h,w]; labels equal to ignore_index (or outside [0,c), where the reference raises)
 * contribute nothing. Scalar loss and optionally d(loss)/d(x). */
size_t cl4_bce_ignore_workspace_bytes(void);
int cl4_bce_ignore_mean(const float* x, const long long* target, int b, int c, int h, int w, int ignore_index, float* loss,
                        float* dx /* or NULL */, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Phase-1 training-path losses: value + gradient on the stride-16 class maps.
 * One workspace size serves all five entry points.
 * ------------------------------------------------------------------------ */
size_t cl4_p1loss_workspace_bytes(int b, int tot, int h, int w);
/* PeakGenerator.forward (training: PAM -> 1x1 conv nc->nc -> nGWP logit, wss/modules.py:354-423) + the peak loss
 * BCE-with-logits(logit, labels) of train.py:334-336,344-345. x [b,ctot,h,w] (the generator reads the last nc channels),
 * weight [nc,nc], bias [nc], labels [b,nc]; writes logit [b,nc], loss [1] and, when dx != NULL, dx [b,ctot,h,w]
 * (old-class channels zero) and ACCUMULATES into dweight [nc,nc] / dbias [nc]. 1 <= nc <= 24. */
int cl4_peakgen_train_loss(const float* x, const float* weight, const float* bias, const float* labels, int b, int ctot, int h,
                           int w, int nc, float alpha, float* logit, float* loss, float* dx, float* dweight, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream);
/* The same generator as a differentiable map for foreign training loops (`peak_logits, _ = peakgenerator(int_masks_raw)`,
 * train.py:334-336, followed by any loss): dlogit == NULL -> forward only (logit [b,nc]); otherwise the vector-Jacobian
 * product for the upstream gradient dlogit [b,nc]: dx [b,ctot,h,w] written, dweight / dbias accumulated. */
int cl4_peakgen_vjp(const float* x, const float* weight, const float* bias, const float* dlogit, int b, int ctot, int h, int w,
                    int nc, float alpha, float* logit, float* dx, float* dweight, float* dbias, void* workspace,
                    size_t workspace_bytes, void* stream);
/* FLAC consistency loss — train.py:314-333. A [3*bs,c,h,h] = PseudoLabeler logits on [x, flip_w(x), rot90(x, angle/90)];
 * angle in {90,180,270}; dA [3*bs,c,h,h] or NULL. */
int cl4_flac_loss(const float* A, int bs, int c, int old_classes, int h, int w, int angle, float* loss, float* dA, void* workspace,
                  size_t workspace_bytes, void* stream);
/* l_loc — train.py:346-354: mean BCE-with-logits(x[:, :cold], sigmoid(old_seg) [* labels with --no_mask]); x [b,ctot,h,w],
 * old_seg [b,cold,h,w], labels [b,ctot-1] or NULL; dx [b,ctot,h,w] (only the first cold channels are written) or NULL. */
int cl4_lloc_loss(const float* x, const float* old_seg, const float* labels, int b, int ctot, int cold, int h, int w, float* loss,
                  float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* nn.MSELoss (lde, train.py:362-363; also the FLAC / distillation building block): loss = mean((x-y)^2),
 * dx = grad_scale * 2 (x - y) / n or NULL. */
int cl4_mse_loss(const float* x, const float* y, long n, float grad_scale, float* loss, float* dx, void* workspace,
                 size_t workspace_bytes, void* stream);
/* l_seg — train.py:374-409: BCE of the new model's seg logits against [sigmoid(old seg) | alpha*binarize + (1-alpha)*soft]
 * with the background rule (icarl_bkg == -1: min), gated per image by the label-set match and normalised by the number of
 * gated images. seg/masks [b,tot,h,w], old_seg [b,cold,h,w], labels [b,tot-1]; dseg [b,tot,h,w] or NULL. */
/* RandropLoss.forward — utils/wss_loss.py:227-248 (train.py:413-416). x [b,tot,h,w] logits, masks [b,tot,h,w] the detached
 * PseudoLabeler logits (reference = sigmoid(masks) * labels), labels [b,tot-1], neg_labels int32 [b,h,w] the randomly
 * drawn old classes in [0,cold) (the caller draws them; the reference uses the CPU generator), mask_labels = labels with
 * --no_mask else NULL. loss = mean over the selected pixels of -log sigmoid(x[neg]) (0 when none); dx or NULL. */
int cl4_randrop_loss(const float* x, const float* masks, const float* labels, const int* neg_labels, const float* mask_labels,
                     int b, int tot, int cold, int h, int w, float* loss, float* dx, void* workspace, size_t workspace_bytes,
                     void* stream);
int cl4_p1_seg_loss(const float* seg, const float* old_seg, const float* masks, const float* labels, int b, int tot, int cold,
                    int h, int w, float alpha, float icarl_bkg, float weight, float* loss, float* dseg, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Building blocks of the train-mode backward of the body / heads (phase 1 / step 0).
 * ------------------------------------------------------------------------ */
/* y[.., s*i, s*j] = x[.., i, j], zero elsewhere: spreads the output gradient of a strided conv onto the stride-1 grid so
 * that its input gradient is cl4_conv2d_fwd with the cl4_transpose_flip_weight weights (models/resnet.py strided convs). */
int cl4_zero_insert(const float* x, float* y, long planes, int h, int w, int stride, int H, int W, void* stream);
/* nn.MaxPool2d(3, stride=2, padding=1) backward (models/resnet.py:63): deterministic gather, first maximum wins. */
int cl4_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, long planes, int hin, int win, int hout, int wout, void* stream);
/* y = act(a + b) — the residual add + in-place activation of modules/residual.py:91-93; dx = dy * act'(.) from the output. */
int cl4_add_act(const float* a, const float* b, float* y, long n, int act, float slope, void* stream);
int cl4_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float slope, void* stream);
/* y[plane][:] += scale * v[plane] — broadcast add of DeepLabV3's image-pooling branch in training mode
 * (modules/deeplab.py:64-67) and, with scale = 1/hw, the backward of the global mean. */
int cl4_add_per_plane(float* y, const float* v, long planes, int hw, float scale, void* stream);
/* nn.Dropout2d in training mode (models/wide_resnet.py:150-155: p = 0.3 in mod6, 0.5 in mod7 of WiderResNet-38): y = keep[plane] ?
 * x / (1 - p) : 0 for every (image, channel) plane; `keep` holds one byte per plane (cl4_dropout_mask draws it). The backward pass is
 * the same call on the gradient. */
int cl4_dropout2d(const float* x, const unsigned char* keep, float* y, long planes, int hw, float p, void* stream);
/* torch.rot90(x, k, dims=(-2,-1)) on square planes = torchvision `rotate` by k*90 degrees (train.py:317,324,330; FLAC). */
int cl4_rot90(const float* x, float* y, long planes, int h, int w, int k, void* stream);
/* y[n,c,:] = x[n,c,:] * scale[c] + shift[c] — `denorm` (utils/utils.py:23-40, train.py:376). */
int cl4_channel_affine(const float* x, float* y, int n, int c, int hw, const float* scale, const float* shift, void* stream);
/* w [cout,cin,k,k] -> wt [cin,cout,k,k] with both spatial axes reversed (the weight of the input-gradient convolution). */
int cl4_transpose_flip_weight(const float* w, float* wt, int cout, int cin, int k, void* stream);

/* ------------------------------------------------------------------------
 * Validation: softmax + arg-max of the full-resolution semantic logits ([nb,c,H,W]; flip != 0: nb == 2 =
 * [image, flipped image], averaged like train.py:709-711) and the confusion-matrix update of
 * metrics/stream_metrics.py:72-79 into hist [n_classes*n_classes] uint64 (accumulated). labels int64 [nimg,H,W]; labels
 * outside [0,n_classes) are ignored; labels in [0,zero_below) count as 0 (eval_pseudolabeler, train.py:737-738). pred (u8
 * [nimg,H,W]) or NULL receives the prediction.
 * ------------------------------------------------------------------------ */
int cl4_semseg_hist(const float* logits, const long long* labels, unsigned long long* hist, unsigned char* pred, int nb, int c, int H,
                    int W, int flip, int n_classes, int zero_below, void* stream);

/* Validation post-processing — get_ins_map (dataset/utils.py:795-892, called from Trainer.validate train.py:622-626):
 * cl4_val_seg: per-sample softmax of the full-resolution logits [nb,c,H,W] (nb = 2 = [image, flipped image] with flip),
 *   flip average of the probabilities, optional --val_clean label mask [c-1], arg-max -> seg_map u8 [H,W] + its probability;
 * cl4_instance_map: per class and 8-connected component of >= min_mask pixels (50 in validation, dataset/utils.py:147):
 *   get_instance_segmentation (thresh / nms_kernel / beta / ignore = --val_thresh / --val_kernel / --beta / --val_ignore) ->
 *   ins_map int32 [H,W] (row of `table`, -1 = none) and table [count,4] = {class (0-based fg), score, pixels, contour};
 *   rows with 0 pixels are centres that attracted no pixel (the reference skips them). Workspace:
 *   cl4_label_workspace_bytes(1, H, W). `center` [classes,H,W] is the (flip-averaged) centre map, `offset` [2,H,W]. */
int cl4_val_seg(const float* logits, const float* cls_label, unsigned char* seg_map, float* prob_win, int nb, int c, int H, int W,
                int flip, void* stream);
typedef struct cl4_insmap_io {
  int32_t H, W, classes, nms_kernel, min_mask, ignore, cap;
  float thresh, beta;
  const uint8_t* seg_map; const float* prob_win; const float* center; const float* offset;
  int32_t* ins_map; float* table; int32_t* count;
} cl4_insmap_io;
int cl4_instance_map(const cl4_insmap_io* d, void* workspace, size_t workspace_bytes, void* stream);

/* ---- input pipeline ---------------------------------------------------------------------------
 * Replaces the per-sample pixel work of the reference's loader: get_dataset's transform stacks (dataset/__init__.py:14-27 over
 * dataset/transform.py: RandomResizedCrop :582-660 / Resize :61-93, RandomHorizontalFlip :317-340, ToTensor :419-445,
 * Normalize :452-478), LabelTransform + instance gating (dataset/dataset.py:104-108,212-214) and label_generation
 * (dataset/utils.py:299-350). The arithmetic is Pillow's (the reference transforms PIL images): BILINEAR = Resample.c two-pass
 * fixed-point convolution, NEAREST = Geometry.c ImagingScaleAffine; results are bit-identical to the reference's tensors. */
typedef struct cl4_sample_desc {
  int64_t img_off;     /* byte offset of this sample's H x W x 3 uint8 RGB photo in the packed `images` buffer */
  int64_t lbl_off;     /* byte offset of its H x W x 2 uint8 (semantic id, instance id) label in `labels` */
  int H, W;            /* source size */
  int ci, cj, ch, cw;  /* crop box: top, left, height, width (RandomResizedCrop.get_params; the whole image for Resize) */
  int flip;            /* RandomHorizontalFlip drew a flip */
  int reserved;
} cl4_sample_desc;

/* Pillow's filter window length for resampling in_size -> out_size with the triangle filter. */
int cl4_input_ksize(int in_size, int out_size);
size_t cl4_input_workspace_bytes(int b, int out_h, int out_w, int max_ch, int ksize);
/* images/labels: DEVICE buffers of packed uint8 photos / labels; descs: HOST array of b descriptors (validated, then copied).
 * mean/std: HOST float[3]. lut: DEVICE uint8[256] LabelTransform mapping or NULL. Outputs (device): out_img float32
 * [b,3,out_h,out_w]; out_seg int64 [b,out_h,out_w] (mapped semantic ids) and out_mask uint8 [b,out_h,out_w] (instance ids gated by
 * seg > 0) when labels != NULL. workspace >= cl4_input_workspace_bytes(b, out_h, out_w, max crop height, max cl4_input_ksize). */
int cl4_input_batch(const uint8_t* images, size_t images_bytes, const uint8_t* labels, size_t labels_bytes,
                    const cl4_sample_desc* descs, int b, int out_h, int out_w, const float* mean, const float* std,
                    const uint8_t* lut, float* out_img, long long* out_seg, uint8_t* out_mask, void* workspace,
                    size_t workspace_bytes, void* stream);
/* label_generation (dataset/utils.py:299-350) for a batch: centre heat-maps [b,num_classes,H,W] (gaussian of dataset/utils.py:
 * 182-192 uploaded as float32 [(6 sigma+3)^2]), offsets [b,2,H,W] and weights [b,1,H,W] from seg_map int64 [b,H,W] and the
 * instance-id map uint8 [b,H,W] (ids 0 and 255 ignored). Outputs are fully overwritten. */
size_t cl4_target_generation_workspace_bytes(int b);
int cl4_target_generation(const long long* seg_map, const uint8_t* mask, int b, int H, int W, int num_classes, int sigma,
                          const float* gauss, float* center, float* offset, float* weight, void* workspace, size_t workspace_bytes,
                          void* stream);

/* InPlaceABN(Sync) (segmentation_module.py:19-27; mapillary/inplace_abn, not vendored) scales with |weight| + eps, not with
 * weight: cl4_abs_eps writes that effective scale (out[i] = |w[i]| + eps), cl4_sign_axpy applies its chain rule
 * (y[i] += w[i] > 0 ? x[i] : -x[i]) to a gradient computed with respect to the effective scale. */
/* Experiments only: phase time stamps of the bf16 forward kernels (tools/probes/probe_timeline.py); NULL = off. */
int cl4_conv_debug_timeline(void* buf);
/* Experiments / tests only: number of launches this process has sent to the unified-wave bf16x3 kernel (csrc/conv_u.hip) and to
 * the planes-input kernel (csrc/conv_s.hip). */
long cl4_conv_debug_u_launches(void);
long cl4_conv_debug_s_launches(void);
long cl4_conv_debug_b_launches(void);   /* launches of the big-tile LDS-DMA planes kernel (conv_b.hip, plan impl 4) */

/* "Planes" activations (csrc/conv_s.hip). The dense convolutions of the frozen networks (modules/residual.py:84-97,
 * modules/deeplab.py:54-70, every nn.Conv2d + InPlaceABN + residual add of the eval-mode forward passes at train.py:253-263) multiply
 * bf16 hi / lo splits of their fp32 operands. A tensor [n][c][h][w] (c % 8 == 0) can be kept pre-split between two such layers:
 *     planes layout   [n][c/8][plane: hi, lo][h*w][8 x bf16],   hi = bf16(x), lo = bf16(x - hi)     (4 bytes per element, like fp32)
 * so that the consuming convolution does no conversion work in its K loop. cl4_to_planes / cl4_from_planes convert a channel slice
 * of an fp32 NCHW tensor to / from that layout (from_planes returns hi + lo: x rounded to 16-17 significant bits — exactly what
 * cl4_conv2d_fwd's bf16x3 arithmetic sees of x anyway).
 * cl4_conv2d_fwd_planes = cl4_conv2d_fwd for a planes INPUT `xs` (descriptor precision 3; 1x1 / 3x3, any stride / pad / dilation,
 * cin % 32 == 0, 8-aligned channel slices, cout % 4 == 0 and >= 64; the packed weights are cl4_pack_conv_weight_bf16x3's). Outputs:
 * `y` (fp32 NCHW, channel slice out_coff of out_ctot) and / or `ys` (planes, the same slice; needs cout % 8 == 0) — at least one.
 * The residual operand is `post` (fp32, [n][cout][hout][wout]) or `post_s` (planes, same shape), not both.
 * Tile widths of 64, 128 or 144 pixels, the persistent pipelined variant (csrc/conv_p.hip) and the LDS-DMA big-tile kernel
 * (csrc/conv_b.hip: 256 or 128 output channels x 144 pixels per workgroup, operands straight into LDS; two or three planes) are
 * launch plans (cl4_conv2d_fwd_set_plan: tile 64 / 144, impl 3, impl 4 / 5 / 6); every plan computes the same products in the same order.
 * THREE planes (r03): cl4_to_planes3 writes [n][c/8][hi, mid, lo][h*w][8 x bf16] (x to 24 significant bits, 6 bytes per element);
 * cl4_conv2d_fwd_planes with descriptor precision 4 multiplies such an input with cl4_pack_conv_weight_bf16x6 weights as six bf16 MFMA
 * products = the bf16x6 arithmetic of the differentiated convolutions (train.py:265-272: the trainable ASPP of the instance decoder
 * reads the frozen res5 features). fp32 output only, no planes residual. */
/* Fused identity bottleneck (r04, csrc/conv_f.hip). Replaces ResidualBlock.forward of modules/residual.py:84-97 for an identity block
 * (no projection shortcut, stride 1, dilation 1) of a FROZEN network in eval mode — the blocks 2..n of res2 / res3 of models/resnet.py:
 * 65-99 on the no-grad forwards of train.py:253-263:
 *     out = leaky( bn3(conv3( leaky(bn2(conv2_3x3( leaky(bn1(conv1(x))) ))) )) + x )
 * as ONE launch: a workgroup owns a 13 x 13 tile of one image, computes conv1 on the tile's 15 x 15 halo, keeps both cm-channel
 * intermediates in LDS (planes format), and reads / writes the 4*cm-channel trunk once. xs / ys: planes tensors [n][4*cm/8][hi, lo]
 * [h*w][8 x bf16] (ys must not alias xs); y: optional fp32 NCHW copy of the output (NULL: none), written for the images [0, n_f32)
 * only (n_f32 < 0: all — train.py:265-267 reads the low-level features of the un-flipped half of the flip-TTA batch only, the rest of
 * `y` is left untouched); w1 / w2 / w3: the three weights packed
 * by cl4_pack_conv_weight_bf16x3; aff: the folded eval-mode ABN affine maps concatenated [scale1 | shift1 | scale2 | shift2] (cm floats
 * each) [scale3 | shift3] (4*cm each); slope1 / slope2 / slope_out: the leaky-ReLU slopes after bn1, bn2 and after the residual sum
 * (bn3 has the identity activation, modules/residual.py:82). Same products in the same order as three cl4_conv2d_fwd_planes calls.
 * cm = 64 (res2, 4-wave workgroups, two per CU) and cm = 128 (res3, one 8-wave workgroup per CU) are built. Enqueue-only on `stream`; no workspace. */
int cl4_bottleneck_fused(const void* xs, void* ys, float* y, int n_f32, const void* w1, const void* w2, const void* w3, const float* aff,
                         int n, int cm, int h, int w, float slope1, float slope2, float slope_out, void* stream);
/* The same for the FIRST block of a stage, the one with a projection shortcut (modules/residual.py:79-82, 90-91; stride 1: res2's block 1,
 * models/resnet.py:68-77): xs holds cm channels, conv1 is cm -> cm, and
 *     out = leaky( bn3(conv3(...)) + bn_p(conv_p(x)) ),      conv_p: 1x1, cm -> 4*cm, weight `wp` (packed like w3),
 * with `aff` = [scale1 | shift1 | scale2 | shift2] (cm each) [scale3 | shift3] [scale_p | shift_p] (4*cm each). The shortcut is rounded through
 * the planes split (hi + lo) before it is added, exactly like the stored shortcut of the cl4_conv2d_fwd_planes path. cm = 64 is built. */
int cl4_bottleneck_fused_proj(const void* xs, void* ys, float* y, int n_f32, const void* w1, const void* w2, const void* w3, const void* wp,
                              const float* aff, int n, int cm, int h, int w, float slope1, float slope2, float slope_out, void* stream);
/* Tile of output pixels per workgroup for middle width cm (CL4_EINVAL: not built). */
int cl4_bottleneck_fused_tile(int cm, int* th, int* tw);
long cl4_conv_debug_f_launches(void);   /* launches of the fused bottleneck kernel by this process (tests) */
int cl4_to_planes(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, void* stream);
int cl4_to_planes3(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, void* stream);
int cl4_from_planes(const void* xs, float* y, int n, int c, int hw, int out_ctot, int out_coff, void* stream);
int cl4_conv2d_fwd_planes(const cl4_conv_desc* d, const void* xs, const float* wp, const float* scale, const float* shift,
                          const float* pre, const float* post, const void* post_s, float* y, void* ys, void* workspace,
                          size_t workspace_bytes, void* stream);
/* fp16x3 (r05): the differentiated convolutions (train.py:272 forward, train.py:537 input gradients) as THREE MFMA products per MAC on
 * split-fp16 operands — x = h + l, both fp16, 22 significant bits: the accuracy of bf16x6's six products (tools/probes/f16x3_probe.hip) —
 * with power-of-two operand scales that keep both halves inside fp16's range:
 *   cl4_pack_conv_weight_f16x3: w [cout][cin][kh][kw] -> [K/8][hi, lo][cout][8 x fp16] of w * 2^12 (cin % 32 == 0, cout % 8 == 0, |w| < 16);
 *   cl4_to_planes_f16: fp32 NCHW channel slice -> [n][c/8][hi, lo][h*w][8 x fp16] of x * s; s = 2^4 (dynamic == 0: activations), or
 *     2^(14 - e) with max |x| < 2^e (dynamic != 0: gradients; one reduction launch in front, no atomics). slot: caller-owned device
 *     memory of cl4_to_planes_f16_slot_floats() floats — [0] receives s, the rest is the reduction's scratch;
 *   cl4_conv2d_fwd_planes_f16: conv (+ the fused affine / activation / fp32 residual group) on such an input, descriptor precision 5,
 *     fp32 NCHW output; xs_scale = &slot[0]. 1x1 / 3x3, any stride / dilation, cout >= 128. */
int cl4_pack_conv_weight_f16x3(const float* w, void* wq, int cout, int cin, int kh, int kw, void* stream);
/* r05: all fp16x3 weight packs of a training iteration in one launch (the trainable convolutions of train.py:272 / :537 re-derive their
 * packed forward weight and their transposed / flipped input-gradient weight after every optimizer step: ~320 small launches in phase 1).
 * jobs: device array of njobs entries; `first` = running sum of the items of the jobs in front, items of a job = (K / 8) * rows with
 * K = kh * kw * (transposed ? cout : cin), rows = transposed ? cin : cout; total_items = their sum. transposed != 0 packs
 * wt[ci][co][kh][kw] = w[co][ci][k-1-kh][k-1-kw] (cl4_transpose_flip_weight followed by cl4_pack_conv_weight_f16x3). Same constraints and
 * bit-identical output as cl4_pack_conv_weight_f16x3 per job. */
typedef struct cl4_pack_job {
  const float* w;   /* [cout][cin][kh][kw] */
  void* wq;         /* the pack: (K / 8) * 2 * rows * 16 bytes */
  long first;
  int cout, cin, khw, transposed;
} cl4_pack_job;
int cl4_pack_conv_weights_f16x3_batched(const cl4_pack_job* jobs, int njobs, long total_items, void* stream);
/* r05: the convolution in front of a train-mode ABN (train.py:272: conv -> InPlaceABNSync) also leaves the per-channel partial sums of its
 * output, so the ABN needs no statistics pass over the tensor: a plain planes convolution (no affine / activation / residual), fp32 NCHW output
 * y, stats[(channel * stats_np + pixel tile) * 2 + {0, 1}] = {sum, sum of squares} in double. stats_np must equal
 * cl4_conv2d_fwd_planes_stats_parts(d) — the pixel tiles of the launch plan this shape currently has, 0 when that plan cannot produce them
 * (K-split plans: run cl4_bn_train_stats instead). precision 3 (xs_scale null) or 5 (fp16 planes with their scale). cl4_bn_stats_from_partials
 * turns such partials (any nparts) into mean / biased variance and updates the running statistics like cl4_bn_train_stats. */
int cl4_conv2d_fwd_planes_stats_parts(const cl4_conv_desc* d);
int cl4_conv2d_fwd_planes_stats(const cl4_conv_desc* d, const void* xs, const float* xs_scale, const float* wp, float* y, double* stats,
                                int stats_np, void* workspace, size_t workspace_bytes, void* stream);
int cl4_bn_stats_from_partials(const double* partial, int nparts, long count, int c, float* mean, float* var, float* running_mean,
                               float* running_var, float momentum, void* stream);
/* ... cross-rank form (InPlaceABNSync, segmentation_module.py:17-19): the partials folded into the packed sums[c * 2] vector the ranks
 * all-reduce — cl4_bn_sync_stats_local without its pass over the tensor; cl4_bn_sync_stats_final follows the exchange. */
int cl4_bn_sync_stats_from_partials(const double* partial, int nparts, double* sums, int c, void* stream);
/* Which two-plane format / arithmetic this build of the library uses for EVERY planes tensor (cl4_to_planes, the ys / post_s operands, the
 * fused bottleneck and stem outputs): 1 = fp16x3 (fp16 hi / lo of 16 x; weights from cl4_pack_conv_weight_f16x3 — the default build),
 * 0 = bf16x3 (bf16 hi / lo of x; weights from cl4_pack_conv_weight_bf16x3; hipcc -DCL4_F16_PLANES=0: the r02-r04 format). */
int cl4_planes_format(void);
size_t cl4_to_planes_f16_slot_floats(void);
int cl4_to_planes_f16(const float* x, void* xs, int n, int ctot, int coff, int c, int hw, int dynamic, float* slot, void* stream);
int cl4_conv2d_fwd_planes_f16(const cl4_conv_desc* d, const void* xs, const float* xs_scale, const float* wp, const float* scale,
                              const float* shift, const float* pre, const float* post, float* y, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Producer-side planes (r05): cl4_bn_apply / cl4_add_act on contiguous [n][c][hw] tensors (c % 8 == 0) that also write the two-plane twin of
 * their output (the build's planes format), so that the convolution behind them multiplies it without a conversion pass of its own:
 * train-mode InPlaceABN apply (segmentation_module.py:17-29) and the residual add + activation of modules/residual.py:91-97. y is bit-identical
 * to the plain entry points. */
int cl4_bn_apply_planes(const float* x, float* y, void* ys, int n, int c, int hw, const float* mean, const float* var, const float* gamma,
                        const float* beta, float eps, int act, float slope, const float* post, int act2, float slope2, void* stream);
                        /* ys may be null; post != null: y = act2(act(bn(x)) + post) (the residual add of a bottleneck in the same pass) */
int cl4_add_act_planes(const float* a, const float* b, float* y, void* ys, int n, int c, int hw, int act, float slope, void* stream);
int cl4_abs_eps(const float* w, float* out, int n, float eps, void* stream);
/* ... of every InPlaceABN layer of a network in one launch: jobs in device memory, `first` = the channels of the jobs in front. */
typedef struct cl4_abs_job {
  const float* w;
  float* out;
  long first;
  int n;
  float eps;
} cl4_abs_job;
int cl4_abs_eps_batched(const cl4_abs_job* jobs, int njobs, long total_items, void* stream);
int cl4_sign_axpy(float* y, const float* x, const float* w, int n, void* stream);

/* --------------------------------------------------------------------------
 * Data-parallel communication (replaces distributed.init_process_group('nccl') run.py:38 and the DistributedDataParallel
 * wrappers of train.py:187-196 on the gradient path). One communicator per process, bound to the calling thread's current
 * HIP device; RCCL is loaded at run time (librccl.so.1). The 128-byte id is created by one rank with cl4_comm_unique_id and
 * handed to every rank by the caller's own bootstrap channel (TCP store, MPI, torch.distributed ...).
 * cl4_allreduce_bucket: in-place fp32 all-reduce (sum, or average when `average` != 0) of `count` floats of a caller-owned
 * buffer, enqueued on `stream` (no host synchronisation; capturable wherever RCCL itself is).
 * -------------------------------------------------------------------------- */
int cl4_comm_unique_id(void* id128);
int cl4_comm_init(int rank, int world, const void* unique_id128);
int cl4_comm_world(void);
int cl4_comm_rank(void);
int cl4_allreduce_bucket(void* buf, size_t count, int average, void* stream);
/* In-place float64 SUM all-reduce of `count` doubles on the same communicator: the cross-rank batch statistics of
 * InPlaceABNSync(group=WORLD) (segmentation_module.py:17-19) — one call per layer and direction on the packed vector
 * [sum x | sum x^2] (forward) / [sum dy | sum dy * xhat] (backward) instead of a torch.distributed call per layer. */
int cl4_allreduce_f64(void* buf, size_t count, void* stream);
int cl4_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* CL4WSIS_H */
