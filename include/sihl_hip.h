/* sihl_hip.h - C ABI of the MI355X (gfx950) kernels behind sihl's backbone -> FPN/BiFPN -> dense-head hot path.
 *
 * The reference (jonregef/sihl, /root/reference/src/sihl) is pure Python: there is no FFI to mirror, the
 * "plugin interface" is nn.Module duck typing.  This header is therefore the boundary a
 * maintainer binds instead of the ATen calls each reference line makes; every entry cites the reference
 * code it replaces.  Binding example (ctypes, what sihl_amd/_C.py does): INTEGRATION.md.
 *
 * Conventions
 *   - layout: activations are NHWC ([N][H][W][C], torch.channels_last storage); weights [Cout][KH][KW][Cin];
 *     rows x C matrices for the MLP heads.  C must be a multiple of the 16-byte vector (4 fp32 / 8 bf16)
 *     unless an entry says otherwise.
 *   - dtype: SIHL_F32 (exact fp32 MFMA, the 1e-4 parity configuration) or SIHL_BF16 (bf16 storage, fp32
 *     accumulation).  Statistics, norm parameters and every gradient of a parameter are fp32.
 *   - ownership: every buffer is caller-owned device memory (PyTorch's caching allocator in sihl_amd);
 *     kernels never allocate.  Scratch comes in through (ws, ws_bytes); the *_ws_bytes helpers size it.
 *   - streams: every launch goes to the hipStream_t argument, no implicit synchronisation; entries are
 *     re-entrant.  Process-wide switches, none of them touched by the product path: the opt-in launch profiler,
 *     the TEST hooks that select between two parity-tested kernels for the same result (sihl_conv2d_tile_override,
 *     *_enable, *_force_*), and the TUNING setters (sihl_conv2d_debug, sihl_conv2d_nbuf_override,
 *     sihl_conv2d_rules_off, sihl_conv2d_krot, sihl_mlp_debug, sihl_mlp_rows_debug), which are live only in a
 *     `make TUNING=1` library: the shipped library stores nothing and answers SIHL_EARG to any non-default value.
 *   - errors: 0 = ok, SIHL_EARG (-1) = bad argument / unsupported shape, SIHL_EWS (-2) = workspace too
 *     small, > 0 = hipError_t.  Nothing throws across the ABI.
 */
#ifndef SIHL_HIP_H
#define SIHL_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIHL_F32 0
#define SIHL_BF16 1
#define SIHL_ACT_NONE 0
#define SIHL_ACT_RELU 1
#define SIHL_ACT_SILU 2
#define SIHL_ACT_SIGMOID 3
#define SIHL_EARG (-1)
#define SIHL_EWS (-2)

/* ---- convolution / linear on the matrix cores ------------------------------------------------------------
 * Replaces nn.Conv2d + activation + norm-apply of ConvNormAct (layers/convblocks.py:37-87), torchvision
 * Conv2dNormActivation (layers/fpn.py:26-37, heads/object_detection.py:52-55) and nn.Linear inside ops.MLP
 * (heads/object_detection.py:51-61; a Linear is a 1x1 conv over rows: N=1, H=1, W=rows).
 * out[m][co] = post( act( pre( conv(in, wt)[m][co] + bias[co] ) ) ), pre/post = per-channel scale*x+shift
 * (any of bias/pre_/post_ may be NULL).  stats_mode 1 / 2 additionally writes per-channel (sum, sumsq)
 * partials of the value after bias / after the activation to stats_ws [sihl_conv2d_stat_rows(M)][2][Cout]
 * (BatchNorm batch statistics, one deterministic row per 128-pixel tile).  out_image_stride (elements,
 * 0 = dense) lets the head's laterals write into a slice of the flat (B, P, C) position buffer
 * (object_detection.py:102-105).  Cout is unconstrained; Cin % vector == 0. */
int sihl_conv2d_stat_rows(long M);
/* Test hook: the conv has two loaders, LDS-DMA (default) and register-staged (used for >= 4 GiB tensors);
 * on != 0 forces the register-staged one so that both stay parity-tested. */
int sihl_conv2d_force_register_staging(int on);
/* Tuning hook: pixel-tile size of the LDS-DMA kernel for Cout > 128 (0 = heuristic, 128 or 256). */
int sihl_conv2d_tile_override(int bm);
int sihl_conv2d_nbuf_override(int n); /* TUNING builds only: LDS stages of the narrow-tile kernels, 0 = default */
/* TUNING builds only - ablation of the LDS-DMA kernel (results are INVALID when non-zero): 1 = no in-loop DMA,
 * 2 = no ds_read/MFMA.  The shipped library refuses any non-zero mode (SIHL_EARG). */
int sihl_conv2d_debug(int mode);
int sihl_conv2d_strided_classes_enable(int on); /* test hook: 0 = zero-dilated read for 3x3 stride-2 dgrads */
int sihl_conv2d_fwd(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                    const float* pre_scale, const float* pre_shift, const float* post_scale,
                    const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                    long out_image_stride, hipStream_t stream);

/* Input gradient of a (possibly strided) conv: din [N][H][W][Cin] from dout [N][Ho][Wo][Cout] with
 * wt_t = sihl_weight_flip_transpose(w, flip=1) = [Cin][KH][KW][Cout]; (N,H,W,Cin,Cout,...) describe the FORWARD conv.
 * Same kernel as the forward (dout is read as if zero-dilated by `stride`). */
int sihl_conv2d_dgrad(const void* dout, const void* wt_t, void* din, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, hipStream_t stream);

/* The same two entry points with caller scratch: tiny pyramid levels (<= 64 workgroups of 128 x 64) slice their K loop
 * over the grid (split-K) when ws_bytes >= sihl_conv2d_ws_bytes(...) of the forward-shaped problem; ws may be NULL. */
long sihl_conv2d_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil);
int sihl_conv2d_fwd_ws(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                       int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                       const float* pre_scale, const float* pre_shift, const float* post_scale,
                       const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                       long out_image_stride, void* ws, long ws_bytes, hipStream_t stream);
/* add: optional dense [N][H][W][Cin] tensor added to the result (identity-branch gradient of a residual block) */
int sihl_conv2d_dgrad_ws(const void* dout, const void* wt_t, void* din, const void* add, int N, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws, long ws_bytes,
                         hipStream_t stream);
/* sihl_conv2d_dgrad_ws whose addend may be COMPACT: with add_stride s > 1, `add` is [N][ceil(H/s)][ceil(W/s)][Cin] -
 * the input gradient of a stride-s 1x1 projection of the same input (ResNet downsample branch) - and is added at the
 * pixels (y % s == 0, x % s == 0) that projection reads; the other pixels get the conv's own gradient only.  Replaces
 * the zero-dilated dgrad of the projection plus autograd's add over the block input. */
int sihl_conv2d_dgrad_add(const void* dout, const void* wt_t, void* din, const void* add, int add_stride, int N, int H,
                          int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws,
                          long ws_bytes, hipStream_t stream);
int sihl_conv2d_splitk_enable(int on); /* tuning / test hook */
int sihl_conv2d_small_enable(int on);  /* test hook: kernel of the 3x3 convs on the small pyramid levels - 1 (default) csrc/conv_pyr.hip where its shapes allow, else csrc/conv_small.hip; 2 = conv_small.hip only; 0 = the general tile kernel */
int sihl_conv2d_small_mode(void);      /* the current value of that hook */
int sihl_conv2d_halo_enable(int mode); /* test hook: csrc/conv_halo.hip (3x3 on 64-wide maps, 256 x 256 tile, input patch resident in LDS) - 1 (default) where its grid fills the chip, 2 wherever the shape allows, 0 off */
int sihl_conv2d_rules_off(int mask);   /* TUNING builds only: disable individual dispatch rules */
int sihl_conv2d_krot(int n);           /* TUNING builds only: stage stride between neighbouring workgroups' K-loop starts (100000 * log2(group) + 1000 * min_stages + stride; default 200013 = groups of 4 workgroups share a start, stride 13; 0 = lockstep) */

/* Weight gradient (autograd of Conv2d.weight / Linear.weight): dw fp32 [Cout][KH][KW][Cin];
 * accumulate != 0 adds into dw.  Cin, Cout % vector == 0. */
/* Test hook: bf16 layers with >= 128 channels use an LDS-DMA 256x256-panel kernel; on != 0 forces the
 * register-staged 128x128 kernel (the fp32 / small-channel path) so both stay parity-tested. */
int sihl_conv2d_wgrad_force_register_staging(int on);
/* target: K-split aim (workgroups) of THIS call - target % 10000 for the LDS-DMA kernel (0 = 256, one per CU),
 * target / 10000 for the register-staged / all-taps kernels (0 = default).  A per-call argument (no process-wide
 * state): the workspace query takes the same value.  The two-stream training step passes 128 / 128 for the weight
 * gradients that run beside the dgrad chain. */
long sihl_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int dil, int dtype, int target);
int sihl_conv2d_wgrad(const void* in, const void* dout, float* dw, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, int accumulate, int target, void* ws,
                      long ws_bytes, hipStream_t stream);

/* [Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout], spatially flipped when flip != 0: the weights with which
 * sihl_conv2d_fwd computes the INPUT gradient of a stride-1 conv (pad' = dil*(K-1) - pad) or of a Linear. */
int sihl_weight_flip_transpose(const void* w, void* o, int Cout, int KH, int KW, int Cin, int flip, int dtype_in,
                               int dtype_out, hipStream_t stream);

/* All weights of a model in one launch: fp32 master [O][I][KH][KW] (any strides) -> bf16 w [Op][KH][KW][I] and
 * bf16 wt [I][KH][KW][Op] (flipped when flip != 0), Op = O padded to the vector width with zero rows.  descs is a
 * DEVICE array of n records {const float* src; bf16* w; bf16* wt; int64 so, si, sky, skx; int32 O, Op, KH, KW, I,
 * flip; int64 first_block, first_tblock} (96 bytes); a record owns blocks [first_block, first_block +
 * ceil(Op*KH*KW*I/2048)) of the w pass and tiles [first_tblock, first_tblock + KH*KW*ceil(Op/64)*ceil(I/64)) of the
 * transposing wt pass.
 * Replaces the per-layer .to(bf16) + sihl_weight_flip_transpose the autocast path of the reference implies. */
int sihl_weight_prepare(const void* descs, int n, long total_blocks, long total_tblocks, hipStream_t stream);

/* ---- BatchNorm2d (convblocks.py:82-85; torch defaults eps 1e-5, momentum 0.1) ------------------------------
 * finalize: partial sums -> batch mean / rstd (biased variance), scale = gamma*rstd, shift = beta - mean*scale,
 * running_mean / running_var updated in place (unbiased variance), any of them may be NULL.
 * eval_affine: scale / shift from the running statistics. */
int sihl_bn_finalize(const float* partials, int n_partials, int C, long count, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, hipStream_t stream);
int sihl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, hipStream_t stream);

/* y = act(x*scale[c] + shift[c]) over [rows][C] (norm-apply; stand-alone SiLU / sigmoid), and its input gradient. */
int sihl_affine_act(const void* x, void* y, long rows, int C, const float* scale, const float* shift, int act,
                    int dtype, hipStream_t stream);
/* y = act(x*scale[c] + shift[c] + res): BatchNorm-apply + residual add + activation of a ResNet block tail.
 * mask (optional, bytes [rows * C / V], V = 8 bf16 / 4 fp32 elements per 16-byte vector): bit e of byte i = (element e of
 * vector i of y > 0) - the ReLU mask for sihl_norm_add_relu_bwd, so the backward need not re-read y. */
int sihl_affine_add_act(const void* x, const void* res, void* y, void* mask, long rows, int C, const float* scale,
                        const float* shift, int act, int dtype, hipStream_t stream);
int sihl_affine_act_bwd(const void* x, const void* dy, void* dx, long rows, int C, const float* scale,
                        const float* shift, int act, int dtype, hipStream_t stream);

/* The ResNet stem: conv 7x7 / stride 2 / pad 3 of a 3-channel image into 64 channels (torchvision resnet.py conv1, wrapped
 * by src/sihl/torchvision_backbone.py:42-49), bf16 on the matrix cores (csrc/stem.hip).  x: [N][3][H][W] fp32 or bf16 with
 * the given element strides (any layout), H and W even; w: the fp32 master weights [64][3][7][7] with their element strides.
 * xp (sihl_stem_xp_bytes bytes): receives the zero-padded NHWC bf16 copy of the image the kernel reads - keep it for the
 * weight gradient; wp: 64 * 7 * 32 * 2 bytes of scratch for the packed weights.  out: [N][H/2][W/2][64] bf16.
 * stats (optional): [sihl_stem_stats_rows(N, H)][2][64] fp32 partial sums / sums of squares of `out` for sihl_bn_finalize. */
long sihl_stem_xp_bytes(int N, int H, int W);
int sihl_stem_stats_rows(int N, int H);
int sihl_stem_conv_fwd(const void* x, int x_dtype, long xsn, long xsc, long xsh, long xsw, const float* w, long wso, long wsc,
                       long wsh, long wsw, void* xp, void* wp, void* out, float* stats, int N, int H, int W,
                       hipStream_t stream);
/* Weight gradient of that conv: dw[64][3][7][7] (fp32, element strides given) from dz[N][H/2][W/2][64] (bf16, the gradient
 * of `out`) and the packed image xp the forward wrote; ws: sihl_stem_wgrad_parts(N, H) * 64 * 7 * 32 floats (fp32 partials,
 * summed in a fixed order: deterministic).  The image needs no gradient: there is no dgrad entry. */
int sihl_stem_wgrad_parts(int N, int H);
int sihl_stem_conv_wgrad(const void* xp, const void* dz, float* dw, long wso, long wsc, long wsh, long wsw, float* ws, int N,
                         int H, int W, hipStream_t stream);

/* Batch statistics of an NHWC tensor another kernel produced (the ResNet stem's MIOpen conv): per-channel (sum, sumsq)
 * partial rows [sihl_bn_stats_rows(...)][2][C] in the layout sihl_bn_finalize reads (torchvision resnet.py stem:
 * conv1 -> bn1 -> relu, wrapped by src/sihl/torchvision_backbone.py:42-49). */
int sihl_bn_stats_rows(long rows, int C, int dtype);
int sihl_bn_stats(const void* x, long rows, int C, float* partials, int n_partials, int dtype, hipStream_t stream);

/* Backward through [activation -> BatchNorm] (mode 0, ConvNormAct: s = act(conv)) or [BatchNorm -> activation]
 * (mode 1, Conv2dNormActivation: s = conv): dz = grad wrt the conv output, dgamma / dbeta fp32 [C].
 * batch_stats != 0 includes the batch-mean / variance terms (training mode). */
long sihl_norm_act_bwd_ws_bytes(long rows, int C, int dtype);
int sihl_norm_act_bwd(const void* s, const void* dy, void* dz, long rows, int C, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float* dgamma, float* dbeta, int mode, int act,
                      int batch_stats, int dtype, float* ws, long ws_bytes, hipStream_t stream);

/* Backward of a residual block's tail y = relu(BatchNorm(s) + identity) (torchvision Bottleneck / BasicBlock merge, wrapped
 * by src/sihl/torchvision_backbone.py): dres = dy * (y > 0) - the identity branch's gradient - and dz = BatchNorm backward
 * of dres, the ReLU mask applied inside the column reduction.  ws as for sihl_norm_act_bwd.
 * The mask comes from `mask` (the bytes sihl_affine_add_act wrote) when given, else from y; one of the two may be NULL. */
int sihl_norm_add_relu_bwd(const void* s, const void* dy, const void* y, const void* mask, void* dres, void* dz, long rows,
                           int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float* dgamma, float* dbeta, int batch_stats, int dtype, float* ws, long ws_bytes,
                           hipStream_t stream);

/* ---- BiFPN fusion nodes, fused with their producer (layers/bifpn.py:10-17,39-53) --------------------------
 * w = softmax(wraw) exactly as FastNormalizedFusion (bifpn.py:16); wraw is the raw nn.Parameter (fp32).
 * fuse_up2 : out[N][H][W][C] = w0 * bilinear_x2(a[N][H/2][W/2][C]) + w1 * b   (Interpolate scale=2,
 *            align_corners=False, scalers.py:36-47).  b == NULL: plain upsample.
 * blur_fuse: out = w0 * blurpool(a) + w1 * b + w2 * c with blurpool = reflect-pad 1, [1,2,1]x[1,2,1]/16, stride 2
 *            (layers/pooling.py:7-26).  b == c == NULL: plain BlurPool2d.
 * fuse_sum : stand-alone FastNormalizedFusion of n = 2 or 3 tensors.
 * *_bwd    : input gradients (NULL = not needed) and dw_raw (softmax Jacobian applied); gacc = SIHL_FUSION_GACC_FLOATS
 *            floats of scratch (one partial row per workgroup of the reducing kernel, summed in a fixed order: the
 *            fusion-weight gradients are bit-reproducible; need not be initialised). */
#define SIHL_FUSION_GACC_FLOATS (4 * 4096)
int sihl_fuse_up2(const void* a, const void* b, const float* wraw, void* out, int N, int H, int W, int C, int dtype,
                  hipStream_t stream);
int sihl_fuse_up2_bwd(const void* dout, const void* a, const void* b, const float* wraw, void* da, void* db,
                      float* dw_raw, float* gacc, int N, int H, int W, int C, int dtype, hipStream_t stream);
/* a_scale / a_shift (fp32 [C], both or neither): `a` stands for a * scale + shift - the training-mode BatchNorm affine of
 * the ConvNormAct that produced it (layers/scalers.py:26-30 AntialiasedDownscaler = conv block -> BlurPool2d), applied to
 * the blurred value instead of in a pass of its own; the backward's `da` is then the gradient of a * scale + shift. */
int sihl_blur_fuse(const void* a, const void* b, const void* c, const float* wraw, const float* a_scale,
                   const float* a_shift, void* out, int N, int H, int W, int C, int dtype, hipStream_t stream);
int sihl_blur_fuse_bwd(const void* dout, const void* a, const void* b, const void* c, const float* wraw,
                       const float* a_scale, const float* a_shift, void* da, void* db, void* dc, float* dw_raw,
                       float* gacc, int N, int H, int W, int C, int dtype, hipStream_t stream);

/* Gradient-norm clipping of n dense fp32 tensors (torch.nn.utils.clip_grad_norm_, the reference's Lightning
 * gradient_clip_val, examples/object_detection.py:288-296): total = sqrt(sum of squares of everything), every tensor *= min(1,
 * max_norm / (total + 1e-6)).  Three launches per 320 tensors instead of the library's per-tensor result tensors.
 * grads: HOST array of n device pointers, taken in groups of `group` (32 or 320: the pointer table rides by value in the launch
 * arguments); group_blocks: HOST int [ceil(n / group)], workgroups per group;
 * map: DEVICE int32 [sum(group_blocks)][2] = (tensor index within its group, 65 536-element chunk) per workgroup and numel:
 * DEVICE int64 [n] - functions of the sizes only, built once by the caller; scratch: sum(group_blocks) + 2 floats; on return
 * (stream order) scratch[nblocks] = the coefficient, scratch[nblocks + 1] = the total norm. */
int sihl_grad_clip(const void* const* grads, int n, const int* map, const int* group_blocks, const long* numel, float max_norm,
                   float* scratch, long scratch_floats, int group, hipStream_t stream);

/* 3x3 / stride 1 / pad 1 conv of the pyramid's TOP levels with its fusion node folded into the loader (bf16; square maps of
 * W = 16, 8 or 4; csrc/conv_pyr.hip).  One launch replaces [fusion kernel -> conv -> split-K finish] of a BiFPNLayer node
 * (layers/bifpn.py:41-52: `up_convs[i](up_fusions[i]([upscalers[i](td[-1]), inputs[..]]))` and
 * `down_convs[i](down_fusions[i]([downscalers[i](bu[-1]), inputs[i + 1], td[i + 1]]))`, conv blocks of
 * layers/convblocks.py:37-87), bit-identical to those kernels run one after the other:
 *   mode 0: out = epilogue(conv3x3(in))
 *   mode 1: m = softmax(fw)_0 * bilinear_x2(a) + softmax(fw)_1 * b          a [N][W/2][W/2][Cin], b [N][W][W][Cin]; W = 8 or 4
 *   mode 2: m = softmax(fw)_0 * (blur_s2(a) * a_scale + a_shift) + softmax(fw)_1 * b + softmax(fw)_2 * c
 *           a [N][2W][2W][Cin] (reflect-padded binomial blur, stride 2: layers/pooling.py:7-26), b, c [N][W][W][Cin];
 *           a_scale / a_shift optional (both or neither), as in sihl_blur_fuse; W = 8 or 4 only
 *   modes 1, 2: out = epilogue(conv3x3(m)); `merged` (optional, [N][W][W][Cin]) receives m (the weight gradient needs it).
 *   emit (inference): the fusion node that CONSUMES `out`, computed in the epilogue from the workgroup's own 32-channel slice
 *   of the whole map - 1: e_out [N][2W][2W][Cout] = softmax(e_fw)_0 * bilinear_x2(out) + softmax(e_fw)_1 * e_b;
 *   2: e_out [N][W/2][W/2][Cout] = softmax(e_fw)_0 * blur_s2(out) + .._1 * e_b + .._2 * e_c; 0: none.  `out` may be NULL
 *   when the emitted node was its only consumer (the AntialiasedDownscaler's conv in front of its blur).
 * wt [Cout][3][3][Cin]; Cin % 64 == 0, Cout % 32 == 0.  Epilogue as sihl_conv2d_fwd (bias -> [stats] -> pre-affine -> act ->
 * [stats] -> post-affine); statistics rows: sihl_pyr_conv_stat_rows(N, W) of [2][Cout] fp32 (one per 128 pixels on 16x16
 * maps, else one per tile: an 8x8 image / four 4x4 images), to be reduced by sihl_bn_finalize. */
int sihl_pyr_conv_supported(int N, int W, int Cin, int Cout, int mode);
int sihl_pyr_conv_stat_rows(int N, int W);
int sihl_pyr_conv_fwd(const void* in, const void* wt, const float* bias, void* out, int N, int W, int Cin, int Cout, int act,
                      const float* pre_scale, const float* pre_shift, const float* post_scale, const float* post_shift,
                      int stats_mode, float* stats, long stats_bytes, int mode, const void* a, const void* b, const void* c,
                      const float* fw, const float* a_scale, const float* a_shift, void* merged, int emit, const void* e_b,
                      const void* e_c, const float* e_fw, void* e_out, hipStream_t stream);
int sihl_fuse_sum(const void* x0, const void* x1, const void* x2, const float* wraw, void* out, long numel, int n,
                  int dtype, hipStream_t stream);
int sihl_fuse_sum_bwd(const void* dout, const void* x0, const void* x1, const void* x2, const float* wraw, void* d0,
                      void* d1, void* d2, float* dw_raw, float* gacc, long numel, int n, int dtype,
                      hipStream_t stream);

/* ---- FPN / SPPM resampling (layers/fpn.py:43-48; heads/semantic_segmentation.py:139,154) -------------------
 * nearest_up2_add: out[N][H][W][C] = nearest_x2(lo) + skip; its backward returns the 2x2 block sums (the skip
 * gradient is dout itself).  resize_bilinear: any size, align_corners=False, optional "+ add". */
int sihl_nearest_up2_add(const void* lo, const void* skip, void* out, int N, int H, int W, int C, int dtype,
                         hipStream_t stream);
int sihl_nearest_up2_add_bwd(const void* dout, void* dlo, int N, int H, int W, int C, int dtype, hipStream_t stream);
int sihl_resize_bilinear(const void* a, const void* add, void* out, int N, int H, int W, int Ho, int Wo, int C,
                         int dtype, hipStream_t stream);
int sihl_resize_bilinear_bwd(const void* dout, void* da, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                             hipStream_t stream);

/* ---- ResNet residual merge (torchvision resnet Bottleneck / BasicBlock: out = relu(bn(conv(..)) + identity)) ------
 * out = act(a + b), act none or relu; backward = sihl_affine_act_bwd(out, dout, relu) for both inputs. */
int sihl_add_act(const void* a, const void* b, void* out, long numel, int act, int dtype, hipStream_t stream);

/* nn.MaxPool2d(3, 2, 1) of the ResNet stem (torchvision_backbone.py:135-138 via torchvision's resnet) on NHWC tensors:
 * y [N][Ho][Wo][C], Ho = (H-1)/2 + 1; idx = one byte per output element (winning tap 0..8, first maximum in row-major
 * window order, NaN wins) consumed by the backward, which gathers per input pixel (no atomics). */
int sihl_maxpool3x3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int dtype, hipStream_t stream);
int sihl_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int dtype,
                          hipStream_t stream);

/* ---- MLP hidden layers: y = act(LayerNorm(z)*gamma + beta) over [rows][C] (object_detection.py:51-61) ----- */
/* The WHOLE MLP in one launch, inference (heads/object_detection.py:51-61,108-121; SURVEY App. D counts an MLP as one op):
 * out[rows][out_stride] = Linear_n(act(LN(Linear_{n-1}(... act(LN(Linear_0(x)))...)))) with a 128-row activation tile
 * resident in LDS across the layers and the weight panels streamed through an LDS ring (csrc/mlp_fused.hip).  bf16 only;
 * x: rows of Cin elements, x_stride elements apart; w / bias / gamma / beta: HOST arrays of device pointers - nhidden + 1
 * weights ([C][Cin], [C][C], ..., [Cout][C]; bf16 row-major) and biases (fp32, entries may be NULL), nhidden LayerNorm
 * scale / shift vectors (fp32).  Cin, C, Cout <= 256; Cin, C multiples of 8; out_stride a multiple of 8, >= Cout (the
 * padding columns are written as 0).  Same arithmetic as sihl_conv2d_fwd + sihl_layernorm_act per layer, up to the order
 * of the two row reductions. */
int sihl_mlp_fwd_supported(long rows, int Cin, int C, int Cout, int nhidden, int act, int dtype);
int sihl_mlp_fwd(const void* x, long x_stride, long rows, int Cin, int C, int nhidden, const void* const* w,
                 const float* const* bias, const float* const* gamma, const float* const* beta, float eps, int act,
                 int Cout, void* out, int out_stride, int dtype, hipStream_t stream);
/* The same MLP with the activations in REGISTERS (csrc/mlp_rows.hip): a wave owns 32 rows and all channels of them, the
 * accumulators of a layer are normalised, activated and packed in place into the next layer's matrix operand; LDS holds
 * only the weight ring, two workgroups share a CU.  As sihl_mlp_fwd, except: nhidden >= 1, C == 256, and w[l]
 * for l >= 1 must be in the K order sihl_mlp_permute_k produces (inside each group of 16 input channels
 * [0-3, 8-11, 4-7, 12-15] - the order in which a lane's accumulators hold a row's channels); w[0] is plain. */
int sihl_mlp_rows_supported(long rows, int Cin, int C, int Cout, int nhidden, int act, int dtype);
/* Up to 4 such MLPs in ONE launch (the class / box heads of a detection head: the same few thousand rows, each MLP alone
 * fills a tenth of the chip for the same 35 us); same activation for all, they may share x. */
typedef struct sihl_mlp_call {
  const void* x; long x_stride; long rows; int Cin, C, nhidden, Cout, out_stride; float eps;
  const void* const* w; const float* const* bias; const float* const* gamma; const float* const* beta;
  void* out;
} sihl_mlp_call;
int sihl_mlp_rows_fwd_multi(const sihl_mlp_call* calls, int n, int act, int dtype, hipStream_t stream);
int sihl_mlp_rows_config(int waves); /* test hook: 4 (default) = two 128-row workgroups per CU, 8 = one 256-row workgroup with a 4-stage weight ring (measured slower) */
int sihl_mlp_rows_debug(int mode); /* timing ablations, `make TUNING=1` builds only (results invalid when non-zero) */
int sihl_mlp_permute_k(const void* w_in, void* w_out, long Cout, int K, hipStream_t stream);
int sihl_mlp_rows_fwd(const void* x, long x_stride, long rows, int Cin, int C, int nhidden, const void* const* w,
                      const float* const* bias, const float* const* gamma, const float* const* beta, float eps, int act,
                      int Cout, void* out, int out_stride, int dtype, hipStream_t stream);
int sihl_mlp_stages(int n); /* tuning hook: LDS stages of the weight ring (2 or 3) */
int sihl_mlp_stamps(void* buf64); /* diagnostic builds (-DSIHL_MLP_STAMPS): 64 x u64 s_memtime marks of workgroup 0 */
int sihl_mlp_debug(int mode); /* timing ablations, `make TUNING=1` builds only (results invalid when non-zero) */
int sihl_layernorm_act(const void* z, void* y, long rows, int C, const float* gamma, const float* beta, float eps,
                       int act, float* mean, float* rstd, int dtype, hipStream_t stream);
int sihl_layernorm_bwd_waves(long rows); /* workgroups = partial rows of the backward (workspace sizing) */
long sihl_layernorm_act_bwd_ws_bytes(long rows, int C);
int sihl_layernorm_act_bwd(const void* z, const void* dy, void* dz, long rows, int C, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, int act, float* dgamma,
                           float* dbeta, int dtype, float* ws, long ws_bytes, hipStream_t stream);

/* out[c] = sum_r x[r][c] (bias gradients). */
long sihl_colsum_ws_bytes(long rows, int C);
int sihl_colsum(const void* x, long rows, int C, float* out, int dtype, float* ws, long ws_bytes, hipStream_t stream);

/* ---- ObjectDetection.training_step loss (heads/object_detection.py:157-208): the four loss sums and their gradients
 * with respect to the head's outputs in one pass over the matching's target tensors.  loc / iou [n_positions], box
 * [n_rows][4], cls [n_rows][C] in `dtype`; targets fp32, tgt_cls int64; loc_norm / iou_norm / wsum / none_matched (bool)
 * are DEVICE scalars.  losses[5] = total, location, box, class, iou; d_* = gradients of the total.  ws:
 * sihl_od_loss_ws_bytes. */
long sihl_od_loss_ws_bytes(long n_positions, int n_rows);
int sihl_od_loss(const void* loc, const void* iou, const void* box, const void* cls, const float* loc_target,
                 const float* rel_iou, const float* cand_off, const float* cand_scale, const float* tgt_box,
                 const float* wts, const long* tgt_cls, const float* loc_norm, const float* iou_norm, const float* wsum,
                 const void* none_matched, long n_positions, int n_rows, int C, void* d_loc, void* d_iou, void* d_box,
                 void* d_cls, float* losses, int dtype, float* ws, long ws_bytes, hipStream_t stream);

/* ---- ObjectDetection.forward decode (heads/object_detection.py:99-122, anchors :83-97) ---------------------
 * topk_rows : per image, the K largest of P position logits (estride elements apart), sorted descending
 *             (object_detection.py:109); vals fp32 [B][K], idx int32 [B][K].
 * gather_rows: out[b][k][:] = src[b][idx[b][k]][:]                                   (:110-112)
 * od_decode : scores = sigmoid(vals), num_instances = #(scores > 0.5), classes = argmax(cls_logits),
 *             boxes = (offsets + scales*exp(box_raw)) * (W,H,W,H) with closed-form cell anchors (:113-121);
 *             level_hw is a HOST array [n_levels][2] of (h, w), bottom level first.
 * od_anchors: the (P,4) offsets / scales tensors of get_offsets_and_scales (:83-97), for the training loss. */
int sihl_topk_select_enable(int on); /* test hook: 0 = the full LDS bitonic sort instead of the radix select */
int sihl_topk_rows(const void* x, int B, int P, int K, int estride, float* vals, int* idx, int dtype,
                   hipStream_t stream);
int sihl_gather_rows(const void* src, const int* idx, void* out, int B, int P, int K, int C, int dtype,
                     hipStream_t stream);
int sihl_od_decode(const float* top_vals, const int* top_idx, const void* cls_logits, long cls_stride, const void* box_raw,
                   long box_stride, const int* level_hw, int n_levels, int B, int K, int ncls, int full_w, int full_h,
                   float* scores, long* classes, float* boxes, long* num_instances, int dtype, hipStream_t stream);
/* (cls_stride / box_stride: elements between the rows of cls_logits / box_raw - views of vector-padded MLP outputs; 0 = dense) */
int sihl_od_anchors(const int* level_hw, int n_levels, float* offsets, float* scales, hipStream_t stream);

/* ---- InstanceSegmentation mask decode (heads/instance_segmentation.py:121-163) -----------
 * CondInst dynamic 1x1 network (10 -> 8 -> 8 -> 1, SiLU, sigmoid; 169 parameters per instance from the kernel
 * MLP) over the mask features (B, h, w, 8) NHWC + relative coordinates, fused with the bilinear resize to
 * (B, K, H, W).  dyn rows: w1[10][8], b1[8], w2[8][8], b2[8], w3[8], b3; dstride = elements between rows. */
int sihl_iseg_mask_decode(const void* feats, const void* dyn, long dstride, const int* top_idx, const int* level_hw,
                          int n_levels, int B, int K, int h, int w, int H, int W, void* out, int dtype,
                          hipStream_t stream);

/* ---- SemanticSegmentation head (heads/semantic_segmentation.py) -------------------------------------------------
 * uafm_fwd: UAFM (:163-182): out = x1*a + x2*(1-a), a = sigmoid(conv3x3_{4->1}([mean_c x1, max_c x1, mean_c x2,
 *   max_c x2]) + b); conv_w is the (1,4,3,3) weight (fp32, contiguous), conv_b one float or NULL.  stats
 *   [N][H][W][4] fp32, arg [N][H][W][2] int32 (argmax channels), alpha [N][H][W] fp32 are outputs kept for uafm_bwd.
 * softmax_max_resize: forward (:83-85) = nearest resize of the logits to (H, W) + softmax + max, fused; any C.
 * ce_resize: training_step (:87-92) = nearest resize to the target size + cross_entropy(ignore_index), fused:
 *   acc = (sum of per-pixel losses, #valid targets); dl = d(sum loss)/d logits * inv_count[0]. */
int sihl_uafm_fwd(const void* x1, const void* x2, const float* conv_w, const float* conv_b, void* out, float* stats,
                  int* arg, float* alpha, int N, int H, int W, int C, int dtype, hipStream_t stream);
long sihl_uafm_bwd_ws_bytes(int N, int H, int W);
int sihl_uafm_bwd(const void* dout, const void* x1, const void* x2, const float* conv_w, const float* stats,
                  const int* arg, const float* alpha, void* dx1, void* dx2, float* dconv_w, float* dconv_b, int N,
                  int H, int W, int C, int dtype, float* ws, long ws_bytes, hipStream_t stream);
int sihl_softmax_max_resize(const void* logits, float* scores, long* classes, int N, int h, int w, int C, int H,
                            int W, int dtype, hipStream_t stream);
/* acc: SIHL_CE_ACC_FLOATS floats - acc[0] = loss sum, acc[1] = valid-target count on return; the rest holds per-workgroup
 * partial sums that are added in a fixed order (bit-reproducible; no initialisation needed). */
#define SIHL_CE_ACC_FLOATS (2 + 2 * 2048)
int sihl_ce_resize(const void* logits, const long* targets, long ignore_index, const float* inv_count, void* dl,
                   float* acc, int N, int h, int w, int C, int H, int W, int dtype, hipStream_t stream);

/* ---- opt-in launch profiler (bench.py roofline leg): HIP events around the matrix-core launches ------------
 * slot 0 = conv (fwd / dgrad / linear), slot 1 = wgrad. */
int sihl_profile_enable(int on);
int sihl_profile_collect(int slot, int dtype, long* launches, double* total_ms, double* total_flops,
                         double* total_bytes);
long sihl_profile_records(int slot, int dtype, double* out, long cap);

#ifdef __cplusplus
}
#endif
#endif /* SIHL_HIP_H */
