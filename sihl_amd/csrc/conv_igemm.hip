// C-ABI entry points of the implicit-GEMM convolution (include/sihl_hip.h): argument checks, ConvParams, the parity
// classes of strided dgrads, the test / tuning hooks.  The kernels and the tile dispatch are in conv_igemm_impl.h,
// instantiated per element type in conv_igemm_bf16.hip / conv_igemm_f32.hip.
#include "common.h"
#include "conv_params.h"
#include "conv_tuning.h"

ConvTuning sihl_conv_tuning = {false, 0, 0, 0, true, true, 0, 200013};

extern "C" {

int sihl_conv2d_fwd_ws(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                       int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                       const float* pre_scale, const float* pre_shift, const float* post_scale,
                       const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                       long out_image_stride, void* ws, long ws_bytes, hipStream_t stream);
int sihl_conv2d_dgrad_ws(const void* dout, const void* wt_t, void* din, const void* add, int N, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws, long ws_bytes,
                         hipStream_t stream);

int sihl_conv2d_dgrad_add(const void* dout, const void* wt_t, void* din, const void* add, int add_stride, int N, int H,
                          int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws,
                          long ws_bytes, hipStream_t stream);

// Test hook: 1 = use the register-staged loader instead of LDS-DMA (both are kept parity-tested).
int sihl_conv2d_force_register_staging(int on) { g_force_reg = on != 0; return 0; }

// The ablation / tuning setters below change process-wide state and (sihl_conv2d_debug) make results INVALID: they exist
// only in `make TUNING=1` libraries.  The shipped library keeps the symbols (one binding for both builds) but stores
// nothing: a non-default request is refused with SIHL_EARG.
#ifdef SIHL_TUNING
#define SIHL_TUNING_SET(var, value) do { (var) = (value); return SIHL_OK; } while (0)
#else
#define SIHL_TUNING_SET(var, value) do { return (value) == (var) ? SIHL_OK : SIHL_EARG; } while (0)
#endif

// Tuning ablation (results invalid when non-zero): 1 = skip the in-loop DMA, 2 = skip ds_read/MFMA.
int sihl_conv2d_debug(int mode) { SIHL_TUNING_SET(g_dbg, mode); }

// Tuning hook: LDS stages of the narrow-tile LDS-DMA kernels (0 = default, 2..4).
int sihl_conv2d_nbuf_override(int n) { SIHL_TUNING_SET(g_nbuf, n); }

// Tuning hook: force the pixel-tile size of the LDS-DMA kernel (0 = heuristic, 128, 256).
int sihl_conv2d_tile_override(int bm) { g_tile_override = bm; return 0; }

// Number of (sum, sumsq) partial rows the conv may write for M output pixels (upper bound over tile sizes).
int sihl_conv2d_stat_rows(long M) { return (int)((M + BM128 - 1) / BM128); }

// fp32 scratch a conv launch may use (split-K of tiny pyramid levels); 0 when the shape never splits.
long sihl_conv2d_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long M = (long)N * Ho * Wo;
  const long small = M < (1L << 30) ? sihl_small_ws_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad, dil) : 0;
  if (((M + 127) / 128) * ((Cout + 63) / 64) > 64) return small;
  const long general = 9L * M * Cout * (long)sizeof(float);
  return general > small ? general : small;
}

// Tuning hook: switch individual dispatch rules off (bit 0: single-stage narrow tiles, bit 1: thin pointwise -> 128x128).
int sihl_conv2d_rules_off(int mask) { SIHL_TUNING_SET(g_rules_off, mask); }

// Test hook: kernel of the 3x3 convs on the small pyramid levels - 1 (default) = conv_pyr.hip where its shapes allow, else
// conv_small.hip; 2 = conv_small.hip only; 0 = the general tile kernel.  All three stay parity-tested.
int sihl_conv2d_small_enable(int on) { sihl_small_set_enabled(on != 0); sihl_pyr_set_mode(on); return 0; }
int sihl_conv2d_small_mode(void) { return sihl_pyr_get_mode(); }

// Test hook: the halo-resident 256 x 256 tile of 3x3 convs on 64-wide maps (conv_halo.hip) - 1 (default) = where its grid fills
// the chip, 2 = wherever the shape allows (small batches in tests), 0 = off (the general tile).  Both stay parity-tested.
int sihl_conv2d_halo_enable(int mode) { sihl_halo_set_mode(mode); return 0; }

// Tuning hook: 100000 * log2(group) + 1000 * min_stages + stride - the stage stride between the K-loop starts of
// neighbouring GROUPS of workgroups (default 200013: groups of 4 share a start and with it their L2 fills, stride 13;
// 0 = lockstep).
int sihl_conv2d_krot(int n) { SIHL_TUNING_SET(g_krot, n < 0 ? 0 : n); }

// Tuning / test hook: 0 disables split-K.
int sihl_conv2d_splitk_enable(int on) { g_splitk = on != 0; return 0; }

// Tuning / test hook: 0 = strided 3x3 dgrads read a zero-dilated dout (one launch) instead of four parity classes.
int sihl_conv2d_strided_classes_enable(int on) { g_strided_classes = on != 0; return 0; }

int sihl_conv2d_fwd(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                    const float* pre_scale, const float* pre_shift, const float* post_scale,
                    const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                    long out_image_stride, hipStream_t stream) {
  return sihl_conv2d_fwd_ws(in, wt, bias, out, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, dtype, act, pre_scale,
                            pre_shift, post_scale, post_shift, stats_mode, stats_ws, stats_ws_bytes, out_image_stride,
                            nullptr, 0, stream);
}

// sihl_conv2d_fwd with caller scratch (ws_bytes >= sihl_conv2d_ws_bytes(...) enables split-K; ws may be NULL).
int sihl_conv2d_fwd_ws(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                       int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                       const float* pre_scale, const float* pre_shift, const float* post_scale,
                       const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                       long out_image_stride, void* ws, long ws_bytes, hipStream_t stream) {
  if (!in || !wt || !out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
      stride <= 0 || dil <= 0 || pad < 0)
    return SIHL_EARG;
  ConvParams p;
  p.in = in; p.wt = wt; p.out = out; p.bias = bias;
  p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.post_scale = post_scale; p.post_shift = post_shift;
  p.stats = stats_ws;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return SIHL_EARG;
  const long M = (long)N * p.Ho * p.Wo;
  if (M > (1L << 30)) return SIHL_EARG;
  p.M = (int)M;
  p.act = act; p.stats_mode = stats_mode;
  p.gridM = p.gridN = 0;
  p.in_dilate = 1;
  p.dbg = g_dbg;  // read by the kernels in SIHL_TUNING builds only
  p.out_image_stride = out_image_stride > 0 ? out_image_stride : (long)p.Ho * p.Wo * Cout;
  if (p.out_image_stride < (long)p.Ho * p.Wo * Cout) return SIHL_EARG;
  if (p.out_image_stride % (dtype == SIHL_BF16 ? 8 : 4)) return SIHL_EARG;
  if (stats_mode == 1 && act != SIHL_ACT_NONE) return SIHL_EARG;  // pre-norm statistics are taken with no activation
  if (stats_mode) {
    if (!stats_ws) return SIHL_EARG;
    if (stats_ws_bytes < (long)sihl_conv2d_stat_rows(M) * 2 * Cout * (long)sizeof(float)) return SIHL_EWS;
  }
  p.splits = 1; p.partial = (float*)ws; p.partial_bytes = ws ? ws_bytes : 0;
  p.add = nullptr; p.add_stride = 1; p.add_H = p.add_W = 0;
  p.w_ntaps = KH * KW; p.w_kw = KW; p.w_ky0 = p.w_kx0 = 0; p.w_kys = p.w_kxs = 1;
  p.k_rotate = 0; p.k_rot_group = 0; p.small_nch = 0;
  p.out_s = 1; p.out_py = p.out_px = 0; p.out_W = p.Wo;
  if (dtype == SIHL_F32) return sihl_conv_dispatch_f32(p, stream);
  if (dtype == SIHL_BF16) return sihl_conv_dispatch_bf16(p, stream);
  return SIHL_EARG;
}

// Input gradient of a (possibly strided) convolution: din[N][H][W][Cin] from dout[N][Ho][Wo][Cout], where
// wt_t = sihl_weight_flip_transpose(w, flip=1) is [Cin][KH][KW][Cout].  dout is read as if zero-dilated by `stride`.
int sihl_conv2d_dgrad(const void* dout, const void* wt_t, void* din, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, hipStream_t stream) {
  return sihl_conv2d_dgrad_ws(dout, wt_t, din, nullptr, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, dtype, nullptr, 0,
                              stream);
}

// sihl_conv2d_dgrad with caller scratch (ws_bytes >= sihl_conv2d_ws_bytes(N, Ho, Wo, Cout, Cin, KH, KW, 1, ...) of the
// equivalent forward problem (output = din) enables split-K) and an optional addend: din = dgrad + add, `add` a dense
// [N][H][W][Cin] tensor of the same dtype (the identity-branch gradient of a residual block), or NULL.
int sihl_conv2d_dgrad_ws(const void* dout, const void* wt_t, void* din, const void* add, int N, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws, long ws_bytes,
                         hipStream_t stream) {
  return sihl_conv2d_dgrad_add(dout, wt_t, din, add, 1, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, dtype, ws, ws_bytes,
                               stream);
}

// sihl_conv2d_dgrad_ws whose addend may be the COMPACT input gradient of a stride-`add_stride` 1x1 projection of the
// same input ([N][ceil(H/s)][ceil(W/s)][Cin]): it is added at the pixels that projection reads, nothing elsewhere.
int sihl_conv2d_dgrad_add(const void* dout, const void* wt_t, void* din, const void* add, int add_stride, int N, int H,
                          int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, void* ws,
                          long ws_bytes, hipStream_t stream) {
  if (add_stride < 1 || (add_stride > 1 && Cin % (dtype == SIHL_BF16 ? 8 : 4) != 0)) return SIHL_EARG;
  if (!dout || !wt_t || !din || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
      stride <= 0 || dil <= 0 || pad < 0)
    return SIHL_EARG;
  const int pad_d_h = dil * (KH - 1) - pad, pad_d_w = dil * (KW - 1) - pad;
  if (pad_d_h < 0 || pad_d_h != pad_d_w) return SIHL_EARG;
  ConvParams p;
  p.in = dout; p.wt = wt_t; p.out = din; p.bias = nullptr;
  p.pre_scale = p.pre_shift = p.post_scale = p.post_shift = nullptr;
  p.stats = nullptr;
  p.N = N;
  p.H = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;  // dout spatial size
  p.W = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  if (p.H <= 0 || p.W <= 0) return SIHL_EARG;
  p.Cin = Cout; p.Cout = Cin; p.KH = KH; p.KW = KW;
  p.stride = 1; p.pad = pad_d_h; p.dil = dil;
  p.Ho = H; p.Wo = W;
  const long M = (long)N * H * W;
  if (M > (1L << 30)) return SIHL_EARG;
  p.M = (int)M;
  p.act = SIHL_ACT_NONE; p.stats_mode = 0;
  p.gridM = p.gridN = 0;
  p.in_dilate = stride;
  p.dbg = 0;
  p.out_image_stride = (long)H * W * Cin;
  p.splits = 1; p.partial = (float*)ws; p.partial_bytes = ws ? ws_bytes : 0;
  p.add = add;
  p.add_stride = add ? add_stride : 1;
  p.add_H = (H + add_stride - 1) / add_stride; p.add_W = (W + add_stride - 1) / add_stride;
  p.w_ntaps = KH * KW; p.w_kw = KW; p.w_ky0 = p.w_kx0 = 0; p.w_kys = p.w_kxs = 1;
  p.k_rotate = 0; p.k_rot_group = 0; p.small_nch = 0;
  p.out_s = 1; p.out_py = p.out_px = 0; p.out_W = W;
  // 3x3 / stride 2 / pad 1 (the strided convs of the ResNet stages): four parity classes of output pixels, each a small
  // dense conv over dout with the 1, 2, 2 or 4 weight taps that meet it (dx[2i+py] = sum_ky dout[(2i+py+1-ky)/2] w[ky]
  // over the ky that make the index whole: ky = 1 for py = 0; ky = 2 (row i) and ky = 0 (row i+1) for py = 1; in the
  // flipped dgrad weights that is tap 1, resp. taps 0 and 2) - the same MACs as the forward, where the zero-dilated
  // read below multiplies 3 zeros out of 4
  const int vs = dtype == SIHL_BF16 ? 2 : 4;
  const bool dma_ok = !g_force_reg && (long)N * p.H * p.W * Cout * vs < (1L << 31) && (long)Cin * 9 * Cout * vs < (1L << 31);
  if (stride == 2 && KH == 3 && KW == 3 && pad == 1 && dil == 1 && !add && dma_ok && g_strided_classes &&
      Cout % (16 / vs) == 0 && (dtype == SIHL_F32 || dtype == SIHL_BF16)) {
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        ConvParams q = p;
        q.in_dilate = 1;
        q.pad = 0;
        q.KH = py ? 2 : 1; q.KW = px ? 2 : 1;
        q.w_ky0 = py ? 0 : 1; q.w_kys = 2; q.w_kx0 = px ? 0 : 1; q.w_kxs = 2;
        q.Ho = (H - py + 1) / 2; q.Wo = (W - px + 1) / 2;  // output rows 2i+py < H
        if (q.Ho <= 0 || q.Wo <= 0) continue;
        q.M = N * q.Ho * q.Wo;
        q.out_s = 2; q.out_py = py; q.out_px = px; q.out_W = W;
        q.partial = nullptr; q.partial_bytes = 0;
        const int rc = dtype == SIHL_F32 ? sihl_conv_dispatch_f32(q, stream) : sihl_conv_dispatch_bf16(q, stream);
        if (rc != SIHL_OK) return rc;
      }
    return SIHL_OK;
  }
  if (dtype == SIHL_F32) return sihl_conv_dispatch_f32(p, stream);
  if (dtype == SIHL_BF16) return sihl_conv_dispatch_bf16(p, stream);
  return SIHL_EARG;
}

}  // extern "C"
