// Launch parameters shared by the implicit-GEMM conv kernels (conv_igemm_impl.h, conv_small.hip).
#pragma once
#include "common.h"

// Tuning ablations (sihl_conv2d_debug) are compiled into the kernels only in SIHL_TUNING builds (make TUNING=1): the
// shipped hot loops carry no debug branches.
#ifdef SIHL_TUNING
#define SIHL_DBG(p) ((p).dbg)
#else
#define SIHL_DBG(p) 0
#endif

struct ConvParams {
  const void* in;
  const void* wt;
  void* out;
  const float* bias;
  const float* pre_scale;
  const float* pre_shift;
  const float* post_scale;
  const float* post_shift;
  float* stats;  // [gridM][2][Cout] or null
  int N, H, W, Cin, Cout, KH, KW, stride, pad, dil, Ho, Wo;
  int M;
  int act, stats_mode;  // stats_mode: 0 none, 1 after bias (pre-affine), 2 after activation
  long out_image_stride;  // elements between consecutive images of the output (>= Ho*Wo*Cout)
  int dbg;                // tuning ablation: 1 = no DMA inside the loop, 2 = no MFMA/ds_read (results invalid)
  int in_dilate;          // >1: the input is read as if zero-dilated by this factor (dgrad of a strided conv)
  int gridM, gridN;
  int splits;       // > 1: split-K - grid.y splits each accumulate a slice of the K stages into `partial`
  float* partial;   // [splits][M][Cout] fp32 (caller workspace)
  long partial_bytes;
  const void* add;  // optional tensor of the output's shape added in the epilogue (dense output only): the identity
                    // branch's gradient riding on a residual block's first dgrad instead of a separate add kernel
  // Parity classes of a strided conv's dgrad (sihl_conv2d_dgrad_add, LDS-DMA kernel only): the launch walks a KHxKW
  // SUBSET of a w_kh x w_kw weight window - window tap (ky, kx) multiplies weight tap (w_ky0 + ky*w_kys, w_kx0 +
  // kx*w_kxs) - and scatters output pixel (i, j) to (i*out_s + out_py, j*out_s + out_px) of an out_W-wide image.
  int w_ntaps, w_kw, w_ky0, w_kys, w_kx0, w_kxs;  // defaults: KH*KW, KW, 0, 1, 0, 1
  int out_s, out_py, out_px, out_W;                // defaults: 1, 0, 0, Wo
  int k_rot_group;  // log2 of the number of neighbouring workgroups that share a K-loop start
  int k_rotate;     // LDS-DMA tile kernel: workgroups start their K loop at different stages (see conv_igemm_dma_kernel)
  int small_nch;    // conv_small.hip (3x3 on the small pyramid levels): channel chunks one workgroup walks
  int add_stride;   // > 1: the addend is [N][add_H][add_W][Cout] and lands on output pixels (y, x) with y % add_stride
  int add_H, add_W; // == 0 and x % add_stride == 0 only (input gradient of a strided 1x1 projection: zero elsewhere)
};

// conv_small.hip: halo-resident 3x3 kernel of the small pyramid levels (bf16)
bool sihl_small_eligible(const ConvParams& p);
int sihl_small_launch(const ConvParams& p, hipStream_t stream);
void sihl_small_set_enabled(bool on);
long sihl_small_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil);
// conv_pyr.hip: image-major 3x3 kernel of the pyramid's top levels (bf16), preferred over conv_small.hip where eligible
bool sihl_pyr_eligible(const ConvParams& p);
int sihl_pyr_launch(const ConvParams& p, hipStream_t stream);
void sihl_pyr_set_mode(int mode);
int sihl_pyr_get_mode();
// conv_halo.hip: 256 x 256 tile with a halo-resident input patch for 3x3 convs on 64-wide maps (bf16)
bool sihl_halo_eligible(const ConvParams& p);
int sihl_halo_launch(const ConvParams& p, hipStream_t stream);
void sihl_halo_set_mode(int mode);
// conv_igemm_bf16.hip: the finishing launch of a split-K conv (sums the fp32 slices, runs the epilogue), for conv_small.hip
int sihl_conv_splitk_finish_bf16(const ConvParams& p, hipStream_t stream);
