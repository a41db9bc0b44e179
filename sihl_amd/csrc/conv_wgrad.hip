// Weight gradient of the NHWC convolution (and of nn.Linear = 1x1 over rows) on the matrix cores.
//
//   dW[co][ky][kx][ci] = sum_m dout[m][co] * in[n, oy*s-p+ky*d, ox*s-p+kx*d, ci]       m = (n*Ho+oy)*Wo+ox
//
// This is the reduction autograd performs for Conv2d.weight / Linear.weight behind the reference's
// ConvNormAct / Conv2dNormActivation / ops.MLP (see conv_igemm.hip for the forward citations).
// GEMM view per tap: [Cout x M] . [M x Cin]; the contraction runs over PIXELS, which in NHWC is
// the slow axis of both operands.  Both tiles are therefore staged as [pixel][channel] rows and the
// MFMA operands (k-contiguous per lane) come from TRANSPOSED LDS reads: ds_read_b64_tr_b16 for bf16,
// plain ds_read_b32 down a column for fp32 (v_mfma_f32_32x32x2_f32 takes one float per lane).
// Workgroup = 128 co x 128 ci of one tap over one K-split of the pixels; fp32 partial tiles go to
// a workspace [split][Cout][taps][Cin] and sihl_wgrad_reduce sums the splits (deterministic).
#include "common.h"
#include "profile.h"

namespace {

struct WgradParams {
  const void* in;
  const void* dout;
  float* ws;  // [splits][Cout][taps][Cin]
  int N, H, W, Cin, Cout, KH, KW, stride, pad, dil, Ho, Wo;
  int M, splits, m_per_split;
  int tiles_co, tiles_ci;
};

constexpr int BCO = 128, BCI = 128;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

template <typename T> struct wg_cfg;
template <> struct wg_cfg<bf16_t> {
  static constexpr int KP = 64;            // pixels per stage
  static constexpr int ROWB = BCO * 2 + 64;  // 4 consecutive pixel rows land on disjoint banks
};
template <> struct wg_cfg<float> {
  static constexpr int KP = 32;
  static constexpr int ROWB = BCO * 4;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KP = wg_cfg<T>::KP, ROWB = wg_cfg<T>::ROWB;
  constexpr int CPR = BCO * (int)sizeof(T) / 16;  // 16-byte chunks per tile row (16 bf16 / 32 fp32)
  constexpr int RPT = KP * CPR / 256;             // rows per loader thread (=4)
  constexpr int RSTEP = 256 / CPR;
  constexpr int TILE = KP * ROWB, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, each 64 co x 64 ci
  const int ntaps = p.KH * p.KW;
  int b = blockIdx.x;
  const int split = b % p.splits; b /= p.splits;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int tap = b;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int k_begin = split * p.m_per_split;
  const int k_end = min(p.M, k_begin + p.m_per_split);
  const int nstages = (k_end - k_begin + KP - 1) / KP;

  const T* __restrict__ in = (const T*)p.in;
  const T* __restrict__ dout = (const T*)p.dout;
  const int lc = tid % CPR, lr = tid / CPR;
  const bool co_ok = co0 + lc * VEC < p.Cout;
  const bool ci_ok = ci0 + lc * VEC < p.Cin;
  const int hw = p.Ho * p.Wo;

  uint4 ra[RPT], rb[RPT];
  auto load_regs = [&](int s) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int m = k_begin + s * KP + lr + RSTEP * i;
      const bool mok = m < k_end;
      ra[i] = (mok && co_ok) ? *(const uint4*)(dout + (long)m * p.Cout + co0 + lc * VEC) : make_uint4(0, 0, 0, 0);
      bool ok = mok && ci_ok;
      long off = 0;
      if (ok) {
        const int n = m / hw, r = m - n * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
        ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        off = (((long)n * p.H + iy) * p.W + ix) * p.Cin + ci0 + lc * VEC;
      }
      rb[i] = ok ? *(const uint4*)(in + off) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE + lr * ROWB + lc * 16;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      *(uint4*)(base + i * RSTEP * ROWB) = ra[i];
      *(uint4*)(base + TILE + i * RSTEP * ROWB) = rb[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const char* As = smem + buf * STAGE;
    const char* Bs = As + TILE;
    if constexpr (sizeof(T) == 2) {
      // transposed read: 16-lane group g reads a block of 4 pixel rows x 16 channels; lane 4q+p gives
      // the address of row q, channels 4p..4p+3 and receives channel (lane&15) of the 4 rows.
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g >> 1;
      const int row_off = (8 * h + q) * ROWB + (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < KP / 16; ++ks) {
        bf16x8_t a[2], bq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* pa = As + ks * 16 * ROWB + row_off + (wm * 64 + t * 32) * 2;
          const char* pb = Bs + ks * 16 * ROWB + row_off + (wn * 64 + t * 32) * 2;
          s16x4_t a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
          s16x4_t a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + 4 * ROWB));
          s16x4_t b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
          s16x4_t b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + 4 * ROWB));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          s16x8_t av = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
          s16x8_t bv = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
          a[t] = __builtin_bit_cast(bf16x8_t, av);
          bq[t] = __builtin_bit_cast(bf16x8_t, bv);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int i32 = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < KP / 2; ++kk) {
        float a[2], bq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t] = *(const float*)(As + (2 * kk + h) * ROWB + (wm * 64 + t * 32 + i32) * 4);
          bq[t] = *(const float*)(Bs + (2 * kk + h) * ROWB + (wn * 64 + t * 32 + i32) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if (nstages > 0) {
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const bool more = s + 1 < nstages;
      if (more) load_regs(s + 1);
      compute(s & 1);
      if (more) store_lds((s + 1) & 1);
      __syncthreads();
    }
  }

  // fp32 partial tile -> workspace
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          p.ws[(((long)split * p.Cout + co) * ntaps + tap) * p.Cin + ci] = acc[i][j][r];
      }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits,
                                    int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws[(long)k * n + i];
  dw[i] = accumulate ? dw[i] + s : s;
}

template <typename T>
int launch(WgradParams p, hipStream_t stream) {
  constexpr int LDS = 2 * 2 * wg_cfg<T>::KP * wg_cfg<T>::ROWB;
  static bool attr_set = false;
  auto kern = conv_wgrad_kernel<T>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = p.KH * p.KW * p.tiles_co * p.tiles_ci * p.splits;
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout) * sizeof(T) + (double)p.Cout * p.KH * p.KW * p.Cin * 4.0;
  sihl_prof_begin(SIHL_PROF_WGRAD, sizeof(T) == 2 ? SIHL_BF16 : SIHL_F32, flops, bytes, stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int choose_splits(long M, int tiles, int kp) {
  // aim for >= 512 workgroups, each with at least 4 stages of work
  long want = (512 + tiles - 1) / tiles;
  long max_by_work = M / (4L * kp);
  if (max_by_work < 1) max_by_work = 1;
  long s = want < max_by_work ? want : max_by_work;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

}  // namespace

extern "C" {

// Workspace bytes sihl_conv2d_wgrad needs for this problem.
long sihl_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int dil, int dtype) {
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long M = (long)N * Ho * Wo;
  const int tiles = KH * KW * ((Cout + BCO - 1) / BCO) * ((Cin + BCI - 1) / BCI);
  const int splits = choose_splits(M, tiles, dtype == SIHL_BF16 ? 64 : 32);
  return (long)splits * Cout * KH * KW * Cin * (long)sizeof(float);
}

// dw: fp32 [Cout][KH][KW][Cin]; accumulate != 0 adds into dw instead of overwriting.
int sihl_conv2d_wgrad(const void* in, const void* dout, float* dw, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, int accumulate, void* ws, long ws_bytes,
                      hipStream_t stream) {
  if (!in || !dout || !dw || !ws || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SIHL_EARG;
  const int vec = dtype == SIHL_BF16 ? 8 : 4;
  if (Cin % vec || Cout % vec) return SIHL_EARG;
  WgradParams p;
  p.in = in; p.dout = dout; p.ws = (float*)ws;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long M = (long)N * p.Ho * p.Wo;
  if (p.Ho <= 0 || p.Wo <= 0 || M > (1L << 30)) return SIHL_EARG;
  p.M = (int)M;
  p.tiles_co = (Cout + BCO - 1) / BCO;
  p.tiles_ci = (Cin + BCI - 1) / BCI;
  const int kp = dtype == SIHL_BF16 ? 64 : 32;
  p.splits = choose_splits(M, KH * KW * p.tiles_co * p.tiles_ci, kp);
  p.m_per_split = (int)(((M + p.splits - 1) / p.splits + kp - 1) / kp * kp);
  const long n = (long)Cout * KH * KW * Cin;
  if (ws_bytes < (long)p.splits * n * (long)sizeof(float)) return SIHL_EWS;
  int rc;
  if (dtype == SIHL_F32) rc = launch<float>(p, stream);
  else if (dtype == SIHL_BF16) rc = launch<bf16_t>(p, stream);
  else return SIHL_EARG;
  if (rc) return rc;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (const float*)ws, dw, n, p.splits, accumulate);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
