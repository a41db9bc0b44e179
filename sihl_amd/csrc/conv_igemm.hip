// NHWC implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// Replaces, for the hot path, the ATen conv + activation + norm passes behind the reference's
// ConvNormAct (src/sihl/layers/convblocks.py:37-87), torchvision Conv2dNormActivation
// (src/sihl/layers/fpn.py:26-37, heads/object_detection.py:52-55) and nn.Linear inside ops.MLP
// (heads/object_detection.py:51-61; a Linear is a 1x1 conv over rows).
//
//   out[m][co] = epilogue( sum_{ky,kx,ci} in[n, oy*s-p+ky*d, ox*s-p+kx*d, ci] * wt[co][ky][kx][ci] )
//   m = (n*Ho + oy)*Wo + ox.   GEMM view: M = N*Ho*Wo pixels, N = Cout, K = KH*KW*Cin.
//
// Tiling: one 256-thread workgroup (4 waves) owns BM=128 pixels x BN (64/128/256) output channels.
// K is walked in stages of (one tap) x (128 bytes of input channels = 64 bf16 / 32 fp32): the A tile
// (BM pixel rows, zero-filled outside the image) and the B tile (BN weight rows) are staged
// global -> registers -> LDS (rows padded to 144 B: conflict-free ds_read_b128 for the 32x32 MFMA
// operand shape), double-buffered, one barrier per stage.  bf16 uses v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation; fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32, for the 1e-4 parity configuration).
// Epilogue: bias -> [stats] -> pre-affine -> activation -> [stats] -> post-affine, the tile is
// transposed through LDS and written with 16-byte row-contiguous stores; per-channel (sum, sumsq)
// partials for BatchNorm batch statistics go to a workspace row per M-tile (deterministic, no atomics).
#include "common.h"
#include "profile.h"

namespace {

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
  int in_dilate;          // >1: the input is read as if zero-dilated by this factor (dgrad of a strided conv)
  int gridM, gridN;
};

constexpr int BM = 128;
constexpr int KCB = 128;         // bytes of K per stage per row
constexpr int LDS_STRIDE = 144;  // padded row (bytes)

template <typename T> __device__ __forceinline__ void mma_step(f32x16_t& c, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma_step<bf16_t>(f32x16_t& c, const uint4& a, const uint4& b) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_step<float>(f32x16_t& c, const uint4& a, const uint4& b) {
  // lane half h holds k = 4h..4h+3 of this 8-wide k-step: MFMA j contracts k = {j, 4+j}
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

template <typename T, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KCE = KCB / (int)sizeof(T);
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int A_BYTES = BM * LDS_STRIDE, B_BYTES = BN * LDS_STRIDE, STAGE = A_BYTES + B_BYTES;
  constexpr int NB = BN / 32;  // weight rows per loader thread
  constexpr int EPI_STRIDE = BN * (int)sizeof(T) + 16;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int L = xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int tile_m = L / p.gridN, tile_n = L % p.gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* __restrict__ in = (const T*)p.in;
  const T* __restrict__ wt = (const T*)p.wt;
  const int ntaps = p.KH * p.KW;
  const int nchunks = (p.Cin + KCE - 1) / KCE;
  const int nstages = nchunks * ntaps;

  // ---- loader geometry: thread -> (16-byte chunk lc of the 128-byte K slice, rows lr + 32*i)
  const int lc = tid & 7, lr = tid >> 3;
  int a_iy0[4], a_ix0[4];
  long a_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lr + 32 * i;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw, r = m - n * hw;
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_base[i] = (long)n * p.H * p.W * p.Cin;
    } else {
      a_iy0[i] = -(1 << 28);  // never in bounds
      a_ix0[i] = 0;
      a_base[i] = 0;
    }
  }
  long b_base[NB];
  bool b_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int co = n0 + lr + 32 * j;
    b_ok[j] = co < p.Cout;
    b_base[j] = (long)co * ntaps * p.Cin;
  }

  uint4 ra[4], rb[NB];
  auto load_regs = [&](int s) {
    const int kc = s / ntaps, tap = s - kc * ntaps;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int ch = kc * KCE + lc * VEC;
    const bool ch_ok = ch < p.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = a_iy0[i] + ky * p.dil, ix = a_ix0[i] + kx * p.dil;
      bool ok = ch_ok && iy >= 0 && ix >= 0;
      if (p.in_dilate > 1) {
        ok = ok && (iy % p.in_dilate == 0) && (ix % p.in_dilate == 0);
        iy /= p.in_dilate;
        ix /= p.in_dilate;
      }
      ok = ok && iy < p.H && ix < p.W;
      ra[i] = ok ? *(const uint4*)(in + a_base[i] + ((long)iy * p.W + ix) * p.Cin + ch) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      rb[j] = (ch_ok && b_ok[j]) ? *(const uint4*)(wt + b_base[j] + (long)tap * p.Cin + ch) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE + lr * LDS_STRIDE + lc * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(uint4*)(base + i * 32 * LDS_STRIDE) = ra[i];
#pragma unroll
    for (int j = 0; j < NB; ++j) *(uint4*)(base + A_BYTES + j * 32 * LDS_STRIDE) = rb[j];
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * LDS_STRIDE + (lane >> 5) * 16;
  auto compute = [&](int buf) {
    const char* As = smem + buf * STAGE + wm * WTM * LDS_STRIDE + frag_off;
    const char* Bs = smem + buf * STAGE + A_BYTES + wn * WTN * LDS_STRIDE + frag_off;
#pragma unroll
    for (int ks = 0; ks < KCB / 32; ++ks) {
      uint4 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const uint4*)(As + i * 32 * LDS_STRIDE + ks * 32);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *(const uint4*)(Bs + j * 32 * LDS_STRIDE + ks * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) mma_step<T>(acc[i][j], a[i], b[j]);
    }
  };

  // ---- main loop: prefetch stage s+1 into registers while stage s is multiplied out of LDS
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

  // ---- epilogue (all staging LDS is free now)
  char* epi = smem;
  float* red = (float*)(smem + BM * EPI_STRIDE);  // [2][WM][BN]
  const int half = lane >> 5;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cl = wn * WTN + j * 32 + (lane & 31);  // column inside the tile
    const int co = n0 + cl;
    const bool cok = co < p.Cout;
    const float bias = (p.bias && cok) ? p.bias[co] : 0.f;
    const float s1 = (p.pre_scale && cok) ? p.pre_scale[co] : 1.f;
    const float t1 = (p.pre_shift && cok) ? p.pre_shift[co] : 0.f;
    const float s2 = (p.post_scale && cok) ? p.post_scale[co] : 1.f;
    const float t2 = (p.post_shift && cok) ? p.post_shift[co] : 0.f;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool rok = (m0 + row) < p.M;
        float v = acc[i][j][r] + bias;
        if (p.stats_mode == 1 && rok) { ssum += v; ssq += v * v; }
        v = v * s1 + t1;
        v = apply_act(v, p.act);
        if (p.stats_mode == 2 && rok) { ssum += v; ssq += v * v; }
        v = v * s2 + t2;
        elem<T>::st((T*)(epi + row * EPI_STRIDE) + cl, v);
      }
    }
    if (p.stats_mode) {
      ssum += __shfl_xor(ssum, 32);
      ssq += __shfl_xor(ssq, 32);
      if (half == 0) {
        red[(0 * WM + wm) * BN + cl] = ssum;
        red[(1 * WM + wm) * BN + cl] = ssq;
      }
    }
  }
  __syncthreads();
  if (p.stats_mode) {
    for (int c = tid; c < BN; c += 256) {
      const int co = n0 + c;
      if (co < p.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s += red[(0 * WM + w) * BN + c]; q += red[(1 * WM + w) * BN + c]; }
        p.stats[((long)tile_m * 2 + 0) * p.Cout + co] = s;
        p.stats[((long)tile_m * 2 + 1) * p.Cout + co] = q;
      }
    }
  }
  // row-contiguous 16-byte stores
  T* __restrict__ out = (T*)p.out;
  constexpr int CHUNKS = BN * (int)sizeof(T) / 16;  // 16-byte chunks per tile row
  const bool vec_ok = (p.Cout % VEC) == 0;
  for (int idx = tid; idx < BM * CHUNKS; idx += 256) {
    const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
    const int m = m0 + row, co = n0 + ch * VEC;
    if (m >= p.M || co >= p.Cout) continue;
    const char* src = epi + row * EPI_STRIDE + ch * 16;
    const int hw_o = p.Ho * p.Wo, n_img = m / hw_o;
    T* dst = out + (long)n_img * p.out_image_stride + (long)(m - n_img * hw_o) * p.Cout + co;
    if (vec_ok && co + VEC <= p.Cout) {
      *(uint4*)dst = *(const uint4*)src;
    } else {
      for (int e = 0; e < VEC && co + e < p.Cout; ++e) dst[e] = ((const T*)src)[e];
    }
  }
}

template <typename T, int BN, int WM, int WN>
int launch(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  p.gridM = (p.M + BM - 1) / BM;
  p.gridN = (p.Cout + BN - 1) / BN;
  constexpr int STAGE = (BM + BN) * LDS_STRIDE;
  constexpr int EPI = BM * (BN * (int)sizeof(T) + 16) + 2 * WM * BN * 4;
  constexpr int LDS = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<T, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * p.KH * p.KW * p.Cin) * sizeof(T);
  sihl_prof_begin(SIHL_PROF_CONV, sizeof(T) == 2 ? SIHL_BF16 : SIHL_F32, flops, bytes, stream);
  hipLaunchKernelGGL(kern, dim3(p.gridM * p.gridN), dim3(256), LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

template <typename T>
int dispatch(const ConvParams& p, hipStream_t stream) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (p.Cin % VEC != 0) return SIHL_EARG;  // 16-byte channel vectors required; caller pads
  if (p.Cout > 128) return launch<T, 256, 2, 2>(p, stream);
  if (p.Cout > 64) return launch<T, 128, 2, 2>(p, stream);
  return launch<T, 64, 4, 1>(p, stream);
}

}  // namespace

extern "C" {

// Number of (sum, sumsq) partial rows the conv writes for M output pixels.
int sihl_conv2d_stat_rows(long M) { return (int)((M + BM - 1) / BM); }

int sihl_conv2d_fwd(const void* in, const void* wt, const float* bias, void* out, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad, int dil, int dtype, int act,
                    const float* pre_scale, const float* pre_shift, const float* post_scale,
                    const float* post_shift, int stats_mode, float* stats_ws, long stats_ws_bytes,
                    long out_image_stride, hipStream_t stream) {
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
  p.out_image_stride = out_image_stride > 0 ? out_image_stride : (long)p.Ho * p.Wo * Cout;
  if (p.out_image_stride < (long)p.Ho * p.Wo * Cout) return SIHL_EARG;
  if (p.out_image_stride % (dtype == SIHL_BF16 ? 8 : 4)) return SIHL_EARG;
  if (stats_mode) {
    if (!stats_ws) return SIHL_EARG;
    if (stats_ws_bytes < (long)sihl_conv2d_stat_rows(M) * 2 * Cout * (long)sizeof(float)) return SIHL_EWS;
  }
  if (dtype == SIHL_F32) return dispatch<float>(p, stream);
  if (dtype == SIHL_BF16) return dispatch<bf16_t>(p, stream);
  return SIHL_EARG;
}

// Input gradient of a (possibly strided) convolution: din[N][H][W][Cin] from dout[N][Ho][Wo][Cout], where
// wt_t = sihl_weight_flip_transpose(w, flip=1) is [Cin][KH][KW][Cout].  dout is read as if zero-dilated by `stride`.
int sihl_conv2d_dgrad(const void* dout, const void* wt_t, void* din, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, hipStream_t stream) {
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
  p.out_image_stride = (long)H * W * Cin;
  if (dtype == SIHL_F32) return dispatch<float>(p, stream);
  if (dtype == SIHL_BF16) return dispatch<bf16_t>(p, stream);
  return SIHL_EARG;
}

}  // extern "C"
