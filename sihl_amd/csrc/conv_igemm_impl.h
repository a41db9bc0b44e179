// NHWC implicit-GEMM convolution on the CDNA4 matrix cores (gfx950): kernels, launchers and the tile dispatch.
// Included by conv_igemm_bf16.hip and conv_igemm_f32.hip (one translation unit per element type: the ~60 kernel
// instantiations compile in parallel); the C-ABI entry points are in conv_igemm.hip.
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
// operand shape), double-buffered, one barrier per stage.  bf16 uses v_mfma_f32_16x16x32_bf16 in the LDS-DMA kernels
// (v_mfma_f32_32x32x16_bf16 in the register-staged fallback) with fp32 accumulation; fp32 uses v_mfma_f32_32x32x2_f32
// (exact fp32, for the 1e-4 parity configuration).
// Epilogue: bias -> [stats] -> pre-affine -> activation -> [stats] -> post-affine, the tile is
// transposed through LDS and written with 16-byte row-contiguous stores; per-channel (sum, sumsq)
// partials for BatchNorm batch statistics go to a workspace row per M-tile (deterministic, no atomics).
#pragma once
#include "common.h"
#include <type_traits>
#include "dma.h"
#include "conv_params.h"
#include "conv_tuning.h"
#include "profile.h"

namespace {


// element offset of output row m in the addend, or -1 where a strided addend has nothing to add
__device__ __forceinline__ long add_offset(const ConvParams& p, int m) {
  if (p.add_stride <= 1) return (long)m * p.Cout;
  const int hw = p.Ho * p.Wo;
  const int n = m / hw, r = m - n * hw;
  const int y = r / p.Wo, x = r - y * p.Wo;
  if ((y % p.add_stride) != 0 || (x % p.add_stride) != 0) return -1;
  return (((long)n * p.add_H + y / p.add_stride) * p.add_W + x / p.add_stride) * p.Cout;
}

constexpr int KCB = 128;         // bytes of K per stage per row
constexpr int LDS_STRIDE = 144;  // padded row (bytes)

// MFMA output-tile geometry.  The bf16 LDS-DMA kernels multiply with v_mfma_f32_16x16x32_bf16 (TILE 16: 4 accumulator
// registers per tile, lane -> column lane % 16, rows 4 * (lane / 16) + r); fp32 (exact v_mfma_f32_32x32x2_f32) and the
// register-staged fallback use 32 x 32 tiles (16 registers: column lane % 32, rows 8 * (r / 4) + 4 * (lane / 32) + r % 4).
// In both, register r of lane group g = lane / TILE is row (r & 3) + (256 / TILE) * (r >> 2) + 4 * g of the tile, and
// registers r, r + 1 (r even) are neighbouring rows.  Why 16x16x32: the same flops per ds_read_b128 and per cycle, but the
// chip holds a higher clock under it (MI355X_MICROARCH.md, bare-loop 1.12-1.15 x); measured on this loop: L3 3x3
// 154 -> 144 us, MLP linears 56 -> 48 (profiles/r02_mfma_shape_timing.txt was the timing-only preview).
template <int TILE> struct AccTile { typedef float type __attribute__((ext_vector_type(TILE * TILE / 64))); };
template <int TILE> __device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + (256 / TILE) * (r >> 2); }

template <typename T, int TILE> __device__ __forceinline__ void mma_step(typename AccTile<TILE>::type& c, const uint4& a, const uint4& b) {
  if constexpr (sizeof(T) == 2 && TILE == 16) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else if constexpr (sizeof(T) == 2) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  } else {
    static_assert(TILE == 32, "fp32: 32x32x2 MFMA");
    // lane half h holds k = 4h..4h+3 of this 8-wide k-step: MFMA j contracts k = {j, 4+j}
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
}
// Shared epilogue: bias -> [stats] -> pre-affine -> act -> [stats] -> post-affine, LDS transpose, row stores.
// ACT and STATS are compile-time inside the element loop (a runtime switch there costs an expf per element).
template <typename T, int BM, int BN, int WM, int WN, int ACT, int STATS, bool ADD = false, int TILE = 32>
__device__ __forceinline__ void conv_epilogue_body(const ConvParams& p,
                                                   typename AccTile<TILE>::type (&acc)[BM / WM / TILE][BN / WN / TILE],
                                                   char* smem, int tile_m, int m0, int n0) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / TILE, NT = WTN / TILE, REGS = TILE * TILE / 64;
  constexpr int EPI_STRIDE = BN * (int)sizeof(T) + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  char* epi = smem;
  float* red = (float*)(smem + BM * EPI_STRIDE);  // [2][WM][BN]
  const int grp = lane / TILE;  // lane group: rows 4 * grp + ... of every tile
  const bool has_pre = p.pre_scale != nullptr, has_post = p.post_scale != nullptr;
  // The element loop is VALU-bound on store-heavy layers (64 elements per lane at ~10 instructions each: a thin
  // pointwise conv spent more time here than loading, multiplying and storing).  The common training launch - no
  // affines, every row of the tile inside M - takes a path with the bias, the activation, the statistics and the
  // conversion only.
  const bool lean = !has_pre && m0 + BM <= p.M;
  if (lean) {
    // (the post-affine of an inference launch - BatchNorm folded behind the activation - stays: one FMA)
    // bf16: TWO rows at a time (accumulator registers r, r + 1 are neighbouring rows of one column): packed fp32
    // add / FMA for the bias, the statistics and the post-affine, one v_cvt_pk_bf16_f32 per pair, the two halves stored
    // with ds_write_b16 / ds_write_b16_d16_hi - 1.5-2 VALU instructions per output instead of 4-5.  Every VALU
    // instruction costs a wave 4 cycles and this loop runs 64-128 outputs per lane, so on store-heavy layers it, not
    // HBM, set the pace (tools/pw_ablate.py).  The statistics are summed as (even rows, odd rows) pairs and folded at the
    // end: the same fp32 partial sums in another order.
    auto body = [&](auto post_tag) {
      constexpr bool POST = decltype(post_tag)::value;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int cl = wn * WTN + j * TILE + (lane & (TILE - 1));
        const bool cok = n0 + cl < p.Cout;
        const float bias = (p.bias && cok) ? p.bias[n0 + cl] : 0.f;
        const float s2 = (POST && cok) ? p.post_scale[n0 + cl] : 1.f;
        const float t2 = (POST && p.post_shift && cok) ? p.post_shift[n0 + cl] : 0.f;
        float ssum = 0.f, ssq = 0.f;
        if constexpr (sizeof(T) == 2) {
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          f32x2_t sum2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
          const f32x2_t b2 = {bias, bias}, s2v = {s2, s2}, t2v = {t2, t2};
          char* col = epi + (wm * WTM + 4 * grp) * EPI_STRIDE + cl * 2;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < REGS; r += 2) {
              const int row = i * TILE + acc_row<TILE>(r);  // (+ wm*WTM + 4*grp in `col`); register r + 1 is row + 1
              f32x2_t v = {acc[i][j][r], acc[i][j][r + 1]};
              v += b2;  // (a bias-free instantiation would save half an instruction per output and double the epilogue code)
              if (STATS == 1) { sum2 += v; sq2 = __builtin_elementwise_fma(v, v, sq2); }
              if (ACT == SIHL_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
              else if (ACT == SIHL_ACT_SILU) { v.x = v.x / (1.f + expf(-v.x)); v.y = v.y / (1.f + expf(-v.y)); }
              else if (ACT == SIHL_ACT_SIGMOID) { v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y)); }
              if (STATS == 2) { sum2 += v; sq2 = __builtin_elementwise_fma(v, v, sq2); }
              if (POST) v = __builtin_elementwise_fma(v, s2v, t2v);
              const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
              *(unsigned short*)(col + row * EPI_STRIDE) = (unsigned short)pk;
              *(unsigned short*)(col + (row + 1) * EPI_STRIDE) = (unsigned short)(pk >> 16);
            }
          }
          ssum = sum2.x + sum2.y;
          ssq = sq2.x + sq2.y;
        } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
          for (int r = 0; r < REGS; ++r) {
            const int row = wm * WTM + i * TILE + acc_row<TILE>(r) + 4 * grp;
            float v = acc[i][j][r] + bias;
            if (STATS == 1) { ssum += v; ssq += v * v; }
            if (ACT == SIHL_ACT_RELU) v = fmaxf(v, 0.f);
            else if (ACT == SIHL_ACT_SILU) v = v / (1.f + expf(-v));
            else if (ACT == SIHL_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
            if (STATS == 2) { ssum += v; ssq += v * v; }
            if (POST) v = v * s2 + t2;
            elem<T>::st((T*)(epi + row * EPI_STRIDE) + cl, v);
          }
        }
        }
        if (STATS) {
#pragma unroll
          for (int o = TILE; o < 64; o <<= 1) {  // fold the lane groups (same column, other rows)
            ssum += __shfl_xor(ssum, o);
            ssq += __shfl_xor(ssq, o);
          }
          if (grp == 0) {
            red[(0 * WM + wm) * BN + cl] = ssum;
            red[(1 * WM + wm) * BN + cl] = ssq;
          }
        }
      }
    };
    if (has_post) body(std::true_type{});
    else body(std::false_type{});
  } else
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cl = wn * WTN + j * TILE + (lane & (TILE - 1));  // column inside the tile
    const int co = n0 + cl;
    const bool cok = co < p.Cout;
    const float bias = (p.bias && cok) ? p.bias[co] : 0.f;
    const float s1 = (has_pre && cok) ? p.pre_scale[co] : 1.f;
    const float t1 = (has_pre && p.pre_shift && cok) ? p.pre_shift[co] : 0.f;
    const float s2 = (has_post && cok) ? p.post_scale[co] : 1.f;
    const float t2 = (has_post && p.post_shift && cok) ? p.post_shift[co] : 0.f;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < REGS; ++r) {
        const int row = wm * WTM + i * TILE + acc_row<TILE>(r) + 4 * grp;
        float v = acc[i][j][r] + bias;
        if (STATS == 1) { const float m = (m0 + row) < p.M ? v : 0.f; ssum += m; ssq += m * m; }
        v = v * s1 + t1;
        if (ACT == SIHL_ACT_RELU) v = fmaxf(v, 0.f);
        else if (ACT == SIHL_ACT_SILU) v = v / (1.f + expf(-v));
        else if (ACT == SIHL_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        if (STATS == 2) { const float m = (m0 + row) < p.M ? v : 0.f; ssum += m; ssq += m * m; }
        v = v * s2 + t2;
        elem<T>::st((T*)(epi + row * EPI_STRIDE) + cl, v);
      }
    }
    if (STATS) {
#pragma unroll
      for (int o = TILE; o < 64; o <<= 1) {
        ssum += __shfl_xor(ssum, o);
        ssq += __shfl_xor(ssq, o);
      }
      if (grp == 0) {
        red[(0 * WM + wm) * BN + cl] = ssum;
        red[(1 * WM + wm) * BN + cl] = ssq;
      }
    }
  }
  __syncthreads();
  if (STATS) {
    // one partial row per 128-pixel sub-tile, whatever BM is: row = tile_m * (BM/128) + half
    constexpr int HALVES = BM / 128, WPH = WM / HALVES;  // waves (along M) per 128-pixel half
    const int nrows = (p.M + 127) / 128;
    for (int idx = tid; idx < BN * HALVES; idx += NTHREADS) {
      const int c = idx % BN, hf = idx / BN;
      const int co = n0 + c, srow = tile_m * HALVES + hf;
      if (co < p.Cout && srow < nrows) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WPH; ++w) {
          s += red[(0 * WM + hf * WPH + w) * BN + c];
          q += red[(1 * WM + hf * WPH + w) * BN + c];
        }
        p.stats[((long)srow * 2 + 0) * p.Cout + co] = s;
        p.stats[((long)srow * 2 + 1) * p.Cout + co] = q;
      }
    }
  }
  // row-contiguous 16-byte stores
  T* __restrict__ out = (T*)p.out;
  constexpr int CHUNKS = BN * (int)sizeof(T) / 16;  // 16-byte chunks per tile row
  const bool vec_ok = (p.Cout % VEC) == 0;
  const int hw_o = p.Ho * p.Wo;
  if constexpr (ADD) if (vec_ok) {
    // + addend (dense output): the addend chunks of U rows are requested together, THEN added and stored - one global
    // round trip per U chunks instead of one per chunk (the naive loop made this epilogue latency-bound).  The U
    // chunks cost 4 U registers on top of the main loop's peak, which is why this path is its own kernel
    // instantiation (ADD): compiled into every kernel it took a wave per SIMD from all of them.
    constexpr int ITER = BM * CHUNKS / NTHREADS, U = ITER < 8 ? ITER : 8;
    static_assert(BM * CHUNKS % NTHREADS == 0 && ITER % U == 0, "tile / thread-count mismatch");
    const T* __restrict__ addp = (const T*)p.add;
    for (int it0 = 0; it0 < ITER; it0 += U) {
      uint4 av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = tid + (it0 + u) * NTHREADS;
        const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
        const int m = m0 + row, co = n0 + ch * VEC;
        const long aoff = m < p.M ? add_offset(p, m) : -1;
        av[u] = (aoff >= 0 && co + VEC <= p.Cout) ? *(const uint4*)(addp + aoff + co) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = tid + (it0 + u) * NTHREADS;
        const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
        const int m = m0 + row, co = n0 + ch * VEC;
        if (m >= p.M || co + VEC > p.Cout) continue;
        float a[VEC], c[VEC];
        unpack16(*(const uint4*)(epi + row * EPI_STRIDE + ch * 16), a, T());
        unpack16(av[u], c, T());
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += c[e];
        *(uint4*)(out + (long)m * p.Cout + co) = pack16(a, T());
      }
    }
    return;
  }
  if (vec_ok && p.out_s == 1 && p.out_image_stride == (long)hw_o * p.Cout) {
    // dense output: row m starts at m * Cout (no per-chunk division by the image size)
    for (int idx = tid; idx < BM * CHUNKS; idx += NTHREADS) {
      const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
      const int m = m0 + row, co = n0 + ch * VEC;
      if (m >= p.M || co + VEC > p.Cout) continue;
      *(uint4*)(out + (long)m * p.Cout + co) = *(const uint4*)(epi + row * EPI_STRIDE + ch * 16);
    }
    return;
  }
  for (int idx = tid; idx < BM * CHUNKS; idx += NTHREADS) {
    const int row = idx / CHUNKS, ch = idx - row * CHUNKS;
    const int m = m0 + row, co = n0 + ch * VEC;
    if (m >= p.M || co >= p.Cout) continue;
    const char* src = epi + row * EPI_STRIDE + ch * 16;
    const int n_img = m / hw_o;
    long pix = m - n_img * hw_o;
    if (p.out_s > 1) {  // parity class of a strided dgrad: (i, j) -> (i*s + py, j*s + px)
      const int i = (int)pix / p.Wo, j = (int)pix - i * p.Wo;
      pix = (long)(i * p.out_s + p.out_py) * p.out_W + j * p.out_s + p.out_px;
    }
    T* dst = out + (long)n_img * p.out_image_stride + pix * p.Cout + co;
    if (vec_ok && co + VEC <= p.Cout) {
      *(uint4*)dst = *(const uint4*)src;
    } else {
      for (int e = 0; e < VEC && co + e < p.Cout; ++e) {
        float v = elem<T>::ld((const T*)src + e);
        if (ADD) { const long aoff = add_offset(p, m); if (aoff >= 0) v += elem<T>::ld((const T*)p.add + aoff + co + e); }
        elem<T>::st(dst + e, v);
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int TILE = 32>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p,
                                              typename AccTile<TILE>::type (&acc)[BM / WM / TILE][BN / WN / TILE], char* smem,
                                              int tile_m, int m0, int n0) {
  // the combinations the hot path uses get their own straight-line body; the rest share the generic ones
  if (p.stats_mode == 0) {
    if (p.act == SIHL_ACT_NONE) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_NONE, 0, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else if (p.act == SIHL_ACT_RELU) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_RELU, 0, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else if (p.act == SIHL_ACT_SILU) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_SILU, 0, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_SIGMOID, 0, false, TILE>(p, acc, smem, tile_m, m0, n0);
  } else if (p.stats_mode == 1) {
    conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_NONE, 1, false, TILE>(p, acc, smem, tile_m, m0, n0);  // conv -> BN -> act
  } else {
    if (p.act == SIHL_ACT_RELU) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_RELU, 2, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else if (p.act == SIHL_ACT_NONE) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_NONE, 2, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else if (p.act == SIHL_ACT_SILU) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_SILU, 2, false, TILE>(p, acc, smem, tile_m, m0, n0);
    else conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_SIGMOID, 2, false, TILE>(p, acc, smem, tile_m, m0, n0);
  }
}

template <typename T, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
  constexpr int BM = BM128;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KCE = KCB / (int)sizeof(T);
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int A_BYTES = BM * LDS_STRIDE, B_BYTES = BN * LDS_STRIDE, STAGE = A_BYTES + B_BYTES;
  constexpr int NB = BN / 32;  // weight rows per loader thread
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
        for (int j = 0; j < NT; ++j) mma_step<T, 32>(acc[i][j], a[i], b[j]);
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

  conv_epilogue<T, BM, BN, WM, WN>(p, acc, smem, tile_m, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the main loop (the default): tiles go global -> LDS directly with
// buffer_load_dwordx4 ... lds (no VGPR staging, no ds_write pass).  One wave-instruction writes 1 KiB of LDS
// linearly = 8 rows x 128 B, so rows are unpadded and bank conflicts are removed by an XOR swizzle applied
// on the SOURCE side: LDS position (row, pos) holds global 16-byte chunk  pos ^ ((row >> 1) & 7)  of that
// row, and the fragment reads apply the same XOR (16 lanes of a ds_read_b128 group then hit 16 distinct
// 4-bank slots).  Out-of-image taps, rows beyond M, channels beyond Cin and weight rows beyond Cout are
// given an out-of-range buffer offset: the hardware bounds check returns zeros, which land in LDS.

// DIL: strided-dgrad instantiation (input read as if zero-dilated); the common case compiles without it.
// NBUF LDS stages: NBUF - 1 stages of DMA are in flight while one is multiplied.  Two suffice for the 256x256 tile
// (a stage of MFMAs outlasts a DMA round trip); the narrow tiles of small / thin layers were bound by one DMA
// latency per 64-deep stage and take 3-4.

// waves per SIMD the tile is tuned for (its workgroups share a CU to hide each other's load -> multiply -> store phases):
// passed to __launch_bounds__ so that an epilogue change cannot silently cost a resident workgroup (the packed bf16
// epilogue did: 128x128 went from 160 to 171 registers, 3 -> 2 workgroups per CU)
template <int BM, int BN, int NW = 8> constexpr int conv_min_waves() {
  return (BM == 128 && BN == 64) ? 5 : (BM == 128 && BN == 128) ? 3 : (BM == 256) ? (NW == 16 ? 4 : 2) : 1;
}

template <typename T, int BM, int BN, int WM, int WN, bool DIL, int NBUF, bool ADD = false>
__global__ __launch_bounds__(WM * WN * 64, (conv_min_waves<BM, BN, WM * WN>())) void conv_igemm_dma_kernel(const ConvParams p) {
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KCE = KCB / (int)sizeof(T);
  constexpr int TILE = sizeof(T) == 2 ? 16 : 32;  // MFMA tile (see AccTile)
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / TILE, NT = WTN / TILE, REGS = TILE * TILE / 64;
  constexpr int A_BYTES = BM * KCB, B_BYTES = BN * KCB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM * 8 / NTHREADS, NBL = BN * 8 / NTHREADS;  // 16-byte slots per thread per stage
  constexpr int NKS = KCB / (TILE == 16 ? 64 : 32);                // k-steps per stage (one ds_read_b128 per lane and k-step)
  constexpr int PER_STAGE = NA + NBL;                              // DMA instructions per thread per stage
  static_assert(NA >= 1 && NBL >= 1, "tile too small for the thread count");
  static_assert(NBUF >= 1 && NBUF <= 4 && (NBUF < 2 || (NBUF - 2) * PER_STAGE <= 63), "vmcnt immediate range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int L = xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int tile_m = L / p.gridN, tile_n = L % p.gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ntaps = p.KH * p.KW;
  const int nchunks = (p.Cin + KCE - 1) / KCE;
  // split-K (tiny pyramid levels): this workgroup multiplies the stages [s_first, s_first + nstages)
  const int total_stages = nchunks * ntaps;
  const int per_split = (total_stages + p.splits - 1) / p.splits;
  // k_rotate (unsplit launches): workgroup L starts its K loop at stage k_rotate * L mod total and wraps around.  Every workgroup
  // reads the SAME weight panel, and workgroups started together walk it in lockstep: each 32 KiB weight stage is then
  // requested by all of an XCD's workgroups at once.  Rotated starts spread the panel's lines over the L2 channels at any
  // moment (the fp32 sum of a tile is taken in another stage order: deterministic, tile by tile).
  const int s_first = p.k_rotate ? (int)((((unsigned)L >> p.k_rot_group) * (unsigned)p.k_rotate) % (unsigned)total_stages) : (int)blockIdx.y * per_split;
  const int nstages = p.k_rotate ? total_stages : min(total_stages, s_first + per_split) - s_first;

  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * p.H * p.W * p.Cin * (long)sizeof(T)));
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * p.w_ntaps * p.Cin * (long)sizeof(T)));
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;

  // ---- per-thread slot geometry (fixed for the whole K loop).  Per K stage only a scalar delta is added:
  //   voff = slot_offset + tap_delta(ky,kx) + kc*128      valid iff bit `tap` of the slot's tap mask is set
  // (the mask folds image borders and rows beyond M; tensors are < 2 GiB, so 0x80000000 + anything is
  // out of range for the buffer bounds check and comes back as zeros).
  unsigned a_off[NA], a_mask[NA];
  int a_ch[NA];
  int a_iy0[NA], a_ix0[NA];  // DIL only
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int q = (wave * NA + j) * 64 + lane, row = q >> 3, pos = q & 7;
    a_ch[j] = (pos ^ ((row >> 1) & 7)) * VEC;
    const int m = m0 + row;
    a_off[j] = 0;
    a_mask[j] = 0;
    a_iy0[j] = a_ix0[j] = 0;
    if (m < p.M && !DIL && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0) {
      // pointwise conv: output pixel m reads input pixel m, no borders (skips two divisions + the tap loop per slot)
      a_mask[j] = 1u;
      a_off[j] = (unsigned)(((long)m * p.Cin + a_ch[j]) * (long)sizeof(T));
    } else if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw, r = m - n * hw;
      const int oy = r / p.Wo, ox = r - oy * p.Wo;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      a_iy0[j] = iy0;
      a_ix0[j] = ix0;
      int t = 0;
      for (int ky = 0; ky < p.KH; ++ky) {
        for (int kx = 0; kx < p.KW; ++kx, ++t) {
          int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
          bool ok = iy >= 0 && ix >= 0;
          if (DIL) {
            ok = ok && (iy % p.in_dilate == 0) && (ix % p.in_dilate == 0);
            iy /= p.in_dilate;
            ix /= p.in_dilate;
          }
          if (ok && iy < p.H && ix < p.W) a_mask[j] |= 1u << t;
        }
      }
      const long pix0 = DIL ? 0 : ((long)iy0 * p.W + ix0);
      a_off[j] = (unsigned)(((long)n * p.H * p.W * p.Cin + pix0 * p.Cin + a_ch[j]) * (long)sizeof(T));
    }
  }
  unsigned b_off[NBL];
  int b_ch[NBL];
#pragma unroll
  for (int j = 0; j < NBL; ++j) {
    const int q = (wave * NBL + j) * 64 + lane, row = q >> 3, pos = q & 7;
    b_ch[j] = (pos ^ ((row >> 1) & 7)) * VEC;
    const int co = n0 + row;
    b_off[j] = co < p.Cout ? (unsigned)(((long)co * p.w_ntaps * p.Cin + b_ch[j]) * (long)sizeof(T)) : 0x80000000u;
  }
  const bool cin_full = (p.Cin % KCE) == 0;

  // scalar state of the stage being prefetched
  int n_kc = 0, n_tap = 0, n_adelta = 0, n_bdelta = 0, n_ky = 0, n_kx = 0;
  unsigned n_abase = 0, n_bbase = 0, n_tapbit = 1;
  // stages are visited in order (tap fastest, then channel chunk): counters advance by one instead of dividing
  // (two runtime integer divisions per stage were ~0.25 us of dependent VALU latency in a loop whose useful work
  // on a narrow tile is ~0.5 us)
  int c_kc = s_first / ntaps, c_tap = s_first % ntaps - 1, c_ky = 0, c_kx = -1;  // state "one before s_first"
  if (c_tap >= 0) { c_ky = c_tap / p.KW; c_kx = c_tap - c_ky * p.KW; }
  auto stage_setup = [&](int /*s*/, int buf) {
    if (++c_tap == ntaps) { c_tap = 0; c_ky = 0; c_kx = 0; if (++c_kc == nchunks) c_kc = 0; }
    else if (++c_kx == p.KW) { c_kx = 0; ++c_ky; }
    n_kc = c_kc; n_tap = c_tap; n_ky = c_ky; n_kx = c_kx;
    n_tapbit = 1u << n_tap;
    n_adelta = ((n_ky * p.dil * p.W + n_kx * p.dil) * p.Cin) * (int)sizeof(T) + n_kc * KCB;
    n_bdelta = ((p.w_ky0 + n_ky * p.w_kys) * p.w_kw + p.w_kx0 + n_kx * p.w_kxs) * p.Cin * (int)sizeof(T) + n_kc * KCB;
    n_abase = lds_base + buf * STAGE + wave * NA * 1024;
    n_bbase = lds_base + buf * STAGE + A_BYTES + wave * NBL * 1024;
  };
  auto issue_a = [&](int j) {
    if ((SIHL_DBG(p) & 128) && n_tap != 0) return;  // tuning ablation: the input tile is fetched for the first tap only
    bool ok = (a_mask[j] & n_tapbit) != 0;
    if (!cin_full) ok = ok && (n_kc * KCE + a_ch[j] < p.Cin);
    unsigned voff;
    if (!DIL) {
      voff = a_off[j] + (unsigned)n_adelta;
    } else {
      const int iy = (a_iy0[j] + n_ky * p.dil) / p.in_dilate, ix = (a_ix0[j] + n_kx * p.dil) / p.in_dilate;
      voff = a_off[j] + (unsigned)(((iy * p.W + ix) * p.Cin) * (int)sizeof(T) + n_kc * KCB);
    }
    dma16(ok ? voff : OOB, n_abase + j * 1024, in_rsrc);
  };
  auto issue_b = [&](int j) {
    if ((SIHL_DBG(p) & 256) && n_tap != 0) return;  // tuning ablation: weights fetched for the first tap only
    unsigned voff = b_off[j] + (unsigned)n_bdelta;
    if (!cin_full) voff = (n_kc * KCE + b_ch[j] < p.Cin) ? voff : OOB;
    dma16(voff, n_bbase + j * 1024, wt_rsrc);
  };

  typename AccTile<TILE>::type acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < REGS; ++r) acc[i][j][r] = 0.f;

  // fragment reads: lane -> row fr of a TILE-row block, 16-byte chunk ks * (64 / TILE) + fg of the stage's 128 bytes of K
  // (16x16x32: the four 16-lane groups take four consecutive chunks = K 0..31 of the step).  With the source-side XOR both
  // shapes put the 16 lanes of every ds_read_b128 group on 16 distinct 4-bank slots.
  const int fr = lane & (TILE - 1), fg = lane / TILE, fsw = (fr >> 1) & 7;
  int koff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) koff[ks] = fr * KCB + (((ks * (64 / TILE) + fg) ^ fsw) << 4);

  // wait until at most `keep` of the newest stage groups are still in flight (vm ops retire in order)
  auto wait_keep = [&](int keep) {
    if (NBUF >= 4 && keep >= 2) wait_vm_keep<2 * PER_STAGE>();
    else if (NBUF >= 3 && keep >= 1) wait_vm_keep<PER_STAGE>();
    else wait_vm_keep<0>();
  };
  if constexpr (NBUF == 1) {
    // One LDS stage: load -> multiply -> load ...  No overlap inside the workgroup; the small footprint lets two or
    // three workgroups share a CU and overlap each other (thin-K pointwise layers: one or two stages in total, where
    // the output store of one workgroup runs under the loads of the next).
    const char* As = smem + wm * WTM * KCB;
    const char* Bs = smem + A_BYTES + wn * WTN * KCB;
    for (int s = 0; s < ((SIHL_DBG(p) & 64) ? 0 : nstages); ++s) {
      stage_setup(s, 0);
      if (!(SIHL_DBG(p) & 1)) {
#pragma unroll
        for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
        for (int j = 0; j < NBL; ++j) issue_b(j);
      }
      wait_vm_keep<0>();
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < ((SIHL_DBG(p) & 2) ? 0 : NKS); ++ks) {
        uint4 fa[MT], fb[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *(const uint4*)(As + i * TILE * KCB + koff[ks]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(Bs + j * TILE * KCB + koff[ks]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_step<T, TILE>(acc[i][j], fa[i], fb[j]);
      }
      __syncthreads();  // everyone is done with the stage before it is overwritten (or reused by the epilogue)
    }
  } else {
#pragma unroll
  for (int s0 = 0; s0 < NBUF - 1; ++s0) {
    if (s0 < nstages) {
      stage_setup(s0, s0);
#pragma unroll
      for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
      for (int j = 0; j < NBL; ++j) issue_b(j);
    }
  }
  wait_keep(min(NBUF - 2, nstages - 1));
  __syncthreads();

  int buf = 0, nbuf_next = NBUF - 1;  // LDS stage being multiplied / being filled
  for (int s = 0; s < ((SIHL_DBG(p) & 64) ? 0 : nstages); ++s) {  // dbg 64: tuning ablation, prologue + epilogue only
    const bool more = (s + NBUF - 1 < nstages) && !(SIHL_DBG(p) & 1);
    if (more) stage_setup(s + NBUF - 1, nbuf_next);
    const char* As = smem + buf * STAGE + wm * WTM * KCB;
    const char* Bs = smem + buf * STAGE + A_BYTES + wn * WTN * KCB;
    // DMA issue schedule for the next stage (SIHL_DBG(p) bits 2-3 select it while tuning):
    //   0: NA/NBL slots spread over the k-steps   1: everything before the first k-step
    //   2: staggered - waves of the first half issue before k-step 0, the others after k-step 1
    //   3: front-loaded - slots spread over the first half of the k-steps
    const int sched = (SIHL_DBG(p) & 16) ? ((SIHL_DBG(p) >> 2) & 3) : ((BM >= 256 && NBUF == 2 && WM * WN == 8) ? 3 : 1);
    const bool early = sched == 1 || (sched == 2 && wave < WM * WN / 2);
    constexpr int NKSH = NKS / 2 > 0 ? NKS / 2 : 1;
    if (!(SIHL_DBG(p) & 2)) {
      uint4 fa[2][MT], fb[2][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) fa[0][i] = *(const uint4*)(As + i * TILE * KCB + koff[0]);
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[0][j] = *(const uint4*)(Bs + j * TILE * KCB + koff[0]);
      if (more && early) {
#pragma unroll
        for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
        for (int j = 0; j < NBL; ++j) issue_b(j);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // (tried: issue priority falling as the wave advances through the stage, common.h SIHL_PRIO - it evens out the waves of
        // the weight-gradient loop, here and in conv_halo.hip it changed nothing: profiles/r04_prio_lib_ab.txt)
        if (ks + 1 < NKS) {  // fragments of the next k-step are in flight while this one multiplies
#pragma unroll
          for (int i = 0; i < MT; ++i) fa[(ks + 1) & 1][i] = *(const uint4*)(As + i * TILE * KCB + koff[ks + 1]);
#pragma unroll
          for (int j = 0; j < NT; ++j) fb[(ks + 1) & 1][j] = *(const uint4*)(Bs + j * TILE * KCB + koff[ks + 1]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma_step<T, TILE>(acc[i][j], fa[ks & 1][i], fb[ks & 1][j]);
        if (more && !early) {
          if (sched == 0) {
#pragma unroll
            for (int j = ks; j < NA; j += NKS) issue_a(j);
#pragma unroll
            for (int j = ks; j < NBL; j += NKS) issue_b(j);
          } else if (sched == 2) {
            if (ks == 1) {
#pragma unroll
              for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
              for (int j = 0; j < NBL; ++j) issue_b(j);
            }
          } else if (ks < NKSH) {
            // (one block per k-step, not a slice after every row of MFMA tiles: interleaved that finely the L3 3x3 took
            // 157 us instead of 143 - the two waves of a SIMD alternate best with one in its DMA block, one multiplying)
#pragma unroll
            for (int j = ks; j < NA; j += NKSH) issue_a(j);
#pragma unroll
            for (int j = ks; j < NBL; j += NKSH) issue_b(j);
          }
        }
      }
    } else if (more) {
#pragma unroll
      for (int j = 0; j < NA; ++j) issue_a(j);
#pragma unroll
      for (int j = 0; j < NBL; ++j) issue_b(j);
    }
    // (the multiplies stay above the wait: with two 32-deep k-steps per stage the scheduler otherwise sinks the second
    // k-step's MFMAs - register operands only - below the barrier, and the next stage's DMA flight is covered by nothing.
    // Also tried on the 256 x 256 tile, round 3: the barrier taken 1-2 MFMA rows before the end of the stage with the next
    // stage's first fragments read behind it, DMA block after 0 / 2 / 4 rows, half the waves issuing early - all within
    // 1 % of this form on L3 3x3, lat3 1x1 and r2 1x1 128>512, gpurun_out mf16/variants2.txt; not kept.)
    __builtin_amdgcn_sched_barrier(0);
    // stage s+1 must have landed: only the stages issued after it may still be in flight ...
    wait_keep(max(0, min(NBUF - 2, nstages - 2 - s)));
    __syncthreads();  // ... for every wave, and everyone is done reading stage s
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    nbuf_next = nbuf_next + 1 == NBUF ? 0 : nbuf_next + 1;
  }
  }  // NBUF > 1
  if (SIHL_DBG(p) & 32) return;  // tuning ablation: no epilogue
  if (p.splits > 1) {  // raw fp32 partial tile; conv_splitk_epilogue_kernel sums the splits and finishes
    float* part = p.partial + (long)blockIdx.y * p.M * p.Cout;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + wn * WTN + j * TILE + (lane & (TILE - 1));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < REGS; ++r) {
          const int m = m0 + wm * WTM + i * TILE + acc_row<TILE>(r) + 4 * fg;
          if (m < p.M && co < p.Cout) part[(long)m * p.Cout + co] = acc[i][j][r];
        }
    }
    return;
  }
  if constexpr (ADD) conv_epilogue_body<T, BM, BN, WM, WN, SIHL_ACT_NONE, 0, true, TILE>(p, acc, smem, tile_m, m0, n0);
  else conv_epilogue<T, BM, BN, WM, WN, TILE>(p, acc, smem, tile_m, m0, n0);
}

// out += add over n elements (fallback for tile configurations without an ADD instantiation)
template <typename T>
__global__ void conv_add_inplace_kernel(T* __restrict__ out, const T* __restrict__ add, long nvec) {
  constexpr int V = 16 / (int)sizeof(T);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float a[V], b[V];
    unpack16(*(const uint4*)(out + i * V), a, T());
    unpack16(*(const uint4*)(add + i * V), b, T());
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += b[e];
    *(uint4*)(out + i * V) = pack16(a, T());
  }
}

// out[n][y*s][x*s][:] += add[n][y][x][:] (strided addend, see ConvParams::add_stride); C % V == 0
template <typename T>
__global__ void conv_add_strided_kernel(T* __restrict__ out, const T* __restrict__ add, long nvec, int C, int aH, int aW,
                                        int oH, int oW, int s) {
  constexpr int V = 16 / (int)sizeof(T);
  const int cv = C / V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long pix = i / cv;
    const int x = (int)(pix % aW); pix /= aW;
    const int y = (int)(pix % aH);
    const long n = pix / aH;
    T* o = out + ((n * oH + (long)y * s) * oW + (long)x * s) * C + c * V;
    float a[V], b[V];
    unpack16(*(const uint4*)o, a, T());
    unpack16(*(const uint4*)(add + i * V), b, T());
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += b[e];
    *(uint4*)o = pack16(a, T());
  }
}

template <typename T>
void launch_add_inplace(const ConvParams& p, hipStream_t stream) {
  constexpr int V = 16 / (int)sizeof(T);
  if (p.add_stride > 1) {
    const long nv = (long)p.N * p.add_H * p.add_W * p.Cout / V;
    long gs = (nv + 255) / 256;
    if (gs > 4096) gs = 4096;
    if (gs < 1) gs = 1;
    hipLaunchKernelGGL(conv_add_strided_kernel<T>, dim3((unsigned)gs), dim3(256), 0, stream, (T*)p.out, (const T*)p.add, nv,
                       p.Cout, p.add_H, p.add_W, p.Ho, p.Wo, p.add_stride);
    return;
  }
  const long nvec = (long)p.M * p.Cout / V;  // dense output, Cout % V == 0
  long g = (nvec + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(conv_add_inplace_kernel<T>, dim3((unsigned)g), dim3(256), 0, stream, (T*)p.out, (const T*)p.add, nvec);
}

// Second half of a split-K conv: out = epilogue(sum_s partial[s]) with the same bias / statistics / affine / activation
// chain as conv_epilogue_body.  Workgroup = 32 * ROWS pixels x 32 output channels, thread = 4 consecutive channels of
// ROWS consecutive pixels.  ROWS = 4: one 128-pixel statistics row per workgroup (training).  ROWS = 1 (no statistics):
// four times the workgroups - an L7 level is 512 pixels, and 32 workgroups each walking nine slabs with one load in
// flight took longer (14 us) than the matrix kernel in front of them.  The slabs of up to four splits are requested
// together, then added in split order (the sum is the same fp32 chain either way).  Dense output only
// (out_image_stride == Ho*Wo*Cout).
template <typename T, int ACT, int STATS, int ROWS>
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvParams p) {
  static_assert(STATS == 0 || ROWS == 4, "a statistics row is 128 pixels");
  __shared__ float red[STATS ? 2 : 1][STATS ? 32 : 1][33];
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int co = blockIdx.y * 32 + c4 * 4;
  const bool cok = co < p.Cout;  // Cout % 4 == 0 (vector width), so the 4 channels are valid together
  const bool has_pre = p.pre_scale != nullptr, has_post = p.post_scale != nullptr;
  float bias[4], s1[4], t1[4], s2[4], t2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bias[e] = (p.bias && cok) ? p.bias[co + e] : 0.f;
    s1[e] = (has_pre && cok) ? p.pre_scale[co + e] : 1.f;
    t1[e] = (has_pre && p.pre_shift && cok) ? p.pre_shift[co + e] : 0.f;
    s2[e] = (has_post && cok) ? p.post_scale[co + e] : 1.f;
    t2[e] = (has_post && p.post_shift && cok) ? p.post_shift[co + e] : 0.f;
  }
  const long slab = (long)p.M * p.Cout;
  const int mbase = blockIdx.x * (32 * ROWS) + rl * ROWS;
  float v[ROWS][4];
#pragma unroll
  for (int k = 0; k < ROWS; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[k][e] = 0.f;
  if (cok) {
    for (int sp0 = 0; sp0 < p.splits; sp0 += 4) {
      float4 t[4][ROWS];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
          t[u][k] = (sp0 + u < p.splits && mbase + k < p.M)
                        ? *(const float4*)(p.partial + (sp0 + u) * slab + (long)(mbase + k) * p.Cout + co)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
          if (sp0 + u < p.splits) { v[k][0] += t[u][k].x; v[k][1] += t[u][k].y; v[k][2] += t[u][k].z; v[k][3] += t[u][k].w; }
        }
    }
  }
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  T* __restrict__ out = (T*)p.out;
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const bool ok = cok && mbase + k < p.M;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = v[k][e] + bias[e];
      if (STATS == 1 && ok) { ssum[e] += x; ssq[e] += x * x; }
      x = x * s1[e] + t1[e];
      if (ACT == SIHL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (ACT == SIHL_ACT_SILU) x = x / (1.f + expf(-x));
      else if (ACT == SIHL_ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
      if (STATS == 2 && ok) { ssum[e] += x; ssq[e] += x * x; }
      o[e] = x * s2[e] + t2[e];
      if (p.add && ok) { const long aoff = add_offset(p, mbase + k); if (aoff >= 0) o[e] += elem<T>::ld((const T*)p.add + aoff + co + e); }
    }
    if (ok) {
      T* dst = out + (long)(mbase + k) * p.Cout + co;
#pragma unroll
      for (int e = 0; e < 4; ++e) elem<T>::st(dst + e, o[e]);
    }
  }
  if constexpr (STATS != 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][rl][c4 * 4 + e] = ssum[e]; red[1][rl][c4 * 4 + e] = ssq[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, col = threadIdx.x & 31;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) a += red[which][k][col];
      if (blockIdx.y * 32 + col < p.Cout) p.stats[((long)blockIdx.x * 2 + which) * p.Cout + blockIdx.y * 32 + col] = a;
    }
  }
}

template <typename T>
void launch_splitk_epilogue(const ConvParams& p, hipStream_t stream) {
  const dim3 grid((p.M + 127) / 128, (p.Cout + 31) / 32), grid1((p.M + 31) / 32, (p.Cout + 31) / 32);
#define SIHL_SKE(A, S) hipLaunchKernelGGL((conv_splitk_epilogue_kernel<T, A, S, 4>), grid, dim3(256), 0, stream, p)
#define SIHL_SKE1(A) hipLaunchKernelGGL((conv_splitk_epilogue_kernel<T, A, 0, 1>), grid1, dim3(256), 0, stream, p)
  if (p.stats_mode == 0) {
    if (p.act == SIHL_ACT_NONE) SIHL_SKE1(SIHL_ACT_NONE);
    else if (p.act == SIHL_ACT_RELU) SIHL_SKE1(SIHL_ACT_RELU);
    else if (p.act == SIHL_ACT_SILU) SIHL_SKE1(SIHL_ACT_SILU);
    else SIHL_SKE1(SIHL_ACT_SIGMOID);
  } else if (p.stats_mode == 1) {
    SIHL_SKE(SIHL_ACT_NONE, 1);
  } else {
    if (p.act == SIHL_ACT_RELU) SIHL_SKE(SIHL_ACT_RELU, 2);
    else if (p.act == SIHL_ACT_NONE) SIHL_SKE(SIHL_ACT_NONE, 2);
    else if (p.act == SIHL_ACT_SILU) SIHL_SKE(SIHL_ACT_SILU, 2);
    else SIHL_SKE(SIHL_ACT_SIGMOID, 2);
  }
#undef SIHL_SKE
#undef SIHL_SKE1
}

template <typename T, int BM, int BN, int WM, int WN, int NBUF = 2>
int launch_dma(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  p.gridM = (p.M + BM - 1) / BM;
  p.gridN = (p.Cout + BN - 1) / BN;
  constexpr int EPI = BM * (BN * (int)sizeof(T) + 16) + 2 * WM * BN * 4;
  constexpr int STAGES2 = NBUF * (BM + BN) * KCB;
  constexpr int LDS = STAGES2 > EPI ? STAGES2 : EPI;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  // the addend epilogue exists for the tiles the identity-block dgrads use; other configurations add in a second pass
  constexpr bool CAN_ADD = (BM == 128 && BN == 128 && NBUF == 1) || (BM == 256 && BN == 256);
  const bool fused_add = p.add && CAN_ADD && p.splits == 1 && p.in_dilate == 1 && p.act == SIHL_ACT_NONE &&
                         p.stats_mode == 0 && !p.bias && !p.pre_scale && !p.post_scale;
  const void* late_add = (p.add && !fused_add && p.splits == 1) ? p.add : nullptr;  // split-K adds in its finisher
  if (late_add) p.add = nullptr;
  {
    const int min_stages = (g_krot / 1000) % 100 ? (g_krot / 1000) % 100 : 8;  // (tuning: sihl_conv2d_krot(100000 * log2(group) + 1000 * min_stages + stride))
    p.k_rot_group = g_krot / 100000;  // 2^k neighbouring workgroups share a start (and their L2 fills)
    p.k_rotate = (!(g_rules_off & 4) && p.splits == 1 && p.gridM * p.gridN >= 64 &&
                  ((p.Cin + KCB / (int)sizeof(T) - 1) / (KCB / (int)sizeof(T))) * p.KH * p.KW >= min_stages) ? g_krot % 1000 : 0;
  }
  auto kern = p.in_dilate > 1 ? conv_igemm_dma_kernel<T, BM, BN, WM, WN, true, NBUF>
                              : conv_igemm_dma_kernel<T, BM, BN, WM, WN, false, NBUF>;
  if constexpr (CAN_ADD) {
    if (fused_add) kern = conv_igemm_dma_kernel<T, BM, BN, WM, WN, false, NBUF, true>;
  }
  if (!attr_set) {
    hipError_t e0 = hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<T, BM, BN, WM, WN, true, NBUF>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e0 != hipSuccess) return (int)e0;
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<T, BM, BN, WM, WN, false, NBUF>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    if constexpr (CAN_ADD) {
      e = hipFuncSetAttribute((const void*)conv_igemm_dma_kernel<T, BM, BN, WM, WN, false, NBUF, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  // algorithmic flops: the zero-dilated read of a strided conv's dgrad multiplies (in_dilate^2 - 1) / in_dilate^2 zeros
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin / ((double)p.in_dilate * p.in_dilate);
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * p.KH * p.KW * p.Cin) * sizeof(T);
  sihl_prof_begin(SIHL_PROF_CONV, sizeof(T) == 2 ? SIHL_BF16 : SIHL_F32, flops, bytes, stream);
  hipLaunchKernelGGL(kern, dim3(p.gridM * p.gridN, p.splits), dim3(WM * WN * 64), LDS, stream, p);
  if (p.splits > 1) launch_splitk_epilogue<T>(p, stream);
  if (late_add) { p.add = late_add; launch_add_inplace<T>(p, stream); }
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

template <typename T, int BN, int WM, int WN>
int launch_reg(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  constexpr int BM = BM128;
  p.gridM = (p.M + BM - 1) / BM;
  p.gridN = (p.Cout + BN - 1) / BN;
  constexpr int EPI = BM * (BN * (int)sizeof(T) + 16) + 2 * WM * BN * 4;
  constexpr int STAGES2 = 2 * (BM + BN) * LDS_STRIDE;
  constexpr int LDS = STAGES2 > EPI ? STAGES2 : EPI;
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<T, BN, WM, WN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  // algorithmic flops: the zero-dilated read of a strided conv's dgrad multiplies (in_dilate^2 - 1) / in_dilate^2 zeros
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin / ((double)p.in_dilate * p.in_dilate);
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * p.KH * p.KW * p.Cin) * sizeof(T);
  sihl_prof_begin(SIHL_PROF_CONV, sizeof(T) == 2 ? SIHL_BF16 : SIHL_F32, flops, bytes, stream);
  const void* late_add = p.add;
  p.add = nullptr;
  hipLaunchKernelGGL(kern, dim3(p.gridM * p.gridN), dim3(256), LDS, stream, p);
  if (late_add) { p.add = late_add; launch_add_inplace<T>(p, stream); }
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// stage-count selection of the narrow tiles (g_nbuf: tuning hook, 0 = default)
// Few LDS stages keep the footprint small, so several workgroups share a CU and hide each other's waits - worth
// more than a deeper pipeline whenever the grid has more workgroups than CUs (measured: L4 3x3 60 us with 2 stages,
// 93 us with 4).  Grids of <= 256 workgroups are alone on their CU anyway and take 4 stages (lat5 1x1 K=2048:
// 42 -> 32 us).
template <typename T, int BN> int stages_for(const ConvParams& p) {
  if (g_nbuf) return g_nbuf;
  const long wgs = ((p.M + 127) / 128) * ((p.Cout + BN - 1) / BN);
  if (wgs <= 256) return 4;
  // Bigger grids: ONE stage (32-48 KB of LDS with the epilogue staging: 3-4 workgroups per CU hide each other's
  // load -> multiply -> store phases) beats two stages for windowed convs and thin-K pointwise layers - r1 3x3 64:
  // 120 -> 88 us, r3 1x1 256>1024: 49 -> 39, L4 3x3: 62 -> 57 - but not for pointwise layers with a long K loop
  // into few channels (r2 1x1 512>128: 46 -> 50, r3 1x1 1024>256: 34 -> 36), which keep two.
  if (!(g_rules_off & 1) && (p.KH * p.KW > 1 || p.Cin <= 256 || p.Cout >= 1024)) return 1;
  return 2;
}
// Split-K plan for the 128x64 tile: levels with <= 64 workgroups (2048 pixels x 256 channels and below) walk their
// whole K loop (36 stages for a 3x3 over 256 channels, ~1 us each) on a fraction of the chip; slicing the stages
// over grid.y fills it (L7 3x3: 16 workgroups x 36 stages -> 144 x 4) at the price of a small fp32 slab + one
// finishing kernel.  Returns 1 when not worth it / no workspace.
template <typename T> int splitk_plan(const ConvParams& p, long ws_bytes) {
  if (!g_splitk || !p.partial || p.in_dilate > 1) return 1;
  if (p.out_image_stride != (long)p.Ho * p.Wo * p.Cout) return 1;  // the finishing kernel writes dense rows
  constexpr int KCE = KCB / (int)sizeof(T);
  const long wgs = ((p.M + 127) / 128) * ((p.Cout + 63) / 64);
  const int stages = ((p.Cin + KCE - 1) / KCE) * p.KH * p.KW;
  if (wgs > 64 || stages < 16) return 1;  // (also tried: <= 256 workgroups with up to 768 slices - no gain on L5/L6)
  long s = stages / 4;
  if (s > 256 / wgs) s = 256 / wgs;
  if (s < 2) return 1;
  const int per = (int)((stages + s - 1) / s);
  s = (stages + per - 1) / per;  // every split non-empty
  if (s < 2 || ws_bytes < s * (long)p.M * p.Cout * (long)sizeof(float)) return 1;
  return (int)s;
}

template <typename T> int launch_n64(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  p.splits = splitk_plan<T>(p0, p0.partial_bytes);
  const int nb = p.splits > 1 ? 4 : stages_for<T, 64>(p);
  if (nb == 1) return launch_dma<T, 128, 64, 4, 1, 1>(p, stream);
  if (nb == 2) return launch_dma<T, 128, 64, 4, 1, 2>(p, stream);
  if (nb == 3) return launch_dma<T, 128, 64, 4, 1, 3>(p, stream);
  return launch_dma<T, 128, 64, 4, 1, 4>(p, stream);
}
template <typename T> int launch_n128(const ConvParams& p, hipStream_t stream) {
  const int nb = stages_for<T, 128>(p);
  if (nb == 1) return launch_dma<T, 128, 128, 2, 2, 1>(p, stream);
  if (nb == 2) return launch_dma<T, 128, 128, 2, 2, 2>(p, stream);
  if (nb == 3) return launch_dma<T, 128, 128, 2, 2, 3>(p, stream);
  return launch_dma<T, 128, 128, 2, 2, 4>(p, stream);
}

template <typename T>
int dispatch(const ConvParams& p, hipStream_t stream) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (p.Cin % VEC != 0) return SIHL_EARG;  // 16-byte channel vectors required; caller pads
  const long in_bytes = (long)p.N * p.H * p.W * p.Cin * (long)sizeof(T);
  const long wt_bytes = (long)p.Cout * p.KH * p.KW * p.Cin * (long)sizeof(T);
  const bool dma = !g_force_reg && in_bytes < (1L << 31) && wt_bytes < (1L << 31) && p.KH * p.KW <= 32;
  if constexpr (sizeof(T) == 2) {
    // 3x3 convs of the small pyramid levels: halo-resident input patch, split-K finished inside the launch (conv_small.hip)
    if (dma && g_tile_override == 0 && !g_nbuf && sihl_pyr_eligible(p)) return sihl_pyr_launch(p, stream);
    if (dma && g_tile_override == 0 && !g_nbuf && sihl_small_eligible(p)) return sihl_small_launch(p, stream);
    // 3x3, 256 -> 256 channels on 64-wide maps: the 256 x 256 tile with the input patch resident in LDS (conv_halo.hip)
    if (dma && g_tile_override == 0 && !g_nbuf && sihl_halo_eligible(p)) return sihl_halo_launch(p, stream);
  }
  if (!dma) {
    if (p.Cout > 128) return launch_reg<T, 256, 2, 2>(p, stream);
    if (p.Cout > 64) return launch_reg<T, 128, 2, 2>(p, stream);
    return launch_reg<T, 64, 4, 1>(p, stream);
  }
  if (p.Cout > 128) {
    // 256-pixel x 256-channel tiles (8 waves) halve the weight-panel traffic per flop; worth it once the
    // grid still fills the chip.  Small pyramid levels are latency-bound (a K loop of KH*KW*Cin/64 stages
    // on a handful of workgroups): narrower channel tiles spread them over 2-4x more CUs.
    const long tiles128 = (p.M + 127) / 128;
    if constexpr (sizeof(T) == 2) {
      // pointwise, K <= 256 into <= 256 channels (ResNet layer1 expansions, the MLP linears): two channel tiles of
      // 128x128, single stage, 4 workgroups per CU - r1 64>256: 114 -> 104 us, mlp 65 -> 60; wider outputs or a
      // window lose (r2 128>512: 63 -> 66, L3 3x3: 156 -> 182)
      if (g_tile_override == 0 && !(g_rules_off & 2) && p.M >= 256 * 256 && p.KH * p.KW == 1 && p.Cin <= 256 && p.Cout <= 256)
        return launch_n128<T>(p, stream);
      // 16 waves of 64 x 64 (4 per SIMD, 118 registers) rather than 8 of 64 x 128: half the LDS-DMA instructions per wave
      // and stage, and four waves per SIMD to cover each other's DMA-issue and barrier phases - L3 3x3 146 -> 143 us,
      // lat3 1x1 56.4 -> 53.7, r2 1x1 128>512 56.2 -> 51.4 (tools/tile_probe.py, profiles/r03_tile_probe_mfma16.txt)
      if (g_tile_override == 2568) return launch_dma<T, 256, 256, 4, 2>(p, stream);  // (A/B: the 8-wave form)
      if (g_tile_override == 256 || (g_tile_override == 0 && p.M >= 256 * 256)) return launch_dma<T, 256, 256, 4, 4>(p, stream);
    }
    if constexpr (sizeof(T) == 2) {  // 128 pixels x 256 channels on 8 waves (2 x 4), 2 / 3 LDS stages
      if (g_tile_override == 2562) return launch_dma<T, 128, 256, 2, 4, 2>(p, stream);
      if (g_tile_override == 2563) return launch_dma<T, 128, 256, 2, 4, 3>(p, stream);
      // windowed convs over 16 k - 64 k pixels into 256 channels (P4 3x3 at batch 32: 256 workgroups, one per CU, the
      // whole weight panel width per workgroup): 46 us against 52 for 512 single-stage 128x128 tiles (tools/tile_probe.py).
      // (A halo-resident variant of this level - 256 pixels x 128 channels per workgroup, the input patch fetched once
      // instead of once per tap, 0.76 MB of LDS-DMA ingest per CU instead of 1.73 - was built in round 3 and measured
      // SLOWER: 2.300 ms per north-star forward with 8 waves of 64 x 64, 2.391 with 4 waves of 64 x 128, against 2.270 for
      // this tile, profiles/r03_halo_p4_ab.txt.  The level is not ingest-bound: its matrix loop alone takes 37 of the 47 us,
      // because 64 x 64 wave tiles need one ds_read_b128 per MFMA and two such waves per SIMD sit exactly at the LDS array's
      // 256 B/clk, while the tiles that halve the LDS traffic leave half the chip or half the SIMD slots empty at this size.)
      if (g_tile_override == 0 && !(g_rules_off & 8) && p.KH * p.KW > 1 && p.Cout == 256 && tiles128 >= 128 && tiles128 <= 384)
        return launch_dma<T, 128, 256, 2, 4, 2>(p, stream);
    }
    if (g_tile_override == 64 || (g_tile_override == 0 && tiles128 * ((p.Cout + 255) / 256) <= 64))
      return launch_n64<T>(p, stream);
    if (g_tile_override == 1280 || (g_tile_override == 0 && tiles128 * ((p.Cout + 255) / 256) <= 256))
      return launch_n128<T>(p, stream);
    // bf16 below 65536 pixels: 128x128 tiles beat 128x256 (r3 1x1 256>1024: 53 vs 68 us; r4 1x1 512>2048: 39 vs 49)
    if constexpr (sizeof(T) == 2) {
      if (g_tile_override == 0) return launch_n128<T>(p, stream);
    }
    if (g_nbuf == 3) return launch_dma<T, 128, 256, 2, 2, 3>(p, stream);
    return launch_dma<T, 128, 256, 2, 2, 2>(p, stream);
  }
  if (p.Cout > 64) return launch_n128<T>(p, stream);
  return launch_n64<T>(p, stream);
}

}  // namespace
