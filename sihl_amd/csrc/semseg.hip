// SemanticSegmentation head kernels (gfx950), reference src/sihl/heads/semantic_segmentation.py:
//   UAFM (:163-182): channel mean/max statistics of both inputs (one wave per pixel, wave reductions over C),
//     the 4->1 3x3 conv + sigmoid attention map, the alpha blend x1*a + x2*(1-a), and their adjoints;
//   forward (:83-85): nearest resize to the input size + softmax + max, fused so that the
//     (B, classes, H, W) tensor is never materialised - one thread per LOGIT pixel fans out to its block;
//   training_step (:87-92): nearest resize to the target size + cross-entropy with ignore_index, fused: one
//     wave per logit pixel builds the class histogram of its target block; the gradient w.r.t. the logits
//     is produced in the same pass (n_valid * softmax - histogram) and only scaled in backward.
#include "common.h"

namespace {

constexpr int TPB = 256;

template <typename T> __device__ __forceinline__ void ldv(const T* p, float (&f)[16 / sizeof(T)]) {
  unpack16(*(const uint4*)p, f, T());
}
template <typename T> __device__ __forceinline__ void stv(T* p, const float (&f)[16 / sizeof(T)]) {
  *(uint4*)p = pack16(f, T());
}

// ------------------------------------------------------------------ UAFM statistics
// stats[pix] = (mean_c x1, max_c x1, mean_c x2, max_c x2) fp32; arg[pix] = (argmax_c x1, argmax_c x2)
template <typename T>
__global__ void uafm_stats_kernel(const T* __restrict__ x1, const T* __restrict__ x2, float* __restrict__ stats,
                                  int* __restrict__ arg, long npix, int C) {
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TPB) >> 6;
  for (long p = wave; p < npix; p += nwaves) {
    float out[4];
    int am[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const T* x = (t == 0 ? x1 : x2) + p * C;
      float s = 0.f, m = -INFINITY;
      int mi = 0x7fffffff;
      for (int cv = lane; cv < cvec; cv += 64) {
        float f[V];
        ldv(x + cv * V, f);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          s += f[e];
          if (f[e] > m) { m = f[e]; mi = cv * V + e; }
        }
      }
      s = wave_sum(s);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {  // (max, lowest index) reduction: torch.max returns the first maximum
        const float om = __shfl_xor(m, o);
        const int oi = __shfl_xor(mi, o);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
      }
      out[2 * t] = s / C;
      out[2 * t + 1] = m;
      am[t] = mi;
    }
    if (lane == 0) {
      *(float4*)(stats + p * 4) = make_float4(out[0], out[1], out[2], out[3]);
      arg[p * 2] = am[0];
      arg[p * 2 + 1] = am[1];
    }
  }
}

// alpha[pix] = sigmoid(bias + sum_{ky,kx,ci} w[ci][ky][kx] * stats[pix + (ky-1, kx-1)][ci]); w is OIHW (1,4,3,3) fp32
__global__ void uafm_alpha_kernel(const float* __restrict__ stats, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ alpha, int N, int H, int W) {
  const long n = (long)N * H * W;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long base = i - (long)y * W - x;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        if (xx < 0 || xx >= W) continue;
        const float4 s = *(const float4*)(stats + (base + (long)yy * W + xx) * 4);
        acc += w[0 * 9 + ky * 3 + kx] * s.x + w[1 * 9 + ky * 3 + kx] * s.y + w[2 * 9 + ky * 3 + kx] * s.z +
               w[3 * 9 + ky * 3 + kx] * s.w;
      }
    }
    alpha[i] = 1.f / (1.f + expf(-acc));
  }
}

// out = x1*alpha + x2*(1-alpha)
template <typename T>
__global__ void uafm_blend_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ alpha,
                                  T* __restrict__ out, long nvec, int cvec) {
  constexpr int V = 16 / sizeof(T);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const float a = alpha[i / cvec];
    float f1[V], f2[V];
    ldv(x1 + i * V, f1);
    ldv(x2 + i * V, f2);
#pragma unroll
    for (int e = 0; e < V; ++e) f1[e] = f1[e] * a + f2[e] * (1.f - a);
    stv(out + i * V, f1);
  }
}

// dpre[pix] = (sum_c dout*(x1-x2)) * alpha*(1-alpha)   (grad wrt the conv's pre-sigmoid output)
template <typename T>
__global__ void uafm_dpre_kernel(const T* __restrict__ dout, const T* __restrict__ x1, const T* __restrict__ x2,
                                 const float* __restrict__ alpha, float* __restrict__ dpre, long npix, int C) {
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TPB) >> 6;
  for (long p = wave; p < npix; p += nwaves) {
    float s = 0.f;
    for (int cv = lane; cv < cvec; cv += 64) {
      float d[V], a[V], b[V];
      ldv(dout + p * C + cv * V, d);
      ldv(x1 + p * C + cv * V, a);
      ldv(x2 + p * C + cv * V, b);
#pragma unroll
      for (int e = 0; e < V; ++e) s += d[e] * (a[e] - b[e]);
    }
    s = wave_sum(s);
    if (lane == 0) { const float a = alpha[p]; dpre[p] = s * a * (1.f - a); }
  }
}

// dstats[pix][ci] = sum_taps dpre[pix - tap] * w[ci][tap]; dw[ci][ky][kx] += sum_pix dpre[pix]*stats[pix+tap][ci];
// dbias += sum dpre.  gacc: one partial row of 40 floats (36 weights + bias) per workgroup, summed in workgroup order by
// uafm_wgrad_finalize_kernel (no atomics: bit-reproducible).
__global__ void uafm_alpha_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ stats,
                                      const float* __restrict__ w, float* __restrict__ dstats, float* gacc, int N,
                                      int H, int W) {
  __shared__ float red[37][TPB / 64];
  const long n = (long)N * H * W;
  float g[37];
#pragma unroll
  for (int k = 0; k < 37; ++k) g[k] = 0.f;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const long base = i - (long)y * W - x;
    const float dp = dpre[i];
    g[36] += dp;
    float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        // forward read stats[y+ky-1][x+kx-1]: weight-gradient term
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const float4 s = *(const float4*)(stats + (base + (long)yy * W + xx) * 4);
          g[0 * 9 + ky * 3 + kx] += dp * s.x;
          g[1 * 9 + ky * 3 + kx] += dp * s.y;
          g[2 * 9 + ky * 3 + kx] += dp * s.z;
          g[3 * 9 + ky * 3 + kx] += dp * s.w;
        }
        // adjoint: this pixel's stats were read by output pixel (y-ky+1, x-kx+1)
        const int oy = y - ky + 1, ox = x - kx + 1;
        if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
          const float d = dpre[base + (long)oy * W + ox];
          ds.x += d * w[0 * 9 + ky * 3 + kx];
          ds.y += d * w[1 * 9 + ky * 3 + kx];
          ds.z += d * w[2 * 9 + ky * 3 + kx];
          ds.w += d * w[3 * 9 + ky * 3 + kx];
        }
      }
    }
    *(float4*)(dstats + i * 4) = ds;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 37; ++k) {
    const float s = wave_sum(g[k]);
    if (lane == 0) red[k][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < 37) {
    float s = 0.f;
    for (int v = 0; v < TPB / 64; ++v) s += red[threadIdx.x][v];
    gacc[(long)blockIdx.x * 40 + threadIdx.x] = s;
  }
}

__global__ void uafm_wgrad_finalize_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ dconv_w,
                                           float* __restrict__ dconv_b) {
  const int k = threadIdx.x;
  if (k >= 37) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[(long)b * 40 + k];
  if (k < 36) { if (dconv_w) dconv_w[k] = s; }
  else if (dconv_b) dconv_b[0] = s;
}

// dx1 = alpha*dout + dstats.mean1/C + [c == argmax1]*dstats.max1 ; dx2 likewise with (1-alpha)
template <typename T>
__global__ void uafm_dx_kernel(const T* __restrict__ dout, const float* __restrict__ alpha,
                               const float* __restrict__ dstats, const int* __restrict__ arg, T* __restrict__ dx1,
                               T* __restrict__ dx2, long nvec, int cvec) {
  constexpr int V = 16 / sizeof(T);
  const float invC = 1.f / (cvec * V);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const long p = i / cvec;
    const int c0 = (int)(i % cvec) * V;
    const float a = alpha[p];
    const float4 ds = *(const float4*)(dstats + p * 4);
    const int a1 = arg[p * 2], a2 = arg[p * 2 + 1];
    float d[V], o[V];
    ldv(dout + i * V, d);
    if (dx1) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = a * d[e] + ds.x * invC + (c0 + e == a1 ? ds.y : 0.f);
      stv(dx1 + i * V, o);
    }
    if (dx2) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = (1.f - a) * d[e] + ds.z * invC + (c0 + e == a2 ? ds.w : 0.f);
      stv(dx2 + i * V, o);
    }
  }
}

// ------------------------------------------------------------------ nearest resize helpers
__device__ __forceinline__ int nearest_src(int dst, int in_size, float scale) {
  return min((int)floorf(dst * scale), in_size - 1);
}

// scores[N][H][W] fp32 = max_c softmax(logits[src]); classes int64 = argmax.  One thread per logit pixel.
template <typename T>
__global__ void softmax_max_resize_kernel(const T* __restrict__ logits, float* __restrict__ scores,
                                          long* __restrict__ classes, int N, int h, int w, int C, int H, int W) {
  const long n = (long)N * h * w;
  const float sy = (float)h / H, sx = (float)w / W;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
    const T* l = logits + i * C;
    float m = elem<T>::ld(l);
    int am = 0;
    for (int c = 1; c < C; ++c) { const float v = elem<T>::ld(l + c); if (v > m) { m = v; am = c; } }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(elem<T>::ld(l + c) - m);
    const float score = 1.f / s;
    // destination block: conservative candidate range, exact membership re-tested
    const int y0 = max(0, (int)floorf(y / sy) - 1), y1 = min(H - 1, (int)ceilf((y + 1) / sy) + 1);
    const int x0 = max(0, (int)floorf(x / sx) - 1), x1 = min(W - 1, (int)ceilf((x + 1) / sx) + 1);
    for (int yy = y0; yy <= y1; ++yy) {
      if (nearest_src(yy, h, sy) != y) continue;
      for (int xx = x0; xx <= x1; ++xx) {
        if (nearest_src(xx, w, sx) != x) continue;
        const long o = ((long)b * H + yy) * W + xx;
        scores[o] = score;
        classes[o] = am;
      }
    }
  }
}

// One wave per logit pixel.  Per workgroup: part[b] = (sum over its valid target pixels of (lse - logit[target]), n_valid),
// summed in workgroup order by ce_finalize_kernel;
// dl[pix][c] = n_valid*softmax_c - hist_c  (unscaled gradient of the SUMMED loss).
template <typename T>
__global__ void ce_resize_kernel(const T* __restrict__ logits, const long* __restrict__ targets, long ignore_index,
                                 const float* __restrict__ inv_count, T* __restrict__ dl, float* acc, int N, int h,
                                 int w, int C, int H, int W) {
  extern __shared__ float smem[];  // per wave: C logits + C histogram
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* lg = smem + wv * 2 * C;
  float* hist = lg + C;
  const long npix = (long)N * h * w;
  const long wave = ((long)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TPB) >> 6;
  const float sy = (float)h / H, sx = (float)w / W;
  float loss = 0.f, cnt = 0.f;
  const float dscale = inv_count ? inv_count[0] : 1.f;
  for (long p = wave; p < npix; p += nwaves) {
    const int x = (int)(p % w), y = (int)((p / w) % h), b = (int)(p / ((long)w * h));
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) {
      const float v = elem<T>::ld(logits + p * C + c);
      lg[c] = v;
      hist[c] = 0.f;
      m = fmaxf(m, v);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(lg[c] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    const int y0 = max(0, (int)floorf(y / sy) - 1), y1 = min(H - 1, (int)ceilf((y + 1) / sy) + 1);
    const int x0 = max(0, (int)floorf(x / sx) - 1), x1 = min(W - 1, (int)ceilf((x + 1) / sx) + 1);
    const int bw = x1 - x0 + 1, total = (y1 - y0 + 1) * bw;
    float nv = 0.f, ls = 0.f;
    for (int k = lane; k < total; k += 64) {
      const int yy = y0 + k / bw, xx = x0 + k % bw;
      if (nearest_src(yy, h, sy) != y || nearest_src(xx, w, sx) != x) continue;
      const long t = targets[((long)b * H + yy) * W + xx];
      if (t == ignore_index || t < 0 || t >= C) continue;
      nv += 1.f;
      ls += lse - lg[t];
      atomicAdd(hist + t, 1.f);
    }
    nv = wave_sum(nv);
    loss += ls;
    if (lane == 0) cnt += nv;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // LDS atomics of this wave are complete before the read below
    for (int c = lane; c < C; c += 64)
      elem<T>::st(dl + p * C + c, (nv * expf(lg[c] - lse) - hist[c]) * dscale);
  }
  loss = wave_sum(loss);
  __shared__ float wsum[TPB / 64][2];
  if (lane == 0) { wsum[wv][0] = loss; wsum[wv][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int v = 0; v < TPB / 64; ++v) t += wsum[v][threadIdx.x];
    acc[2 + 2 * (long)blockIdx.x + threadIdx.x] = t;
  }
}

__global__ void ce_finalize_kernel(float* __restrict__ acc, int nblocks) {
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int b = 0; b < nblocks; ++b) t += acc[2 + 2 * (long)b + threadIdx.x];
    acc[threadIdx.x] = t;
  }
}

inline int grid_for(long n) {
  long g = (n + TPB - 1) / TPB;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline int grid_for_waves(long npix) {
  long g = (npix + 3) / 4;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define DISPATCH_DTYPE(dtype, ...)                                   \
  if (dtype == SIHL_F32) { typedef float T; __VA_ARGS__; }           \
  else if (dtype == SIHL_BF16) { typedef bf16_t T; __VA_ARGS__; }    \
  else return SIHL_EARG;

extern "C" {

// UAFM forward: stats fp32 [N][H][W][4], arg int32 [N][H][W][2], alpha fp32 [N][H][W] are outputs kept for backward.
// conv_w: the (1,4,3,3) conv weight, fp32 contiguous; conv_b: 1 float (may be NULL).
int sihl_uafm_fwd(const void* x1, const void* x2, const float* conv_w, const float* conv_b, void* out, float* stats,
                  int* arg, float* alpha, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!x1 || !x2 || !conv_w || !out || !stats || !arg || !alpha || N <= 0 || H <= 0 || W <= 0) return SIHL_EARG;
  const long npix = (long)N * H * W;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(uafm_stats_kernel<T>, dim3(grid_for_waves(npix)), dim3(TPB), 0, stream, (const T*)x1,
                       (const T*)x2, stats, arg, npix, C);
    hipLaunchKernelGGL(uafm_alpha_kernel, dim3(grid_for(npix)), dim3(TPB), 0, stream, (const float*)stats, conv_w,
                       conv_b, alpha, N, H, W);
    hipLaunchKernelGGL(uafm_blend_kernel<T>, dim3(grid_for(npix * (C / V))), dim3(TPB), 0, stream, (const T*)x1,
                       (const T*)x2, (const float*)alpha, (T*)out, npix * (C / V), C / V);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// ws: floats, at least 5*N*H*W + 512*40 (dpre, dstats, per-workgroup partial rows).  dconv_w: 36 floats, dconv_b: 1 float
// (may be NULL).
long sihl_uafm_bwd_ws_bytes(int N, int H, int W) { return ((long)N * H * W * 5 + 512 * 40) * (long)sizeof(float); }

int sihl_uafm_bwd(const void* dout, const void* x1, const void* x2, const float* conv_w, const float* stats,
                  const int* arg, const float* alpha, void* dx1, void* dx2, float* dconv_w, float* dconv_b, int N,
                  int H, int W, int C, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
  if (!dout || !x1 || !x2 || !conv_w || !stats || !arg || !alpha || !ws || N <= 0) return SIHL_EARG;
  if (ws_bytes < sihl_uafm_bwd_ws_bytes(N, H, W)) return SIHL_EWS;
  const long npix = (long)N * H * W;
  float* dpre = ws;
  float* dstats = ws + npix;
  float* gacc = ws + npix * 5;
  const int nblk = grid_for(npix) > 512 ? 512 : grid_for(npix);
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(uafm_dpre_kernel<T>, dim3(grid_for_waves(npix)), dim3(TPB), 0, stream, (const T*)dout,
                       (const T*)x1, (const T*)x2, alpha, dpre, npix, C);
    hipLaunchKernelGGL(uafm_alpha_bwd_kernel, dim3(nblk), dim3(TPB), 0, stream,
                       (const float*)dpre, stats, conv_w, dstats, gacc, N, H, W);
    hipLaunchKernelGGL(uafm_dx_kernel<T>, dim3(grid_for(npix * (C / V))), dim3(TPB), 0, stream, (const T*)dout, alpha,
                       (const float*)dstats, arg, (T*)dx1, (T*)dx2, npix * (C / V), C / V);
  });
  if (dconv_w || dconv_b)
    hipLaunchKernelGGL(uafm_wgrad_finalize_kernel, dim3(1), dim3(64), 0, stream, (const float*)gacc, nblk, dconv_w, dconv_b);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// scores fp32 [N][H][W], classes int64 [N][H][W] from logits [N][h][w][C] (any C): nearest resize + softmax + max
int sihl_softmax_max_resize(const void* logits, float* scores, long* classes, int N, int h, int w, int C, int H,
                            int W, int dtype, hipStream_t stream) {
  if (!logits || !scores || !classes || N <= 0 || h <= 0 || w <= 0 || C <= 0 || H <= 0 || W <= 0) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(softmax_max_resize_kernel<T>, dim3(grid_for((long)N * h * w)), dim3(TPB), 0, stream,
                       (const T*)logits, scores, classes, N, h, w, C, H, W);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// Cross-entropy of nearest-resized logits against targets int64 [N][H][W] with ignore_index.
// acc: SIHL_CE_ACC_FLOATS floats; acc[0] = loss sum, acc[1] = valid count on return (the rest: per-workgroup partials, summed
// in a fixed order - bit-reproducible); dl [N][h][w][C] = d(sum loss)/d logits * inv_count[0]
// (inv_count: device scalar = 1 / #valid targets for the mean reduction; NULL = 1).
int sihl_ce_resize(const void* logits, const long* targets, long ignore_index, const float* inv_count, void* dl,
                   float* acc, int N, int h, int w, int C, int H, int W, int dtype, hipStream_t stream) {
  if (!logits || !targets || !dl || !acc || N <= 0 || h <= 0 || w <= 0 || C <= 0 || C > 4096) return SIHL_EARG;
  const size_t lds = (size_t)(TPB / 64) * 2 * C * sizeof(float);
  const int nblk = grid_for_waves((long)N * h * w);
  DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(ce_resize_kernel<T>, dim3(nblk), dim3(TPB), lds, stream,
                       (const T*)logits, targets, ignore_index, inv_count, (T*)dl, acc, N, h, w, C, H, W);
  });
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, stream, acc, nblk);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
