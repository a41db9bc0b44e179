// HBM-bound NHWC elementwise / reduction kernels of the sihl hot path (gfx950).
// Every kernel moves 16-byte channel vectors (8 bf16 / 4 fp32 per lane), computes in fp32 and is
// written so that each logical tensor is read once and written once:
//   * BatchNorm batch-statistics finalize + per-channel affine/activation   (convblocks.py:82-85, fpn.py:26-37)
//   * FastNormalizedFusion fused with its producer: bilinear x2 upsample + 2-way weighted sum, and
//     reflect-pad binomial blur (stride 2) + 3-way weighted sum               (bifpn.py:10-17,39-53; scalers.py:36-47;
//                                                                              pooling.py:7-26)
//   * their adjoints (input grads + fusion-weight grads as block reductions)
//   * LayerNorm + SiLU of the MLP hidden layers and its backward             (heads/object_detection.py:51-61)
#include "common.h"

// Every kernel of this file runs on the training step's MAIN stream; the weight gradients run beside them on a second stream
// (1 024-thread workgroups that hold a CU's LDS but not all of its wave slots).  s_setprio is SIMD-wide, across kernels: at
// priority 3 these memory-bound kernels take the issue slot as soon as their data is back instead of queueing behind the
// neighbour's MFMA stream - step 27.10 -> 26.75 ms (profiles/r04_ewprio_lib_ab.txt, alternating processes; -DSIHL_EW_NO_PRIO is
// the A/B build).  The same on the conv kernels changed nothing (r04_mainprio_lib_ab.txt) and is not there.
#ifdef SIHL_EW_NO_PRIO
#define EW_PRIO() do { } while (0)
#else
#define EW_PRIO() SIHL_PRIO(3)
#endif

namespace {

constexpr int TPB = 256;

template <typename T> __device__ __forceinline__ void ldv(const T* p, float (&f)[16 / sizeof(T)]) {
  unpack16(*(const uint4*)p, f, T());
}
template <typename T> __device__ __forceinline__ void stv(T* p, const float (&f)[16 / sizeof(T)]) {
  *(uint4*)p = pack16(f, T());
}
// V consecutive per-channel fp32 parameters starting at channel c (c % 4 == 0, arrays 16-byte aligned) as 16-byte
// loads that are all issued before the first use; a NULL array yields the default.  (Element-wise `p ? p[c+e] : d`
// compiles to V dependent load/wait round trips - a serial prologue of several microseconds per wave.)
template <int V> __device__ __forceinline__ void ldparam(const float* __restrict__ p, int c, float (&out)[V], float dflt) {
  if (p) {
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
      const float4 t = *(const float4*)(p + c + 4 * k);
      out[4 * k] = t.x; out[4 * k + 1] = t.y; out[4 * k + 2] = t.z; out[4 * k + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < V; ++e) out[e] = dflt;
  }
}

__device__ __forceinline__ void softmax_w(const float* raw, int n, float (&w)[3]) {
  w[0] = w[1] = w[2] = 0.f;
  if (!raw) { w[0] = 1.f; return; }
  float m = raw[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, raw[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) { w[i] = expf(raw[i] - m); s += w[i]; }
  for (int i = 0; i < n; ++i) w[i] /= s;
}

// block-wide sum of up to 3 values -> this workgroup's partial row acc[blockIdx.x][0..n) (row stride 4; no atomics: the
// rows are summed in a fixed order by fusion_wgrad_kernel, so the fusion-weight gradients are bit-reproducible)
__device__ __forceinline__ void block_accumulate(float (&v)[3], int n, float* acc) {
  __shared__ float red[3][TPB / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = 0; i < n; ++i) {
    const float s = wave_sum(v[i]);
    if (lane == 0) red[i][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x < n) {
    float s = 0.f;
    for (int w = 0; w < TPB / 64; ++w) s += red[threadIdx.x][w];
    acc[(long)blockIdx.x * 4 + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------ BN finalize
// partials [R][2][C] -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; running stats updated.
__global__ void bn_finalize_kernel(const float* __restrict__ part, int R, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* mean_out,
                                   float* rstd_out, float* scale, float* shift) {
  EW_PRIO();
  // One workgroup per 4 channels (layers with few channels still spread over C/4 workgroups).  When C % 4 == 0 every
  // thread owns whole partial rows and reads the 4 sums and 4 squared sums of a row as two 16-byte loads (the
  // stage-1 layers have 4096 partial rows: 64 scalar loads per thread took ~15 us); the 256 threads are then folded
  // in double through wave shuffles and LDS.
  __shared__ double sh[2][64][5];
  const int cx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cx;
  double s = 0.0, q = 0.0;
  int nrows = 64;  // rows of `sh` that hold partial sums
  if ((C & 3) == 0) {
    double ds[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
    const int c0 = blockIdx.x * 4;
    for (int r = threadIdx.x; r < R; r += 256) {
      const float4 a = *(const float4*)(part + ((long)r * 2 + 0) * C + c0);
      const float4 b4 = *(const float4*)(part + ((long)r * 2 + 1) * C + c0);
      ds[0] += (double)a.x; ds[1] += (double)a.y; ds[2] += (double)a.z; ds[3] += (double)a.w;
      dq[0] += (double)b4.x; dq[1] += (double)b4.y; dq[2] += (double)b4.z; dq[3] += (double)b4.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { ds[k] += __shfl_xor(ds[k], o); dq[k] += __shfl_xor(dq[k], o); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // park the 4 wave totals in rows 0..3 of the table the common tail sums over
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { sh[0][wave][k] = ds[k]; sh[1][wave][k] = dq[k]; }
    }
    nrows = 4;
  } else {
    if (c < C)
      for (int r = ry; r < R; r += 64) {
        s += (double)part[((long)r * 2 + 0) * C + c];
        q += (double)part[((long)r * 2 + 1) * C + c];
      }
    sh[0][ry][cx] = s;
    sh[1][ry][cx] = q;
  }
  __syncthreads();
  if (ry == 0 && c < C) {
    s = q = 0.0;
    for (int k = 0; k < nrows; ++k) { s += sh[0][k][cx]; q += sh[1][k][cx]; }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
    scale[c] = g * rstd;
    shift[c] = b - (float)mean * g * rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// scale/shift from running statistics (eval mode)
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  EW_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * rstd;
  shift[c] = b - rm[c] * g * rstd;
}

// ------------------------------------------------------------------ y = act(x*scale[c] + shift[c])
// sigmoid with the hardware exp2 / reciprocal instructions (1 ulp each): two quarter-rate transcendentals instead of
// the ~40-instruction expf + IEEE division, which made the SiLU kernels VALU-bound instead of HBM-bound.
__device__ __forceinline__ float fast_sigmoid(float v) {
  const float e = __builtin_amdgcn_exp2f(fminf(-v * 1.4426950408889634f, 126.f));
  return __builtin_amdgcn_rcpf(1.f + e);
}
template <int ACT> __device__ __forceinline__ float act_c(float v) {
  if (ACT == SIHL_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == SIHL_ACT_SILU) return v * fast_sigmoid(v);
  if (ACT == SIHL_ACT_SIGMOID) return fast_sigmoid(v);
  return v;
}
template <int ACT> __device__ __forceinline__ float act_grad_c(float v) {
  if (ACT == SIHL_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (ACT == SIHL_ACT_SILU) { const float s = fast_sigmoid(v); return s * (1.f + v * (1.f - s)); }
  if (ACT == SIHL_ACT_SIGMOID) { const float s = fast_sigmoid(v); return s * (1.f - s); }
  return 1.f;
}

// The launchers pick a grid whose total thread count is a multiple of cvec whenever cvec divides it, so a thread
// always meets the same channel vector and keeps its scale/shift in registers (FIXED = true).
template <typename T, int ACT, bool FIXED>
__global__ void affine_act_kernel(const T* __restrict__ x, T* __restrict__ y, long nvec, int cvec,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const T* __restrict__ res, unsigned char* __restrict__ mask) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  float sc[V], sh[V];
  const long i0 = (long)blockIdx.x * TPB + threadIdx.x;
  if (FIXED) {
    const int c = (int)(i0 % cvec) * V;
    ldparam<V>(scale, c, sc, 1.f);
    ldparam<V>(scale ? shift : nullptr, c, sh, 0.f);
  }
  // (a sweep from the tensor's END - the tail the producing conv left in the Infinity Cache first - was measured and
  // changes nothing here: 15.29 vs 15.30 ms per 10 steps, profiles/r03_ew_order_ab.txt)
  for (long i = i0; i < nvec; i += (long)gridDim.x * TPB) {
    if (!FIXED) {
      const int c = (int)(i % cvec) * V;
      ldparam<V>(scale, c, sc, 1.f);
      ldparam<V>(scale ? shift : nullptr, c, sh, 0.f);
    }
    float f[V];
    ldv(x + i * V, f);
    if (res) {  // residual merge of a ResNet block: act(BN(x) + res) in one pass
      float r[V];
      ldv(res + i * V, r);
#pragma unroll
      for (int e = 0; e < V; ++e) f[e] = act_c<ACT>(f[e] * sc[e] + sh[e] + r[e]);
      if (mask) {  // bit e = (output e > 0): the ReLU mask the block's backward needs, 1/16 of re-reading y for it
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) m |= (f[e] > 0.f ? 1u : 0u) << e;
        mask[i] = (unsigned char)m;
      }
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) f[e] = act_c<ACT>(f[e] * sc[e] + sh[e]);
    }
    stv(y + i * V, f);
  }
}

// dx = dy * act'(x*scale+shift) * scale ; used for stand-alone activations (silu / sigmoid / relu)
template <typename T, int ACT, bool FIXED>
__global__ void affine_act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long nvec,
                                      int cvec, const float* __restrict__ scale, const float* __restrict__ shift) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  float sc[V], sh[V];
  const long i0 = (long)blockIdx.x * TPB + threadIdx.x;
  if (FIXED) {
    const int c = (int)(i0 % cvec) * V;
    ldparam<V>(scale, c, sc, 1.f);
    ldparam<V>(scale ? shift : nullptr, c, sh, 0.f);
  }
  for (long i = i0; i < nvec; i += (long)gridDim.x * TPB) {
    if (!FIXED) {
      const int c = (int)(i % cvec) * V;
      ldparam<V>(scale, c, sc, 1.f);
      ldparam<V>(scale ? shift : nullptr, c, sh, 0.f);
    }
    float f[V], g[V];
    ldv(x + i * V, f);
    ldv(dy + i * V, g);
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = g[e] * act_grad_c<ACT>(f[e] * sc[e] + sh[e]) * sc[e];
    stv(dx + i * V, g);
  }
}


// ------------------------------------------------------------------ out = act(a + b)  (ResNet residual merge)
template <typename T, int ACT>
__global__ void add_act_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long nvec) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    float fa[V], fb[V];
    ldv(a + i * V, fa);
    ldv(b + i * V, fb);
#pragma unroll
    for (int e = 0; e < V; ++e) fa[e] = act_c<ACT>(fa[e] + fb[e]);
    stv(out + i * V, fa);
  }
}

// ------------------------------------------------------------------ bilinear x2 (align_corners=False) helpers
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp up2_src(int dst, int in_size) {
  float src = 0.5f * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp r;
  r.i0 = (int)src;
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = src - r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// out = w0 * up2(a) + w1 * b          a: [N][H/2][W/2][C], b/out: [N][H][W][C]
// One output ROW per blockIdx.y (the row's image, source rows and row weights are wave-uniform: scalar registers),
// blockIdx.x * TPB + tid = (column, channel vector) of that row, 32-bit index arithmetic.  The earlier form - one flat
// 64-bit index per vector, split by three runtime divisions - spent ~600 instructions per 16-byte output, and that, not
// HBM, set its pace (L3: 36 us for 75 MB; rocprofv3 r03_ns_forward).
template <typename T>
__global__ void fuse_up2_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ wraw,
                                T* __restrict__ out, int N, int H, int W, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const unsigned cvec = C / V, h2 = H / 2, w2 = W / 2;
  const unsigned xc = blockIdx.x * TPB + threadIdx.x;
  if (xc >= (unsigned)W * cvec) return;
  const unsigned x = xc / cvec, cv = xc - x * cvec;
  const unsigned row = blockIdx.z * 65535u + blockIdx.y;
  if (row >= (unsigned)N * (unsigned)H) return;
  const unsigned n = row / (unsigned)H, y = row - n * (unsigned)H;
  float w[3];
  softmax_w(wraw, 2, w);
  if (!b) { w[0] = 1.f; w[1] = 0.f; }
  const Lerp ly = up2_src((int)y, (int)h2), lx = up2_src((int)x, (int)w2);
  const T* a0 = a + ((long)n * h2 + ly.i0) * w2 * C + cv * V;
  const T* a1 = a + ((long)n * h2 + ly.i1) * w2 * C + cv * V;
  const long i = ((long)row * W + x) * C + cv * V;
  float f00[V], f01[V], f10[V], f11[V], o[V];
  ldv(a0 + lx.i0 * C, f00);
  ldv(a0 + lx.i1 * C, f01);
  ldv(a1 + lx.i0 * C, f10);
  ldv(a1 + lx.i1 * C, f11);
  if (b) ldv(b + i, o);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const float up = node_up2(ly.l0, ly.l1, lx.l0, lx.l1, f00[e], f01[e], f10[e], f11[e]);
    o[e] = b ? node_fuse2(w[0], w[1], up, o[e]) : up;
  }
  stv(out + i, o);
}

// high-res pass of the adjoint: db = w1*dout, g0 += <dout, up2(a)>, g1 += <dout, b>
template <typename T>
__global__ void fuse_up2_bwd_hi_kernel(const T* __restrict__ dout, const T* __restrict__ a, const T* __restrict__ b,
                                       const float* __restrict__ wraw, T* __restrict__ db, float* gacc, int N, int H,
                                       int W, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, h2 = H / 2, w2 = W / 2;
  const long nvec = (long)N * H * W * cvec;
  float w[3];
  softmax_w(wraw, 2, w);
  float g[3] = {0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    float d[V], bb[V];
    ldv(dout + i * V, d);
    ldv(b + i * V, bb);
    if (gacc) {
      const Lerp ly = up2_src(y, h2), lx = up2_src(x, w2);
      const T* an = a + (long)n * h2 * w2 * C + cv * V;
      float f00[V], f01[V], f10[V], f11[V];
      ldv(an + ((long)ly.i0 * w2 + lx.i0) * C, f00);
      ldv(an + ((long)ly.i0 * w2 + lx.i1) * C, f01);
      ldv(an + ((long)ly.i1 * w2 + lx.i0) * C, f10);
      ldv(an + ((long)ly.i1 * w2 + lx.i1) * C, f11);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float up = ly.l0 * (lx.l0 * f00[e] + lx.l1 * f01[e]) + ly.l1 * (lx.l0 * f10[e] + lx.l1 * f11[e]);
        g[0] += d[e] * up;
        g[1] += d[e] * bb[e];
      }
    }
    if (db) {
#pragma unroll
      for (int e = 0; e < V; ++e) d[e] *= w[1];
      stv(db + i * V, d);
    }
  }
  if (gacc) block_accumulate(g, 2, gacc);
}

// low-res pass of the adjoint: da[i][j] = w0 * sum_{y,x} cy(y,i) cx(x,j) dout[y][x]
template <typename T>
__global__ void up2_adjoint_kernel(const T* __restrict__ dout, const float* __restrict__ wraw, T* __restrict__ da,
                                   int N, int H, int W, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, h2 = H / 2, w2 = W / 2;
  const long nvec = (long)N * h2 * w2 * cvec;
  float w[3];
  softmax_w(wraw, 2, w);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int xj = (int)(pix % w2); pix /= w2;
    const int yi = (int)(pix % h2);
    const int n = (int)(pix / h2);
    float wy[4], wx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int y = 2 * yi - 1 + t, x = 2 * xj - 1 + t;
      wy[t] = wx[t] = 0.f;
      if (y >= 0 && y < H) { const Lerp l = up2_src(y, h2); wy[t] = (l.i0 == yi ? l.l0 : 0.f) + (l.i1 == yi ? l.l1 : 0.f); }
      if (x >= 0 && x < W) { const Lerp l = up2_src(x, w2); wx[t] = (l.i0 == xj ? l.l0 : 0.f) + (l.i1 == xj ? l.l1 : 0.f); }
    }
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const T* dn = dout + (long)n * H * W * C + cv * V;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      if (wy[ty] == 0.f) continue;
      const int y = 2 * yi - 1 + ty;
#pragma unroll
      for (int tx = 0; tx < 4; ++tx) {
        if (wx[tx] == 0.f) continue;
        const int x = 2 * xj - 1 + tx;
        float d[V];
        ldv(dn + ((long)y * W + x) * C, d);
        const float k = wy[ty] * wx[tx];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += k * d[e];
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] *= w[0];
    stv(da + i * V, acc);
  }
}


// ------------------------------------------------------------------ nearest x2 upsample + add (FPN top-down, fpn.py:43-48)
template <typename T>
__global__ void nearest_up2_add_kernel(const T* __restrict__ lo, const T* __restrict__ skip, T* __restrict__ out,
                                       int N, int H, int W, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, h2 = H / 2, w2 = W / 2;
  const long nvec = (long)N * H * W * cvec;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    float a[V], b[V];
    ldv(lo + ((((long)n * h2 + y / 2) * w2 + x / 2) * cvec + cv) * V, a);
    ldv(skip + i * V, b);
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += b[e];
    stv(out + i * V, a);
  }
}

// d_lo[i][j] = sum of the 2x2 block of dout
template <typename T>
__global__ void nearest_up2_adjoint_kernel(const T* __restrict__ dout, T* __restrict__ dlo, int N, int H, int W,
                                           int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V, h2 = H / 2, w2 = W / 2;
  const long nvec = (long)N * h2 * w2 * cvec;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % w2); pix /= w2;
    const int y = (int)(pix % h2);
    const int n = (int)(pix / h2);
    float acc[V], d[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        ldv(dout + ((((long)n * H + 2 * y + dy) * W + 2 * x + dx) * cvec + cv) * V, d);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += d[e];
      }
    stv(dlo + i * V, acc);
  }
}

// ------------------------------------------------------------------ general bilinear resize, align_corners=False
__device__ __forceinline__ Lerp resize_src(int dst, int in_size, float scale) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp r;
  r.i0 = min((int)src, in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = src - r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// out[N][Ho][Wo][C] = resize(a[N][H][W][C]) (+ add)
template <typename T>
__global__ void resize_bilinear_kernel(const T* __restrict__ a, const T* __restrict__ add, T* __restrict__ out, int N,
                                       int H, int W, int Ho, int Wo, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  const long nvec = (long)N * Ho * Wo * cvec;
  const float sy = (float)H / Ho, sx = (float)W / Wo;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % Wo); pix /= Wo;
    const int y = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const Lerp ly = resize_src(y, H, sy), lx = resize_src(x, W, sx);
    const T* an = a + (long)n * H * W * C + cv * V;
    float f00[V], f01[V], f10[V], f11[V], o[V];
    ldv(an + ((long)ly.i0 * W + lx.i0) * C, f00);
    ldv(an + ((long)ly.i0 * W + lx.i1) * C, f01);
    ldv(an + ((long)ly.i1 * W + lx.i0) * C, f10);
    ldv(an + ((long)ly.i1 * W + lx.i1) * C, f11);
    if (add) ldv(add + i * V, o);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float v = ly.l0 * (lx.l0 * f00[e] + lx.l1 * f01[e]) + ly.l1 * (lx.l0 * f10[e] + lx.l1 * f11[e]);
      o[e] = add ? o[e] + v : v;
    }
    stv(out + i * V, o);
  }
}

// da[y][x] = sum over the output pixels whose 2x2 footprint touches (y, x)
template <typename T>
__global__ void resize_bilinear_adjoint_kernel(const T* __restrict__ dout, T* __restrict__ da, int N, int H, int W,
                                               int Ho, int Wo, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  const long nvec = (long)N * H * W * cvec;
  const float sy = (float)H / Ho, sx = (float)W / Wo;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    // conservative candidate ranges; exact membership is re-tested with the forward formula
    const int oy_lo = max(0, (int)floorf((y - 0.5f) / sy - 0.5f) - 1), oy_hi = min(Ho - 1, (int)ceilf((y + 1.5f) / sy - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf((x - 0.5f) / sx - 0.5f) - 1), ox_hi = min(Wo - 1, (int)ceilf((x + 1.5f) / sx - 0.5f) + 1);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const T* dn = dout + (long)n * Ho * Wo * C + cv * V;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const Lerp ly = resize_src(oy, H, sy);
      const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const Lerp lx = resize_src(ox, W, sx);
        const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
        if (wx == 0.f) continue;
        float d[V];
        ldv(dn + ((long)oy * Wo + ox) * C, d);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += wy * wx * d[e];
      }
    }
    stv(da + i * V, acc);
  }
}

// ------------------------------------------------------------------ blur (reflect, [1,2,1]^2/16, stride 2) + fuse
// out = w0*blur(a) + w1*b + w2*c       a: [N][H][W][C]; b,c,out: [N][Ho][Wo][C]; b==null -> out = blur(a)
// a_scale / a_shift (optional, per channel): `a` stands for a * scale + shift - the BatchNorm affine of the conv block
// that produced it (training: conv -> ReLU -> BN), applied to the blurred value (the taps sum to 1) instead of in a
// pass of its own over the full-resolution tensor.
// One output row per blockIdx.y (image and the three reflected source rows: scalar), 32-bit arithmetic - see fuse_up2_kernel.
template <typename T>
__global__ void blur_fuse_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                 const float* __restrict__ wraw, const float* __restrict__ a_scale,
                                 const float* __restrict__ a_shift, T* __restrict__ out, int N, int H, int W, int Ho,
                                 int Wo, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const unsigned cvec = C / V;
  const unsigned xc = blockIdx.x * TPB + threadIdx.x;
  if (xc >= (unsigned)Wo * cvec) return;
  const unsigned ox = xc / cvec, cv = xc - ox * cvec;
  const unsigned row = blockIdx.z * 65535u + blockIdx.y;
  if (row >= (unsigned)N * (unsigned)Ho) return;
  const unsigned n = row / (unsigned)Ho, oy = row - n * (unsigned)Ho;
  float w[3];
  softmax_w(b ? wraw : nullptr, 3, w);
  const float k1[3] = {0.25f, 0.5f, 0.25f};
  const T* an = a + (long)n * H * W * C + cv * V;
  int xs[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) xs[dx] = reflect1(2 * (int)ox + dx - 1, W) * C;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const T* ar = an + (long)reflect1(2 * (int)oy + dy - 1, H) * W * C;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float f[V];
      ldv(ar + xs[dx], f);
      const float k = k1[dy] * k1[dx];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = __fmaf_rn(k, f[e], acc[e]);
    }
  }
  if (a_scale) {
    float sc[V], sf[V];
    ldparam<V>(a_scale, cv * V, sc, 1.f);
    ldparam<V>(a_shift, cv * V, sf, 0.f);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = __fmaf_rn(acc[e], sc[e], sf[e]);
  }
  const long i = ((long)row * Wo + ox) * C + cv * V;
  if (b) {
    float fb[V], fc[V];
    ldv(b + i, fb);
    ldv(c + i, fc);
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = node_fuse3(w[0], w[1], w[2], acc[e], fb[e], fc[e]);
  }
  stv(out + i, acc);
}

// low-res pass of the adjoint: db = w1*dout, dc = w2*dout, g += <dout, {blur(a), b, c}>
template <typename T>
__global__ void blur_fuse_bwd_lo_kernel(const T* __restrict__ dout, const T* __restrict__ a, const T* __restrict__ b,
                                        const T* __restrict__ c, const float* __restrict__ wraw,
                                        const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                        T* __restrict__ db, T* __restrict__ dc, float* gacc, int N, int H, int W, int Ho,
                                        int Wo, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  const long nvec = (long)N * Ho * Wo * cvec;
  float w[3];
  softmax_w(wraw, 3, w);
  const float k1[3] = {0.25f, 0.5f, 0.25f};
  float g[3] = {0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    float d[V], fb[V], fc[V];
    ldv(dout + i * V, d);
    ldv(b + i * V, fb);
    ldv(c + i * V, fc);
    if (gacc) {
      const T* an = a + (long)n * H * W * C + cv * V;
      float bl[V];
#pragma unroll
      for (int e = 0; e < V; ++e) bl[e] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int y = reflect1(2 * oy + dy - 1, H);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int x = reflect1(2 * ox + dx - 1, W);
          float f[V];
          ldv(an + ((long)y * W + x) * C, f);
          const float k = k1[dy] * k1[dx];
#pragma unroll
          for (int e = 0; e < V; ++e) bl[e] += k * f[e];
        }
      }
      if (a_scale) {
        float sc[V], sf[V];
        ldparam<V>(a_scale, cv * V, sc, 1.f);
        ldparam<V>(a_shift, cv * V, sf, 0.f);
#pragma unroll
        for (int e = 0; e < V; ++e) bl[e] = bl[e] * sc[e] + sf[e];
      }
#pragma unroll
      for (int e = 0; e < V; ++e) { g[0] += d[e] * bl[e]; g[1] += d[e] * fb[e]; g[2] += d[e] * fc[e]; }
    }
    if (db) {
#pragma unroll
      for (int e = 0; e < V; ++e) fb[e] = w[1] * d[e];
      stv(db + i * V, fb);
    }
    if (dc) {
#pragma unroll
      for (int e = 0; e < V; ++e) fc[e] = w[2] * d[e];
      stv(dc + i * V, fc);
    }
  }
  if (gacc) block_accumulate(g, 3, gacc);
}

// high-res pass of the adjoint: da[y][x] = w0 * sum_{oy,ox} ky(y,oy) kx(x,ox) dout[oy][ox]
template <typename T>
__global__ void blur_adjoint_kernel(const T* __restrict__ dout, const float* __restrict__ wraw, T* __restrict__ da,
                                    int N, int H, int W, int Ho, int Wo, int C) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  const long nvec = (long)N * H * W * cvec;
  float w[3];
  softmax_w(wraw, 3, w);
  const float k1[3] = {0.25f, 0.5f, 0.25f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    const int cv = (int)(i % cvec);
    long pix = i / cvec;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const int n = (int)(pix / H);
    // candidate output rows: those whose 3-tap window (after reflection) can touch y
    const int oy_lo = max(0, (y - 2) / 2), ox_lo = max(0, (x - 2) / 2);
    float wy[3], wx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      wy[t] = wx[t] = 0.f;
      const int oy = oy_lo + t, ox = ox_lo + t;
      if (oy < Ho)
        for (int d = 0; d < 3; ++d) if (reflect1(2 * oy + d - 1, H) == y) wy[t] += k1[d];
      if (ox < Wo)
        for (int d = 0; d < 3; ++d) if (reflect1(2 * ox + d - 1, W) == x) wx[t] += k1[d];
    }
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const T* dn = dout + (long)n * Ho * Wo * C + cv * V;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      if (wy[ty] == 0.f) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        if (wx[tx] == 0.f) continue;
        float d[V];
        ldv(dn + ((long)(oy_lo + ty) * Wo + (ox_lo + tx)) * C, d);
        const float k = wy[ty] * wx[tx];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += k * d[e];
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] *= w[0];
    stv(da + i * V, acc);
  }
}


// ------------------------------------------------------------------ stand-alone FastNormalizedFusion (n = 2 or 3)
template <typename T>
__global__ void fuse_sum_kernel(const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2,
                                const float* __restrict__ wraw, T* __restrict__ out, long nvec, int n) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  float w[3];
  softmax_w(wraw, n, w);
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    float a[V], b[V], c[V];
    ldv(x0 + i * V, a);
    ldv(x1 + i * V, b);
    if (n > 2) ldv(x2 + i * V, c);
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = w[0] * a[e] + w[1] * b[e] + (n > 2 ? w[2] * c[e] : 0.f);
    stv(out + i * V, a);
  }
}

template <typename T>
__global__ void fuse_sum_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x0, const T* __restrict__ x1,
                                    const T* __restrict__ x2, const float* __restrict__ wraw, T* __restrict__ d0,
                                    T* __restrict__ d1, T* __restrict__ d2, float* gacc, long nvec, int n) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  float w[3];
  softmax_w(wraw, n, w);
  float g[3] = {0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (long)gridDim.x * TPB) {
    float d[V], a[V], b[V], c[V], o[V];
    ldv(dout + i * V, d);
    if (gacc) {
      ldv(x0 + i * V, a);
      ldv(x1 + i * V, b);
      if (n > 2) ldv(x2 + i * V, c);
#pragma unroll
      for (int e = 0; e < V; ++e) { g[0] += d[e] * a[e]; g[1] += d[e] * b[e]; if (n > 2) g[2] += d[e] * c[e]; }
    }
    if (d0) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = w[0] * d[e];
      stv(d0 + i * V, o);
    }
    if (d1) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = w[1] * d[e];
      stv(d1 + i * V, o);
    }
    if (n > 2 && d2) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = w[2] * d[e];
      stv(d2 + i * V, o);
    }
  }
  if (gacc) block_accumulate(g, n, gacc);
}

// g = sum of the producing kernel's per-workgroup partial rows part[nblocks][4] (fixed order: strided per thread, then a
// tree through LDS); d raw_j = w_j * (g_j - sum_i w_i g_i)  (softmax Jacobian) -> dw_raw
__global__ void fusion_wgrad_kernel(const float* wraw, const float* __restrict__ part, int nblocks, float* dw_raw, int n) {
  EW_PRIO();
  __shared__ float red[3][256];
  float a[3] = {0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < nblocks; b += 256)
    for (int i = 0; i < n; ++i) a[i] += part[(long)b * 4 + i];
  for (int i = 0; i < 3; ++i) red[i][threadIdx.x] = a[i];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int i = 0; i < n; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float w[3], g[3];
    softmax_w(wraw, n, w);
    float dot = 0.f;
    for (int i = 0; i < n; ++i) { g[i] = red[i][0]; dot += w[i] * g[i]; }
    for (int j = 0; j < n; ++j) dw_raw[j] = w[j] * (g[j] - dot);
  }
}

// ------------------------------------------------------------------ column sums: partial stage + final stage
// mode 0: sums of (dy) and (dy * xhat)            [act -> norm, s = post-activation tensor]
// mode 1: g = dy*act'(xhat*gamma+beta); sums of g and g*xhat   [norm -> act, s = pre-norm tensor]
// Thread -> (channel vector cv, row lane rl): a block's rows are dealt to nrl = TPB/cvec row lanes, folded
// through LDS into ONE partial row per block: part[block][2][C].
// MASK 1 / 2 (the tail of a residual block, y = relu(BN(s) + identity)): the incoming gradient is first masked by y > 0
// (1: read from y; 2: from the byte-per-vector mask the forward wrote - a sixteenth of y's bytes) - the
// gradient of both merge inputs, written to `dres` for the identity branch and for the apply pass - in the SAME pass that
// folds it into the column sums: one kernel and 4 tensor passes (s, dy, y read; dres written) where a separate ReLU
// backward + this reduction took 5.
template <typename T, int MODE, int ACT, int MASK = 0>
__global__ void norm_bwd_reduce_kernel(const T* __restrict__ s, const T* __restrict__ dy, long rows, int C,
                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ part, int rows_per_block, int nrl, int rev,
                                       const void* __restrict__ ymask = nullptr, T* __restrict__ dres = nullptr) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  // rows of a workgroup: a contiguous range (rev 0), or - rev - row lanes dealt round by round over the whole grid and
  // walked from the tensor's END to its start.  All workgroups are resident together, so with contiguous ranges what is
  // left in the 256 MB Infinity Cache at the end is a stripe of every range; walked in descending rounds the pass ends on
  // the HEAD of s and dy, which is where the apply pass (ascending rounds) starts: its first 64 - 128 MB per tensor are
  // then served on-die.  (Deterministic either way: a workgroup's rows and their order are fixed by the launch.)
  const int rl = threadIdx.x / cvec;
  long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block), rstep = nrl;
  if (rev) {
    const long round = (long)gridDim.x * nrl;
    r0 = (long)blockIdx.x * nrl;                                      // first round's row of lane 0
    r0 = r0 + rl < rows ? r0 + (rows - 1 - r0 - rl) / round * round : rows;  // last round holding a row for this lane
    r1 = rows;
    rstep = -round;
  }
  extern __shared__ float red_lds[];  // [nrl][2][C]
  for (int cv = threadIdx.x % cvec; cv < cvec && rl < nrl; cv += TPB) {
    float sb[V], sg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sb[e] = sg[e] = 0.f;
    float mu[V], rs[V], ga[V], be[V];
    ldparam<V>(mean, cv * V, mu, 0.f);
    ldparam<V>(rstd, cv * V, rs, 1.f);
    ldparam<V>(MODE == 1 ? gamma : nullptr, cv * V, ga, 1.f);
    ldparam<V>(MODE == 1 ? beta : nullptr, cv * V, be, 0.f);
    for (long r = r0 + rl; r >= 0 && r < r1; r += rstep) {
      float fs[V], fd[V];
      ldv(s + (r * cvec + cv) * V, fs);
      ldv(dy + (r * cvec + cv) * V, fd);
      if (MASK == 1) {  // mask from the block's output y
        float fy[V];
        ldv((const T*)ymask + (r * cvec + cv) * V, fy);
#pragma unroll
        for (int e = 0; e < V; ++e) fd[e] = fy[e] > 0.f ? fd[e] : 0.f;
        stv(dres + (r * cvec + cv) * V, fd);
      } else if (MASK == 2) {  // mask bits written by the forward (sihl_affine_add_act): a byte per vector
        const unsigned m = ((const unsigned char*)ymask)[r * cvec + cv];
#pragma unroll
        for (int e = 0; e < V; ++e) fd[e] = ((m >> e) & 1u) ? fd[e] : 0.f;
        stv(dres + (r * cvec + cv) * V, fd);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float xh = (fs[e] - mu[e]) * rs[e];
        float g = fd[e];
        if (MODE == 1) g *= act_grad_c<ACT>(xh * ga[e] + be[e]);
        sb[e] += g;
        sg[e] += g * xh;
      }
    }
    // fold the nrl row lanes of this block through LDS: one partial row per block
#pragma unroll
    for (int e = 0; e < V; ++e) {
      red_lds[(rl * 2 + 0) * C + cv * V + e] = sb[e];
      red_lds[(rl * 2 + 1) * C + cv * V + e] = sg[e];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * C; idx += TPB) {
    float t = 0.f;
    for (int l = 0; l < nrl; ++l) t += red_lds[l * 2 * C + idx];
    part[(long)blockIdx.x * 2 * C + idx] = t;
  }
}

// out[k][c] = sum_r part[r][k][c]   (k < K).  256 threads = 4 columns x 64 row lanes: fp32 partials are summed in
// double per lane, the 64 lanes are combined through LDS (grid = K*C/4 workgroups, so even 64-channel layers
// spread over 32+ CUs).
__global__ void colsum_finalize_kernel(const float* __restrict__ part, int R, int K, int C, float* __restrict__ out0,
                                       float* __restrict__ out1) {
  EW_PRIO();
  __shared__ double sh[64][5];
  const int cx = threadIdx.x & 3, ry = threadIdx.x >> 2;
  const int idx = blockIdx.x * 4 + cx, KC = K * C;
  double s0 = 0.0, s1 = 0.0;
  if (idx < KC) {
    int r = ry;
    for (; r + 64 < R; r += 128) {
      s0 += (double)part[(long)r * KC + idx];
      s1 += (double)part[(long)(r + 64) * KC + idx];
    }
    for (; r < R; r += 64) s0 += (double)part[(long)r * KC + idx];
  }
  sh[ry][cx] = s0 + s1;
  __syncthreads();
  if (ry == 0 && idx < KC) {
    double t = 0.0;
    for (int k = 0; k < 64; ++k) t += sh[k][cx];
    if (idx < C) out0[idx] = (float)t; else out1[idx - C] = (float)t;  // K <= 2: row 0 -> out0, row 1 -> out1
  }
}

// dz for conv->act->BN (MODE 0) or conv->BN->act (MODE 1); batch_stats: include the mean/var terms
template <typename T, int MODE, int ACT, bool FIXED>
__global__ void norm_bwd_apply_kernel(const T* __restrict__ s, const T* __restrict__ dy, T* __restrict__ dz, long nvec,
                                      int cvec, const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ sum_g /*dbeta*/, const float* __restrict__ sum_gx /*dgamma*/,
                                      float inv_count,
                                      int batch_stats) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  const int C = cvec * V;
  float mu[V], rs[V], ga[V], be[V], k0[V], k1[V];
  auto load_params = [&](int c0) {
    ldparam<V>(mean, c0, mu, 0.f);
    ldparam<V>(rstd, c0, rs, 1.f);
    ldparam<V>(gamma, c0, ga, 1.f);
    ldparam<V>(MODE == 1 ? beta : nullptr, c0, be, 0.f);
    ldparam<V>(batch_stats ? sum_g : nullptr, c0, k0, 0.f);
    ldparam<V>(batch_stats ? sum_gx : nullptr, c0, k1, 0.f);
#pragma unroll
    for (int e = 0; e < V; ++e) { k0[e] *= inv_count; k1[e] *= inv_count; }
  };
  const long i0 = (long)blockIdx.x * TPB + threadIdx.x;
  if (FIXED) load_params((int)(i0 % cvec) * V);
  for (long i = i0; i < nvec; i += (long)gridDim.x * TPB) {
    if (!FIXED) load_params((int)(i % cvec) * V);
    float fs[V], fd[V];
    ldv(s + i * V, fs);
    ldv(dy + i * V, fd);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float xh = (fs[e] - mu[e]) * rs[e];
      float g = fd[e];
      if (MODE == 1) g *= act_grad_c<ACT>(xh * ga[e] + be[e]);
      float dr = (g - k0[e] - xh * k1[e]) * ga[e] * rs[e];
      if (MODE == 0) dr *= act_grad_c<ACT>(fs[e]);  // s is post-activation: relu mask from s > 0
      fd[e] = dr;
    }
    stv(dz + i * V, fd);
  }
}

// ------------------------------------------------------------------ LayerNorm + activation over rows of C
// SUB lanes cooperate on one row (64/SUB rows per wave); a lane owns NK 16-byte chunks: sub, sub+SUB, ...
// (C <= NK*SUB*V).  NK = 1 whenever the row fits (256 bf16 channels = 32 lanes x 8): half the live registers of the
// two-chunk form, which spilled in the backward kernel.
template <int SUB> __device__ __forceinline__ float sub_sum(float v) {
#pragma unroll
  for (int o = SUB / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int SUB, int NK, int ACT>
__global__ void __launch_bounds__(TPB)
layernorm_act_kernel(const T* __restrict__ z, T* __restrict__ y, long rows, int C, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float eps, float* __restrict__ mean_out,
                     float* __restrict__ rstd_out) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T), RPW = 64 / SUB;
  const int cvec = C / V, lane = threadIdx.x & 63, sub = lane % SUB, rsel = lane / SUB;
  const long wave = ((long)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TPB) >> 6;
  const float inv_c = 1.f / (float)C;
  float ga[NK][V], be[NK][V];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int cv = sub + SUB * k;
    ldparam<V>(cv < cvec ? gamma : nullptr, cv * V, ga[k], 0.f);
    ldparam<V>(cv < cvec ? beta : nullptr, cv * V, be[k], 0.f);
  }
  for (long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const long r = r0 + rsel;
    const bool rok = r < rows;
    float f[NK][V];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int cv = sub + SUB * k;
#pragma unroll
      for (int e = 0; e < V; ++e) f[k][e] = 0.f;
      if (rok && cv < cvec) ldv(z + (r * cvec + cv) * V, f[k]);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int e = 0; e < V; ++e) s += f[k][e];
    const float mu = sub_sum<SUB>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
      if (sub + SUB * k < cvec) {
#pragma unroll
        for (int e = 0; e < V; ++e) { const float d = f[k][e] - mu; q += d * d; }
      }
    const float rs = 1.f / sqrtf(sub_sum<SUB>(q) * inv_c + eps);
    if (rok && sub == 0 && mean_out) { mean_out[r] = mu; rstd_out[r] = rs; }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int cv = sub + SUB * k;
      if (rok && cv < cvec) {
#pragma unroll
        for (int e = 0; e < V; ++e) f[k][e] = act_c<ACT>((f[k][e] - mu) * rs * ga[k][e] + be[k][e]);
        stv(y + (r * cvec + cv) * V, f[k]);
      }
    }
  }
}

// dz = rstd * (gh - mean_c(gh) - xhat * mean_c(gh*xhat)), gh = dy*act'(u)*gamma ; per-wave column partials
// part: [gridDim.x][2][C] (dbeta, dgamma), the workgroup's waves folded through LDS
template <typename T, int SUB, int NK, int ACT>
__global__ void __launch_bounds__(TPB)
layernorm_act_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dy, T* __restrict__ dz, long rows, int C,
                         const float* __restrict__ gamma, const float* __restrict__ beta,
                         const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ part) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T), RPW = 64 / SUB;
  const int cvec = C / V, lane = threadIdx.x & 63, sub = lane % SUB, rsel = lane / SUB;
  const long wave = ((long)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * TPB) >> 6;
  const float inv_c = 1.f / (float)C;
  float pb[NK][V], pg[NK][V], ga[NK][V], be[NK][V];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int cv = sub + SUB * k;
#pragma unroll
    for (int e = 0; e < V; ++e) pb[k][e] = pg[k][e] = 0.f;
    ldparam<V>(cv < cvec ? gamma : nullptr, cv * V, ga[k], 0.f);
    ldparam<V>(cv < cvec ? beta : nullptr, cv * V, be[k], 0.f);
  }
  for (long r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {
    const long r = r0 + rsel;
    const bool rok = r < rows;
    float xh[NK][V], gh[NK][V];
#pragma unroll
    for (int k = 0; k < NK; ++k) {  // issue every load of the row group before the first use
      const int cv = sub + SUB * k;
#pragma unroll
      for (int e = 0; e < V; ++e) xh[k][e] = gh[k][e] = 0.f;
      if (rok && cv < cvec) {
        ldv(z + (r * cvec + cv) * V, xh[k]);
        ldv(dy + (r * cvec + cv) * V, gh[k]);
      }
    }
    const float mu = rok ? mean[r] : 0.f, rs = rok ? rstd[r] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float x = (xh[k][e] - mu) * rs;
        const float g = gh[k][e] * act_grad_c<ACT>(x * ga[k][e] + be[k][e]);
        pb[k][e] += g;
        pg[k][e] += g * x;
        xh[k][e] = x;
        gh[k][e] = g * ga[k][e];
        s1 += gh[k][e];
        s2 += gh[k][e] * x;
      }
    s1 = sub_sum<SUB>(s1) * inv_c;
    s2 = sub_sum<SUB>(s2) * inv_c;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int cv = sub + SUB * k;
      if (rok && cv < cvec) {
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = rs * (gh[k][e] - s1 - xh[k][e] * s2);
        stv(dz + (r * cvec + cv) * V, o);
      }
    }
  }
  // fold the row groups of the wave together, then lanes < SUB write the wave's partial row
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) {
#pragma unroll
      for (int o = 32; o >= SUB; o >>= 1) {
        pb[k][e] += __shfl_xor(pb[k][e], o);
        pg[k][e] += __shfl_xor(pg[k][e], o);
      }
    }
  extern __shared__ float ln_red[];  // [TPB/64 waves][2][C]
  const int wib = threadIdx.x >> 6;
  if (rsel == 0) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int cv = sub + SUB * k;
      if (cv < cvec) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          ln_red[(wib * 2 + 0) * C + cv * V + e] = pb[k][e];
          ln_red[(wib * 2 + 1) * C + cv * V + e] = pg[k][e];
        }
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * C; idx += TPB) {  // one partial row per workgroup
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) t += ln_red[w * 2 * C + idx];
    part[(long)blockIdx.x * 2 * C + idx] = t;
  }
}

// column sums of a [rows][C] tensor.  Thread = (channel [vector], row lane); the block's row lanes are folded through
// LDS into ONE partial row per block: part [nblk][C]  (a narrow bias gradient, C = 8, used to leave 256 lane rows per
// block for the finalize kernel - hundreds of microseconds for two workgroups).
template <typename T, bool VECTOR>
__global__ void colsum_partial_kernel(const T* __restrict__ x, long rows, int C, float* __restrict__ part,
                                      int rows_per_block, int nrl) {
  EW_PRIO();
  constexpr int V = 16 / sizeof(T);
  __shared__ float red[TPB * V];  // nrl * C <= TPB * V floats whenever nrl > 1
  const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float* dst = part + (long)blockIdx.x * C;
  if (VECTOR) {
    const int cvec = C / V, rl = threadIdx.x / cvec;
    for (int cv = threadIdx.x % cvec; cv < cvec && rl < nrl; cv += TPB) {
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
      long r = r0 + rl;
      for (; r + nrl < r1; r += 2 * nrl) {  // two rows in flight
        float f[V], g[V];
        ldv(x + (r * cvec + cv) * V, f);
        ldv(x + ((r + nrl) * cvec + cv) * V, g);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += f[e] + g[e];
      }
      if (r < r1) {
        float f[V];
        ldv(x + (r * cvec + cv) * V, f);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += f[e];
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if (nrl > 1) red[rl * C + cv * V + e] = acc[e]; else dst[cv * V + e] = acc[e];
      }
    }
  } else {
    const int rl = threadIdx.x / C;
    for (int c = threadIdx.x % C; c < C && rl < nrl; c += TPB) {
      float s = 0.f;
      for (long r = r0 + rl; r < r1; r += nrl) s += elem<T>::ld(x + r * C + c);
      if (nrl > 1) red[rl * C + c] = s; else dst[c] = s;
    }
  }
  if (nrl > 1) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += TPB) {
      float t = 0.f;
      for (int l = 0; l < nrl; ++l) t += red[l * C + c];
      dst[c] = t;
    }
  }
}

// [Cout][KH][KW][Cin] -> [Cin][KH][KW][Cout] with both spatial axes flipped (dgrad weights), optional dtype change
template <typename TI, typename TO>
__global__ void weight_flip_transpose_kernel(const TI* __restrict__ w, TO* __restrict__ o, int Cout, int KH, int KW,
                                             int Cin, int flip) {
  EW_PRIO();
  const long n = (long)Cout * KH * KW * Cin;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    long t = i;
    const int co = (int)(t % Cout); t /= Cout;
    const int kx = (int)(t % KW); t /= KW;
    const int ky = (int)(t % KH);
    const int ci = (int)(t / KH);
    const int sy = flip ? KH - 1 - ky : ky, sx = flip ? KW - 1 - kx : kx;
    elem<TO>::st(o + i, elem<TI>::ld(w + (((long)co * KH + sy) * KW + sx) * Cin + ci));
  }
}

inline int grid_for(long n) {
  long g = (n + TPB - 1) / TPB;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}
// grid for per-channel kernels: total threads a multiple of cvec when possible (threads then keep their channels)
inline int grid_fixed(long nvec, int cvec, bool* fixed) {
  int g = grid_for(nvec);
  *fixed = false;
  if (cvec <= TPB && TPB % cvec == 0) { *fixed = true; return g; }
  // otherwise look for a block count with (g * TPB) % cvec == 0 close below g
  for (int t = g; t >= 1 && t > g - 64; --t)
    if (((long)t * TPB) % cvec == 0) { *fixed = true; return t; }
  return g;
}

// ------------------------------------------------------------------ per-step operand copies of ALL weights
// One launch turns every fp32 master weight (logical [O][I][KH][KW], any strides) into the two bf16 operands the
// conv kernels read: w [Op][KH][KW][I] (forward / wgrad layout, rows O..Op-1 zero) and wt [I][KH][KW][Op] (spatially
// flipped when flip != 0: the dgrad operand).  Replaces one cast kernel + one flip/transpose kernel per layer per
// step (~240 launches).  A workgroup covers 2048 consecutive destination elements of one weight.
struct WeightDesc {
  const float* src; bf16_t* w; bf16_t* wt;
  long so, si, sky, skx;  // source element strides
  int O, Op, KH, KW, I, flip;
  long first_block;   // w copy: blocks of 2048 destination elements
  long first_tblock;  // wt copy: 64 o x 64 i tiles per tap
};

__device__ __forceinline__ const WeightDesc* find_desc(const WeightDesc* descs, int n, long block, bool tiles) {
  int lo = 0, hi = n - 1;  // last descriptor whose first block <= block
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((tiles ? descs[mid].first_tblock : descs[mid].first_block) <= block) lo = mid; else hi = mid - 1;
  }
  return descs + lo;
}

// w [Op][KH][KW][I]: same element order as a channels-last master weight, so reads and writes are both contiguous
__global__ void weight_prepare_kernel(const WeightDesc* __restrict__ descs, int n) {
  EW_PRIO();
  const WeightDesc d = *find_desc(descs, n, blockIdx.x, false);
  const long total = (long)d.Op * d.KH * d.KW * d.I;
  const long e0 = ((long)blockIdx.x - d.first_block) * 2048 + threadIdx.x * 8;
  if (e0 >= total) return;
  const int cnt = (int)min(8L, total - e0);
  long t = e0;
  int i = (int)(t % d.I); t /= d.I;
  int kx = (int)(t % d.KW); t /= d.KW;
  int ky = (int)(t % d.KH);
  int o = (int)(t / d.KH);
  float f[8];
  const float* row = d.src + o * d.so + i * d.si + ky * d.sky + kx * d.skx;
  if (d.si == 1 && (d.I & 7) == 0 && cnt == 8 && (((unsigned long)row) & 15) == 0) {
    // channels-last master weight: the 8 destination elements are 8 consecutive source floats
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (o < d.O) { a = *(const float4*)row; c = *(const float4*)(row + 4); }
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f[k] = (k < cnt && o < d.O) ? d.src[o * d.so + i * d.si + ky * d.sky + kx * d.skx] : 0.f;
      if (++i == d.I) { i = 0; if (++kx == d.KW) { kx = 0; if (++ky == d.KH) { ky = 0; ++o; } } }
    }
  }
  if (cnt == 8) *(uint4*)(d.w + e0) = pack16(f, bf16_t());
  else for (int k = 0; k < cnt; ++k) elem<bf16_t>::st(d.w + e0 + k, f[k]);
}

// wt [I][KH][KW][Op] (tap mirrored when flip): a 64 o x 64 i tile of one tap goes through LDS so that the master
// weight is read along i (its contiguous axis when channels-last) and the copy is written along o.  (Gathering the
// source per destination element fetched 11x the bytes: 2.2 GB for 193 MB of weights.)
__global__ void weight_prepare_t_kernel(const WeightDesc* __restrict__ descs, int n) {
  EW_PRIO();
  __shared__ float tile[64][65];
  const WeightDesc d = *find_desc(descs, n, blockIdx.x, true);
  long b = (long)blockIdx.x - d.first_tblock;
  const int tiles_i = (d.I + 63) / 64, tiles_o = (d.Op + 63) / 64;
  const int ti = (int)(b % tiles_i); b /= tiles_i;
  const int to = (int)(b % tiles_o); b /= tiles_o;
  const int tap = (int)b, ky = tap / d.KW, kx = tap - ky * d.KW;
  const int sy = d.flip ? d.KH - 1 - ky : ky, sx = d.flip ? d.KW - 1 - kx : kx;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;  // 256 threads: 64 columns x 4 rows per pass
  const int i_in = ti * 64 + lx;
  for (int r = ly; r < 64; r += 4) {
    const int o = to * 64 + r;
    tile[r][lx] = (o < d.O && i_in < d.I) ? d.src[o * d.so + i_in * d.si + sy * d.sky + sx * d.skx] : 0.f;
  }
  __syncthreads();
  const int o_out = to * 64 + lx;
  for (int r = ly; r < 64; r += 4) {
    const int i = ti * 64 + r;
    if (i < d.I && o_out < d.Op)
      elem<bf16_t>::st(d.wt + (((long)i * d.KH + ky) * d.KW + kx) * d.Op + o_out, tile[lx][r]);
  }
}


// ---- 3x3 / stride 2 / pad 1 max pooling, NHWC (the ResNet stem's pool).  Forward keeps the winning tap (0..8, first
// maximum in row-major window order, NaN wins - ATen's rule) as one byte per element; backward is a gather: every
// INPUT pixel sums the output gradients of the <= 4 windows that picked it (no atomics, one pass, deterministic).
template <typename T>
__global__ void maxpool3x3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ idx,
                                        long nvec, int H, int W, int Ho, int Wo, int cv) {
  EW_PRIO();
  constexpr int V = 16 / (int)sizeof(T);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long pix = i / cv;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const long n = pix / Ho;
    // all nine loads are issued together from clamped addresses (branches around them serialised the round trips)
    uint4 raw[9];
    bool ok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
      ok[k] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
      raw[k] = *(const uint4*)(x + (((n * H + iyc) * W + ixc) * (long)cv + c) * V);
    }
    float best[V];
    int arg[V];
    bool first = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float v[V];
      unpack16(raw[k], v, T());
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const bool take = ok[k] && (first || v[e] > best[e] || v[e] != v[e]);
        best[e] = take ? v[e] : (first ? 0.f : best[e]);
        arg[e] = take ? k : (first ? 0 : arg[e]);
      }
      first = first && !ok[k];
    }
    *(uint4*)(y + i * V) = pack16(best, T());
    unsigned char* ip = idx + i * V;
    if (V == 8) {
      uint2 w;
      w.x = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
      w.y = (unsigned)arg[4 % V] | ((unsigned)arg[5 % V] << 8) | ((unsigned)arg[6 % V] << 16) | ((unsigned)arg[7 % V] << 24);
      *(uint2*)ip = w;
    } else {
      *(unsigned*)ip = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
    }
  }
}

template <typename T>
__global__ void maxpool3x3s2_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx, T* __restrict__ dx,
                                        long nvec, int H, int W, int Ho, int Wo, int cv) {
  EW_PRIO();
  constexpr int V = 16 / (int)sizeof(T);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv);
    long pix = i / cv;
    const int ix = (int)(pix % W); pix /= W;
    const int iy = (int)(pix % H);
    const long n = pix / H;
    // windows (oy, ox) with 2*oy - 1 <= iy <= 2*oy + 1: oy in {iy/2, (iy+1)/2} (one window for even iy); the four
    // candidates are loaded together from clamped addresses and masked afterwards
    uint4 g4[4];
    unsigned w0[4], w1[4];
    int kk[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = q >> 1, b = q & 1;
      const int oy = (iy + a) >> 1, ox = (ix + b) >> 1;
      ok[q] = oy < Ho && ox < Wo && !(a == 1 && !(iy & 1)) && !(b == 1 && !(ix & 1));
      kk[q] = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));
      const long o = (((n * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 1)) * (long)cv + c);
      g4[q] = *(const uint4*)(dy + o * V);
      const unsigned char* ip = idx + o * V;
      if (V == 8) { const uint2 w = *(const uint2*)ip; w0[q] = w.x; w1[q] = w.y; }
      else { w0[q] = *(const unsigned*)ip; w1[q] = 0; }
    }
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float g[V];
      unpack16(g4[q], g, T());
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const unsigned t = ((e < 4 ? w0[q] : w1[q]) >> (8 * (e & 3))) & 0xffu;
        acc[e] += (ok[q] && (int)t == kk[q]) ? g[e] : 0.f;
      }
    }
    *(uint4*)(dx + i * V) = pack16(acc, T());
  }
}

}  // namespace

// sweep order of norm_bwd_reduce (bit 1 set: descending rounds, the default; SIHL_EW_ORDER=0 for the A/B:
// profiles/r03_ew_order_ab.txt)
static int ew_order() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("SIHL_EW_ORDER"); v = e ? atoi(e) : 2; }
  return v;
}

#define SIHL_AFF(A, F) hipLaunchKernelGGL((affine_act_kernel<T, A, F>), dim3(g), dim3(TPB), 0, stream, (const T*)x, (T*)y, nvec, C / V, scale, shift, (const T*)res, (unsigned char*)mask)
#define SIHL_AFF_A(A) do { if (fixed) SIHL_AFF(A, true); else SIHL_AFF(A, false); } while (0)
#define SIHL_AFB(A, F) hipLaunchKernelGGL((affine_act_bwd_kernel<T, A, F>), dim3(g), dim3(TPB), 0, stream, (const T*)x, (const T*)dy, (T*)dx, nvec, C / V, scale, shift)
#define SIHL_AFB_A(A) do { if (fixed) SIHL_AFB(A, true); else SIHL_AFB(A, false); } while (0)
#define SIHL_NBR(M, A) hipLaunchKernelGGL((norm_bwd_reduce_kernel<T, M, A>), dim3(nblk), dim3(TPB), red_lds, stream, (const T*)s, (const T*)dy, rows, C, mean, rstd, gamma, beta, ws, rpb, nrl, (ew_order() >> 1) & 1)
#define SIHL_NBA(M, A, F) hipLaunchKernelGGL((norm_bwd_apply_kernel<T, M, A, F>), dim3(g), dim3(TPB), 0, stream, (const T*)s, (const T*)dy, (T*)dz, nvec, C / V, mean, rstd, gamma, beta, (const float*)s0, (const float*)s1, 1.f / (float)rows, batch_stats)
#define SIHL_NBA_F(M, A) do { if (fixed) SIHL_NBA(M, A, true); else SIHL_NBA(M, A, false); } while (0)

#define SIHL_LN(S, K, A) hipLaunchKernelGGL((layernorm_act_kernel<T, S, K, A>), dim3((int)g), dim3(TPB), 0, stream, (const T*)z, (T*)y, rows, C, gamma, beta, eps, mean, rstd)
#define SIHL_LNB(S, K, A) hipLaunchKernelGGL((layernorm_act_bwd_kernel<T, S, K, A>), dim3(nblk), dim3(TPB), (size_t)(TPB / 64) * 2 * C * sizeof(float), stream, (const T*)z, (const T*)dy, (T*)dz, rows, C, gamma, beta, mean, rstd, ws)

// lanes per row / chunks per lane by row width (cvec = 16-byte chunks per row)
#define SIHL_LN_ALL(L, A) do { if (cvec <= 16) L(16, 1, A); else if (cvec <= 32) L(32, 1, A); else if (cvec <= 64) L(64, 1, A); else L(64, 2, A); } while (0)

#define DISPATCH_DTYPE(dtype, ...)                                   \
  if (dtype == SIHL_F32) { typedef float T; __VA_ARGS__; }           \
  else if (dtype == SIHL_BF16) { typedef bf16_t T; __VA_ARGS__; }    \
  else return SIHL_EARG;

extern "C" {

int sihl_bn_finalize(const float* partials, int n_partials, int C, long count, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                     float* scale, float* shift, hipStream_t stream) {
  if (!partials || n_partials <= 0 || C <= 0 || count <= 0 || !mean || !rstd || !scale || !shift) return SIHL_EARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, partials, n_partials, C,
                     (double)count, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, hipStream_t stream) {
  if (!running_mean || !running_var || !scale || !shift || C <= 0) return SIHL_EARG;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, gamma, beta, running_mean,
                     running_var, eps, C, scale, shift);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_affine_add_act(const void* x, const void* res, void* y, void* mask, long rows, int C, const float* scale,
                        const float* shift, int act, int dtype, hipStream_t stream);

int sihl_affine_act(const void* x, void* y, long rows, int C, const float* scale, const float* shift, int act,
                    int dtype, hipStream_t stream) {
  return sihl_affine_add_act(x, nullptr, y, nullptr, rows, C, scale, shift, act, dtype, stream);
}

// y = act(x*scale[c] + shift[c] + res): BatchNorm-apply + residual add + activation in one pass (res may be NULL).
// mask (optional, with res): one byte per 16-byte vector of y, bit e = (y element e > 0) - for sihl_norm_add_relu_bwd
int sihl_affine_add_act(const void* x, const void* res, void* y, void* mask, long rows, int C, const float* scale,
                        const float* shift, int act, int dtype, hipStream_t stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (mask && !res)) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long nvec = rows * (C / V);
    bool fixed;
    const int g = grid_fixed(nvec, C / V, &fixed);
    switch (act) {
      case SIHL_ACT_RELU: SIHL_AFF_A(SIHL_ACT_RELU); break;
      case SIHL_ACT_SILU: SIHL_AFF_A(SIHL_ACT_SILU); break;
      case SIHL_ACT_SIGMOID: SIHL_AFF_A(SIHL_ACT_SIGMOID); break;
      default: SIHL_AFF_A(SIHL_ACT_NONE); break;
    }
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_affine_act_bwd(const void* x, const void* dy, void* dx, long rows, int C, const float* scale,
                        const float* shift, int act, int dtype, hipStream_t stream) {
  if (!x || !dy || !dx || rows <= 0 || C <= 0) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long nvec = rows * (C / V);
    bool fixed;
    const int g = grid_fixed(nvec, C / V, &fixed);
    switch (act) {
      case SIHL_ACT_RELU: SIHL_AFB_A(SIHL_ACT_RELU); break;
      case SIHL_ACT_SILU: SIHL_AFB_A(SIHL_ACT_SILU); break;
      case SIHL_ACT_SIGMOID: SIHL_AFB_A(SIHL_ACT_SIGMOID); break;
      default: SIHL_AFB_A(SIHL_ACT_NONE); break;
    }
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// out[N][H][W][C] = w0*bilinear_x2(a[N][H/2][W/2][C]) + w1*b ; w = softmax(wraw[0:2]); b==null -> plain upsample
int sihl_fuse_up2(const void* a, const void* b, const float* wraw, void* out, int N, int H, int W, int C, int dtype,
                  hipStream_t stream) {
  if (!a || !out || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (b && !wraw)) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long rows = (long)N * H;  // grid.(y, z) = output rows
    if (rows > 65535L * 65535L || (long)W * C >= (1L << 31)) return SIHL_EARG;
    hipLaunchKernelGGL(fuse_up2_kernel<T>, dim3((W * (C / V) + TPB - 1) / TPB, (unsigned)(rows < 65535 ? rows : 65535), (unsigned)((rows + 65534) / 65535)), dim3(TPB), 0, stream,
                       (const T*)a, (const T*)b, wraw, (T*)out, N, H, W, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// gacc: SIHL_FUSION_GACC_FLOATS floats of scratch (per-workgroup partial sums); da/db may be null when not needed; dw_raw
// (2 floats) written if non-null
int sihl_fuse_up2_bwd(const void* dout, const void* a, const void* b, const float* wraw, void* da, void* db,
                      float* dw_raw, float* gacc, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!dout || N <= 0 || (H & 1) || (W & 1) || (dw_raw && !gacc)) return SIHL_EARG;
  if ((db || dw_raw) && (!a || !b || !wraw)) return SIHL_EARG;  // b == null: plain upsample, only da
  int nblk = 0;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    nblk = grid_for((long)N * H * W * (C / V));
    if (db || dw_raw)
      hipLaunchKernelGGL(fuse_up2_bwd_hi_kernel<T>, dim3(nblk), dim3(TPB), 0, stream,
                         (const T*)dout, (const T*)a, (const T*)b, wraw, (T*)db, dw_raw ? gacc : nullptr, N, H, W, C);
    if (da)
      hipLaunchKernelGGL(up2_adjoint_kernel<T>, dim3(grid_for((long)N * (H / 2) * (W / 2) * (C / V))), dim3(TPB), 0,
                         stream, (const T*)dout, b ? wraw : nullptr, (T*)da, N, H, W, C);
  });
  if (dw_raw) hipLaunchKernelGGL(fusion_wgrad_kernel, dim3(1), dim3(256), 0, stream, wraw, gacc, nblk, dw_raw, 2);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}


// out[N][H][W][C] = nearest_x2(lo[N][H/2][W/2][C]) + skip
int sihl_nearest_up2_add(const void* lo, const void* skip, void* out, int N, int H, int W, int C, int dtype,
                         hipStream_t stream) {
  if (!lo || !skip || !out || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(nearest_up2_add_kernel<T>, dim3(grid_for((long)N * H * W * (C / V))), dim3(TPB), 0, stream,
                       (const T*)lo, (const T*)skip, (T*)out, N, H, W, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// dlo[N][H/2][W/2][C] = 2x2 block sums of dout[N][H][W][C]   (the skip gradient is dout itself)
int sihl_nearest_up2_add_bwd(const void* dout, void* dlo, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!dout || !dlo || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(nearest_up2_adjoint_kernel<T>, dim3(grid_for((long)N * (H / 2) * (W / 2) * (C / V))), dim3(TPB),
                       0, stream, (const T*)dout, (T*)dlo, N, H, W, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// out[N][Ho][Wo][C] = bilinear_resize(a[N][H][W][C]) (+ add if non-null), align_corners=False
int sihl_resize_bilinear(const void* a, const void* add, void* out, int N, int H, int W, int Ho, int Wo, int C,
                         int dtype, hipStream_t stream) {
  if (!a || !out || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(resize_bilinear_kernel<T>, dim3(grid_for((long)N * Ho * Wo * (C / V))), dim3(TPB), 0, stream,
                       (const T*)a, (const T*)add, (T*)out, N, H, W, Ho, Wo, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_resize_bilinear_bwd(const void* dout, void* da, int N, int H, int W, int Ho, int Wo, int C, int dtype,
                             hipStream_t stream) {
  if (!dout || !da || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    hipLaunchKernelGGL(resize_bilinear_adjoint_kernel<T>, dim3(grid_for((long)N * H * W * (C / V))), dim3(TPB), 0,
                       stream, (const T*)dout, (T*)da, N, H, W, Ho, Wo, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// 3x3 / stride 2 / pad 1 max pooling over NHWC x [N][H][W][C] -> y [N][Ho][Wo][C], Ho = (H - 1) / 2 + 1; idx (one byte
// per output element) receives the winning tap for sihl_maxpool3x3s2_bwd.  C % (16 / sizeof(T)) == 0.
int sihl_maxpool3x3s2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SIHL_EARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long nvec = (long)N * Ho * Wo * (C / V);
    hipLaunchKernelGGL((maxpool3x3s2_fwd_kernel<T>), dim3(grid_for(nvec)), dim3(TPB), 0, stream, (const T*)x, (T*)y,
                       (unsigned char*)idx, nvec, H, W, Ho, Wo, C / V);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// dx [N][H][W][C] of sihl_maxpool3x3s2_fwd from dy [N][Ho][Wo][C] and the forward's idx (every element written).
int sihl_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int N, int H, int W, int C, int dtype,
                          hipStream_t stream) {
  if (!dy || !idx || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SIHL_EARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long nvec = (long)N * H * W * (C / V);
    hipLaunchKernelGGL((maxpool3x3s2_bwd_kernel<T>), dim3(grid_for(nvec)), dim3(TPB), 0, stream, (const T*)dy,
                       (const unsigned char*)idx, (T*)dx, nvec, H, W, Ho, Wo, C / V);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// out = act(a + b) over `numel` elements (act: none or relu) - the residual merge of a ResNet block.
// Its backward is sihl_affine_act_bwd(out, dout, ..., relu): both inputs receive dout * (out > 0).
int sihl_add_act(const void* a, const void* b, void* out, long numel, int act, int dtype, hipStream_t stream) {
  if (!a || !b || !out || numel <= 0 || (act != SIHL_ACT_NONE && act != SIHL_ACT_RELU)) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (numel % V) return SIHL_EARG;
    if (act == SIHL_ACT_RELU)
      hipLaunchKernelGGL((add_act_kernel<T, SIHL_ACT_RELU>), dim3(grid_for(numel / V)), dim3(TPB), 0, stream, (const T*)a, (const T*)b, (T*)out, numel / V);
    else
      hipLaunchKernelGGL((add_act_kernel<T, SIHL_ACT_NONE>), dim3(grid_for(numel / V)), dim3(TPB), 0, stream, (const T*)a, (const T*)b, (T*)out, numel / V);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// out = sum_i softmax(wraw)_i * x_i over n (2 or 3) same-shaped tensors of `numel` elements
int sihl_fuse_sum(const void* x0, const void* x1, const void* x2, const float* wraw, void* out, long numel, int n,
                  int dtype, hipStream_t stream) {
  if (!x0 || !x1 || (n == 3 && !x2) || !wraw || !out || numel <= 0 || n < 2 || n > 3) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (numel % V) return SIHL_EARG;
    hipLaunchKernelGGL(fuse_sum_kernel<T>, dim3(grid_for(numel / V)), dim3(TPB), 0, stream, (const T*)x0, (const T*)x1,
                       (const T*)x2, wraw, (T*)out, numel / V, n);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_fuse_sum_bwd(const void* dout, const void* x0, const void* x1, const void* x2, const float* wraw, void* d0,
                      void* d1, void* d2, float* dw_raw, float* gacc, long numel, int n, int dtype,
                      hipStream_t stream) {
  if (!dout || !wraw || numel <= 0 || n < 2 || n > 3 || (dw_raw && (!gacc || !x0 || !x1 || (n == 3 && !x2))))
    return SIHL_EARG;
  int nblk = 0;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (numel % V) return SIHL_EARG;
    nblk = grid_for(numel / V);
    hipLaunchKernelGGL(fuse_sum_bwd_kernel<T>, dim3(nblk), dim3(TPB), 0, stream, (const T*)dout,
                       (const T*)x0, (const T*)x1, (const T*)x2, wraw, (T*)d0, (T*)d1, (T*)d2,
                       dw_raw ? gacc : nullptr, numel / V, n);
  });
  if (dw_raw) hipLaunchKernelGGL(fusion_wgrad_kernel, dim3(1), dim3(256), 0, stream, wraw, gacc, nblk, dw_raw, n);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// ------------------------------------------------------------------ gradient-norm clipping over many tensors
// torch.nn.utils.clip_grad_norm_(params, max_norm) (the reference's Lightning gradient_clip_val) on up to 320 fp32
// tensors per launch: total = sqrt(sum of all squares); coef = min(1, max_norm / (total + 1e-6)); every gradient *= coef.
// Three launches (partial sums of squares per 64 Ki-element chunk, ordered finish, scale) instead of the ~14 of the
// multi-tensor library routines and, above all, without their 320 result tensors (2.8 ms of host time per step).
}  // extern "C" (reopened below)
namespace {
constexpr int CLIP_CHUNK = 1 << 16;
template <int MAXT> struct ClipPtrs { float* p[MAXT]; };
template <int MAXT>
__global__ void clip_sumsq_kernel(const ClipPtrs<MAXT> ptrs, const int* __restrict__ map, const long* __restrict__ numel, float* __restrict__ part) {
  EW_PRIO();
  const int t = map[2 * blockIdx.x], c = map[2 * blockIdx.x + 1];
  const float* __restrict__ x = ptrs.p[t] + (long)c * CLIP_CHUNK;
  const long left = numel[t] - (long)c * CLIP_CHUNK;
  const int len = left < CLIP_CHUNK ? (int)left : CLIP_CHUNK;
  float a = 0.f;
  const bool vec = (((unsigned long)x) & 15) == 0;
  if (vec) {
    const int n4 = len >> 2;
    for (int i = threadIdx.x; i < n4; i += TPB) {
      const float4 v = ((const float4*)x)[i];
      a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < len; i += TPB) a += x[i] * x[i];
  } else {
    for (int i = threadIdx.x; i < len; i += TPB) a += x[i] * x[i];
  }
  __shared__ float red[TPB / 64];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < TPB / 64; ++w) s += red[w];
    part[blockIdx.x] = s;
  }
}
__global__ void clip_finish_kernel(const float* __restrict__ part, int nblocks, float max_norm, float* __restrict__ out) {
  EW_PRIO();
  __shared__ double red[256];
  double a = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) a += (double)part[b];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]);
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
    out[0] = coef;
    out[1] = total;
  }
}
template <int MAXT>
__global__ void clip_scale_kernel(const ClipPtrs<MAXT> ptrs, const int* __restrict__ map, const long* __restrict__ numel, const float* __restrict__ out) {
  EW_PRIO();
  const float coef = out[0];
  if (coef >= 1.f) return;  // (NaN fails the test and scales, like the reference)
  const int t = map[2 * blockIdx.x], c = map[2 * blockIdx.x + 1];
  float* __restrict__ x = ptrs.p[t] + (long)c * CLIP_CHUNK;
  const long left = numel[t] - (long)c * CLIP_CHUNK;
  const int len = left < CLIP_CHUNK ? (int)left : CLIP_CHUNK;
  if ((((unsigned long)x) & 15) == 0) {
    const int n4 = len >> 2;
    for (int i = threadIdx.x; i < n4; i += TPB) {
      float4 v = ((float4*)x)[i];
      v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
      ((float4*)x)[i] = v;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < len; i += TPB) x[i] *= coef;
  } else {
    for (int i = threadIdx.x; i < len; i += TPB) x[i] *= coef;
  }
}

template <int MAXT>
int grad_clip_launch(const void* const* grads, int n, const int* map, const int* group_blocks, const long* numel, float max_norm,
                     float* scratch, long scratch_floats, hipStream_t stream) {
  const int groups = (n + MAXT - 1) / MAXT;
  long nblocks = 0;
  for (int g = 0; g < groups; ++g) {
    if (group_blocks[g] <= 0) return SIHL_EARG;
    nblocks += group_blocks[g];
  }
  if (scratch_floats < nblocks + 2 || nblocks > (1 << 24)) return SIHL_EARG;
  auto group_ptrs = [&](int g) {
    ClipPtrs<MAXT> ptrs;
    const int first = g * MAXT, m = n - first < MAXT ? n - first : MAXT;
    for (int k = 0; k < MAXT; ++k) ptrs.p[k] = (float*)grads[first + (k < m ? k : 0)];
    return ptrs;
  };
  long b0 = 0;
  for (int g = 0; g < groups; b0 += group_blocks[g], ++g)
    hipLaunchKernelGGL(clip_sumsq_kernel<MAXT>, dim3(group_blocks[g]), dim3(TPB), 0, stream, group_ptrs(g), map + 2 * b0,
                       numel + (long)g * MAXT, scratch + b0);
  hipLaunchKernelGGL(clip_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)scratch, (int)nblocks, max_norm, scratch + nblocks);
  b0 = 0;
  for (int g = 0; g < groups; b0 += group_blocks[g], ++g)
    hipLaunchKernelGGL(clip_scale_kernel<MAXT>, dim3(group_blocks[g]), dim3(TPB), 0, stream, group_ptrs(g), map + 2 * b0,
                       numel + (long)g * MAXT, (const float*)(scratch + nblocks));
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}
}  // namespace
extern "C" {

// grads: HOST array of n device pointers to dense fp32 tensors, taken in groups of `group` (32 or 320: one launch's by-value
// pointer table); group_blocks: HOST int [ceil(n / group)], the workgroups of each group; map: DEVICE int32
// [sum of group_blocks][2] = (tensor index WITHIN its group, 64 Ki-element chunk) of every workgroup, group after group;
// numel: DEVICE int64 [n] - map and numel depend on the sizes only (the caller builds them once per model); scratch: DEVICE
// floats, nblocks + 2: the partial sums, then (coefficient, total norm).
int sihl_grad_clip(const void* const* grads, int n, const int* map, const int* group_blocks, const long* numel, float max_norm,
                   float* scratch, long scratch_floats, int group, hipStream_t stream) {
  if (!grads || n <= 0 || !map || !group_blocks || !numel || !scratch || !(max_norm > 0.f)) return SIHL_EARG;
  for (int k = 0; k < n; ++k)
    if (!grads[k] || (((unsigned long)grads[k]) & 3)) return SIHL_EARG;
  if (group == 320) return grad_clip_launch<320>(grads, n, map, group_blocks, numel, max_norm, scratch, scratch_floats, stream);
  if (group == 32) return grad_clip_launch<32>(grads, n, map, group_blocks, numel, max_norm, scratch, scratch_floats, stream);
  return SIHL_EARG;
}

// out[N][Ho][Wo][C] = w0*blurpool_s2(a[N][H][W][C]) + w1*b + w2*c ; b==c==null -> plain blur pool
int sihl_blur_fuse(const void* a, const void* b, const void* c, const float* wraw, const float* a_scale,
                   const float* a_shift, void* out, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!a || !out || N <= 0 || H < 2 || W < 2 || ((b != nullptr) != (c != nullptr)) || (b && !wraw)) return SIHL_EARG;
  if ((a_scale != nullptr) != (a_shift != nullptr)) return SIHL_EARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const long rows = (long)N * Ho;  // grid.(y, z) = output rows
    if (rows > 65535L * 65535L || (long)W * C >= (1L << 31)) return SIHL_EARG;
    hipLaunchKernelGGL(blur_fuse_kernel<T>, dim3((Wo * (C / V) + TPB - 1) / TPB, (unsigned)(rows < 65535 ? rows : 65535), (unsigned)((rows + 65534) / 65535)), dim3(TPB), 0, stream,
                       (const T*)a, (const T*)b, (const T*)c, wraw, a_scale, a_shift, (T*)out, N, H, W, Ho, Wo, C);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// fused (b != null): gacc = SIHL_FUSION_GACC_FLOATS floats of scratch; plain blur (b == null): only da is produced
// a_scale / a_shift: as in sihl_blur_fuse; `da` is then the gradient of a * scale + shift (what the producing conv
// block's BatchNorm backward takes), the fusion-weight gradient uses the blurred affine value
int sihl_blur_fuse_bwd(const void* dout, const void* a, const void* b, const void* c, const float* wraw,
                       const float* a_scale, const float* a_shift, void* da, void* db, void* dc, float* dw_raw,
                       float* gacc, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!dout || N <= 0 || H < 2 || W < 2 || (dw_raw && (!gacc || !b))) return SIHL_EARG;
  if ((a_scale != nullptr) != (a_shift != nullptr)) return SIHL_EARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  int nblk = 0;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    nblk = grid_for((long)N * Ho * Wo * (C / V));
    if (b && (db || dc || dw_raw))
      hipLaunchKernelGGL(blur_fuse_bwd_lo_kernel<T>, dim3(nblk), dim3(TPB), 0,
                         stream, (const T*)dout, (const T*)a, (const T*)b, (const T*)c, wraw, a_scale, a_shift, (T*)db,
                         (T*)dc, dw_raw ? gacc : nullptr, N, H, W, Ho, Wo, C);
    if (da)
      hipLaunchKernelGGL(blur_adjoint_kernel<T>, dim3(grid_for((long)N * H * W * (C / V))), dim3(TPB), 0, stream,
                         (const T*)dout, b ? wraw : nullptr, (T*)da, N, H, W, Ho, Wo, C);
  });
  if (dw_raw) hipLaunchKernelGGL(fusion_wgrad_kernel, dim3(1), dim3(256), 0, stream, wraw, gacc, nblk, dw_raw, 3);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// Backward through BatchNorm (+ activation).  mode 0: s = act(conv) (post-activation, pre-norm), y = BN(s);
// mode 1: s = conv (pre-norm), y = act(BN(s)).  Writes dgamma/dbeta (fp32 [C]) and dz (grad wrt conv output).
// workgroups of a column reduction: every thread folds >= 8 rows (nrl row lanes per block), at most 1024 partial rows
static int reduce_blocks(long rows, int nrl = 1) {
  long nb = (rows + 8L * nrl - 1) / (8L * nrl);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}
static int row_lanes(int threads_per_row) { return threads_per_row >= TPB ? 1 : TPB / threads_per_row; }

// Workspace bytes for sihl_norm_act_bwd / sihl_colsum.
long sihl_norm_act_bwd_ws_bytes(long rows, int C, int dtype) {
  const int cvec = C / (dtype == SIHL_BF16 ? 8 : 4);
  (void)cvec;
  return ((long)reduce_blocks(rows) * 2 * C + 2L * C) * (long)sizeof(float);  // one partial row per workgroup + sums
}
long sihl_colsum_ws_bytes(long rows, int C) { return (long)reduce_blocks(rows) * C * (long)sizeof(float); }

// Per-channel (sum, sumsq) partial rows [n][2][C] of an NHWC tensor some OTHER kernel produced (the MIOpen stem conv of
// the ResNet trunk), in the layout sihl_bn_finalize reads: the batch statistics of BatchNorm without the conv epilogue.
// It is the BatchNorm-backward column reduction with s = dy = x, mean 0 and rstd 1: sum g = sum x, sum g * xhat = sum x^2.
int sihl_bn_stats_rows(long rows, int C, int dtype) {
  if (rows <= 0 || C <= 0) return 0;
  return reduce_blocks(rows, row_lanes(C / (dtype == SIHL_BF16 ? 8 : 4)));
}
int sihl_bn_stats(const void* x, long rows, int C, float* partials, int n_partials, int dtype, hipStream_t stream) {
  if (!x || rows <= 0 || C <= 0 || !partials) return SIHL_EARG;
  const int nrl = row_lanes(C / (dtype == SIHL_BF16 ? 8 : 4));
  const int nblk = reduce_blocks(rows, nrl);
  if (n_partials != nblk) return SIHL_EARG;
  const int rpb = (int)((rows + nblk - 1) / nblk);
  const void* s = x;
  const void* dy = x;
  const float* mean = nullptr;
  const float* rstd = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* ws = partials;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const size_t red_lds = (size_t)nrl * 2 * C * sizeof(float);
    SIHL_NBR(0, SIHL_ACT_NONE);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// Backward of a residual block's tail  y = relu(BN(s) + identity)  (conv -> BatchNorm, no activation of its own, mode 1):
// dres = dy * (y > 0) (the gradient of the identity branch AND of BN's output), dz = BN backward of dres.  The mask is
// applied inside the column reduction (norm_bwd_reduce_kernel<.., MASK>), not in a pass of its own.
int sihl_norm_add_relu_bwd(const void* s, const void* dy, const void* y, const void* mask, void* dres, void* dz, long rows,
                           int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float* dgamma, float* dbeta, int batch_stats, int dtype, float* ws, long ws_bytes,
                           hipStream_t stream) {
  if (!s || !dy || (!y && !mask) || !dres || !dz || rows <= 0 || C <= 0 || !mean || !rstd || !ws) return SIHL_EARG;
  const int nrl = row_lanes(C / (dtype == SIHL_BF16 ? 8 : 4));
  const int nblk = reduce_blocks(rows, nrl);
  if (ws_bytes < sihl_norm_act_bwd_ws_bytes(rows, C, dtype)) return SIHL_EWS;
  const int rpb = (int)((rows + nblk - 1) / nblk);
  float* sums = ws + (long)reduce_blocks(rows) * 2 * C;
  float* s0 = dbeta ? dbeta : sums;
  float* s1 = dgamma ? dgamma : sums + C;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const size_t red_lds = (size_t)nrl * 2 * C * sizeof(float);
    if (mask)
      hipLaunchKernelGGL((norm_bwd_reduce_kernel<T, 1, SIHL_ACT_NONE, 2>), dim3(nblk), dim3(TPB), red_lds, stream,
                         (const T*)s, (const T*)dy, rows, C, mean, rstd, gamma, beta, ws, rpb, nrl, (ew_order() >> 1) & 1,
                         mask, (T*)dres);
    else
      hipLaunchKernelGGL((norm_bwd_reduce_kernel<T, 1, SIHL_ACT_NONE, 1>), dim3(nblk), dim3(TPB), red_lds, stream,
                         (const T*)s, (const T*)dy, rows, C, mean, rstd, gamma, beta, ws, rpb, nrl, (ew_order() >> 1) & 1,
                         y, (T*)dres);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, (const float*)ws, nblk, 2, C, s0,
                       s1);
    const long nvec = rows * (C / V);
    bool fixed;
    const int g = grid_fixed(nvec, C / V, &fixed);
    const void* dy_apply = dres;  // the apply pass reads the masked gradient
    {
      const void* dy = dy_apply;
      SIHL_NBA_F(1, SIHL_ACT_NONE);
    }
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_norm_act_bwd(const void* s, const void* dy, void* dz, long rows, int C, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float* dgamma, float* dbeta, int mode, int act,
                      int batch_stats, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
  if (!s || !dy || !dz || rows <= 0 || C <= 0 || !mean || !rstd || !ws) return SIHL_EARG;
  const int nrl = row_lanes(C / (dtype == SIHL_BF16 ? 8 : 4));
  const int nblk = reduce_blocks(rows, nrl);
  if (ws_bytes < sihl_norm_act_bwd_ws_bytes(rows, C, dtype)) return SIHL_EWS;
  const int rpb = (int)((rows + nblk - 1) / nblk);
  float* sums = ws + (long)reduce_blocks(rows) * 2 * C;
  // the column sums ARE the parameter gradients: finalize straight into the caller's buffers when given
  float* s0 = dbeta ? dbeta : sums;
  float* s1 = dgamma ? dgamma : sums + C;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V) return SIHL_EARG;
    const size_t red_lds = (size_t)nrl * 2 * C * sizeof(float);
    if (mode == 0) SIHL_NBR(0, SIHL_ACT_NONE);
    else if (act == SIHL_ACT_RELU) SIHL_NBR(1, SIHL_ACT_RELU);
    else if (act == SIHL_ACT_SILU) SIHL_NBR(1, SIHL_ACT_SILU);
    else if (act == SIHL_ACT_SIGMOID) SIHL_NBR(1, SIHL_ACT_SIGMOID);
    else SIHL_NBR(1, SIHL_ACT_NONE);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, (const float*)ws,
                       nblk, 2, C, s0, s1);
    const long nvec = rows * (C / V);
    bool fixed;
    const int g = grid_fixed(nvec, C / V, &fixed);
    if (mode == 0) {
      if (act == SIHL_ACT_RELU) SIHL_NBA_F(0, SIHL_ACT_RELU);
      else if (act == SIHL_ACT_NONE) SIHL_NBA_F(0, SIHL_ACT_NONE);
      else return SIHL_EARG;  // act->norm backward needs the pre-activation for silu/sigmoid: not a hot-path block
    } else {
      switch (act) {
        case SIHL_ACT_RELU: SIHL_NBA_F(1, SIHL_ACT_RELU); break;
        case SIHL_ACT_SILU: SIHL_NBA_F(1, SIHL_ACT_SILU); break;
        case SIHL_ACT_SIGMOID: SIHL_NBA_F(1, SIHL_ACT_SIGMOID); break;
        default: SIHL_NBA_F(1, SIHL_ACT_NONE); break;
      }
    }
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_layernorm_act(const void* z, void* y, long rows, int C, const float* gamma, const float* beta, float eps,
                       int act, float* mean, float* rstd, int dtype, hipStream_t stream) {
  if (!z || !y || rows <= 0 || C <= 0 || !gamma || !beta) return SIHL_EARG;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V || C / V > 128) return SIHL_EARG;
    long g = (rows + 3) / 4;
    if (g > 256 * 8) g = 256 * 8;
    const int cvec = C / V;
    if (act != SIHL_ACT_SILU && act != SIHL_ACT_NONE) return SIHL_EARG;
    if (act == SIHL_ACT_SILU) SIHL_LN_ALL(SIHL_LN, SIHL_ACT_SILU); else SIHL_LN_ALL(SIHL_LN, SIHL_ACT_NONE);
  });
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// workgroups (= partial rows) of the LayerNorm backward: 5 per CU matches the kernel's 5 waves/SIMD occupancy
int sihl_layernorm_bwd_waves(long rows) {
  long g = (rows + 3) / 4;
  if (g > 256 * 5) g = 256 * 5;
  if (g < 1) g = 1;
  return (int)g;
}

long sihl_layernorm_act_bwd_ws_bytes(long rows, int C) {
  return ((long)sihl_layernorm_bwd_waves(rows) * 2 * C + 2L * C) * (long)sizeof(float);
}

// ws: sihl_layernorm_act_bwd_ws_bytes(rows, C)
int sihl_layernorm_act_bwd(const void* z, const void* dy, void* dz, long rows, int C, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, int act, float* dgamma,
                           float* dbeta, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
  if (!z || !dy || !dz || rows <= 0 || !gamma || !beta || !mean || !rstd || !ws) return SIHL_EARG;
  const int nblk = sihl_layernorm_bwd_waves(rows);
  if (ws_bytes < (long)(nblk * 2L * C + 2L * C) * (long)sizeof(float)) return SIHL_EWS;
  float* sums = ws + (long)nblk * 2 * C;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V || C / V > 128) return SIHL_EARG;
    const int cvec = C / V;
    if (act != SIHL_ACT_SILU && act != SIHL_ACT_NONE) return SIHL_EARG;
    if (act == SIHL_ACT_SILU) SIHL_LN_ALL(SIHL_LNB, SIHL_ACT_SILU); else SIHL_LN_ALL(SIHL_LNB, SIHL_ACT_NONE);
  });
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, (const float*)ws, nblk,
                     2, C, dbeta ? dbeta : sums, dgamma ? dgamma : sums + C);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// out[c] = sum_r x[r][c]   (bias gradients).  ws: sihl_colsum_ws_bytes(rows, C)
int sihl_colsum(const void* x, long rows, int C, float* out, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
  if (!x || !out || rows <= 0 || C <= 0 || !ws) return SIHL_EARG;
  if (ws_bytes < sihl_colsum_ws_bytes(rows, C)) return SIHL_EWS;
  int nrl = row_lanes(C), nblk = 1, rpb = 1;
  DISPATCH_DTYPE(dtype, {
    constexpr int V = 16 / sizeof(T);
    if (C % V == 0) nrl = row_lanes(C / V);
    nblk = reduce_blocks(rows, nrl);
    rpb = (int)((rows + nblk - 1) / nblk);
    if (C % V == 0) {
      hipLaunchKernelGGL((colsum_partial_kernel<T, true>), dim3(nblk), dim3(TPB), 0, stream, (const T*)x, rows, C, ws, rpb, nrl);
    } else {
      hipLaunchKernelGGL((colsum_partial_kernel<T, false>), dim3(nblk), dim3(TPB), 0, stream, (const T*)x, rows, C, ws, rpb, nrl);
    }
  });
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, (const float*)ws,
                     nblk, 1, C, out, out);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// w [Cout][KH][KW][Cin] (dtype_in) -> o [Cin][KH][KW][Cout] (dtype_out), spatially flipped when flip != 0
int sihl_weight_flip_transpose(const void* w, void* o, int Cout, int KH, int KW, int Cin, int flip, int dtype_in,
                               int dtype_out, hipStream_t stream) {
  if (!w || !o || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return SIHL_EARG;
  const int g = grid_for((long)Cout * KH * KW * Cin);
  if (dtype_in == SIHL_F32 && dtype_out == SIHL_F32)
    hipLaunchKernelGGL((weight_flip_transpose_kernel<float, float>), dim3(g), dim3(TPB), 0, stream, (const float*)w, (float*)o, Cout, KH, KW, Cin, flip);
  else if (dtype_in == SIHL_F32 && dtype_out == SIHL_BF16)
    hipLaunchKernelGGL((weight_flip_transpose_kernel<float, bf16_t>), dim3(g), dim3(TPB), 0, stream, (const float*)w, (bf16_t*)o, Cout, KH, KW, Cin, flip);
  else if (dtype_in == SIHL_BF16 && dtype_out == SIHL_BF16)
    hipLaunchKernelGGL((weight_flip_transpose_kernel<bf16_t, bf16_t>), dim3(g), dim3(TPB), 0, stream, (const bf16_t*)w, (bf16_t*)o, Cout, KH, KW, Cin, flip);
  else return SIHL_EARG;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// descs: DEVICE array of n sihl weight descriptors (layout of struct WeightDesc above, 96 bytes each; built once by
// the host mirror), total_blocks = sum over weights of ceil(Op*KH*KW*I / 2048), total_tblocks = sum of
// KH*KW*ceil(Op/64)*ceil(I/64).
int sihl_weight_prepare(const void* descs, int n, long total_blocks, long total_tblocks, hipStream_t stream) {
  if (!descs || n <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffL || total_tblocks <= 0 ||
      total_tblocks > 0x7fffffffL)
    return SIHL_EARG;
  static_assert(sizeof(WeightDesc) == 96, "descriptor layout is part of the C-ABI");
  hipLaunchKernelGGL(weight_prepare_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream,
                     (const WeightDesc*)descs, n);
  hipLaunchKernelGGL(weight_prepare_t_kernel, dim3((unsigned)total_tblocks), dim3(256), 0, stream,
                     (const WeightDesc*)descs, n);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
