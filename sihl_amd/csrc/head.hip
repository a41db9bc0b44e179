// Detection-head decode kernels (gfx950): per-image top-K over all pyramid positions, row gather,
// and the anchor-free box decode of ObjectDetection.forward
// (src/sihl/heads/object_detection.py:99-122, anchors :83-97) with closed-form anchors - the
// (P,4) offsets/scales tensors of the reference are never materialised.
#include "common.h"

namespace {

// ---- top-K: one workgroup per image sorts (value, index) pairs in LDS with a bitonic network.
// Order: value descending, ties by ascending index (torch.topk leaves tie order unspecified).
__device__ __forceinline__ bool before(float va, int ia, float vb, int ib) {
  return va > vb || (va == vb && ia < ib);
}

template <typename T>
__global__ __launch_bounds__(1024) void topk_rows_kernel(const T* __restrict__ x, int P, int P2, int K, int estride,
                                                         float* __restrict__ vals, int* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* v = (float*)smem;
  int* id = (int*)(smem + (size_t)P2 * 4);
  const T* row = x + (long)blockIdx.x * P * estride;
  for (int i = threadIdx.x; i < P2; i += blockDim.x) {
    v[i] = i < P ? elem<T>::ld(row + (long)i * estride) : -INFINITY;
    id[i] = i < P ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < P2 / 2; t += blockDim.x) {
        const int lo = 2 * t - (t & (j - 1));  // index with bit j cleared
        const int hi = lo + j;
        const bool up = (lo & k) == 0;  // this sub-sequence sorted "best first"
        const float a = v[lo], b = v[hi];
        const int ia = id[lo], ib = id[hi];
        const bool swap = up ? before(b, ib, a, ia) : before(a, ia, b, ib);
        if (swap) { v[lo] = b; v[hi] = a; id[lo] = ib; id[hi] = ia; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    vals[(long)blockIdx.x * K + i] = v[i];
    idx[(long)blockIdx.x * K + i] = id[i];
  }
}

// ---- top-K by radix SELECT (the default): the bitonic network above sorts all P2 = 8192 slots of a 5456-position row
// through 91 barrier-separated stages (112 us at bs 32) to deliver 100 of them.  Here every position gets a unique
// composite key - (monotone image of the value) : (0xFFFF - index), so "larger key" is exactly the order of `before` -
// the K-th largest key is found digit by digit (8 bits per pass: LDS histogram of the positions that still match the
// prefix, suffix scan by one wave), the K positions at or above it are compacted and ordered by rank counting.  bf16 rows
// have 16-bit value keys: 4 passes; fp32: 6.  One workgroup of 1024 threads per image, 3 barriers per pass.
template <typename T> struct TopkKey;
template <> struct TopkKey<float> {
  static constexpr int BITS = 48;
  __device__ static __forceinline__ unsigned long long make(float v, int i) {
    const unsigned u = __float_as_uint(v);
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)k << 16) | (unsigned)(0xFFFF - i);
  }
};
template <> struct TopkKey<bf16_t> {
  static constexpr int BITS = 32;
  __device__ static __forceinline__ unsigned long long make(float v, int i) {
    const unsigned u = __float_as_uint(v);  // low 16 bits are zero: a bf16 value
    const unsigned k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (unsigned long long)(((k >> 16) << 16) | (unsigned)(0xFFFF - i));
  }
};

template <typename T>
__global__ __launch_bounds__(1024) void topk_select_kernel(const T* __restrict__ x, int P, int K, int estride,
                                                           float* __restrict__ vals, int* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BITS = TopkKey<T>::BITS;
  float* v = (float*)smem;                                  // [P] values
  unsigned long long* cand = (unsigned long long*)(v + ((P + 1) & ~1));  // [K] keys of the selected positions
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_krem, s_ncand;
  const int tid = threadIdx.x, lane = tid & 63;
  const T* row = x + (long)blockIdx.x * P * estride;
  for (int i = tid; i < P; i += 1024) v[i] = elem<T>::ld(row + (long)i * estride);
  if (tid == 0) { s_prefix = 0ULL; s_krem = K; s_ncand = 0; }
  __syncthreads();
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    for (int i0 = 0; i0 < P; i0 += 1024) {
      const int i = i0 + tid;
      const unsigned long long key = i < P ? TopkKey<T>::make(v[i], i) : 0ULL;
      const bool live = i < P && (shift + 8 >= BITS || (key >> (shift + 8)) == prefix);
      const unsigned d = (unsigned)(key >> shift) & 255u;
      // a wave whose live lanes all carry one digit (the sign / exponent byte of a row of similar logits) adds once
      const unsigned long long m = __ballot(live);
      if (m) {
        const unsigned d0 = __builtin_amdgcn_readlane(d, __ffsll((long long)m) - 1);
        if (__ballot(live && d != d0) == 0ULL) {
          if (lane == 0) atomicAdd(&hist[d0], (unsigned)__popcll(m));
        } else if (live) {
          atomicAdd(&hist[d], 1u);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {  // bins 4*lane .. 4*lane + 3; suffix sums from the top bin down
      const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const unsigned own = h0 + h1 + h2 + h3;
      unsigned above = own;  // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(above, o);
        if (lane + o < 64) above += t;
      }
      above -= own;  // positions in bins above this lane's four
      const unsigned krem = (unsigned)s_krem;
      // the digit of the K-th key: the highest bin b with (count above b) < krem <= (count above b) + hist[b]
      unsigned a = above;
      int found = -1;
      unsigned found_above = 0;
      if (a < krem && krem <= a + h3) { found = 3; found_above = a; }
      a += h3;
      if (found < 0 && a < krem && krem <= a + h2) { found = 2; found_above = a; }
      a += h2;
      if (found < 0 && a < krem && krem <= a + h1) { found = 1; found_above = a; }
      a += h1;
      if (found < 0 && a < krem && krem <= a + h0) { found = 0; found_above = a; }
      if (found >= 0) {
        s_prefix = (prefix << 8) | (unsigned long long)(4 * lane + found);
        s_krem = (int)(krem - found_above);
      }
    }
    __syncthreads();
  }
  // s_prefix is the K-th largest key (keys are unique): exactly K positions are at or above it
  const unsigned long long kth = s_prefix;
  for (int i = tid; i < P; i += 1024) {
    const unsigned long long key = TopkKey<T>::make(v[i], i);
    if (key >= kth) cand[atomicAdd(&s_ncand, 1)] = key;
  }
  __syncthreads();
  for (int t = tid; t < K; t += 1024) {
    const unsigned long long mine = cand[t];
    int rank = 0;
    for (int j = 0; j < K; ++j) rank += cand[j] > mine ? 1 : 0;
    const int i = 0xFFFF - (int)(mine & 0xFFFFULL);
    vals[(long)blockIdx.x * K + rank] = v[i];
    idx[(long)blockIdx.x * K + rank] = i;
  }
}

// out[b][k][:] = src[b][idx[b][k]][:]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx, T* __restrict__ out, int B,
                                   int P, int K, int C) {
  constexpr int V = 16 / sizeof(T);
  const int cvec = C / V;
  const long n = (long)B * K * cvec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cvec);
    const long bk = i / cvec;
    const int b = (int)(bk / K);
    const int r = idx[bk];
    *(uint4*)(out + (bk * cvec + cv) * V) = *(const uint4*)(src + (((long)b * P + r) * cvec + cv) * V);
  }
}

struct Levels { int n; int h[8]; int w[8]; };

// scores = sigmoid(topk logits); classes = argmax_c cls_logits; boxes = (offsets + scales*exp(box))*full
// num_instances[b] = #(scores > 0.5).  One workgroup per image; 8 lanes share an instance: each takes every eighth
// class of the argmax (the first maximum wins, as torch.max: ties go to the smaller class index) and lane 0 decodes the box;
// the count is a wave ballot + one LDS add per wave (no atomics on global memory, no memset in front).
// cls_logits / box_raw: rows cls_stride / box_stride elements apart (views of vector-padded MLP outputs).
template <typename T>
__global__ __launch_bounds__(256) void od_decode_kernel(const float* __restrict__ top_vals, const int* __restrict__ top_idx,
                                                        const T* __restrict__ cls_logits, long cls_stride,
                                                        const T* __restrict__ box_raw, long box_stride, Levels lv, int K, int ncls,
                                                        float full_w, float full_h, float* __restrict__ scores,
                                                        long* __restrict__ classes, float* __restrict__ boxes,
                                                        long* __restrict__ num_inst) {
  __shared__ int count;
  const int b = blockIdx.x, sub = threadIdx.x & 7;
  if (threadIdx.x == 0) count = 0;
  __syncthreads();
  int mine = 0;
  for (int k0 = 0; k0 < K; k0 += 32) {
    const int k = k0 + (threadIdx.x >> 3);
    const bool ok = k < K;
    const long i = (long)b * K + (ok ? k : 0);
    const T* cl = cls_logits + i * cls_stride;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = sub; c < ncls; c += 8) {
      const float v = elem<T>::ld(cl + c);
      if (v > best || (v != v && best == best)) { best = v; arg = c; }  // NaN wins, as torch.max
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const float ob = __shfl_xor(best, o);
      const int oa = __shfl_xor(arg, o);
      const bool take = (ob > best) || (ob != ob && best == best) || (ob == best && oa < arg) || (ob != ob && best != best && oa < arg);
      if (take) { best = ob; arg = oa; }
    }
    if (ok && sub == 0) {
      const float s = 1.f / (1.f + expf(-top_vals[i]));
      scores[i] = s;
      mine += s > 0.5f ? 1 : 0;
      classes[i] = arg;
      // position index -> level, cell
      int p = top_idx[i], l = 0;
      while (l < lv.n - 1 && p >= lv.h[l] * lv.w[l]) { p -= lv.h[l] * lv.w[l]; ++l; }
      const int h = lv.h[l], w = lv.w[l];
      const int cy = p / w, cx = p - cy * w;
      const float hx = 0.5f / w, hy = 0.5f / h;
      const float ox = (cx + 0.5f) / w, oy = (cy + 0.5f) / h;
      const T* br = box_raw + i * box_stride;
      float4 o;
      o.x = (ox - hx * expf(elem<T>::ld(br + 0))) * full_w;
      o.y = (oy - hy * expf(elem<T>::ld(br + 1))) * full_h;
      o.z = (ox + hx * expf(elem<T>::ld(br + 2))) * full_w;
      o.w = (oy + hy * expf(elem<T>::ld(br + 3))) * full_h;
      *(float4*)(boxes + i * 4) = o;
    }
  }
  mine = (int)wave_sum((float)mine);
  if ((threadIdx.x & 63) == 0) atomicAdd(&count, mine);
  __syncthreads();
  if (threadIdx.x == 0) num_inst[b] = count;
}

// ------------------------------------------------------------------ CondInst mask decode (instance segmentation)
// Per instance (b, k): a 3-layer 1x1 network whose 169 parameters were emitted by the kernel MLP runs over the 8
// mask-feature channels + 2 coordinates relative to the instance's location, at the mask level (h x w); sigmoid;
// bilinear resize (align_corners = False) to the full H x W.  The reference materialises (B, K, 10, h, w), three
// einsum results and the low-resolution masks before F.interpolate (instance_segmentation.py:121-163); here a
// workgroup owns one 64 x 64 OUTPUT tile of one instance, evaluates the network for the low-resolution pixels that
// tile interpolates from (into LDS) and writes the tile: the only HBM traffic besides the (tiny) inputs is the
// output itself.  Parameter layout of a row: w1[10][8], b1[8], w2[8][8], b2[8], w3[8], b3.
constexpr int ISEG_C = 8, ISEG_NP = (ISEG_C + 2) * ISEG_C + ISEG_C + ISEG_C * ISEG_C + ISEG_C + ISEG_C + 1;
constexpr int ISEG_LOW = 64 + 3;  // low-resolution window edge a tile may need (64 x 64 tile at scale 1; 128 x 128 at <= 1/2)

__device__ __forceinline__ float iseg_sigmoid(float v) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fminf(-v * 1.4426950408889634f, 126.f)));
}

template <typename T, int ISEG_OT>
__global__ void __launch_bounds__(256) iseg_mask_decode_kernel(const T* __restrict__ feats, const T* __restrict__ dyn, long dstride,
                                        const int* __restrict__ top_idx, Levels lv, int K, int h, int w, int H,
                                        int W, int tiles_x, int tiles_y, T* __restrict__ out) {
  __shared__ float par[ISEG_NP + 2];
  __shared__ float low[ISEG_LOW * ISEG_LOW];
  const int tile = blockIdx.x % (tiles_x * tiles_y), inst = blockIdx.x / (tiles_x * tiles_y);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x, b = inst / K;
  for (int i = threadIdx.x; i < ISEG_NP; i += blockDim.x) par[i] = elem<T>::ld(dyn + (long)inst * dstride + i);
  if (threadIdx.x == 0) {  // normalised centre of the instance's pyramid position
    int p = top_idx[inst], l = 0;
    while (l < lv.n - 1 && p >= lv.h[l] * lv.w[l]) { p -= lv.h[l] * lv.w[l]; ++l; }
    const int cy = p / lv.w[l], cx = p - cy * lv.w[l];
    par[ISEG_NP] = (cx + 0.5f) / lv.w[l];
    par[ISEG_NP + 1] = (cy + 0.5f) / lv.h[l];
  }
  // low-resolution window this output tile reads
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int oy0 = ty * ISEG_OT, ox0 = tx * ISEG_OT;
  const int oy1 = min(oy0 + ISEG_OT, H), ox1 = min(ox0 + ISEG_OT, W);
  const int ly0 = (int)fmaxf(sy * (oy0 + 0.5f) - 0.5f, 0.f), lx0 = (int)fmaxf(sx * (ox0 + 0.5f) - 0.5f, 0.f);
  const int ly1 = min(h - 1, (int)fmaxf(sy * (oy1 - 0.5f) - 0.5f, 0.f) + 1);
  const int lx1 = min(w - 1, (int)fmaxf(sx * (ox1 - 0.5f) - 0.5f, 0.f) + 1);
  const int nly = ly1 - ly0 + 1, nlx = lx1 - lx0 + 1;
  __syncthreads();
  const float ox = par[ISEG_NP], oy = par[ISEG_NP + 1];
  const float* w1 = par;
  const float* b1 = w1 + (ISEG_C + 2) * ISEG_C;
  const float* w2 = b1 + ISEG_C;
  const float* b2 = w2 + ISEG_C * ISEG_C;
  const float* w3 = b2 + ISEG_C;
  const float b3 = w3[ISEG_C];
  for (int i = threadIdx.x; i < nly * nlx; i += blockDim.x) {
    // the 169 parameters stay in LDS: hoisting them out of this (usually single-trip) loop into registers spilled
    asm volatile("" ::: "memory");
    const int ry = i / nlx, rx = i - ry * nlx, y = ly0 + ry, x = lx0 + rx;
    float in[ISEG_C + 2];
    const T* f = feats + (((long)b * h + y) * w + x) * ISEG_C;
#pragma unroll
    for (int c = 0; c < ISEG_C; ++c) in[c] = elem<T>::ld(f + c);
    in[ISEG_C] = (x + 0.5f) / w - ox;
    in[ISEG_C + 1] = (y + 0.5f) / h - oy;
    float a[ISEG_C], a2[ISEG_C];
#pragma unroll
    for (int d = 0; d < ISEG_C; ++d) {
      float t = b1[d];
#pragma unroll
      for (int c = 0; c < ISEG_C + 2; ++c) t += in[c] * w1[c * ISEG_C + d];
      a[d] = t * iseg_sigmoid(t);
    }
#pragma unroll
    for (int d = 0; d < ISEG_C; ++d) {
      float t = b2[d];
#pragma unroll
      for (int c = 0; c < ISEG_C; ++c) t += a[c] * w2[c * ISEG_C + d];
      a2[d] = t * iseg_sigmoid(t);
    }
    float t = b3;
#pragma unroll
    for (int c = 0; c < ISEG_C; ++c) t += a2[c] * w3[c];
    low[ry * ISEG_LOW + rx] = iseg_sigmoid(t);
  }
  __syncthreads();
  // bilinear write of the tile: a thread owns 8 consecutive output columns (their x taps are row-independent and
  // computed once) of rows (tid / 8) + 32 j; full 8-column groups leave as 16-byte stores
  constexpr int CG = ISEG_OT / 8;  // 8-column groups per tile row
  const int cx8 = (threadIdx.x % CG) * 8;
  int xa[8], xb[8];
  float wxb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int oxx = min(ox0 + cx8 + e, W - 1);
    const float fx = fmaxf(sx * (oxx + 0.5f) - 0.5f, 0.f);
    const int x0 = (int)fx;
    xa[e] = x0 - lx0;
    xb[e] = x0 + (x0 < w - 1 ? 1 : 0) - lx0;
    wxb[e] = fx - x0;
  }
  const bool full = ox0 + cx8 + 8 <= ox1 && (W % 8) == 0;
  for (int r = threadIdx.x / CG; r < oy1 - oy0; r += blockDim.x / CG) {
    const int oyy = oy0 + r;
    const float fy = fmaxf(sy * (oyy + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float wy1 = fy - y0, wy0 = 1.f - wy1;
    const float* r0 = low + (y0 - ly0) * ISEG_LOW;
    const float* r1 = low + (y1 - ly0) * ISEG_LOW;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float wxa = 1.f - wxb[e];  // same association as ATen's upsample_bilinear2d
      v[e] = wy0 * (wxa * r0[xa[e]] + wxb[e] * r0[xb[e]]) + wy1 * (wxa * r1[xa[e]] + wxb[e] * r1[xb[e]]);
    }
    T* o = out + ((long)inst * H + oyy) * W + ox0 + cx8;
    if (full) {
      if constexpr (sizeof(T) == 2) {
        *(uint4*)o = pack16(v, T());
      } else {
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ox0 + cx8 + e < ox1) elem<T>::st(o + e, v[e]);
    }
  }
}

// offsets (cx,cy,cx,cy) and scales (-1/2w,-1/2h,1/2w,1/2h) for every pyramid position (training-side anchors)
__global__ void od_anchors_kernel(Levels lv, int P, float* __restrict__ offsets, float* __restrict__ scales) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int p = i, l = 0;
  while (l < lv.n - 1 && p >= lv.h[l] * lv.w[l]) { p -= lv.h[l] * lv.w[l]; ++l; }
  const int h = lv.h[l], w = lv.w[l];
  const int cy = p / w, cx = p - cy * w;
  const float hx = 0.5f / w, hy = 0.5f / h, ox = (cx + 0.5f) / w, oy = (cy + 0.5f) / h;
  offsets[(long)i * 4 + 0] = ox; offsets[(long)i * 4 + 1] = oy; offsets[(long)i * 4 + 2] = ox; offsets[(long)i * 4 + 3] = oy;
  scales[(long)i * 4 + 0] = -hx; scales[(long)i * 4 + 1] = -hy; scales[(long)i * 4 + 2] = hx; scales[(long)i * 4 + 3] = hy;
}


// ---------------------------------------------------------------------------------------------------------------------
// Detection loss (ObjectDetection.training_step, src/sihl/heads/object_detection.py:157-208) in one pass: the four loss
// sums AND their gradients with respect to the head's outputs, from the target-only tensors of the matching.
//   location  BCE-with-logits(loc_logits, rel_iou == 1).sum() / loc_norm                                   (:157-163)
//   iou       MSE(iou_preds, rel_iou).sum() / iou_norm                                                      (:175-180)
//   box       (w * CIoU_loss(offsets + scales * exp(box_raw), gt / full)).sum() / wsum, x10 in the total    (:186-197)
//   class     (w * CE(cls_logits, gt class)).sum() / wsum                                                   (:199-208)
// over the fixed-size candidate rows of the matching (w = rel_iou where the anchor's assigned ground truth is the
// candidate's, 0 otherwise).  Every gradient is elementwise given the normalisers (inputs), so the backward needs no
// second pass: d_loc / d_iou / d_box / d_cls are written here, for an upstream gradient of 1.
// CIoU follows torchvision's complete_box_iou_loss (published definition): alpha is a constant in the
// gradient; max / min pass the gradient to the larger / smaller argument (half each on ties, as ATen does).
struct OdLossArgs {
  const void *loc, *iou, *box, *cls;                 // head outputs: [N1], [N1], [R][4], [R][C]
  const float *loc_target, *rel_iou;                 // [N1]
  const float *cand_off, *cand_scale, *tgt_box, *wts;  // [R][4] x3, [R]
  const long* tgt_cls;                               // [R]
  const float *loc_norm, *iou_norm, *wsum;           // device scalars
  const bool* none_matched;                          // device scalar
  void *d_loc, *d_iou, *d_box, *d_cls;
  float* partial;                                    // [blocks][4]
  long N1; int R, C;
};

__device__ __forceinline__ float gmax_w(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }  // d max(a,b)/da

template <typename T>
__global__ __launch_bounds__(256) void od_loss_kernel(const OdLossArgs a) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const bool nm = *a.none_matched;
  const float keep = nm ? 0.f : 1.f;  // degenerate ground truths only: the total is the location loss
  // (then wsum and iou_norm are 0 as well: the reciprocals are taken as 0 instead, so the gradients of the three unused
  // terms are exact zeros rather than 0 * inf)
  const float inv_iou_norm = nm ? 0.f : 1.f / *a.iou_norm;
  float s_loc = 0.f, s_iou = 0.f, s_box = 0.f, s_cls = 0.f;
  if (gid < a.N1) {
    const float x = elem<T>::ld((const T*)a.loc + gid), t = a.loc_target[gid];
    s_loc = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    elem<T>::st((T*)a.d_loc + gid, (1.f / (1.f + expf(-x)) - t) / *a.loc_norm);
    const float q = elem<T>::ld((const T*)a.iou + gid), r = a.rel_iou[gid], d = q - r;
    s_iou = d * d;
    elem<T>::st((T*)a.d_iou + gid, keep * 2.f * d * inv_iou_norm);
  }
  if (gid < a.R) {
    const int r = (int)gid;
    const float w = a.wts[r], inv_wsum = nm ? 0.f : 1.f / *a.wsum;
    // ---- box: pred = off + scale * exp(raw)
    float raw[4], e[4], b[4], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      raw[k] = elem<T>::ld((const T*)a.box + r * 4 + k);
      e[k] = expf(raw[k]);
      b[k] = a.cand_off[r * 4 + k] + a.cand_scale[r * 4 + k] * e[k];
      g[k] = a.tgt_box[r * 4 + k];
    }
    const float eps = 1e-7f;
    const float x1 = b[0], y1 = b[1], x2 = b[2], y2 = b[3], X1 = g[0], Y1 = g[1], X2 = g[2], Y2 = g[3];
    const float xk1 = fmaxf(x1, X1), yk1 = fmaxf(y1, Y1), xk2 = fminf(x2, X2), yk2 = fminf(y2, Y2);
    const bool valid = yk2 > yk1 && xk2 > xk1;
    const float iw = xk2 - xk1, ih = yk2 - yk1;
    const float inter = valid ? iw * ih : 0.f;
    const float pw = x2 - x1, ph = y2 - y1, gw = X2 - X1, gh = Y2 - Y1;
    const float uni = pw * ph + gw * gh - inter, ue = uni + eps;
    const float iouv = inter / ue;
    const float xc1 = fminf(x1, X1), yc1 = fminf(y1, Y1), xc2 = fmaxf(x2, X2), yc2 = fmaxf(y2, Y2);
    const float cw = xc2 - xc1, chh = yc2 - yc1, c2 = cw * cw + chh * chh + eps;
    const float sx = 0.5f * ((x1 + x2) - (X1 + X2)), sy = 0.5f * ((y1 + y2) - (Y1 + Y2));
    const float rho2 = sx * sx + sy * sy;
    const float kv = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float da = atanf(gw / gh) - atanf(pw / ph);
    const float v = kv * da * da;
    const float alpha = v / (1.f - iouv + v + eps);
    s_box = w * (1.f - iouv + rho2 / c2 + alpha * v);
    // gradient with respect to (x1, y1, x2, y2)
    float dint[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      dint[0] = -ih * gmax_w(x1, X1);  // xk1 = max(x1, X1)
      dint[1] = -iw * gmax_w(y1, Y1);
      dint[2] = ih * gmax_w(X2, x2);   // xk2 = min(x2, X2): x2 carries it when it is the smaller
      dint[3] = iw * gmax_w(Y2, y2);
    }
    const float darea[4] = {-ph, -pw, ph, pw};
    const float drho[4] = {sx, sy, sx, sy};
    const float dc2[4] = {-2.f * cw * gmax_w(X1, x1), -2.f * chh * gmax_w(Y1, y1), 2.f * cw * gmax_w(x2, X2),
                          2.f * chh * gmax_w(y2, Y2)};
    const float q = pw / ph, dat = 1.f / (1.f + q * q);  // d atan(q) / dq
    const float dv_dw = -2.f * kv * da * dat / ph, dv_dh = 2.f * kv * da * dat * pw / (ph * ph);
    const float dv[4] = {-dv_dw, -dv_dh, dv_dw, dv_dh};
    const float scale = keep * 10.f * w * inv_wsum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float duni = darea[k] - dint[k];
      const float diou = (dint[k] * ue - inter * duni) / (ue * ue);
      const float dL = -diou + (drho[k] * c2 - rho2 * dc2[k]) / (c2 * c2) + alpha * dv[k];
      // (a zero-weight row - padding, or every row when nothing matched - gets an exact zero: a degenerate target box
      // makes dL infinite or NaN, and 0 * that is NaN)
      elem<T>::st((T*)a.d_box + r * 4 + k, scale == 0.f ? 0.f : scale * dL * a.cand_scale[r * 4 + k] * e[k]);
    }
    // ---- class: cross-entropy over C logits
    const T* lg = (const T*)a.cls + (long)r * a.C;
    float m = -INFINITY;
    for (int c = 0; c < a.C; ++c) m = fmaxf(m, elem<T>::ld(lg + c));
    float z = 0.f;
    for (int c = 0; c < a.C; ++c) z += expf(elem<T>::ld(lg + c) - m);
    const int tc = (int)a.tgt_cls[r];
    const float lse = m + logf(z);
    s_cls = w * (lse - elem<T>::ld(lg + tc));
    const float cs = keep * w * inv_wsum, iz = 1.f / z;
    T* dl = (T*)a.d_cls + (long)r * a.C;
    for (int c = 0; c < a.C; ++c) elem<T>::st(dl + c, cs == 0.f ? 0.f : cs * (expf(elem<T>::ld(lg + c) - m) * iz - (c == tc ? 1.f : 0.f)));
  }
  // block sums (fixed order inside the block; blocks are summed in order by the finalize kernel)
  __shared__ float red[4][4];
  s_loc = wave_sum(s_loc); s_iou = wave_sum(s_iou); s_box = wave_sum(s_box); s_cls = wave_sum(s_cls);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = s_loc; red[wv][1] = s_iou; red[wv][2] = s_box; red[wv][3] = s_cls; }
  __syncthreads();
  if (threadIdx.x < 4)
    a.partial[(long)blockIdx.x * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// losses[5] = total, location, box, class, iou (the last three are 0 when none_matched, as the reference's early-out)
__global__ __launch_bounds__(256) void od_loss_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                               const float* loc_norm, const float* iou_norm,
                                                               const float* wsum, const bool* none_matched,
                                                               float* __restrict__ losses) {
  __shared__ double red[4][4];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += (double)partial[(long)b * 4 + k];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    const float loc = (float)(t[0] / (double)*loc_norm), iou = (float)(t[1] / (double)*iou_norm);
    const float box = (float)(t[2] / (double)*wsum), cls = (float)(t[3] / (double)*wsum);
    const bool nm = *none_matched;
    losses[0] = nm ? loc : loc + 10.f * box + cls + iou;
    losses[1] = loc;
    losses[2] = nm ? 0.f : box;
    losses[3] = nm ? 0.f : cls;
    losses[4] = nm ? 0.f : iou;
  }
}

}  // namespace

static bool g_topk_select = true;

extern "C" {

// Test hook: 0 = the full bitonic sort (kept as the fallback for rows beyond 65535 positions or K > 4096).
int sihl_topk_select_enable(int on) { g_topk_select = on != 0; return SIHL_OK; }

// x: [B][P] with `estride` elements between consecutive positions (dtype) -> vals fp32 [B][K] (sorted
// descending), idx int32 [B][K]
int sihl_topk_rows(const void* x, int B, int P, int K, int estride, float* vals, int* idx, int dtype,
                   hipStream_t stream) {
  if (!x || !vals || !idx || B <= 0 || P <= 0 || K <= 0 || K > P || estride <= 0) return SIHL_EARG;
  if (g_topk_select && P <= 0xFFFF && K <= 4096 && (size_t)(P + 2) * 4 + (size_t)K * 8 <= 150 * 1024) {
    const size_t lds_sel = (size_t)((P + 1) & ~1) * 4 + (size_t)K * 8;
    static bool sel_attr = false;
    if (!sel_attr) {
      hipError_t e1 = hipFuncSetAttribute((const void*)topk_select_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      hipError_t e2 = hipFuncSetAttribute((const void*)topk_select_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      if (e1 != hipSuccess) return (int)e1;
      if (e2 != hipSuccess) return (int)e2;
      sel_attr = true;
    }
    if (dtype == SIHL_F32)
      hipLaunchKernelGGL(topk_select_kernel<float>, dim3(B), dim3(1024), lds_sel, stream, (const float*)x, P, K, estride, vals, idx);
    else if (dtype == SIHL_BF16)
      hipLaunchKernelGGL(topk_select_kernel<bf16_t>, dim3(B), dim3(1024), lds_sel, stream, (const bf16_t*)x, P, K, estride, vals, idx);
    else return SIHL_EARG;
    SIHL_CHECK_LAUNCH();
    return SIHL_OK;
  }
  int P2 = 1;
  while (P2 < P) P2 <<= 1;
  const size_t lds = (size_t)P2 * 8;
  if (lds > 160 * 1024) return SIHL_EARG;  // > 20480 positions per image: not supported by the LDS sort
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute((const void*)topk_rows_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute((const void*)topk_rows_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    attr_set = true;
  }
  if (dtype == SIHL_F32)
    hipLaunchKernelGGL(topk_rows_kernel<float>, dim3(B), dim3(1024), lds, stream, (const float*)x, P, P2, K, estride, vals, idx);
  else if (dtype == SIHL_BF16)
    hipLaunchKernelGGL(topk_rows_kernel<bf16_t>, dim3(B), dim3(1024), lds, stream, (const bf16_t*)x, P, P2, K, estride, vals, idx);
  else return SIHL_EARG;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_gather_rows(const void* src, const int* idx, void* out, int B, int P, int K, int C, int dtype,
                     hipStream_t stream) {
  if (!src || !idx || !out || B <= 0 || P <= 0 || K <= 0 || C <= 0) return SIHL_EARG;
  const int V = dtype == SIHL_BF16 ? 8 : 4;
  if (C % V) return SIHL_EARG;
  long n = (long)B * K * (C / V);
  int g = (int)((n + 255) / 256);
  if (g > 4096) g = 4096;
  if (dtype == SIHL_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)src, idx, (float*)out, B, P, K, C);
  else if (dtype == SIHL_BF16)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)src, idx, (bf16_t*)out, B, P, K, C);
  else return SIHL_EARG;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// level_hw: host array [n_levels][2] = (h, w) of each pyramid level, bottom first.  cls_logits [B*K] rows of ncls values
// cls_stride elements apart, box_raw [B*K] rows of 4 values box_stride elements apart (0 = dense).
int sihl_od_decode(const float* top_vals, const int* top_idx, const void* cls_logits, long cls_stride, const void* box_raw,
                   long box_stride, const int* level_hw, int n_levels, int B, int K, int ncls, int full_w, int full_h,
                   float* scores, long* classes, float* boxes, long* num_instances, int dtype, hipStream_t stream) {
  if (!top_vals || !top_idx || !cls_logits || !box_raw || !level_hw || n_levels <= 0 || n_levels > 8 || B <= 0 ||
      K <= 0 || ncls <= 0 || !scores || !classes || !boxes || !num_instances)
    return SIHL_EARG;
  if (cls_stride == 0) cls_stride = ncls;
  if (box_stride == 0) box_stride = 4;
  if (cls_stride < ncls || box_stride < 4) return SIHL_EARG;
  Levels lv;
  lv.n = n_levels;
  for (int i = 0; i < n_levels; ++i) { lv.h[i] = level_hw[2 * i]; lv.w[i] = level_hw[2 * i + 1]; }
  if (dtype == SIHL_F32)
    hipLaunchKernelGGL(od_decode_kernel<float>, dim3(B), dim3(256), 0, stream, top_vals, top_idx, (const float*)cls_logits, cls_stride, (const float*)box_raw, box_stride, lv, K, ncls, (float)full_w, (float)full_h, scores, classes, boxes, num_instances);
  else if (dtype == SIHL_BF16)
    hipLaunchKernelGGL(od_decode_kernel<bf16_t>, dim3(B), dim3(256), 0, stream, top_vals, top_idx, (const bf16_t*)cls_logits, cls_stride, (const bf16_t*)box_raw, box_stride, lv, K, ncls, (float)full_w, (float)full_h, scores, classes, boxes, num_instances);
  else return SIHL_EARG;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int sihl_od_anchors(const int* level_hw, int n_levels, float* offsets, float* scales, hipStream_t stream) {
  if (!level_hw || n_levels <= 0 || n_levels > 8 || !offsets || !scales) return SIHL_EARG;
  Levels lv;
  lv.n = n_levels;
  int P = 0;
  for (int i = 0; i < n_levels; ++i) { lv.h[i] = level_hw[2 * i]; lv.w[i] = level_hw[2 * i + 1]; P += lv.h[i] * lv.w[i]; }
  hipLaunchKernelGGL(od_anchors_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, lv, P, offsets, scales);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// masks (B, K, H, W) from mask features (B, h, w, 8) NHWC, per-instance parameter rows dyn[(b*K+k)*dstride .. +169)
// and the instances' pyramid positions top_idx (B, K) (level_hw as in sihl_od_decode).  All tensors dtype.
int sihl_iseg_mask_decode(const void* feats, const void* dyn, long dstride, const int* top_idx, const int* level_hw,
                          int n_levels, int B, int K, int h, int w, int H, int W, void* out, int dtype,
                          hipStream_t stream) {
  if (!feats || !dyn || !top_idx || !level_hw || !out || n_levels <= 0 || n_levels > 8 || B <= 0 || K <= 0 ||
      h <= 0 || w <= 0 || H < h || W < w || dstride < ISEG_NP)
    return SIHL_EARG;
  Levels lv;
  lv.n = n_levels;
  for (int i = 0; i < n_levels; ++i) { lv.h[i] = level_hw[2 * i]; lv.w[i] = level_hw[2 * i + 1]; }
  // 128 x 128 output tiles when the low-resolution window they interpolate from still fits the LDS table
  const int ot = (2 * h <= H && 2 * w <= W) ? 128 : 64;
  const int tiles_x = (W + ot - 1) / ot, tiles_y = (H + ot - 1) / ot;
  const long blocks = (long)B * K * tiles_x * tiles_y;
  if (blocks > 0x7fffffffL) return SIHL_EARG;
#define SIHL_ISEG(T, OT) hipLaunchKernelGGL((iseg_mask_decode_kernel<T, OT>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)feats, (const T*)dyn, dstride, top_idx, lv, K, h, w, H, W, tiles_x, tiles_y, (T*)out)
  if (dtype == SIHL_F32) { if (ot == 128) SIHL_ISEG(float, 128); else SIHL_ISEG(float, 64); }
  else if (dtype == SIHL_BF16) { if (ot == 128) SIHL_ISEG(bf16_t, 128); else SIHL_ISEG(bf16_t, 64); }
#undef SIHL_ISEG
  else return SIHL_EARG;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}


// Workspace floats of sihl_od_loss: one row of four partial sums per 256-thread block.
long sihl_od_loss_ws_bytes(long n_positions, int n_rows) {
  const long n = n_positions > n_rows ? n_positions : n_rows;
  return ((n + 255) / 256) * 4L * (long)sizeof(float);
}

// Detection loss and its gradients in one pass (see od_loss_kernel).  loc / iou: [n_positions] head outputs (dtype),
// box [n_rows][4], cls [n_rows][C]; targets fp32 (tgt_cls int64); normalisers and none_matched are DEVICE scalars.
// losses[5] = total, location, box, class, iou; d_* = gradients of the total (same dtype / shape as the head outputs).
int sihl_od_loss(const void* loc, const void* iou, const void* box, const void* cls, const float* loc_target,
                 const float* rel_iou, const float* cand_off, const float* cand_scale, const float* tgt_box,
                 const float* wts, const long* tgt_cls, const float* loc_norm, const float* iou_norm, const float* wsum,
                 const void* none_matched, long n_positions, int n_rows, int C, void* d_loc, void* d_iou, void* d_box,
                 void* d_cls, float* losses, int dtype, float* ws, long ws_bytes, hipStream_t stream) {
  if (!loc || !iou || !box || !cls || !loc_target || !rel_iou || !cand_off || !cand_scale || !tgt_box || !wts ||
      !tgt_cls || !loc_norm || !iou_norm || !wsum || !none_matched || !d_loc || !d_iou || !d_box || !d_cls || !losses ||
      !ws || n_positions <= 0 || n_rows <= 0 || C <= 0)
    return SIHL_EARG;
  if (ws_bytes < sihl_od_loss_ws_bytes(n_positions, n_rows)) return SIHL_EWS;
  OdLossArgs a;
  a.loc = loc; a.iou = iou; a.box = box; a.cls = cls; a.loc_target = loc_target; a.rel_iou = rel_iou;
  a.cand_off = cand_off; a.cand_scale = cand_scale; a.tgt_box = tgt_box; a.wts = wts; a.tgt_cls = tgt_cls;
  a.loc_norm = loc_norm; a.iou_norm = iou_norm; a.wsum = wsum; a.none_matched = (const bool*)none_matched;
  a.d_loc = d_loc; a.d_iou = d_iou; a.d_box = d_box; a.d_cls = d_cls; a.partial = ws;
  a.N1 = n_positions; a.R = n_rows; a.C = C;
  const long n = n_positions > n_rows ? n_positions : n_rows;
  const int blocks = (int)((n + 255) / 256);
  if (dtype == SIHL_F32) hipLaunchKernelGGL(od_loss_kernel<float>, dim3(blocks), dim3(256), 0, stream, a);
  else if (dtype == SIHL_BF16) hipLaunchKernelGGL(od_loss_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, a);
  else return SIHL_EARG;
  hipLaunchKernelGGL(od_loss_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, blocks, loc_norm, iou_norm,
                     wsum, (const bool*)none_matched, losses);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
