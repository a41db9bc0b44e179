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
// num_instances[b] = #(scores > 0.5).  One thread per (b, k).
template <typename T>
__global__ void od_decode_kernel(const float* __restrict__ top_vals, const int* __restrict__ top_idx,
                                 const T* __restrict__ cls_logits, const T* __restrict__ box_raw, Levels lv, int B,
                                 int K, int ncls, float full_w, float full_h, float* __restrict__ scores,
                                 long* __restrict__ classes, float* __restrict__ boxes, long* __restrict__ num_inst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K;
  const float s = 1.f / (1.f + expf(-top_vals[i]));
  scores[i] = s;
  if (s > 0.5f) atomicAdd((unsigned long long*)(num_inst + b), 1ULL);
  // argmax (first maximum wins, as torch.max)
  const T* cl = cls_logits + (long)i * ncls;
  float best = elem<T>::ld(cl);
  int arg = 0;
  for (int c = 1; c < ncls; ++c) {
    const float v = elem<T>::ld(cl + c);
    if (v > best) { best = v; arg = c; }
  }
  classes[i] = arg;
  // position index -> level, cell
  int p = top_idx[i], l = 0;
  while (l < lv.n - 1 && p >= lv.h[l] * lv.w[l]) { p -= lv.h[l] * lv.w[l]; ++l; }
  const int h = lv.h[l], w = lv.w[l];
  const int cy = p / w, cx = p - cy * w;
  const float hx = 0.5f / w, hy = 0.5f / h;
  const float ox = (cx + 0.5f) / w, oy = (cy + 0.5f) / h;
  const T* br = box_raw + (long)i * 4;
  boxes[(long)i * 4 + 0] = (ox - hx * expf(elem<T>::ld(br + 0))) * full_w;
  boxes[(long)i * 4 + 1] = (oy - hy * expf(elem<T>::ld(br + 1))) * full_h;
  boxes[(long)i * 4 + 2] = (ox + hx * expf(elem<T>::ld(br + 2))) * full_w;
  boxes[(long)i * 4 + 3] = (oy + hy * expf(elem<T>::ld(br + 3))) * full_h;
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

}  // namespace

extern "C" {

// x: [B][P] with `estride` elements between consecutive positions (dtype) -> vals fp32 [B][K] (sorted
// descending), idx int32 [B][K]
int sihl_topk_rows(const void* x, int B, int P, int K, int estride, float* vals, int* idx, int dtype,
                   hipStream_t stream) {
  if (!x || !vals || !idx || B <= 0 || P <= 0 || K <= 0 || K > P || estride <= 0) return SIHL_EARG;
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

// level_hw: host array [n_levels][2] = (h, w) of each pyramid level, bottom first.
int sihl_od_decode(const float* top_vals, const int* top_idx, const void* cls_logits, const void* box_raw,
                   const int* level_hw, int n_levels, int B, int K, int ncls, int full_w, int full_h, float* scores,
                   long* classes, float* boxes, long* num_instances, int dtype, hipStream_t stream) {
  if (!top_vals || !top_idx || !cls_logits || !box_raw || !level_hw || n_levels <= 0 || n_levels > 8 || B <= 0 ||
      K <= 0 || ncls <= 0 || !scores || !classes || !boxes || !num_instances)
    return SIHL_EARG;
  Levels lv;
  lv.n = n_levels;
  for (int i = 0; i < n_levels; ++i) { lv.h[i] = level_hw[2 * i]; lv.w[i] = level_hw[2 * i + 1]; }
  hipError_t e = hipMemsetAsync(num_instances, 0, (size_t)B * sizeof(long), stream);
  if (e != hipSuccess) return (int)e;
  const int g = (B * K + 255) / 256;
  if (dtype == SIHL_F32)
    hipLaunchKernelGGL(od_decode_kernel<float>, dim3(g), dim3(256), 0, stream, top_vals, top_idx, (const float*)cls_logits, (const float*)box_raw, lv, B, K, ncls, (float)full_w, (float)full_h, scores, classes, boxes, num_instances);
  else if (dtype == SIHL_BF16)
    hipLaunchKernelGGL(od_decode_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, top_vals, top_idx, (const bf16_t*)cls_logits, (const bf16_t*)box_raw, lv, B, K, ncls, (float)full_w, (float)full_h, scores, classes, boxes, num_instances);
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

}  // extern "C"
