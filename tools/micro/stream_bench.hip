// Developer microbenchmark (GPU box): which loop structure streams best for the 2-read/1-write "apply" pattern and
// the 2-read "reduce" pattern on MI355X.  Buffers rotate over >1 GB so the Infinity Cache cannot serve repeats.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned short u16;
typedef unsigned vu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) { vu4 v = __builtin_nontemporal_load((const vu4*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(uint4 r, uint4* p) { vu4 v = {r.x, r.y, r.z, r.w}; __builtin_nontemporal_store(v, (vu4*)p); }
__device__ __forceinline__ void unpack(uint4 v, float* f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ unsigned rn(float f) { unsigned u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }
__device__ __forceinline__ uint4 pack(const float* f) {
  uint4 v; v.x = rn(f[0]) | (rn(f[1]) << 16); v.y = rn(f[2]) | (rn(f[3]) << 16); v.z = rn(f[4]) | (rn(f[5]) << 16); v.w = rn(f[6]) | (rn(f[7]) << 16);
  return v;
}
__device__ __forceinline__ uint4 work(uint4 a, uint4 b, const float* A, const float* B, const float* D) {
  float fa[8], fb[8];
  unpack(a, fa); unpack(b, fb);
#pragma unroll
  for (int e = 0; e < 8; ++e) { float r = A[e] * fb[e] + B[e] * fa[e] + D[e]; fa[e] = fa[e] > 0.f ? r : 0.f; }
  return pack(fa);
}
// U vectors in flight per thread; grid-stride
template <int U, bool NT>
__global__ void apply_loop(const uint4* __restrict__ s, const uint4* __restrict__ dy, uint4* __restrict__ dz, long nvec, int cvec,
                           const float* __restrict__ pa, const float* __restrict__ pb, const float* __restrict__ pd) {
  float A[8], B[8], D[8];
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(i0 % cvec) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { A[e] = pa[c + e]; B[e] = pb[c + e]; D[e] = pd[c + e]; }
  const long stride = (long)gridDim.x * 256;
  for (long i = i0; i < nvec; i += stride * U) {
    uint4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      if (j < nvec) {
        if (NT) { a[u] = ntload(s + j); b[u] = ntload(dy + j); }
        else { a[u] = s[j]; b[u] = dy[j]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * stride;
      if (j < nvec) { uint4 r = work(a[u], b[u], A, B, D); if (NT) ntstore(r, dz + j); else dz[j] = r; }
    }
  }
}
// reduce: block = rows [r0, r1); thread = (cv, rl); U rows in flight; LDS fold to one partial row per block
template <int U>
__global__ void reduce_loop(const uint4* __restrict__ s, const uint4* __restrict__ dy, long rows, int cvec, float* __restrict__ part,
                            int rpb) {
  __shared__ float red[8 * 2 * 256];
  const int nrl = 256 / cvec, rl = threadIdx.x / cvec, cv = threadIdx.x % cvec, C = cvec * 8;
  const long r0 = (long)blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  float sb[8], sg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sb[e] = sg[e] = 0.f;
  for (long r = r0 + rl; r < r1; r += (long)nrl * U) {
    uint4 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long rr = r + (long)u * nrl; if (rr < r1) { a[u] = s[rr * cvec + cv]; b[u] = dy[rr * cvec + cv]; } }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * nrl;
      if (rr < r1) {
        float fa[8], fb[8];
        unpack(a[u], fa); unpack(b[u], fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sb[e] += fb[e]; sg[e] += fb[e] * fa[e]; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[(rl * 2 + 0) * C + cv * 8 + e] = sb[e]; red[(rl * 2 + 1) * C + cv * 8 + e] = sg[e]; }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * C; idx += 256) {
    float t = 0.f;
    for (int l = 0; l < nrl; ++l) t += red[l * 2 * C + idx];
    part[(long)blockIdx.x * 2 * C + idx] = t;
  }
}
int main() {
  const int C = 256, cvec = C / 8;
  for (long rows : {131072L, 524288L}) {
    const long nvec = rows * cvec;
    const size_t bytes = (size_t)rows * C * 2;
    const int nbuf = (int)(1.6e9 / (3 * bytes)) < 2 ? 2 : (int)(1.6e9 / (3 * bytes));
    std::vector<uint4*> s(nbuf), d(nbuf), z(nbuf);
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&s[i], bytes)); CK(hipMalloc(&d[i], bytes)); CK(hipMalloc(&z[i], bytes)); CK(hipMemset(s[i], 0x3c, bytes)); CK(hipMemset(d[i], 0x3d, bytes)); }
    float *pa, *part;
    CK(hipMalloc(&pa, 3 * C * 4)); CK(hipMemset(pa, 0, 3 * C * 4));
    CK(hipMalloc(&part, 8192L * 2 * C * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, int passes, auto launch) {
      for (int i = 0; i < nbuf; ++i) launch(i);
      hipEventRecord(e0);
      const int iters = 30;
      for (int i = 0; i < iters; ++i) launch(i % nbuf);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / iters;
      printf("rows %ld %-28s %7.1f us  %5.2f TB/s\n", rows, name, us, passes * (double)bytes / us / 1e6);
    };
#define APPLY(U, NT, G) run("apply U" #U " nt" #NT " g" #G, 3, [&](int i) { hipLaunchKernelGGL((apply_loop<U, NT>), dim3(G), dim3(256), 0, 0, s[i], d[i], z[i], nvec, cvec, pa, pa + C, pa + 2 * C); })
    APPLY(1, false, 2048); APPLY(1, false, 4096); APPLY(1, false, 8192); APPLY(1, false, (int)(nvec / 256));
    APPLY(2, false, 2048); APPLY(2, false, 4096); APPLY(2, false, (int)(nvec / 512));
    APPLY(4, false, 2048); APPLY(4, false, 4096); APPLY(4, false, (int)(nvec / 1024));
    APPLY(1, true, 4096); APPLY(2, true, 4096); APPLY(4, true, 2048); APPLY(1, true, (int)(nvec / 256)); APPLY(2, true, (int)(nvec / 512));
#define REDUCE(U, NB) run("reduce U" #U " nb" #NB, 2, [&](int i) { const int nb = NB; hipLaunchKernelGGL((reduce_loop<U>), dim3(nb), dim3(256), 0, 0, s[i], d[i], rows, cvec, part, (int)((rows + nb - 1) / nb)); })
    REDUCE(1, 512); REDUCE(1, 1024); REDUCE(1, 2048); REDUCE(1, 4096);
    REDUCE(2, 1024); REDUCE(2, 2048); REDUCE(4, 1024); REDUCE(4, 2048); REDUCE(4, 4096); REDUCE(8, 2048);
    for (int i = 0; i < nbuf; ++i) { hipFree(s[i]); hipFree(d[i]); hipFree(z[i]); }
    hipFree(pa); hipFree(part);
  }
  return 0;
}
