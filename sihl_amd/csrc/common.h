// Shared device/host helpers for the sihl hot-path kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SIHL_F32 0
#define SIHL_BF16 1
#define SIHL_ACT_NONE 0
#define SIHL_ACT_RELU 1
#define SIHL_ACT_SILU 2
#define SIHL_ACT_SIGMOID 3

#define SIHL_OK 0
#define SIHL_EARG (-1)    // bad argument / unsupported shape
#define SIHL_EWS (-2)     // workspace too small

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct bf16_t { uint16_t bits; };  // storage-only tag type for bf16 tensors

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

template <typename T> struct elem;
template <> struct elem<float> {
  static constexpr int VEC = 4;  // elements per 16 B
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->bits); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }
};

// unpack / pack one 16-byte vector of T to/from floats
__device__ __forceinline__ void unpack16(const uint4& v, float (&f)[4], float) {
  f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ uint4 pack16(const float (&f)[4], float) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
__device__ __forceinline__ void unpack16(const uint4& v, float (&f)[8], bf16_t) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack16(const float (&f)[8], bf16_t) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16(f[2 * i]) | ((uint32_t)f32_to_bf16(f[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SIHL_ACT_RELU: return fmaxf(v, 0.f);
    case SIHL_ACT_SILU: return v / (1.f + expf(-v));
    case SIHL_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}
// d act(v)/dv given the pre-activation v
__device__ __forceinline__ float act_grad(float v, int act) {
  switch (act) {
    case SIHL_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case SIHL_ACT_SILU: { float s = 1.f / (1.f + expf(-v)); return s * (1.f + v * (1.f - s)); }
    case SIHL_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-v)); return s * (1.f - s); }
    default: return 1.f;
  }
}

// The two BiFPN fusion nodes, element by element, with the rounding points spelled out (explicit fma / mul: hipcc contracts
// a * b + c on its own, and differently from one kernel to the next).  elementwise.hip (stand-alone nodes) and conv_pyr.hip
// (node inside the conv loader) both use these, so the two forms give the same bits.
__device__ __forceinline__ float node_up2(float ly0, float ly1, float lx0, float lx1, float f00, float f01, float f10, float f11) {
  const float t0 = __fmaf_rn(lx1, f01, __fmul_rn(lx0, f00));
  const float t1 = __fmaf_rn(lx1, f11, __fmul_rn(lx0, f10));
  return __fmaf_rn(ly1, t1, __fmul_rn(ly0, t0));
}
__device__ __forceinline__ float node_fuse2(float w0, float w1, float up, float b) { return __fmaf_rn(w1, b, __fmul_rn(w0, up)); }
__device__ __forceinline__ float node_fuse3(float w0, float w1, float w2, float a, float b, float c) {
  return __fmaf_rn(w2, c, __fmaf_rn(w1, b, __fmul_rn(w0, a)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// XCD-aware bijective block remap: blocks b and b+8 share an XCD (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

#define SIHL_CHECK_LAUNCH()                     \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

// Progress-based issue priority inside a K-loop stage (s_setprio, 0-3): a wave lowers its priority as it advances from one stage
// barrier to the next.  The SIMD's arbiter serves its OLDEST ready wave first, so without this the waves of a SIMD finish a
// stage one after the other: the oldest idles at the barrier (43 % of the weight-gradient loop), the youngest runs the end
// of every stage alone with its LDS latencies exposed (tools/wgrad_stamps.py).  Measured (profiles/r04_prio_lib_ab.txt,
// r04_wgrad_stamps.txt): weight-gradient loop 3 086 -> 2 890 cycles per stage; the forward tiles (conv_igemm 256 x 256,
// conv_halo) unchanged.  The priority is SIMD-wide, across kernels: a weight gradient running BESIDE the dgrad chain on a
// second stream took issue slots from the critical path's kernels (step 26.83 -> 27.16 ms), so the kernel raises it only
// when it has the device to itself (WgradParams::prio).
#define SIHL_PRIO(n) __builtin_amdgcn_s_setprio(n)
