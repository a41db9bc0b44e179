// The ResNet stem's 7x7 / stride 2 / pad 3 convolution over 3 input channels (torchvision resnet.py conv1, wrapped by the
// reference's src/sihl/torchvision_backbone.py:42-49), bf16 on the matrix cores.
//
// The general implicit-GEMM kernel walks K as (tap) x (64 channels): with 3 channels it would multiply 95 % zeros.  Here K is
// (kernel row ky) x (one 32-element segment of the input row): in NHWC with C = 3 the 7 x 3 = 21 values under a kernel row
// are CONTIGUOUS in memory, so for output pixel (oy, ox) and kernel row ky the operand is the 32 consecutive bf16 values that
// start one pixel left of the window (a 4-byte aligned address; 24 of them meet the window and its left neighbour, the last
// 8 belong to pixels further right) and the weights are laid out to match: wp[co][ky][j], j = 3 * (kx + 1) + c, zeros at
// j < 3 and j >= 24.  7 k-steps of 32 = 224 against 147 useful (1.5 x), where the general kernel would spend 21 x.
//
//   * the image is first packed into a zero-padded NHWC bf16 copy xp[N][H + 6][Wp][3] (3 rows above / below, 4 columns
//     left, enough right that every 64-byte segment of a 64-pixel chunk stays inside the row): no border cases in the loop,
//     and the copy is what the weight gradient reads as well;
//   * MFMA v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the first operand: D[co][pixel], lane = pixel (lane % 16), its 4
//     accumulator registers = 4 consecutive output channels -> 8-byte stores, 32 contiguous bytes per pixel and instruction;
//     the input fragment of a lane is ONE 16-byte global load straight into registers (neighbouring pixels' segments overlap
//     by 52 of 64 bytes: L1 / L2 traffic, not HBM); the 28 KiB of packed weights sit in LDS;
//   * a workgroup (4 waves) walks ROWS output rows, a wave 64 pixels of a row at a time (4 pixel tiles x 4 channel tiles =
//     16 accumulator tiles); per-channel sums and sums of squares of the output (BatchNorm batch statistics) are kept per
//     lane and folded once per workgroup into one partial row [2][64] in sihl_bn_finalize's layout.
#include "common.h"

namespace {

constexpr int SKY = 7, SSEG = 32, SCO = 64;          // kernel rows, elements per segment, output channels
constexpr int SLEFT = 4, STOP = 3;                   // padding of the packed image (columns left, rows above)
constexpr int SROWS = 4;                             // output rows per workgroup

// a 16-byte operand at a 4-byte aligned address (segments start at multiples of 12 bytes)
struct __attribute__((packed, aligned(4))) seg16_t { uint32_t a, b, c, d; };
__device__ __forceinline__ uint4 ld_seg(const uint16_t* p) {
  const seg16_t v = *(const seg16_t*)p;
  return make_uint4(v.a, v.b, v.c, v.d);
}

__host__ __device__ inline int stem_wp(int W) {      // padded row length in pixels (multiple of 4)
  const int Wo = (W - 1) / 2 + 1, Wor = (Wo + 63) / 64 * 64;
  const int need = 2 * Wor + 12;
  const int wp = need > W + SLEFT + 4 ? need : W + SLEFT + 4;
  return (wp + 3) / 4 * 4;
}

// x: [N][3][H][W] with element strides (sn, sc, sh, sw), fp32 or bf16 -> xp bf16 [N][H + 6][Wp][3], zero border.
// A thread writes 4 pixels = 24 bytes (three 8-byte stores).
template <typename TI, bool VEC4 = false>
__global__ void stem_pack_image_kernel(const TI* __restrict__ x, long sn, long sc, long sh, long sw, uint16_t* __restrict__ xp,
                                       int N, int H, int W, int Wp) {
  const int Hp = H + 2 * STOP, q = Wp / 4;
  const long total = (long)N * Hp * q;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int px0 = (int)(i % q) * 4;
    long r = i / q;
    const int py = (int)(r % Hp);
    const long n = r / Hp;
    const int iy = py - STOP;
    uint16_t v[12];
    if (VEC4) {  // fp32 planes with unit pixel stride, 16-byte aligned rows: one float4 per channel (px0 - 4 is a multiple of 4)
      const int ix0 = px0 - SLEFT;
      const bool ok = iy >= 0 && iy < H && ix0 >= 0 && ix0 + 3 < W;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) f = *(const float4*)((const float*)x + n * sn + c * sc + (long)iy * sh + ix0);
        v[c] = f32_to_bf16(f.x); v[3 + c] = f32_to_bf16(f.y); v[6 + c] = f32_to_bf16(f.z); v[9 + c] = f32_to_bf16(f.w);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ix = px0 + k - SLEFT;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float f = 0.f;
          if (ok) {
            const TI* p = x + n * sn + c * sc + (long)iy * sh + (long)ix * sw;
            if constexpr (sizeof(TI) == 4) f = *(const float*)p;
            else f = bf16_to_f32(*(const uint16_t*)p);
          }
          v[k * 3 + c] = f32_to_bf16(f);
        }
      }
    }
    uint2* dst = (uint2*)(xp + (((long)n * Hp + py) * Wp + px0) * 3);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dst[k] = make_uint2((uint32_t)v[4 * k] | ((uint32_t)v[4 * k + 1] << 16), (uint32_t)v[4 * k + 2] | ((uint32_t)v[4 * k + 3] << 16));
  }
}

// w: [64][3][7][7] (OIHW, element strides so, sc, sh, sw), fp32 -> wp bf16 [64][7][32], j = 3 * (kx + 1) + c
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, long so, long sc, long sh, long sw, uint16_t* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= SCO * SKY * SSEG) return;
  const int j = i % SSEG, ky = (i / SSEG) % SKY, co = i / (SSEG * SKY);
  float f = 0.f;
  if (j >= 3 && j < 24) {
    const int kx = j / 3 - 1, c = j % 3;
    f = w[co * so + c * sc + ky * sh + kx * sw];
  }
  wp[i] = f32_to_bf16(f);
}

__global__ __launch_bounds__(256, 2) void stem_conv_fwd_kernel(const uint16_t* __restrict__ xp, const uint16_t* __restrict__ wp,
                                                             uint16_t* __restrict__ out, float* __restrict__ stats, int N, int H,
                                                             int W, int Wp, int Ho, int Wo) {
  __shared__ __attribute__((aligned(16))) uint16_t wl[SCO * SKY * SSEG];  // 28 KiB
  __shared__ float red[4][2][SCO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, g = lane >> 4;
  const int Hp = H + 2 * STOP;
  // LDS image [ky][co][4 chunks of 16 B], chunk c of channel co stored at position c ^ swz(co): the 16 lanes of every
  // ds_read_b128 group (channels p = 0-3, 12-15 at chunk g and 4-11 at chunk g + 1) then hit 16 distinct 4-bank slots
  for (int i = tid; i < SCO * SKY * SSEG / 8; i += 256) {
    const int c = i & 3, co = (i >> 2) % SCO, ky = i / (4 * SCO);
    ((uint4*)wl)[i] = ((const uint4*)wp)[(co * SKY + ky) * 4 + (c ^ ((0x78 >> (2 * ((co >> 2) & 3))) & 3))];
  }
  __syncthreads();
  const int wsw = 8 * (g ^ ((0x78 >> (2 * (p >> 2))) & 3));  // this lane's chunk position (p >> 2 = (co >> 2) & 3 for co = 16 t + p)
  const int row_groups = (Ho + SROWS - 1) / SROWS;
  const int n = blockIdx.x / row_groups, oy0 = (blockIdx.x % row_groups) * SROWS;
  float ssum[4][4], ssq[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[t][r] = ssq[t][r] = 0.f;
  const int nchunks = (Wo + 63) / 64;  // 64-pixel chunks of a row, dealt to the waves
  for (int rr = 0; rr < SROWS; ++rr) {
    const int oy = oy0 + rr;
    if (oy >= Ho) break;
    for (int ch = wave; ch < nchunks; ch += 4) {
      const int ox0 = ch * 64;
      f32x4_t acc[4][4];  // [channel tile][pixel tile]
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // lane's segment of pixel tile m, kernel row ky: xp[n][2 oy + ky][elements 6 (ox0 + 16 m + p) + 8 g ..]
      const uint16_t* xrow = xp + (((long)n * Hp + 2 * oy) * Wp) * 3 + 6 * (ox0 + p) + 8 * g;
      uint4 xf[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) xf[0][m] = ld_seg(xrow + 96 * m);
#pragma unroll
      for (int ky = 0; ky < SKY; ++ky) {
        if (ky + 1 < SKY) {
          const uint16_t* xn = xrow + (long)(ky + 1) * Wp * 3;
#pragma unroll
          for (int m = 0; m < 4; ++m) xf[(ky + 1) & 1][m] = ld_seg(xn + 96 * m);
        }
        // (compiler fence: without it the 28 weight fragments are hoisted out of the row / chunk loops - 112 registers,
        // spills; re-reading 4 fragments per kernel row costs the LDS 16 B/clk per wave)
        asm volatile("" ::: "memory");
        uint4 wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = *(const uint4*)(wl + (ky * SCO + 16 * t + p) * SSEG + wsw);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[t]),
                                                                __builtin_bit_cast(bf16x8_t, xf[ky & 1][m]), acc[t][m], 0, 0, 0);
      }
      // D[co = 16 t + 4 g + r][pixel = ox0 + 16 m + p]
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int ox = ox0 + 16 * m + p;
        if (ox < Wo) {
          uint16_t* o = out + (((long)n * Ho + oy) * Wo + ox) * SCO + 4 * g;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f32x4_t v = acc[t][m];
            *(uint2*)(o + 16 * t) = make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                                               (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
            if (stats) {
#pragma unroll
              for (int r = 0; r < 4; ++r) { ssum[t][r] += v[r]; ssq[t][r] += v[r] * v[r]; }
            }
          }
        }
      }
    }
  }
  if (stats) {
    // fold the 16 pixels of a lane group (same channels), then the 4 waves: one partial row per workgroup
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = ssum[t][r], b = ssq[t][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (p == 0) { red[wave][0][16 * t + 4 * g + r] = a; red[wave][1][16 * t + 4 * g + r] = b; }
      }
    __syncthreads();
    if (tid < 2 * SCO) {
      const int which = tid / SCO, co = tid % SCO;
      stats[((long)blockIdx.x * 2 + which) * SCO + co] = red[0][which][co] + red[1][which][co] + red[2][which][co] + red[3][which][co];
    }
  }
}

}  // namespace

extern "C" {

// bytes of the packed image copy / element count of the padded row (the caller allocates xp and keeps it for the backward)
long sihl_stem_xp_bytes(int N, int H, int W) { return (long)N * (H + 2 * STOP) * stem_wp(W) * 3 * 2; }
int sihl_stem_stats_rows(int N, int H) { return N * ((((H - 1) / 2 + 1) + SROWS - 1) / SROWS); }

// out[N][Ho][Wo][64] (bf16) = conv7x7 / stride 2 / pad 3 of x[N][3][H][W] (fp32 or bf16, element strides given) with the
// fp32 weights w[64][3][7][7] (element strides given); xp: sihl_stem_xp_bytes, wp: 64 * 7 * 32 * 2 bytes; stats (optional):
// [sihl_stem_stats_rows][2][64] partial sums / sums of squares of the output for sihl_bn_finalize.
int sihl_stem_conv_fwd(const void* x, int x_dtype, long xsn, long xsc, long xsh, long xsw, const float* w, long wso, long wsc,
                       long wsh, long wsw, void* xp, void* wp, void* out, float* stats, int N, int H, int W,
                       hipStream_t stream) {
  if (!x || !w || !xp || !wp || !out || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return SIHL_EARG;
  const int Wp = stem_wp(W), Ho = H / 2, Wo = W / 2;
  const long quads = (long)N * (H + 2 * STOP) * (Wp / 4);
  long gp = (quads + 255) / 256;
  if (gp > 65535) gp = 65535;
  const bool vec4 = x_dtype == SIHL_F32 && xsw == 1 && (W % 4) == 0 && (xsh % 4) == 0 && (xsc % 4) == 0 && (xsn % 4) == 0 &&
                    ((uintptr_t)x % 16) == 0;
  if (vec4)
    hipLaunchKernelGGL((stem_pack_image_kernel<float, true>), dim3((unsigned)gp), dim3(256), 0, stream, (const float*)x, xsn, xsc, xsh,
                       xsw, (uint16_t*)xp, N, H, W, Wp);
  else if (x_dtype == SIHL_F32)
    hipLaunchKernelGGL(stem_pack_image_kernel<float>, dim3((unsigned)gp), dim3(256), 0, stream, (const float*)x, xsn, xsc, xsh, xsw,
                       (uint16_t*)xp, N, H, W, Wp);
  else if (x_dtype == SIHL_BF16)
    hipLaunchKernelGGL(stem_pack_image_kernel<uint16_t>, dim3((unsigned)gp), dim3(256), 0, stream, (const uint16_t*)x, xsn, xsc, xsh,
                       xsw, (uint16_t*)xp, N, H, W, Wp);
  else return SIHL_EARG;
  hipLaunchKernelGGL(stem_pack_weight_kernel, dim3((SCO * SKY * SSEG + 255) / 256), dim3(256), 0, stream, w, wso, wsc, wsh, wsw,
                     (uint16_t*)wp);
  const int grid = N * ((Ho + SROWS - 1) / SROWS);
  hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(grid), dim3(256), 0, stream, (const uint16_t*)xp, (const uint16_t*)wp,
                     (uint16_t*)out, stats, N, H, W, Wp, Ho, Wo);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
