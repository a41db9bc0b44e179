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
//     accumulator registers = 4 consecutive output channels (8 bytes), staged through a per-wave LDS tile into 16-byte stores;
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
constexpr int OTROW = SCO * 2 + 16;                  // bytes per pixel row of a wave's output staging tile (padded)

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
template <typename TI, int VEC4 = 0>  // VEC4: 1 = fp32 planes (NCHW), 2 = fp32 channels-last (NHWC): 16-byte loads
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
    if (VEC4 == 2) {  // fp32 NHWC: the 4 pixels are 12 consecutive floats
      const int ix0 = px0 - SLEFT;
      float f[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) f[k] = 0.f;
      if (iy >= 0 && iy < H && ix0 >= 0 && ix0 + 3 < W) {
        const float4* src = (const float4*)((const float*)x + n * sn + (long)iy * sh + (long)ix0 * 3);
        const float4 a = src[0], b = src[1], c = src[2];
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        f[8] = c.x; f[9] = c.y; f[10] = c.z; f[11] = c.w;
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = f32_to_bf16(f[k]);
    } else if (VEC4 == 1) {  // fp32 planes with unit pixel stride, 16-byte aligned rows: one float4 per channel (px0 - 4 is a multiple of 4)
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
  __shared__ __attribute__((aligned(16))) char ot[4][64 * OTROW];  // per wave: its 64 pixels x 64 channels, for 16-byte row stores
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
      // D[co = 16 t + 4 g + r][pixel = ox0 + 16 m + p]: 8 bytes (4 channels) per lane and tile - transposed through the wave's
      // LDS tile so that the global stores are 16 bytes per lane and 1 KiB contiguous per instruction (8-byte stores
      // straight from the accumulators: 115 us for the bs-32 stem; this form: 95)
      char* tile = ot[wave];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool ok = ox0 + 16 * m + p < Wo;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4_t v = acc[t][m];
          *(uint2*)(tile + (16 * m + p) * OTROW + (16 * t + 4 * g) * 2) =
              make_uint2((uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16),
                         (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16));
          if (stats && ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { ssum[t][r] += v[r]; ssq[t][r] += v[r] * v[r]; }
          }
        }
      }
      uint16_t* orow = out + (((long)n * Ho + oy) * Wo + ox0) * SCO;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int id = lane + 64 * k, px = id >> 3, c = id & 7;
        if (ox0 + px < Wo) *(uint4*)(orow + px * SCO + c * 8) = *(const uint4*)(tile + px * OTROW + c * 16);
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

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dwp[co][ky][j] = sum over output pixels of dz[pixel][co] * xp[n][2 oy + ky][6 ox + j] - the contraction runs
// over PIXELS, the slow axis of both operands, so 64-pixel tiles are staged [pixel][channel] in LDS (dz: 64 x 64; the
// image: per kernel row the 64 pixels' 32-element segments, 64 B rows) and the MFMA operands are read TRANSPOSED
// (ds_read_b64_tr_b16, as conv_wgrad.hip).  The 4 x 14 accumulator tiles (16 channels x 16 j per (kernel row, j half)) are
// dealt 2 x 7 to each of the four waves; a workgroup walks `rows_per_wg` output rows and writes ONE fp32 partial
// [64][224]; stem_wgrad_reduce_kernel sums the partials in order and scatters them to the [64][3][7][7] gradient.
constexpr int GK = 64;                    // pixels per stage
constexpr int DZROW = SCO * 2 + 16;       // bytes per dz row in LDS (padded)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const uint16_t* __restrict__ xp, const uint16_t* __restrict__ dz,
                                                          float* __restrict__ part, int N, int H, int W, int Wp, int Ho, int Wo,
                                                          int rows_per_wg) {
  __shared__ __attribute__((aligned(16))) char dzl[GK * DZROW];             // 9 KiB
  __shared__ __attribute__((aligned(16))) uint16_t xl[SKY][GK][SSEG];       // 28 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int Hp = H + 2 * STOP;
  // wave (wa, wb): output channels 32 wa .. 32 wa + 31 (two 16-row tiles) x seven of the 14 (kernel row, j tile) column
  // tiles (tile id 7 wb + k = 2 ky + jt): 18 transposed reads per 14 MFMAs (one wave per channel tile took 30)
  const int wa = wave & 1, wb = wave >> 1;
  f32x4_t acc[2][SKY];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int k = 0; k < SKY; ++k) acc[ci][k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const long nrows = (long)N * Ho;
  const long r_begin = (long)blockIdx.x * rows_per_wg, r_end = r_begin + rows_per_wg < nrows ? r_begin + rows_per_wg : nrows;
  const int spr = (Wo + GK - 1) / GK;                       // stages per output row
  const int nstages = (int)(r_end - r_begin) * spr;
  // stage s = (row r_begin + s / spr, pixels (s % spr) * 64 ...): its global loads are issued one stage ahead, so they
  // fly under the previous stage's multiplies (a stage is ~40 MFMA-cycles of work per byte loaded: latency, not bandwidth)
  uint4 rd[2], rx[SKY];
  auto load_stage = [&](int st) {
    const long r = r_begin + st / spr;
    const int ox0 = (st % spr) * GK;
    const int n = (int)(r / Ho), oy = (int)(r % Ho);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + 256 * k, px = idx >> 3, c = idx & 7;
      rd[k] = (ox0 + px < Wo) ? *(const uint4*)(dz + ((r * Wo + ox0 + px) * SCO + c * 8)) : make_uint4(0, 0, 0, 0);
    }
    const int px = tid >> 2, c = tid & 3;
    const uint16_t* xs = xp + (((long)n * Hp + 2 * oy) * Wp) * 3 + 6 * (ox0 + px) + 8 * c;
#pragma unroll
    for (int ky = 0; ky < SKY; ++ky) rx[ky] = ld_seg(xs + (long)ky * Wp * 3);
  };
  if (nstages > 0) load_stage(0);
  for (int st = 0; st < nstages; ++st) {
    {
      __syncthreads();  // the previous stage has been multiplied
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, px = idx >> 3, c = idx & 7;
        *(uint4*)(dzl + px * DZROW + c * 16) = rd[k];
      }
      {
        const int px = tid >> 2, c = tid & 3;
#pragma unroll
        // 16-byte chunk c of pixel px sits at position c ^ 2 in rows 8-15, 24-31, ...: the two lane groups of a half wave
        // (pixel rows 8 apart) then read different bank halves
        for (int ky = 0; ky < SKY; ++ky) *(uint4*)(&xl[ky][px][(c ^ (((px >> 3) & 1) << 1)) * 8]) = rx[ky];
      }
      __syncthreads();
      if (st + 1 < nstages) load_stage(st + 1);
      // ---- multiply: two 32-pixel k-steps
#pragma unroll
      for (int ks = 0; ks < GK / 32; ++ks) {
        const int row = ks * 32 + 8 * g + q;  // (+ 4 for the second transposed read)
        bf16x8_t a[2];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          const char* pa = dzl + row * DZROW + (16 * (2 * wa + ci) + 4 * pp) * 2;
          const s16x4_t alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pa);
          const s16x4_t ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + 4 * DZROW));
          a[ci] = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int k = 0; k < SKY; ++k) {
          const int t = 7 * wb + k, ky = t >> 1, jt = t & 1;
          const uint16_t* pb = &xl[ky][row][16 * (jt ^ (g & 1)) + 4 * pp];  // ((row >> 3) & 1) == (g & 1)
          const s16x4_t blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)pb);
          const s16x4_t bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + 4 * SSEG));
          const bf16x8_t b = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
          for (int ci = 0; ci < 2; ++ci) acc[ci][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ci], b, acc[ci][k], 0, 0, 0);
        }
      }
    }
  }
  // D[co = 32 wa + 16 ci + 4 g + r][column tile 7 wb + k][j = lane % 16]
  float* dst = part + (long)blockIdx.x * SCO * SKY * SSEG;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int k = 0; k < SKY; ++k) {
      const int t = 7 * wb + k, ky = t >> 1, jt = t & 1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        dst[((32 * wa + 16 * ci + 4 * g + rr) * SKY + ky) * SSEG + 16 * jt + i16] = acc[ci][k][rr];
    }
}

// dw[co][c][ky][kx] (element strides given) = sum_k part[k][co][ky][3 (kx + 1) + c].  256 threads = 32 outputs x 8 slices of
// the partials (k = kk, kk + 8, ...), folded through LDS in slice order: deterministic.
__global__ void stem_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dw, long so, long sc,
                                         long sh, long sw) {
  __shared__ float red[8][32];
  const int ii = threadIdx.x & 31, kk = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + ii;  // SCO * SKY * SSEG is a multiple of 32
  float a0 = 0.f, a1 = 0.f;
  int k = kk;
  for (; k + 8 < nparts; k += 16) {
    a0 += part[(long)k * SCO * SKY * SSEG + i];
    a1 += part[(long)(k + 8) * SCO * SKY * SSEG + i];
  }
  if (k < nparts) a0 += part[(long)k * SCO * SKY * SSEG + i];
  red[kk][ii] = a0 + a1;
  __syncthreads();
  if (kk == 0) {
    const int j = i % SSEG, ky = (i / SSEG) % SKY, co = i / (SSEG * SKY);
    if (j >= 3 && j < 24) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q][ii];
      dw[co * so + (j % 3) * sc + ky * sh + (j / 3 - 1) * sw] = t;
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
  const bool nhwc4 = x_dtype == SIHL_F32 && xsc == 1 && xsw == 3 && (W % 4) == 0 && (xsh % 4) == 0 && (xsn % 4) == 0 &&
                     ((uintptr_t)x % 16) == 0;
  if (vec4)
    hipLaunchKernelGGL((stem_pack_image_kernel<float, 1>), dim3((unsigned)gp), dim3(256), 0, stream, (const float*)x, xsn, xsc, xsh,
                       xsw, (uint16_t*)xp, N, H, W, Wp);
  else if (nhwc4)
    hipLaunchKernelGGL((stem_pack_image_kernel<float, 2>), dim3((unsigned)gp), dim3(256), 0, stream, (const float*)x, xsn, xsc, xsh,
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

// workgroups (= fp32 partials of 64 * 7 * 32 floats each) of the weight gradient
int sihl_stem_wgrad_parts(int N, int H) {
  const long nrows = (long)N * (H / 2);
  long rpw = (nrows + 511) / 512;  // (1 024 workgroups: the kernel 115 -> 103 us, the reduction 12 -> 22: no gain)
  if (rpw < 1) rpw = 1;
  return (int)((nrows + rpw - 1) / rpw);
}

// dw[64][3][7][7] (fp32, element strides given) = weight gradient of the stem conv for dz[N][H/2][W/2][64] (bf16), from the
// packed image xp that sihl_stem_conv_fwd wrote.  ws: sihl_stem_wgrad_parts(N, H) * 64 * 7 * 32 floats.
int sihl_stem_conv_wgrad(const void* xp, const void* dz, float* dw, long wso, long wsc, long wsh, long wsw, float* ws, int N,
                         int H, int W, hipStream_t stream) {
  if (!xp || !dz || !dw || !ws || N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return SIHL_EARG;
  const int Wp = stem_wp(W), Ho = H / 2, Wo = W / 2;
  const long nrows = (long)N * Ho;
  const int parts = sihl_stem_wgrad_parts(N, H);
  const int rpw = (int)((nrows + parts - 1) / parts);
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(parts), dim3(256), 0, stream, (const uint16_t*)xp, (const uint16_t*)dz, ws, N, H, W, Wp,
                     Ho, Wo, rpw);
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(SCO * SKY * SSEG / 32), dim3(256), 0, stream, (const float*)ws, parts,
                     dw, wso, wsc, wsh, wsw);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
