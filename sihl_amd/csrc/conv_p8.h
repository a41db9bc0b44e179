// Persistent 256x256 implicit-GEMM convolution for the big bf16 layers (gfx950): one 8-wave workgroup per CU walks
// its output tiles; the K loop is ONE continuous stream of 64-deep K-tiles that runs across tile boundaries, so the
// LDS-DMA prefetch of tile i+1 is already in flight while tile i is finished and stored.
//
// Same operation as conv_igemm_dma_kernel (ConvNormAct of src/sihl/layers/convblocks.py:37-87, its input gradient,
// nn.Linear of torchvision.ops.MLP: heads/object_detection.py:51-61):
//   out[m][co] = epilogue( sum_{ky,kx,ci} in[n, oy*s-p+ky*d, ox*s-p+kx*d, ci] * wt[co][ky][kx][ci] ),  m = (n*Ho+oy)*Wo+ox
//
// Structure (the "8-phase" schedule: two K-tiles = eight phases per loop iteration):
//  * K-tile = (one tap) x 64 input channels = four 16 KiB half-tiles in LDS: A0/A1 = pixel rows 0-127 / 128-255,
//    B0/B1 = output channels 0-127 / 128-255, 128-byte rows, two K-tile buffers (128 KiB).  Half-tiles arrive by
//    LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, XOR swizzle on the SOURCE address), ONE
//    half-tile per phase, six phases ahead of the phase that first reads it; the only wait in the loop is a counted
//    vmcnt(4) once per K-tile, so loads stay in flight across barriers and across output tiles.
//  * A wave owns 2 x 32 pixels (32 of each pixel half) x 2 x 64 channels (64 of each channel half) and multiplies ONE
//    quadrant per phase: 16 x v_mfma_f32_16x16x32_bf16.  Quadrant order (C0,P0) (C0,P1) (C1,P1) (C1,P0) with the P0
//    fragments kept in registers: phases read 12 / 4 / 8 / 0 x ds_read_b128, and each half-tile slot is read in
//    exactly one phase (A0,B0: 0; A1: 1; B1: 2), which is what lets it be restaged two phases later.
//  * Barriers: ONE per phase (every wave reads its fragments, multiplies, and meets the others at the end of the
//    phase; the two waves of a SIMD drift apart between barriers and cover each other's LDS and DMA-issue time), and
//    the phase's half-tile DMA is issued behind the first eight MFMAs.  Measured on L3 3x3 (profiles/
//    r02_p8_ablation.txt): the template's two-barrier schedule with waves 4-7 staggered one barrier behind waves 0-3 -
//    kept under tuning bits 32 / 256 - is 10 % slower here; in-kernel stamps put 35-40 % of a K-tile into barrier
//    waits and 20 % into issuing the eight LDS-DMA instructions per wave (~200 cycles each), not into waiting for data.
//  * Operands are swapped - weights are the MFMA's A (row) operand, pixels the B (column) operand - so a lane holds
//    FOUR CONSECUTIVE CHANNELS of one pixel per accumulator tile.  The epilogue packs them to bf16 (8 bytes), transposes
//    through a wave-PRIVATE 2 KiB LDS scratch (no workgroup barrier) and stores whole 128-byte pixel rows with 16-byte
//    stores; BatchNorm batch statistics are reduced over the 16 pixel lanes with DPP row rotations, written as per-wave
//    partial rows to LDS and summed in fixed order (deterministic) one phase later by all threads.
#pragma once
#include <type_traits>
#include "common.h"
#include "conv_params.h"
#include "dma.h"

namespace {

constexpr int P8_BM = 256, P8_BN = 256;
constexpr int P8_HALF = 16384, P8_KTILE = 65536;
constexpr int P8_SCR = 4096;                       // per wave: 2 KiB transpose + 2 KiB statistics partials
constexpr int P8_LDS = 2 * P8_KTILE + 8 * P8_SCR;  // 163 840 B = all of a CU's LDS
constexpr int P8_THREADS = 512;

// In-kernel cycle stamps (SIHL_TUNING builds, debug bit 64 only): shares of the loop per segment, per wave.
#ifdef SIHL_TUNING
#define P8_STAMP(var)                                                                       \
  do {                                                                                      \
    if (dbg & 64) {                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      unsigned long long t__;                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");           \
      __builtin_amdgcn_sched_barrier(0);                                                    \
      var += t__ - t_last;                                                                  \
      t_last = t__;                                                                         \
    }                                                                                       \
  } while (0)
#else
#define P8_STAMP(var) do {} while (0)
#endif

#define P8_BAR()                               \
  do {                                         \
    asm volatile("s_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

template <int CTRL> __device__ __forceinline__ float p8_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float p8_row_sum(float v) {
  v += p8_dpp<0x128>(v);  // row_ror:8
  v += p8_dpp<0x124>(v);  // row_ror:4
  v += p8_dpp<0x122>(v);  // row_ror:2
  v += p8_dpp<0x121>(v);  // row_ror:1
  return v;
}

__device__ __forceinline__ unsigned p8_pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 f = {a, b};
  bf2 h = __builtin_convertvector(f, bf2);
  return __builtin_bit_cast(unsigned, h);
}

template <int ACT> __device__ __forceinline__ float p8_act(float v) {
  if (ACT == SIHL_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == SIHL_ACT_SILU) return v / (1.f + expf(-v));
  if (ACT == SIHL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

// Epilogue of one wave's 64 pixels x 128 channels.  acc[ct][pt]: channel tile ct = hc*4 + t (channels n0 + hc*128 +
// wn*64 + t*16 + (lane>>4)*4 + r), pixel tile pt = h*2 + u (pixel m0 + h*128 + wm*32 + u*16 + (lane&15)).
// The activation is compile-time (a select chain over activations would make every element pay for an exp); bias,
// affines and the statistics mode are wave-uniform branches around whole 4-value groups.
template <int ACT, bool ADD>
__device__ __forceinline__ void p8_epilogue(const ConvParams& p, f32x4_t (&acc)[8][4], char* scr, int m0, int n0, int wm,
                                            int wn, int lane) {
  const int px = lane & 15, g = lane >> 4;
  float* st = (float*)(scr + 2048);  // [h][which][hc][64]
  bf16_t* __restrict__ out = (bf16_t*)p.out;
  const bool has_pre = p.pre_scale != nullptr, has_post = p.post_scale != nullptr, has_bias = p.bias != nullptr;
  const int stats = ADD ? 0 : p.stats_mode;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const int cob = n0 + hc * 128 + wn * 64;
      float ssum[4][4], ssq[4][4];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) ssum[b][c] = ssq[b][c] = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = h * 2 + u;
        const int mrow = m0 + h * 128 + wm * 32 + u * 16;  // first pixel of this 16-pixel tile
        const float valid = (mrow + px) < p.M ? 1.f : 0.f;
        uint4 addv[2];
        if (ADD) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int rr = (lane >> 3) + 8 * i, c = lane & 7;
            const int m = mrow + rr, co = cob + c * 8;
            addv[i] = (m < p.M && co < p.Cout) ? *(const uint4*)((const bf16_t*)p.add + (long)m * p.Cout + co)
                                               : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4_t v = acc[hc * 4 + t][pt];
          const int co0 = cob + t * 16 + g * 4;
          const bool cok = co0 < p.Cout;  // Cout % 4 == 0: the four channels are valid together
          if (has_bias) {
            const float4 b = cok ? *(const float4*)(p.bias + co0) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
          if (stats == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float m = v[r] * valid; ssum[t][r] += m; ssq[t][r] += m * m; }
          }
          if (has_pre) {
            const float4 s = cok ? *(const float4*)(p.pre_scale + co0) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sh = (cok && p.pre_shift) ? *(const float4*)(p.pre_shift + co0) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] = v[0] * s.x + sh.x; v[1] = v[1] * s.y + sh.y; v[2] = v[2] * s.z + sh.z; v[3] = v[3] * s.w + sh.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = p8_act<ACT>(v[r]);
          if (stats == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float m = v[r] * valid; ssum[t][r] += m; ssq[t][r] += m * m; }
          }
          if (has_post) {
            const float4 s = cok ? *(const float4*)(p.post_scale + co0) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 sh = (cok && p.post_shift) ? *(const float4*)(p.post_shift + co0) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0] = v[0] * s.x + sh.x; v[1] = v[1] * s.y + sh.y; v[2] = v[2] * s.z + sh.z; v[3] = v[3] * s.w + sh.w;
          }
          const int q = t * 4 + g;  // 8-byte chunk of the 128-byte row
          *(uint2*)(scr + px * 128 + (((q >> 1) ^ (px & 7)) << 4) + (q & 1) * 8) =
              make_uint2(p8_pack2(v[0], v[1]), p8_pack2(v[2], v[3]));
        }
        // rows back: lane = (row rr, 16-byte chunk c); LDS operations of one wave execute in order, so nothing is
        // needed between the writes above, these reads and the next block's writes
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int rr = (lane >> 3) + 8 * i, c = lane & 7;
          uint4 val = *(const uint4*)(scr + rr * 128 + ((c ^ (rr & 7)) << 4));
          const int m = mrow + rr, co = cob + c * 8;
          if (ADD) {
            float a[8], b[8];
            unpack16(val, a, bf16_t());
            unpack16(addv[i], b, bf16_t());
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
            val = pack16(a, bf16_t());
          }
          if (m < p.M && co < p.Cout) *(uint4*)(out + (long)m * p.Cout + co) = val;
        }
      }
      if (stats) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float4 s, q;
          s.x = p8_row_sum(ssum[t][0]); s.y = p8_row_sum(ssum[t][1]);
          s.z = p8_row_sum(ssum[t][2]); s.w = p8_row_sum(ssum[t][3]);
          q.x = p8_row_sum(ssq[t][0]); q.y = p8_row_sum(ssq[t][1]);
          q.z = p8_row_sum(ssq[t][2]); q.w = p8_row_sum(ssq[t][3]);
          if (px == 0) {
            *(float4*)(st + ((h * 2 + 0) * 2 + hc) * 64 + t * 16 + g * 4) = s;
            *(float4*)(st + ((h * 2 + 1) * 2 + hc) * 64 + t * 16 + g * 4) = q;
          }
        }
      }
    }
  }
  if (stats) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // partial rows are in LDS before the next barrier
}

// Statistics of one finished tile: every thread sums the four pixel-quarter partials of one channel (fixed order) for
// one 128-pixel half and writes the two partial rows (sum, sum of squares).
__device__ __forceinline__ void p8_stats_flush(const ConvParams& p, const char* smem, int tile_m, int n0, int tid) {
  const int co = tid & 255, h = tid >> 8;
  const int hc = co >> 7, wn = (co >> 6) & 1, c64 = co & 63;
  const int nrows = (p.M + 127) / 128, srow = tile_m * 2 + h;
  if (n0 + co >= p.Cout || srow >= nrows) return;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float s = 0.f;
#pragma unroll
    for (int wm = 0; wm < 4; ++wm) {
      const float* st = (const float*)(smem + 2 * P8_KTILE + (wn * 4 + wm) * P8_SCR + 2048);
      s += st[((h * 2 + which) * 2 + hc) * 64 + c64];
    }
    p.stats[((long)srow * 2 + which) * p.Cout + n0 + co] = s;
  }
}

template <bool ADD>
__global__ __launch_bounds__(P8_THREADS) void conv_p8_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;  // wn doubles as the stagger group: waves w and w+4 share a SIMD
  const int G = gridDim.x;
  const int wgl = xcd_remap(blockIdx.x, G);  // an XCD's workgroups take neighbouring tiles
  const int ntiles = p.gridM * p.gridN;
  const int nmy = (ntiles - wgl + G - 1) / G;  // >= 1: the grid never exceeds the tile count
  const int ntaps = p.KH * p.KW, nchunks = p.Cin / 64;
  const int KT = ntaps * nchunks;
  const int total = nmy * KT;  // K-tiles in this workgroup's stream

  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * p.H * p.W * p.Cin * 2L));
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * p.w_ntaps * p.Cin * 2L));
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;

  // ------------------------------------------------------------------------------------------------ issuer state
  // A half-tile (16 KiB) is 16 pieces of 1 KiB = 8 rows x 128 B; wave w moves pieces 2w and 2w+1 of every half-tile.
  // Slot i = half*2 + j: row (half*128 +) (2w+j)*8 + (lane>>3); LDS position lane&7 of a row holds the global 16-byte
  // chunk (lane&7) ^ ((row>>1)&7) = (lane&7) ^ ((lane>>4) + 4j).  Pixel slots keep an offset and a tap mask (image
  // borders); rows beyond M or Cout need nothing: their offsets lie beyond the buffer (zeros) or, for a windowed conv,
  // in its last image (finite garbage in rows that are never stored and are masked out of the statistics).
  unsigned a_off[4], a_mask[2];  // a_mask[half]: 16 tap bits of slot j in bits 16j .. 16j+15
  unsigned b_off0;      // slot 0 of the weight rows; the other three differ by wave-uniform strides and ...
  int b_swd;            // ... +-64 bytes: the swizzled chunk of the j = 1 rows relative to the j = 0 rows
  int is_k = 0, is_kc = 0, is_tap = -1, is_ky = 0, is_kx = -1, is_buf = 1;
  bool is_more = true;
  unsigned n_adelta = 0, n_bdelta = 0, n_tapbit = 1;
  const bool pointwise = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
  const int hw_o = p.Ho * p.Wo;
  const unsigned b_rowbytes = (unsigned)(p.w_ntaps * p.Cin * 2);
  // K-tile to K-tile steps of the two scalar deltas (tap fastest, then channel chunk)
  const int a_sx = p.dil * p.Cin * 2, a_sy = (p.dil * p.W - (p.KW - 1) * p.dil) * p.Cin * 2;
  const int a_wrap = 128 - ((p.KH - 1) * p.dil * p.W + (p.KW - 1) * p.dil) * p.Cin * 2;
  const int b_sx = p.w_kxs * p.Cin * 2, b_sy = (p.w_kys * p.w_kw - (p.KW - 1) * p.w_kxs) * p.Cin * 2;
  const int b_wrap = 128 - ((p.KH - 1) * p.w_kys * p.w_kw + (p.KW - 1) * p.w_kxs) * p.Cin * 2;
  const unsigned b_d0 = (unsigned)((p.w_ky0 * p.w_kw + p.w_kx0) * p.Cin * 2);
  auto load_tile = [&](int k) {
    const int L = wgl + k * G;
    const int tm = L / p.gridN, tn = L - tm * p.gridN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (i >> 1) * 128 + (2 * wave + (i & 1)) * 8 + (lane >> 3);
      const int ch = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
      const int m = tm * P8_BM + row;
      unsigned off, mask = 1u;
      if (pointwise) {
        off = (unsigned)(((long)m * p.Cin + ch) * 2L);
      } else {
        const int n = m / hw_o, r = m - n * hw_o;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        mask = 0;
        int t = 0;
        for (int ky = 0; ky < p.KH; ++ky)
          for (int kx = 0; kx < p.KW; ++kx, ++t) {
            const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
            if (iy >= 0 && ix >= 0 && iy < p.H && ix < p.W) mask |= 1u << t;
          }
        off = (unsigned)((((long)n * p.H * p.W + (long)iy0 * p.W + ix0) * p.Cin + ch) * 2L);
      }
      a_off[i] = off;
      if ((i & 1) == 0) a_mask[i >> 1] = mask; else a_mask[i >> 1] |= mask << 16;
    }
    const int row0 = 2 * wave * 8 + (lane >> 3), sw0 = lane >> 4;
    const int ch0 = ((lane & 7) ^ sw0) * 8, ch1 = ((lane & 7) ^ (sw0 + 4)) * 8;
    b_off0 = (unsigned)(((long)(tn * P8_BN + row0) * p.w_ntaps * p.Cin + ch0) * 2L);
    b_swd = (ch1 - ch0) * 2;
  };
  auto advance = [&]() {  // next K-tile of the stream
    is_buf ^= 1;
    if (++is_tap == ntaps) {
      is_tap = 0; is_ky = 0; is_kx = 0;
      n_adelta += (unsigned)a_wrap; n_bdelta += (unsigned)b_wrap;
      if (++is_kc == nchunks) {
        is_kc = 0;
        n_adelta = 0; n_bdelta = b_d0;
        if (++is_k >= nmy) { is_more = false; return; }
        load_tile(is_k);
      }
    } else if (++is_kx == p.KW) {
      is_kx = 0; ++is_ky;
      n_adelta += (unsigned)a_sy; n_bdelta += (unsigned)b_sy;
    } else {
      n_adelta += (unsigned)a_sx; n_bdelta += (unsigned)b_sx;
    }
    n_tapbit = 1u << is_tap;
  };
  // half-tile hq of the K-tile the issuer stands on: 0 = A0, 1 = B0, 2 = A1, 3 = B1 (LDS order inside a K-tile buffer)
  const int dbg = SIHL_DBG(p) ^ (32 | 256);  // shipped schedule: one barrier per phase, DMA issued behind the first MFMAs  // tuning ablations (results invalid): 1 no DMA in the loop, 2 no MFMA, 8 no stagger,
                          // 16 no epilogue; 32 (a valid schedule): one barrier per phase, no stagger
  bool in_loop = false;
  unsigned long long t_last = 0, t_read = 0, t_issue = 0, t_bar = 0, t_mma = 0, t_wait = 0, t_epi = 0;
  (void)t_last; (void)t_read; (void)t_issue; (void)t_bar; (void)t_mma; (void)t_wait; (void)t_epi;
  auto issue = [&](int hq) {
    if (hq == 0) advance();
    if (!is_more) return;
    if ((dbg & 1) && in_loop) return;
    const unsigned dst = lds_base + is_buf * P8_KTILE + hq * P8_HALF + 2 * wave * 1024;
    if ((hq & 1) == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned off = hq == 0 ? a_off[j] : a_off[2 + j], mask = hq == 0 ? a_mask[0] : a_mask[1];
        dma16((mask & (n_tapbit << (16 * j))) ? off + n_adelta : OOB, dst + j * 1024, in_rsrc);
      }
    } else {
      const unsigned base = b_off0 + n_bdelta + (hq == 3 ? 128u * b_rowbytes : 0u);
      dma16(base, dst, wt_rsrc);
      dma16(base + 8u * b_rowbytes + (unsigned)b_swd, dst + 1024, wt_rsrc);
    }
  };

  // ------------------------------------------------------------------------------------------------ consumer state
  f32x4_t acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 fc[4][2], fp0[2][2], fp1[2][2];  // fragments: channel tiles of the current channel half, pixel tiles of each half
  const int frow = lane & 15, fsw = (frow >> 1) & 7, fg = lane >> 4;
  const int loff0 = frow * 128 + ((fg ^ fsw) << 4), loff1 = frow * 128 + (((4 + fg) ^ fsw) << 4);
  const char* cbase = smem + P8_HALF + wn * 64 * 128;  // + buf*KTILE + hc*2*HALF + t*2048 + loff
  const char* pbase = smem + wm * 32 * 128;            // + buf*KTILE + h*2*HALF + u*2048 + loff
  char* scr = smem + 2 * P8_KTILE + wave * P8_SCR;

  int c_k = 0, c_kin = 0;  // consumer: ordinal of the current tile in this workgroup's list, K-tile inside it
  bool flush_pending = false;
  int flush_tm = 0, flush_n0 = 0;

  auto mma_quad = [&](auto hc_tag, auto h_tag, int hq_inside) {
    constexpr int HC = decltype(hc_tag)::value, HH = decltype(h_tag)::value;
    if (dbg & 2) { if (hq_inside >= 0) issue(hq_inside); return; }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4& b = HH == 0 ? fp0[u][ks] : fp1[u][ks];
          acc[HC * 4 + t][HH * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, fc[t][ks]), __builtin_bit_cast(bf16x8_t, b), acc[HC * 4 + t][HH * 2 + u], 0, 0, 0);
        }
      if (ks == 0 && hq_inside >= 0) {  // tuning bit 256: the half-tile's DMA goes out behind the first 8 MFMAs
        __builtin_amdgcn_sched_barrier(0);
        issue(hq_inside);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  const bool dma_inside = (dbg & 256) != 0;
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;

  // ------------------------------------------------------------------------------------------------ prologue
  load_tile(0);
  n_adelta = (unsigned)(-a_sx);  // the first advance() steps onto tap (0, 0) of chunk 0
  n_bdelta = b_d0 - (unsigned)b_sx;
  for (int q = 0; q < 6; ++q) issue(q & 3);  // K-tile 0 and the first two half-tiles of K-tile 1
  if (is_more) wait_vm_keep<4>(); else wait_vm_keep<0>();
  P8_BAR();
  if (wn == 1 && !(dbg & 40)) P8_BAR();  // stagger: waves 4-7 run one barrier behind waves 0-3
  in_loop = true;
  if ((dbg & 128) && wn == 1) __builtin_amdgcn_s_setprio(1);  // tuning bit 128: the younger waves of each SIMD

  P8_STAMP(t_epi);
  t_epi = 0;
  for (int g = 0; g < total; ++g) {
    // one K-tile = four phases, fragments from K-tile buffer g & 1
    const char* cb = cbase + (g & 1) * P8_KTILE;
    const char* pb = pbase + (g & 1) * P8_KTILE;
    // phase 0: (C0, P0)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fc[t][0] = *(const uint4*)(cb + t * 2048 + loff0);
      fc[t][1] = *(const uint4*)(cb + t * 2048 + loff1);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fp0[u][0] = *(const uint4*)(pb + u * 2048 + loff0);
      fp0[u][1] = *(const uint4*)(pb + u * 2048 + loff1);
    }
    P8_STAMP(t_read);
    if (!dma_inside) issue(2);
    P8_STAMP(t_issue);
    if (!(dbg & 32)) P8_BAR();
    P8_STAMP(t_bar);
    mma_quad(T0(), T0(), dma_inside ? 2 : -1);
    P8_STAMP(t_mma);
    P8_BAR();
    P8_STAMP(t_bar);
    // phase 1: (C0, P1).  Every wave's partial statistics of the previous tile are in LDS by now (two barriers ago).
    if (flush_pending) {
      p8_stats_flush(p, smem, flush_tm, flush_n0, tid);
      flush_pending = false;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      fp1[u][0] = *(const uint4*)(pb + 2 * P8_HALF + u * 2048 + loff0);
      fp1[u][1] = *(const uint4*)(pb + 2 * P8_HALF + u * 2048 + loff1);
    }
    P8_STAMP(t_read);
    if (!dma_inside) issue(3);
    P8_STAMP(t_issue);
    if (!(dbg & 32)) P8_BAR();
    P8_STAMP(t_bar);
    mma_quad(T0(), T1(), dma_inside ? 3 : -1);
    P8_STAMP(t_mma);
    P8_BAR();
    P8_STAMP(t_bar);
    // phase 2: (C1, P1)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fc[t][0] = *(const uint4*)(cb + 2 * P8_HALF + t * 2048 + loff0);
      fc[t][1] = *(const uint4*)(cb + 2 * P8_HALF + t * 2048 + loff1);
    }
    P8_STAMP(t_read);
    if (!dma_inside) issue(0);
    P8_STAMP(t_issue);
    if (!(dbg & 32)) P8_BAR();
    P8_STAMP(t_bar);
    mma_quad(T1(), T1(), dma_inside ? 0 : -1);
    P8_STAMP(t_mma);
    P8_BAR();
    P8_STAMP(t_bar);
    // phase 3: (C1, P0); the next K-tile must have landed - only the two half-tiles issued after it stay in flight
    issue(1);
    P8_STAMP(t_issue);
    if (is_more) wait_vm_keep<4>(); else wait_vm_keep<0>();
    P8_STAMP(t_wait);
    if (!(dbg & 32)) P8_BAR();
    P8_STAMP(t_bar);
    mma_quad(T1(), T0(), -1);
    P8_STAMP(t_mma);
    if (++c_kin == KT) {
      c_kin = 0;
      const int L = wgl + c_k * G;
      const int tm = L / p.gridN, tn = L - tm * p.gridN;
      if (dbg & 16) {}
      else if (ADD) p8_epilogue<SIHL_ACT_NONE, true>(p, acc, scr, tm * P8_BM, tn * P8_BN, wm, wn, lane);
      else if (p.act == SIHL_ACT_RELU) p8_epilogue<SIHL_ACT_RELU, false>(p, acc, scr, tm * P8_BM, tn * P8_BN, wm, wn, lane);
      else if (p.act == SIHL_ACT_SILU) p8_epilogue<SIHL_ACT_SILU, false>(p, acc, scr, tm * P8_BM, tn * P8_BN, wm, wn, lane);
      else p8_epilogue<SIHL_ACT_NONE, false>(p, acc, scr, tm * P8_BM, tn * P8_BN, wm, wn, lane);
      if (!ADD && p.stats_mode) { flush_pending = true; flush_tm = tm; flush_n0 = tn * P8_BN; }
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      ++c_k;
      P8_STAMP(t_epi);
    }
    P8_BAR();
    P8_STAMP(t_bar);
  }
#ifdef SIHL_TUNING
  if ((dbg & 64) && p.partial && lane == 0) {
    unsigned long long* o = (unsigned long long*)p.partial + ((long)blockIdx.x * 8 + wave) * 8;
    o[0] = t_read; o[1] = t_issue; o[2] = t_bar; o[3] = t_mma; o[4] = t_wait; o[5] = t_epi; o[6] = (unsigned long long)total; o[7] = 0;
  }
#endif
  if (wn == 0 && !(dbg & 40)) P8_BAR();  // matches the last barrier of the staggered group
  if (flush_pending) p8_stats_flush(p, smem, flush_tm, flush_n0, tid);
}

}  // namespace
