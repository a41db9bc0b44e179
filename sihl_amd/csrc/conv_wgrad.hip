// Weight gradient of the NHWC convolution (and of nn.Linear = 1x1 over rows) on the matrix cores.
//
//   dW[co][ky][kx][ci] = sum_m dout[m][co] * in[n, oy*s-p+ky*d, ox*s-p+kx*d, ci]       m = (n*Ho+oy)*Wo+ox
//
// This is the reduction autograd performs for Conv2d.weight / Linear.weight behind the reference's
// ConvNormAct / Conv2dNormActivation / ops.MLP (see conv_igemm.hip for the forward citations).
// GEMM view per tap: [Cout x M] . [M x Cin]; the contraction runs over PIXELS, which in NHWC is
// the slow axis of both operands.  Both tiles are therefore staged as [pixel][channel] rows and the
// MFMA operands (k-contiguous per lane) come from TRANSPOSED LDS reads: ds_read_b64_tr_b16 for bf16,
// plain ds_read_b32 down a column for fp32 (v_mfma_f32_32x32x2_f32 takes one float per lane).
// Workgroup = 128 co x 128 ci of one tap over one K-split of the pixels; fp32 partial tiles go to
// a workspace [split][Cout][taps][Cin] and sihl_wgrad_reduce sums the splits (deterministic).
#include "common.h"
#include "dma.h"
#include "profile.h"

namespace {

struct WgradParams {
  const void* in;
  const void* dout;
  float* ws;  // [splits][Cout][taps][Cin]
  int N, H, W, Cin, Cout, KH, KW, stride, pad, dil, Ho, Wo;
  int M, splits, m_per_split;
  int tiles_co, tiles_ci;
  int prio;  // LDS-DMA kernel: progress-based wave priority (common.h SIHL_PRIO) - set when the launch aims at the whole chip
};

constexpr int BCO = 128, BCI = 128;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

template <typename T> struct wg_cfg;
template <> struct wg_cfg<bf16_t> {
  static constexpr int KP = 64;            // pixels per stage
  static constexpr int ROWB = BCO * 2 + 64;  // 4 consecutive pixel rows land on disjoint banks
};
template <> struct wg_cfg<float> {
  static constexpr int KP = 32;
  static constexpr int ROWB = BCO * 4;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int KP = wg_cfg<T>::KP, ROWB = wg_cfg<T>::ROWB;
  constexpr int CPR = BCO * (int)sizeof(T) / 16;  // 16-byte chunks per tile row (16 bf16 / 32 fp32)
  constexpr int RPT = KP * CPR / 256;             // rows per loader thread (=4)
  constexpr int RSTEP = 256 / CPR;
  constexpr int TILE = KP * ROWB, STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, each 64 co x 64 ci
  const int ntaps = p.KH * p.KW;
  int b = blockIdx.x;
  const int split = b % p.splits; b /= p.splits;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int tap = b;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int k_begin = split * p.m_per_split;
  const int k_end = min(p.M, k_begin + p.m_per_split);
  const int nstages = (k_end - k_begin + KP - 1) / KP;

  const T* __restrict__ in = (const T*)p.in;
  const T* __restrict__ dout = (const T*)p.dout;
  const int lc = tid % CPR, lr = tid / CPR;
  const bool co_ok = co0 + lc * VEC < p.Cout;
  const bool ci_ok = ci0 + lc * VEC < p.Cin;
  const int hw = p.Ho * p.Wo;

  uint4 ra[RPT], rb[RPT];
  auto load_regs = [&](int s) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int m = k_begin + s * KP + lr + RSTEP * i;
      const bool mok = m < k_end;
      ra[i] = (mok && co_ok) ? *(const uint4*)(dout + (long)m * p.Cout + co0 + lc * VEC) : make_uint4(0, 0, 0, 0);
      bool ok = mok && ci_ok;
      long off = 0;
      if (ok) {
        const int n = m / hw, r = m - n * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy = oy * p.stride - p.pad + ky * p.dil, ix = ox * p.stride - p.pad + kx * p.dil;
        ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        off = (((long)n * p.H + iy) * p.W + ix) * p.Cin + ci0 + lc * VEC;
      }
      rb[i] = ok ? *(const uint4*)(in + off) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE + lr * ROWB + lc * 16;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      *(uint4*)(base + i * RSTEP * ROWB) = ra[i];
      *(uint4*)(base + TILE + i * RSTEP * ROWB) = rb[i];
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const char* As = smem + buf * STAGE;
    const char* Bs = As + TILE;
    if constexpr (sizeof(T) == 2) {
      // transposed read: 16-lane group g reads a block of 4 pixel rows x 16 channels; lane 4q+p gives
      // the address of row q, channels 4p..4p+3 and receives channel (lane&15) of the 4 rows.
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g >> 1;
      const int row_off = (8 * h + q) * ROWB + (16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
      for (int ks = 0; ks < KP / 16; ++ks) {
        bf16x8_t a[2], bq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* pa = As + ks * 16 * ROWB + row_off + (wm * 64 + t * 32) * 2;
          const char* pb = Bs + ks * 16 * ROWB + row_off + (wn * 64 + t * 32) * 2;
          s16x4_t a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa));
          s16x4_t a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pa + 4 * ROWB));
          s16x4_t b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb));
          s16x4_t b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(pb + 4 * ROWB));
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          s16x8_t av = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
          s16x8_t bv = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
          a[t] = __builtin_bit_cast(bf16x8_t, av);
          bq[t] = __builtin_bit_cast(bf16x8_t, bv);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int i32 = lane & 31, h = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < KP / 2; ++kk) {
        float a[2], bq[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t] = *(const float*)(As + (2 * kk + h) * ROWB + (wm * 64 + t * 32 + i32) * 4);
          bq[t] = *(const float*)(Bs + (2 * kk + h) * ROWB + (wn * 64 + t * 32 + i32) * 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ONE LDS stage: the registers holding the next stage's loads are the second buffer.  40 KB instead of 80 KB lets
  // three workgroups share a CU and overlap each other's load / multiply phases (an extra barrier per stage buys it).
  if (nstages > 0) {
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const bool more = s + 1 < nstages;
      if (more) load_regs(s + 1);
      compute(0);
      __syncthreads();
      if (more) store_lds(0);
      __syncthreads();
    }
  }

  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          p.ws[(((long)split * p.Cout + co) * ntaps + tap) * p.Cin + ci] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 LDS-DMA variant (the default for >= 128-channel layers): one workgroup of 8 waves owns the whole
// 256 co x 256 ci panel of ONE tap over a K-split of the pixels.  Per stage 64 pixels of dout and of the
// tap-shifted input go global -> LDS by buffer_load ... lds (rows of 512 B, unpadded); the MFMA operands are
// k-contiguous per lane, so they are read TRANSPOSED with ds_read_b64_tr_b16.  A 16-lane group of that
// instruction touches 4 consecutive pixel rows x 64 B, which on 512-byte rows would all hit the same bank
// window: the 64-byte block index of a row is XOR-ed with (row & 3) on the DMA source side and on the read.
#ifdef SIHL_WGRAD_STAMPS
__device__ unsigned long long* g_wgrad_stamps = nullptr;
#define WG_T(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
constexpr int WB = 256;        // channels per panel side
constexpr int WKP = 64;        // pixels per stage
constexpr int WROW = WB * 2;   // bytes per pixel row in LDS

// NW waves: 8 (2 x 4, each 128 co x 64 ci) or 16 (4 x 4, each 64 co x 64 ci: four waves per SIMD cover each other's
// DMA-issue and barrier phases, as in the 256 x 256 forward tile; half the DMA pieces per wave).
template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_wgrad_dma_kernel(const WgradParams p) {
  constexpr int TILE = WKP * WROW, STAGE = 2 * TILE;  // 32 KiB per operand, 64 KiB per stage
  constexpr int NP = TILE / 1024 / NW;                // DMA pieces per wave per operand per stage (4 / 2)
  constexpr int MI = 32 / NW;                         // 32-row co tiles per wave (4 / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SIHL_WGRAD_STAMPS
  unsigned long long w_entry, w_loop_end = 0, w_rt0 = 0, w_rt1 = 0;
  WG_T(w_entry);
#endif
  const int wm = wave >> 2, wn = wave & 3;  // (NW / 4) x 4 waves, each MI * 32 co x 64 ci
  const int ntaps = p.KH * p.KW;
  // Block -> (group, tap) with the taps of one (co panel, ci panel, K-split) group on ONE XCD and adjacent in
  // dispatch order (blocks b and b+8 share an XCD): they stream the same pixels at about the same time, so
  // dout is fetched from the fabric once instead of KH*KW times and the shifted inputs mostly hit L2.
  const int ngroups = p.tiles_co * p.tiles_ci * p.splits;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int grp = (idx / ntaps) * 8 + xcd;
  const int tap = idx % ntaps;
  if (grp >= ngroups) return;
  int b = grp;
  const int split = b % p.splits; b /= p.splits;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int co0 = tco * WB, ci0 = tci * WB;
  const int k_begin = split * p.m_per_split;
  const int k_end = min(p.M, k_begin + p.m_per_split);
  const int nstages = (k_end - k_begin + WKP - 1) / WKP;

  const v4i_t dy_rsrc = make_rsrc(p.dout, (unsigned)((long)p.M * p.Cout * 2));
  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * p.H * p.W * p.Cin * 2));
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  const int hw = p.Ho * p.Wo;

  // slot j of this wave covers LDS rows (wave*NP + j)*2 + (lane >> 5); lane & 31 is the 16-byte position, which
  // holds source chunk  pos ^ ((row & 3) << 2).  Each slot walks the pixels k_begin + row, +64, +128, ...: its
  // (image, oy, ox) is advanced incrementally (no division inside the K loop).
  int s_ch[NP], s_m[NP], s_n[NP], s_oy[NP], s_ox[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int row = (wave * NP + j) * 2 + (lane >> 5);
    s_ch[j] = ((lane & 31) ^ ((row & 3) << 2)) * 8;  // channel offset inside the 256-wide panel
    s_m[j] = k_begin + row;
    const int mm = min(s_m[j], p.M - 1);
    s_n[j] = mm / hw;
    const int r = mm - s_n[j] * hw;
    s_oy[j] = r / p.Wo;
    s_ox[j] = r - s_oy[j] * p.Wo;
  }
  const int step_ox = WKP % p.Wo, step_rows = WKP / p.Wo;
  const int step_oy = step_rows % p.Ho, step_n = step_rows / p.Ho;
  auto issue_slot = [&](int j, unsigned abase, unsigned bbase) {
    const int m = s_m[j];
    const bool mok = m < k_end;
    const int co = co0 + s_ch[j], ci = ci0 + s_ch[j];
    dma16((mok && co < p.Cout) ? (unsigned)(m * p.Cout + co) * 2u : OOB, abase + j * 1024, dy_rsrc);
    const int iy = s_oy[j] * p.stride - p.pad + ky * p.dil, ix = s_ox[j] * p.stride - p.pad + kx * p.dil;
    const bool ok = mok && ci < p.Cin && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    dma16(ok ? (unsigned)(((s_n[j] * p.H + iy) * p.W + ix) * p.Cin + ci) * 2u : OOB, bbase + j * 1024, in_rsrc);
    // advance this slot by one stage (64 pixels)
    s_m[j] += WKP;
    s_ox[j] += step_ox;
    s_oy[j] += step_oy;
    s_n[j] += step_n;
    if (s_ox[j] >= p.Wo) { s_ox[j] -= p.Wo; s_oy[j] += 1; }
    if (s_oy[j] >= p.Ho) { s_oy[j] -= p.Ho; s_n[j] += 1; }
  };

  f32x16_t acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read geometry: group g = lane>>4 reads 4 pixel rows x 16 channels; lane 4q+pp gives row q,
  // channels 4pp..4pp+3 and receives channel (lane & 15) of the 4 rows.  (row & 3) == q for both reads.
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g >> 1;
  const int rowsel = 8 * h + q;                        // + 16*ks (+4 for the second read)
  const int colb = (16 * (g & 1) + 4 * pp) * 2;        // byte column inside a 32-channel MFMA tile
  auto tr_addr = [&](int row, int chan_base_bytes) {   // swizzled byte offset of (row, channel byte) in a tile
    const int cb = chan_base_bytes + colb;
    return row * WROW + ((((cb >> 6) ^ (row & 3)) << 6) | (cb & 63));
  };
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  // (a thin layer fills part of the 256 x 256 panel: waves whose 64 x 64 block lies outside issue their DMA pieces and keep the
  // barriers, nothing else)
  const bool wave_live = co0 + wm * (MI * 32) < p.Cout && ci0 + wn * 64 < p.Cin;
  auto compute = [&](int buf, bool more, unsigned abase) {
    const char* As = smem + buf * STAGE;
    const char* Bs = As + TILE;
    if (!wave_live) {
      if (more) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_slot(j, abase, abase + TILE);
      }
      return;
    }
    // operand fragments are double-buffered in registers: the transposed reads of k-step ks+1 are in flight while
    // k-step ks multiplies (reading and multiplying back to back left the matrix pipe idle for an LDS round trip
    // per k-step)
    bf16x8_t a[2][MI], bq[2][2];
    auto load_frags = [&](int ks, int slot) {
      const int row = ks * 16 + rowsel;
#pragma unroll
      for (int t = 0; t < MI; ++t) {
        const int off = tr_addr(row, (wm * (MI * 32) + t * 32) * 2);
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + off));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(As + off + 4 * WROW));
        a[slot][t] = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int off = tr_addr(row, (wn * 64 + t * 32) * 2);
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + off));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(Bs + off + 4 * WROW));
        bq[slot][t] = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int ks = 0; ks < WKP / 16; ++ks) {
      // issue priority falls as the wave advances through the stage (common.h SIHL_PRIO: 3 086 -> 2 890 cycles per stage) -
      // only when no other stream's kernels share the SIMDs
      if (p.prio) { if (ks == 0) SIHL_PRIO(3); else if (ks == 1) SIHL_PRIO(2); else if (ks == 2) SIHL_PRIO(1); else SIHL_PRIO(0); }
      if (ks + 1 < WKP / 16) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], bq[ks & 1][j], acc[i][j], 0, 0, 0);
      // the next stage's DMA goes out behind the first two k-steps' MFMAs (front-loaded, as in conv_igemm)
      if (more && ks < 2) {
#pragma unroll
        for (int j = ks; j < NP; j += 2) issue_slot(j, abase, abase + TILE);
      }
    }
  };

  if (nstages > 0) {
    {
      const unsigned abase = lds_base + wave * NP * 1024;
#pragma unroll
      for (int j = 0; j < NP; ++j) issue_slot(j, abase, abase + TILE);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SIHL_WGRAD_STAMPS
    unsigned long long w_t0, w_t1, w_t2, w_t3, w_comp = 0, w_wait = 0, w_bar = 0, w_start;
    WG_T(w_start);
    w_rt0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (int s = 0; s < nstages; ++s) {
      const bool more = s + 1 < nstages;
      const unsigned abase = lds_base + ((s + 1) & 1) * STAGE + wave * NP * 1024;
#ifdef SIHL_WGRAD_STAMPS
      WG_T(w_t0);
#endif
      compute(s & 1, more, abase);
#ifdef SIHL_WGRAD_STAMPS
      WG_T(w_t1);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SIHL_WGRAD_STAMPS
      WG_T(w_t2);
#endif
      __syncthreads();
#ifdef SIHL_WGRAD_STAMPS
      WG_T(w_t3);
      w_comp += w_t1 - w_t0; w_wait += w_t2 - w_t1; w_bar += w_t3 - w_t2;
#endif
    }
#ifdef SIHL_WGRAD_STAMPS
    if (g_wgrad_stamps && blockIdx.x == 0 && lane == 0) {
      unsigned long long* o = g_wgrad_stamps + wave * 5;
      o[0] = w_t3 - w_start; o[1] = w_comp; o[2] = w_wait; o[3] = w_bar; o[4] = (unsigned long long)nstages;
      g_wgrad_stamps[80 + wave * 4 + 0] = w_start - w_entry;
    }
    w_loop_end = w_t3;
    w_rt1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  }

  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          p.ws[(((long)split * p.Cout + co) * ntaps + tap) * p.Cin + ci] = acc[i][j][r];
      }
    }
#ifdef SIHL_WGRAD_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long w_end;
  WG_T(w_end);
  if (g_wgrad_stamps && blockIdx.x == 0 && lane == 0) {
    g_wgrad_stamps[80 + wave * 4 + 1] = w_end - w_loop_end;
    g_wgrad_stamps[80 + wave * 4 + 2] = w_rt1 - w_rt0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// bf16 ALL-TAPS variant for thin layers (Cin, Cout <= 128 with a KxK window, e.g. ResNet layer1/layer2 3x3 convs):
// one workgroup owns 64 co x 64 ci of EVERY tap over a K-split of the pixels.  Per stage of 32 pixels the dout tile
// is staged once and the input tile once per tap (the 9 shifted views overlap, so they come from L1/L2, not HBM);
// with one tap per workgroup (kernel above) a 64-channel layer streamed both activations 9 times and used a quarter
// of each 128x128 tile: 451 us for the 64->64 3x3 conv at 128^2 against a ~30 us HBM floor.
// Waves 2x2: each holds the 32 co x 32 ci block of all taps (9 x 16 accumulator registers).
constexpr int SCO = 64, SCI = 64, SKP = 32, SROW = SCO * 2 + 64, STILE = SKP * SROW, MAXTAPS = 9;

__global__ __launch_bounds__(256) void conv_wgrad_alltaps_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntaps = p.KH * p.KW;
  const int STAGE = (1 + ntaps) * STILE;
  int b = blockIdx.x;
  const int split = b % p.splits; b /= p.splits;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b;
  const int co0 = tco * SCO, ci0 = tci * SCI;
  const int k_begin = split * p.m_per_split;
  const int k_end = min(p.M, k_begin + p.m_per_split);
  const int nstages = (k_end - k_begin + SKP - 1) / SKP;

  const bf16_t* __restrict__ in = (const bf16_t*)p.in;
  const bf16_t* __restrict__ dout = (const bf16_t*)p.dout;
  const int lc = tid & 7, lr = tid >> 3;  // 16-byte chunk of the 64-channel row, pixel row of the stage
  const bool co_ok = co0 + lc * 8 < p.Cout;
  const bool ci_ok = ci0 + lc * 8 < p.Cin;
  const int hw = p.Ho * p.Wo;

  uint4 ra, rb[MAXTAPS];
  auto load_regs = [&](int s) {
    const int m = k_begin + s * SKP + lr;
    const bool mok = m < k_end;
    ra = (mok && co_ok) ? *(const uint4*)(dout + (long)m * p.Cout + co0 + lc * 8) : make_uint4(0, 0, 0, 0);
    const int n = m / hw, r = m - n * hw;
    const int oy = r / p.Wo, ox = r - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const long base = (long)n * p.H * p.W * p.Cin + ci0 + lc * 8;
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) {
      if (t < ntaps) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int iy = iy0 + ky * p.dil, ix = ix0 + kx * p.dil;
        const bool ok = mok && ci_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        rb[t] = ok ? *(const uint4*)(in + base + ((long)iy * p.W + ix) * p.Cin) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE + lr * SROW + lc * 16;
    *(uint4*)base = ra;
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t)
      if (t < ntaps) *(uint4*)(base + (1 + t) * STILE) = rb[t];
  };

  f32x16_t acc[MAXTAPS];
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposed fragment read (see conv_wgrad_kernel): lane 4q+pp of 16-lane group g addresses pixel row 8h+q,
  // channels 16(g&1)+4pp.. of a 32-channel block and receives one channel of 4 pixel rows
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g >> 1;
  const int row_off = (8 * h + q) * SROW + (16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto frag = [&](const char* ptr) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ptr));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ptr + 4 * SROW));
    s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto compute = [&](int buf) {
    const char* As = smem + buf * STAGE + row_off + wm * 32 * 2;
    const char* Bs = smem + buf * STAGE + STILE + row_off + wn * 32 * 2;
#pragma unroll
    for (int ks = 0; ks < SKP / 16; ++ks) {
      const bf16x8_t a = frag(As + ks * 16 * SROW);
#pragma unroll
      for (int t = 0; t < MAXTAPS; ++t)
        if (t < ntaps) {
          const bf16x8_t bq = frag(Bs + t * STILE + ks * 16 * SROW);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[t], 0, 0, 0);
        }
    }
  };

  // (a second register set prefetching two stages ahead was measured slower: 162 vs 145 us on the 64->64 3x3 layer)
  if (nstages > 0) {
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
  }

  const int half = lane >> 5;
  const int ci = ci0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < MAXTAPS; ++t)
    if (t < ntaps) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          p.ws[(((long)split * p.Cout + co) * ntaps + t) * p.Cin + ci] = acc[t][r];
      }
    }
}

// Small weight tensors (fewer than 256 workgroups in the kernel below): a workgroup = 16 weight vectors x 16 split
// groups - thread (v, g) sums the slabs k = g, g + 16, ... of vector v, the 16 groups are folded through LDS in a fixed
// order (deterministic).  A 64x64 weight with 1 024 slabs (16 MB to read) then runs on 64 workgroups instead of 4.
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n,
                                                                 int splits, int accumulate) {
  __shared__ float4 red[16][17];
  const int v = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = ((long)blockIdx.x * 16 + v) * 4;  // n % 4 == 0 (caller)
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n)
    for (int k = g; k < splits; k += 16) {
      const float4 t = *(const float4*)(ws + (long)k * n + i);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
  red[g][v] = a;
  __syncthreads();
  if (g == 0 && i < n) {
    float4 r = red[0][v];
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float4 t = red[q][v]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    if (accumulate) {
      const float4 o = *(const float4*)(dw + i);
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    *(float4*)(dw + i) = r;
  }
}

// dw[i] (+)= sum_k ws[k][i]: every thread owns 4 consecutive weights (16-byte loads) and keeps four split slabs in
// flight; the scalar version (one float per thread, one slab at a time) ran at ~1.6 TB/s over up to 64 MB of slabs.
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n, int splits,
                                    int accumulate) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n && (n & 3) == 0) {
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      const float4 v0 = *(const float4*)(ws + (long)k * n + i);
      const float4 v1 = *(const float4*)(ws + (long)(k + 1) * n + i);
      const float4 v2 = *(const float4*)(ws + (long)(k + 2) * n + i);
      const float4 v3 = *(const float4*)(ws + (long)(k + 3) * n + i);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; k < splits; ++k) {
      const float4 v0 = *(const float4*)(ws + (long)k * n + i);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    float4 r = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                           (a0.w + a1.w) + (a2.w + a3.w));
    if (accumulate) {
      const float4 o = *(const float4*)(dw + i);
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    *(float4*)(dw + i) = r;
  } else {
    for (long j = i; j < n && j < i + 4; ++j) {
      float t = 0.f;
      for (int k = 0; k < splits; ++k) t += ws[(long)k * n + j];
      dw[j] = accumulate ? dw[j] + t : t;
    }
  }
}

bool g_wgrad_force_reg = false;

bool use_dma(int Cin, int Cout, long in_bytes, long dy_bytes, int dtype) {
  // exactly 128x128 channels would leave three quarters of the 256x256 panel empty (228 us against 130 us for the
  // 128x128 register-staged kernel on ResNet layer2's 3x3)
  return !g_wgrad_force_reg && dtype == SIHL_BF16 && Cin >= 128 && Cout >= 128 && (Cin > 128 || Cout > 128) &&
         in_bytes < (1L << 31) && dy_bytes < (1L << 31);
}

// thin layers with a spatial window: all taps in one workgroup (bf16)
bool use_alltaps(int Cin, int Cout, int KH, int KW, int dtype) {
  return !g_wgrad_force_reg && dtype == SIHL_BF16 && Cin <= 64 && Cout <= 64 && KH * KW > 1 && KH * KW <= MAXTAPS;
}

int launch_alltaps(WgradParams p, hipStream_t stream) {
  const int LDS = 2 * (1 + p.KH * p.KW) * STILE;  // 120 KiB for 3x3
  static int attr_lds = 0;
  if (attr_lds < LDS) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_alltaps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_lds = LDS;
  }
  const int grid = p.tiles_co * p.tiles_ci * p.splits;
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout) * 2 + (double)p.Cout * p.KH * p.KW * p.Cin * 4.0;
  sihl_prof_begin(SIHL_PROF_WGRAD, SIHL_BF16, flops, bytes, stream);
  hipLaunchKernelGGL(conv_wgrad_alltaps_kernel, dim3(grid), dim3(256), LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// Tiling / split decision shared by the workspace query and the launcher.
struct WgradPlan { int mode /*0 reg, 1 dma, 2 all-taps*/, tiles_co, tiles_ci, splits, kp; };
int choose_splits(long M, int tiles, int kp, int target, long n_weights);
// target: K-split aim in workgroups, per call: target % 10000 for the LDS-DMA kernel (0 = 256, one per CU),
// target / 10000 for the register-staged / all-taps kernels (0 = default)
WgradPlan plan_wgrad(long M, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dtype, int target) {
  const int g_wgrad_target = target > 0 ? target % 10000 : 0, g_wgrad_target_reg = target > 0 ? target / 10000 : 0;
  const int vs = dtype == SIHL_BF16 ? 2 : 4;
  WgradPlan pl;
  const long n_weights = (long)Cout * KH * KW * Cin;
  if (use_alltaps(Cin, Cout, KH, KW, dtype)) {
    pl.mode = 2; pl.kp = SKP;
    pl.tiles_co = (Cout + SCO - 1) / SCO; pl.tiles_ci = (Cin + SCI - 1) / SCI;
    // one workgroup per CU (120 KiB of LDS): 256 workgroups, at least 8 stages each
    long s = (g_wgrad_target_reg > 0 ? g_wgrad_target_reg : 256) / (pl.tiles_co * pl.tiles_ci), by_work = M / (8L * SKP);
    if (s > by_work) s = by_work;
    if (s < 1) s = 1;
    pl.splits = (int)s;
    return pl;
  }
  // thin POINTWISE layers (ResNet layer1's 64 <-> 256 and 64 -> 64 convs, the 80-class logits: hundreds of MB of activations for
  // a few K weights) are bandwidth-bound: the panel kernel's LDS-DMA loader streams them faster than the register-staged 128 x 128
  // kernel although a quarter or less of its 256 x 256 tile is used (waves whose channel range is empty skip their fragment
  // reads and multiplies): 64 -> 256 95.9 -> 76.7 us (4.4 TB/s), 256 -> 64 96.9 -> 74.3, 64 -> 64 53.2 -> 42.9, 256 -> 80 on P3
  // 35.4 -> 29.1 (tools/wgrad_thin_probe.py, profiles/r04_wgrad_thin_probe.txt)
  // Only for launches that aim at the whole chip: beside the dgrad chain (aim 128) the register-staged kernel's 256 small
  // workgroups fit around the main stream's kernels better than 128 panel workgroups (in-step weight-gradient time 6.08 -> 6.19 ms,
  // profiles/r04_thin_lib_ab.txt).
  const bool thin_pw = !g_wgrad_force_reg && dtype == SIHL_BF16 && KH * KW == 1 && (Cin < 128 || Cout < 128) &&
                       (g_wgrad_target == 0 || g_wgrad_target >= 256) &&
                       (Cin < Cout ? Cin : Cout) >= 64 && (Cin % 8) == 0 && (Cout % 8) == 0 &&
                       (long)N * H * W * Cin * vs < (1L << 31) && M * Cout * vs < (1L << 31) && !getenv("SIHL_WGRAD_NO_THIN_DMA");
  const bool dma = thin_pw || use_dma(Cin, Cout, (long)N * H * W * Cin * vs, M * Cout * vs, dtype);
  const int tb = dma ? WB : BCO;
  pl.mode = dma ? 1 : 0;
  pl.kp = dtype == SIHL_BF16 ? 64 : 32;
  pl.tiles_co = (Cout + tb - 1) / tb; pl.tiles_ci = (Cin + tb - 1) / tb;
  pl.splits = choose_splits(M, KH * KW * pl.tiles_co * pl.tiles_ci, pl.kp, dma ? (g_wgrad_target > 0 ? g_wgrad_target : 256) : (g_wgrad_target_reg > 0 ? 2 * g_wgrad_target_reg : 512), n_weights);
  if (dma && KH * KW > 1) {
    // The LDS-DMA kernel keeps the KH*KW tap-workgroups of a (tile, split) group on ONE XCD, groups dealt round-robin over
    // the 8 XCDs, one workgroup per CU: an XCD given more than 32 workgroups runs a second, mostly empty round (3x3,
    // 256 -> 256 aimed at 256 workgroups: 28 groups = 36 workgroups on four of the XCDs, 253 us where one round of 24
    // groups takes 160).  Fewer splits, so that every XCD's share fits its 32 CUs at once.
    const int ntaps = KH * KW, tiles = pl.tiles_co * pl.tiles_ci;
    const long per_xcd = ((long)tiles * pl.splits + 7) / 8 * ntaps;
    const long one_round = 8L * (32 / ntaps) / tiles;  // splits that fill one round
    if (per_xcd > 32 && per_xcd <= 48 && one_round >= 1) pl.splits = (int)one_round;
  }
  return pl;
}

int launch_dma(WgradParams p, hipStream_t stream) {
  constexpr int LDS = 2 * 2 * WKP * WROW;  // 128 KiB
  static bool attr_set = false;
  static int nw = 16;  // SIHL_WGRAD_WAVES=8: the 8-wave form (A/B)
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    const char* env = getenv("SIHL_WGRAD_WAVES");
    if (env && atoi(env) == 8) nw = 8;
    attr_set = true;
  }
  const int ngroups = p.tiles_co * p.tiles_ci * p.splits;
  const int grid = ((ngroups + 7) / 8) * 8 * p.KH * p.KW;
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout) * 2 + (double)p.Cout * p.KH * p.KW * p.Cin * 4.0;
  sihl_prof_begin(SIHL_PROF_WGRAD, SIHL_BF16, flops, bytes, stream);
  if (nw == 16) hipLaunchKernelGGL(conv_wgrad_dma_kernel<16>, dim3(grid), dim3(1024), LDS, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_dma_kernel<8>, dim3(grid), dim3(512), LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

template <typename T>
int launch(WgradParams p, hipStream_t stream) {
  constexpr int LDS = 2 * wg_cfg<T>::KP * wg_cfg<T>::ROWB;  // one stage: dout tile + input tile
  static bool attr_set = false;
  auto kern = conv_wgrad_kernel<T>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = p.KH * p.KW * p.tiles_co * p.tiles_ci * p.splits;
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout) * sizeof(T) + (double)p.Cout * p.KH * p.KW * p.Cin * 4.0;
  sihl_prof_begin(SIHL_PROF_WGRAD, sizeof(T) == 2 ? SIHL_BF16 : SIHL_F32, flops, bytes, stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

int choose_splits(long M, int tiles, int kp, int target, long n_weights) {
  // aim for >= `target` workgroups, each with at least 4 stages of work; small weight tensors (few tiles) may
  // take many more K-splits - their fp32 partial slabs stay small - so that a streaming reduction over a huge
  // activation (ResNet layer1: M = 524288 rows for a 64x64 weight) still spreads over the whole chip
  long want = (target + tiles - 1) / tiles;
  long max_by_work = M / (4L * kp);
  if (max_by_work < 1) max_by_work = 1;
  long s = want < max_by_work ? want : max_by_work;
  long cap = (64L << 20) / (n_weights * 4);
  if (cap < 64) cap = 64;
  if (cap > 1024) cap = 1024;
  if (s < 1) s = 1;
  if (s > cap) s = cap;
  return (int)s;
}

}  // namespace

extern "C" {

#ifdef SIHL_WGRAD_STAMPS
int sihl_wgrad_stamps(void* buf) {  // diagnostic builds: device buffer of 16 x 5 u64 that workgroup 0 of the LDS-DMA kernel fills
  unsigned long long* b = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamps), &b, sizeof(b));
}
#endif

// Test hook: 1 = always use the register-staged 128x128 kernel (the fp32 / small-channel path).
int sihl_conv2d_wgrad_force_register_staging(int on) { g_wgrad_force_reg = on != 0; return 0; }

// Workspace bytes sihl_conv2d_wgrad needs for this problem.
long sihl_conv2d_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                int dil, int dtype, int target) {
  const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long M = (long)N * Ho * Wo;
  const int splits = plan_wgrad(M, N, H, W, Cin, Cout, KH, KW, dtype, target).splits;
  return (long)splits * Cout * KH * KW * Cin * (long)sizeof(float);
}

// dw: fp32 [Cout][KH][KW][Cin]; accumulate != 0 adds into dw instead of overwriting.  target: K-split aim of THIS call
// (0 = default; fewer, longer workgroups write fewer fp32 partial slabs and leave CUs to kernels of another stream) -
// the same value must be given to sihl_conv2d_wgrad_ws_bytes.
int sihl_conv2d_wgrad(const void* in, const void* dout, float* dw, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int dil, int dtype, int accumulate, int target, void* ws,
                      long ws_bytes, hipStream_t stream) {
  if (!in || !dout || !dw || !ws || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SIHL_EARG;
  const int vec = dtype == SIHL_BF16 ? 8 : 4;
  if (Cin % vec || Cout % vec) return SIHL_EARG;
  WgradParams p;
  p.in = in; p.dout = dout; p.ws = (float*)ws;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad = pad; p.dil = dil;
  p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  const long M = (long)N * p.Ho * p.Wo;
  if (p.Ho <= 0 || p.Wo <= 0 || M > (1L << 30)) return SIHL_EARG;
  p.M = (int)M;
  const WgradPlan pl = plan_wgrad(M, N, H, W, Cin, Cout, KH, KW, dtype, target);
  p.prio = (target <= 0 || target % 10000 == 0 || target % 10000 >= 256) ? 1 : 0;  // a smaller aim = beside another stream's kernels
  const bool dma = pl.mode == 1;
  const int kp = pl.kp;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.splits = pl.splits;
  p.m_per_split = (int)(((M + p.splits - 1) / p.splits + kp - 1) / kp * kp);
  const long n = (long)Cout * KH * KW * Cin;
  if (ws_bytes < (long)p.splits * n * (long)sizeof(float)) return SIHL_EWS;
  int rc;
  if (pl.mode == 2) rc = launch_alltaps(p, stream);
  else if (dma) rc = launch_dma(p, stream);
  else if (dtype == SIHL_F32) rc = launch<float>(p, stream);
  else if (dtype == SIHL_BF16) rc = launch<bf16_t>(p, stream);
  else return SIHL_EARG;
  if (rc) return rc;
  if ((n + 1023) / 1024 < 256 && (n & 3) == 0 && p.splits >= 16)
    hipLaunchKernelGGL(wgrad_reduce_small_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, stream,
                       (const float*)ws, dw, n, p.splits, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream,
                       (const float*)ws, dw, n, p.splits, accumulate);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // extern "C"
