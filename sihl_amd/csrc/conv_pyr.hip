// 3x3 / stride 1 / pad 1 convolution of the pyramid's TOP levels with its fusion node folded into the loader (bf16; square
// maps 16x16, 8x8, 4x4: BiFPN P5-P7 at 512^2; reference src/sihl/layers/bifpn.py:39-53 - `up_convs[k](up_fusions[k]([up(td),
// x]))` and `down_convs[k](down_fusions[k]([down(bu), x, td]))` - through convblocks.py:37-87, scalers.py:36-47,
// pooling.py:7-26).  Round 4; conv_small.hip stays as the parity-tested alternative (sihl_conv2d_small_enable(2)).
//
// Why: P5-P7 are 8 % of the north-star forward's flops and were 21 % of its time - per layer boundary a serial chain of
// 17 launches of 5-20 us (profiles/r03_ns_forward_final_summary.txt: fusion kernel -> conv -> split-K finish, seven times).
// The hardware guide's verdict on such chains (MI355X_MICROARCH.md, price list; cdna_hip_programming.md 5.6) and round 3's
// own in-launch split-K experiment agree that a kernel boundary is the cheapest coherence point on this chip, so the
// chain is not made persistent; it is made SHORT: 8 launches, each doing more.
//   * IMAGE-MAJOR tiles: a workgroup owns a whole map (one 16x16 or 8x8 image, four 4x4 images) and 32 output channels.
//     bs 32 x 256 channels = 256 workgroups at every level, no split-K and no finishing launch for P6 / P7, and - because
//     the workgroup sees the whole map - the spatial ops of the fusion nodes need no neighbour:
//   * FUSED PRODUCERS (MODE): the conv input is never read from HBM; the loader computes it into the LDS patch from the
//     fusion node's inputs - MODE 1: w0 * bilinear_x2(a) + w1 * b (FastNormalizedFusion(2) behind Interpolate, the
//     top-down path); MODE 2: w0 * (blur_s2(a) * a_scale + a_shift) + w1 * b + w2 * c (FastNormalizedFusion(3) behind the
//     AntialiasedDownscaler's blur, the bottom-up path; the optional per-channel affine is the training-mode BatchNorm of
//     the downscaler's conv block, as in blur_fuse_kernel).  Same fp32 arithmetic, same bf16 rounding of the merged value
//     as the stand-alone kernels (elementwise.hip), so the result is bit-identical to [fusion kernel -> conv].  Training
//     needs the merged tensor for the weight gradient: the workgroups of output slice 0 also store it (`merged`).
//   * 8 waves = 4 pixel groups x 2 K halves (the two waves of a SIMD split each stage's k-steps and hide each other's LDS
//     latency without reading anything twice); v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as first operand (a lane
//     holds 4 consecutive channels of a pixel); the two halves meet in the fp32 staging tiles of the epilogue.
//   * the input patch (map + one-pixel zero halo) is resident in LDS per 64-channel chunk (by LDS-DMA in MODE 0), weights
//     stream through a 4-slot ring of (kernel row x 64 channels x 32 out-channels = 12 KiB); 16-byte chunks XOR-swizzled by
//     2 * (patch column / 2) [+ 4 * patch row on 4x4 maps]: every ds_read_b128 lane group of the 16x16x32 operand shape
//     hits 16 distinct bank slots for all nine tap shifts (brute-forced over the family a * (col / 2) + b * row + c * (col & 1)).
#include "common.h"
#include "conv_params.h"
#include "conv_tuning.h"
#include "dma.h"
#include "profile.h"

namespace {

constexpr int PTHREADS = 512, PBN = 32, PKCB = 128;
constexpr int PW_CHUNK = 9 * PBN * PKCB;    // the weights of one 64-channel chunk: 9 taps x 32 out-channels x 128 B = 36 KiB
constexpr int PW_PIECES = PW_CHUNK / 1024;  // 36 LDS-DMA wave-instructions: waves 0-3 issue five, waves 4-7 four
constexpr int PST_STRIDE = PBN * 4 + 16;    // fp32 staging row of the epilogue
constexpr int PPAR_BYTES = 5 * PBN * 4;     // bias, pre-scale, pre-shift, post-scale, post-shift of the slice (fp32)
constexpr int PRED_BYTES = 16 * 2 * PBN * 4;  // column-sum partials of the statistics pass

struct PyrParams {
  const void* in;  // MODE 0: [N][W][W][Cin]
  const void* wt;  // [Cout][3][3][Cin]
  void* out;       // [N][W][W][Cout]
  const void* a;   // MODE 1: [N][W/2][W/2][Cin]; MODE 2: [N][2W][2W][Cin]
  const void* b;   // [N][W][W][Cin]
  const void* c;   // MODE 2: [N][W][W][Cin]
  const float* fw;       // raw fusion weights (softmax inside): 2 (MODE 1) or 3 (MODE 2)
  const float* a_scale;  // MODE 2, optional: per-channel affine of `a`
  const float* a_shift;
  void* merged;    // optional: the merged conv input [N][W][W][Cin], written by the workgroups of output slice 0
  const float* bias;
  const float* pre_scale;
  const float* pre_shift;
  const float* post_scale;
  const float* post_shift;
  float* stats;    // [rows][2][Cout]: one row per 128 pixels (16x16 maps) or per tile (8x8: per image; 4x4: per 4 images)
  // EMIT: the fusion node that CONSUMES this conv's output y, computed in the epilogue for the workgroup's 32-channel slice
  // (no recomputation: the slice of the whole map is here).  emit 1: e_out[N][2W][2W][Cout] = w0 * bilinear_x2(y) + w1 * e_b;
  // emit 2: e_out[N][W/2][W/2][Cout] = w0 * blur_s2(y) + w1 * e_b + w2 * e_c (e_b, e_c, e_out of that size).  y is the bf16
  // value the output tensor holds, so the result equals the stand-alone node kernel run on `out`.
  const void* e_b;
  const void* e_c;
  const float* e_fw;
  void* e_out;
  int emit, write_y;  // write_y == 0: `out` itself is not stored (its only consumer was the emitted node)
  int N, Cin, Cout, M, act, stats_mode, gridN, tiles, k_rotate;
  unsigned long long* stamps;  // -DSIHL_PYR_STAMPS diagnostic builds: (s_memtime, s_memrealtime) marks of workgroup 0
};

#ifdef SIHL_PYR_STAMPS
#define PYR_STAMP(i)                                                                                            \
  do {                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    unsigned long long t__, r__;                                                                                \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__), "=s"(r__)::"memory"); \
    if (tid == 0 && blockIdx.x == 0 && p.stamps) { p.stamps[2 * (i)] = t__; p.stamps[2 * (i) + 1] = r__; }       \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  } while (0)
#else
#define PYR_STAMP(i) do {} while (0)
#endif

template <int W> struct PyrGeo {
  static constexpr int HW = W * W;
  static constexpr int TM = W == 16 ? 256 : 64;     // pixels per tile
  static constexpr int G = TM / HW;                 // maps per tile
  static constexpr int PW = W + 2, PR = W + 2;      // patch columns / rows per map
  static constexpr int NP = G * PR * PW;            // patch pixels
  static constexpr int PIECES = (NP + 7) / 8;       // 1 KiB pieces (8 pixels x 128 B): 41 / 13 / 18
  static constexpr int APW = (PIECES + 7) / 8;      // pieces per wave (piece q belongs to wave q % 8)
  static constexpr int A_BYTES = PIECES * 1024;
  static constexpr int BUF = A_BYTES + PW_CHUNK;    // one chunk: patch + weights
  static constexpr int MTW = TM / 64;               // 16-pixel MFMA tiles per wave (4 pixel groups)
  static constexpr int ITEMS = TM * 8 / PTHREADS;   // (pixel, 16-byte piece) items per thread per chunk of a fused loader
  static constexpr int EPI = TM * 4 / PTHREADS > 0 ? TM * 4 / PTHREADS : 1;  // (pixel, 8 channels) items per thread
  static constexpr int ROWS = TM / 128 > 0 ? TM / 128 : 1;  // statistics rows per tile
  static constexpr int Y_OFF = (2 * TM * PST_STRIDE + 15) / 16 * 16;  // bf16 y tile of an emitting launch, behind the staging tiles
  static constexpr int PAR_OFF = 2 * BUF, RED_OFF = PAR_OFF + PPAR_BYTES;
  static constexpr int LDS = RED_OFF + PRED_BYTES;
  __device__ static __forceinline__ int swz(int pr, int pc) { return (2 * (pc >> 1) + (W == 4 ? 4 : 0) * pr) & 7; }
};

__device__ __forceinline__ void pyr_softmax(const float* raw, int n, float (&w)[3]) {  // as softmax_w of elementwise.hip
  w[0] = w[1] = w[2] = 0.f;
  float m = raw[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, raw[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) { w[i] = expf(raw[i] - m); s += w[i]; }
  for (int i = 0; i < n; ++i) w[i] /= s;
}
struct PLerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ PLerp pyr_up2_src(int dst, int in_size) {  // as up2_src of elementwise.hip
  float src = 0.5f * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  PLerp r;
  r.i0 = (int)src;
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.l1 = src - r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
__device__ __forceinline__ int pyr_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ __forceinline__ float pyr_act(float x, int act) {
  if (act == SIHL_ACT_RELU) return fmaxf(x, 0.f);
  if (act == SIHL_ACT_SILU) return x / (1.f + expf(-x));
  if (act == SIHL_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
  return x;
}

// K is walked in FOUR stages per 256 input channels - one per 64-channel chunk, all nine taps - with the whole chunk
// (patch 13-41 KiB + weights 36 KiB) double-buffered: 49-77 KiB in flight per CU while the previous chunk multiplies.  The
// first version of this kernel streamed the weights per kernel ROW through a 4-slot ring (12 stages, 36 KiB in flight): its
// in-kernel stamps (tools/pyr_stamps.py, profiles/r04_pyr_stamps_v1.txt) showed 1 240 cycles per stage on a 4x4 level that
// has 100 cycles of multiplies in it and 1 810 on P5 against 768 of multiplies - barrier + DMA round trip per stage (the
// slices' weights come from the Infinity Cache: each line is read by at most four workgroups per XCD), not bandwidth.
template <int W, int MODE>
__global__ __launch_bounds__(PTHREADS, 2) void conv_pyr_kernel(const PyrParams p) {
  using Geo = PyrGeo<W>;
  constexpr int MTW = Geo::MTW, ITEMS = Geo::ITEMS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  PYR_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave & 3, kh = wave >> 2;
  const int l16 = lane & 15, g = lane >> 4;
  // blocks b and b + 8 share an XCD (observed, speed only): with tile_m = b % tiles the 8 output slices of a map run on ONE
  // XCD - its input is pulled into one L2 - while every XCD streams all the weights (1.2 MB: L2-resident)
  const int tile_m = blockIdx.x % p.tiles, tile_n = blockIdx.x / p.tiles;
  const int n0 = tile_n * PBN, img0 = tile_m * Geo::G, m0 = tile_m * Geo::TM;
  const int nch = p.Cin >> 6;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * 9 * p.Cin * 2));
  // every workgroup walks the channel chunks from another start (see conv_small.hip / ConvParams::k_rotate)
  const int rc = p.k_rotate ? tile_m % nch : 0;

  // ---- weight pieces: piece j = wave + 8 u holds rows 8 j .. 8 j + 7 of the chunk's [tap][32 out-channels] x 128 B
  unsigned b_off[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int row = (wave + 8 * u) * 8 + (lane >> 3), pos = lane & 7;
    const int tap = row >> 5, co = row & 31;
    b_off[u] = (unsigned)((((long)(n0 + co) * 9 + tap) * p.Cin) * 2 + ((pos ^ ((co >> 1) & 7)) << 4));
  }
  auto issue_b = [&](int c, int buf) {
    const int cc = c + rc >= nch ? c + rc - nch : c + rc;
    const unsigned dst = lds_base + buf * Geo::BUF + Geo::A_BYTES + wave * 1024;
    const unsigned delta = (unsigned)(cc * PKCB);
#pragma unroll
    for (int u = 0; u < 4; ++u) dma16(b_off[u] + delta, dst + u * 8192, wt_rsrc);
    if (wave < 4) dma16(b_off[4] + delta, dst + 4 * 8192, wt_rsrc);
  };
  issue_b(0, 0);  // first thing: the weights do not depend on anything computed below

  // ---- MODE 0: the patch by LDS-DMA; piece q = wave + 8 j holds patch pixels 8 q .. 8 q + 7
  unsigned a_off[Geo::APW];
  unsigned a_ok = 0;
  v4i_t in_rsrc = wt_rsrc;
  if constexpr (MODE == 0) {
    in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * Geo::HW * p.Cin * 2));
#pragma unroll
    for (int j = 0; j < Geo::APW; ++j) {
      const int P = (wave + 8 * j) * 8 + (lane >> 3), pos = lane & 7;
      const int seg = P / (Geo::PR * Geo::PW), r = P - seg * (Geo::PR * Geo::PW);
      const int pr = r / Geo::PW, pc = r - pr * Geo::PW;
      const int n = img0 + seg, iy = pr - 1, ix = pc - 1;
      const bool ok = P < Geo::NP && n < p.N && iy >= 0 && iy < W && ix >= 0 && ix < W;
      a_off[j] = (unsigned)((((long)n * W + iy) * W + ix) * p.Cin * 2 + ((pos ^ Geo::swz(pr, pc)) << 4));
      a_ok |= ok ? (1u << j) : 0u;
    }
  }
  auto issue_a = [&](int c, int buf) {
    const unsigned dst = lds_base + buf * Geo::BUF + wave * 1024;
    const int cc = c + rc >= nch ? c + rc - nch : c + rc;
    const unsigned delta = (unsigned)(cc * PKCB);
#pragma unroll
    for (int j = 0; j < Geo::APW; ++j)
      if (wave + 8 * j < Geo::PIECES) dma16(((a_ok >> j) & 1) ? a_off[j] + delta : OOB, dst + j * 8192, in_rsrc);
  };
  if constexpr (MODE == 0) issue_a(0, 0);

  // ---- MODE 1 / 2: the patch computed from the fusion node's inputs.  Item = (pixel, 16-byte LDS position) of the chunk.
  float fw[3] = {0.f, 0.f, 0.f};
  constexpr int NLD = MODE == 1 ? 5 : (MODE == 2 ? 11 : 1);
  uint4 ld[ITEMS][NLD];
  int it_lds[ITEMS], it_ch[ITEMS];  // LDS byte offset inside a patch buffer; first channel of the item inside a chunk
  long it_out[ITEMS];               // element offset of the item's pixel in a [N][W][W][Cin] tensor, or -1
  if constexpr (MODE != 0) {
    pyr_softmax(p.fw, MODE == 1 ? 2 : 3, fw);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int it = tid + PTHREADS * k, pl = it >> 3, pos = it & 7;
      const int seg = pl / Geo::HW, wi = pl - seg * Geo::HW, y = wi / W, x = wi - y * W;
      const int pr = y + 1, pc = x + 1;
      it_lds[k] = (seg * Geo::PR * Geo::PW + pr * Geo::PW + pc) * PKCB + pos * 16;
      it_ch[k] = (pos ^ Geo::swz(pr, pc)) * 8;
      const int n = img0 + seg;
      it_out[k] = n < p.N ? (((long)n * W + y) * W + x) * p.Cin : -1;
    }
  }
  auto fused_load = [&](int c) {  // global loads of chunk c's inputs (in flight under the previous chunk's multiplies)
    const int cc = c + rc >= nch ? c + rc - nch : c + rc;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int it = tid + PTHREADS * k, pl = it >> 3;
      const int seg = pl / Geo::HW, wi = pl - seg * Geo::HW, y = wi / W, x = wi - y * W;
      const int n = img0 + seg, ch = cc * 64 + it_ch[k];
      if (it_out[k] < 0) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) ld[k][q] = make_uint4(0, 0, 0, 0);
        continue;
      }
      if constexpr (MODE == 1) {
        constexpr int W2 = W / 2;
        const PLerp ly = pyr_up2_src(y, W2), lx = pyr_up2_src(x, W2);
        const bf16_t* a0 = (const bf16_t*)p.a + (((long)n * W2 + ly.i0) * W2) * p.Cin + ch;
        const bf16_t* a1 = (const bf16_t*)p.a + (((long)n * W2 + ly.i1) * W2) * p.Cin + ch;
        ld[k][0] = *(const uint4*)(a0 + lx.i0 * p.Cin);
        ld[k][1] = *(const uint4*)(a0 + lx.i1 * p.Cin);
        ld[k][2] = *(const uint4*)(a1 + lx.i0 * p.Cin);
        ld[k][3] = *(const uint4*)(a1 + lx.i1 * p.Cin);
        ld[k][4] = *(const uint4*)((const bf16_t*)p.b + it_out[k] + ch);
      } else if constexpr (MODE == 2) {
        constexpr int WA = 2 * W;
        const bf16_t* an = (const bf16_t*)p.a + (long)n * WA * WA * p.Cin + ch;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const bf16_t* ar = an + (long)pyr_reflect(2 * y + dy - 1, WA) * WA * p.Cin;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) ld[k][dy * 3 + dx] = *(const uint4*)(ar + pyr_reflect(2 * x + dx - 1, WA) * p.Cin);
        }
        ld[k][9] = *(const uint4*)((const bf16_t*)p.b + it_out[k] + ch);
        ld[k][10] = *(const uint4*)((const bf16_t*)p.c + it_out[k] + ch);
      }
    }
  };
  auto fused_finish = [&](int c, int buf) {  // the merged values of chunk c: into the patch buffer (and to `merged`)
    const int cc = c + rc >= nch ? c + rc - nch : c + rc;
    char* dstbuf = smem + buf * Geo::BUF;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      float o[8];
      if constexpr (MODE == 1) {
        const int it = tid + PTHREADS * k, pl = it >> 3;
        const int wi = pl % Geo::HW, y = wi / W, x = wi - y * W;
        const PLerp ly = pyr_up2_src(y, W / 2), lx = pyr_up2_src(x, W / 2);
        float f00[8], f01[8], f10[8], f11[8];
        unpack16(ld[k][0], f00, bf16_t());
        unpack16(ld[k][1], f01, bf16_t());
        unpack16(ld[k][2], f10, bf16_t());
        unpack16(ld[k][3], f11, bf16_t());
        unpack16(ld[k][4], o, bf16_t());
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float up = node_up2(ly.l0, ly.l1, lx.l0, lx.l1, f00[e], f01[e], f10[e], f11[e]);
          o[e] = node_fuse2(fw[0], fw[1], up, o[e]);
        }
      } else if constexpr (MODE == 2) {
        const float k1[3] = {0.25f, 0.5f, 0.25f};
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            float f[8];
            unpack16(ld[k][dy * 3 + dx], f, bf16_t());
            const float kk = k1[dy] * k1[dx];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __fmaf_rn(kk, f[e], acc[e]);
          }
        if (p.a_scale) {
          const int ch = cc * 64 + it_ch[k];
          const float4 s0 = *(const float4*)(p.a_scale + ch), s1 = *(const float4*)(p.a_scale + ch + 4);
          const float4 t0 = *(const float4*)(p.a_shift + ch), t1 = *(const float4*)(p.a_shift + ch + 4);
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float sf[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = __fmaf_rn(acc[e], sc[e], sf[e]);
        }
        float fb[8], fc[8];
        unpack16(ld[k][9], fb, bf16_t());
        unpack16(ld[k][10], fc, bf16_t());
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = node_fuse3(fw[0], fw[1], fw[2], acc[e], fb[e], fc[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
      }
      uint4 v = pack16(o, bf16_t());
      if (it_out[k] < 0) v = make_uint4(0, 0, 0, 0);
      *(uint4*)(dstbuf + it_lds[k]) = v;
      if (p.merged && tile_n == 0 && it_out[k] >= 0) *(uint4*)((bf16_t*)p.merged + it_out[k] + cc * 64 + it_ch[k]) = v;
    }
  };
  if constexpr (MODE != 0) {
    fused_load(0);
    // the halo of both patch areas is zero for the whole launch; the loaders write the interior only
    for (int bufi = 0; bufi < 2; ++bufi)
      for (int o = tid * 16; o < Geo::A_BYTES; o += PTHREADS * 16) *(uint4*)(smem + bufi * Geo::BUF + o) = make_uint4(0, 0, 0, 0);
  }

  // ---- the slice's epilogue parameters: into LDS now (their load latency hides behind the first chunk's DMA)
  if (tid < 5 * PBN) {
    const int which = tid / PBN, col = tid % PBN;
    const float* src = which == 0 ? p.bias : which == 1 ? p.pre_scale : which == 2 ? p.pre_shift : which == 3 ? p.post_scale : p.post_shift;
    const float dflt = (which == 1 || which == 3) ? 1.f : 0.f;
    ((float*)(smem + Geo::PAR_OFF))[tid] = src ? src[n0 + col] : dflt;
  }

  // ---- fragment geometry: m-tile i of this wave covers pixels (pg * MTW + i) * 16 + l16 of the tile
  int q0[MTW], py[MTW], px[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int pl = (pg * MTW + i) * 16 + l16;
    const int seg = pl / Geo::HW, wi = pl - seg * Geo::HW;
    py[i] = wi / W;
    px[i] = wi - py[i] * W;
    q0[i] = seg * Geo::PR * Geo::PW + py[i] * Geo::PW + px[i];
  }
  const int kpiece = 4 * kh + g;  // this lane's 16-byte piece of a pixel's / weight row's 128 B: K half kh, lane group g
  int w_off[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int row = nt * 16 + l16;
    w_off[nt] = row * PKCB + ((kpiece ^ ((row >> 1) & 7)) << 4);
  }

  f32x4_t acc[MTW][2];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][nt][r] = 0.f;

  if constexpr (MODE != 0) {
    __syncthreads();  // the zero fill is complete before anyone writes an interior pixel
    fused_finish(0, 0);
  }
  PYR_STAMP(1);

  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    wait_vm_keep<0>();  // chunk c's DMA pieces of this wave have landed ...
    __syncthreads();    // ... and everyone's; everyone is done with the other buffer (chunk c - 1)
    if (c == 0) PYR_STAMP(2);
    if (c + 1 < nch) {
      issue_b(c + 1, buf ^ 1);
      if constexpr (MODE == 0) issue_a(c + 1, buf ^ 1);
      else fused_load(c + 1);
    }
    const char* As = smem + buf * Geo::BUF;
    const char* Bs = As + Geo::A_BYTES;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        uint4 fa[MTW], wf[2];
#pragma unroll
        for (int i = 0; i < MTW; ++i)
          fa[i] = *(const uint4*)(As + (q0[i] + ky * Geo::PW + kx) * PKCB + ((kpiece ^ Geo::swz(py[i] + ky, px[i] + kx)) << 4));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) wf[nt] = *(const uint4*)(Bs + (ky * 3 + kx) * PBN * PKCB + w_off[nt]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[nt]), __builtin_bit_cast(bf16x8_t, fa[i]),
                                                                acc[i][nt], 0, 0, 0);
      }
    if constexpr (MODE != 0) if (c + 1 < nch) fused_finish(c + 1, buf ^ 1);
  }

  PYR_STAMP(3);
  // ---- the two K halves meet in LDS: staging tile kh, row = pixel, 32 fp32 channels (lane: channels nt * 16 + 4 g + 0..3)
  __syncthreads();  // everyone is done reading the patch and the weights
  {
    char* st = smem + kh * (Geo::TM * PST_STRIDE);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int pl = (pg * MTW + i) * 16 + l16;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        *(float4*)(st + pl * PST_STRIDE + (nt * 16 + 4 * g) * 4) = make_float4(acc[i][nt][0], acc[i][nt][1], acc[i][nt][2], acc[i][nt][3]);
    }
  }
  __syncthreads();
  PYR_STAMP(4);

  // ---- epilogue: item = (pixel, 8 consecutive channels): bias -> [stats] -> affine -> act -> [stats] -> affine
  constexpr int EPI = Geo::EPI, ROWS = Geo::ROWS;
  const float* par = (const float*)(smem + Geo::PAR_OFF);
  {
    const int cg = tid & 3;
    const int co = n0 + cg * 8;
    float bias[8], s1[8], t1[8], s2[8], t2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bias[e] = par[0 * PBN + cg * 8 + e];
      s1[e] = par[1 * PBN + cg * 8 + e];
      t1[e] = par[2 * PBN + cg * 8 + e];
      s2[e] = par[3 * PBN + cg * 8 + e];
      t2[e] = par[4 * PBN + cg * 8 + e];
    }
    bf16_t* __restrict__ out = (bf16_t*)p.out;
#pragma unroll
    for (int k = 0; k < EPI; ++k) {
      const int it = tid + PTHREADS * k, pl = it >> 2;
      if (pl >= Geo::TM) break;  // (64-pixel tiles: threads 0-255 only)
      const int m = m0 + pl;
      float v[8];
      {
        const char* r0 = smem + pl * PST_STRIDE + cg * 32;
        const char* r1 = r0 + Geo::TM * PST_STRIDE;
        const float4 a0 = *(const float4*)r0, a1 = *(const float4*)(r0 + 16), b0 = *(const float4*)r1, b1 = *(const float4*)(r1 + 16);
        v[0] = a0.x + b0.x; v[1] = a0.y + b0.y; v[2] = a0.z + b0.z; v[3] = a0.w + b0.w;
        v[4] = a1.x + b1.x; v[5] = a1.y + b1.y; v[6] = a1.z + b1.z; v[7] = a1.w + b1.w;
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = pyr_act((v[e] + bias[e]) * s1[e] + t1[e], p.act) * s2[e] + t2[e];
      const uint4 packed = pack16(o, bf16_t());
      if (m < p.M && p.write_y) *(uint4*)(out + (long)m * p.Cout + co) = packed;
      if (p.emit) *(uint4*)(smem + Geo::Y_OFF + pl * (PBN * 2) + cg * 16) = packed;  // the slice's y, bf16: [pixel][32 channels]
    }
  }
  if (p.emit) {
    __syncthreads();
    float ew[3];
    pyr_softmax(p.e_fw, p.emit == 1 ? 2 : 3, ew);
    const char* ys = smem + Geo::Y_OFF;
    if (p.emit == 1) {
      constexpr int WO = 2 * W, NIT = Geo::G * WO * WO * 4;  // (output pixel, 8 channels) items of the tile
      for (int it = tid; it < NIT; it += PTHREADS) {
        const int cg = it & 3, op = it >> 2;
        const int seg = op / (WO * WO), r = op - seg * (WO * WO), oy = r / WO, ox = r - oy * WO;
        const int n = img0 + seg;
        if (n >= p.N) continue;
        const PLerp ly = pyr_up2_src(oy, W), lx = pyr_up2_src(ox, W);
        const char* y0 = ys + (seg * Geo::HW + ly.i0 * W) * (PBN * 2) + cg * 16;
        const char* y1 = ys + (seg * Geo::HW + ly.i1 * W) * (PBN * 2) + cg * 16;
        float f00[8], f01[8], f10[8], f11[8], o[8];
        unpack16(*(const uint4*)(y0 + lx.i0 * (PBN * 2)), f00, bf16_t());
        unpack16(*(const uint4*)(y0 + lx.i1 * (PBN * 2)), f01, bf16_t());
        unpack16(*(const uint4*)(y1 + lx.i0 * (PBN * 2)), f10, bf16_t());
        unpack16(*(const uint4*)(y1 + lx.i1 * (PBN * 2)), f11, bf16_t());
        const long eo = (((long)n * WO + oy) * WO + ox) * p.Cout + n0 + cg * 8;
        unpack16(*(const uint4*)((const bf16_t*)p.e_b + eo), o, bf16_t());
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = node_fuse2(ew[0], ew[1], node_up2(ly.l0, ly.l1, lx.l0, lx.l1, f00[e], f01[e], f10[e], f11[e]), o[e]);
        *(uint4*)((bf16_t*)p.e_out + eo) = pack16(o, bf16_t());
      }
    } else {
      constexpr int WO = W / 2, NIT = Geo::G * WO * WO * 4;
      const float k1[3] = {0.25f, 0.5f, 0.25f};
      for (int it = tid; it < NIT; it += PTHREADS) {
        const int cg = it & 3, op = it >> 2;
        const int seg = op / (WO * WO), r = op - seg * (WO * WO), oy = r / WO, ox = r - oy * WO;
        const int n = img0 + seg;
        if (n >= p.N) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const char* yr = ys + (seg * Geo::HW + pyr_reflect(2 * oy + dy - 1, W) * W) * (PBN * 2) + cg * 16;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            float f[8];
            unpack16(*(const uint4*)(yr + pyr_reflect(2 * ox + dx - 1, W) * (PBN * 2)), f, bf16_t());
            const float kk = k1[dy] * k1[dx];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __fmaf_rn(kk, f[e], acc[e]);
          }
        }
        const long eo = (((long)n * WO + oy) * WO + ox) * p.Cout + n0 + cg * 8;
        float fb[8], fc[8];
        unpack16(*(const uint4*)((const bf16_t*)p.e_b + eo), fb, bf16_t());
        unpack16(*(const uint4*)((const bf16_t*)p.e_c + eo), fc, bf16_t());
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = node_fuse3(ew[0], ew[1], ew[2], acc[e], fb[e], fc[e]);
        *(uint4*)((bf16_t*)p.e_out + eo) = pack16(acc, bf16_t());
      }
    }
  }
  PYR_STAMP(5);
  if (p.stats_mode) {
    // column sums in a fixed order: thread = (column, one of 16 runs of consecutive pixels) over the staged fp32 tile, then
    // the runs of a statistics row.  (The first version summed inside the store loop and folded lanes with 128 shuffles per
    // wave: 3.6 us on P5 by its stamps.)
    constexpr int PP = Geo::TM / 16;
    const int col = tid & 31, part = tid >> 5;
    const float bias = par[col], s1 = par[PBN + col], t1 = par[2 * PBN + col];
    float sum = 0.f, sq = 0.f;
    float xa[PP], xb[PP];
#pragma unroll
    for (int i = 0; i < PP; ++i) {  // all reads in flight together (a serial read -> add chain cost 2 us here)
      xa[i] = *(const float*)(smem + (part * PP + i) * PST_STRIDE + col * 4);
      xb[i] = *(const float*)(smem + (Geo::TM + part * PP + i) * PST_STRIDE + col * 4);
    }
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      float x = xa[i] + xb[i] + bias;
      if (p.stats_mode == 2) x = pyr_act(x * s1 + t1, p.act);
      if (m0 + part * PP + i < p.M) { sum += x; sq += x * x; }
    }
    float* red = (float*)(smem + Geo::RED_OFF);  // [16 runs][2][32]
    red[(part * 2 + 0) * PBN + col] = sum;
    red[(part * 2 + 1) * PBN + col] = sq;
    __syncthreads();
    if (tid < ROWS * 2 * PBN) {
      constexpr int RUNS = 16 / ROWS;  // runs per statistics row
      const int r = tid / (2 * PBN), which = (tid / PBN) & 1, c2 = tid % PBN;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < RUNS; ++w) a += red[((r * RUNS + w) * 2 + which) * PBN + c2];
      p.stats[(((long)tile_m * ROWS + r) * 2 + which) * p.Cout + n0 + c2] = a;
    }
  }
  PYR_STAMP(6);
}

unsigned long long* g_pyr_stamps = nullptr;
int g_pyr = 1;  // test hook (sihl_conv2d_small_enable): 1 = this kernel, 2 = conv_small.hip, 0 = the general tile kernel

template <int W, int MODE>
int launch_pyr(const PyrParams& p, hipStream_t stream) {
  using Geo = PyrGeo<W>;
  static_assert(Geo::LDS <= 160 * 1024 && Geo::Y_OFF + Geo::TM * PBN * 2 <= 2 * Geo::BUF, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_pyr_kernel<W, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  PyrParams q = p;
  q.tiles = (p.N + Geo::G - 1) / Geo::G;
  hipLaunchKernelGGL((conv_pyr_kernel<W, MODE>), dim3(q.tiles * q.gridN), dim3(PTHREADS), Geo::LDS, stream, q);
  return SIHL_OK;
}

template <int MODE>
int launch_pyr_w(const PyrParams& p, int W, hipStream_t stream) {
  if (W == 16) {
    // 16x16 maps take no fused producer: four items per thread (80 / 176 registers of loads in flight), and every one of a
    // map's 8 output slices recomputes the whole node - measured slower than the stand-alone fusion kernel + this conv
    if constexpr (MODE != 0) return SIHL_EARG;
    else return launch_pyr<16, MODE>(p, stream);
  }
  if (W == 8) return launch_pyr<8, MODE>(p, stream);
  return launch_pyr<4, MODE>(p, stream);
}

bool pyr_shape_ok(int N, int W, int Cin, int Cout, int mode) {
  if (N <= 0 || (W != 16 && W != 8 && W != 4) || Cin <= 0 || Cin % 64 || Cout <= 0 || Cout % PBN) return false;
  if (mode < 0 || mode > 2 || (mode != 0 && W == 16)) return false;
  const long scale = mode == 2 ? 4 : 1;
  return (long)N * W * W * Cin * 2 * scale < (1L << 31) && (long)Cout * 9 * Cin * 2 < (1L << 31);
}

int pyr_launch(PyrParams p, int W, int mode, hipStream_t stream) {
  p.gridN = p.Cout / PBN;
  p.M = p.N * W * W;
  p.k_rotate = g_krot % 1000 != 0;
  p.stamps = g_pyr_stamps;
  const double flops = 2.0 * p.M * (double)p.Cout * 9 * p.Cin;
  const double bytes = ((double)p.M * p.Cin * (mode == 0 ? 1.0 : (mode == 1 ? 1.25 : 6.0)) + (double)p.M * p.Cout +
                        (double)p.Cout * 9 * p.Cin) * 2.0;
  sihl_prof_begin(SIHL_PROF_CONV, SIHL_BF16, flops, bytes, stream);
  int rc;
  if (mode == 0) rc = launch_pyr_w<0>(p, W, stream);
  else if (mode == 1) rc = launch_pyr_w<1>(p, W, stream);
  else rc = launch_pyr_w<2>(p, W, stream);
  sihl_prof_end(stream);
  if (rc != SIHL_OK) return rc;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // namespace

void sihl_pyr_set_mode(int mode) { g_pyr = mode; }
#ifdef SIHL_PYR_STAMPS
extern "C" int sihl_pyr_stamps(void* buf) { g_pyr_stamps = (unsigned long long*)buf; return SIHL_OK; }  // 14 x u64, device memory
#endif
int sihl_pyr_get_mode() { return g_pyr; }

// The generic conv entry (sihl_conv2d_fwd_ws / dgrad) takes this kernel for plain 3x3 convs on the small maps when the
// statistics rows it would write are the generic ones (one per 128 pixels: 16x16 maps, or no statistics at all).
bool sihl_pyr_eligible(const ConvParams& p) {
  if (g_pyr != 1 || !sihl_small_eligible(p)) return false;
  if (p.stats_mode && p.W != 16) return false;
  return pyr_shape_ok(p.N, p.W, p.Cin, p.Cout, 0);
}

int sihl_pyr_launch(const ConvParams& c, hipStream_t stream) {
  PyrParams p = {};
  p.in = c.in; p.wt = c.wt; p.out = c.out;
  p.bias = c.bias; p.pre_scale = c.pre_scale; p.pre_shift = c.pre_shift; p.post_scale = c.post_scale; p.post_shift = c.post_shift;
  p.stats = c.stats;
  p.N = c.N; p.Cin = c.Cin; p.Cout = c.Cout; p.act = c.act; p.stats_mode = c.stats_mode;
  p.write_y = 1;
  return pyr_launch(p, c.W, 0, stream);
}

extern "C" {

// 1 when sihl_pyr_conv_fwd covers the shape: bf16, square maps of 16, 8 or 4, Cin % 64 == 0, Cout % 32 == 0; the fused
// producers (modes 1, 2) on 8x8 and 4x4 maps only.
int sihl_pyr_conv_supported(int N, int W, int Cin, int Cout, int mode) { return pyr_shape_ok(N, W, Cin, Cout, mode) ? 1 : 0; }

// Rows of (sum, sum of squares) partials the launch writes: one per 128 pixels on 16x16 maps, one per tile otherwise.
int sihl_pyr_conv_stat_rows(int N, int W) {
  if (W == 16) return 2 * N;
  if (W == 8) return N;
  return (N + 3) / 4;
}

int sihl_pyr_conv_fwd(const void* in, const void* wt, const float* bias, void* out, int N, int W, int Cin, int Cout, int act,
                      const float* pre_scale, const float* pre_shift, const float* post_scale, const float* post_shift,
                      int stats_mode, float* stats, long stats_bytes, int mode, const void* a, const void* b, const void* c,
                      const float* fw, const float* a_scale, const float* a_shift, void* merged, int emit, const void* e_b,
                      const void* e_c, const float* e_fw, void* e_out, hipStream_t stream) {
  if (!wt || !pyr_shape_ok(N, W, Cin, Cout, mode)) return SIHL_EARG;
  if (emit < 0 || emit > 2 || (!out && !emit)) return SIHL_EARG;
  if (emit && (!e_b || !e_fw || !e_out || (emit == 2 && (!e_c || W % 2)))) return SIHL_EARG;
  if (mode == 0 ? !in : (!a || !b || !fw || (mode == 2 && !c))) return SIHL_EARG;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_scale && mode != 2)) return SIHL_EARG;
  if (act < SIHL_ACT_NONE || act > SIHL_ACT_SIGMOID || stats_mode < 0 || stats_mode > 2) return SIHL_EARG;
  if (stats_mode && (!stats || stats_bytes < (long)sihl_pyr_conv_stat_rows(N, W) * 2 * Cout * 4)) return SIHL_EWS;
  PyrParams p = {};
  p.in = in; p.wt = wt; p.out = out; p.a = a; p.b = b; p.c = c; p.fw = fw; p.a_scale = a_scale; p.a_shift = a_shift;
  p.merged = merged;
  p.emit = emit; p.e_b = e_b; p.e_c = e_c; p.e_fw = e_fw; p.e_out = e_out; p.write_y = out != nullptr;
  p.bias = bias; p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.post_scale = post_scale; p.post_shift = post_shift;
  p.stats = stats_mode ? stats : nullptr;
  p.N = N; p.Cin = Cin; p.Cout = Cout; p.act = act; p.stats_mode = stats_mode;
  return pyr_launch(p, W, mode, stream);
}

}  // extern "C"
