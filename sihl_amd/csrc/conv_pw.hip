// Persistent weight-stationary pointwise (1x1, stride 1) bf16 convolution for the HBM-bound layers of the hot path:
// ResNet bottleneck expansions / reductions (src/sihl/torchvision_backbone.py:50-67 wraps torchvision's resnet50),
// the BiFPN laterals and the Linear layers of the heads' MLPs (src/sihl/heads/object_detection.py:51-61) when the
// contraction is short (Cin <= 256).  out[m][co] = epilogue(sum_ci in[m][ci] * wt[co][ci]).
//
// Why a second kernel: with K <= 256 an output tile needs 1-4 LDS stages of MFMA work and the layer is bound by moving
// M x (Cin + Cout) x 2 bytes.  The one-tile-per-workgroup kernel (conv_igemm.hip) launches M/128 x Cout/128 workgroups
// whose load -> multiply -> store phases only overlap through co-residency (3 per CU); measured on r1 64>256 (335 MB):
// loads alone 47 us, stores alone 56 us, the whole kernel 114 us - the two phases add up (profiles/r02_thin_ablate.txt).
// Here ONE workgroup per CU stays resident:
//   * its 128 output channels' weights (Cin x 128 x 2 B <= 64 KiB) are DMA'd to LDS once;
//   * it walks the pixel tiles  slot, slot + nslots, ...  of its channel tile; the A operand streams through a ring of
//     NR 16-KiB chunks (128 pixels x 64 channels) filled by LDS-DMA up to NR - 1 chunks ahead, so the loads of the next
//     tiles are in flight while this tile's epilogue converts and stores;
//   * counted s_waitcnt vmcnt: the wait for chunk g allows the (NR - 2) x 4 DMA instructions issued after it to stay in
//     flight.  The epilogue's global stores count in vmcnt too and retire in order with the DMAs, so this is a LOWER
//     bound of what is younger than chunk g - never an under-wait (the wait may additionally retire older stores).
//     Beyond its last tile the ring keeps issuing out-of-range DMAs (zero fill, never read) so the count stays uniform.
//   * channel tiles of one pixel tile run on the SAME XCD (logical id = xcd-major), so the A tile is fetched from HBM
//     once and re-read from that XCD's L2.
// Same MFMA shape, fragment layout, XOR source-side swizzle and epilogue arithmetic (bias -> [stats] -> act -> [stats],
// BatchNorm partial rows per 128-pixel tile) as conv_igemm_dma_kernel: the two kernels are interchangeable bit for bit.
#include "common.h"
#include "dma.h"
#include "conv_params.h"
#include "profile.h"

namespace {

// 0 off (default), 1 on, 2 on without the minimum-work rule (tests: small shapes reach the kernel).  OPT-IN: once the tile
// kernel's epilogue was packed as well (and its register budget pinned to 4 workgroups per CU) it matches or beats this
// kernel on every shape of the step but r2 256>128 (-5 %): r1 64>256 78 vs 81 us, mlp 45 vs 57, r3 256>1024 29 vs 43
// (profiles/r02_pw_bench.txt).  One wave per SIMD cannot hide its own ds_read -> MFMA latency, and a CU whose four waves
// are blocked issuing stores (HBM write back-pressure) issues no loads: the phases still alternate per CU.
int g_pw = 0;

constexpr int PW_BM = 128, PW_BN = 128, PW_KCB = 128;  // tile; bytes of K per row per chunk (64 bf16)
constexpr int PW_CHUNK = PW_BM * PW_KCB;               // 16 KiB: one A chunk, one resident B chunk
constexpr int PW_EPI_STRIDE = PW_BN * 2 + 16;          // staging row (bytes), padded: conflict-free b16 writes
constexpr int PW_EPI = PW_BM * PW_EPI_STRIDE + 2 * 2 * PW_BN * 4;  // staging + [2][WM=2][BN] statistics partials
constexpr int PW_NA = 4;                               // DMA instructions per wave per chunk (16 per chunk, 4 waves)

__device__ __forceinline__ void pw_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// (s_nop 4 in front of every inline-asm buffer instruction of this file: the kernel spills descriptor SGPRs to VGPR lanes,
// the compiler reloads them with v_readlane right before the statement, and "VALU writes SGPR -> VMEM reads it" needs
// 5 wait states that the hazard recogniser cannot insert for an instruction it does not see.  Without them the
// statistics rows were silently dropped: the store ran with a stale descriptor.)
__device__ __forceinline__ void pw_dma16(unsigned voff, unsigned lds_dst, v4i_t rsrc) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 3\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(rsrc)
      : "memory");
}
// (s_nop 1 BEHIND the 16-byte store: "VMEM store of more than 8 bytes followed by a VALU write of its data registers" needs
// a wait state, and the compiler does recycle v0 of the data quad for the next address in the very next instruction -
// lanes 12-15 of every 16 then stored the new address instead of their first two channels.  Same blind spot.)
// Output / statistics stores as raw buffer instructions through inline asm: every wave issues EXACTLY the same number per
// tile (rows beyond M fall outside the descriptor and are dropped by the bounds check instead of being branched around),
// which is what lets the ring wait with an exact vmcnt (below).
__device__ __forceinline__ void pw_store16(u32x4_t v, unsigned voff, v4i_t rsrc) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pw_store4(float v, unsigned voff, v4i_t rsrc) {
  asm volatile("s_nop 4\n\tbuffer_store_dword %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
// Wait until chunk g has landed.  Vector-memory operations retire in issue order; younger than chunk g are the NR - 2
// chunks requested after it (PW_NA instructions each) and the stores of the n_epi tile epilogues that ran in between (ST
// instructions each): exactly that many may stay in flight.
template <int NR, int ST> __device__ __forceinline__ void pw_wait_chunk(int n_epi) {
  static_assert((NR - 2) * PW_NA + (NR - 1) * ST <= 63, "vmcnt immediate");
  constexpr int D = (NR - 2) * PW_NA;
  switch (n_epi) {
    case 0: wait_vm_keep<D>(); break;
    case 1: wait_vm_keep<D + ST>(); break;
    case 2: wait_vm_keep<D + 2 * ST>(); break;
    case 3: if (NR > 3) { wait_vm_keep<D + (NR > 3 ? 3 : 0) * ST>(); break; }
    case 4: if (NR > 4) { wait_vm_keep<D + (NR > 4 ? 4 : 0) * ST>(); break; }
    default: wait_vm_keep<D + (NR - 1) * ST>(); break;
  }
}

template <int ACT> __device__ __forceinline__ float pw_act(float v) {
  if (ACT == SIHL_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == SIHL_ACT_SILU) return v / (1.f + expf(-v));
  return v;
}

// acc (2 x 2 MFMA tiles of 32x32 per wave) -> staging rows in LDS (bf16) + per-wave-row statistics partials
template <int ACT, int STATS>
__device__ __forceinline__ void pw_stage_tile(const ConvParams& p, f32x16_t (&acc)[2][2], char* epi, float* red, int m0,
                                              int wm, int wn, int lane, const float (&bias)[2]) {
  const int half = lane >> 5;
  const int lim = p.M - m0;  // rows of this tile inside M (>= 128 for every tile but the last)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cl = wn * 64 + j * 32 + (lane & 31);
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[i][j][r] + bias[j];
        if (STATS == 1) { const float m = row < lim ? v : 0.f; ssum += m; ssq += m * m; }
        v = pw_act<ACT>(v);
        if (STATS == 2) { const float m = row < lim ? v : 0.f; ssum += m; ssq += m * m; }
        elem<bf16_t>::st((bf16_t*)(epi + row * PW_EPI_STRIDE) + cl, v);
      }
    }
    if (STATS) {
      ssum += __shfl_xor(ssum, 32);
      ssq += __shfl_xor(ssq, 32);
      if (half == 0) {
        red[(0 * 2 + wm) * PW_BN + cl] = ssum;
        red[(1 * 2 + wm) * PW_BN + cl] = ssq;
      }
    }
  }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// The same staging for a tile that lies wholly inside M, two rows at a time.  The element loop is what bounds a resident
// workgroup (one wave per SIMD, 64 outputs per lane, every VALU instruction 4 cycles): packed fp32 adds / FMAs for the
// bias and the statistics, one v_cvt_pk_bf16_f32 per row pair, no row mask.  Per lane the statistics are summed as
// (even rows, odd rows) pairs and folded at the end: a different order from pw_stage_tile / the tile kernel (equal to
// fp32 rounding, not bit for bit).
template <int ACT, int STATS, bool HASB>
__device__ __forceinline__ void pw_stage_full(f32x16_t (&acc)[2][2], char* epi, float* red, int wm, int wn, int lane,
                                              const float (&bias)[2]) {
  const int half = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cl = wn * 64 + j * 32 + (lane & 31);
    char* col = epi + (wm * 64 + 4 * half) * PW_EPI_STRIDE + cl * 2;
    f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
    const f32x2_t b2 = {bias[j], bias[j]};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2);  // (+ wm*64 + 4*half in `col`); element r + 1 is row + 1
        f32x2_t v = {acc[i][j][r], acc[i][j][r + 1]};
        if (HASB) v += b2;
        if (STATS == 1) { s2 += v; q2 = __builtin_elementwise_fma(v, v, q2); }
        if (ACT == SIHL_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
        else if (ACT == SIHL_ACT_SILU) { v.x = v.x / (1.f + expf(-v.x)); v.y = v.y / (1.f + expf(-v.y)); }
        if (STATS == 2) { s2 += v; q2 = __builtin_elementwise_fma(v, v, q2); }
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        *(unsigned short*)(col + row * PW_EPI_STRIDE) = (unsigned short)pk;
        *(unsigned short*)(col + (row + 1) * PW_EPI_STRIDE) = (unsigned short)(pk >> 16);
      }
    }
    if (STATS) {
      float ssum = s2.x + s2.y, ssq = q2.x + q2.y;
      ssum += __shfl_xor(ssum, 32);
      ssq += __shfl_xor(ssq, 32);
      if (half == 0) {
        red[(0 * 2 + wm) * PW_BN + cl] = ssum;
        red[(1 * 2 + wm) * PW_BN + cl] = ssq;
      }
    }
  }
}

template <int NR>
__global__ __launch_bounds__(256) void conv_pw_kernel(const ConvParams p, int nslots, int tiles_m) {
  static_assert(NR >= 3 && NR <= 6 && (NR - 2) * PW_NA <= 63, "ring depth / vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // logical id, XCD-major: the gridN channel tiles of one pixel slot are neighbours on one XCD (gridDim.x % 8 == 0)
  const int L = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const int tile_n = L % p.gridN, slot = L / p.gridN;
  if (slot >= tiles_m) return;  // (whole workgroup)
  const int n0 = tile_n * PW_BN;
  const int KC = p.Cin / 64;                                           // chunks per tile
  const int my_tiles = (tiles_m - slot + nslots - 1) / nslots;
  const int G = my_tiles * KC;

  char* bres = smem;                       // [KC][128 rows][128 B] resident weights
  char* ring = smem + KC * PW_CHUNK;       // [NR][128 rows][128 B]
  char* epi = ring + NR * PW_CHUNK;
  float* red = (float*)(epi + PW_BM * PW_EPI_STRIDE);
  const unsigned ring_lds = (unsigned)(unsigned long)(lds_ptr_t)ring;
  const unsigned bres_lds = (unsigned)(unsigned long)(lds_ptr_t)bres;

  const unsigned row_bytes = (unsigned)p.Cin * 2u;
  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.M * p.Cin * 2L));     // rows >= M read as zeros
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * p.Cin * 2L));

  // slot geometry of this lane: DMA instruction j of the wave fills LDS rows (wave*4 + j)*8 .. +7, 16 B per lane;
  // position `pos` of row `row` receives global chunk pos ^ ((row >> 1) & 7) of that row's 128-byte K slice
  unsigned a_off[PW_NA];
#pragma unroll
  for (int j = 0; j < PW_NA; ++j) {
    const int q = (wave * PW_NA + j) * 64 + lane, row = q >> 3, pos = q & 7;
    a_off[j] = (unsigned)row * row_bytes + (unsigned)((pos ^ ((row >> 1) & 7)) << 4);
  }
  // resident weights: rows n0 .. n0+127 of wt[Cout][Cin], chunk c = channels 64c .. 64c+63
  for (int c = 0; c < KC; ++c) {
#pragma unroll
    for (int j = 0; j < PW_NA; ++j)
      pw_dma16(a_off[j] + (unsigned)n0 * row_bytes + (unsigned)c * PW_KCB, bres_lds + c * PW_CHUNK + (wave * PW_NA + j) * 1024,
            wt_rsrc);
  }

  // issue state: next chunk to request
  int i_c = 0, i_tile = slot, i_slot = 0;
  unsigned i_base = (unsigned)slot * (unsigned)PW_BM * row_bytes;
  const unsigned tile_step = (unsigned)nslots * (unsigned)PW_BM * row_bytes;
  auto issue_next = [&]() {
    const bool valid = i_tile < tiles_m;
    const unsigned base = i_base + (unsigned)i_c * PW_KCB;
    const unsigned dst = ring_lds + i_slot * PW_CHUNK + wave * PW_NA * 1024;
#pragma unroll
    for (int j = 0; j < PW_NA; ++j) pw_dma16(valid ? a_off[j] + base : OOB, dst + j * 1024, in_rsrc);
    if (++i_c == KC) { i_c = 0; i_tile += nslots; i_base += tile_step; }
    i_slot = i_slot + 1 == NR ? 0 : i_slot + 1;
  };
#pragma unroll
  for (int s = 0; s < NR - 1; ++s) issue_next();

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = fr * PW_KCB + (((ks * 2 + fh) ^ fsw) << 4);

  float bias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bias[j] = p.bias ? p.bias[n0 + wn * 64 + j * 32 + (lane & 31)] : 0.f;
  // The bias is consumed HERE (empty asm with the values as inputs), so the compiler's s_waitcnt vmcnt(0) for this load
  // sits in front of the loop - it drains the weight and first ring DMAs once.  Left to the first real use (the
  // epilogue) the full drain would be re-executed every tile and empty the ring.  Nothing inside the loop reads global
  // memory through the compiler.
  asm volatile("" ::"v"(bias[0]), "v"(bias[1]));

  const v4i_t out_rsrc = make_rsrc(p.out, (unsigned)((long)p.M * p.Cout * 2L));  // rows >= M: stores dropped
  const v4i_t st_rsrc = make_rsrc(p.stats, p.stats ? (unsigned)((long)tiles_m * 2L * p.Cout * 4L) : 0u);
  unsigned epi_hist = 0;  // bit k: iteration g - 1 - k ended a tile (its epilogue's stores sit between the ring requests)
  int c = 0, r_slot = 0, tile_m = slot;
  for (int g = 0; g < G; ++g) {
    // chunk g (and the weights) have landed for this wave ...
    const int n_epi = (p.dbg & 256) ? 0 : __builtin_popcount(epi_hist & ((1u << (NR - 1)) - 1u));
    if (p.stats_mode) pw_wait_chunk<NR, 9>(n_epi); else pw_wait_chunk<NR, 8>(n_epi);
    epi_hist <<= 1;
    pw_barrier();                      // ... and for every wave; everyone is done reading chunk g - 1
    issue_next();                      // refill the slot chunk g - 1 occupied
    const char* As = ring + r_slot * PW_CHUNK + wm * 64 * PW_KCB;
    const char* Bs = bres + c * PW_CHUNK + wn * 64 * PW_KCB;
    if (!(SIHL_DBG(p) & 2)) {
      uint4 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[0][i] = *(const uint4*)(As + i * 32 * PW_KCB + koff[0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[0][j] = *(const uint4*)(Bs + j * 32 * PW_KCB + koff[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[(ks + 1) & 1][i] = *(const uint4*)(As + i * 32 * PW_KCB + koff[ks + 1]);
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[(ks + 1) & 1][j] = *(const uint4*)(Bs + j * 32 * PW_KCB + koff[ks + 1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[ks & 1][i]),
                                                                __builtin_bit_cast(bf16x8_t, fb[ks & 1][j]), acc[i][j], 0, 0, 0);
      }
    }
    r_slot = r_slot + 1 == NR ? 0 : r_slot + 1;
    if (++c < KC) continue;
    c = 0;
    // ---- epilogue of tile (tile_m, tile_n); the ring keeps streaming the next tiles meanwhile
    const int m0 = tile_m * PW_BM;
    if (!(SIHL_DBG(p) & 32)) {
      const bool full = m0 + PW_BM <= p.M;
      const bool hasb = p.bias != nullptr;
#define PW_FULL(A, S, B) pw_stage_full<A, S, B>(acc, epi, red, wm, wn, lane, bias)
#define PW_EDGE(A, S) pw_stage_tile<A, S>(p, acc, epi, red, m0, wm, wn, lane, bias)
      if (full && !(hasb && p.stats_mode)) {  // (conv -> BatchNorm layers carry no bias; the rare one that does goes below)
        if (p.stats_mode == 0) {
          if (p.act == SIHL_ACT_RELU) { if (hasb) PW_FULL(SIHL_ACT_RELU, 0, true); else PW_FULL(SIHL_ACT_RELU, 0, false); }
          else if (p.act == SIHL_ACT_SILU) { if (hasb) PW_FULL(SIHL_ACT_SILU, 0, true); else PW_FULL(SIHL_ACT_SILU, 0, false); }
          else { if (hasb) PW_FULL(SIHL_ACT_NONE, 0, true); else PW_FULL(SIHL_ACT_NONE, 0, false); }
        } else if (p.stats_mode == 1) {
          PW_FULL(SIHL_ACT_NONE, 1, false);
        } else {
          if (p.act == SIHL_ACT_RELU) PW_FULL(SIHL_ACT_RELU, 2, false);
          else if (p.act == SIHL_ACT_SILU) PW_FULL(SIHL_ACT_SILU, 2, false);
          else PW_FULL(SIHL_ACT_NONE, 2, false);
        }
      } else if (p.stats_mode == 0) {
        if (p.act == SIHL_ACT_RELU) PW_EDGE(SIHL_ACT_RELU, 0);
        else if (p.act == SIHL_ACT_SILU) PW_EDGE(SIHL_ACT_SILU, 0);
        else PW_EDGE(SIHL_ACT_NONE, 0);
      } else if (p.stats_mode == 1) {
        PW_EDGE(SIHL_ACT_NONE, 1);
      } else {
        if (p.act == SIHL_ACT_RELU) PW_EDGE(SIHL_ACT_RELU, 2);
        else if (p.act == SIHL_ACT_SILU) PW_EDGE(SIHL_ACT_SILU, 2);
        else PW_EDGE(SIHL_ACT_NONE, 2);
      }
#undef PW_FULL
#undef PW_EDGE
      pw_barrier();
      if (p.stats_mode) {  // one partial row per 128-pixel tile, [tile_m][2][Cout]: thread -> (sum | sumsq, channel)
        const int which = tid >> 7, ch = tid & 127;
        const float t = red[(which * 2 + 0) * PW_BN + ch] + red[(which * 2 + 1) * PW_BN + ch];
        pw_store4(t, (unsigned)((((long)tile_m * 2 + which) * p.Cout + n0 + ch) * 4L), st_rsrc);
      }
      {  // 16-byte row-contiguous stores: 256 B of each tile row; thread -> (row tid/16 + 16 it, chunk tid%16)
        const int row0 = tid >> 4, ch = tid & 15;
        const unsigned dst = (unsigned)(((long)(m0 + row0) * p.Cout + n0 + ch * 8) * 2L);
        const unsigned step = 16u * (unsigned)p.Cout * 2u;
        const char* src = epi + row0 * PW_EPI_STRIDE + ch * 16;
        // all eight LDS reads first: the asm stores clobber "memory", a read placed after one could not move above it
        // (one LDS round trip per store otherwise: +1000 cycles per tile)
        u32x4_t v[PW_BM / 16];
#pragma unroll
        for (int it = 0; it < PW_BM / 16; ++it) v[it] = *(const u32x4_t*)(src + it * 16 * PW_EPI_STRIDE);
        if (p.dbg & 512) {  // experiment: compiler-generated global stores (conservative waits only)
          char* o = (char*)p.out + dst;
#pragma unroll
          for (int it = 0; it < PW_BM / 16; ++it)
            if (m0 + row0 + it * 16 < p.M) *(u32x4_t*)(o + (long)it * step) = v[it];
        } else {
#pragma unroll
        for (int it = 0; it < PW_BM / 16; ++it) pw_store16(v[it], dst + it * step, out_rsrc);
        }
      }
      epi_hist |= 1u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    tile_m += nslots;
  }
  wait_vm_keep<0>();  // the trailing (out-of-range) ring requests have retired before the LDS is released
}

int g_cus = 0;

}  // namespace

void sihl_pw_set_enabled(int mode) { g_pw = mode; }

// ring depth for a contraction of Cin channels: whatever fits next to the resident weights and the staging tile
static int pw_ring(int Cin) {
  const int kc = Cin / 64;
  int nr = (160 * 1024 - PW_EPI - kc * PW_CHUNK) / PW_CHUNK;
  return nr > 6 ? 6 : nr;
}

bool sihl_pw_eligible(const ConvParams& p) {
  if (!(g_pw & 3) || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.in_dilate != 1 || p.splits != 1) return false;
  if (p.Cin % 64 || p.Cin < 64 || p.Cin > 256 || p.Cout % PW_BN) return false;
  if (p.add || p.pre_scale || p.post_scale || p.ln_gamma || p.act == SIHL_ACT_SIGMOID) return false;
  if (p.stats_mode && !p.stats) return false;
  if (p.out_s != 1 || p.out_image_stride != (long)p.Ho * p.Wo * p.Cout || p.Ho != p.H || p.Wo != p.W) return false;
  if (((long)p.M + PW_BM) * p.Cin * 2L >= (1L << 31) || (long)p.Cout * p.Cin * 2L >= (1L << 31)) return false;
  if (((long)p.M + PW_BM) * p.Cout * 2L >= (1L << 31)) return false;  // 32-bit buffer offsets of the output stores
  const int gridN = p.Cout / PW_BN;
  if (gridN > 32 || (gridN & (gridN - 1))) return false;  // channel tiles of a pixel slot share an XCD: 256 % (8 gridN) == 0
  if (pw_ring(p.Cin) < 3) return false;
  const long tiles_m = ((long)p.M + PW_BM - 1) / PW_BM;
  return (g_pw & 2) || tiles_m * gridN >= 4 * 256;  // at least ~4 tiles per workgroup: below that the one-tile kernel's wider grid wins
}

int sihl_pw_launch(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  if (!g_cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return SIHL_EARG;
    g_cus = n;
  }
  p.gridN = p.Cout / PW_BN;
  p.dbg = (p.dbg & 0xff) | ((g_pw & 4) ? 256 : 0) | ((g_pw & 8) ? 512 + 256 : 0);  // experiment: conservative ring waits
  const int tiles_m = (p.M + PW_BM - 1) / PW_BM;
  p.gridM = tiles_m;
  int nwg = g_cus - g_cus % (8 * p.gridN);  // one workgroup per CU; a multiple of 8 XCDs x gridN channel tiles
  if (nwg <= 0) return SIHL_EARG;
  const int nslots = nwg / p.gridN;
  const int nr = pw_ring(p.Cin);
  const int lds = (p.Cin / 64 + nr) * PW_CHUNK + PW_EPI;
  const double flops = 2.0 * p.M * (double)p.Cout * p.Cin;
  const double bytes = ((double)p.M * p.Cin + (double)p.M * p.Cout + (double)p.Cout * p.Cin) * 2.0;
#define SIHL_PW(NR_)                                                                                              \
  do {                                                                                                            \
    static bool attr = false;                                                                                     \
    if (!attr) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)conv_pw_kernel<NR_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         160 * 1024);                                                             \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr = true;                                                                                                \
    }                                                                                                             \
    sihl_prof_begin(SIHL_PROF_CONV, SIHL_BF16, flops, bytes, stream);                                             \
    hipLaunchKernelGGL(conv_pw_kernel<NR_>, dim3(nwg), dim3(256), lds, stream, p, nslots, tiles_m);               \
    sihl_prof_end(stream);                                                                                        \
  } while (0)
  if (nr == 6) SIHL_PW(6);
  else if (nr == 5) SIHL_PW(5);
  else if (nr == 4) SIHL_PW(4);
  else SIHL_PW(3);
#undef SIHL_PW
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}
