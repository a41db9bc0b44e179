// 3x3 / stride 1 / pad 1 convolution of the LARGE pyramid levels with a HALO-RESIDENT input patch (bf16; 256 -> 256 channels on
// 64-wide maps: BiFPN P3 at 512^2 - reference src/sihl/layers/bifpn.py:39-53 through convblocks.py:37-87 - forward, and input
// gradient with the flipped / transposed weights).  Round 4, review item 2.
//
// The general tile (conv_igemm_impl.h) walks K in 36 stages of (one tap) x (64 channels): per stage 32 KiB of input rows AND
// 32 KiB of weights come in by LDS-DMA, one barrier each, and every input pixel is fetched nine times - 4.6 MB of L2 -> LDS
// ingest per CU for the two tiles it owns, 64 DMA wave-instructions and a barrier per 2 048 matrix cycles.  Its stamps and
// ablations (profiles/r01_s2_big_tile_ablation.txt, r02_p8_ablation.txt) put the loss in exactly that skeleton - barrier skew and
// DMA issue per K-tile - not in data latency.  Here the same 256 pixel x 256 channel tile, the same 16 waves of 64 x 64 and the
// same epilogue run on another K walk:
//   * the tile is 4 whole image rows; its input PATCH (6 rows x 66 columns at a pitch of 72, zeros outside the image) is resident
//     in LDS per 32-channel chunk - fetched once (27 KiB) instead of nine times (9 x 16 KiB); the nine taps read it at shifted
//     addresses;
//   * a stage is one KERNEL ROW of one chunk: 3 taps x 256 out-channels x 64 B = 48 KiB of weights; 24 stages and 24 barriers
//     per tile instead of 36, each over 3 072 matrix cycles instead of 2 048; per stage 48 + 8.4 KiB come in (57 DMA
//     wave-instructions per 3 072 cycles against 64 per 2 048: -42 % per flop);
//   * patch and weights double-buffered: 2 x 27 + 2 x 48 = 150 KiB;
//   * rows are 64 B (32 channels): a 16-lane ds_read_b128 group then spans 16 rows, and bit 1 of the 16-byte piece index is
//     flipped where bit 2 of the row index is set - on the DMA source side and on the read - which is conflict-free for every
//     alignment of 16 consecutive rows (brute-forced), i.e. for every tap shift of the patch and for the weight rows.
#include "conv_igemm_impl.h"

namespace {

constexpr int HKCB = 64;                   // bytes of K per row and chunk (32 bf16 channels)
constexpr int HBN = 256;                   // out-channels per workgroup
constexpr int HB_STAGE = 3 * HBN * HKCB;   // weights of one kernel row of one chunk: 48 KiB
constexpr int HB_PIECES = HB_STAGE / 1024; // 48 LDS-DMA wave-instructions (16 rows x 64 B each)

template <int W, int BM> struct HaloGeo {
  static constexpr int R = BM / W;                 // image rows per tile
  static constexpr int PW = (W + 2 + 7) / 8 * 8, PR = R + 2;  // patch row pitch: W + 2 pixels padded to a multiple of 8 - the
  // swizzle term of a patch pixel (bit 2 of its index) is then the same for every kernel row, and the LDS offset of a lane's
  // fragment for (m-tile, kx) is a per-launch constant: the stage adds one scalar.  (With the pitch at 66 the six VALU
  // instructions per fragment address - 72 per wave and stage, four waves per SIMD - competed with the MFMAs for issue slots.)
  static constexpr int NP = PR * PW;               // patch pixels (incl. the padding columns)
  static constexpr int PIECES = (NP + 15) / 16;    // 1 KiB pieces of 16 pixels x 64 B
  static constexpr int A_BYTES = PIECES * 1024;
  static constexpr int NW = BM / 64 * 4;           // waves: (BM / 64) x 4, each 64 pixels x 64 channels
  static constexpr int APW = (PIECES + NW - 1) / NW, BPW = HB_PIECES / NW;
  static constexpr int BUF = A_BYTES + HB_STAGE;
  static constexpr int EPI = BM * (HBN * 2 + 16) + 2 * (BM / 64) * HBN * 4;
  static constexpr int LDS = 2 * BUF > EPI ? 2 * BUF : EPI;
  static_assert(HB_PIECES % NW == 0, "weight pieces per wave");
};

__device__ __forceinline__ int hswz(int row) { return ((row >> 2) & 1) << 1; }

#ifdef SIHL_HALO_STAMPS
unsigned long long* g_halo_stamps = nullptr;
#define HALO_T(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HALO_T(x) do {} while (0)
#endif

template <int W, int BM, bool MIDBAR>
__global__ __launch_bounds__((BM / 64 * 4 * 64), (BM / 64)) void conv_halo_kernel(const ConvParams p) {
  using Geo = HaloGeo<W, BM>;
  constexpr int WM = BM / 64, WN = 4, NW = Geo::NW, TILE = 16, MT = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef SIHL_HALO_STAMPS
  unsigned long long t_entry = 0;
  HALO_T(t_entry);
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, g = lane >> 4;
  const int L = xcd_remap(blockIdx.x, p.gridM * p.gridN);
  const int tile_m = L / p.gridN, tile_n = L % p.gridN;
  const int m0 = tile_m * BM, n0 = tile_n * HBN;
  const int tiles_per_image = p.H / Geo::R;
  const int img = tile_m / tiles_per_image, y0 = (tile_m - img * tiles_per_image) * Geo::R;
  const int nch = p.Cin >> 5, nstages = nch * 3;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * p.H * W * p.Cin * 2));
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * 9 * p.Cin * 2));
  // every workgroup walks the channel chunks from another start (ConvParams::k_rotate: all of them read the same weights);
  // whole chunks, so that a chunk's three stages - which share its patch - stay together
  const int rot = p.k_rotate ? 3 * (int)(((unsigned)L * 5u) % (unsigned)nch) : 0;

  // ---- weight pieces: piece j = wave + NW * u holds rows 16 j .. 16 j + 15 of the stage's [kx][256 out-channels] x 64 B
  unsigned b_off[Geo::BPW];
#pragma unroll
  for (int u = 0; u < Geo::BPW; ++u) {
    const int row = (wave + NW * u) * 16 + (lane >> 2), pos = lane & 3;
    const int kx = row >> 8, co = row & 255;
    b_off[u] = (unsigned)((((long)(n0 + co) * 9 + kx) * p.Cin) * 2 + ((pos ^ hswz(co)) << 4));
  }
  // ---- patch pieces: piece q = wave + NW * j holds patch pixels 16 q .. 16 q + 15
  unsigned a_off[Geo::APW];
  unsigned a_ok = 0;
#pragma unroll
  for (int j = 0; j < Geo::APW; ++j) {
    const int P = (wave + NW * j) * 16 + (lane >> 2), pos = lane & 3;
    const int pr = P / Geo::PW, pc = P - pr * Geo::PW;
    const int iy = y0 - 1 + pr, ix = pc - 1;
    const bool ok = P < Geo::NP && iy >= 0 && iy < p.H && ix >= 0 && ix < W;  // (padding columns: ix >= W)
    a_off[j] = (unsigned)((((long)img * p.H + iy) * W + ix) * p.Cin * 2 + ((pos ^ hswz(P)) << 4));
    a_ok |= ok ? (1u << j) : 0u;
  }
  auto stage_of = [&](int s, int& cc, int& ky) {  // rotated stage index -> (chunk, kernel row)
    int t = s + rot;
    if (t >= nstages) t -= nstages;
    cc = t / 3;
    ky = t - 3 * cc;
  };
  auto issue_b = [&](int s, int buf) {
    int cc, ky;
    stage_of(s, cc, ky);
    const unsigned dst = lds_base + buf * Geo::BUF + Geo::A_BYTES + wave * 1024;
    const unsigned delta = (unsigned)((ky * 3 * p.Cin) * 2 + cc * HKCB);
#pragma unroll
    for (int u = 0; u < Geo::BPW; ++u) dma16(b_off[u] + delta, dst + u * NW * 1024, wt_rsrc);
  };
  auto issue_a = [&](int cc, int buf) {
    const unsigned dst = lds_base + buf * Geo::BUF + wave * 1024;
    const unsigned delta = (unsigned)(cc * HKCB);
#pragma unroll
    for (int j = 0; j < Geo::APW; ++j)
      if (wave + NW * j < Geo::PIECES) dma16(((a_ok >> j) & 1) ? a_off[j] + delta : OOB, dst + j * NW * 1024, in_rsrc);
  };
  // The patch of a chunk serves the chunk's three stages: the stage sequence visits the chunks in runs of three, run r uses
  // patch buffer r & 1.
  int cc0, ky0;
  stage_of(0, cc0, ky0);
  issue_b(0, 0);
  issue_a(cc0, 0);

  // ---- fragment geometry
  int aoff[MT][3];  // LDS byte offset of this lane's fragment of m-tile i for tap column kx, kernel row 0
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int pix = wm * 64 + i * 16 + l16;
    const int r = pix / W, c = pix - r * W;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int q = r * Geo::PW + c + kx;
      aoff[i][kx] = q * HKCB + ((g ^ hswz(q)) << 4);
    }
  }
  int boff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * 64 + j * 16 + l16;
    boff[j] = row * HKCB + ((g ^ hswz(row)) << 4);
  }

  typename AccTile<TILE>::type acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // patch run of a stage (stages come in runs of three that share a chunk's patch): with whole-chunk rotation run = s / 3
  auto frag_a = [&](const char* As, int ky, int kx, uint4 (&fa)[MT]) {
    const char* base = As + ky * (Geo::PW * HKCB);  // (PW % 8 == 0: the swizzle inside aoff holds for every kernel row)
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = *(const uint4*)(base + aoff[i][kx]);
  };
  auto frag_b = [&](const char* Bs, int kx, uint4 (&fb)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *(const uint4*)(Bs + kx * (HBN * HKCB) + boff[j]);
  };
  auto mma_all = [&](const uint4 (&fa)[MT], const uint4 (&fb)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) mma_step<bf16_t, TILE>(acc[i][j], fa[i], fb[j]);
  };
#ifdef SIHL_HALO_STAMPS
  unsigned long long t_start = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, sum_wait = 0, sum_bar = 0, sum_comp = 0;
  HALO_T(t_start);
#endif
  if constexpr (MIDBAR) {
    // The stage barrier sits between the LAST tap's fragment reads and its multiplies.  In-kernel stamps of the plain form
    // (barrier, then the stage: tools/halo_stamps.py) showed 4 430 cycles per stage against 3 072 matrix cycles of the SIMD, the
    // DMA data always there (1-3 % wait) - and the four waves of a SIMD 45 / 24 / 19 / 3 % of their time at the barrier: behind it
    // all sixteen waves read their first fragments at once (128 KiB through a 256 B/clk LDS) with nothing in the matrix pipe.
    // Here every wave crosses the barrier with 16 multiplies in hand, the next stage's first reads queue behind them, and the
    // DMA of stage s + 2 (into the buffer the barrier has just freed) goes out behind those multiplies.
    if (nstages > 1) issue_b(1, 1);
    wait_vm_keep<0>();
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
      const int buf = s & 1, run = s / 3;
      int cc, ky;
      stage_of(s, cc, ky);
      HALO_T(t2);
      const char* As = smem + (run & 1) * Geo::BUF;
      const char* Bs = smem + buf * Geo::BUF + Geo::A_BYTES;
      uint4 fa[MT], fb[NT];
#pragma unroll
      for (int kx = 0; kx < 2; ++kx) {
        frag_a(As, ky, kx, fa);
        frag_b(Bs, kx, fb);
        mma_all(fa, fb);
      }
      frag_a(As, ky, 2, fa);
      frag_b(Bs, 2, fb);
      HALO_T(t0);
      wait_vm_keep<0>();  // stage s + 1 (issued a stage ago) has landed for this wave ...
      HALO_T(t1);
      __syncthreads();    // ... and for everyone; and everyone has read the last of stage s (the fragments are in registers)
      HALO_T(t3);
      mma_all(fa, fb);
      if (s + 2 < nstages) {
        issue_b(s + 2, buf);
        if ((s + 2) % 3 == 0) {  // stage s + 2 opens a new run: its patch goes to the buffer the run before this one used
          int ncc, nky;
          stage_of(s + 2, ncc, nky);
          issue_a(ncc, ((s + 2) / 3) & 1);
        }
      }
#ifdef SIHL_HALO_STAMPS
      sum_wait += t1 - t0; sum_bar += t3 - t1; sum_comp += t0 - t2;
#endif
    }
  } else {
    for (int s = 0; s < nstages; ++s) {
      const int buf = s & 1, run = s / 3;
      int cc, ky;
      stage_of(s, cc, ky);
      HALO_T(t0);
      wait_vm_keep<0>();  // this wave's pieces of stage s (and of its patch, if it is new) have landed ...
      HALO_T(t1);
      __syncthreads();    // ... and everyone's; everyone is done with stage s - 1's weights and, at a run's start, the previous patch
      HALO_T(t2);
      const char* As = smem + (run & 1) * Geo::BUF;
      const char* Bs = smem + buf * Geo::BUF + Geo::A_BYTES;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        uint4 fa[MT], fb[NT];
        frag_a(As, ky, kx, fa);
        frag_b(Bs, kx, fb);
        mma_all(fa, fb);
        // the next stage's DMA goes out behind the first tap's multiplies: issued right after the barrier (16 waves x 3-5 DMA
        // instructions at ~100 cycles each) it delayed every wave's first fragment reads - 141.6 against 137.1 us on P3
        // (tools/halo_probe.py, interleaved medians; behind the second tap: 139.0)
        if (kx == 0 && s + 1 < nstages) {
          issue_b(s + 1, buf ^ 1);
          if ((s + 1) % 3 == 0) {
            int ncc, nky;
            stage_of(s + 1, ncc, nky);
            issue_a(ncc, ((s + 1) / 3) & 1);
          }
        }
      }
#ifdef SIHL_HALO_STAMPS
      HALO_T(t3);
      sum_wait += t1 - t0; sum_bar += t2 - t1; sum_comp += t3 - t2;
#endif
    }
  }
#ifdef SIHL_HALO_STAMPS
  HALO_T(t3);
  if (p.partial && blockIdx.x == 0 && lane == 0) {  // (diagnostic build: the stamps ride in the unused split-K workspace pointer)
    unsigned long long* o = (unsigned long long*)p.partial + wave * 4;
    o[0] = t3 - t_start; o[1] = sum_wait; o[2] = sum_bar; o[3] = sum_comp;
  }
#endif
  __syncthreads();  // everyone is done reading the last stage: LDS is free for the epilogue's transpose
  conv_epilogue<bf16_t, BM, HBN, WM, WN, TILE>(p, acc, smem, tile_m, m0, n0);
#ifdef SIHL_HALO_STAMPS
  {
    unsigned long long t_end;
    HALO_T(t_end);
    if (p.partial && blockIdx.x == 0 && lane == 0) {
      unsigned long long* o = (unsigned long long*)p.partial + 64 + wave * 2;
      o[0] = t_start - t_entry; o[1] = t_end - t3;  // prologue (entry -> K loop), epilogue (K loop end -> stores issued)
    }
  }
#endif
}

// test hook (sihl_conv2d_halo_enable): 0 = off, 1 = where the grid fills the chip, 2 = wherever the shape allows, 3 = as 2 with
// the plain loop form (stage barrier in front of the stage)
int g_halo = 1;

template <int W, int BM>
int launch_halo(const ConvParams& p0, hipStream_t stream) {
  using Geo = HaloGeo<W, BM>;
  static_assert(Geo::LDS <= 160 * 1024, "LDS budget");
  ConvParams p = p0;
  p.gridM = p.M / BM;
  p.gridN = p.Cout / HBN;
  p.splits = 1;
  p.k_rotate = g_krot % 1000 != 0;
  const bool midbar = g_halo != 3;  // (test hook value 3: the plain form - barrier in front of the stage - kept as the A/B arm)
  auto kern = midbar ? conv_halo_kernel<W, BM, true> : conv_halo_kernel<W, BM, false>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_halo_kernel<W, BM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)conv_halo_kernel<W, BM, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const double flops = 2.0 * p.M * (double)p.Cout * 9 * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * 9 * p.Cin) * 2.0;
  sihl_prof_begin(SIHL_PROF_CONV, SIHL_BF16, flops, bytes, stream);
  hipLaunchKernelGGL(kern, dim3(p.gridM * p.gridN), dim3(Geo::NW * 64), Geo::LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // namespace

void sihl_halo_set_mode(int mode) { g_halo = mode; }

bool sihl_halo_eligible(const ConvParams& p) {
  if (!g_halo || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.in_dilate != 1) return false;
  if ((p.W != 64 && p.W != 32) || p.H % 4 || p.Ho != p.H || p.Wo != p.W || p.Cin % 32 || p.Cout % HBN) return false;
  if (p.add || p.out_s != 1 || p.out_image_stride != (long)p.Ho * p.Wo * p.Cout) return false;
  if (p.w_ntaps != 9 || p.w_kw != 3 || p.w_ky0 || p.w_kx0 || p.w_kys != 1 || p.w_kxs != 1) return false;
  if ((long)p.N * p.H * p.W * p.Cin * 2 >= (1L << 31) || (long)p.Cout * 9 * p.Cin * 2 >= (1L << 31)) return false;
  // at least one tile per CU (else the narrower general tiles fill the chip better)
  const long per = p.W == 64 ? 256 : 128;  // pixels of the smallest tile for this width
  return g_halo >= 2 || (long)p.M / per * (p.Cout / HBN) >= 256;
}

int sihl_halo_launch(const ConvParams& p, hipStream_t stream) {
  if (p.W == 32) {
    // 256 pixels (8 rows of 32) on 16 waves where that still gives every CU a tile, else 128 pixels (4 rows) on 8 waves (P4 at
    // batch 32: 256 workgroups; at batch 64 the 128-pixel form was slower than the general 256 x 256 tile, 77.8 against 71.1 us)
    if (p.H % 8 == 0 && (long)p.M / 256 * (p.Cout / HBN) >= 256) return launch_halo<32, 256>(p, stream);
    return launch_halo<32, 128>(p, stream);
  }
  if ((long)p.M / 256 * (p.Cout / HBN) < 256 && g_halo < 2) return SIHL_EARG;  // (sihl_halo_eligible has ruled this out)
  return launch_halo<64, 256>(p, stream);
}
