// 3x3 / stride 1 / pad 1 convolution of the SMALL pyramid levels (bf16; square maps 16x16, 8x8, 4x4: BiFPN P5-P7 at
// 512^2, reference src/sihl/layers/bifpn.py:39-53 through convblocks.py:37-87) - forward, and input gradient with the
// flipped / transposed weights.
//
// These launches are 3 % of the north-star forward's flops and were 20 % of its time (profiles/r03_ns_forward_summary.txt:
// 28 us for 9.7 GFLOP on P5, 14 + 5 us for 2.4 GFLOP on P6): the general kernel walks 36 (tap, 64-channel) stages with a
// barrier each and re-fetches a pixel's channels once per tap (9 x), and the levels that need split-K to fill the chip pay a
// second launch to add the slices up.  Here:
//   * the INPUT PATCH of a 128-pixel tile - its rows plus a one-pixel halo, zero outside the image - is resident in LDS per
//     64-channel chunk: fetched once by LDS-DMA (22-36 KiB) instead of nine times, the nine taps read it at shifted
//     addresses.  Tile = 128 consecutive output pixels = 8 rows of a 16-wide map, two 8x8 maps or eight 4x4 maps.
//     The 16-byte chunks of a patch pixel are XOR-swizzled by bits of its patch COLUMN and ROW (not of its linear index):
//     a ds_read_b128 lane group spans two to four image rows, and this keeps its 16 lanes on 16 different bank slots for
//     every tap shift.
//   * weights stream through a 3-slot ring of (one kernel ROW = 3 taps) x 64 out-channels x 64 channels (24 KiB): 3 stages
//     and 3 barriers per chunk instead of 9.
//   * MFMA 32x32x16 with the WEIGHT fragment as the first operand: a lane holds 4 consecutive channels of one pixel.
//   * split-K over the channel chunks (levels with <= 2048 pixels): every slice stores its fp32 tile and the general
//     kernel's finishing launch (conv_splitk_epilogue_kernel) adds the slices in slice order and runs the epilogue.
//     (Tried and dropped: finishing INSIDE the launch - sc1 stores of the slices, one agent-scope ticket per tile, the
//     workgroup that draws the last ticket adds the slices.  Correct and deterministic, but on this chip the chain store ->
//     write acknowledge -> atomic round trip -> sc1 loads costs 12 us per launch (P6: 17.9 us against 5.5 us for the
//     multiply alone), more than the finishing launch's 5 us: kernel boundaries are the cheaper coherence point here.)
//   * an unsplit launch (P5) runs the epilogue itself (bias -> [stats] -> affine -> act -> [stats] -> affine, BatchNorm
//     partial row per 128 pixels, as conv_igemm_impl.h) on 8-channel vectors with 16-byte output stores.
//
// Not for the thin 3x3 convs of the ResNet trunk (tried, round 3: instantiated for 128x128 maps with 64 channels and 64x64 with
// 128, parity-green): 148 us against 80-89 for the general kernel on layer1's 64 -> 64, 98 against 50 on layer2's 128 -> 128.
// At thousands of tiles per launch the general kernel's small single-stage tiles (3-4 workgroups per CU hiding each other's
// load and store phases) beat one 125-147 KiB workgroup per CU that loads its patch, then multiplies.
#include "common.h"
#include "conv_params.h"
#include "conv_tuning.h"
#include "dma.h"
#include "profile.h"

namespace {

constexpr int SBM = 128, SBN = 64, SKCB = 128, STHREADS = 256;
constexpr int SB_STAGE = 3 * SBN * SKCB;  // one kernel row: 3 taps x 64 out-channels x 128 B
constexpr int SB_PIECES = SB_STAGE / 1024 / 4;  // LDS-DMA wave-instructions per wave per weight stage (6)
constexpr int SB_RING = 3 * SB_STAGE;
constexpr int STILE_STRIDE = SBN * 4 + 16;  // fp32 tile row in LDS (finisher staging of an unsplit launch)

template <int W> struct SmallGeo {
  static constexpr int HW = W * W;
  static constexpr int G = HW >= SBM ? 1 : SBM / HW;  // maps (segments) per tile
  static constexpr int SR = HW >= SBM ? SBM / W : W;   // output rows per segment
  static constexpr int PW = W + 2, PR = SR + 2;        // patch columns / rows per segment
  static constexpr int NP = G * PR * PW;               // patch pixels
  static constexpr int PIECES = (NP + 7) / 8;          // 1 KiB LDS-DMA pieces (8 pixels x 128 B)
  static constexpr int APW = (PIECES + 3) / 4;         // pieces per wave
  static constexpr int A_BYTES = APW * 4 * 1024;
  static constexpr int NB = W >= 16 ? 3 : (W == 8 ? 2 : 1);  // swizzle bits taken from the patch column
  static constexpr int LDS = 2 * A_BYTES + SB_RING;
  __device__ static __forceinline__ int swz(int pr, int pc) {
    return ((pc >> 1) & ((1 << NB) - 1)) | ((pr & ((1 << (3 - NB)) - 1)) << NB);
  }
};


template <int W>
__global__ __launch_bounds__(STHREADS, 1) void conv_small_kernel(const ConvParams p) {
  using Geo = SmallGeo<W>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  const int nsplit = p.splits, nch = p.small_nch;
  int b = blockIdx.x;
  const int split = b % nsplit; b /= nsplit;
  const int tile_n = b % p.gridN, tile_m = b / p.gridN;
  const int m0 = tile_m * SBM, n0 = tile_n * SBN;
  const int kc0 = split * nch;
  const int img0 = m0 / Geo::HW, y0 = Geo::G == 1 ? (m0 % Geo::HW) / W : 0;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  const unsigned ring_base = lds_base + 2 * Geo::A_BYTES;
  const v4i_t in_rsrc = make_rsrc(p.in, (unsigned)((long)p.N * Geo::HW * p.Cin * 2));
  const v4i_t wt_rsrc = make_rsrc(p.wt, (unsigned)((long)p.Cout * 9 * p.Cin * 2));

  // ---- patch pieces of this wave: piece q = wave * APW + j holds patch pixels 8 q .. 8 q + 7, 128 B each
  unsigned a_off[Geo::APW];
  unsigned a_ok = 0;
#pragma unroll
  for (int j = 0; j < Geo::APW; ++j) {
    const int P = (wave * Geo::APW + j) * 8 + (lane >> 3), pos = lane & 7;
    const int seg = P / (Geo::PR * Geo::PW), r = P - seg * (Geo::PR * Geo::PW);
    const int pr = r / Geo::PW, pc = r - pr * Geo::PW;
    const int n = img0 + seg, iy = y0 - 1 + pr, ix = pc - 1;
    const bool ok = P < Geo::NP && n < p.N && iy >= 0 && iy < W && ix >= 0 && ix < W;
    a_off[j] = (unsigned)((((long)n * W + iy) * W + ix) * p.Cin * 2 + ((pos ^ Geo::swz(pr, pc)) << 4));
    a_ok |= ok ? (1u << j) : 0u;
  }
  // ---- weight pieces: stage rows = 3 taps (kx) x 64 out-channels
  unsigned b_off[SB_PIECES];
#pragma unroll
  for (int j = 0; j < SB_PIECES; ++j) {
    const int row = (wave * SB_PIECES + j) * 8 + (lane >> 3), pos = lane & 7;
    const int kx = row >> 6, col = row & 63;
    b_off[j] = (unsigned)((((long)(n0 + col) * 9 + kx) * p.Cin) * 2 + ((pos ^ ((col >> 1) & 7)) << 4));
  }
  // Every workgroup of a launch reads the same weights; started together they would request the same 24 KiB stage at the
  // same moment.  Workgroup (tile_m, tile_n) therefore walks its chunks from chunk rc on and its kernel rows from row rk on
  // (wrapping around): the fp32 sum of a tile is taken in another order, tile by tile (deterministic).
  const int rot = p.k_rotate ? tile_m * p.gridN + tile_n : 0;
  const int rc = rot % nch, rk = (rot / nch) % 3;
  auto issue_a = [&](int c, int buf) {  // channel chunk kc0 + (c + rc) % nch of the patch
    const unsigned dst = lds_base + buf * Geo::A_BYTES + wave * Geo::APW * 1024;
    const int cc = c + rc >= nch ? c + rc - nch : c + rc;
    const unsigned delta = (unsigned)((kc0 + cc) * SKCB);
#pragma unroll
    for (int j = 0; j < Geo::APW; ++j) dma16(((a_ok >> j) & 1) ? a_off[j] + delta : OOB, dst + j * 1024, in_rsrc);
  };
  auto issue_b = [&](int s) {  // stage s = (chunk s / 3, kernel row s % 3) into ring slot s % 3
    const int c = s / 3;
    int ky = s - 3 * c + rk, cc = c + rc;
    if (ky >= 3) ky -= 3;
    if (cc >= nch) cc -= nch;
    const unsigned dst = ring_base + (s % 3) * SB_STAGE + wave * SB_PIECES * 1024;
    const unsigned delta = (unsigned)((ky * 3 * p.Cin) * 2 + (kc0 + cc) * SKCB);
#pragma unroll
    for (int j = 0; j < SB_PIECES; ++j) dma16(b_off[j] + delta, dst + j * 1024, wt_rsrc);
  };

  // ---- this lane's output pixel inside the patch (tap (0, 0) position)
  const int ml = wave * 32 + fr;
  const int seg_l = ml / (Geo::SR * W), wi = ml - seg_l * (Geo::SR * W);
  const int pr0 = wi / W, pc0 = wi - pr0 * W;
  const int pbase = seg_l * (Geo::PR * Geo::PW) + pr0 * Geo::PW + pc0;

  f32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int nstages = nch * 3;
  if (!(SIHL_DBG(p) & 1)) {
    issue_a(0, 0);
    issue_b(0);
    issue_b(1);
  }
  int c = 0, ky = 0;
  for (int s = 0; s < nstages; ++s) {
    // stage s (and, at a chunk's first stage, its patch) has landed; what may still be in flight was issued behind it:
    // the next weight stage and, during a chunk's 2nd and 3rd stage, the next chunk's patch
    const bool more_b = s + 1 < nstages, a_pending = ky != 0 && c + 1 < nch;
    if (more_b) {
      if (a_pending) wait_vm_keep<SB_PIECES + Geo::APW>();
      else wait_vm_keep<SB_PIECES>();
    } else {
      wait_vm_keep<0>();
    }
    __syncthreads();  // ... for every wave; and everyone is done with the slot / patch buffer refilled below
    if (s + 2 < nstages && !(SIHL_DBG(p) & 1)) issue_b(s + 2);
    if (ky == 0 && c + 1 < nch && !(SIHL_DBG(p) & 1)) issue_a(c + 1, (c + 1) & 1);
    const char* As = smem + (c & 1) * Geo::A_BYTES;
    const char* Bs = smem + 2 * Geo::A_BYTES + (s % 3) * SB_STAGE + fr * SKCB;
    const int kyr = ky + rk >= 3 ? ky + rk - 3 : ky + rk;  // the kernel row this stage's weights belong to
    if (!(SIHL_DBG(p) & 2))
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const char* ap = As + (pbase + kyr * Geo::PW + kx) * SKCB;
      const int f = Geo::swz(pr0 + kyr, pc0 + kx);
      const char* bp = Bs + kx * SBN * SKCB;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 fa = *(const uint4*)(ap + (((ks * 2 + fh) ^ f) << 4));
        const uint4 w0 = *(const uint4*)(bp + (((ks * 2 + fh) ^ fsw) << 4));
        const uint4 w1 = *(const uint4*)(bp + 32 * SKCB + (((ks * 2 + fh) ^ fsw) << 4));
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, fa), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, fa), acc[1], 0, 0, 0);
      }
    }
    if (++ky == 3) { ky = 0; ++c; }
  }

  // ---- this slice's fp32 tile: lane (fr, fh) holds, for pixel m0 + wave * 32 + fr, channels 32 i + 8 g + 4 fh + {0..3}
  if (SIHL_DBG(p) & 32) return;  // tuning ablation: no epilogue at all
  if (nsplit > 1) {  // the finishing launch adds the slices up
    const int m = m0 + ml;
    if (m < p.M) {
      float* part = p.partial + ((long)split * p.M + m) * p.Cout + n0 + 4 * fh;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(part + i * 32 + 8 * g) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    }
    return;
  }
  __syncthreads();  // everyone is done reading the patch and the ring: LDS is free for the staging tile
  {
    char* row = smem + ml * STILE_STRIDE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(row + (i * 32 + 8 * g + 4 * fh) * 4) = make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    __syncthreads();
  }

  // ---- epilogue: thread = 8 consecutive channels (cg) of pixels prow, prow + 32, prow + 64, prow + 96 of the tile
  const int cg = tid & 7, prow = tid >> 3;
  const int co = n0 + cg * 8;
  const bool has_pre = p.pre_scale != nullptr, has_post = p.post_scale != nullptr;
  float bias[8], s1[8], t1[8], s2[8], t2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bias[e] = p.bias ? p.bias[co + e] : 0.f;
    s1[e] = has_pre ? p.pre_scale[co + e] : 1.f;
    t1[e] = (has_pre && p.pre_shift) ? p.pre_shift[co + e] : 0.f;
    s2[e] = has_post ? p.post_scale[co + e] : 1.f;
    t2[e] = (has_post && p.post_shift) ? p.post_shift[co + e] : 0.f;
  }
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
  bf16_t* __restrict__ out = (bf16_t*)p.out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = prow + 32 * k, m = m0 + pl;
    const bool ok = m < p.M;
    float v[8];
    {
      const float4 a = *(const float4*)(smem + pl * STILE_STRIDE + cg * 32), c4 = *(const float4*)(smem + pl * STILE_STRIDE + cg * 32 + 16);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] + bias[e];
      if (p.stats_mode == 1 && ok) { ssum[e] += x; ssq[e] += x * x; }
      x = x * s1[e] + t1[e];
      if (p.act == SIHL_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == SIHL_ACT_SILU) x = x / (1.f + expf(-x));
      else if (p.act == SIHL_ACT_SIGMOID) x = 1.f / (1.f + expf(-x));
      if (p.stats_mode == 2 && ok) { ssum[e] += x; ssq[e] += x * x; }
      o[e] = x * s2[e] + t2[e];
    }
    if (ok) *(uint4*)(out + (long)m * p.Cout + co) = pack16(o, bf16_t());
  }
  if (p.stats_mode) {  // one partial row per 128-pixel tile: column sums over the 32 threads of a channel group, fixed order
    __syncthreads();  // the staging tile has been read
    float* red = (float*)smem;  // [2][32][64]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * 32 + prow) * 64 + cg * 8 + e] = ssum[e];
      red[(1 * 32 + prow) * 64 + cg * 8 + e] = ssq[e];
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, col = tid & 63;
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) a += red[(which * 32 + r) * 64 + col];
      p.stats[((long)tile_m * 2 + which) * p.Cout + n0 + col] = a;
    }
  }
}

bool g_small = true;

template <int W>
int launch_small(const ConvParams& p, hipStream_t stream) {
  using Geo = SmallGeo<W>;
  static_assert(Geo::LDS <= 160 * 1024 && SBM * STILE_STRIDE <= Geo::LDS, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_small_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = ((p.M + SBM - 1) / SBM) * p.gridN * p.splits;
  hipLaunchKernelGGL(conv_small_kernel<W>, dim3(grid), dim3(STHREADS), Geo::LDS, stream, p);
  return SIHL_OK;
}

}  // namespace

void sihl_small_set_enabled(bool on) { g_small = on; }

// K-split of an eligible launch: all channel chunks as slices while that keeps the grid within ~1.5 waves of workgroups
static int small_splits(const ConvParams& p) {
  const long tiles = ((p.M + SBM - 1) / SBM) * (p.Cout / SBN);
  const int nchunks = p.Cin / 64;
  int s = nchunks;
  while (s > 1 && (tiles * s > 384 || nchunks % s)) --s;
  if (s > 8) s = 8;
  while (s > 1 && nchunks % s) --s;
  return s;
}

// fp32 scratch the launch would need for this forward-shaped problem (0: not this kernel's shape, or no split)
long sihl_small_ws_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
  if (KH != 3 || KW != 3 || stride != 1 || pad != 1 || dil != 1 || H != W || (W != 16 && W != 8 && W != 4)) return 0;
  if (Cin % 64 || Cout % SBN) return 0;
  ConvParams p;
  p.M = N * H * W; p.Cin = Cin; p.Cout = Cout;
  const int s = small_splits(p);
  return s > 1 ? (long)s * p.M * Cout * 4 : 0;
}

bool sihl_small_eligible(const ConvParams& p) {
  if (!g_small || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.dil != 1 || p.in_dilate != 1) return false;
  if (p.H != p.W || (p.W != 16 && p.W != 8 && p.W != 4) || p.Ho != p.H || p.Wo != p.W) return false;
  if (p.Cin % 64 || p.Cout % SBN || p.add || p.out_s != 1 || p.out_image_stride != (long)p.Ho * p.Wo * p.Cout) return false;
  if (p.w_ntaps != 9 || p.w_kw != 3 || p.w_ky0 || p.w_kx0 || p.w_kys != 1 || p.w_kxs != 1) return false;
  if ((long)p.N * p.H * p.W * p.Cin * 2 >= (1L << 31) || (long)p.Cout * 9 * p.Cin * 2 >= (1L << 31)) return false;
  const int s = small_splits(p);
  if (s > 1) {
    const long tiles = ((p.M + SBM - 1) / SBM) * (p.Cout / SBN);
    (void)tiles;
    if (!p.partial || p.partial_bytes < (long)s * p.M * p.Cout * 4 || p.Cout % 4) return false;
  }
  return true;
}

int sihl_small_launch(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  p.gridM = (p.M + SBM - 1) / SBM;
  p.gridN = p.Cout / SBN;
  p.splits = small_splits(p);
  p.small_nch = p.Cin / 64 / p.splits;
  p.dbg = g_dbg;  // read by the kernel in SIHL_TUNING builds only
  p.k_rotate = g_krot % 1000 != 0;
  const double flops = 2.0 * p.M * (double)p.Cout * 9 * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * 9 * p.Cin) * 2.0;
  sihl_prof_begin(SIHL_PROF_CONV, SIHL_BF16, flops, bytes, stream);
  int rc;
  if (p.W == 16) rc = launch_small<16>(p, stream);
  else if (p.W == 8) rc = launch_small<8>(p, stream);
  else rc = launch_small<4>(p, stream);
  if (rc == SIHL_OK && p.splits > 1) rc = sihl_conv_splitk_finish_bf16(p, stream);
  sihl_prof_end(stream);
  if (rc != SIHL_OK) return rc;
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}
