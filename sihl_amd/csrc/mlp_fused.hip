// Whole-MLP forward in ONE launch (inference): torchvision.ops.MLP as the dense heads use it
// (src/sihl/heads/object_detection.py:51-61,108-121): [Linear -> LayerNorm -> SiLU] x nhidden -> Linear.
//
// SURVEY App. D counts an MLP as one op: its activations never need to leave the chip between layers.  Layer by layer
// (sihl_conv2d_fwd + sihl_layernorm_act) the loc head over the 174 592 pyramid positions of a 512^2 batch of 32 was
// 4 x (49 us Linear + 37 us LayerNorm) + 20 us = 0.37 ms and moved each 89 MB activation tensor four times through
// HBM per layer; here a workgroup keeps a 128-row tile of activations in LDS for the whole chain:
//
//   X tile   [4 K-chunks][128 rows][128 B] bf16 (64 KiB) - the A operand of every layer, in the LDS-DMA layout of the
//            conv kernel (16-byte chunk `pos` of row r holds channels (pos ^ ((r >> 1) & 7)) * 8 ... of that K-chunk)
//   W ring   NST stages of [256 out-channels][128 B] (32 KiB each): the five weight panels stream through it by
//            LDS-DMA as ONE sequence of stages across layer boundaries, so the next layer's first panels are already
//            in flight while a layer is being normalised
//   8 waves  4 (output channels) x 2 (rows), wave tile 64 x 64, v_mfma_f32_32x32x16_bf16 with the WEIGHT fragment as the
//            first operand: the accumulator layout then gives a lane 32 channels of two ROWS of the tile (in groups of
//            four consecutive channels), fp32
//   epilogue of a hidden layer, on the accumulators (the pre-norm row never exists in memory; the layer-by-layer path
//            rounds it to bf16 and normalises that): per row and 64-channel wave chunk (sum, M2 about the chunk mean) from
//            in-register sums and one exchange between the lane halves; the four chunks meet through 4 KiB of LDS and are
//            combined by the pairwise update of Chan et al. (as robust as a two-pass variance); then
//            act(LayerNorm(z) * gamma + beta) in packed fp32 arithmetic, rounded to bf16 and written into the X tile as
//            the next layer's operand - 8-byte ds_write_b64 pieces, 16 lanes = 16 different bank slots.  The accumulators
//            start at the bias.
//   last layer: only the 32-channel blocks that hold real outputs are multiplied; rows are stored straight from the
//            accumulators, 8 bytes (4 channels) per store.
#include "common.h"
#include "dma.h"

namespace {

constexpr int MLP_MAXL = 8;  // hidden layers
struct MlpParams {
  const void* x;
  void* out;
  long x_stride;   // elements between rows of x
  int out_stride;  // elements between rows of out (a multiple of 8, >= Cout; columns beyond Cout are written as 0)
  int rows, Cin, C, Cout, nhidden;
  float eps;
  const void* w[MLP_MAXL + 1];      // layer l: [Cout_l][K_l] bf16, K_0 = Cin, K_l = C
  const float* bias[MLP_MAXL + 1];  // may be null
  const float* gamma[MLP_MAXL];
  const float* beta[MLP_MAXL];
  unsigned long long* stamps;  // SIHL_MLP_STAMPS builds: s_memtime marks of workgroup 0 (2-stage ring only)
  int dbg;  // SIHL_TUNING builds: timing ablations (results invalid): 1 no row normalisation, 2 no z write, 4 no MFMA loop, 8 no weight DMA
};
// In-kernel stamps (diagnostic build -DSIHL_MLP_STAMPS, 2-stage ring: the marks live in the unused 32 KiB of LDS)
#ifdef SIHL_MLP_STAMPS
#define MLP_STAMP(i)                                                                                  \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    unsigned long long t__;                                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                       \
    if (tid == 0) ((unsigned long long*)(smem + XBYTES + 2 * WSTAGE))[i] = t__;                        \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define MLP_STAMP(i) do {} while (0)
#endif
#ifdef SIHL_TUNING
#define MLP_DBG(p) ((p).dbg)
#else
#define MLP_DBG(p) 0
#endif

constexpr int BM = 128, BN = 256, KCB = 128, KCE = 64;
constexpr int XPLANE = BM * KCB;    // one K-chunk of the activation tile
constexpr int XBYTES = 4 * XPLANE;  // up to 256 channels
constexpr int WSTAGE = BN * KCB;
constexpr int NTHREADS = 512;
constexpr int W_PIECES = BN * KCB / 1024 / (NTHREADS / 64);  // LDS-DMA wave-instructions per wave per weight stage (4)

__device__ __forceinline__ float fast_sigmoid_(float v) {
  const float ex = __builtin_amdgcn_exp2f(fminf(-v * 1.4426950408889634f, 126.f));
  return __builtin_amdgcn_rcpf(1.f + ex);
}

template <int NST, int ACT>
__global__ __launch_bounds__(NTHREADS, 1) void mlp_fused_kernel(const MlpParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 1, wp = wave & 1;  // this wave's 64 output channels / 64 rows of the tile
  const int m0 = blockIdx.x * BM;
  MLP_STAMP(0);
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  const unsigned ring_base = lds_base + XBYTES;
  const int nk_in = (p.Cin + KCE - 1) / KCE, nk_h = (p.C + KCE - 1) / KCE;
  const int total = nk_in + p.nhidden * nk_h;  // weight stages of all layers

  // ---- the activation tile: rows m0 .. m0 + 127, every K-chunk of the input (rows beyond M / channels beyond Cin: zeros)
  {
    const v4i_t x_rsrc = make_rsrc(p.x, (unsigned)(((long)(p.rows - 1) * p.x_stride + p.Cin) * 2));
    for (int kc = 0; kc < nk_in; ++kc) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j;  // 8 rows x 128 B per wave-instruction
        const int row = piece * 8 + (lane >> 3), pos = lane & 7;
        const int ch = kc * KCE + ((pos ^ ((row >> 1) & 7)) << 3);
        const int m = m0 + row;
        const bool ok = m < p.rows && ch < p.Cin;
        dma16(ok ? (unsigned)(((long)m * p.x_stride + ch) * 2) : OOB, lds_base + kc * XPLANE + piece * 1024, x_rsrc);
      }
    }
  }

  // ---- weight stages: one sequence over all layers (layer, K-chunk), slot = stage % NST
  int i_l = 0, i_kc = 0, i_slot = 0, issued = 0;
  auto issue_stage = [&]() {
    const int K = i_l == 0 ? p.Cin : p.C;
    const int Co = i_l < p.nhidden ? p.C : p.Cout;
    const v4i_t w_rsrc = make_rsrc(p.w[i_l], (unsigned)((long)Co * K * 2));
    const unsigned dst = ring_base + i_slot * WSTAGE;
#pragma unroll
    for (int j = 0; j < W_PIECES; ++j) {
      const int piece = wave * W_PIECES + j;
      const int row = piece * 8 + (lane >> 3), pos = lane & 7;
      const int ch = i_kc * KCE + ((pos ^ ((row >> 1) & 7)) << 3);
      const bool ok = row < Co && ch < K;
      dma16(ok ? (unsigned)(((long)row * K + ch) * 2) : OOB, dst + piece * 1024, w_rsrc);
    }
    ++issued;
    i_slot = i_slot + 1 == NST ? 0 : i_slot + 1;
    if (++i_kc == (i_l == 0 ? nk_in : nk_h)) { i_kc = 0; ++i_l; }
  };
  // all but the `keep` newest stages of this wave's DMA have landed (vector-memory operations retire in order)
  auto wait_keep = [&](int keep) {
    if (NST >= 3 && keep >= 1) wait_vm_keep<W_PIECES>();
    else wait_vm_keep<0>();
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (issued < total) issue_stage();
  MLP_STAMP(1);
  wait_keep(min(NST - 2, issued - 1));
  __syncthreads();
  MLP_STAMP(2);

  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) koff[ks] = fr * KCB + (((ks * 2 + fh) ^ fsw) << 4);
  const float inv_c = 1.f / (float)p.C;
  // Accumulator layout of v_mfma_f32_32x32x16 with the WEIGHTS as the first operand: lane (fr, fh) holds, for tile row
  // wp*64 + j*32 + fr, the output channels wc*64 + i*32 + 8*g + 4*fh + {0..3} in registers 4*g .. 4*g + 3 - a lane owns
  // 32 channels of two rows, so the row statistics are in-register sums plus one exchange between the lane halves and
  // one between the four channel-waves.
  const int chan0 = wc * 64 + 4 * fh;  // + i*32 + 8*g

  int g = 0, slot = 0;
  for (int l = 0; l <= p.nhidden; ++l) {
    const int nk = l == 0 ? nk_in : nk_h;
    const bool last = l == p.nhidden;
    const int Co = last ? p.Cout : p.C;
    // 32-channel blocks of this wave that hold real outputs (both, in a hidden layer of full width)
    const int nblk = (Co + 31) >> 5;
    const bool on0 = wc * 2 < nblk, on1 = wc * 2 + 1 < nblk;

    // the accumulators start at the bias
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = chan0 + i * 32 + 8 * q;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias[l] && c < Co) {
          if (c + 4 <= Co) b = *(const float4*)(p.bias[l] + c);
          else {  // the last layer's Cout need not be a multiple of 4
            b.x = p.bias[l][c];
            if (c + 1 < Co) b.y = p.bias[l][c + 1];
            if (c + 2 < Co) b.z = p.bias[l][c + 2];
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j][4 * q + 0] = b.x; acc[i][j][4 * q + 1] = b.y; acc[i][j][4 * q + 2] = b.z; acc[i][j][4 * q + 3] = b.w;
        }
      }
    }

    for (int kc = 0; kc < nk; ++kc, ++g) {
      if (issued < total && !(MLP_DBG(p) & 8)) issue_stage();  // into the slot of stage g - 1: every wave is past the barrier behind it
      const char* Ws = smem + XBYTES + slot * WSTAGE + wc * 64 * KCB;
      const char* Xs = smem + kc * XPLANE + wp * 64 * KCB;
      if (on0 && !(MLP_DBG(p) & 4)) {
        uint4 fw[2][2], fx[2][2];
        fw[0][0] = *(const uint4*)(Ws + koff[0]);
        if (on1) fw[0][1] = *(const uint4*)(Ws + 32 * KCB + koff[0]);
#pragma unroll
        for (int j = 0; j < 2; ++j) fx[0][j] = *(const uint4*)(Xs + j * 32 * KCB + koff[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) {
            fw[(ks + 1) & 1][0] = *(const uint4*)(Ws + koff[ks + 1]);
            if (on1) fw[(ks + 1) & 1][1] = *(const uint4*)(Ws + 32 * KCB + koff[ks + 1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fx[(ks + 1) & 1][j] = *(const uint4*)(Xs + j * 32 * KCB + koff[ks + 1]);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fw[ks & 1][0]),
                                                                 __builtin_bit_cast(bf16x8_t, fx[ks & 1][j]), acc[0][j], 0, 0, 0);
            if (on1)
              acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fw[ks & 1][1]),
                                                                   __builtin_bit_cast(bf16x8_t, fx[ks & 1][j]), acc[1][j], 0, 0, 0);
          }
        }
      }
      // stage g + 1 must have landed for every wave, and everyone is done reading stage g (and, in the layer's last
      // iteration, the activation tile)
      wait_keep(min(NST - 2, issued - g - 2));
      __syncthreads();
      slot = slot + 1 == NST ? 0 : slot + 1;
    }
    MLP_STAMP(3 + 4 * l);

    if (!last) {
      // ---- LayerNorm + activation on the accumulators.  Per row: this wave's 64-channel chunk gives (sum, M2 about the
      // chunk mean); the four chunks are combined by the pairwise update of Chan et al. - as robust as the two-pass
      // variance of layernorm_act_kernel, with ONE exchange through LDS.  The exchange buffer is the ring slot of the
      // stage just multiplied (free until the next K loop issues into it, which happens behind this epilogue's barrier).
      float4 gam[2][4], bet[2][4];
      const int nvalid = max(0, min(64, p.C - wc * 64));  // real channels in this wave's chunk (a multiple of 8)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = chan0 + i * 32 + 8 * q;
          const bool ok = c < p.C;
          gam[i][q] = ok ? *(const float4*)(p.gamma[l] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          bet[i][q] = ok ? *(const float4*)(p.beta[l] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      float2* xch = (float2*)(smem + XBYTES + (slot == 0 ? NST - 1 : slot - 1) * WSTAGE);  // [128 rows][4 chunks]
      float csum[2], cm2[2];
      if (!(MLP_DBG(p) & 1)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x2_t s2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) s2 += (f32x2_t){acc[i][j][r], acc[i][j][r + 1]};  // channels beyond C hold 0
        float sum = s2.x + s2.y;
        sum += __shfl_xor(sum, 32);
        const float cmean = nvalid ? sum / (float)nvalid : 0.f;
        const f32x2_t nm = {-cmean, -cmean};
        f32x2_t q2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            if (chan0 + i * 32 + 8 * (r >> 2) < p.C) {
              const f32x2_t d = (f32x2_t){acc[i][j][r], acc[i][j][r + 1]} + nm;
              q2 = __builtin_elementwise_fma(d, d, q2);
            }
          }
        float m2 = q2.x + q2.y;
        m2 += __shfl_xor(m2, 32);
        csum[j] = sum;
        cm2[j] = m2;
        if (fh == 0) xch[(wp * 64 + j * 32 + fr) * 4 + wc] = make_float2(sum, m2);
      }
      }
      MLP_STAMP(4 + 4 * l);
      __syncthreads();
      if (!(MLP_DBG(p) & 1)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wp * 64 + j * 32 + fr;
        const float4 a01 = *(const float4*)(xch + row * 4), a23 = *(const float4*)(xch + row * 4 + 2);
        const float mu = (a01.x + a01.z + a23.x + a23.z) * inv_c;
        float m2 = a01.y + a01.w + a23.y + a23.w;
        const float cs[4] = {a01.x, a01.z, a23.x, a23.z};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int nw = max(0, min(64, p.C - w * 64));
          if (nw) { const float d = cs[w] / (float)nw - mu; m2 += (float)nw * d * d; }
        }
        const float rs = 1.f / sqrtf(m2 * inv_c + p.eps);
        const f32x2_t rs2 = {rs, rs}, nmr = {-mu * rs, -mu * rs};
        char* xrow = smem + wc * XPLANE + row * KCB + 8 * fh;  // this wave's channels are K-chunk wc of the next layer
        const int rsw = (row >> 1) & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 ga = gam[i][q], be = bet[i][q];
            f32x2_t t0 = __builtin_elementwise_fma((f32x2_t){acc[i][j][4 * q], acc[i][j][4 * q + 1]}, rs2, nmr);
            f32x2_t t1 = __builtin_elementwise_fma((f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]}, rs2, nmr);
            t0 = __builtin_elementwise_fma(t0, (f32x2_t){ga.x, ga.y}, (f32x2_t){be.x, be.y});
            t1 = __builtin_elementwise_fma(t1, (f32x2_t){ga.z, ga.w}, (f32x2_t){be.z, be.w});
            if (ACT == SIHL_ACT_SILU) {
              const f32x2_t k2 = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
              f32x2_t e0 = t0 * k2, e1 = t1 * k2;
              e0 = (f32x2_t){__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)} + one;
              e1 = (f32x2_t){__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)} + one;
              t0 *= (f32x2_t){__builtin_amdgcn_rcpf(e0.x), __builtin_amdgcn_rcpf(e0.y)};
              t1 *= (f32x2_t){__builtin_amdgcn_rcpf(e1.x), __builtin_amdgcn_rcpf(e1.y)};
            } else if (ACT == SIHL_ACT_RELU) {
              t0 = (f32x2_t){fmaxf(t0.x, 0.f), fmaxf(t0.y, 0.f)};
              t1 = (f32x2_t){fmaxf(t1.x, 0.f), fmaxf(t1.y, 0.f)};
            }
            uint2 pk;
            pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, bf16x2_t));
            pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, bf16x2_t));
            if (chan0 + i * 32 + 8 * q >= p.C) pk = make_uint2(0u, 0u);  // keep the operand's padding channels at zero
            if (!(MLP_DBG(p) & 2)) *(uint2*)(xrow + (((i * 4 + q) ^ rsw) << 4)) = pk;
          }
      }
      }
      MLP_STAMP(5 + 4 * l);
      __syncthreads();
      MLP_STAMP(6 + 4 * l);
    } else {
      // ---- output rows from the accumulators: 4 consecutive channels per lane and register group = one 8-byte store
      // (channels beyond Cout come out as 0: zero weight rows, zero bias)
      bf16_t* __restrict__ out = (bf16_t*)p.out;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m0 + wp * 64 + j * 32 + fr;
        if (m >= p.rows) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = chan0 + i * 32 + 8 * q;
            if (c >= p.out_stride) continue;
            uint2 pk;
            pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[i][j][4 * q], acc[i][j][4 * q + 1]}, bf16x2_t));
            pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]}, bf16x2_t));
            *(uint2*)(out + (long)m * p.out_stride + c) = pk;
          }
      }
    }
  }
#ifdef SIHL_MLP_STAMPS
  MLP_STAMP(4 + 4 * p.nhidden);
  __syncthreads();
  if (blockIdx.x == 0 && tid < 64 && p.stamps) p.stamps[tid] = ((unsigned long long*)(smem + XBYTES + 2 * WSTAGE))[tid];
#endif
}

int g_mlp_stages = 3;
int g_mlp_dbg = 0;
unsigned long long* g_mlp_stamps = nullptr;

template <int NST, int ACT>
int launch_mlp(const MlpParams& p, hipStream_t stream) {
#ifdef SIHL_MLP_STAMPS
  constexpr int LDS = XBYTES + NST * WSTAGE + (NST == 2 ? 1024 : 0);
  if (NST != 2) return SIHL_EARG;
#else
  constexpr int LDS = XBYTES + NST * WSTAGE;
#endif
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_fused_kernel<NST, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((mlp_fused_kernel<NST, ACT>), dim3((p.rows + BM - 1) / BM), dim3(NTHREADS), LDS, stream, p);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

}  // namespace

extern "C" {

// Tuning hook: LDS stages of the weight ring (2 or 3).
int sihl_mlp_stages(int n) {
  if (n != 2 && n != 3) return SIHL_EARG;
  g_mlp_stages = n;
  return SIHL_OK;
}

// Tuning ablation (SIHL_TUNING builds only; results invalid when non-zero): see MlpParams::dbg.
#ifdef SIHL_TUNING
int sihl_mlp_debug(int mode) { g_mlp_dbg = mode; return SIHL_OK; }
#else
int sihl_mlp_debug(int mode) { return mode == 0 ? SIHL_OK : SIHL_EARG; }  // the shipped library has no ablation state
#endif

// Diagnostic builds (-DSIHL_MLP_STAMPS): device buffer of 64 x u64 that workgroup 0 fills with its s_memtime marks.
int sihl_mlp_stamps(void* buf) { g_mlp_stamps = (unsigned long long*)buf; return SIHL_OK; }

// 1 when sihl_mlp_fwd covers the shape (bf16, Cin / C / Cout <= 256, 16-byte channel vectors, <= 8 hidden layers).
int sihl_mlp_fwd_supported(long rows, int Cin, int C, int Cout, int nhidden, int act, int dtype) {
  return dtype == SIHL_BF16 && rows > 0 && rows < (1L << 30) && Cin > 0 && Cin <= 256 && Cin % 8 == 0 && Cout > 0 &&
         Cout <= 256 && nhidden >= 0 && nhidden <= MLP_MAXL && (nhidden == 0 || (C > 0 && C <= 256 && C % 8 == 0)) &&
         (act == SIHL_ACT_SILU || act == SIHL_ACT_RELU || act == SIHL_ACT_NONE);
}

// out[rows][out_stride] = Linear_n( act(LN(Linear_{n-1}( ... act(LN(Linear_0(x))) ... ))) ), one launch.
// x: [rows] rows of Cin bf16, x_stride elements apart.  w / bias / gamma / beta: HOST arrays of device pointers -
// nhidden + 1 weights ([C][Cin], [C][C] ..., [Cout][C]; bf16, row-major, at least that many rows) and biases (fp32, entries
// may be NULL), nhidden LayerNorm scale / shift vectors (fp32, C entries).  out_stride: a multiple of 8, >= Cout.
int sihl_mlp_fwd(const void* x, long x_stride, long rows, int Cin, int C, int nhidden, const void* const* w,
                 const float* const* bias, const float* const* gamma, const float* const* beta, float eps, int act,
                 int Cout, void* out, int out_stride, int dtype, hipStream_t stream) {
  if (!x || !out || !w || !bias || (nhidden > 0 && (!gamma || !beta))) return SIHL_EARG;
  if (!sihl_mlp_fwd_supported(rows, Cin, C, Cout, nhidden, act, dtype)) return SIHL_EARG;
  if (out_stride < Cout || out_stride % 8 || out_stride > 256 || x_stride < Cin || x_stride % 8) return SIHL_EARG;
  if ((rows * x_stride) * 2 >= (1L << 31)) return SIHL_EARG;  // 32-bit buffer offsets
  MlpParams p;
  p.x = x; p.out = out; p.x_stride = x_stride; p.out_stride = out_stride;
  p.dbg = g_mlp_dbg;
  p.stamps = g_mlp_stamps;
  p.rows = (int)rows; p.Cin = Cin; p.C = nhidden ? C : Cin; p.Cout = Cout; p.nhidden = nhidden; p.eps = eps;
  for (int l = 0; l <= MLP_MAXL; ++l) { p.w[l] = nullptr; p.bias[l] = nullptr; }
  for (int l = 0; l < MLP_MAXL; ++l) { p.gamma[l] = nullptr; p.beta[l] = nullptr; }
  for (int l = 0; l <= nhidden; ++l) {
    if (!w[l]) return SIHL_EARG;
    p.w[l] = w[l];
    p.bias[l] = bias[l];
  }
  for (int l = 0; l < nhidden; ++l) {
    if (!gamma[l] || !beta[l]) return SIHL_EARG;
    p.gamma[l] = gamma[l];
    p.beta[l] = beta[l];
  }
#define SIHL_MLP(A)                                            \
  do {                                                         \
    if (g_mlp_stages == 2) return launch_mlp<2, A>(p, stream); \
    return launch_mlp<3, A>(p, stream);                        \
  } while (0)
  if (act == SIHL_ACT_SILU) SIHL_MLP(SIHL_ACT_SILU);
  if (act == SIHL_ACT_RELU) SIHL_MLP(SIHL_ACT_RELU);
  SIHL_MLP(SIHL_ACT_NONE);
#undef SIHL_MLP
}

}  // extern "C"
