// Whole-MLP forward in ONE launch, activations in REGISTERS (inference): torchvision.ops.MLP as the dense heads use it
// (src/sihl/heads/object_detection.py:51-61,108-121): [Linear -> LayerNorm -> SiLU] x nhidden -> Linear.
//
// mlp_fused.hip keeps a 128-row activation tile in LDS and splits the 256 output channels over the waves: every hidden
// layer pays an LDS round trip of the tile, a cross-wave exchange of the row statistics and three workgroup barriers, and
// the one workgroup a CU can hold (160 KiB) has nothing to run while it normalises (stamps: 35 % of the kernel).  Here a
// WAVE owns 32 rows and ALL channels of them:
//
//   v_mfma_f32_32x32x16_bf16 with the WEIGHT fragment as the first operand: a lane (fr = lane & 31, fh = lane >> 5) then
//   holds, for row fr, the output channels  32 i + 8 g + 4 fh + e  (block i < 8, g < 4, e < 4) in accumulator register
//   4 g + e of block i - 128 fp32 registers for the 256 channels of the row, split between the two lane halves.
//   * LayerNorm: the row statistics are in-register sums plus ONE exchange between the lane halves (two-pass variance).
//     No LDS, no barrier, and the waves of a workgroup drift apart only by what the weight ring allows.
//   * The next layer's operand never leaves the registers either.  The second MFMA operand wants, for k-step kappa, the 8
//     contraction indices 8 fh .. 8 fh + 7 of row fr; the contraction order is free as long as both operands agree, so
//     k-slot (8 fh + e) of k-step kappa = (i, h) is DEFINED as channel 32 i + 16 h + (e < 4 ? 4 fh + e : 8 + 4 fh + e - 4):
//     exactly accumulator registers 8 h .. 8 h + 7 of block i, normalised, activated and packed to bf16.  The weights of
//     layers >= 1 are stored with that K order (sihl_mlp_permute_k, once per weight version): inside each group of 16
//     input channels [0-3, 8-11, 4-7, 12-15].  Layer 0 reads x from global memory in the plain order.
//   * LDS holds only the weight ring - 2 stages of [256 out-channels][64 k] bf16 (the conv kernel's LDS-DMA layout) -
//     and the fp32 bias / gamma / beta vectors: 64 + 13 KiB for four hidden layers, so TWO workgroups (4 waves, 128 rows
//     each) share a CU and one's LayerNorm + SiLU (VALU) runs under the other's matrix instructions.
//   Per wave and 64-deep weight stage: 32 ds_read_b128 + 32 MFMA; per layer 128 MFMA = 4 096 matrix cycles against
//   ~4 300 VALU cycles of normalisation for the same 32 x 256 outputs.
#include "common.h"
#include "dma.h"

namespace {

constexpr int MLPR_MAXL = 8;  // hidden layers
struct MlpRowsParams {
  const void* x;
  void* out;
  long x_stride;   // elements between rows of x
  int out_stride;  // elements between rows of out (a multiple of 8, >= Cout; columns beyond Cout are written as 0)
  int rows, Cin, C, Cout, nhidden;
  float eps;
  const void* w[MLPR_MAXL + 1];      // layer 0: [C][Cin] plain; layer l >= 1: [Cout_l][C] in the permuted K order
  const float* bias[MLPR_MAXL + 1];  // may be null
  const float* gamma[MLPR_MAXL];
  const float* beta[MLPR_MAXL];
  int dbg;  // SIHL_TUNING builds: timing ablations (results invalid): 1 no normalisation, 2 no weight DMA in the loop, 4 no MFMA
};
#ifdef SIHL_TUNING
#define MLPR_DBG(p) ((p).dbg)
#else
#define MLPR_DBG(p) 0
#endif

constexpr int MLPR_MAXN = 4;  // MLPs per launch
struct MlpRowsBatch {
  MlpRowsParams p[MLPR_MAXN];
  int first_block[MLPR_MAXN + 1];  // workgroups first_block[k] .. first_block[k + 1] - 1 run MLP k
};

constexpr int RKCB = 128, RKCE = 64;
constexpr int RSTAGE = 256 * RKCB;  // one weight stage: 256 out-channels x 64 k (32 KiB)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}

// One 64-deep weight stage times NB 32-channel blocks: steps of (up to) 4 blocks x one k-step; the weight fragments of
// step t + 1 are requested before step t multiplies, and scheduling barriers keep hipcc from hoisting all fragment reads
// of the stage (128 registers) above the first MFMA.
template <int NB, int KC>
__device__ __forceinline__ void mlp_rows_stage(f32x16_t (&acc)[8], const uint4 (&xk)[16], const char* Ws, int fh, int fsw) {
  constexpr int G = NB < 4 ? NB : 4;       // blocks per step
  constexpr int NG = (NB + G - 1) / G;     // block groups per k-step
  constexpr int STEPS = 4 * NG;
  uint4 fw[2][G];
#pragma unroll
  for (int j = 0; j < G; ++j) fw[0][j] = *(const uint4*)(Ws + j * 32 * RKCB + ((fh ^ fsw) << 4));
#pragma unroll
  for (int t = 0; t < STEPS; ++t) {
    if (t + 1 < STEPS) {
      const int ks1 = (t + 1) / NG, i1 = ((t + 1) % NG) * G;
#pragma unroll
      for (int j = 0; j < G; ++j)
        if (i1 + j < NB) fw[(t + 1) & 1][j] = *(const uint4*)(Ws + (i1 + j) * 32 * RKCB + (((ks1 * 2 + fh) ^ fsw) << 4));
    }
    const int ks = t / NG, i0 = (t % NG) * G;
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (i0 + j < NB)
        acc[i0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fw[t & 1][j]),
                                                              __builtin_bit_cast(bf16x8_t, xk[KC * 4 + ks]), acc[i0 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// SILU: 1 = SiLU, 0 = max(t, floor) (floor = 0: ReLU, -inf: none).  LASTB: 32-channel blocks of the last layer (Cout <= 32 LASTB).
// Hidden layers are full width (C == 256: 8 blocks, compile-time - a run-time block count puts the accumulators behind
// branches and hipcc then spills them).
// NW waves of 32 rows each per workgroup, NS weight stages in the ring.  (4, 2): two workgroups per CU, each streaming every
// layer's weights for its 128 rows; (8, NS >= 3): ONE workgroup of 256 rows per CU - the same 8 waves per CU, half the
// weight bytes through L2 -> LDS per row (tools/mlp_probe.py ablations: the weight stream was the largest phase) and
// NS - 1 stages in flight instead of one.
template <int SILU, int LASTB, int NW, int NS>
__global__ __launch_bounds__(NW * 64, 2) void mlp_rows_kernel(const MlpRowsBatch pb, const float floor_) {
  constexpr int RBM = NW * 32, RRING = NS * RSTAGE, PPW = 32 / NW;  // rows per workgroup; ring bytes; DMA pieces per wave and stage
  // several MLPs in one launch (the class and box heads of a detection head run over the same few thousand rows: each
  // alone fills a tenth of the chip for the same 35 us)
  int which = 0;
#pragma unroll
  for (int k = 1; k < MLPR_MAXN; ++k) which += (int)blockIdx.x >= pb.first_block[k] ? 1 : 0;
  const MlpRowsParams& p = pb.p[which];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
  const int m0 = ((int)blockIdx.x - pb.first_block[which]) * RBM;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_ptr_t)smem;
  // fp32 vectors behind the ring: bias[nhidden + 1][256], gamma[nhidden][256], beta[nhidden][256] (zeros beyond the widths)
  float* pbias = (float*)(smem + RRING);
  float* pgamma = pbias + (p.nhidden + 1) * 256;
  float* pbeta = pgamma + p.nhidden * 256;
  const int nk_in = (p.Cin + RKCE - 1) / RKCE;
  const int total = nk_in + p.nhidden * 4;  // weight stages of all layers

  // ---- this lane's row of x, every k-step of layer 0 (plain channel order; rows beyond M / channels beyond Cin: zeros)
  uint4 xk[16];
  {
    const int m = m0 + wave * 32 + fr;
    const bf16_t* xr = (const bf16_t*)p.x + (long)m * p.x_stride + 8 * fh;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      xk[k] = (m < p.rows && 16 * k + 8 * fh < p.Cin) ? *(const uint4*)(xr + 16 * k) : make_uint4(0u, 0u, 0u, 0u);
  }
  if (tid < 256)
    for (int l = 0; l <= p.nhidden; ++l) {
      const int Co = l < p.nhidden ? 256 : p.Cout;
      pbias[l * 256 + tid] = (p.bias[l] && tid < Co) ? p.bias[l][tid] : 0.f;
      if (l < p.nhidden) {
        pgamma[l * 256 + tid] = p.gamma[l][tid];
        pbeta[l * 256 + tid] = p.beta[l][tid];
      }
    }

  // ---- weight stages: one sequence over all layers (layer, K-chunk), slot = stage % NS
  int i_l = 0, i_kc = 0, i_slot = 0, issued = 0;
  auto issue_stage = [&]() {
    const int K = i_l == 0 ? p.Cin : 256;
    const int Co = i_l < p.nhidden ? 256 : p.Cout;
    const int live = i_l < p.nhidden ? 256 : LASTB * 32;  // rows the multiply reads (rows beyond Co: zeros)
    const v4i_t w_rsrc = make_rsrc(p.w[i_l], (unsigned)((long)Co * K * 2));
    const unsigned dst = lds_base + i_slot * RSTAGE;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave * PPW + j;  // 8 rows x 128 B per wave-instruction
      if (NS == 2 && piece * 8 >= live) continue;  // wave-uniform (deeper rings count their pieces: every piece is issued)
      const int row = piece * 8 + (lane >> 3), pos = lane & 7;
      const int ch = i_kc * RKCE + ((pos ^ ((row >> 1) & 7)) << 3);
      const bool ok = row < Co && ch < K;
      dma16(ok ? (unsigned)(((long)row * K + ch) * 2) : OOB, dst + piece * 1024, w_rsrc);
    }
    ++issued;
    i_slot = i_slot + 1 == NS ? 0 : i_slot + 1;
    if (++i_kc == (i_l == 0 ? nk_in : 4)) { i_kc = 0; ++i_l; }
  };
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (issued < total) issue_stage();
  __syncthreads();  // the parameter vectors are in LDS
  // (Tried: a start delay for alternate workgroups, to put the two workgroups of a CU out of phase - one normalising while
  // the other multiplies.  Monotonically slower, 143.5 -> 160.9 us for 0 -> 6 x 4096 cycles: not kept.)

  const float inv_c = 1.f / 256.f;
  int slot = 0, chunk = 0;
  f32x16_t acc[8];
  // stage `chunk` has landed when at most the stages issued behind it are in flight: min(NS - 2, total - 1 - chunk) of them
  auto wait_stage = [&]() {
    if constexpr (NS == 2) {
      wait_vm_keep<0>();
    } else {
      const int later = min(NS - 2, total - 1 - chunk);
      if (later >= 2) wait_vm_keep<2 * PPW>();
      else if (later == 1) wait_vm_keep<PPW>();
      else wait_vm_keep<0>();
    }
  };
  // one K chunk: stage (l, kc) has landed for every wave and everyone is done reading the other slot; refill that one
#define MLPR_CHUNK(NB, KC)                                                            \
  do {                                                                                \
    wait_stage();                                                                     \
    __syncthreads();                                                                  \
    if (issued < total && !(MLPR_DBG(p) & 2)) issue_stage();                          \
    if (!(MLPR_DBG(p) & 4)) mlp_rows_stage<NB, KC>(acc, xk, smem + slot * RSTAGE + fr * RKCB, fh, fsw); \
    slot = slot + 1 == NS ? 0 : slot + 1;                                             \
    ++chunk;                                                                          \
  } while (0)

  for (int l = 0; l < p.nhidden; ++l) {
    // the accumulators start at the bias
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4*)(pbias + l * 256 + i * 32 + 8 * g + 4 * fh);
        acc[i][4 * g + 0] = b.x; acc[i][4 * g + 1] = b.y; acc[i][4 * g + 2] = b.z; acc[i][4 * g + 3] = b.w;
      }
    const int nk = l == 0 ? nk_in : 4;
    MLPR_CHUNK(8, 0);
    if (nk > 1) MLPR_CHUNK(8, 1);
    if (nk > 2) MLPR_CHUNK(8, 2);
    if (nk > 3) MLPR_CHUNK(8, 3);

    if (MLPR_DBG(p) & 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          xk[2 * i + h] = make_uint4(pack_bf16x2(acc[i][8 * h], acc[i][8 * h + 1]), pack_bf16x2(acc[i][8 * h + 2], acc[i][8 * h + 3]),
                                     pack_bf16x2(acc[i][8 * h + 4], acc[i][8 * h + 5]), pack_bf16x2(acc[i][8 * h + 6], acc[i][8 * h + 7]));
      continue;
    }
    // ---- LayerNorm + activation on the accumulators (two-pass variance, in registers + one exchange between the halves)
    f32x2_t s2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) s2 += (f32x2_t){acc[i][r], acc[i][r + 1]};
    float sum = s2.x + s2.y;
    sum += __shfl_xor(sum, 32);
    const float mu = sum * inv_c;
    const f32x2_t nm = {-mu, -mu};
    f32x2_t q2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2_t d = (f32x2_t){acc[i][r], acc[i][r + 1]} + nm;
        q2 = __builtin_elementwise_fma(d, d, q2);
      }
    float m2 = q2.x + q2.y;
    m2 += __shfl_xor(m2, 32);
    const float rs = 1.f / sqrtf(m2 * inv_c + p.eps);
    const f32x2_t rs2 = {rs, rs}, nmr = {-mu * rs, -mu * rs};
    const float* gl = pgamma + l * 256 + 4 * fh;
    const float* bl = pbeta + l * 256 + 4 * fh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned pk[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int g = 2 * h + a;
          const float4 ga = *(const float4*)(gl + i * 32 + 8 * g), be = *(const float4*)(bl + i * 32 + 8 * g);
          f32x2_t t0 = __builtin_elementwise_fma((f32x2_t){acc[i][4 * g], acc[i][4 * g + 1]}, rs2, nmr);
          f32x2_t t1 = __builtin_elementwise_fma((f32x2_t){acc[i][4 * g + 2], acc[i][4 * g + 3]}, rs2, nmr);
          t0 = __builtin_elementwise_fma(t0, (f32x2_t){ga.x, ga.y}, (f32x2_t){be.x, be.y});
          t1 = __builtin_elementwise_fma(t1, (f32x2_t){ga.z, ga.w}, (f32x2_t){be.z, be.w});
          if (SILU) {
            const f32x2_t k2 = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.f, 1.f};
            f32x2_t e0 = t0 * k2, e1 = t1 * k2;
            // (no clamp: exp2 -> inf gives 1 / (1 + inf) = 0 and t * 0 = 0, the limit of SiLU for t -> -inf)
            e0 = (f32x2_t){__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)} + one;
            e1 = (f32x2_t){__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)} + one;
            t0 *= (f32x2_t){__builtin_amdgcn_rcpf(e0.x), __builtin_amdgcn_rcpf(e0.y)};
            t1 *= (f32x2_t){__builtin_amdgcn_rcpf(e1.x), __builtin_amdgcn_rcpf(e1.y)};
          } else {
            t0 = (f32x2_t){fmaxf(t0.x, floor_), fmaxf(t0.y, floor_)};
            t1 = (f32x2_t){fmaxf(t1.x, floor_), fmaxf(t1.y, floor_)};
          }
          pk[2 * a] = pack_bf16x2(t0.x, t0.y);
          pk[2 * a + 1] = pack_bf16x2(t1.x, t1.y);
        }
        xk[2 * i + h] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
      __builtin_amdgcn_sched_barrier(0);  // (hipcc would otherwise fetch the gamma / beta of all blocks first)
    }
  }

  // ---- last layer: only the LASTB blocks that hold real outputs
  {
    const int l = p.nhidden;
#pragma unroll
    for (int i = 0; i < LASTB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *(const float4*)(pbias + l * 256 + i * 32 + 8 * g + 4 * fh);
        acc[i][4 * g + 0] = b.x; acc[i][4 * g + 1] = b.y; acc[i][4 * g + 2] = b.z; acc[i][4 * g + 3] = b.w;
      }
    MLPR_CHUNK(LASTB, 0);
    MLPR_CHUNK(LASTB, 1);
    MLPR_CHUNK(LASTB, 2);
    MLPR_CHUNK(LASTB, 3);
    // output rows from the accumulators: 4 consecutive channels per lane and register group = one 8-byte store
    // (channels beyond Cout come out as 0: zero weight rows, zero bias)
    bf16_t* __restrict__ out = (bf16_t*)p.out;
    const int m = m0 + wave * 32 + fr;
    if (m < p.rows) {
#pragma unroll
      for (int i = 0; i < LASTB; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = i * 32 + 8 * g + 4 * fh;
          if (c >= p.out_stride) continue;
          uint2 pk;
          pk.x = pack_bf16x2(acc[i][4 * g], acc[i][4 * g + 1]);
          pk.y = pack_bf16x2(acc[i][4 * g + 2], acc[i][4 * g + 3]);
          *(uint2*)(out + (long)m * p.out_stride + c) = pk;
        }
    }
  }
#undef MLPR_CHUNK
}

// w_out[co][16 q + 8 a + 4 b + d] = w_in[co][16 q + 8 b + 4 a + d]: the K order the register-resident operand has
__global__ void mlp_permute_k_kernel(const uint2* __restrict__ in, uint2* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long q = i >> 2;
    const int s = (int)(i & 3), a = s >> 1, b = s & 1;
    out[i] = in[(q << 2) + 2 * b + a];
  }
}

int g_mlp_rows_dbg = 0;
// test hook (sihl_mlp_rows_config).  Default 4: two 4-wave workgroups per CU with a 2-stage ring.  8 - ONE 256-row workgroup
// per CU, 4-stage ring: half the weight bytes through L2 -> LDS per row and three stages in flight - was built in round 4 on
// the round-3 ablation that named the weight stream the longest phase, and is SLOWER: north-star forward 2.111 against
// 2.080 ms in one process (profiles/r04_ns_ab_mlp_waves.txt).  Two independent workgroups drift apart - one normalises
// (VALU) while the other multiplies - and eight waves behind one barrier per stage do not.
int g_mlp_rows_waves = 4;
template <int SILU, int LASTB, int NW, int NS>
int launch_mlp_rows_cfg(const MlpRowsBatch& pb, int n, float floor_, hipStream_t stream) {
  int nh = 0;
  for (int k = 0; k < n; ++k) nh = pb.p[k].nhidden > nh ? pb.p[k].nhidden : nh;
  const int lds = NS * RSTAGE + (3 * nh + 1) * 256 * (int)sizeof(float);
  if (lds > 160 * 1024) return SIHL_EARG;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_rows_kernel<SILU, LASTB, NW, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((mlp_rows_kernel<SILU, LASTB, NW, NS>), dim3(pb.first_block[n]), dim3(NW * 64), lds, stream, pb, floor_);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}
template <int SILU, int LASTB>
int launch_mlp_rows(const MlpRowsBatch& pb, int n, float floor_, hipStream_t stream) {
  if (g_mlp_rows_waves == 8) return launch_mlp_rows_cfg<SILU, LASTB, 8, 4>(pb, n, floor_, stream);
  return launch_mlp_rows_cfg<SILU, LASTB, 4, 2>(pb, n, floor_, stream);
}

}  // namespace

extern "C" {

// 1 when sihl_mlp_rows_fwd covers the shape: bf16, hidden width exactly 256 (the dense heads' num_channels default), Cin a
// multiple of 8 up to 256, Cout <= 256, 1 .. 8 hidden layers.
int sihl_mlp_rows_supported(long rows, int Cin, int C, int Cout, int nhidden, int act, int dtype) {
  return dtype == SIHL_BF16 && rows > 0 && rows < (1L << 30) && Cin > 0 && Cin <= 256 && Cin % 8 == 0 && Cout > 0 &&
         Cout <= 256 && nhidden >= 1 && nhidden <= MLPR_MAXL && C == 256 &&
         (act == SIHL_ACT_SILU || act == SIHL_ACT_RELU || act == SIHL_ACT_NONE);
}

// Test hook: waves per workgroup of sihl_mlp_rows_fwd - 4 (default: two 128-row workgroups per CU, 2-stage weight ring) or 8
// (one 256-row workgroup per CU, 4-stage ring: measured slower, kept parity-tested as the A/B arm).
int sihl_mlp_rows_config(int waves) {
  if (waves != 4 && waves != 8) return SIHL_EARG;
  g_mlp_rows_waves = waves;
  return SIHL_OK;
}

// Tuning ablation (`make TUNING=1` builds only; results invalid when non-zero): see MlpRowsParams::dbg.
#ifdef SIHL_TUNING
int sihl_mlp_rows_debug(int mode) { g_mlp_rows_dbg = mode; return SIHL_OK; }
#else
int sihl_mlp_rows_debug(int mode) { return mode == 0 ? SIHL_OK : SIHL_EARG; }  // the shipped library has no ablation state
#endif

// The K order sihl_mlp_rows_fwd wants for the weights of layers >= 1: w_out [Cout][K] from w_in [Cout][K] (bf16, K a
// multiple of 16), inside each group of 16 input channels [0-3, 8-11, 4-7, 12-15].  Its own inverse.
int sihl_mlp_permute_k(const void* w_in, void* w_out, long Cout, int K, hipStream_t stream) {
  if (!w_in || !w_out || Cout <= 0 || K <= 0 || K % 16) return SIHL_EARG;
  const long n4 = Cout * K / 4;
  long g = (n4 + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(mlp_permute_k_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const uint2*)w_in, (uint2*)w_out, n4);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}

// One MLP of a sihl_mlp_rows_fwd_multi launch (include/sihl_hip.h: sihl_mlp_call)
struct sihl_mlp_call {
  const void* x; long x_stride; long rows; int Cin, C, nhidden, Cout, out_stride; float eps;
  const void* const* w; const float* const* bias; const float* const* gamma; const float* const* beta;
  void* out;
};

// n <= 4 MLPs in ONE launch, each as sihl_mlp_rows_fwd (same activation for all; they may share x).
int sihl_mlp_rows_fwd_multi(const sihl_mlp_call* calls, int n, int act, int dtype, hipStream_t stream) {
  if (!calls || n < 1 || n > MLPR_MAXN) return SIHL_EARG;
  MlpRowsBatch pb;
  int maxco = 0, blocks = 0;
  for (int k = 0; k < MLPR_MAXN; ++k) pb.first_block[k] = 0x7fffffff;
  for (int k = 0; k < n; ++k) {
    const sihl_mlp_call& c = calls[k];
    if (!c.x || !c.out || !c.w || !c.bias || !c.gamma || !c.beta) return SIHL_EARG;
    if (!sihl_mlp_rows_supported(c.rows, c.Cin, c.C, c.Cout, c.nhidden, act, dtype)) return SIHL_EARG;
    if (c.out_stride < c.Cout || c.out_stride % 8 || c.out_stride > 256 || c.x_stride < c.Cin || c.x_stride % 8) return SIHL_EARG;
    MlpRowsParams& p = pb.p[k];
    p.x = c.x; p.out = c.out; p.x_stride = c.x_stride; p.out_stride = c.out_stride;
    p.rows = (int)c.rows; p.Cin = c.Cin; p.C = c.C; p.Cout = c.Cout; p.nhidden = c.nhidden; p.eps = c.eps;
    p.dbg = g_mlp_rows_dbg;
    for (int l = 0; l <= MLPR_MAXL; ++l) { p.w[l] = nullptr; p.bias[l] = nullptr; }
    for (int l = 0; l < MLPR_MAXL; ++l) { p.gamma[l] = nullptr; p.beta[l] = nullptr; }
    for (int l = 0; l <= c.nhidden; ++l) {
      if (!c.w[l]) return SIHL_EARG;
      p.w[l] = c.w[l];
      p.bias[l] = c.bias[l];
    }
    for (int l = 0; l < c.nhidden; ++l) {
      if (!c.gamma[l] || !c.beta[l]) return SIHL_EARG;
      p.gamma[l] = c.gamma[l];
      p.beta[l] = c.beta[l];
    }
    pb.first_block[k] = blocks;
    const int rbm = g_mlp_rows_waves == 8 ? 256 : 128;  // rows per workgroup of the configuration launch_mlp_rows picks
    blocks += (p.rows + rbm - 1) / rbm;
    maxco = c.Cout > maxco ? c.Cout : maxco;
  }
  for (int k = n; k <= MLPR_MAXN; ++k) pb.first_block[k] = k == n ? blocks : 0x7fffffff;
  pb.first_block[n] = blocks;
  for (int k = n; k < MLPR_MAXN; ++k) pb.p[k] = pb.p[0];
  const float floor_ = act == SIHL_ACT_RELU ? 0.f : -__builtin_inff();
#define SIHL_MLPR(S)                                                          \
  do {                                                                        \
    if (maxco <= 32) return launch_mlp_rows<S, 1>(pb, n, floor_, stream);     \
    if (maxco <= 96) return launch_mlp_rows<S, 3>(pb, n, floor_, stream);     \
    return launch_mlp_rows<S, 8>(pb, n, floor_, stream);                      \
  } while (0)
  if (act == SIHL_ACT_SILU) SIHL_MLPR(1);
  SIHL_MLPR(0);
#undef SIHL_MLPR
}

// out[rows][out_stride] = Linear_n( act(LN(Linear_{n-1}( ... act(LN(Linear_0(x))) ... ))) ), one launch, activations in
// registers.  Arguments as sihl_mlp_fwd, except: nhidden >= 1, C == 256, and w[l] for l >= 1 in the K order of
// sihl_mlp_permute_k (w[0] plain).
int sihl_mlp_rows_fwd(const void* x, long x_stride, long rows, int Cin, int C, int nhidden, const void* const* w,
                      const float* const* bias, const float* const* gamma, const float* const* beta, float eps, int act,
                      int Cout, void* out, int out_stride, int dtype, hipStream_t stream) {
  sihl_mlp_call c;
  c.x = x; c.x_stride = x_stride; c.rows = rows; c.Cin = Cin; c.C = C; c.nhidden = nhidden; c.Cout = Cout;
  c.out_stride = out_stride; c.eps = eps; c.w = w; c.bias = bias; c.gamma = gamma; c.beta = beta; c.out = out;
  return sihl_mlp_rows_fwd_multi(&c, 1, act, dtype, stream);
}

}  // extern "C"
