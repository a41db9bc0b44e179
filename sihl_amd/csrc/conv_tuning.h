// Test / tuning switches of the implicit-GEMM conv (set through the sihl_conv2d_* hooks of conv_igemm.hip, read by the
// dispatch in conv_igemm_impl.h).  One object for the whole library: the dispatch lives in two translation units.
#pragma once
#include <hip/hip_runtime.h>

struct ConvTuning {
  bool force_reg;        // test hook: use the register-staged loader
  int dbg;               // tuning ablations (SIHL_TUNING builds)
  int tile_override;     // test / tuning hook: 0 = heuristic, 128 / 256 = force that pixel-tile size
  int nbuf;              // tuning hook: LDS stages of the narrow LDS-DMA tiles (0 = default)
  bool splitk;           // tuning / test hook: split-K for tiny pyramid levels
  bool strided_classes;  // tuning / test hook: parity-class dgrad of 3x3 stride-2 convs (else zero-dilated read)
  int rules_off;         // tuning hook: bit 0 = no single-stage narrow tiles, bit 1 = no 128x128 routing of thin pointwise layers, bit 2 = K loops in lockstep
  int krot;              // tuning hook: stage stride between neighbouring workgroups' K-loop starts (default 200013 = groups of 4 workgroups, stride 13; see sihl_conv2d_krot)
};
extern ConvTuning sihl_conv_tuning;
#define g_force_reg (sihl_conv_tuning.force_reg)
#define g_dbg (sihl_conv_tuning.dbg)
#define g_tile_override (sihl_conv_tuning.tile_override)
#define g_nbuf (sihl_conv_tuning.nbuf)
#define g_splitk (sihl_conv_tuning.splitk)
#define g_strided_classes (sihl_conv_tuning.strided_classes)
#define g_rules_off (sihl_conv_tuning.rules_off)
#define g_krot (sihl_conv_tuning.krot)

constexpr int BM128 = 128;  // pixels per BatchNorm partial row

struct ConvParams;
int sihl_conv_dispatch_bf16(const ConvParams& p, hipStream_t stream);  // conv_igemm_bf16.hip
int sihl_conv_dispatch_f32(const ConvParams& p, hipStream_t stream);   // conv_igemm_f32.hip
