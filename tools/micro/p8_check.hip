// Developer harness (GPU box): the persistent 8-phase conv kernel (csrc/conv_p8.h) against the two-stage 256x256 tile
// it replaces - element-wise agreement of outputs and BatchNorm partial rows, a CPU spot check in double precision,
// and HIP-event timings on rotating operands.  Build: make -C tools/micro p8_check; run: tools/micro/p8_check [quick]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../include/sihl_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

struct Shape { const char* name; int N, H, W, Cin, Cout, K, stride, pad; };

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  std::vector<Shape> shapes = {
      {"L3 3x3", 32, 64, 64, 256, 256, 3, 1, 1},
      {"ragged M 3x3", 3, 150, 150, 64, 256, 3, 1, 1},        // 67500 pixels: last tile partly beyond M
      {"r2 1x1 128>512", 32, 64, 64, 128, 512, 1, 1, 0},
      {"lat3 1x1 512>256", 32, 64, 64, 512, 256, 1, 1, 0},
      {"stride2 3x3", 8, 256, 256, 64, 256, 3, 2, 1},
      {"r1 1x1 64>256", 32, 128, 128, 64, 256, 1, 1, 0},
      {"mlp 1x1 256>256", 1, 1, 174592, 256, 256, 1, 1, 0},
      {"Cout 320 1x1", 4, 128, 128, 128, 320, 1, 1, 0},       // channel tail: second channel tile has 64 of 256
  };
  int bad = 0;
  for (const Shape& s : shapes) {
    const int Ho = (s.H + 2 * s.pad - (s.K - 1) - 1) / s.stride + 1, Wo = (s.W + 2 * s.pad - (s.K - 1) - 1) / s.stride + 1;
    const long M = (long)s.N * Ho * Wo;
    const long nin = (long)s.N * s.H * s.W * s.Cin, nw = (long)s.Cout * s.K * s.K * s.Cin, nout = M * s.Cout;
    const int NB = 4;
    std::vector<uint16_t> hin(nin), hw(nw);
    for (auto& v : hin) v = f2bf(urand());
    const float wscale = 1.0f / sqrtf((float)(s.K * s.K * s.Cin));
    for (auto& v : hw) v = f2bf(urand() * wscale * 1.7f);
    std::vector<float> hbias(s.Cout), hsc(s.Cout), hsh(s.Cout);
    for (int c = 0; c < s.Cout; ++c) { hbias[c] = urand() * 0.3f; hsc[c] = 0.5f + fabsf(urand()); hsh[c] = urand(); }
    uint16_t* din[NB]; uint16_t *dw, *dout0, *dout1;
    float *dbias, *dsc, *dsh, *dst0, *dst1;
    for (int i = 0; i < NB; ++i) { CK(hipMalloc(&din[i], nin * 2)); CK(hipMemcpy(din[i], hin.data(), nin * 2, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&dw, nw * 2)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout0, nout * 2)); CK(hipMalloc(&dout1, nout * 2));
    CK(hipMalloc(&dbias, s.Cout * 4)); CK(hipMalloc(&dsc, s.Cout * 4)); CK(hipMalloc(&dsh, s.Cout * 4));
    CK(hipMemcpy(dbias, hbias.data(), s.Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsc, hsc.data(), s.Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsh, hsh.data(), s.Cout * 4, hipMemcpyHostToDevice));
    const int rows = sihl_conv2d_stat_rows(M);
    const long stb = (long)rows * 2 * s.Cout * 4;
    CK(hipMalloc(&dst0, stb)); CK(hipMalloc(&dst1, stb));
    std::vector<uint16_t> o0(nout), o1(nout);
    std::vector<float> s0((size_t)rows * 2 * s.Cout), s1((size_t)rows * 2 * s.Cout);
    // mode 0: relu + stats after act (training ConvNormAct); 1: bias + stats before act (conv -> BN -> act); 2: eval: relu +
    // post-affine; 3: silu, no stats; 4: bias + pre-affine + relu + post-affine
    for (int mode = 0; mode < (quick ? 1 : 5); ++mode) {
      const int act = mode == 0 ? 1 : mode == 1 ? 0 : mode == 2 ? 1 : mode == 3 ? 2 : 1;
      const int stats = mode == 0 ? 2 : mode == 1 ? 1 : 0;
      const float* bias = (mode == 1 || mode == 4) ? dbias : nullptr;
      const float* pre_s = mode == 4 ? dsc : nullptr; const float* pre_t = mode == 4 ? dsh : nullptr;
      const float* post_s = (mode == 2 || mode == 4) ? dsc : nullptr; const float* post_t = (mode == 2 || mode == 4) ? dsh : nullptr;
      for (int v = 0; v < 2; ++v) {
        sihl_conv2d_p8_enable(v);
        sihl_conv2d_debug(v && getenv("P8_DBG") ? atoi(getenv("P8_DBG")) : 0);
        sihl_conv2d_tile_override(256);
        CK(hipMemset(v ? dout1 : dout0, 0xFF, nout * 2));
        CK(hipMemset(v ? dst1 : dst0, 0xFF, stb));
        int rc = sihl_conv2d_fwd(din[0], dw, bias, v ? dout1 : dout0, s.N, s.H, s.W, s.Cin, s.Cout, s.K, s.K, s.stride, s.pad, 1,
                                 1 /*bf16*/, act, pre_s, pre_t, post_s, post_t, stats, stats ? (v ? dst1 : dst0) : nullptr, stb, 0, 0);
        if (rc) { printf("%s mode %d v%d: rc %d\n", s.name, mode, v, rc); return 2; }
        CK(hipDeviceSynchronize());
      }
      sihl_conv2d_debug(0);
      CK(hipMemcpy(o0.data(), dout0, nout * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(o1.data(), dout1, nout * 2, hipMemcpyDeviceToHost));
      double maxref = 0, maxdiff = 0; long nbig = 0;
      for (long i = 0; i < nout; ++i) {
        const float a = bf2f(o0[i]), b = bf2f(o1[i]);
        maxref = std::max(maxref, (double)fabsf(a));
        const double d = fabs((double)a - b);
        if (!(d <= 0.02 * std::max(1.0, (double)fabsf(a)))) ++nbig;  // also catches NaN
        maxdiff = std::max(maxdiff, d);
      }
      double sdiff = 0, smax = 0;
      if (stats) {
        CK(hipMemcpy(s0.data(), dst0, stb, hipMemcpyDeviceToHost));
        CK(hipMemcpy(s1.data(), dst1, stb, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < s0.size(); ++i) { smax = std::max(smax, (double)fabsf(s0[i])); sdiff = std::max(sdiff, fabs((double)s0[i] - s1[i])); if (!(fabs((double)s0[i] - s1[i]) <= 1e-3 * std::max(1.0, (double)fabsf(s0[i])))) ++nbig; }
      }
      // CPU spot check of the new kernel (double accumulation over the bf16 operands), mode 0 only
      double cpudiff = 0;
      if (mode == 0) {
        for (int t = 0; t < 512; ++t) {
          const long m = (t < 16) ? (M - 1 - t) : (long)((rng_state = rng_state * 1664525u + 1013904223u) % (unsigned long)M);
          const int co = (int)((rng_state = rng_state * 1664525u + 1013904223u) % (unsigned)s.Cout);
          const int n = (int)(m / ((long)Ho * Wo)), r = (int)(m % ((long)Ho * Wo)), oy = r / Wo, ox = r % Wo;
          double acc = 0;
          for (int ky = 0; ky < s.K; ++ky) for (int kx = 0; kx < s.K; ++kx) {
            const int iy = oy * s.stride - s.pad + ky, ix = ox * s.stride - s.pad + kx;
            if (iy < 0 || ix < 0 || iy >= s.H || ix >= s.W) continue;
            const uint16_t* ip = &hin[(((long)n * s.H + iy) * s.W + ix) * s.Cin];
            const uint16_t* wp = &hw[(((long)co * s.K + ky) * s.K + kx) * s.Cin];
            for (int c = 0; c < s.Cin; ++c) acc += (double)bf2f(ip[c]) * bf2f(wp[c]);
          }
          const double want = acc > 0 ? acc : 0;
          cpudiff = std::max(cpudiff, fabs(want - bf2f(o1[m * s.Cout + co])) / std::max(1.0, fabs(want)));
        }
        if (cpudiff > 0.02) ++nbig;
      }
      printf("%-18s mode %d: max|ref| %.3f  max|new-old| %.4f  stats max|d| %.4f (of %.1f)  cpu rel %.4f  mismatches %ld %s\n", s.name, mode,
             maxref, maxdiff, sdiff, smax, cpudiff, nbig, nbig ? "FAIL" : "ok");
      bad += nbig != 0;
    }
    // timing: training-mode launch (relu + stats), rotating inputs
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * M * s.Cout * s.K * s.K * s.Cin;
    for (int v = 0; v < 2; ++v) {
      sihl_conv2d_p8_enable(v);
      sihl_conv2d_tile_override(v ? 256 : 0);  // old: whatever the heuristic picks for this shape
      float best = 1e9f, tot = 0; const int reps = 5, iters = 10;
      for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i)
          sihl_conv2d_fwd(din[i % NB], dw, nullptr, dout1, s.N, s.H, s.W, s.Cin, s.Cout, s.K, s.K, s.stride, s.pad, 1, 1, 1, nullptr,
                          nullptr, nullptr, nullptr, 2, dst1, stb, 0, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { best = std::min(best, ms / iters); tot += ms / iters; }
      }
      printf("    %-10s %8.1f us best %8.1f us avg  %7.0f TFLOP/s  (out %.0f MB -> %.2f TB/s)\n", v ? "p8" : "heuristic", best * 1e3, tot / reps * 1e3,
             flops / (best * 1e-3) / 1e12, nout * 2 / 1e6, (nout * 2 + nin * 2) / (best * 1e-3) / 1e12);
    }
    if (getenv("P8_ABLATE")) {
      sihl_conv2d_p8_enable(1); sihl_conv2d_tile_override(256);
      const int modes[] = {0, 32, 32 | 128, 32 | 256, 32 | 128 | 256, 256, 128 | 256, 32 | 16, 32 | 16 | 256};
      for (int dm : modes) {
        sihl_conv2d_debug(dm);
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
          CK(hipEventRecord(e0, 0));
          for (int i = 0; i < 10; ++i)
            sihl_conv2d_fwd(din[i % NB], dw, nullptr, dout1, s.N, s.H, s.W, s.Cin, s.Cout, s.K, s.K, s.stride, s.pad, 1, 1, 1, nullptr,
                            nullptr, nullptr, nullptr, 2, dst1, stb, 0, 0);
          CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (r) best = std::min(best, ms / 10);
        }
        printf("    ablate dbg=%2d (16 noepi 32 one-barrier 128 prio-w4-7 256 dma-in-mma): %8.1f us\n", dm, best * 1e3);
      }
      // cycle shares per segment (debug bit 64): per-wave sums over the launch, averaged over workgroups
      unsigned long long* dstamp; CK(hipMalloc(&dstamp, 256 * 8 * 8 * 8)); 
      for (int dm : {64 | 32, 64 | 32 | 256}) {
        CK(hipMemset(dstamp, 0, 256 * 8 * 8 * 8));
        sihl_conv2d_debug(dm);
        sihl_conv2d_fwd_ws(din[0], dw, nullptr, dout1, s.N, s.H, s.W, s.Cin, s.Cout, s.K, s.K, s.stride, s.pad, 1, 1, 1, nullptr,
                           nullptr, nullptr, nullptr, 2, dst1, stb, 0, dstamp, 256 * 8 * 8 * 8, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(256 * 8 * 8);
        CK(hipMemcpy(hs.data(), dstamp, hs.size() * 8, hipMemcpyDeviceToHost));
        for (int grp = 0; grp < 2; ++grp) {
          double a[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
          for (int wg = 0; wg < 256; ++wg) for (int w = grp * 4; w < grp * 4 + 4; ++w) { const unsigned long long* o = &hs[(wg * 8 + w) * 8]; if (!o[6]) continue; for (int k = 0; k < 7; ++k) a[k] += (double)o[k]; ++n; }
          if (!n) continue;
          const double kt = a[6] / n;
          printf("    stamps dbg=%3d waves %d-%d: per K-tile cycles: read %6.0f issue %6.0f barrier %6.0f mma %6.0f vmwait %6.0f | epilogue per tile %8.0f  (K-tiles/WG %.0f)\n",
                 dm, grp * 4, grp * 4 + 3, a[0] / n / kt, a[1] / n / kt, a[2] / n / kt, a[3] / n / kt, a[4] / n / kt, a[5] / n / (kt / (s.K * s.K * (s.Cin / 64))), kt);
        }
      }
      CK(hipFree(dstamp));
      sihl_conv2d_debug(0);
    }
    sihl_conv2d_p8_enable(1); sihl_conv2d_tile_override(0);
    for (int i = 0; i < NB; ++i) CK(hipFree(din[i]));
    CK(hipFree(dw)); CK(hipFree(dout0)); CK(hipFree(dout1)); CK(hipFree(dbias)); CK(hipFree(dsc)); CK(hipFree(dsh)); CK(hipFree(dst0)); CK(hipFree(dst1));
    fflush(stdout);
  }
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
