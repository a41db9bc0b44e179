// Host side of the persistent 256x256 bf16 conv kernel (conv_p8.h): eligibility and launch.
#include "conv_p8.h"
#include "profile.h"

namespace {
bool g_p8 = false;  // opt-in (sihl_conv2d_p8_enable): measured at parity with the two-stage tile on L3 3x3, slower on thin layers
int p8_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
}  // namespace

bool sihl_p8_eligible(const ConvParams& p) {
  if (!g_p8 || p.in_dilate != 1 || p.splits != 1 || p.Cin % 64 != 0 || p.Cout % 8 != 0 || p.out_s != 1) return false;
  if (p.out_image_stride != (long)p.Ho * p.Wo * p.Cout || p.KH * p.KW > 16) return false;
  if (p.add && (p.add_stride != 1 || p.act != SIHL_ACT_NONE || p.stats_mode != 0 || p.bias || p.pre_scale || p.post_scale))
    return false;
  if (p.act == SIHL_ACT_SIGMOID) return false;  // rare (tiny attention convs): the generic kernel keeps it
  return true;
}
int sihl_p8_launch(const ConvParams& p0, hipStream_t stream) {
  ConvParams p = p0;
  p.gridM = (p.M + P8_BM - 1) / P8_BM;
  p.gridN = (p.Cout + P8_BN - 1) / P8_BN;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_p8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)conv_p8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = p.gridM * p.gridN;
  const int grid = ntiles < p8_cus() ? ntiles : p8_cus();
  const double flops = 2.0 * p.M * (double)p.Cout * p.KH * p.KW * p.Cin;
  const double bytes = ((double)p.N * p.H * p.W * p.Cin + (double)p.M * p.Cout + (double)p.Cout * p.KH * p.KW * p.Cin) * 2.0;
  sihl_prof_begin(SIHL_PROF_CONV, SIHL_BF16, flops, bytes, stream);
  if (p.add) hipLaunchKernelGGL(conv_p8_kernel<true>, dim3(grid), dim3(P8_THREADS), P8_LDS, stream, p);
  else hipLaunchKernelGGL(conv_p8_kernel<false>, dim3(grid), dim3(P8_THREADS), P8_LDS, stream, p);
  sihl_prof_end(stream);
  SIHL_CHECK_LAUNCH();
  return SIHL_OK;
}


void sihl_p8_set_enabled(bool on) { g_p8 = on; }
