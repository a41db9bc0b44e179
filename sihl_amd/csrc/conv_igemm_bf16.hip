// bf16 instantiations of the implicit-GEMM conv kernels (conv_igemm_impl.h).
#include "conv_igemm_impl.h"

int sihl_conv_dispatch_bf16(const ConvParams& p, hipStream_t stream) { return dispatch<bf16_t>(p, stream); }

int sihl_conv_splitk_finish_bf16(const ConvParams& p, hipStream_t stream) {
  launch_splitk_epilogue<bf16_t>(p, stream);
  return SIHL_OK;
}
