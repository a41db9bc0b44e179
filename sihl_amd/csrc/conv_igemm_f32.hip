// fp32 instantiations (exact-fp32 MFMA, the 1e-4 parity configuration) of the implicit-GEMM conv kernels (conv_igemm_impl.h).
#include "conv_igemm_impl.h"

int sihl_conv_dispatch_f32(const ConvParams& p, hipStream_t stream) { return dispatch<float>(p, stream); }
