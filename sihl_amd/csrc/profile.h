// Optional per-launch timing of the two matrix-core kernels with HIP events recorded on the launch stream
// (bench.py's roofline leg).  Off by default: one branch per launch.
#pragma once
#include <hip/hip_runtime.h>

#define SIHL_PROF_CONV 0   // conv_igemm (forward / dgrad / linear)
#define SIHL_PROF_WGRAD 1  // conv_wgrad main kernel

void sihl_prof_begin(int slot, int dtype, double flops, double bytes, hipStream_t stream);
void sihl_prof_end(hipStream_t stream);
