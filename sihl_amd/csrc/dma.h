// LDS-DMA helpers shared by the matrix-core kernels (gfx950): raw buffer descriptors and the
// buffer_load_dwordx4 ... lds wave-instruction (64 lanes x 16 B -> 1 KiB of LDS, linear in lane order).
#pragma once
#include "common.h"

namespace {

constexpr unsigned OOB = 0xFFFFFFF0u;  // voffset beyond num_records: the bounds check returns zeros

typedef int v4i_t __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (base, stride 0, num_records bytes, raw 32-bit format) from wave-uniform values.
__device__ __forceinline__ v4i_t make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long a = (unsigned long)ptr;
  v4i_t r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B -> 1 KiB of LDS at the wave-uniform byte address lds_dst.
// Issued through inline asm ON PURPOSE: hipcc would otherwise put s_waitcnt vmcnt(0) in front of the next
// ds_read (it cannot tell that the DMA targets the OTHER stage buffer) and serialise DMA with MFMA.  The
// main loop therefore waits for the DMA itself (s_waitcnt vmcnt(0) before the barrier that publishes the
// stage).  M0 (the LDS-DMA destination base) is written in the same statement that reads it; nothing else in
// these kernels uses M0 (gfx9 ds_* instructions do not).
// s_nop 3 (with the s_mov: 5 wait states): these kernels spill scalar registers to VGPR lanes (100-600 slots), the
// compiler may reload the descriptor with v_readlane right in front of this statement, and "VALU writes SGPR -> VMEM
// reads that SGPR" needs 5 wait states which the hazard recogniser cannot insert for an instruction inside inline asm.
// (s_nop 0 only covered M0.  Found in conv_pw.hip, where a statistics store ran with a stale descriptor and was
// dropped; whether a DMA of the older kernels ever issued inside the window depends on what else the CU was doing.)
__device__ __forceinline__ void dma16(unsigned voff, unsigned lds_dst, v4i_t rsrc) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 3\n\t"
      "buffer_load_dwordx4 %0, %2, 0 offen lds"
      :
      : "v"(voff), "s"(lds_dst), "s"(rsrc)
      : "memory");
}

// wait until at most N of this wave's newest vector-memory operations are still in flight (they retire in order)
template <int N> __device__ __forceinline__ void wait_vm_keep() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
