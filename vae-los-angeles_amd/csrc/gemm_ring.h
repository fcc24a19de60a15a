// LDS-ring main loop for the bf16 TN GEMM (gfx950; the NT twin was removed once the register-staged NT kernel overtook it): operands go HBM/L2 -> LDS directly
// (buffer_load ... lds / global_load_lds, 16 bytes per lane, no VGPR staging) into a 4-deep ring
// of 32 KiB stages; up to three K steps are in flight behind a COUNTED s_waitcnt vmcnt(N) and one
// raw s_barrier per K step.  The skinny GEMMs of this model are HBM/L2-latency bound, not MFMA
// bound: what matters is bytes in flight per CU, so the ring takes the whole LDS (one workgroup
// of 4 waves per CU) instead of registers.
//
// The LDS image of a DMA is lane-linear (wave-uniform base + lane*16), so the XOR swizzles that
// make ds_read_b128 / ds_read_b64_tr_b16 conflict-free are applied to the per-lane SOURCE address.
// A / P / Q go through a buffer resource: rows past M read as zeros with no bounds branches.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace mm {

constexpr int RING_NS = 4;                 // stages
constexpr int RING_STAGE = 32768;          // bytes per stage: 16 KiB A (or P) + 16 KiB W (or Q)
constexpr int RING_LPS = 8;                // DMA instructions per wave per stage (4 + 4 one-KiB pieces)
constexpr int RING_LDS = RING_NS * RING_STAGE + 2048;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void ring_wait(int stages_in_flight) {      // wave-uniform argument
    if (stages_in_flight >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (stages_in_flight == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void ring_barrier() { asm volatile("s_barrier" ::: "memory"); }

// whether the ring kernel may serve this source: bf16, 16-byte aligned rows, whole tensor < 4 GiB
static inline bool ring_ok(const void* p, long ld, int rows) {
    // Opt-in (MMVAE_RING=1): with one 4-wave workgroup per CU the ~6 us fixed cost per tile (pipeline fill +
    // epilogue) is no longer hidden by a co-resident workgroup, so on this model's short-K GEMMs the ring is
    // currently SLOWER than the register-staged kernels (tools/bench_gemm.py).
    static const bool enabled = getenv("MMVAE_RING") != nullptr;
    if (!enabled) return false;
    return ld % 8 == 0 && ((uintptr_t)p & 15) == 0 && (long)rows * ld * 2 < (1L << 32) - (1 << 20);
}

}  // namespace mm
