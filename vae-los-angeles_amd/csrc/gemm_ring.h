// LDS-ring main loops for the bf16 GEMMs (gfx950): operands go HBM/L2 -> LDS directly
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
#include "gemm_nt_epi.h"

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

// ------------------------------------------------------------------------------------------
// NT:  C[M,N] = epi( A[M,K] (bf16, rows padded to 8 elements) x W[N,K]^T )
// ------------------------------------------------------------------------------------------
template <typename Epi>
__global__ __launch_bounds__(NTHREADS, 1)
void gemm_nt_ring_kernel(const bf16* __restrict__ A, long lda, unsigned a_bytes, const bf16* __restrict__ W, long ldw,
                         int M, int N, int K, int gx, int gy, Epi epi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)(smem + RING_NS * RING_STAGE);

    const int L = blockIdx.x, slot = L >> 3;
    const int ct = slot % gy, rt = (slot / gy) * 8 + (L & 7);
    if (rt >= gx) return;
    const int row0 = rt * TILE, col0 = ct * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    // piece p (1 KiB) = tile rows 8p..8p+7; lane -> row 8p + (lane>>3), physical chunk lane&7, which holds
    // logical 16-byte chunk (lane&7) ^ (row&7) of that row (same involution as swz() on the read side)
    const int lrow = lane >> 3, lch = (lane & 7) ^ (lrow & 7);
    unsigned a_off[4];
    const bf16* w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (wid + 4 * i) + lrow;
        a_off[i] = (unsigned)(((long)(row0 + r) * lda + lch * 8) * 2);
        w_ptr[i] = W + (long)(col0 + r) * ldw + lch * 8;
    }
    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % RING_NS) * RING_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wid + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(st + p * 1024), 16, a_off[i] + kt * 128, 0, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(w_ptr[i] + kt * 64), (lds_void*)(st + 16384 + p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + 63) / 64;
    EpiPrefetch pf;
    if (!epi.accum()) nt_epilogue_prefetch<bf16, Epi, 2>(pf, epi, row0, col0, M, N, tid);
#pragma unroll
    for (int s = 0; s < RING_NS - 1; ++s) if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        ring_wait(min(RING_NS - 2, nk - 1 - kt));        // this wave's share of stage kt has landed
        ring_barrier();                                   // ... and everyone's; stage kt-1 is no longer being read
        if (kt + RING_NS - 1 < nk) issue(kt + RING_NS - 1);
        const unsigned char* sA = smem + (kt % RING_NS) * RING_STAGE;
        const unsigned char* sB = sA + 16384;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
            const int ch = s * 4 + (lane >> 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) { const int r = wr * 64 + m * 16 + (lane & 15); af[m] = *(const bf16x8*)(sA + r * 128 + ((ch ^ (r & 7)) << 4)); }
#pragma unroll
            for (int n = 0; n < 4; ++n) { const int r = wc * 64 + n * 16 + (lane & 15); bfr[n] = *(const bf16x8*)(sB + r * 128 + ((ch ^ (r & 7)) << 4)); }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<bf16>::mma(acc[m][n], af[m], bfr[n]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ring_barrier();                                       // ring is free: the epilogue reuses it as scratch
    nt_epilogue<bf16, Epi, 2>(smem, red, acc, epi, pf, row0, col0, M, N, tid, lane, wr, wc);
}

template <typename Epi>
static int launch_nt_ring(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int gx = (M + TILE - 1) / TILE, gy = (N + TILE - 1) / TILE;
    const int grid = ((gx + 7) / 8) * 8 * gy;
    const unsigned a_bytes = (unsigned)((long)M * lda * 2);
    hipLaunchKernelGGL((gemm_nt_ring_kernel<Epi>), dim3(grid), dim3(NTHREADS), RING_LDS, st,
                       (const bf16*)A, lda, a_bytes, (const bf16*)W, ldw, M, N, K, gx, gy, epi);
    MM_CHECK_LAUNCH();
    return 0;
}

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
