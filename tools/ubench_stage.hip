// Micro-benchmark (standalone): cost of getting one K step (A tile 128x64 + W tile 128x64 bf16 = 32 KiB) from global memory
// into LDS next to the MFMAs of the previous step, for the two ways gfx950 offers:
//   V=0  register-staged: 8 x global_load_dwordx4 per lane (two steps ahead) -> 8 x ds_write_b128      (what gemm_nt.hip does)
//   V=1  LDS-DMA:         8 x global_load_lds_dwordx4 per lane straight into a 3-stage ring            (no VGPRs, no ds_write)
//   V=2  no operand traffic at all (LDS reads + MFMA + the same barriers): the floor of this loop structure
// Same loop skeleton as the product kernel: double/triple-buffered LDS, ONE barrier per K step, fragment registers
// double-buffered (reads one fragment step ahead), 4 waves per workgroup, 2 workgroups per CU.  Operands come from a
// buffer that fits L2 (so HBM does not interfere) and the tiles are laid out swizzled as in the product.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_stage.hip -o tools/bin/ubench_stage && tools/bin/ubench_stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int swz(int r, int ch) { return r * 128 + ((ch ^ (r & 7)) << 4); }

template <int V>
__global__ __launch_bounds__(256, 2) void k(const unsigned char* __restrict__ g, long tile_stride, int ntiles, float* out, int nk) {
    constexpr int NBUF = V == 1 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // NBUF x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const unsigned char* src = g + (long)(blockIdx.x % ntiles) * tile_stride;          // this workgroup's operand stream
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    bf16x8 f0a[4], f0b[4], f1a[4], f1b[4];
    f32x4 ra[2][8];                                            // V=0: two register sets of 8 x 16 bytes
    auto fetch = [&](int set, int kt) {                       // V=0
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i;                      // chunk 0..2047 of the 32 KiB step: row c>>3, 16-byte chunk c&7
            ra[set][i] = *(const f32x4*)(src + (long)kt * 32768 + (long)c * 16);
        }
    };
    auto stage = [&](int set, int buf) {                      // V=0: registers -> LDS (swizzled)
        unsigned char* s = smem + buf * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + 256 * i, r = (c >> 3) & 127, ch = c & 7, half = c >> 10;
            *(f32x4*)(s + half * 16384 + swz(r, ch)) = ra[set][i];
        }
    };
    auto dma = [&](int kt, int buf) {                         // V=1: 8 pieces of 1 KiB per wave, lane-linear LDS image
        unsigned char* s = smem + buf * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = wid + 4 * i;                        // 1 KiB piece 0..31 = 8 rows
            const int lrow = lane >> 3, lch = (lane & 7) ^ (lrow & 7);      // the swizzle goes on the SOURCE address
            const unsigned char* q = src + (long)kt * 32768 + (long)p * 1024 + lrow * 128 + lch * 16;
            __builtin_amdgcn_global_load_lds((glb_void*)q, (lds_void*)(s + p * 1024), 16, 0, 0);
        }
    };
    auto rd = [&](bf16x8 (&af)[4], bf16x8 (&bf)[4], int buf, int s) {
        const unsigned char* sA = smem + buf * 32768;
        const unsigned char* sB = sA + 16384;
        const int ch = s * 4 + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *(const bf16x8*)(sA + swz(wr * 64 + m * 16 + (lane & 15), ch));
#pragma unroll
        for (int n = 0; n < 4; ++n) bf[n] = *(const bf16x8*)(sB + swz(wc * 64 + n * 16 + (lane & 15), ch));
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
    };

    if constexpr (V == 0) {
        fetch(0, 0); fetch(1, 1); stage(0, 0); fetch(0, 2);
        __syncthreads();
        rd(f0a, f0b, 0, 0);
        for (int kt = 0; kt + 4 < nk; kt += 2) {
            rd(f1a, f1b, 0, 1); mma(f0a, f0b); stage(1, 1); __syncthreads(); fetch(1, kt + 3); rd(f0a, f0b, 1, 0); mma(f1a, f1b);
            rd(f1a, f1b, 1, 1); mma(f0a, f0b); stage(0, 0); __syncthreads(); fetch(0, kt + 4); rd(f0a, f0b, 0, 0); mma(f1a, f1b);
        }
    } else if constexpr (V == 3) {
        // V=0 with the 8 ds_writes spread between the MFMAs of the first fragment step and the 8 global loads between the
        // MFMAs of the second (sched_group_barrier: 2 MFMA, 1 memory op, ...) instead of two bursts
        auto mma_stage = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[4], int set, int buf) {
            mma(af, bf); stage(set, buf);
#pragma unroll
            for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
        };
        auto mma_fetch = [&](const bf16x8 (&af)[4], const bf16x8 (&bf)[4], int set, int kt) {
            mma(af, bf); fetch(set, kt);
#pragma unroll
            for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        };
        fetch(0, 0); fetch(1, 1); stage(0, 0); fetch(0, 2);
        __syncthreads();
        rd(f0a, f0b, 0, 0);
        for (int kt = 0; kt + 4 < nk; kt += 2) {
            rd(f1a, f1b, 0, 1); mma_stage(f0a, f0b, 1, 1); __syncthreads(); rd(f0a, f0b, 1, 0); mma_fetch(f1a, f1b, 1, kt + 3);
            rd(f1a, f1b, 1, 1); mma_stage(f0a, f0b, 0, 0); __syncthreads(); rd(f0a, f0b, 0, 0); mma_fetch(f1a, f1b, 0, kt + 4);
        }
    } else if constexpr (V == 1) {
        // ring of 3: step kt computes buffer kt%3 while kt+1 has landed or is landing and kt+2 is issued
        dma(0, 0); dma(1, 1);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // step 0 has landed (this wave's pieces), then everyone's
        rd(f0a, f0b, 0, 0);
        int b0 = 0, b1 = 1, b2 = 2;
        for (int kt = 0; kt + 4 < nk; ++kt) {
            rd(f1a, f1b, b0, 1);
            mma(f0a, f0b);
            dma(kt + 2, b2);                                  // buffer b2 was read last in step kt-1 (behind the barrier below)
            // step kt+1 has landed (only the 8 pieces just issued may be pending), LDS reads drained; a raw barrier:
            // __syncthreads() would add vmcnt(0) for the LDS-writing loads and drain the ring
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            rd(f0a, f0b, b1, 0);
            mma(f1a, f1b);
            const int t = b0; b0 = b1; b1 = b2; b2 = t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        __syncthreads();
        rd(f0a, f0b, 0, 0);
        for (int kt = 0; kt + 4 < nk; kt += 2) {
            rd(f1a, f1b, 0, 1); mma(f0a, f0b); __syncthreads(); rd(f0a, f0b, 1, 0); mma(f1a, f1b);
            rd(f1a, f1b, 1, 1); mma(f0a, f0b); __syncthreads(); rd(f0a, f0b, 0, 0); mma(f1a, f1b);
        }
    }
    float sink = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) sink += acc[m][n][0] + acc[m][n][3];
    if (sink == 12345.678f) out[blockIdx.x * 256 + tid] = sink;
}

template <int V>
static void run(const char* name, const unsigned char* g, long tile_stride, int ntiles, float* out, int nk, double ghz) {
    const int blocks = 256 * 2 * 2;                 // 2 rounds of 2 workgroups per CU
    const int lds = (V == 1 ? 3 : 2) * 32768;
    (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<V><<<blocks, 256, lds>>>(g, tile_stride, ntiles, out, nk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<V><<<blocks, 256, lds>>>(g, tile_stride, ntiles, out, nk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double steps_per_cu = (double)blocks / 256 * (nk - 4);
    const double ns = ms * 1e6 / steps_per_cu;
    printf("%-44s %8.3f ms  %7.1f ns / tile-K-step / CU  (%5.0f cycles @ %.1f GHz)  %6.0f TFLOP/s-eq  %5.2f TB/s operand ingest\n", name, ms, ns,
           ns * ghz, ghz, 2.0 * 128 * 128 * 64 / ns / 1e3 * 256, V == 2 ? 0.0 : 32768.0 / ns * 256 / 1e3);
}

int main(int argc, char** argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 260;        // K steps per workgroup
    const int ntiles = 8;                                  // distinct operand streams: 8 x nk x 32 KiB (68 MB at nk=260: L2 + Infinity Cache)
    const long tile_stride = (long)nk * 32768;
    unsigned char* g; (void)hipMalloc(&g, (size_t)ntiles * tile_stride + 65536);
    (void)hipMemset(g, 0x3c, (size_t)ntiles * tile_stride + 65536);
    float* out; (void)hipMalloc(&out, 1024 * 256 * 4);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    printf("device clock %.2f GHz, %d K steps per workgroup, 2 workgroups of 4 waves per CU\n", ghz, nk);
    run<2>("no operand traffic (reads + MFMA + barriers)", g, tile_stride, ntiles, out, nk, ghz);
    run<0>("register-staged (global_load -> ds_write_b128)", g, tile_stride, ntiles, out, nk, ghz);
    run<1>("LDS-DMA (global_load_lds_dwordx4, ring of 3)", g, tile_stride, ntiles, out, nk, ghz);
    run<3>("register-staged, memory ops spread between MFMAs", g, tile_stride, ntiles, out, nk, ghz);
    return 0;
}
