// Micro-benchmark (standalone, no torch): what bounds the inner product step of the bf16 GEMM tiles on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_mfma.hip -o gpurun_out/ubench && gpurun_out/ubench
// Each workgroup = 4 waves working on a 128 x 128 x 64 tile-K-step held in LDS (32 KiB, XOR swizzled as in gemm_nt.hip),
// repeated ITER times; 2 workgroups per CU as in the product kernel.  Variants:
//   0  MFMA only            16x16x32, operands stay in registers
//   1  LDS reads only       8 x ds_read_b128 per fragment step, results consumed by a cheap XOR
//   2  LDS + MFMA           the product loop (16x16x32: 8 reads -> 16 MFMA)
//   3  LDS + MFMA 32x32x16  4 x ds_read_b128 -> 4 MFMA per 16-wide k slice (same LDS bytes per flop: the wave tile sets them)
//   4  MFMA only            32x32x16
// Prints cycles per tile-K-step per CU (wall time x clock / steps), to be compared with 512 = the MFMA floor.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int swz(int r, int ch) { return r * 128 + ((ch ^ (r & 7)) << 4); }

template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    for (int i = tid; i < 2 * 16384 / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const unsigned char* sA = smem;
    const unsigned char* sB = smem + 16384;
    float sink = 0.f;
    if constexpr (V == 0 || V == 1 || V == 2) {
        f32x4 acc[4][4];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
        bf16x8 af[4], bf[4];
        for (int m = 0; m < 4; ++m) { af[m] = *(const bf16x8*)(sA + swz(wr * 64 + m * 16 + (lane & 15), lane >> 4)); bf[m] = af[m]; }
        unsigned x = 0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");                    // re-read LDS every iteration
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ch = s * 4 + (lane >> 4);
                if constexpr (V != 0) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[m] = *(const bf16x8*)(sA + swz(wr * 64 + m * 16 + (lane & 15), ch));
#pragma unroll
                    for (int n = 0; n < 4; ++n) bf[n] = *(const bf16x8*)(sB + swz(wc * 64 + n * 16 + (lane & 15), ch));
                }
                if constexpr (V == 1) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) asm volatile("" :: "v"(af[m]), "v"(bf[m]));     // consume, no ALU work
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
                }
            }
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) sink += acc[m][n][0] + acc[m][n][3];
        sink += (float)x;
    } else {
        // 32x32x16: A fragment = lane (l&31) row, k = (l>>5)*8 .. +8 (bf16x8), one MFMA consumes k=16
        f32x16 acc[2][2];
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
        bf16x8 af[2], bf[2];
        for (int m = 0; m < 2; ++m) { af[m] = *(const bf16x8*)(sA + swz(wr * 64 + m * 32 + (lane & 31), lane >> 5)); bf[m] = af[m]; }
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");                    // re-read LDS every iteration
#pragma unroll
            for (int s = 0; s < 4; ++s) {                       // 4 k-slices of 16
                const int ch = s * 2 + (lane >> 5);
                if constexpr (V == 3) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) af[m] = *(const bf16x8*)(sA + swz(wr * 64 + m * 32 + (lane & 31), ch));
#pragma unroll
                    for (int n = 0; n < 2; ++n) bf[n] = *(const bf16x8*)(sB + swz(wc * 64 + n * 32 + (lane & 31), ch));
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
            }
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) sink += acc[m][n][0] + acc[m][n][15];
    }
    if (sink == 12345.678f) out[blockIdx.x * 256 + tid] = sink;
}

// Variant 5/6: the TN (dW) kernel's inner step: operands by ds_read_b64_tr_b16 (two per 16x16x32 fragment) from a
// [64 m][256 B] tile pair, 32 transpose reads + 32 MFMAs per wave and step.  WPS = workgroups per CU (2 -> 2 waves/SIMD, 4 -> 4).
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int WPS>
__global__ __launch_bounds__(256, WPS) void ktr(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    for (int i = tid; i < 2 * 16384 / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto frag = [&](const unsigned char* tile, int colbase, int s) {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int r = s * 32 + 8 * g + q, c32 = colbase >> 4;
        const int f = (r & 3) | (((r >> 3) & 1) << 2);
        const int off0 = r * 256 + ((c32 ^ f) << 5) + (p << 3);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0 + 1024));
        union { struct { s16x4 a, b; } s; bf16x8 v; } u; u.s.a = lo; u.s.b = hi;
        return u.v;
    };
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = frag(smem, wr * 64 + m * 16, s);
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = frag(smem + 16384, wc * 64 + n * 16, s);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
    }
    float sink = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) sink += acc[m][n][0] + acc[m][n][3];
    if (sink == 12345.678f) out[blockIdx.x * 256 + tid] = sink;
}

template <int WPS>
static void run_tr(const char* name, float* out, int iters, double ghz) {
    const int blocks = 256 * WPS * 4;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    ktr<WPS><<<blocks, 256>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    ktr<WPS><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double us_per_step = ms * 1e3 / ((double)blocks / 256 * iters);
    printf("%-24s %8.3f ms  %7.1f ns / tile-step / CU  (%6.0f cycles @ %.1f GHz)  %7.1f TFLOP/s-equivalent\n", name, ms, us_per_step * 1e3,
           us_per_step * 1e3 * ghz, ghz, 2.0 * 128 * 128 * 64 * blocks * iters / (ms * 1e-3) / 1e12);
}

template <int V>
static void run(const char* name, float* out, int iters, double ghz) {
    const int blocks = 256 * 2 * 4;                 // 4 rounds of 2 workgroups per CU
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<V><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<V><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double steps_per_cu = (double)blocks / 256 * iters;
    const double us_per_step = ms * 1e3 / steps_per_cu;
    const double tf = 2.0 * 128 * 128 * 64 * blocks * iters / (ms * 1e-3) / 1e12;
    printf("%-24s %8.3f ms  %7.1f ns / tile-K-step / CU  (%6.0f cycles @ %.1f GHz)  %7.1f TFLOP/s-equivalent\n", name, ms, us_per_step * 1e3,
           us_per_step * 1e3 * ghz, ghz, tf);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float* out; hipMalloc(&out, 256 * 4 * 4 * 256 * 4);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const double ghz = clk / 1e6;
    printf("device clock %.2f GHz, %d iterations per workgroup\n", ghz, iters);
    run<0>("mfma16x16x32 only", out, iters, ghz);
    run<4>("mfma32x32x16 only", out, iters, ghz);
    run<1>("lds b128 reads only", out, iters, ghz);
    run<2>("lds + mfma16x16x32", out, iters, ghz);
    run<3>("lds + mfma32x32x16", out, iters, ghz);
    run_tr<2>("tr_b64 + mfma, 2 wg/CU", out, iters, ghz);
    run_tr<4>("tr_b64 + mfma, 4 wg/CU", out, iters, ghz);
    return 0;
}
