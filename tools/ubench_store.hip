// Micro-benchmark (standalone): HBM write rate of a [M][ROWB-byte] row-major matrix as a function of the shape of ONE
// wave-wide 16-byte-per-lane store instruction:  SEG = contiguous bytes per row per instruction (a wave covers 1024/SEG rows).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o tools/bin/ubench_store && tools/bin/ubench_store
// SEG=1024: fully linear (what a fill does); 128: 8 rows x one full line; 64: 16 rows x half a line (MFMA-layout epilogue,
// 4 lanes per row); 32: 32 rows x quarter line.  A second instruction writes the neighbouring segment of the same rows
// right after, as the epilogue does, until the wave's [rows][WSPAN bytes] block is complete.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SEG, int WSPAN>
__global__ __launch_bounds__(256) void k(unsigned char* out, long rowb, int rows_total) {
    // a wave owns RW = 1024/SEG rows x WSPAN bytes per pass over the columns; block = 4 waves stacked over rows
    constexpr int RW = 1024 / SEG, LPR = SEG / 16;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nspan = (int)(rowb / WSPAN);
    const long blk = blockIdx.x;
    const int span = (int)(blk % nspan);
    const long rowblk = blk / nspan;
    const long row = (rowblk * 4 + wid) * RW + lane / LPR;
    if (row >= rows_total) return;
    unsigned char* p = out + row * rowb + (long)span * WSPAN + (lane % LPR) * 16;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
#pragma unroll
    for (int s = 0; s < WSPAN / SEG; ++s) *(f32x4*)(p + s * SEG) = v;
}

template <int SEG, int WSPAN>
static void run(unsigned char* out, long rowb, int rows) {
    constexpr int RW = 1024 / SEG;
    const long blocks = (long)(rows / (4 * RW)) * (rowb / WSPAN);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) k<SEG, WSPAN><<<dim3((unsigned)blocks), 256>>>(out, rowb, rows);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<SEG, WSPAN><<<dim3((unsigned)blocks), 256>>>(out + (size_t)(i % 3) * rows * rowb, rowb, rows);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)rows * rowb;
    printf("segment %4d B per row per instruction, wave block %4d B wide: %7.1f us  %5.2f TB/s\n", SEG, WSPAN, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e12);
}

int main() {
    const long rowb = 1024;                 // 512 bf16 columns
    const int rows = 65536 * 2;             // 128 MiB per buffer, 3 buffers rotated
    unsigned char* out; (void)hipMalloc(&out, (size_t)3 * rows * rowb);
    run<1024, 1024>(out, rowb, rows);
    run<128, 128>(out, rowb, rows);
    run<128, 256>(out, rowb, rows);
    run<64, 128>(out, rowb, rows);
    run<64, 256>(out, rowb, rows);
    run<32, 128>(out, rowb, rows);
    return 0;
}
