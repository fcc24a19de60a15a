// Streaming micro-benchmark shaped like the MSE part of vae_loss: read two f32 arrays, write one bf16 array, sum of squares.
// Variants: vector width V (8- or 16-byte loads), U vectors in flight per thread, workgroups per CU, grid-stride vs block-contiguous.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_read tools/ubench_read.hip && tools/bin/ubench_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short bf(float f) { unsigned u = __float_as_uint(f); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

template <int V, int U, bool CONTIG, bool WRITE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const float* __restrict__ t, unsigned short* g, unsigned nvec, float* out) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const unsigned nthreads = gridDim.x * 256u;
    float acc = 0.f;
    unsigned i, step, end;
    if (CONTIG) {                                   // each workgroup walks its own contiguous range, 256*U vectors per iteration
        const unsigned per = (nvec + gridDim.x - 1) / gridDim.x;
        i = blockIdx.x * per + threadIdx.x; end = min(nvec, (blockIdx.x + 1) * per); step = 256u;
    } else { i = blockIdx.x * 256u + threadIdx.x; end = nvec; step = nthreads; }
    for (; i < end; i += U * step) {
        vec a[U], b[U]; unsigned idx[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ok[u] = i + u * step < end; idx[u] = ok[u] ? i + u * step : 0u; a[u] = ((const vec*)x)[idx[u]]; b[u] = ((const vec*)t)[idx[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            unsigned short o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { const float d = a[u][e] - b[u][e]; acc += d * d; o[e] = bf(2.f * d); }
            if (WRITE) {
                if (V == 4) *(uint2*)(g + (size_t)idx[u] * 4) = make_uint2(o[0] | (unsigned)o[1] << 16, o[2] | (unsigned)o[3] << 16);
                else *(unsigned*)(g + (size_t)idx[u] * 2) = o[0] | (unsigned)o[1] << 16;
            }
        }
    }
    if (acc == 12345.678f) *out = acc;
}

template <int V, int U, bool CONTIG, bool WRITE>
void run(const float* x0, const float* t0, unsigned short* g0, size_t n, float* out, int wg_per_cu) {
    // 4 buffer sets in rotation: 2 GB per round, nothing comes back from the 256 MB Infinity Cache
#define ROT const float* x = x0 + (size_t)(w & 3) * n; const float* t = t0 + (size_t)(w & 3) * n; unsigned short* g = g0 + (size_t)(w & 3) * n;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const unsigned nvec = (unsigned)(n / V);
    const int grid = 256 * wg_per_cu;
    for (int w = 0; w < 2; ++w) { ROT hipLaunchKernelGGL((k<V, U, CONTIG, WRITE>), dim3(grid), dim3(256), 0, 0, x, t, g, nvec, out); }
    hipEventRecord(s);
    const int it = 12;
    for (int w = 0; w < it; ++w) { ROT hipLaunchKernelGGL((k<V, U, CONTIG, WRITE>), dim3(grid), dim3(256), 0, 0, x, t, g, nvec, out); }
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= it;
    const double bytes = (double)n * (8 + (WRITE ? 2 : 0));
    printf("V=%d U=%d %s %s wg/CU=%d : %7.1f us  %5.2f TB/s\n", V, U, CONTIG ? "contig" : "stride", WRITE ? "rw" : "ro", wg_per_cu, ms * 1e3, bytes / ms / 1e9);
}

int main() {
    const size_t n = (size_t)65536 * 782;           // recon_a / a of the workload
    float *x, *t, *out; unsigned short* g;
    hipMalloc(&x, 4 * n * 4); hipMalloc(&t, 4 * n * 4); hipMalloc(&g, 4 * n * 2); hipMalloc(&out, 4);
    hipMemset(x, 0x3c, 4 * n * 4); hipMemset(t, 0x3b, 4 * n * 4);
    for (int wg : {4, 8}) {
        run<2, 4, false, true>(x, t, g, n, out, wg);
        run<4, 4, false, true>(x, t, g, n, out, wg);
        run<4, 2, false, true>(x, t, g, n, out, wg);
        run<4, 8, false, true>(x, t, g, n, out, wg);
        run<4, 4, true, true>(x, t, g, n, out, wg);
        run<2, 4, true, true>(x, t, g, n, out, wg);
        run<4, 4, false, false>(x, t, g, n, out, wg);
        run<2, 4, false, false>(x, t, g, n, out, wg);
    }
    run<4, 4, false, true>(x, t, g, n, out, 16);
    run<4, 4, true, true>(x, t, g, n, out, 16);
    return 0;
}
