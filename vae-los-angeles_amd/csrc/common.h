// Shared device helpers for the MI355X (gfx950 / CDNA4) MultiModalVAE kernels.
// Wavefront = 64 lanes; MFMA tiles are 16x16 (bf16: K=32, f32: K=4) with f32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WAVE = 64;
constexpr int TILE = 128;          // output tile edge of both GEMM kernels
constexpr int NTHREADS = 256;      // 4 waves, 2x2, 64x64 each
constexpr int ROW_BYTES = 128;     // bytes of one LDS row in the NT kernel (= one K step)

// ---- MFMA wrappers: one "fragment step" consumes 16 bytes per lane of A and of B ------------
template <typename CT> struct Mma;

template <> struct Mma<bf16> {
    static constexpr int EPC = 8;            // elements per 16-byte chunk
    static constexpr int KSTEP = 32;         // reduction elements per fragment step
    typedef bf16x8 frag;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};

template <> struct Mma<float> {
    static constexpr int EPC = 4;
    static constexpr int KSTEP = 16;
    typedef f32x4 frag;
    // lane group g = lane>>4 holds reduction indices 4g..4g+3 of the step; MFMA sub-step s
    // multiplies element s of every group (any bijection of k is valid as long as A and B agree).
    static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }

// 16-byte register image of EPC compute-type elements
template <typename CT> struct Chunk;
template <> struct Chunk<bf16> {
    bf16x8 v;
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
    __device__ __forceinline__ void zero() { v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Chunk<float> {
    f32x4 v;
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
    __device__ __forceinline__ void zero() { v = f32x4{0.f, 0.f, 0.f, 0.f}; }
};

// Vector load of `VEC` elements of type T into floats (VEC*sizeof(T) <= 16, naturally aligned).
template <typename T, int VEC> struct VLoad;
template <> struct VLoad<float, 1> { static __device__ __forceinline__ void ld(const float* p, float* o) { o[0] = p[0]; } };
template <> struct VLoad<float, 2> { static __device__ __forceinline__ void ld(const float* p, float* o) { f32x2 v = *(const f32x2*)p; o[0] = v[0]; o[1] = v[1]; } };
template <> struct VLoad<float, 4> { static __device__ __forceinline__ void ld(const float* p, float* o) { f32x4 v = *(const f32x4*)p; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; } };
template <> struct VLoad<bf16, 1> { static __device__ __forceinline__ void ld(const bf16* p, float* o) { o[0] = (float)p[0]; } };
template <> struct VLoad<bf16, 2> { static __device__ __forceinline__ void ld(const bf16* p, float* o) { typedef __attribute__((ext_vector_type(2))) __bf16 v2; v2 v = *(const v2*)p; o[0] = (float)v[0]; o[1] = (float)v[1]; } };
template <> struct VLoad<bf16, 4> { static __device__ __forceinline__ void ld(const bf16* p, float* o) { bf16x4 v = *(const bf16x4*)p; for (int i = 0; i < 4; ++i) o[i] = (float)v[i]; } };
template <> struct VLoad<bf16, 8> { static __device__ __forceinline__ void ld(const bf16* p, float* o) { bf16x8 v = *(const bf16x8*)p; for (int i = 0; i < 8; ++i) o[i] = (float)v[i]; } };

// One LDS-DMA wave-instruction (64 lanes x 16 bytes -> 1 KB of LDS at `lds_addr`, lane-linear) as inline assembly.  Through
// __builtin_amdgcn_global_load_lds hipcc's wait-count pass treats later LDS reads as possibly reading the DMA's destination and, in
// loops that wait with counted s_waitcnt vmcnt(N) of their own, drains every transfer in flight with an s_waitcnt vmcnt(0) right
// behind the issue or in front of the next one -- a ring deeper than two slots then buys nothing (seen in the ISA of
// gemm_tn_wide.hip and gemm_nt3.h).  Issued from assembly the transfers are invisible to that pass; the kernel's own counted waits
// order them (vmcnt counts them in issue order like any other vector-memory operation).  m0 is written: kernels that use this
// helper must not use the builtin form as well; the compiler's own value of m0 (it reserves the register and knows nothing of this
// write: a clobber entry for it is only warned about) is saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(g) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// Natural logarithm as ONE v_log_f32 (log2, about 1 ulp) and one multiply.  hipcc expands __logf / logf into 13 vector instructions
// (denormal pre-scaling, a compensated multiply by ln 2, an inf / nan select): in the BCE term that was 22 of the 58 instructions per
// element (round 3, read off the ISA).  Differs from logf only in the last bits and for denormal arguments (below 1.18e-38: -inf here,
// i.e. torch's -100 clamp, where torch still has -87.3 .. -103): no sigmoid output of a finite logit above -87 is that small.
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Philox4x32-10 counter-based generator (Salmon et al. 2011); used for dropout masks and eps.
struct Philox {
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        // ONE 32 x 32 -> 64-bit multiply-add (v_mad_u64_u32) per product: written as __umulhi + a 32-bit product hipcc emits
        // v_mul_hi_u32 + v_mul_lo_u32 (the noise launch is bound by its multiplies; step -2.6 us)
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0], p1 = (uint64_t)M1 * (uint64_t)c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    }
    static __device__ __forceinline__ void gen(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) { round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

}  // namespace mm

#define MM_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
