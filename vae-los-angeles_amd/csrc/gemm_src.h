// Operand sources shared by the NT and TN GEMM kernels: how a 16-byte LDS chunk of the
// compute type is produced from global memory (plain load + convert, or the fused
// BatchNorm-normalise + ReLU + Dropout of the previous encoder layer).
#pragma once
#include "common.h"

namespace mm {

// ------------------------------------------------------------------------------------------
// A-operand sources
// ------------------------------------------------------------------------------------------
template <typename AT, int EPC> struct RawVec;
template <> struct RawVec<bf16, 8> {
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
};
template <int EPC> struct RawVec<float, EPC> {
    float v[EPC];
    __device__ __forceinline__ float get(int i) const { return v[i]; }
};

template <int EPC, int VEC>
__device__ __forceinline__ void load_chunk_f32(RawVec<float, EPC>& r, const float* p, int k, int K, bool row_ok) {
#pragma unroll
    for (int j = 0; j < EPC; j += VEC) {
        if (row_ok && k + j < K) {
            VLoad<float, VEC>::ld(p + j, &r.v[j]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) r.v[j + e] = 0.f;
        }
    }
}

// Plain A[M][K] (element type AT, row stride lda, vectors of VEC elements; K % VEC == 0).
template <typename CT, typename AT, int VEC>
struct SrcPlain {
    static constexpr int EPC = Mma<CT>::EPC;
    static constexpr bool NEEDS_AUX = false;
    const AT* p; long lda; int M, K;
    typedef RawVec<AT, EPC> Raw;
    __device__ __forceinline__ void init(float*, int) const {}
    __device__ __forceinline__ void fetch(Raw& r, int row, int k) const {
        bool ok = row < M;
        const AT* q = p + (long)row * lda + k;
        if constexpr (sizeof(AT) == 2) {
            if (ok && k < K) {
                r.v = *(const bf16x8*)q;                 // internal buffers: rows padded to 8 elements
                if (k + 8 > K) {                         // never let pad garbage meet the zero weights
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (k + i >= K) r.v[i] = (bf16)0.f;
                }
            } else r.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        } else {
            load_chunk_f32<EPC, VEC>(r, q, k, K, ok);
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, int, int, Chunk<CT>& o, const float*) const {
        if constexpr (sizeof(AT) == 2) {
            o.v = r.v;
        } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) o.set(i, r.v[i]);
        }
    }
};

// A = dropout(relu(y * scale + shift)) where y is the previous layer's pre-BatchNorm output,
// scale = gamma * rstd, shift = beta - mean * scale (per column, staged in LDS), and the keep
// mask is a uint8 [M][ldm] tensor (1 = kept) or NULL (eval / p = 0).
template <typename CT>
struct SrcBnReluDrop {
    static constexpr int EPC = Mma<CT>::EPC;
    static constexpr bool NEEDS_AUX = true;
    const CT* y; long ldy; int M, K;
    const float* scale; const float* shift;
    const uint8_t* mask; long ldm; float inv_keep;
    struct Raw { RawVec<CT, EPC> y; uint32_t m[EPC / 4]; };
    __device__ __forceinline__ void init(float* aux, int tid) const {
        for (int i = tid; i < K; i += NTHREADS) { aux[i] = scale[i]; aux[512 + i] = shift[i]; }
    }
    __device__ __forceinline__ void fetch(Raw& r, int row, int k) const {
        bool ok = row < M && k < K;
        const CT* q = y + (long)row * ldy + k;
        if constexpr (sizeof(CT) == 2) {
            if (ok) r.y.v = *(const bf16x8*)q; else r.y.v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        } else {
            if (ok) { f32x4 t = *(const f32x4*)q; r.y.v[0] = t[0]; r.y.v[1] = t[1]; r.y.v[2] = t[2]; r.y.v[3] = t[3]; }
            else { r.y.v[0] = r.y.v[1] = r.y.v[2] = r.y.v[3] = 0.f; }
        }
        if (mask != nullptr && ok) {
            const uint32_t* mp = (const uint32_t*)(mask + (long)row * ldm + k);
#pragma unroll
            for (int i = 0; i < EPC / 4; ++i) r.m[i] = mp[i];
        } else {
#pragma unroll
            for (int i = 0; i < EPC / 4; ++i) r.m[i] = mask != nullptr ? 0u : 0x01010101u;
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, int row, int k, Chunk<CT>& o, const float* aux) const {
        bool ok = row < M && k < K;
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            float sc = ok ? aux[k + i] : 0.f, sh = ok ? aux[512 + k + i] : 0.f;
            float v = fmaxf(r.y.get(i) * sc + sh, 0.f);
            float keep = ((r.m[i >> 2] >> (8 * (i & 3))) & 0xffu) ? inv_keep : 0.f;
            o.set(i, v * keep);
        }
    }
};

}  // namespace mm
