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

// Every fetch below is BRANCH-FREE: the address is clamped into the tensor and the load always issues; rows past M and
// columns past K are zeroed in finish(), after the wait the compiler places at the first use.  An `if (ok) load else zero`
// makes hipcc wrap each load in an exec-masked block whose else-side writes the same VGPRs, which forces an
// s_waitcnt vmcnt(0) after every load and serialises the whole K-step prefetch (seen in the gfx950 ISA).
template <int EPC, int VEC>
__device__ __forceinline__ void load_chunk_f32(RawVec<float, EPC>& r, const float* p, unsigned row_off, int k, int K) {
#pragma unroll
    for (int j = 0; j < EPC; j += VEC) VLoad<float, VEC>::ld(p + (row_off + (unsigned)min(k + j, K - VEC)), &r.v[j]);     // K % VEC == 0, K >= VEC
}

// Plain A[M][K] (element type AT, row stride lda, vectors of VEC elements; K % VEC == 0).
template <typename CT, typename AT, int VEC>
struct SrcPlain {
    static constexpr int EPC = Mma<CT>::EPC;
    static constexpr bool NEEDS_AUX = false;
    static constexpr bool PAD_TAIL = sizeof(AT) == 2;      // 2-byte rows are read in whole 16-byte chunks: a row may end inside one
    const AT* p; long lda; int M, K;
    typedef RawVec<AT, EPC> Raw;
    __device__ __forceinline__ void init(float*, int, int) const {}
    __device__ __forceinline__ void fetch(Raw& r, int row, int k) const {
        // 32-bit element offsets (the entry points reject operands of 4 GiB or more): the loads take the
        // scalar-base + 32-bit-VGPR-offset form, one address VGPR per load instead of a 64-bit pair
        const unsigned ro = (unsigned)min(row, M - 1) * (unsigned)lda;
        if constexpr (sizeof(AT) == 2) {
            r.v = *(const bf16x8*)(p + (ro + (unsigned)min(k, ((K + 7) & ~7) - 8)));      // internal buffers: rows padded to 8 elements
        } else {
            load_chunk_f32<EPC, VEC>(r, p, ro, k, K);
        }
    }
    // finish_fast: NO masking.  What lies beyond the operand (rows clamped to M-1, columns clamped into the row, zero pads of
    // the activation buffers) is finite, and it only ever meets zero weight padding (NT), a zeroed P row (TN) or an output
    // element that is never stored -- the masking selects and the tail branch were ~100 cycles per chunk of the registers ->
    // LDS stage, 8 chunks per K step.  finish_rows: rows >= M become zeros (the P operand of the dW GEMM: batch rows past the
    // split must not enter the reduction).  finish: rows and columns masked (kept for callers that need exact zeros).
    __device__ __forceinline__ void finish_fast(const Raw& r, int, Chunk<CT>& o, const float*) const {
        if constexpr (sizeof(AT) == 2) o.v = r.v;
        else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) o.set(i, r.v[i]);
        }
    }
    __device__ __forceinline__ void finish_rows(const Raw& r, int row, int, Chunk<CT>& o, const float*) const {
        const bool ok = row < M;
        if constexpr (sizeof(AT) == 2) {
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            o.v = ok ? r.v : z;
        } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) o.set(i, ok ? r.v[i] : 0.f);
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, int row, int k, Chunk<CT>& o, const float*) const {
        const bool ok = row < M;
        if constexpr (sizeof(AT) == 2) {
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            o.v = (ok && k < K) ? r.v : z;
            if (k + 8 > K) {                             // never let pad garbage meet the zero weights
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k + i >= K) o.v[i] = (bf16)0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPC; ++i) o.set(i, (ok && k + i < K) ? r.v[i] : 0.f);
        }
    }
};

// A = dropout(relu(y * scale + shift)) where y is the previous layer's pre-BatchNorm output,
// scale = gamma * rstd, shift = beta - mean * scale (per column, staged in LDS), and the keep
// mask is a uint8 [M][ldm] tensor (1 = kept) or NULL (eval / p = 0).
// mmvae_bn_finalize folded into the consumer of its result (round 3): the BatchNorm constants of the operand's columns straight from
// the f64 column sums of the producing GEMM.  sum == nullptr: off.  The arithmetic is bn_finalize_kernel's (elementwise.hip), so the
// tables are bit-identical to the two-launch form; workgroup 0 also writes that kernel's outputs (backward reads them, later launches).
struct BnFin {
    const double* sum = nullptr; const double* sumsq = nullptr; const float* gamma = nullptr; const float* beta = nullptr;
    float eps = 0.f, momentum = 0.f;
    float* running_mean = nullptr; float* running_var = nullptr; long long* nbt = nullptr;
    float* mean = nullptr; float* rstd = nullptr; float* scale = nullptr; float* shift = nullptr; int M = 0;
    // scale / shift of column i (as mmvae_bn_finalize rounds them); first_wg: this workgroup stores the outputs
    __device__ __forceinline__ void column(int i, bool first_wg, float& sc, float& sh) const {
        const double mean_ = sum[i] / M;
        double var = sumsq[i] / M - mean_ * mean_;
        if (var < 0.0) var = 0.0;
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        sc = gamma[i] * rs; sh = beta[i] - (float)mean_ * sc;
        if (first_wg) {
            mean[i] = (float)mean_; rstd[i] = rs; scale[i] = sc; shift[i] = sh;
            if (running_mean) {
                running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)mean_;
                const double unbiased = var * ((double)M / (double)(M - 1));
                running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unbiased;
            }
            if (i == 0 && nbt) *nbt += 1;
        }
    }
};

template <typename CT>
struct SrcBnReluDrop {
    static constexpr int EPC = Mma<CT>::EPC;
    static constexpr bool NEEDS_AUX = true;
    static constexpr bool PAD_TAIL = false;                // K = a hidden width (multiple of 64)
    const CT* y; long ldy; int M, K;
    const float* scale; const float* shift;
    const uint8_t* mask; long ldm; float inv_keep;
    BnFin fin = BnFin{};
    struct Raw { RawVec<CT, EPC> y; uint32_t m[EPC / 4]; };
    __device__ __forceinline__ void init(float* aux, int tid, int) const {
        // relu(y sc + sh) * inv_keep == relu(y (sc inv_keep) + sh inv_keep) for inv_keep > 0: folded here once per workgroup.
        // The keep bytes (0 or 1, mmvae_noise) then multiply as floats (v_cvt_f32_ubyteN): 4 VALU per element instead of 7 in
        // the registers->LDS stage, which is VALU-sensitive (see the BatchNorm-backward source below).
        if (fin.sum) {
            const bool first_wg = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
            for (int i = tid; i < K; i += NTHREADS) { float sc, sh; fin.column(i, first_wg, sc, sh); aux[i] = sc * inv_keep; aux[512 + i] = sh * inv_keep; }
            return;
        }
        for (int i = tid; i < K; i += NTHREADS) { aux[i] = scale[i] * inv_keep; aux[512 + i] = shift[i] * inv_keep; }
    }
    __device__ __forceinline__ void fetch(Raw& r, int row, int k) const {        // K % EPC == 0 (hidden widths)
        const int rc = min(row, M - 1), kc = min(k, K - EPC);
        const CT* q = y + ((unsigned)rc * (unsigned)ldy + (unsigned)kc);
        if constexpr (sizeof(CT) == 2) {
            r.y.v = *(const bf16x8*)q;
        } else {
            f32x4 t = *(const f32x4*)q; r.y.v[0] = t[0]; r.y.v[1] = t[1]; r.y.v[2] = t[2]; r.y.v[3] = t[3];
        }
        if (mask != nullptr) {                           // kernel-argument uniform: a scalar branch
            const uint32_t* mp = (const uint32_t*)(mask + ((unsigned)rc * (unsigned)ldm + (unsigned)kc));
#pragma unroll
            for (int i = 0; i < EPC / 4; ++i) r.m[i] = mp[i];
        } else {
#pragma unroll
            for (int i = 0; i < EPC / 4; ++i) r.m[i] = 0x01010101u;
        }
    }
    __device__ __forceinline__ void finish_fast(const Raw& r, int k, Chunk<CT>& o, const float* aux) const {      // k < K (hidden widths)
        const int kk = min(k, 512 - EPC);
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            const float v = fmaxf(r.y.get(i) * aux[kk + i] + aux[512 + kk + i], 0.f);
            o.set(i, v * (float)((r.m[i >> 2] >> (8 * (i & 3))) & 0xffu));
        }
    }
    __device__ __forceinline__ void finish_rows(const Raw& r, int row, int k, Chunk<CT>& o, const float* aux) const { finish(r, row, k, o, aux); }
    __device__ __forceinline__ void finish(const Raw& r, int row, int k, Chunk<CT>& o, const float* aux) const {
        bool ok = row < M && k < K;
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            float sc = ok ? aux[k + i] : 0.f, sh = ok ? aux[512 + k + i] : 0.f;
            float v = fmaxf(r.y.get(i) * sc + sh, 0.f);
            o.set(i, v * (float)((r.m[i >> 2] >> (8 * (i & 3))) & 0xffu));
        }
    }
};

// P = c0 * (d - c1 - xhat * c2), xhat = (y - mean) * rstd: the BatchNorm-backward correction (mmvae_bn_bwd_apply) applied on
// the operand load of the dW GEMM.  For a FIRST layer nothing else consumes dL/dy, so the element-wise pass over d (read d, y,
// write d: 3 x 67 MB on the critical path of the step) disappears; d stays the raw epilogue output of the dX contraction.
// Per-column constants of the workgroup's 128-column tile sit in LDS (aux: 5 x 128 floats, index = column & 127).
template <typename CT>
struct SrcBnBwdApply {
    static constexpr int EPC = Mma<CT>::EPC;
    static constexpr bool NEEDS_AUX = true;
    static constexpr bool PAD_TAIL = false;
    const CT* d; long ldd; const CT* y; long ldy; int M, K;          // K = number of columns (the layer width)
    const float* mean; const float* rstd; const float* coef;          // coef: [3][K]
    struct Raw { RawVec<CT, EPC> d, y; };
    __device__ __forceinline__ void init(float* aux, int tid, int col0) const {
        if (tid < TILE) {
            const int c = col0 + tid;
            const bool ok = c < K;
            // P = c0 (d - c1 - (y - mean) rstd c2) = c0 d - ((y - mean) (c0 c2 rstd) + c0 c1): per element one subtraction and two
            // fused multiply-adds on four per-column constants (the difference y - mean is still formed first: no cancellation)
            const float c0 = ok ? coef[c] : 0.f;
            aux[tid] = ok ? mean[c] : 0.f; aux[TILE + tid] = c0;
            aux[2 * TILE + tid] = ok ? c0 * coef[2 * K + c] * rstd[c] : 0.f; aux[3 * TILE + tid] = ok ? c0 * coef[K + c] : 0.f;
        }
    }
    __device__ __forceinline__ void fetch(Raw& r, int row, int k) const {          // K % EPC == 0 (hidden widths), rows padded
        const unsigned rc = (unsigned)min(row, M - 1), kc = (unsigned)min(k, K - EPC);
        if constexpr (sizeof(CT) == 2) {
            r.d.v = *(const bf16x8*)(d + (rc * (unsigned)ldd + kc));
            r.y.v = *(const bf16x8*)(y + (rc * (unsigned)ldy + kc));
        } else {
            const f32x4 a = *(const f32x4*)(d + (rc * (unsigned)ldd + kc)), b = *(const f32x4*)(y + (rc * (unsigned)ldy + kc));
#pragma unroll
            for (int i = 0; i < 4; ++i) { r.d.v[i] = a[i]; r.y.v[i] = b[i]; }
        }
    }
    __device__ __forceinline__ void finish_fast(const Raw& r, int k, Chunk<CT>& o, const float* aux) const { finish(r, 0, k, o, aux); }
    __device__ __forceinline__ void finish_rows(const Raw& r, int row, int k, Chunk<CT>& o, const float* aux) const { finish(r, row, k, o, aux); }
    __device__ __forceinline__ void finish(const Raw& r, int row, int k, Chunk<CT>& o, const float* aux) const {
        const bool ok = row < M && k < K;
        const int c = k & (TILE - 1);
#pragma unroll
        for (int i = 0; i < EPC; ++i) {
            const float t = fmaf(r.y.get(i) - aux[c + i], aux[2 * TILE + c + i], aux[3 * TILE + c + i]);
            const float v = fmaf(aux[TILE + c + i], r.d.get(i), -t);
            o.set(i, ok ? v : 0.f);
        }
    }
};

}  // namespace mm
