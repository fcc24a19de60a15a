// Epilogues of the NT GEMM kernels (register-staged kernel in gemm_nt.hip, LDS-ring kernel in
// gemm_ring.hip): what happens to the f32 accumulators of a 128x128 tile owned by 4 waves (2x2,
// 64x64 each, 4x4 MFMA 16x16 tiles per wave).
#pragma once
#include "common.h"

namespace mm {

// ------------------------------------------------------------------------------------------
// Epilogues.  compute() maps one f32 accumulator (plus the staged operand element / mask byte)
// to the value that is stored; s1/s2 are per-column partial sums, reduced over the tile's rows
// and added (f64 atomics) to stat1/stat2 when STATS is set.
// ------------------------------------------------------------------------------------------
template <typename OT, bool STATS_>
struct EpiStore {               // C = act(acc + bias) (+ C) ; stats = (sum C, sum C^2) of the ROUNDED output
    static constexpr bool STATS = STATS_;
    static constexpr int NEED = 0;            // operand tiles to stage: 0 none, 1 = H, 2 = H + mask
    typedef OT out_t; typedef OT h_t;
    OT* C; long ldc; const float* bias; int act; int accumulate;
    const OT* H; long ldh; const uint8_t* mask; long ldm;     // unused
    double* stat1; double* stat2;
    struct Col { float b; };
    __device__ __forceinline__ bool stores() const { return true; }
    __device__ __forceinline__ bool accum() const { return accumulate != 0; }
    bool accumulate_requested() const { return accumulate != 0; }
    __device__ __forceinline__ Col col(int c, int N) const { return Col{(bias && c < N) ? bias[c] : 0.f}; }
    __device__ __forceinline__ float compute(float v, float, unsigned, const Col& cc, bool count, float& s1, float& s2) const {
        v += cc.b;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = 1.f / (1.f + expf(-v));
        const float r = to_f32(from_f32<OT>(v));
        if (STATS && count) { s1 += r; s2 += r * r; }
        return r;
    }
};

template <typename OT, typename HT>
struct EpiReluMask {            // dH = (H > 0) ? acc : 0
    static constexpr bool STATS = false;
    static constexpr int NEED = 1;
    typedef OT out_t; typedef HT h_t;
    OT* C; long ldc; const HT* H; long ldh; const uint8_t* mask; long ldm;
    double* stat1; double* stat2;
    struct Col {};
    __device__ __forceinline__ bool stores() const { return true; }
    __device__ __forceinline__ bool accum() const { return false; }
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ Col col(int, int) const { return Col{}; }
    __device__ __forceinline__ float compute(float v, float h, unsigned, const Col&, bool, float&, float&) const {
        return h > 0.f ? v : 0.f;
    }
};

template <typename OT, typename YT>
struct EpiBnBwd {               // BatchNorm+ReLU+Dropout backward around the dX contraction, two phases:
    // d = acc * keep * (y*scale+shift > 0), xhat = (y-mean)*rstd
    //   phase 0: nothing stored; stats = (sum d, sum d*xhat)  -> mmvae_bn_bwd_finalize -> coef
    //   phase 1: C = coef0 * (d - coef1 - xhat*coef2)          (d recomputed from the f32 accumulators, so
    //            the cancellation happens before the single rounding to the activation type)
    //   phase 2: C = d and the same statistics as phase 0 (one contraction; mmvae_bn_bwd_apply finishes in place)
    static constexpr bool STATS = true;
    static constexpr int NEED = 2;
    typedef OT out_t; typedef YT h_t;
    OT* C; long ldc; const YT* H; long ldh; const uint8_t* mask; long ldm;
    const float* scale; const float* shift; const float* mean; const float* rstd; float inv_keep;
    const float* coef; int phase;
    double* stat1; double* stat2;
    struct Col { float sc, sh, mu, rs, c0, c1, c2; };
    __device__ __forceinline__ bool stores() const { return phase != 0; }
    __device__ __forceinline__ bool accum() const { return false; }
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ Col col(int c, int N) const {
        Col k{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c < N) {
            k.sc = scale[c]; k.sh = shift[c]; k.mu = mean[c]; k.rs = rstd[c];
            if (phase == 1) { k.c0 = coef[c]; k.c1 = coef[N + c]; k.c2 = coef[2 * N + c]; }
        }
        return k;
    }
    __device__ __forceinline__ float compute(float v, float y, unsigned mb, const Col& cc, bool count, float& s1, float& s2) const {
        const float keep = mask ? (mb ? inv_keep : 0.f) : 1.f;
        const float d = (y * cc.sc + cc.sh > 0.f) ? v * keep : 0.f;
        const float xh = (y - cc.mu) * cc.rs;
        if (phase != 1) { if (count) { s1 += d; s2 += d * xh; } return d; }      // 0: statistics only; 2: statistics + store d
        return cc.c0 * (d - cc.c1 - xh * cc.c2);
    }
};

// Epilogue operand tiles (ReLU input / pre-BN output, keep mask) fetched into REGISTERS; issued before the main loop
// so that their HBM latency is hidden under the contraction instead of being exposed between main loop and stores.
struct EpiPrefetch { f32x4 h[8]; f32x4 m[4]; };

// WN = waves along N: the tile is 128 x (64*WN), owned by 2*WN waves (128*WN threads).  The per-thread chunk counts of the
// cooperative tile loads/stores below are independent of WN (8 x 16 B of a 2-byte tile, 4 x 16 B of the mask).
template <typename CT, typename Epi, int WN = 2>
__device__ __forceinline__ void nt_epilogue_prefetch(EpiPrefetch& pf, const Epi& epi, int row0, int col0, int M, int N, int tid) {
    typedef typename Epi::h_t HT;
    constexpr int BN = 64 * WN, NTH = 128 * WN;
    if constexpr (sizeof(CT) == 2 && Epi::NEED >= 1 && sizeof(HT) == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + NTH * i, r = c / (BN / 8), ch = c % (BN / 8);
            const int gr = row0 + r, gc = col0 + ch * 8;
            pf.h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gr < M && gc < N) pf.h[i] = *(const f32x4*)(epi.H + (long)gr * epi.ldh + gc);     // rows padded to 8 elements
        }
        if (Epi::NEED >= 2 && epi.mask != nullptr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + NTH * i, r = c / (BN / 16), ch = c % (BN / 16);
                const int gr = row0 + r, gc = col0 + ch * 16;
                pf.m[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (gr < M && gc < N) pf.m[i] = *(const f32x4*)(epi.mask + (long)gr * epi.ldm + gc);   // N % 16 == 0 checked on the host
            }
        }
    }
}

// smem: >= 24 KiB * WN scratch (free to overwrite), red: 1 KiB * WN.  wr/wc: wave row/column inside the tile.
template <typename CT, typename Epi, int WN = 2>
__device__ __forceinline__ void nt_epilogue(unsigned char* smem, float* red, f32x4 (&acc)[4][4], const Epi& epi, const EpiPrefetch& pf,
                                            int row0, int col0, int M, int N, int tid, int lane, int wr, int wc)
{
    constexpr int BN = 64 * WN, NTH = 128 * WN, RB = BN * 2;      // RB: row bytes of a staged 2-byte tile
    typedef typename Epi::out_t OT;
    typedef typename Epi::h_t HT;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lg = lane >> 4;
    // Staged form: 2-byte activation tiles go through LDS so that every global access of the epilogue is a
    // full-line 16-byte access.  The f32 precision mode (parity tool) and accumulate-into-C use the direct form.
    constexpr bool CAN_STAGE = sizeof(CT) == 2 && (Epi::NEED == 0 || sizeof(HT) == 2);
    if (CAN_STAGE && !epi.accum()) {
        unsigned char* sT = smem;                               // [128][128] of a 2-byte type, or [64][128] f32
        unsigned char* sM = smem + TILE * RB;                    // [128][BN] mask bytes
        if (Epi::NEED >= 1) {                                   // operand tiles: registers (prefetched) -> LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = tid + NTH * i, r = c / (BN / 8), ch = c % (BN / 8);
                *(f32x4*)(sT + r * RB + ch * 16) = pf.h[i];
            }
            if (Epi::NEED >= 2 && epi.mask != nullptr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = tid + NTH * i, r = c / (BN / 16), ch = c % (BN / 16);
                    *(f32x4*)(sM + r * BN + ch * 16) = pf.m[i];
                }
            }
            __syncthreads();
        }
        constexpr int HALVES = sizeof(OT) == 2 ? 1 : 2;         // f32 outputs are staged 64 rows at a time
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf) {
            if (HALVES == 1 || wr == hf) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int cl = wc * 64 + n * 16 + li, c = col0 + cl;
                    typename Epi::Col cc = epi.col(c, N);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int rl = wr * 64 + m * 16 + lg * 4 + j;
                            const bool ok = (row0 + rl < M) && (c < N);
                            float hv = 0.f; unsigned mb = 1;
                            if (Epi::NEED >= 1) hv = to_f32(*(const HT*)(sT + rl * RB + cl * 2));
                            if (Epi::NEED >= 2) mb = sM[rl * BN + cl];
                            float o = epi.compute(acc[m][n][j], hv, mb, cc, ok, s1[n], s2[n]);
                            if (c >= N) o = 0.f;
                            const int rs = HALVES == 1 ? rl : rl - hf * 64;
                            *(OT*)(sT + (rs * BN + cl) * (int)sizeof(OT)) = from_f32<OT>(o);
                        }
                }
            }
            if (epi.stores()) {
                __syncthreads();
                constexpr int CPR = BN * (int)sizeof(OT) / 16, EPCO = 16 / (int)sizeof(OT);
                const bool v16 = ((epi.ldc * sizeof(OT)) % 16 == 0) && (((uintptr_t)epi.C & 15) == 0);
                const bool v8 = ((epi.ldc * sizeof(OT)) % 8 == 0) && (((uintptr_t)epi.C & 7) == 0);
                const int n_store = (int)min((long)((N + EPCO - 1) / EPCO * EPCO), epi.ldc);   // pad columns of internal buffers get zeros
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = tid + NTH * i, r = c / CPR, ch = c % CPR;
                    const int gr = row0 + hf * (TILE / HALVES) + r, gc = col0 + ch * EPCO;
                    if (gr >= M || gc >= n_store) continue;
                    const unsigned char* sp = sT + r * (BN * (int)sizeof(OT)) + ch * 16;
                    OT* gp = epi.C + (long)gr * epi.ldc + gc;
                    if (gc + EPCO <= n_store && v16) *(f32x4*)gp = *(const f32x4*)sp;
                    else if (gc + EPCO <= n_store && v8) { ((f32x2*)gp)[0] = ((const f32x2*)sp)[0]; ((f32x2*)gp)[1] = ((const f32x2*)sp)[1]; }
                    else {
#pragma unroll
                        for (int e = 0; e < EPCO; ++e) if (gc + e < n_store) gp[e] = ((const OT*)sp)[e];
                    }
                }
                if (HALVES == 2 && hf == 0) __syncthreads();
            }
        }
    } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int c = col0 + wc * 64 + n * 16 + li;
            if (c < N) {
                typename Epi::Col cc = epi.col(c, N);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = row0 + wr * 64 + m * 16 + lg * 4 + j;
                        if (r < M) {
                            float hv = 0.f; unsigned mb = 1;
                            if (Epi::NEED >= 1) hv = to_f32(epi.H[(long)r * epi.ldh + c]);
                            if (Epi::NEED >= 2 && epi.mask) mb = epi.mask[(long)r * epi.ldm + c];
                            float o = epi.compute(acc[m][n][j], hv, mb, cc, true, s1[n], s2[n]);
                            if (epi.stores()) {
                                OT* q = epi.C + (long)r * epi.ldc + c;
                                if (epi.accum()) o += to_f32(*q);
                                *q = from_f32<OT>(o);
                            }
                        }
                    }
            }
        }
    }
    if (Epi::STATS && (epi.stat1 != nullptr || epi.stat2 != nullptr)) {
        __syncthreads();                                        // red[] is separate from the staging area, but order the reuse
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s1[n] += __shfl_xor(s1[n], 16, 64); s1[n] += __shfl_xor(s1[n], 32, 64);
            s2[n] += __shfl_xor(s2[n], 16, 64); s2[n] += __shfl_xor(s2[n], 32, 64);
            if (lane < 16) {
                red[(wr * 2 + 0) * BN + wc * 64 + n * 16 + lane] = s1[n];
                red[(wr * 2 + 1) * BN + wc * 64 + n * 16 + lane] = s2[n];
            }
        }
        __syncthreads();
        if (tid < BN && col0 + tid < N) {
            if (epi.stat1) unsafeAtomicAdd(epi.stat1 + col0 + tid, (double)(red[0 * BN + tid] + red[2 * BN + tid]));
            if (epi.stat2) unsafeAtomicAdd(epi.stat2 + col0 + tid, (double)(red[1 * BN + tid] + red[3 * BN + tid]));
        }
    }
}

}  // namespace mm
