// NT GEMM for gfx950:  C[M,N] = epilogue( prologue(A)[M,K] * W[N,K]^T )
//
// Replaces, on the MultiModalVAE training path (reference file:line):
//   * every nn.Linear forward  (src/models/encoders.py:13,18-19,31,35,40-41,54-55;
//     src/models/decoders.py:13,15,27,29,31,44,46)  -> aten::addmm
//   * the dX half of every Linear backward (autograd mm, optimize_hyperparameters.py:112),
//     with W^T prepared as the "weight" operand
//   * fused around the contraction: BatchNorm1d-normalise + ReLU + Dropout of the PREVIOUS
//     layer as the A-operand prologue (encoders.py:14-16,32-34,36-38), bias / ReLU / Sigmoid
//     epilogues (decoders.py:14,28,30,32), BatchNorm batch statistics (sum, sum of squares per
//     column) and the ReLU / BN-ReLU-Dropout backward masks as epilogues.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 each = 4x4 MFMA
// 16x16 tiles), one K step = 128 bytes per LDS row (64 bf16 / 32 f32), XOR-swizzled 16-byte
// chunks so that ds_read_b128 fragment reads are bank-conflict free.  A and W tiles are
// register-staged (global -> VGPR -> [convert / prologue] -> LDS) with the next K step's
// global loads in flight under the current step's MFMAs.  blockIdx -> tile mapping keeps all
// column tiles of one row tile on one XCD (shared L2) in adjacent dispatch slots.
#include "common.h"
#include "mmvae_hip.h"
#include "gemm_src.h"
#include "gemm_nt_epi.h"
#include "gemm_ring.h"

namespace mm {

// ------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

// LDS: main loop 2 x { A [128][128 B] + W [64*WN][128 B] }; the epilogue reuses it as scratch (2-byte tile [128][64*WN] + mask bytes)
template <int WN> struct NtLds {
    static constexpr int BUF = (TILE + 64 * WN) * ROW_BYTES;      // one K step of A and W
    static constexpr int MAIN = 2 * BUF;                          // double buffered
    static constexpr int SCRATCH = TILE * 64 * WN * 3;            // 48 KiB (WN=2) / 96 KiB (WN=4)
    static constexpr int STAGE = MAIN > SCRATCH ? MAIN : SCRATCH;
    static constexpr int TOTAL = STAGE + 4096 + 4 * 64 * WN * 4;   // + BN prologue scale/shift + column-sum scratch
};

// WN = 2: 128x128 tile, 4 waves, 2 workgroups per CU.  WN = 4: 128x256 tile, 8 waves, 1 workgroup per CU -- the A tile is
// fetched once for 256 output columns, which halves the L2->CU operand ingest of the N = 256 / 512 layers.
template <typename CT, typename Src, typename Epi, int WN>
__global__ __launch_bounds__(128 * WN, 2)
void gemm_nt_kernel(Src src, const CT* __restrict__ W, long ldw, int M, int N, int K, int gx, int gy, Epi epi)
{
    constexpr int EPC = Mma<CT>::EPC;
    constexpr int BK = ROW_BYTES / (int)sizeof(CT);
    constexpr int BN = 64 * WN, NTH = 128 * WN, A_PER = 8 / WN;
    typedef typename Mma<CT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* aux = (float*)(smem + NtLds<WN>::STAGE);
    float* red = (float*)(smem + NtLds<WN>::STAGE + 4096);

    // XCD-aware tile assignment: linear id L runs on XCD L%8 (round-robin dispatch, speed only).
    const int L = blockIdx.x;
    const int slot = L >> 3;
    const int ct = slot % gy;
    const int rt = (slot / gy) * 8 + (L & 7);
    if (rt >= gx) return;
    const int row0 = rt * TILE, col0 = ct * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;

    if (Src::NEEDS_AUX) { src.init(aux, tid); __syncthreads(); }

    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // TWO register sets: while K step kt is multiplied, the loads of kt+1 AND kt+2 are in flight (one step of look-ahead
    // left every step waiting a full L2/HBM round trip: the loop was latency-bound, not byte-bound).
    typename Src::Raw ra0[A_PER], ra1[A_PER];
    Chunk<CT> rb0[4], rb1[4];
    const int nk = (K + BK - 1) / BK;

    auto fetch = [&](typename Src::Raw (&ra)[A_PER], Chunk<CT> (&rb)[4], int kt) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            src.fetch(ra[i], row0 + r, kt * BK + ch * EPC);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            rb[i].v = *(const decltype(rb[i].v)*)(W + (long)(col0 + r) * ldw + kt * BK + ch * EPC);
        }
    };
    auto stage = [&](typename Src::Raw (&ra)[A_PER], Chunk<CT> (&rb)[4], int kt, int buf) {      // registers -> LDS buffer
        unsigned char* sA = smem + buf * NtLds<WN>::BUF;
        unsigned char* sB = sA + TILE * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            Chunk<CT> o;
            src.finish(ra[i], row0 + r, kt * BK + ch * EPC, o, aux);
            *(decltype(o.v)*)(sA + swz(r, ch)) = o.v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            *(decltype(rb[i].v)*)(sB + swz(r, ch)) = rb[i].v;
        }
    };
    auto compute = [&](int buf, int s) {                      // fragment step s (0/1) of an LDS buffer
        const unsigned char* sA = smem + buf * NtLds<WN>::BUF;
        const unsigned char* sB = sA + TILE * ROW_BYTES;
        frag af[4], bf[4];
        const int ch = s * 4 + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *(const frag*)(sA + swz(wr * 64 + m * 16 + (lane & 15), ch));
#pragma unroll
        for (int n = 0; n < 4; ++n) bf[n] = *(const frag*)(sB + swz(wc * 64 + n * 16 + (lane & 15), ch));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<CT>::mma(acc[m][n], af[m], bf[n]);
    };

    // Double-buffered LDS, ONE barrier per K step: while buffer kt&1 is multiplied, tile kt+1 is written into the other
    // buffer between the two fragment steps and tiles kt+2 / kt+3 are in flight from HBM/L2 in the two register sets.
    //
    // hipcc's s_waitcnt insertion is only as precise as the control flow lets it be: a fetch under `if (kt + 3 < nk)`
    // means "maybe 8 fewer loads in flight" at the next stage(), and the wait degrades to vmcnt(0) -- every K step then
    // drains BOTH register sets and the look-ahead is gone (this is what made load phase + MFMA phase add up).  So the
    // steady-state loop below has NO conditional fetches and is entered only with both sets in flight; the last 3-4
    // K steps and the short-K layers (nk <= 4) run the conditional form.
    if (nk >= 5) {
        fetch(ra0, rb0, 0);
        fetch(ra1, rb1, 1);
        stage(ra0, rb0, 0, 0);
        fetch(ra0, rb0, 2);
        __syncthreads();
        int kt = 0;
        for (; kt + 4 < nk; kt += 2) {
            compute(0, 0);
            stage(ra1, rb1, kt + 1, 1);
            compute(0, 1);
            fetch(ra1, rb1, kt + 3);
            __syncthreads();
            compute(1, 0);
            stage(ra0, rb0, kt + 2, 0);
            compute(1, 1);
            fetch(ra0, rb0, kt + 4);
            __syncthreads();
        }
        const bool four = kt + 3 < nk;                        // 3 or 4 K steps left: kt in LDS, kt+1 / kt+2 in flight
        compute(0, 0);
        stage(ra1, rb1, kt + 1, 1);
        compute(0, 1);
        if (four) fetch(ra1, rb1, kt + 3);
        __syncthreads();
        compute(1, 0);
        stage(ra0, rb0, kt + 2, 0);
        compute(1, 1);
        __syncthreads();
        compute(0, 0);
        if (four) stage(ra1, rb1, kt + 3, 1);
        compute(0, 1);
        __syncthreads();
        if (four) {
            compute(1, 0);
            compute(1, 1);
            __syncthreads();
        }
    } else {
        fetch(ra0, rb0, 0);
        if (nk > 1) fetch(ra1, rb1, 1);
        stage(ra0, rb0, 0, 0);
        if (nk > 2) fetch(ra0, rb0, 2);
        __syncthreads();
        compute(0, 0);
        if (nk > 1) stage(ra1, rb1, 1, 1);
        compute(0, 1);
        if (nk > 3) fetch(ra1, rb1, 3);
        __syncthreads();
        if (nk > 1) {
            compute(1, 0);
            if (nk > 2) stage(ra0, rb0, 2, 0);
            compute(1, 1);
            __syncthreads();
        }
        if (nk > 2) {
            compute(0, 0);
            if (nk > 3) stage(ra1, rb1, 3, 1);
            compute(0, 1);
            __syncthreads();
        }
        if (nk > 3) {
            compute(1, 0);
            compute(1, 1);
            __syncthreads();
        }
    }

    EpiPrefetch pf;                      // epilogue operand tiles (fetched here: the two K-step register sets take the VGPR room)
    if (!epi.accum()) nt_epilogue_prefetch<CT, Epi, WN>(pf, epi, row0, col0, M, N, tid);
    nt_epilogue<CT, Epi, WN>(smem, red, acc, epi, pf, row0, col0, M, N, tid, lane, wr, wc);
}

struct RingSrc { const void* a; long lda; };     // tag: plain bf16 A served by the LDS-ring kernel (gemm_ring.h)

template <typename CT, typename Epi>
static int launch_nt(const RingSrc& src, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    if constexpr (sizeof(CT) == 2) return launch_nt_ring(src.a, src.lda, W, ldw, M, N, K, epi, st);
    return MMVAE_ERR_DTYPE;
}

template <typename CT, typename Src, typename Epi, int WN>
static int launch_nt_wn(const Src& src, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<CT, Src, Epi, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, NtLds<WN>::TOTAL);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int gx = (M + TILE - 1) / TILE, gy = (N + 64 * WN - 1) / (64 * WN);
    const int grid = ((gx + 7) / 8) * 8 * gy;
    hipLaunchKernelGGL((gemm_nt_kernel<CT, Src, Epi, WN>), dim3(grid), dim3(128 * WN), NtLds<WN>::TOTAL, st,
                       src, (const CT*)W, ldw, M, N, K, gx, gy, epi);
    MM_CHECK_LAUNCH();
    return 0;
}

static int g_wide_min_m = 256 * 128;      // 128x256 tiles only when there are >= 256 row tiles (mmvae_set_tuning key 0)

static inline bool nt_wide_ok(int M, int N) {
    static const bool off = getenv("MMVAE_NO_WIDE_TILES") != nullptr;      // A/B switch
    return !off && N % 256 == 0 && M >= g_wide_min_m;       // the prepared W has ceil128(N) rows: whole 256-column tiles only
}

template <typename CT, typename Src, typename Epi>
static int launch_nt(const Src& src, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    if constexpr (sizeof(CT) == 2) {
        if (nt_wide_ok(M, N) && !epi.accumulate_requested()) return launch_nt_wn<CT, Src, Epi, 4>(src, W, ldw, M, N, K, epi, st);
    }
    return launch_nt_wn<CT, Src, Epi, 2>(src, W, ldw, M, N, K, epi, st);
}

template <typename CT, typename Src>
static int dispatch_epi(const mmvae_gemm_nt_args* a, const Src& src, hipStream_t st) {
    const bool out_lp = a->c_dtype == MMVAE_BF16;
    if (a->prec == MMVAE_PREC_F32 && out_lp) return MMVAE_ERR_DTYPE;
    typedef CT LP;   // activation type of this precision mode
    switch (a->epilogue) {
    case MMVAE_EPI_STORE: {
        const bool stats = a->stat1 != nullptr || a->stat2 != nullptr;
        if (a->c_dtype == MMVAE_F32) {
            if (stats) { EpiStore<float, true> e{(float*)a->c, a->ldc, a->bias, a->act, a->accumulate, nullptr, 0, nullptr, 0, a->stat1, a->stat2};
                         return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st); }
            EpiStore<float, false> e{(float*)a->c, a->ldc, a->bias, a->act, a->accumulate, nullptr, 0, nullptr, 0, nullptr, nullptr};
            return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st);
        }
        if constexpr (sizeof(CT) == 2) {
            if (a->ldc % 8 || ((uintptr_t)a->c & 15)) return MMVAE_ERR_ARG;
            if (stats) { EpiStore<bf16, true> e{(bf16*)a->c, a->ldc, a->bias, a->act, a->accumulate, nullptr, 0, nullptr, 0, a->stat1, a->stat2};
                         return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st); }
            EpiStore<bf16, false> e{(bf16*)a->c, a->ldc, a->bias, a->act, a->accumulate, nullptr, 0, nullptr, 0, nullptr, nullptr};
            return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st);
        }
        return MMVAE_ERR_DTYPE;
    }
    case MMVAE_EPI_RELU_MASK: {     // C and H are activation-typed (bf16 in bf16 mode, f32 in f32 mode)
        if (a->h == nullptr) return MMVAE_ERR_ARG;
        if ((a->c_dtype == MMVAE_BF16) != (sizeof(LP) == 2)) return MMVAE_ERR_DTYPE;
        if (sizeof(LP) == 2 && (a->ldc % 8 || a->ldh % 8 || ((uintptr_t)a->c & 15) || ((uintptr_t)a->h & 15))) return MMVAE_ERR_ARG;
        EpiReluMask<LP, LP> e{(LP*)a->c, a->ldc, (const LP*)a->h, a->ldh, nullptr, 0, nullptr, nullptr};
        return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st);
    }
    case MMVAE_EPI_BN_BWD: {
        if (a->h == nullptr || !a->bn_scale || !a->bn_shift || !a->bn_mean || !a->bn_rstd) return MMVAE_ERR_ARG;
        if (a->bn_phase < 0 || a->bn_phase > 2) return MMVAE_ERR_ARG;
        if (a->bn_phase != 1 && (!a->stat1 || !a->stat2)) return MMVAE_ERR_ARG;
        if (a->bn_phase == 1 && !a->bn_coef) return MMVAE_ERR_ARG;
        if (a->bn_phase != 0 && !a->c) return MMVAE_ERR_ARG;
        if (a->bn_phase && (a->c_dtype == MMVAE_BF16) != (sizeof(LP) == 2)) return MMVAE_ERR_DTYPE;
        if (sizeof(LP) == 2) {
            if (a->ldh % 8 || ((uintptr_t)a->h & 15)) return MMVAE_ERR_ARG;
            if (a->bn_phase && (a->ldc % 8 || ((uintptr_t)a->c & 15))) return MMVAE_ERR_ARG;
            if (a->epi_mask && (a->N % 16 || a->ld_epi_mask % 16 || ((uintptr_t)a->epi_mask & 15))) return MMVAE_ERR_ARG;
        }
        EpiBnBwd<LP, LP> e{(LP*)a->c, a->ldc, (const LP*)a->h, a->ldh, a->epi_mask, a->ld_epi_mask,
                           a->bn_scale, a->bn_shift, a->bn_mean, a->bn_rstd, a->epi_inv_keep, a->bn_coef, a->bn_phase,
                           a->bn_phase == 1 ? nullptr : a->stat1, a->bn_phase == 1 ? nullptr : a->stat2};
        return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st);
    }
    }
    return MMVAE_ERR_ARG;
}

template <typename CT>
static int dispatch_src(const mmvae_gemm_nt_args* a, hipStream_t st) {
    constexpr int EPC = Mma<CT>::EPC;
    if (a->prologue == MMVAE_PRO_BN_RELU_DROP) {
        // A must be the activation type of this precision mode, 16-byte aligned rows
        if ((a->a_dtype == MMVAE_BF16) != (sizeof(CT) == 2)) return MMVAE_ERR_DTYPE;
        if (a->K > 512 || a->lda % EPC || ((uintptr_t)a->a & 15) || !a->pro_scale || !a->pro_shift) return MMVAE_ERR_ARG;
        if (a->pro_mask && (a->ld_pro_mask % 4 || ((uintptr_t)a->pro_mask & 3))) return MMVAE_ERR_ARG;
        SrcBnReluDrop<CT> s{(const CT*)a->a, a->lda, a->M, a->K, a->pro_scale, a->pro_shift, a->pro_mask, a->ld_pro_mask, a->pro_inv_keep};
        return dispatch_epi<CT>(a, s, st);
    }
    if (a->a_dtype == MMVAE_BF16) {
        if constexpr (sizeof(CT) == 2) {
            if (a->lda % 8 || ((uintptr_t)a->a & 15)) return MMVAE_ERR_ARG;
            if (ring_ok(a->a, a->lda, a->M)) { RingSrc s{a->a, a->lda}; return dispatch_epi<CT>(a, s, st); }
            SrcPlain<CT, bf16, 8> s{(const bf16*)a->a, a->lda, a->M, a->K};
            return dispatch_epi<CT>(a, s, st);
        }
        return MMVAE_ERR_DTYPE;
    }
    // f32 source with whatever alignment the caller's tensor has (e.g. (B,782): 8-byte rows)
    const uintptr_t p = (uintptr_t)a->a;
    if (a->lda % 4 == 0 && a->K % 4 == 0 && (p & 15) == 0) {
        SrcPlain<CT, float, 4> s{(const float*)a->a, a->lda, a->M, a->K};
        return dispatch_epi<CT>(a, s, st);
    }
    if (a->lda % 2 == 0 && a->K % 2 == 0 && (p & 7) == 0) {
        SrcPlain<CT, float, 2> s{(const float*)a->a, a->lda, a->M, a->K};
        return dispatch_epi<CT>(a, s, st);
    }
    SrcPlain<CT, float, 1> s{(const float*)a->a, a->lda, a->M, a->K};
    return dispatch_epi<CT>(a, s, st);
}

}  // namespace mm

extern "C" int mmvae_set_tuning(int32_t key, int32_t value) {
    if (key == 0) { mm::g_wide_min_m = value; return 0; }
    return MMVAE_ERR_ARG;
}

extern "C" int mmvae_gemm_nt(const mmvae_gemm_nt_args* a, void* stream) {
    if (!a || !a->a || !a->w || (!a->c && !(a->epilogue == MMVAE_EPI_BN_BWD && a->bn_phase == 0))) return MMVAE_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MMVAE_ERR_ARG;
    if (a->ldw % 64 || ((uintptr_t)a->w & 15)) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (a->prec == MMVAE_PREC_BF16) return mm::dispatch_src<mm::bf16>(a, st);
    if (a->prec == MMVAE_PREC_F32) return mm::dispatch_src<float>(a, st);
    return MMVAE_ERR_ARG;
}
