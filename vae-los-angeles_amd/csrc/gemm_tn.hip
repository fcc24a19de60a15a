// TN GEMM for gfx950:  dW[N,K] += P[M,N]^T * Q[M,K]   (+ db[N] += column sums of P)
//
// Replaces the dW / db half of every nn.Linear backward on the MultiModalVAE training path
// (autograd mm + sum, reference caller optimize_hyperparameters.py:112; layers as listed in
// gemm_nt.hip).  The reduction index is the batch row m, which is the SLOW index of both
// operands in memory, so both MFMA operands need a transposed fragment:
//   * bf16: tiles are staged row-major in LDS ([64 m][128 cols], 32-byte column chunks
//     XOR-swizzled by row) and fragments are fetched with ds_read_b64_tr_b16, the gfx950
//     hardware transpose read (two per 16x16x32 operand), bank-conflict free;
//   * f32 : v_mfma_f32_16x16x4_f32 takes ONE element per lane, so plain ds_read_b32 of the
//     row-major tile ([32 m][128 cols], 64-byte chunks swizzled by row parity) suffices.
// Q may carry the BN-normalise + ReLU + Dropout prologue (it is then the previous layer's
// pre-BN output), so post-activation tensors are never materialised.
// The batch is split over `nsplit` workgroups per output tile; partial tiles are combined with
// f32 global atomics (memory-side adds on gfx950; dW must be zeroed by the caller).  All tiles of
// one batch split run on one XCD so that P / Q rows are fetched from HBM once and shared in L2.
#include "common.h"
#include "mmvae_hip.h"
#include "gemm_src.h"
#include "gemm_ring.h"

namespace mm {

template <typename CT> struct TnGeom;
template <> struct TnGeom<bf16> {
    static constexpr int ROWB = 256, CPR = 16, MT = 64;
    static __device__ __forceinline__ int f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
    // byte offset of 16-byte chunk ch of row r
    static __device__ __forceinline__ int chunk_off(int r, int ch) { return r * ROWB + (((((ch >> 1) ^ f(r)) << 1) | (ch & 1)) << 4); }
    static __device__ __forceinline__ int elem_off(int r, int col) { return r * ROWB + ((((col >> 4) ^ f(r))) << 5) + ((col & 15) << 1); }
};
template <> struct TnGeom<float> {
    static constexpr int ROWB = 512, CPR = 32, MT = 32;
    static __device__ __forceinline__ int chunk_off(int r, int ch) { return r * ROWB + ((ch ^ ((r & 1) << 2)) << 4); }
    static __device__ __forceinline__ int elem_off(int r, int col) { return r * ROWB + ((((col >> 2) ^ ((r & 1) << 2))) << 4) + ((col & 3) << 2); }
};

// transposed operand fragment for MFMA tile at columns [colbase, colbase+16), fragment step s
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* tile, int colbase, int s, int lane, bf16*) {
    typedef TnGeom<bf16> G;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r = s * 32 + 8 * g + q;
    const int c32 = colbase >> 4;
    const int off0 = r * G::ROWB + ((c32 ^ G::f(r)) << 5) + (p << 3);
    const int off1 = off0 + 4 * G::ROWB;            // rows +4: f() unchanged
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}
__device__ __forceinline__ f32x4 tn_frag(const unsigned char* tile, int colbase, int s, int lane, float*) {
    typedef TnGeom<float> G;
    const int g = lane >> 4, col = colbase + (lane & 15);
    f32x4 v;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const float*)(tile + G::elem_off(s * 16 + 4 * u + g, col));
    return v;
}

template <typename CT, typename PSrc, typename QSrc>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_tn_kernel(PSrc ps, QSrc qs, float* __restrict__ dW, long ldw, float* __restrict__ db,
                    int M, int N, int K, int ntk, int ntiles, int nsplit, int rows_per_split, float* __restrict__ slab)
{
    typedef TnGeom<CT> G;
    constexpr int EPC = Mma<CT>::EPC;
    typedef typename Mma<CT>::frag frag;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::MT * G::ROWB + 4096];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + G::MT * G::ROWB;
    float* aux = (float*)(smem + 2 * G::MT * G::ROWB);

    const int L = blockIdx.x, slot = L >> 3;
    const int tile = slot % ntiles;
    const int zz = (slot / ntiles) * 8 + (L & 7);
    if (zz >= nsplit) return;
    const int tn = tile / ntk, tk = tile % ntk;
    const int n0 = tn * TILE, k0 = tk * TILE;
    const int m_begin = zz * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    if (m_begin >= m_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    if (QSrc::NEEDS_AUX) { qs.init(aux, tid); __syncthreads(); }

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (db != nullptr) && tk == 0 && wc == 0;     // wave-uniform

    typename PSrc::Raw rp[4];
    typename QSrc::Raw rq[4];
    const int nt = (m_end - m_begin + G::MT - 1) / G::MT;

    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTHREADS * i, r = c / G::CPR, ch = c % G::CPR;
            int m = m_begin + t * G::MT + r;
            if (m >= m_end) m = M;            // rows past this split read as zeros
            ps.fetch(rp[i], m, n0 + ch * EPC);
            qs.fetch(rq[i], m, k0 + ch * EPC);
        }
    };
    auto stage = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTHREADS * i, r = c / G::CPR, ch = c % G::CPR;
            int m = m_begin + t * G::MT + r;
            if (m >= m_end) m = M;
            Chunk<CT> o;
            ps.finish(rp[i], m, n0 + ch * EPC, o, aux);
            *(decltype(o.v)*)(sP + G::chunk_off(r, ch)) = o.v;
            qs.finish(rq[i], m, k0 + ch * EPC, o, aux);
            *(decltype(o.v)*)(sQ + G::chunk_off(r, ch)) = o.v;
        }
    };

    fetch(0);
    for (int t = 0; t < nt; ++t) {
        stage(t);
        __syncthreads();
        if (t + 1 < nt) fetch(t + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            frag af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = tn_frag(sP, wr * 64 + a * 16, s, lane, (CT*)nullptr);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = tn_frag(sQ, wc * 64 + b * 16, s, lane, (CT*)nullptr);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Mma<CT>::mma(acc[a][b], af[a], bf[b]);
            if (do_bias) {                      // column sums of P straight from the A fragments (VALU is idle here)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < Mma<CT>::EPC; ++j) bsum[a] += to_f32(af[a][j]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wr * 64 + a * 16 + (lane >> 4) * 4 + j;
            if (n >= N) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = k0 + wc * 64 + b * 16 + (lane & 15);
                if (k < K) {
                    // slab form: this split's partial tile is stored plainly into slab[zz] and summed by tn_reduce_kernel
                    // (deterministic, no memory-side atomics); otherwise f32 atomics straight into dW
                    if (slab) slab[((long)zz * N + n) * K + k] = acc[a][b][j];
                    else unsafeAtomicAdd(dW + (long)n * ldw + k, acc[a][b][j]);
                }
            }
        }
    if (do_bias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            const int n = n0 + wr * 64 + a * 16 + lane;
            if (lane < 16 && n < N) unsafeAtomicAdd(db + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-ring variant for plain bf16 operands (see gemm_ring.h): P and Q tiles ([64 m][128 cols] bf16,
// 16 KiB each) are DMA'd straight into a 4-stage ring; the tr16 swizzle goes on the source address.
// Rows past M read as zeros through the buffer resource; columns past N / K only feed accumulators
// that are never stored.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 1)
void gemm_tn_ring_kernel(const bf16* __restrict__ P, long ldp, unsigned p_bytes, const bf16* __restrict__ Q, long ldq, unsigned q_bytes,
                         float* __restrict__ dW, long ldw, float* __restrict__ db,
                         int M, int N, int K, int ntk, int ntiles, int nsplit, int rows_per_split, float* __restrict__ slab)
{
    typedef TnGeom<bf16> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = blockIdx.x, slot = L >> 3;
    const int tile = slot % ntiles;
    const int zz = (slot / ntiles) * 8 + (L & 7);
    if (zz >= nsplit) return;
    const int tn = tile / ntk, tk = tile % ntk;
    const int n0 = tn * TILE, k0 = tk * TILE;
    const int m_begin = zz * rows_per_split;              // multiple of 64: only the global tail is ragged
    const int m_end = min(M, m_begin + rows_per_split);
    if (m_begin >= m_end) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;

    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, p_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, q_bytes, 0x00020000);
    // piece p (1 KiB) = tile rows 4p..4p+3 (256-byte rows); lane -> row 4p + (lane>>4), physical chunk lane&15,
    // holding logical chunk (((phys>>1) ^ f(row)) << 1) | (phys & 1)   (inverse of TnGeom::chunk_off)
    unsigned p_off[4], q_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * (wid + 4 * i) + (lane >> 4), ph = lane & 15;
        const int ch = ((((ph >> 1) ^ G::f(r)) << 1) | (ph & 1));
        p_off[i] = (unsigned)(((long)(m_begin + r) * ldp + n0 + ch * 8) * 2);
        q_off[i] = (unsigned)(((long)(m_begin + r) * ldq + k0 + ch * 8) * 2);
    }
    const unsigned p_step = (unsigned)(64 * ldp * 2), q_step = (unsigned)(64 * ldq * 2);
    auto issue = [&](int t) {
        unsigned char* st = smem + (t % RING_NS) * RING_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wid + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_void*)(st + p * 1024), 16, p_off[i] + t * p_step, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void*)(st + 16384 + p * 1024), 16, q_off[i] + t * q_step, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (db != nullptr) && tk == 0 && wc == 0;
    const int nt = (m_end - m_begin + G::MT - 1) / G::MT;

#pragma unroll
    for (int s = 0; s < RING_NS - 1; ++s) if (s < nt) issue(s);
    for (int t = 0; t < nt; ++t) {
        ring_wait(min(RING_NS - 2, nt - 1 - t));
        ring_barrier();
        if (t + RING_NS - 1 < nt) issue(t + RING_NS - 1);
        const unsigned char* sP = smem + (t % RING_NS) * RING_STAGE;
        const unsigned char* sQ = sP + 16384;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = tn_frag(sP, wr * 64 + a * 16, s, lane, (bf16*)nullptr);
#pragma unroll
            for (int b = 0; b < 4; ++b) bfr[b] = tn_frag(sQ, wc * 64 + b * 16, s, lane, (bf16*)nullptr);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Mma<bf16>::mma(acc[a][b], af[a], bfr[b]);
            if (do_bias) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bsum[a] += to_f32(af[a][j]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wr * 64 + a * 16 + (lane >> 4) * 4 + j;
            if (n >= N) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = k0 + wc * 64 + b * 16 + (lane & 15);
                if (k < K) {
                    if (slab) slab[((long)zz * N + n) * K + k] = acc[a][b][j];
                    else unsafeAtomicAdd(dW + (long)n * ldw + k, acc[a][b][j]);
                }
            }
        }
    if (do_bias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            const int n = n0 + wr * 64 + a * 16 + lane;
            if (lane < 16 && n < N) unsafeAtomicAdd(db + n, v);
        }
    }
}

// dW[n][k] += sum_z slab[z][n][k]   (fixed summation order -> bitwise reproducible weight gradients)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, int nsplit, long nk, float* __restrict__ dW,
                                                        long ldw, int K) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll 8
        for (int z = 0; z < nsplit; ++z) s += slab[z * nk + i];       // independent loads: 8 in flight per thread
        const long n = i / K, k = i - n * K;
        dW[n * ldw + k] += s;
    }
}

// Slabs pay when a LARGE weight matrix is split a few dozen ways (tens of MB of atomics otherwise); a small matrix split
// hundreds of ways (heads, 40 x 256) would turn the reduce into a latency-bound crawl: those keep the f32 atomics.
static bool tn_use_slab(const mmvae_gemm_tn_args* a, int nsplit) {
    const long nk = (long)a->N * a->K;
    return a->slab && nsplit > 1 && nsplit <= 64 && nk >= 32768 && nsplit * nk <= a->slab_elems;
}

static int tn_reduce(const mmvae_gemm_tn_args* a, int nsplit, hipStream_t st) {
    const long nk = (long)a->N * a->K;
    int grid = (int)((nk + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(grid), dim3(256), 0, st, a->slab, nsplit, nk, a->dw, a->lddw, a->K);
    MM_CHECK_LAUNCH();
    return 0;
}

static void tn_split(int M, int N, int K, int MT, int nsplit_req, int& ntk, int& ntiles, int& nsplit, int& rps) {
    const int ntn = (N + TILE - 1) / TILE;
    ntk = (K + TILE - 1) / TILE; ntiles = ntn * ntk;
    nsplit = nsplit_req;
    if (nsplit <= 0) {
        // ONE resident round: at most 2 workgroups per CU (512) in total and -- because split z runs on XCD z % 8 --
        // a multiple of 8 splits so that every XCD gets the same share (a 520-block grid costs a whole extra round).
        // Every split adds one f32 atomic per output element; keep at least 4 m-tiles of work per workgroup.
        nsplit = 512 / ntiles;
        if (nsplit >= 8) nsplit &= ~7;
        int max_split = (M + 4 * MT - 1) / (4 * MT);
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    }
    rps = (M + nsplit - 1) / nsplit;
    rps = ((rps + MT - 1) / MT) * MT;
    nsplit = (M + rps - 1) / rps;
}

static int launch_tn_ring(const mmvae_gemm_tn_args* a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    int ntk, ntiles, nsplit, rps;
    tn_split(a->M, a->N, a->K, 64, a->nsplit, ntk, ntiles, nsplit, rps);
    const int grid = ((nsplit + 7) / 8) * 8 * ntiles;
    float* slab = tn_use_slab(a, nsplit) ? a->slab : nullptr;
    hipLaunchKernelGGL(gemm_tn_ring_kernel, dim3(grid), dim3(NTHREADS), RING_LDS, st,
                       (const bf16*)a->p, a->ldp, (unsigned)((long)a->M * a->ldp * 2), (const bf16*)a->q, a->ldq,
                       (unsigned)((long)a->M * a->ldq * 2), a->dw, a->lddw, a->db, a->M, a->N, a->K, ntk, ntiles, nsplit, rps, slab);
    MM_CHECK_LAUNCH();
    return slab ? tn_reduce(a, nsplit, st) : 0;
}

template <typename CT, typename PSrc, typename QSrc>
static int launch_tn(const mmvae_gemm_tn_args* a, const PSrc& ps, const QSrc& qs, hipStream_t st) {
    typedef TnGeom<CT> G;
    int ntk, ntiles, nsplit, rps;
    tn_split(a->M, a->N, a->K, G::MT, a->nsplit, ntk, ntiles, nsplit, rps);
    const int grid = ((nsplit + 7) / 8) * 8 * ntiles;
    float* slab = tn_use_slab(a, nsplit) ? a->slab : nullptr;
    hipLaunchKernelGGL((gemm_tn_kernel<CT, PSrc, QSrc>), dim3(grid), dim3(NTHREADS), 0, st, ps, qs,
                       a->dw, a->lddw, a->db, a->M, a->N, a->K, ntk, ntiles, nsplit, rps, slab);
    MM_CHECK_LAUNCH();
    return slab ? tn_reduce(a, nsplit, st) : 0;
}

template <typename CT, typename PSrc>
static int tn_dispatch_q(const mmvae_gemm_tn_args* a, const PSrc& ps, hipStream_t st) {
    constexpr int EPC = Mma<CT>::EPC;
    if (a->q_prologue == MMVAE_PRO_BN_RELU_DROP) {
        if ((a->q_dtype == MMVAE_BF16) != (sizeof(CT) == 2)) return MMVAE_ERR_DTYPE;
        if (a->K > 512 || a->ldq % EPC || ((uintptr_t)a->q & 15) || !a->pro_scale || !a->pro_shift) return MMVAE_ERR_ARG;
        if (a->pro_mask && (a->ld_pro_mask % 4 || ((uintptr_t)a->pro_mask & 3))) return MMVAE_ERR_ARG;
        SrcBnReluDrop<CT> q{(const CT*)a->q, a->ldq, a->M, a->K, a->pro_scale, a->pro_shift, a->pro_mask, a->ld_pro_mask, a->pro_inv_keep};
        return launch_tn<CT>(a, ps, q, st);
    }
    if (a->q_dtype == MMVAE_BF16) {
        if constexpr (sizeof(CT) == 2) {
            if (a->ldq % 8 || ((uintptr_t)a->q & 15)) return MMVAE_ERR_ARG;
            SrcPlain<CT, bf16, 8> q{(const bf16*)a->q, a->ldq, a->M, a->K};
            return launch_tn<CT>(a, ps, q, st);
        }
        return MMVAE_ERR_DTYPE;
    }
    const uintptr_t p = (uintptr_t)a->q;
    if (a->ldq % 4 == 0 && a->K % 4 == 0 && (p & 15) == 0) { SrcPlain<CT, float, 4> q{(const float*)a->q, a->ldq, a->M, a->K}; return launch_tn<CT>(a, ps, q, st); }
    if (a->ldq % 2 == 0 && a->K % 2 == 0 && (p & 7) == 0) { SrcPlain<CT, float, 2> q{(const float*)a->q, a->ldq, a->M, a->K}; return launch_tn<CT>(a, ps, q, st); }
    SrcPlain<CT, float, 1> q{(const float*)a->q, a->ldq, a->M, a->K};
    return launch_tn<CT>(a, ps, q, st);
}

template <typename CT>
static int tn_dispatch_p(const mmvae_gemm_tn_args* a, hipStream_t st) {
    if (a->p_dtype == MMVAE_BF16) {
        if constexpr (sizeof(CT) == 2) {
            if (a->ldp % 8 || ((uintptr_t)a->p & 15)) return MMVAE_ERR_ARG;
            SrcPlain<CT, bf16, 8> p{(const bf16*)a->p, a->ldp, a->M, a->N};
            return tn_dispatch_q<CT>(a, p, st);
        }
        return MMVAE_ERR_DTYPE;
    }
    const uintptr_t pp = (uintptr_t)a->p;
    if (a->ldp % 4 == 0 && a->N % 4 == 0 && (pp & 15) == 0) { SrcPlain<CT, float, 4> p{(const float*)a->p, a->ldp, a->M, a->N}; return tn_dispatch_q<CT>(a, p, st); }
    if (a->ldp % 2 == 0 && a->N % 2 == 0 && (pp & 7) == 0) { SrcPlain<CT, float, 2> p{(const float*)a->p, a->ldp, a->M, a->N}; return tn_dispatch_q<CT>(a, p, st); }
    SrcPlain<CT, float, 1> p{(const float*)a->p, a->ldp, a->M, a->N};
    return tn_dispatch_q<CT>(a, p, st);
}

}  // namespace mm

extern "C" int mmvae_gemm_tn(const mmvae_gemm_tn_args* a, void* stream) {
    if (!a || !a->p || !a->q || !a->dw) return MMVAE_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MMVAE_ERR_ARG;
    // the operand sources address P, Q and the prologue mask with 32-bit byte offsets from a scalar base
    const long lim = 1L << 32;
    if ((long)a->M * a->ldp * (a->p_dtype == MMVAE_BF16 ? 2 : 4) >= lim || (long)a->M * a->ldq * (a->q_dtype == MMVAE_BF16 ? 2 : 4) >= lim) return MMVAE_ERR_ARG;
    if (a->pro_mask && (long)a->M * a->ld_pro_mask >= lim) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (a->prec == MMVAE_PREC_BF16) {
        if (a->p_dtype == MMVAE_BF16 && a->q_dtype == MMVAE_BF16 && a->q_prologue == MMVAE_PRO_NONE &&
            mm::ring_ok(a->p, a->ldp, a->M) && mm::ring_ok(a->q, a->ldq, a->M))
            return mm::launch_tn_ring(a, st);
        return mm::tn_dispatch_p<mm::bf16>(a, st);
    }
    if (a->prec == MMVAE_PREC_F32) return mm::tn_dispatch_p<float>(a, st);
    return MMVAE_ERR_ARG;
}
