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
// Pipeline: two global register sets (one for f32 sources), double-buffered LDS with one barrier per
// batch step, steady-state loop without conditional fetches (see gemm_nt.hip for why).  tools/stamp_tn.py
// shows where a step goes: the global loads are never waited for; the registers -> LDS stage and the
// transpose reads are what the MFMAs wait on.
// The batch is split over `nsplit` workgroups per output tile; partial tiles are combined with
// f32 global atomics (memory-side adds on gfx950; dW must be zeroed by the caller).  All tiles of
// one batch split run on one XCD so that P / Q rows are fetched from HBM once and shared in L2.
#include "common.h"
#include "mmvae_hip.h"
#include "gemm_src.h"

#ifndef MM_REDUCE_U
#define MM_REDUCE_U 16      // loads a reduce thread keeps in flight (step at B = 65 536: 4 -> +8 us, 8 -> +2.5, 32 -> +8.5 against 16)
#endif

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

#ifdef MM_STAMP
// Diagnostic build only (make STAMP=1, tools/stamp_tn.py): cycles per batch step of {fragment step 0, stage, fragment step 1
// + fetch issue, barrier}, steps, waves, whole-kernel cycles, cycles before the loop, cycles after it.
__device__ unsigned long long mm_stamps_tn[12];
#define MT_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define MT_T(x)
#endif

// One workgroup's share of a dW GEMM: output tile and batch split from its LOCAL block id L (0 .. grid of this problem), shared
// by the one-problem kernel and the grouped kernel below.
// DMA = true (both operands plain and already in the compute type, every batch step of every split a full MT rows): the tiles
// are moved by LDS-DMA (global_load_lds_dwordx4, 4 rows x 256 B per wave-instruction, the chunk swizzle applied to the SOURCE
// address) instead of global -> VGPR -> ds_write; the registers -> LDS pass was what the MFMAs of this kernel waited for
// (tools/stamp_tn.py: stage 24 % of a step + the fragment step that follows it stalled on it).
// NG = 2 (DMA form only): EIGHT waves in two groups of four; every wave still owns a 64x64 piece of the 128x128 tile, group g
// multiplies rows [32 g, 32 g + 32) of every 64-row step and the two partial tiles are added through LDS before the slab
// store -- twice the rows per workgroup at the same number of waves per CU, i.e. HALF the slab (its store + the reduce's
// read were 67 MB next to 100 MB of operands for a 512x256 dW at batch 65 536).
template <typename CT, typename PSrc, typename QSrc, bool DMA = false, int NG = 1>
__device__ __forceinline__
void tn_body(const PSrc& ps, const QSrc& qs, float* __restrict__ dW, long ldw, float* __restrict__ db,
             int M, int N, int K, int ntk, int ntiles, int nsplit, int rows_per_split, float* __restrict__ slab,
             const int L, unsigned char* smem)
{
    typedef TnGeom<CT> G;
    constexpr int EPC = Mma<CT>::EPC;
    typedef typename Mma<CT>::frag frag;
    constexpr int BUF = 2 * G::MT * G::ROWB;                 // one batch step: P tile + Q tile (32 KiB)
    float* aux = (float*)(smem + 2 * BUF);                   // Q prologue: BN scale / shift
    float* auxp = aux + 1024;                                // P prologue: BN-backward constants of this tile's 128 columns

    const int slot = L >> 3;
    const int tile = slot % ntiles;
    const int zz = (slot / ntiles) * 8 + (L & 7);
    if (zz >= nsplit) return;
    const int tn = tile / ntk, tk = tile % ntk;
    const int n0 = tn * TILE, k0 = tk * TILE;
    const int m_begin = zz * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    if (m_begin >= m_end) return;

    static_assert(NG == 1 || (NG == 2 && DMA), "two wave groups: DMA form only");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = (wid & 3) >> 1, wc = wid & 1;

    if (QSrc::NEEDS_AUX) qs.init(aux, tid, k0);
    if (PSrc::NEEDS_AUX) ps.init(auxp, tid, n0);
    if (QSrc::NEEDS_AUX || PSrc::NEEDS_AUX) __syncthreads();

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (db != nullptr) && tk == 0 && wc == 0;     // wave-uniform

    // Same pipeline as the NT kernel (gemm_nt.hip): two global register sets, double-buffered LDS with ONE barrier per
    // batch step, fragment registers double-buffered so the transpose reads run one fragment step ahead of the MFMAs,
    // and a steady-state loop without conditional fetches so that hipcc can count its vmcnt waits.
    typename PSrc::Raw rp0[4], rp1[4];
    typename QSrc::Raw rq0[4], rq1[4];
    const int nt = (m_end - m_begin + G::MT - 1) / G::MT;

    auto fetch = [&](typename PSrc::Raw (&rp)[4], typename QSrc::Raw (&rq)[4], int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTHREADS * i, r = c / G::CPR, ch = c % G::CPR;
            int m = m_begin + t * G::MT + r;
            m = m >= m_end ? M : m;           // rows past this split read as zeros (finish() zeroes rows >= M)
            ps.fetch(rp[i], m, n0 + ch * EPC);
            qs.fetch(rq[i], m, k0 + ch * EPC);
        }
    };
    auto stage = [&](typename PSrc::Raw (&rp)[4], typename QSrc::Raw (&rq)[4], int t, int buf) {
        unsigned char* sP = smem + buf * BUF;
        unsigned char* sQ = sP + G::MT * G::ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTHREADS * i, r = c / G::CPR, ch = c % G::CPR;
            int m = m_begin + t * G::MT + r;
            m = m >= m_end ? M : m;
            Chunk<CT> o;
            ps.finish_rows(rp[i], m, n0 + ch * EPC, o, auxp);        // P rows past the split are zeros: they drop out of the reduction
            *(decltype(o.v)*)(sP + G::chunk_off(r, ch)) = o.v;
            qs.finish_fast(rq[i], k0 + ch * EPC, o, aux);             // Q unmasked (finite, and multiplied by the zeroed P rows)
            *(decltype(o.v)*)(sQ + G::chunk_off(r, ch)) = o.v;
        }
    };
    // ONE fragment set here (the NT kernel double-buffers it): the transpose reads need two LDS addresses per fragment and
    // the bias sums ride along; a second fragment set on top of two global register sets spilled inside the loop, and a
    // spill reload is a vmcnt(0) wait that drains the prefetch.
    auto compute = [&](int buf, int s) {
        const unsigned char* sP = smem + buf * BUF;
        const unsigned char* sQ = sP + G::MT * G::ROWB;
        frag af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = tn_frag(sP, wr * 64 + a * 16, s, lane, (CT*)nullptr);
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = tn_frag(sQ, wc * 64 + b * 16, s, lane, (CT*)nullptr);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) Mma<CT>::mma(acc[a][b], bf[b], af[a]);     // swapped: a lane holds 4 consecutive k of one n (epilogue)
        if (do_bias) {                      // column sums of P straight from the A fragments (VALU is idle here)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < Mma<CT>::EPC; ++j) bsum[a] += to_f32(af[a][j]);
        }
    };
    // one batch step: tile t is multiplied from LDS buffer `buf` while tile t+1 goes registers -> other buffer between the
    // two fragment steps and a later tile is fetched into the registers just freed; ONE barrier per step
#ifdef MM_STAMP
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, t_first = 0, t_last = 0, st_acc5 = 0;
    MT_T(t_begin);
#endif
    // registers -> LDS writes and the global loads are spread between the MFMAs of the two fragment steps (see gemm_nt.hip)
    auto kstep = [&](auto& rp_n, auto& rq_n, int t, int buf, bool has_next, bool do_fetch, int fetch_t) {
        MT_T(t0);
        compute(buf, 0);
        MT_T(t1);
#ifdef MM_STAMP
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // steady state: the set about to be staged has landed
        MT_T(t1b);
        st_acc5 += t1b - t1;
#endif
        if (has_next) stage(rp_n, rq_n, t + 1, buf ^ 1);
#ifndef MM_STAMP
#pragma unroll
        for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#endif
        MT_T(t2);
        compute(buf, 1);
        if (do_fetch) fetch(rp_n, rq_n, fetch_t);               // into the registers just staged
#ifndef MM_STAMP
#pragma unroll
        for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#endif
#ifdef MM_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        MT_T(t3);
        __syncthreads();
        MT_T(t4);
#ifdef MM_STAMP
        if (st_acc[4] == 0) t_first = t0;
        t_last = t4;
        st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += 1;
#endif
    };
    // f32 sources hold 32 bytes per chunk in flight (two sets spill) and the BN-prologue source measured slower with two:
    // those run one register set, one step ahead
    constexpr bool TWO_SETS = sizeof(typename PSrc::Raw) + sizeof(typename QSrc::Raw) <= 32;       // plain bf16 x plain bf16 only
    if constexpr (DMA) {
        static_assert(sizeof(CT) == 2, "the DMA form moves bf16 tiles");
        typedef __attribute__((address_space(3))) void lds_void;
        typedef __attribute__((address_space(1))) const void gbl_void;
        const int wv = __builtin_amdgcn_readfirstlane(wid);
        const int prow = lane >> 4, ppos = lane & 15;
        const int ncap = ((N + 7) & ~7) - 8, kcap = ((K + 7) & ~7) - 8;        // last whole 16-byte chunk inside a row
        auto issue = [&](int t, int buf) {
            unsigned char* sP = smem + buf * BUF;
            unsigned char* sQ = sP + G::MT * G::ROWB;
            const int m0 = m_begin + t * G::MT;                  // every step is a full MT rows (host-checked)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = wv + 4 * i, r = p * 4 + prow;
                const int ch = (((ppos >> 1) ^ G::f(r)) << 1) | (ppos & 1);      // chunk whose swizzled position is ppos
                const unsigned rp_ = (unsigned)(m0 + r);
                const bf16* gp = ps.p + (rp_ * (unsigned)ps.lda + (unsigned)min(n0 + ch * 8, ncap));
                const bf16* gq = qs.p + (rp_ * (unsigned)qs.lda + (unsigned)min(k0 + ch * 8, kcap));
                __builtin_amdgcn_global_load_lds((gbl_void*)gp, (lds_void*)(sP + p * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)gq, (lds_void*)(sQ + p * 1024), 16, 0, 0);
            }
        };
        if constexpr (NG == 2) {
            // One workgroup per CU: a ring of FOUR 64-row buffers, three steps under way while one is multiplied (with one step
            // ahead a lone workgroup had 32 KB in flight per CU and waited for the trip from HBM in every step).  Issued from
            // inline assembly and waited for by counted vmcnt (see tnw_dma16 in gemm_tn_wide.hip for why not the builtin);
            // the ring is unrolled so that every buffer is a compile-time offset.
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
            auto dma16 = [&](const void* g, unsigned lds_addr) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(g) : "memory");     // nothing else here uses m0
            };
            auto issue4 = [&](int t, int slot) {
                const unsigned sP = lds0 + slot * BUF, sQ = sP + G::MT * G::ROWB;
                const int m0 = m_begin + t * G::MT;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int p = wv + 8 * i, r = p * 4 + prow;
                    const int ch = (((ppos >> 1) ^ G::f(r)) << 1) | (ppos & 1);
                    const unsigned rp_ = (unsigned)(m0 + r);
                    dma16(ps.p + (rp_ * (unsigned)ps.lda + (unsigned)min(n0 + ch * 8, ncap)), sP + p * 1024);
                    dma16(qs.p + (rp_ * (unsigned)qs.lda + (unsigned)min(k0 + ch * 8, kcap)), sQ + p * 1024);
                }
            };
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) if (s_ < nt) issue4(s_, s_);
            for (int t0 = 0; t0 < nt; t0 += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u;
                    if (t >= nt) break;
                    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // steps t + 1, t + 2 (4 transfers each) may still be under way
                    else if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                // everybody's pieces of step t have landed; nobody still reads step t - 1
                    asm volatile("" ::: "memory");
                    if (t + 3 < nt) issue4(t + 3, (u + 3) & 3);  // into the buffer step t - 1 was read from
                    compute(u, wv >> 2);                         // (fragment registers double-buffered across steps: measured 1-3 us slower)
                }
            }
        } else {
        issue(0, 0);
        for (int t = 0; t < nt; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own pieces of tile t have landed
            __syncthreads();                                     // everybody's have; nobody still reads the other buffer
            if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
            compute(t & 1, 0);
            compute(t & 1, 1);
        }
        }
        if constexpr (NG == 2) {
            // group 1's partial tile (+ its column sums of P) -> LDS -> added by group 0, which stores
            __syncthreads();                                     // the ring is free
            float* red = (float*)smem + (wid & 3) * 4096;        // 16 KB per wave pair, 16 bytes per lane and accumulator: conflict-free
            float* redb = (float*)(smem + 2 * BUF) + (wid & 3) * 256;            // behind the 64 KB of partial tiles, inside the (free) ring
            if (wv >> 2) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) *(f32x4*)(red + ((a * 4 + b) * 64 + lane) * 4) = acc[a][b];
                if (do_bias) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) redb[a * 64 + lane] = bsum[a];
                }
            }
            __syncthreads();
            if (wv >> 2) return;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] += *(const f32x4*)(red + ((a * 4 + b) * 64 + lane) * 4);
            if (do_bias) {
#pragma unroll
                for (int a = 0; a < 4; ++a) bsum[a] += redb[a * 64 + lane];
            }
        }
    } else if constexpr (!TWO_SETS) {
        fetch(rp0, rq0, 0);
        stage(rp0, rq0, 0, 0);
        if (nt > 1) fetch(rp0, rq0, 1);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            compute(t & 1, 0);
            if (t + 1 < nt) stage(rp0, rq0, t + 1, (t + 1) & 1);
            if (t + 2 < nt) fetch(rp0, rq0, t + 2);               // one set: issue the next loads the moment the registers are free
            compute(t & 1, 1);
            __syncthreads();
        }
    } else if (nt >= 5) {
        fetch(rp0, rq0, 0);
        fetch(rp1, rq1, 1);
        stage(rp0, rq0, 0, 0);
        fetch(rp0, rq0, 2);
        __syncthreads();
        int t = 0;
        for (; t + 4 < nt; t += 2) {
            kstep(rp1, rq1, t, 0, true, true, t + 3);
            kstep(rp0, rq0, t + 1, 1, true, true, t + 4);
        }
        const bool four = t + 3 < nt;
        kstep(rp1, rq1, t, 0, true, four, t + 3);
        kstep(rp0, rq0, t + 1, 1, true, false, 0);
        kstep(rp1, rq1, t + 2, 0, four, false, 0);
        if (four) kstep(rp1, rq1, t + 3, 1, false, false, 0);
    } else {
        fetch(rp0, rq0, 0);
        if (nt > 1) fetch(rp1, rq1, 1);
        stage(rp0, rq0, 0, 0);
        if (nt > 2) fetch(rp0, rq0, 2);
        __syncthreads();
        kstep(rp1, rq1, 0, 0, nt > 1, nt > 3, 3);
        if (nt > 1) kstep(rp0, rq0, 1, 1, nt > 2, false, 0);
        if (nt > 2) kstep(rp1, rq1, 2, 0, nt > 3, false, 0);
        if (nt > 3) kstep(rp1, rq1, 3, 1, false, false, 0);
    }

    // The MFMA operands are swapped (Q fragment as "A", P fragment as "B"), so accumulator (a, b) holds, in lane (li = lane & 15,
    // lg = lane >> 4), dW[n = 16a + li][k = 16b + 4lg + j], j = 0..3: FOUR CONSECUTIVE k.  A split's partial tile goes to the slab
    // as 16-byte stores (16 instead of 64 store instructions per lane: the scalar form cost 10-29 k cycles per workgroup, more
    // than the 4 batch steps of a tiny-output GEMM).
    const bool vec4 = slab != nullptr && (K & 3) == 0 && ((uintptr_t)slab & 15) == 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n0 + wr * 64 + a * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + wc * 64 + b * 16 + (lane >> 4) * 4;
            if (k >= K) continue;
            if (slab) {
                // slab form: this split's partial tile is stored plainly into slab[zz] and summed by tn_reduce_kernel
                float* sp = slab + ((long)zz * N + n) * K + k;
                if (vec4 && k + 4 <= K) *(f32x4*)sp = acc[a][b];
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < K) sp[j] = acc[a][b][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < K) unsafeAtomicAdd(dW + (long)n * ldw + k + j, acc[a][b][j]);       // f32 atomics straight into dW
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
#ifdef MM_STAMP
    if (tid == 0 && (blockIdx.x & 7) == 3) {
        MT_T(t_end);
        for (int i = 0; i < 5; ++i) atomicAdd(&mm_stamps_tn[i], st_acc[i]);
        atomicAdd(&mm_stamps_tn[5], 1ull);
        atomicAdd(&mm_stamps_tn[6], t_end - t_begin);
        atomicAdd(&mm_stamps_tn[7], st_acc5);
        atomicAdd(&mm_stamps_tn[8], t_first - t_begin);
        atomicAdd(&mm_stamps_tn[9], t_end - t_last);
    }
#endif
}

template <typename CT, typename PSrc, typename QSrc, bool DMA, int NG = 1>
__global__ __launch_bounds__(NTHREADS * NG, NG == 2 ? 1 : 2)
void gemm_tn_kernel(PSrc ps, QSrc qs, float* __restrict__ dW, long ldw, float* __restrict__ db,
                    int M, int N, int K, int ntk, int ntiles, int nsplit, int rows_per_split, float* __restrict__ slab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 buffers + 4 KiB prologue scale/shift
    tn_body<CT, PSrc, QSrc, DMA, NG>(ps, qs, dW, ldw, db, M, N, K, ntk, ntiles, nsplit, rows_per_split, slab, (int)blockIdx.x, smem);
}

template <typename T> struct TnPlainBf16 { static constexpr bool value = false; };
template <> struct TnPlainBf16<SrcPlain<bf16, bf16, 8>> { static constexpr bool value = true; };
// DMA form: plain bf16 operands and every split a whole number of full batch steps
static inline bool tn_dma_ok(int M, int MT) {
    static const bool off = getenv("MMVAE_NO_TN_DMA") != nullptr;           // A/B switch
    return !off && M % MT == 0;
}

// ------------------------------------------------------------------------------------------
// Grouped launch: several SMALL-output dW GEMMs (latent / class widths: the heads of the encoders, the first layers of the
// decoders, DecoderC) in ONE launch + ONE reduce.  Alone each of them is a latency chain (launch ramp, first loads, a few batch
// steps, slab store, reduce launch: 22-29 us for a few MB), six of them took 157 us + 12 reduce launches of 6 us per step;
// side by side their chains overlap.  Operand-type combinations a problem may have:
//   0: P f32 (d_heads), Q activation type through the BatchNorm+ReLU+Dropout prologue   (encoder heads)
//   1: P, Q activation type                                                               (decoder first layers)
//   2: P f32 (loss gradient of the class logits), Q activation type                       (DecoderC.L1)
// ------------------------------------------------------------------------------------------
struct TnProblem {
    int combo, dma, M, N, K, ntk, ntiles, nsplit, rps, block0, nblocks;
    float* dW; long ldw; float* db; float* slab;
    const void* p; long ldp; const void* q; long ldq;
    const float* pro_scale; const float* pro_shift; const uint8_t* pro_mask; long ld_pro_mask; float pro_inv_keep;
};
struct TnGroup { TnProblem pr[MMVAE_TN_GROUP_MAX]; int n; };

template <typename CT>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_tn_group_kernel(const TnGroup g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MMVAE_TN_GROUP_MAX; ++i) if (i < g.n && (int)blockIdx.x >= g.pr[i].block0) pi = i;      // block ranges are ascending
    const TnProblem& r = g.pr[pi];
    const int L = (int)blockIdx.x - r.block0;
    typedef SrcPlain<CT, float, 4> PF;
    typedef SrcPlain<CT, CT, Mma<CT>::EPC> PA;          // activation-typed plain operand (EPC elements = 16 bytes)
    if (r.combo == 0) {
        const PF ps{(const float*)r.p, r.ldp, r.M, r.N};
        const SrcBnReluDrop<CT> qs{(const CT*)r.q, r.ldq, r.M, r.K, r.pro_scale, r.pro_shift, r.pro_mask, r.ld_pro_mask, r.pro_inv_keep};
        tn_body<CT>(ps, qs, r.dW, r.ldw, r.db, r.M, r.N, r.K, r.ntk, r.ntiles, r.nsplit, r.rps, r.slab, L, smem);
    } else if (r.combo == 1) {
        const PA ps{(const CT*)r.p, r.ldp, r.M, r.N};
        const PA qs{(const CT*)r.q, r.ldq, r.M, r.K};
        if constexpr (sizeof(CT) == 2) {
            if (r.dma) { tn_body<CT, PA, PA, true>(ps, qs, r.dW, r.ldw, r.db, r.M, r.N, r.K, r.ntk, r.ntiles, r.nsplit, r.rps, r.slab, L, smem); return; }
        }
        tn_body<CT>(ps, qs, r.dW, r.ldw, r.db, r.M, r.N, r.K, r.ntk, r.ntiles, r.nsplit, r.rps, r.slab, L, smem);
    } else {
        const PF ps{(const float*)r.p, r.ldp, r.M, r.N};
        const PA qs{(const CT*)r.q, r.ldq, r.M, r.K};
        tn_body<CT>(ps, qs, r.dW, r.ldw, r.db, r.M, r.N, r.K, r.ntk, r.ntiles, r.nsplit, r.rps, r.slab, L, smem);
    }
}

// sum_z p[z * stride], z = 0 .. n-1, added in that order; U loads are requested before the first is added (a thread of a reduce
// launch has nothing else to do; MM_REDUCE_U = 16 measured best)
template <int U>
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, long stride, int n) {
    float s = 0.f;
    int z = 0;
    for (; z + U <= n; z += U) {                    // whole batches without predicates (n is small for the big matrices:
        float t[U];                                 //  per-load predication cost them +11 us per launch at the scaled widths)
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = p[(long)(z + u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += t[u];
    }
    for (; z + 4 <= n; z += 4) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = p[(long)(z + u) * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += t[u];
    }
    for (; z < n; ++z) s += p[(long)z * stride];
    return s;
}

// dW_p[n][k] += sum_z slab_p[z][n][k] for every problem of a group, fixed order (bitwise reproducible)
struct TnGroupReduce { const float* slab[MMVAE_TN_GROUP_MAX]; float* dW[MMVAE_TN_GROUP_MAX]; int nsplit[MMVAE_TN_GROUP_MAX]; int nk[MMVAE_TN_GROUP_MAX];
                       int first[MMVAE_TN_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void tn_group_reduce_kernel(const TnGroupReduce g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.first[g.n]) return;
    int pi = 0;
#pragma unroll
    for (int j = 1; j < MMVAE_TN_GROUP_MAX; ++j) if (j < g.n && i >= g.first[j]) pi = j;
    const int e = i - g.first[pi], nk = g.nk[pi], ns = g.nsplit[pi];
    const float* sl = g.slab[pi] + e;
    const float old = g.dW[pi][e];                       // requested with the first slab loads, not behind the sum (one round trip fewer)
    g.dW[pi][e] = old + slab_sum<MM_REDUCE_U>(sl, nk, ns);                                   // lddw == K for these (contiguous gradient views)
}

#ifdef MM_STAMP
extern "C" int mmvae_debug_stamps_tn(unsigned long long* out12, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out12, HIP_SYMBOL(mm::mm_stamps_tn), 12 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    if (reset) { unsigned long long z[12] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(mm::mm_stamps_tn), z, sizeof(z)); }
    return (int)e;
}
#endif

// dW[n][k] += sum_z slab[z][n][k].  Splits are summed in groups of TN_RG (blockIdx.y): ONE group (<= TN_RG splits, the large
// weight matrices) is a fixed-order sum -> bitwise reproducible gradients; more groups (small matrices split hundreds of
// ways) add their partial sums with f32 atomics, nsplit / TN_RG ways per address instead of nsplit ways from the GEMM itself.
constexpr int TN_RG = 128;     // the wide-tile kernels split the batch 64 and 128 ways: still one fixed-order group
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, int nsplit, long nk, float* __restrict__ dW,
                                                        long ldw, int K) {
    const int z0 = blockIdx.y * TN_RG, z1 = min(nsplit, z0 + TN_RG);
    const bool single = gridDim.y == 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (long)gridDim.x * blockDim.x) {
        const long n = i / K, k = i - n * K;
        const float old = single ? dW[n * ldw + k] : 0.f;   // requested with the first slab loads, not behind the sum (one round trip fewer)
        const float s = slab_sum<MM_REDUCE_U>(slab + z0 * nk + i, nk, z1 - z0);
        if (single) dW[n * ldw + k] = old + s;
        else unsafeAtomicAdd(dW + n * ldw + k, s);
    }
}

// Slabs (plain stores of every split's partial tile + this reduce) instead of f32 atomics from the GEMM epilogue whenever the
// workspace holds them: atomics from hundreds of splits onto the same few thousand addresses were what the small dW GEMMs
// (latent / class widths) spent their time on.
static bool tn_use_slab(const mmvae_gemm_tn_args* a, int nsplit) {
    const long nk = (long)a->N * a->K;
    return a->slab && nsplit > 1 && nsplit * nk <= a->slab_elems;
}

static int tn_reduce(const mmvae_gemm_tn_args* a, int nsplit, hipStream_t st) {
    const long nk = (long)a->N * a->K;
    int grid = (int)((nk + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(grid, (nsplit + TN_RG - 1) / TN_RG), dim3(256), 0, st, a->slab, nsplit, nk, a->dw, a->lddw, a->K);
    MM_CHECK_LAUNCH();
    return 0;
}

static void tn_split(int M, int N, int K, int MT, int nsplit_req, int& ntk, int& ntiles, int& nsplit, int& rps, int wg_target = 512) {
    const int ntn = (N + TILE - 1) / TILE;
    ntk = (K + TILE - 1) / TILE; ntiles = ntn * ntk;
    nsplit = nsplit_req;
    if (nsplit <= 0) {
        // ONE resident round: at most 2 workgroups per CU (512) in total and -- because split z runs on XCD z % 8 --
        // a multiple of 8 splits so that every XCD gets the same share (a 520-block grid costs a whole extra round).
        // Every split adds one f32 atomic per output element; keep at least 4 m-tiles of work per workgroup.
        nsplit = wg_target / ntiles;
        if (nsplit >= 8) nsplit &= ~7;
        int max_split = (M + 4 * MT - 1) / (4 * MT);
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
    }
    rps = (M + nsplit - 1) / nsplit;
    rps = ((rps + MT - 1) / MT) * MT;
    nsplit = (M + rps - 1) / rps;
}

template <typename CT, typename PSrc, typename QSrc>
static int launch_tn(const mmvae_gemm_tn_args* a, const PSrc& ps, const QSrc& qs, hipStream_t st) {
    typedef TnGeom<CT> G;
    int ntk, ntiles, nsplit, rps;
    tn_split(a->M, a->N, a->K, G::MT, a->nsplit, ntk, ntiles, nsplit, rps);
    const int grid = ((nsplit + 7) / 8) * 8 * ntiles;
    float* slab = tn_use_slab(a, nsplit) ? a->slab : nullptr;
    constexpr int LDS = 4 * G::MT * G::ROWB + 4096 + 4096;
    if constexpr (sizeof(CT) == 2 && TnPlainBf16<PSrc>::value && TnPlainBf16<QSrc>::value) {
        // Two wave groups (one 8-wave workgroup per CU, half the slab) when an automatic split fills the chip that way: whole
        // multiples of 8 splits (XCD balance) on >= 7/8 of the CUs -- 8 or 7 tiles (512x256, 256x512, 782x128: 36-38 against 40-41 us
        // with the reduce).  The 20 tiles of 572x512 would make 160 workgroups, or 240 with 12 splits of which the last four are
        // shared by two XCDs each (tried: 69 against 64-65 us alone, the same inside the step).
        static const int ng_env = getenv("MMVAE_TN_NG") ? atoi(getenv("MMVAE_TN_NG")) : 0;      // A/B switch: 1 / 2 = never / always
        bool two = false;
        if (tn_dma_ok(a->M, G::MT) && ng_env != 1) {
            tn_split(a->M, a->N, a->K, G::MT, a->nsplit, ntk, ntiles, nsplit, rps, 256);
            two = ng_env == 2 || (a->nsplit <= 0 && nsplit % 8 == 0 && nsplit * ntiles >= 224);
            if (!two) tn_split(a->M, a->N, a->K, G::MT, a->nsplit, ntk, ntiles, nsplit, rps);
        }
        if (two) {
            const int grid2 = ((nsplit + 7) / 8) * 8 * ntiles;
            float* slab2 = tn_use_slab(a, nsplit) ? a->slab : nullptr;
            constexpr int LDS2 = 8 * G::MT * G::ROWB + 4096;          // four 64-row P + Q buffers
            static bool attr_dma2 = false;
            if (!attr_dma2) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<CT, PSrc, QSrc, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);
                if (e != hipSuccess) return (int)e;
                attr_dma2 = true;
            }
            hipLaunchKernelGGL((gemm_tn_kernel<CT, PSrc, QSrc, true, 2>), dim3(grid2), dim3(NTHREADS * 2), LDS2, st, ps, qs,
                               a->dw, a->lddw, a->db, a->M, a->N, a->K, ntk, ntiles, nsplit, rps, slab2);
            MM_CHECK_LAUNCH();
            return slab2 ? tn_reduce(a, nsplit, st) : 0;
        }
        if (tn_dma_ok(a->M, G::MT)) {
            static bool attr_dma = false;
            if (!attr_dma) {
                hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<CT, PSrc, QSrc, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                if (e != hipSuccess) return (int)e;
                attr_dma = true;
            }
            hipLaunchKernelGGL((gemm_tn_kernel<CT, PSrc, QSrc, true>), dim3(grid), dim3(NTHREADS), LDS, st, ps, qs,
                               a->dw, a->lddw, a->db, a->M, a->N, a->K, ntk, ntiles, nsplit, rps, slab);
            MM_CHECK_LAUNCH();
            return slab ? tn_reduce(a, nsplit, st) : 0;
        }
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel<CT, PSrc, QSrc, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_tn_kernel<CT, PSrc, QSrc, false>), dim3(grid), dim3(NTHREADS), LDS, st, ps, qs,
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
    constexpr int EPC = Mma<CT>::EPC;
    if (a->p_prologue == MMVAE_PRO_BN_BWD_APPLY) {
        if ((a->p_dtype == MMVAE_BF16) != (sizeof(CT) == 2)) return MMVAE_ERR_DTYPE;
        if (!a->p_y || !a->p_mean || !a->p_rstd || !a->p_coef || a->N % EPC || a->ldp % EPC || a->ld_py % EPC ||
            ((uintptr_t)a->p & 15) || ((uintptr_t)a->p_y & 15)) return MMVAE_ERR_ARG;
        if ((long)a->M * a->ld_py * (long)sizeof(CT) >= (1L << 32)) return MMVAE_ERR_ARG;
        SrcBnBwdApply<CT> p{(const CT*)a->p, a->ldp, (const CT*)a->p_y, a->ld_py, a->M, a->N, a->p_mean, a->p_rstd, a->p_coef};
        return tn_dispatch_q<CT>(a, p, st);
    }
    if (a->p_prologue != MMVAE_PRO_NONE) return MMVAE_ERR_ARG;
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

// host side of the grouped launch: classify every problem; -1 = not groupable (the caller launches it alone)
template <typename CT>
static int tn_group_combo(const mmvae_gemm_tn_args* a) {
    constexpr int EPC = Mma<CT>::EPC;
    const int act = sizeof(CT) == 2 ? MMVAE_BF16 : MMVAE_F32;
    if (a->p_prologue != MMVAE_PRO_NONE || a->lddw != a->K || !a->slab) return -1;
    const bool p_f32 = a->p_dtype == MMVAE_F32 && a->ldp % 4 == 0 && a->N % 4 == 0 && ((uintptr_t)a->p & 15) == 0;
    const bool p_act = a->p_dtype == act && a->ldp % EPC == 0 && ((uintptr_t)a->p & 15) == 0 && (sizeof(CT) == 2 || a->N % 4 == 0);
    const bool q_act = a->q_dtype == act && a->ldq % EPC == 0 && ((uintptr_t)a->q & 15) == 0 && (sizeof(CT) == 2 || a->K % 4 == 0);
    if (a->q_prologue == MMVAE_PRO_BN_RELU_DROP) {
        if (!p_f32 || !q_act || a->K > 512 || !a->pro_scale || !a->pro_shift) return -1;
        if (a->pro_mask && (a->ld_pro_mask % 4 || ((uintptr_t)a->pro_mask & 3))) return -1;
        return 0;
    }
    if (a->q_prologue != MMVAE_PRO_NONE || !q_act) return -1;
    if (sizeof(CT) == 4) return p_f32 ? 1 : -1;           // f32 mode: every plain operand is f32
    if (p_act) return 1;
    return p_f32 ? 2 : -1;
}

template <typename CT>
static int launch_tn_group(const mmvae_gemm_tn_args* args, int n, hipStream_t st) {
    typedef TnGeom<CT> G;
    TnGroup g; TnGroupReduce rd;
    g.n = rd.n = n;
    int block = 0, elems = 0;
    // batch splits: ONE resident round for the whole group (2 workgroups per CU), at least 4 batch steps per workgroup
    int tiles_total = 0;
    for (int i = 0; i < n; ++i) tiles_total += ((args[i].N + TILE - 1) / TILE) * ((args[i].K + TILE - 1) / TILE);
    int auto_split = (512 / tiles_total) & ~7;
    if (auto_split < 8) auto_split = 8;
    if (auto_split > MMVAE_TN_GROUP_SPLITS) auto_split = MMVAE_TN_GROUP_SPLITS;
    for (int i = 0; i < n; ++i) {
        const mmvae_gemm_tn_args* a = &args[i];
        TnProblem& r = g.pr[i];
        r.combo = tn_group_combo<CT>(a);
        if (r.combo < 0) return MMVAE_ERR_ARG;
        int want = a->nsplit > 0 ? a->nsplit : auto_split;
        const int max_split = (a->M + 4 * G::MT - 1) / (4 * G::MT);
        if (want > max_split) want = max_split;
        if (want > MMVAE_TN_GROUP_SPLITS) want = MMVAE_TN_GROUP_SPLITS;
        if (want < 1) want = 1;
        tn_split(a->M, a->N, a->K, G::MT, want, r.ntk, r.ntiles, r.nsplit, r.rps);
        if ((long)r.nsplit * a->N * a->K > a->slab_elems) return MMVAE_ERR_ARG;
        r.M = a->M; r.N = a->N; r.K = a->K;
        r.dma = (r.combo == 1 && sizeof(CT) == 2 && tn_dma_ok(a->M, G::MT)) ? 1 : 0;
        r.block0 = block; r.nblocks = ((r.nsplit + 7) / 8) * 8 * r.ntiles; block += r.nblocks;
        r.dW = a->dw; r.ldw = a->lddw; r.db = a->db; r.slab = a->slab;
        r.p = a->p; r.ldp = a->ldp; r.q = a->q; r.ldq = a->ldq;
        r.pro_scale = a->pro_scale; r.pro_shift = a->pro_shift; r.pro_mask = a->pro_mask; r.ld_pro_mask = a->ld_pro_mask; r.pro_inv_keep = a->pro_inv_keep;
        rd.slab[i] = a->slab; rd.dW[i] = a->dw; rd.nsplit[i] = r.nsplit; rd.nk[i] = a->N * a->K; rd.first[i] = elems; elems += a->N * a->K;
    }
    for (int i = n; i <= MMVAE_TN_GROUP_MAX; ++i) rd.first[i] = elems;
    for (int i = n; i < MMVAE_TN_GROUP_MAX; ++i) g.pr[i].block0 = 1 << 30;
    constexpr int LDS = 4 * G::MT * G::ROWB + 4096 + 4096;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_group_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_tn_group_kernel<CT>), dim3(block), dim3(NTHREADS), LDS, st, g);
    MM_CHECK_LAUNCH();
    hipLaunchKernelGGL(tn_group_reduce_kernel, dim3((elems + 255) / 256), dim3(256), 0, st, rd);
    MM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mm

extern "C" int mmvae_gemm_tn_group(const mmvae_gemm_tn_args* args, int32_t n, void* stream) {
    if (!args || n <= 0 || n > MMVAE_TN_GROUP_MAX) return MMVAE_ERR_ARG;
    const long lim = 1L << 32;
    for (int i = 0; i < n; ++i) {
        const mmvae_gemm_tn_args* a = &args[i];
        if (!a->p || !a->q || !a->dw || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->prec != args[0].prec) return MMVAE_ERR_ARG;
        if ((long)a->M * a->ldp * (a->p_dtype == MMVAE_BF16 ? 2 : 4) >= lim || (long)a->M * a->ldq * (a->q_dtype == MMVAE_BF16 ? 2 : 4) >= lim) return MMVAE_ERR_ARG;
        if (a->pro_mask && (long)a->M * a->ld_pro_mask >= lim) return MMVAE_ERR_ARG;
        for (int j = 0; j < i; ++j)            // every problem needs a slab of its own: they run side by side
            if (args[j].slab == a->slab) return MMVAE_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (args[0].prec == MMVAE_PREC_BF16) return mm::launch_tn_group<mm::bf16>(args, n, st);
    if (args[0].prec == MMVAE_PREC_F32) return mm::launch_tn_group<float>(args, n, st);
    return MMVAE_ERR_ARG;
}

namespace mm {
extern long g_block_bytes, g_split_bytes;                       // gemm_nt.hip (mmvae_set_tuning key 3)
int launch_tn_wide(const mmvae_gemm_tn_args* a, hipStream_t st, int* nsplit_out);      // gemm_tn_wide.hip
}
extern "C" int mmvae_gemm_tn(const mmvae_gemm_tn_args* a, void* stream) {
    if (!a || !a->p || !a->q || !a->dw) return MMVAE_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MMVAE_ERR_ARG;
    // The operand sources address P, Q and the prologue mask with 32-bit offsets from a scalar base.  Operands of 4 GiB or more
    // (scaled omics widths) go through in row blocks: the batch rows are the reduction index and dW / db are accumulated.
    const long p_row = (long)a->ldp * (a->p_dtype == MMVAE_BF16 ? 2 : 4), q_row = (long)a->ldq * (a->q_dtype == MMVAE_BF16 ? 2 : 4);
    const long py_row = a->p_prologue ? (long)a->ld_py * (a->prec == MMVAE_PREC_BF16 ? 2 : 4) : 0;
    long row_bytes = p_row > q_row ? p_row : q_row;
    if (py_row > row_bytes) row_bytes = py_row;
    if ((long)a->ld_pro_mask > row_bytes) row_bytes = a->ld_pro_mask;
    if ((long)a->M * row_bytes >= mm::g_split_bytes) {
        if (a->p_prologue != MMVAE_PRO_NONE && !a->p_coef) return MMVAE_ERR_ARG;      // every block would add dgamma / dbeta again: finalise separately
        long rows = mm::g_block_bytes / row_bytes;          // block < split threshold: the recursion below ends after one level
        if (rows <= 0) return MMVAE_ERR_ARG;
        const long nblk = (a->M + rows - 1) / rows;         // equal blocks (see mmvae_gemm_nt)
        long even = (a->M + nblk - 1) / nblk;
        if (even >= 256) even = (even + 255) & ~255L;
        if (even <= rows) rows = even;
        else if (rows >= 256) rows &= ~255L;
        for (long r0 = 0; r0 < a->M; r0 += rows) {
            mmvae_gemm_tn_args s = *a;
            s.M = (int32_t)((a->M - r0 < rows) ? a->M - r0 : rows);
            s.p = (const char*)a->p + r0 * p_row;
            s.q = (const char*)a->q + r0 * q_row;
            if (a->p_y) s.p_y = (const char*)a->p_y + r0 * py_row;
            if (a->pro_mask) s.pro_mask = a->pro_mask + r0 * a->ld_pro_mask;
            const int rc = mmvae_gemm_tn(&s, stream);
            if (rc) return rc;
        }
        return 0;
    }
    hipStream_t st = (hipStream_t)stream;
    {   // the large weight gradients (first encoder layers, last decoder layer): wide tiles, gemm_tn_wide.hip
        int ns = 0;
        const int rc = mm::launch_tn_wide(a, st, &ns);
        if (rc != -100) return rc ? rc : mm::tn_reduce(a, ns, st);
    }
    if (a->prec == MMVAE_PREC_BF16) return mm::tn_dispatch_p<mm::bf16>(a, st);
    if (a->prec == MMVAE_PREC_F32) return mm::tn_dispatch_p<float>(a, st);
    return MMVAE_ERR_ARG;
}
