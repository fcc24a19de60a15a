// NT GEMM, 256 x 256 tiles (gfx950):  C[M,N] = epilogue( A[M,K] * W[N,K]^T ), A and W already in the compute type (bf16).
//
// Why a third tile shape: the ablation of the second-generation kernel (gemm_nt2.h, MMVAE_NT2_ABLATE) shows
// that the 128-wide kernels of this path are bound by the bytes a CU can INGEST (L2 -> LDS) and store, not by HBM and not by the
// MFMA pipe: operand DMA alone, MFMA alone and the store epilogue alone each take about half of DecoderB.L2.fwd, and memory
// traffic of different phases does not overlap (DMA and stores share the CU's vector-memory path, ~60 GB/s per CU sustained).
// A 128 x 128 tile ingests 32 KiB per 2.1 MFLOP; 256 x 256 ingests 64 KiB per 8.4 MFLOP -- half the L2->CU traffic per FLOP.
//
// Structure: 512 threads = 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = TWO of the 64 x 64 register tiles the epilogues
// (gemm_nt_epi.h) are written for; K steps of 32 (64-byte LDS rows) so that the ring can hold FOUR slots of {A [256][64 B],
// W [256][64 B]} = 128 KiB with three of them in flight (one slot in flight left the DMA latency-bound at one workgroup per
// CU); counted s_waitcnt vmcnt(8) + raw s_barrier per K step (a __syncthreads() would drain the DMA ring); persistent over tiles
// with the next tile's first slots in flight under the epilogue (gemm_nt2.h).
#pragma once
#include "common.h"
#include "gemm_nt_epi.h"

namespace mm {

struct Nt3 {
    static constexpr int BM = 256, BN = 256, BK = 32, ROWB = 64, NWAVE = 8, NTH = 512, SLOTS = 4;
    static constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, SLOT = A_BYTES + W_BYTES;      // 32 KiB
    static constexpr int RING = SLOTS * SLOT;                     // 128 KiB
    static constexpr int RED = RING;                              // column-sum scratch: 4 wave rows x 2 sums x BN floats = 8 KiB
    static constexpr int ECOL = RED + 8 * BN * 4;                 // per-column constants of the epilogue: 8 x BN floats = 8 KiB
    static constexpr int TOTAL = ECOL + 8 * BN * 4;               // 144 KiB: one workgroup per CU
    // chunk swizzle of a 64-byte row (4 chunks of 16 B): chunk c of row r sits at position c ^ g[(r >> 2) & 3]; with this g every
    // 16-lane group of a ds_read_b128 fragment read (rows base + 0..15, one chunk index per lane group) hits 16 distinct slots
    static __device__ __forceinline__ int g(int r) { return (0x1230 >> (((r >> 2) & 3) * 4)) & 3; }      // {0, 3, 2, 1}
};

template <typename Epi>
__global__ __launch_bounds__(512, 2)
void gemm_nt3_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ W, long ldw, int w_rows, int M, int N, int K, int gx, int gy, Epi epi)
{
    typedef bf16 CT;
    typedef Nt3 G;
    typedef Mma<CT>::frag frag;
    typedef EpiCols<sizeof(typename Epi::out_t) == 2> EC;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)(smem + G::RED);
    float* ecol = (float*)(smem + G::ECOL);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr2 = wid >> 2, wc = wid & 3;                        // wave: rows [128 wr2, +128), columns [64 wc, +64)
    const int nk = (K + G::BK - 1) / G::BK;
    const int kmax = ((K + 7) & ~7) - 8;
    const int ntiles = ((gx + 7) / 8) * 8 * gy;
    auto tile_rc = [&](int T, int& rt, int& ct) { const int slot = T >> 3; ct = slot % gy; rt = (slot / gy) * 8 + (T & 7); };
    auto next_tile = [&](int T) {
        for (T += gridDim.x; T < ntiles; T += gridDim.x) { int rt, ct; tile_rc(T, rt, ct); if (rt < gx) return T; }
        return -1;
    };
    int T = blockIdx.x;
    { int rt, ct; tile_rc(T, rt, ct); if (rt >= gx) T = next_tile(T); }
    if (T < 0) return;

    // one ring slot = 16 + 16 pieces of 1 KiB (16 rows x 64 B); 4 pieces per wave: 2 of A, 2 of W
    const int prow = lane >> 2, ppos = lane & 3;
    auto issue = [&](int tile, int kt, int slot) {
        int rt, ct; tile_rc(tile, rt, ct);
        const int row0 = rt * G::BM, col0 = ct * G::BN;
        unsigned char* sA = smem + slot * G::SLOT;
        unsigned char* sW = sA + G::A_BYTES;
        const bf16* Ak = A + kt * G::BK;
        const bf16* Wk = W + kt * G::BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wid + G::NWAVE * i, r = p * 16 + prow;
            const int c = ppos ^ G::g(r);
            const int kc = min(c * 8, kmax - kt * G::BK);          // chunks past the padded row end re-read its last chunk (x zero weights)
            const unsigned offa = (unsigned)min(row0 + r, M - 1) * (unsigned)lda + (unsigned)kc;
            lds_dma16(Ak + offa, lds_addr_of(sA + p * 1024));
            const int wrw = (r & ~63) + EC::wrow(r & 63);           // the epilogue's column order inside a wave's 64 columns
            const unsigned offw = (unsigned)min(col0 + wrw, w_rows - 1) * (unsigned)ldw + (unsigned)(c * 8);      // W has ceil128(N) rows
            lds_dma16(Wk + offw, lds_addr_of(sW + p * 1024));
        }
    };

    f32x4 acc[8][4];
    auto compute = [&](int slot) {
        const unsigned char* sA = smem + slot * G::SLOT;
        const unsigned char* sW = sA + G::A_BYTES;
        const int ch = lane >> 4;
        frag bf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int r = wc * 64 + n * 16 + (lane & 15); bf[n] = *(const frag*)(sW + r * G::ROWB + ((ch ^ G::g(r)) << 4)); }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            frag af[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { const int r = wr2 * 128 + h * 64 + m * 16 + (lane & 15); af[m] = *(const frag*)(sA + r * G::ROWB + ((ch ^ G::g(r)) << 4)); }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<CT>::mma(acc[h * 4 + m][n], bf[n], af[m]);      // swapped operands: transposed accumulator
        }
    };

    // the ring runs continuously across tiles: step s (global) uses slot s & 3; steps s+1 .. s+3 are in flight
    struct Cur { int tile, kt; };
    auto advance = [&](Cur c) { if (c.kt + 1 < nk) return Cur{c.tile, c.kt + 1}; return Cur{c.tile >= 0 ? next_tile(c.tile) : -1, 0}; };
    Cur head{T, 0};                                                 // next step to ISSUE
    int issued = 0;
    for (; issued < G::SLOTS - 1 && head.tile >= 0; ++issued) { issue(head.tile, head.kt, issued & 3); head = advance(head); }
    int s = 0;                                                      // global step being computed
    for (;;) {
        const int Tn = next_tile(T);
        int rt, ct; tile_rc(T, rt, ct);
        const int row0 = rt * G::BM, col0 = ct * G::BN;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++s) {
            // own pieces of step s have landed when at most the pieces of the (issued - s - 1) later steps are outstanding -- unless
            // this tile's predecessor just stored its epilogue (stores share the counter): then drain (once per tile)
            const int later = issued - s - 1;
            if (kt == 0 || later <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // everybody's pieces of step s are in LDS; nobody still reads slot (s - 1) & 3
            if (kt == 0) nt_epilogue_fill_cols<Epi, 4>(ecol, epi, col0, N, tid);      // previous epilogue is over; visible after the next barrier
            if (head.tile >= 0) { issue(head.tile, head.kt, issued & 3); head = advance(head); ++issued; }
            compute(s & 3);
        }
        if (nk == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        // epilogue: the wave's 128 x 64 tile as two 64 x 64 halves, virtual wave row v = 2 wr2 + h of a 256-row tile
        const bool want_stats = Epi::STATS && (epi.stat1 != nullptr || epi.stat2 != nullptr);
        const bool vec = (Epi::NEED < 1 || epi_h_vec(epi)) && (Epi::NEED < 2 || epi.mask == nullptr || epi_m_vec(epi));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = wr2 * 2 + h;
            f32x4 (&ah)[4][4] = *reinterpret_cast<f32x4 (*)[4][4]>(&acc[h * 4]);
            EpiOperands<Epi> eops;
            if (!vec) {
                nt_epilogue_body<Epi, -1, false, false>(ecol, ah, epi, eops, red, want_stats, G::BN, row0, col0, M, N, lane, v, wc, G::BM);
            } else {
                nt_epilogue_prefetch<Epi, 0>(eops, epi, row0, col0, M, G::BN, lane, v, wc);
                nt_epilogue_prefetch<Epi, 1>(eops, epi, row0, col0, M, G::BN, lane, v, wc);
                const int a = epi.act_code();
                if (a == 0) nt_epilogue_body<Epi, 0, false, true>(ecol, ah, epi, eops, red, want_stats, G::BN, row0, col0, M, N, lane, v, wc, G::BM);
                else if (a == 1) nt_epilogue_body<Epi, 1, false, true>(ecol, ah, epi, eops, red, want_stats, G::BN, row0, col0, M, N, lane, v, wc, G::BM);
                else nt_epilogue_body<Epi, 2, false, true>(ecol, ah, epi, eops, red, want_stats, G::BN, row0, col0, M, N, lane, v, wc, G::BM);
            }
        }
        if (want_stats) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < G::BN && col0 + tid < N) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) { s1 += red[(2 * v) * G::BN + tid]; s2 += red[(2 * v + 1) * G::BN + tid]; }
                if (epi.stat1) unsafeAtomicAdd(epi.stat1 + col0 + tid, (double)s1);
                if (epi.stat2) unsafeAtomicAdd(epi.stat2 + col0 + tid, (double)s2);
            }
        }
        if (Tn < 0) break;
        T = Tn;
    }
}

template <typename Epi>
static int launch_nt3(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt3_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, Nt3::TOTAL);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int gx = (M + Nt3::BM - 1) / Nt3::BM, gy = (N + Nt3::BN - 1) / Nt3::BN;
    const int ntiles = ((gx + 7) / 8) * 8 * gy;
    int grid = 256;
    if (grid > ntiles) grid = ntiles;
    const int w_rows = (N + TILE - 1) / TILE * TILE;               // rows of the prepared operand
    hipLaunchKernelGGL((gemm_nt3_kernel<Epi>), dim3(grid), dim3(512), Nt3::TOTAL, st,
                       (const bf16*)A, lda, (const bf16*)W, ldw, w_rows, M, N, K, gx, gy, epi);
    MM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mm
