// NT GEMM, second generation (gfx950):  C[M,N] = epilogue( A[M,K] * W[N,K]^T )  for operands that need no arithmetic on their way
// into the MFMA -- A already in the compute type (bf16 activations / gradients), W the prepared bf16 weights.
//
// What the first-generation kernel (gemm_nt.hip) spends its time on at B = 65 536 (tools/stamp_nt.py):
//   * a tile's first loads are issued when the workgroup starts and its last stores when it ends; every workgroup of the launch
//     runs the same program on the same amount of data, so the whole chip sits in its prologue (HBM saturated, MFMA idle), then
//     in its main loops, then in its epilogues (HBM read path idle) together: 20-25 % of a workgroup's life on either side of
//     an 8-13 step main loop;
//   * operands travel global -> VGPR -> LDS: ~100 VGPRs of look-ahead and a registers -> LDS pass per K step that the MFMAs wait for.
// This kernel:
//   * moves both operands with LDS-DMA (global_load_lds_dwordx4: no VGPR destination, no ds_write pass); the XOR swizzle of the
//     LDS image is applied to the per-lane SOURCE address (the LDS side of a DMA is lane-linear), 8 rows x one full 128-byte
//     line per wave-instruction;
//   * is PERSISTENT: a workgroup walks a list of tiles and treats (tile, K step) as one continuous stream of ring slots, so the
//     DMA of the next tile's first K step is in flight while the current tile's epilogue runs -- no per-tile load prologue;
//   * keeps the first generation's fragment layout (same LDS image, same swapped-operand MFMA order), hence its epilogues
//     (gemm_nt_epi.h) unchanged: BatchNorm statistics, ReLU masks, BatchNorm backward, full-line stores.
// Ring: 2 slots of {A [128][128 B], W [64*WN][128 B]}; per K step: wait for the own DMA of the slot (vmcnt), ONE barrier (every
// wave's DMA has landed and every wave has finished reading the other slot), issue the DMA of the next step into the other slot,
// multiply the current one.
#pragma once
#include "common.h"
#include "gemm_nt_epi.h"

namespace mm {

template <int WN> struct Nt2Lds {
    static constexpr int A_BYTES = TILE * ROW_BYTES;               // 16 KiB
    static constexpr int W_BYTES = 64 * WN * ROW_BYTES;            // 8 KiB * WN
    static constexpr int SLOT = A_BYTES + W_BYTES;
    static constexpr int RING = 2 * SLOT;
    static constexpr int RED = RING;                               // column-sum scratch of STATS epilogues: 4 * 64*WN floats
    static constexpr int ECOL = RED + 4 * 64 * WN * 4;             // per-column constants of the epilogue: 8 * 64*WN floats
    static constexpr int TOTAL = ECOL + 8 * 64 * WN * 4;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// Epilogue of EpiLoss: the 128 x 128 fp32 tile goes through the ring's LDS (64 KB, free once the main loop is over: this
// epilogue's kernel does not start the next tile's DMA early) and leaves it in a ROW-COALESCED pass: thread (row r0 + 8 i,
// chunk c) handles 4 consecutive columns, so target loads and gradient stores are 32 lanes x 16 / 8 bytes along a row.  In
// accumulator layout the 782- and 572-element fp32 target rows (8- / 16-byte aligned) were touched in 64-byte pieces of 16
// different rows per instruction, which cost more than the stream kernel this replaces.  The 16-byte
// chunks of the LDS image are XOR-ed with (row & 7): conflict-free for the accumulator writes (8 rows per lane group) and
// for the row reads.
template <typename Epi, int WN>
__device__ __forceinline__ void nt2_loss_epilogue(unsigned char* smem, const float* ecol, float* red, f32x4 (&acc)[4][4], const Epi& epi,
                                                  int row0, int col0, int M, int N, int tid, int lane, int wr, int wc)
{
    static_assert(WN == 2, "128 x 128 tiles: the fp32 tile is exactly the 64 KB ring");
    const int li = lane & 15, lg = lane >> 4;
    const int c = tid & 31, r0 = tid >> 5;                         // row-coalesced pass: 8 rows per pass, 16 passes, 4 columns per thread
    const int colg = col0 + 4 * c;
    const float* __restrict__ T = epi.T;
    bf16* __restrict__ G = epi.G;
    // ALL 16 target vectors of the thread are requested first (64 registers -- the accumulators die in the LDS image below): with 4
    // in flight per thread the pass ran at the latency-bound 3.6 TB/s of 32 KB in flight per CU; the LDS transposition now covers
    // their trip.  Branch-free clamped addresses (rows past M, columns past N are masked when used).
    // Interior tiles (all but the last row / column tiles): no clamps on the target addresses, no bounds selects on the terms.
    const bool interior = row0 + TILE <= M && col0 + 128 <= N;                      // workgroup-uniform
    float t[16][4];
    if (interior) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* tp = T + (long)(row0 + r0 + 8 * i) * epi.ldt + colg;
#pragma unroll
            for (int e = 0; e < 4; e += Epi::VT) VLoad<float, Epi::VT>::ld(tp + e, &t[i][e]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* tp = T + (long)min(row0 + r0 + 8 * i, M - 1) * epi.ldt;
#pragma unroll
            for (int e = 0; e < 4; e += Epi::VT) VLoad<float, Epi::VT>::ld(tp + min(colg + e, N - Epi::VT), &t[i][e]);
        }
    }
    __syncthreads();                                               // every wave has finished reading the ring
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = wr * 64 + m * 16 + li, ch = 16 * wc + 4 * n + lg;
            *(f32x4*)(smem + row * 512 + ((ch ^ (row & 7)) << 4)) = acc[m][n];
        }
    __syncthreads();
    const int gcols = (int)min((long)((N + 7) & ~7), epi.ldg);      // gradient columns that exist (pads are written as zeros)
    const unsigned char* src = smem + r0 * 512 + ((c ^ (r0 & 7)) << 4);     // (r0 + 8 i) & 7 == r0 & 7
    const f32x2 b01 = {ecol[4 * c], ecol[4 * c + 1]}, b23 = {ecol[4 * c + 2], ecol[4 * c + 3]};
    f32x2 lsum2 = {0.f, 0.f};
    if (interior) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 z = *(const f32x4*)(src + i * 8 * 512);
            f32x2 g01, g23;
            lsum2 += epi.term2(f32x2{z[0], z[1]} + b01, f32x2{t[i][0], t[i][1]}, g01);
            lsum2 += epi.term2(f32x2{z[2], z[3]} + b23, f32x2{t[i][2], t[i][3]}, g23);
            const bf16x4 o = {(bf16)g01[0], (bf16)g01[1], (bf16)g23[0], (bf16)g23[1]};
            *(bf16x4*)(G + (long)(row0 + r0 + 8 * i) * epi.ldg + colg) = o;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int grow = row0 + r0 + 8 * i;
            const f32x4 z = *(const f32x4*)(src + i * 8 * 512);
            f32x2 g01, g23;
            const f32x2 l01 = epi.term2(f32x2{z[0], z[1]} + b01, f32x2{t[i][0], t[i][1]}, g01);
            const f32x2 l23 = epi.term2(f32x2{z[2], z[3]} + b23, f32x2{t[i][2], t[i][3]}, g23);
            const float l[4] = {l01[0], l01[1], l23[0], l23[1]}, ge[4] = {g01[0], g01[1], g23[0], g23[1]};
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = grow < M && colg + e < N;
                lsum2[e & 1] += ok ? l[e] : 0.f;
                g[e] = ok ? ge[e] : 0.f;
            }
            if (grow < M && colg < gcols) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)g[e];
                *(bf16x4*)(G + (long)grow * epi.ldg + colg) = o;
            }
        }
    }
    const float lsum = lsum2[0] + lsum2[1];
    const float ws = wave_sum(lsum);
    if (lane == 0) red[tid >> 6] = ws;
    __syncthreads();                                               // also: everybody is done with the LDS tile
    if (tid == 0) {
        const double v = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
        if (v != 0.0) unsafeAtomicAdd(epi.sum, v);
    }
}

// ReLU-mask dX epilogue in the row-coalesced form (EpiReluMaskStream): 8 bytes of the saved activation per thread and pass, one select
// per element, 8-byte stores along rows.
template <typename Epi, int WN>
__device__ __forceinline__ void nt2_relumask_epilogue(unsigned char* smem, f32x4 (&acc)[4][4], const Epi& epi,
                                                      int row0, int col0, int M, int N, int tid, int lane, int wr, int wc)
{
    static_assert(WN == 2, "128 x 128 tiles: the fp32 tile is exactly the 64 KB ring");
    const int li = lane & 15, lg = lane >> 4;
    const int c = tid & 31, r0 = tid >> 5;
    const int colg = col0 + 4 * c;
    const bf16* __restrict__ H = epi.H;
    bf16* __restrict__ C = epi.C;
    const int ch_ = min(colg, ((N + 7) & ~7) - 4);                   // the saved activation has ceil8(N) columns (H may be a column slice: not ldh)
    bf16x4 hv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[i] = *(const bf16x4*)(H + (long)min(row0 + r0 + 8 * i, M - 1) * epi.ldh + ch_);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = wr * 64 + m * 16 + li, ch = 16 * wc + 4 * n + lg;
            *(f32x4*)(smem + row * 512 + ((ch ^ (row & 7)) << 4)) = acc[m][n];
        }
    __syncthreads();
    const int ccols = (int)min((long)((N + 7) & ~7), epi.ldc);
    const unsigned char* src = smem + r0 * 512 + ((c ^ (r0 & 7)) << 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int grow = row0 + r0 + 8 * i;
        const f32x4 z = *(const f32x4*)(src + i * 8 * 512);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)((colg + e < N && (float)hv[i][e] > 0.f) ? z[e] : 0.f);
        if (grow < M && colg < ccols) *(bf16x4*)(C + (long)grow * epi.ldc + colg) = o;
    }
    __syncthreads();                                               // the ring may be refilled
}

// BatchNorm-backward dX epilogue in the same row-coalesced form (EpiBnBwdStream): thread (row r0 + 8 i, 4 columns) loads 8 bytes of
// the saved pre-BN output y and 4 mask bytes per pass, computes d = acc * keep * (y*scale + shift > 0) and xhat, keeps its four
// columns' partial sums of d and d*xhat in registers over its 16 rows (no butterfly), stores bf16 d; the 8 row groups are added
// through LDS and each column's two sums go out as f64 atomics.  The accumulator-layout form (gemm_nt_epi.h) ran EncoderB.L0.dX at
// 2.2 TB/s on its 200 MB.
template <typename Epi, int WN>
__device__ __forceinline__ void nt2_bnbwd_epilogue(unsigned char* smem, const float* ecol, float* red, f32x4 (&acc)[4][4], const Epi& epi,
                                                   int row0, int col0, int M, int N, int tid, int lane, int wr, int wc)
{
    static_assert(WN == 2, "128 x 128 tiles: the fp32 tile is exactly the 64 KB ring");
    constexpr int BN = 128;
    const int li = lane & 15, lg = lane >> 4;
    const int c = tid & 31, r0 = tid >> 5;
    const int colg = col0 + 4 * c;
    const bf16* __restrict__ Y = epi.Y;
    const uint8_t* __restrict__ K_ = epi.mask;
    bf16* __restrict__ C = epi.C;
    const bool has_mask = K_ != nullptr;
    const int cy = min(colg, ((N + 7) & ~7) - 4);                    // y rows are padded to 8 columns: a 4-column load below ceil8(N) is in bounds
    bf16x4 yv[16]; uint32_t mv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long grow = min(row0 + r0 + 8 * i, M - 1);
        yv[i] = *(const bf16x4*)(Y + grow * epi.ldy + cy);
        mv[i] = has_mask ? *(const uint32_t*)(K_ + grow * epi.ldm + min(colg, ((N + 3) & ~3) - 4)) : 0x01010101u;
    }
    __syncthreads();                                               // every wave has finished reading the ring
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = wr * 64 + m * 16 + li, ch = 16 * wc + 4 * n + lg;
            *(f32x4*)(smem + row * 512 + ((ch ^ (row & 7)) << 4)) = acc[m][n];
        }
    __syncthreads();
    const int ccols = (int)min((long)((N + 7) & ~7), epi.ldc);
    const unsigned char* src = smem + r0 * 512 + ((c ^ (r0 & 7)) << 4);
    float sc[4], sh[4], mu[4], rs[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[e] = ecol[4 * c + e]; sh[e] = ecol[BN + 4 * c + e]; mu[e] = ecol[2 * BN + 4 * c + e]; rs[e] = ecol[3 * BN + 4 * c + e]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int grow = row0 + r0 + 8 * i;
        const f32x4 z = *(const f32x4*)(src + i * 8 * 512);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = grow < M && colg + e < N;
            const float y = (float)yv[i][e];
            const float keep = ((mv[i] >> (8 * e)) & 0xffu) ? epi.inv_keep : 0.f;
            const float d = (ok && y * sc[e] + sh[e] > 0.f) ? z[e] * (has_mask ? keep : 1.f) : 0.f;
            const float xh = (y - mu[e]) * rs[e];
            s1[e] += d; s2[e] += d * xh;
            o[e] = (bf16)d;
        }
        if (grow < M && colg < ccols) *(bf16x4*)(C + (long)grow * epi.ldc + colg) = o;
    }
    __syncthreads();                                               // everybody is done with the LDS tile: reuse it for the column sums
    float* part = (float*)smem;                                    // [8 row groups][2][128 columns]
#pragma unroll
    for (int e = 0; e < 4; ++e) { part[(r0 * 2 + 0) * BN + 4 * c + e] = s1[e]; part[(r0 * 2 + 1) * BN + 4 * c + e] = s2[e]; }
    __syncthreads();
    if (tid < 2 * BN) {
        const int which = tid >> 7, col = tid & (BN - 1);
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += part[(g * 2 + which) * BN + col];
        if (col0 + col < N) unsafeAtomicAdd((which ? epi.stat2 : epi.stat1) + col0 + col, (double)v);
    }
    __syncthreads();                                               // the ring may be refilled
}

template <typename Epi, int WN>
__global__ __launch_bounds__(128 * WN, 2)
void gemm_nt2_kernel(const bf16* __restrict__ A, long lda, const bf16* __restrict__ W, long ldw, int M, int N, int K, int gx, int gy, Epi epi)
{
    typedef bf16 CT;
    typedef Nt2Lds<WN> LD;
    constexpr int BK = 64, BN = 64 * WN, NW = 2 * WN;
    constexpr int A_PIECES = TILE / 8, W_PIECES = BN / 8;                        // 1 KiB pieces (8 rows x 128 B) per slot
    constexpr int A_PER = A_PIECES / NW, W_PER = W_PIECES / NW;
    typedef Mma<CT>::frag frag;
    typedef EpiCols<sizeof(typename Epi::out_t) == 2> EC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)(smem + LD::RED);
    float* ecol = (float*)(smem + LD::ECOL);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WN, wc = wid % WN;
    const int nk = (K + BK - 1) / BK;
    const int kmax = ((K + 7) & ~7) - 8;                           // last 16-byte chunk that lies inside an A row (rows are padded to 8 elements)
    const int ntiles = ((gx + 7) / 8) * 8 * gy;                    // tile ids incl. the padding of gx to a multiple of 8

    // tile id T -> (row tile, column tile): ids that differ by 8 run on one XCD; the column tiles of a row tile are adjacent there
    auto tile_rc = [&](int T, int& rt, int& ct) { const int slot = T >> 3; ct = slot % gy; rt = (slot / gy) * 8 + (T & 7); };
    auto next_tile = [&](int T) {
        for (T += gridDim.x; T < ntiles; T += gridDim.x) { int rt, ct; tile_rc(T, rt, ct); if (rt < gx) return T; }
        return -1;
    };
    int T = blockIdx.x;
    { int rt, ct; tile_rc(T, rt, ct); if (rt >= gx) T = next_tile(T); }
    if (T < 0) return;
    // DMA of one ring slot: lane l of a piece writes LDS (row = 8 p + l / 8, position l % 8) and reads chunk (position ^ (row & 7))
    // of that row: the LDS image is the first generation's swz() image.
    const int prow = lane >> 3, ppos = lane & 7;
    auto issue = [&](int tile, int kt, int slot) {
        int rt, ct; tile_rc(tile, rt, ct);
        const int row0 = rt * TILE, col0 = ct * BN;
        unsigned char* sA = smem + slot * LD::SLOT;
        unsigned char* sW = sA + LD::A_BYTES;
        const bf16* Ak = A + kt * BK;                               // wave-uniform part of the address
        const bf16* Wk = W + kt * BK;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int p = wid + NW * i;
            const int r = p * 8 + prow;
            const int c = ppos ^ (r & 7);
            const int kc = min(c * 8, kmax - kt * BK);               // chunks past the row end re-read its last chunk (x zero weights)
            const unsigned off = (unsigned)min(row0 + r, M - 1) * (unsigned)lda + (unsigned)kc;
            __builtin_amdgcn_global_load_lds((gbl_void*)(Ak + off), (lds_void*)(sA + p * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int p = wid + NW * i, r = p * 8 + prow;
            const int c = ppos ^ (r & 7);
            const int wrw = (r & ~63) + EC::wrow(r & 63);           // the epilogue's column order inside a wave's 64 columns (gemm_nt_epi.h)
            const unsigned off = (unsigned)(col0 + wrw) * (unsigned)ldw + (unsigned)(c * 8);
            __builtin_amdgcn_global_load_lds((gbl_void*)(Wk + off), (lds_void*)(sW + p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
    frag f0a[4], f0b[4], f1a[4], f1b[4];
    auto rd = [&](frag (&af)[4], frag (&bf)[4], int slot, int s) {
        const unsigned char* sA = smem + slot * LD::SLOT;
        const unsigned char* sW = sA + LD::A_BYTES;
        const int ch = s * 4 + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = wr * 64 + m * 16 + (lane & 15);
            af[m] = *(const frag*)(sA + r * ROW_BYTES + ((ch ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int r = wc * 64 + n * 16 + (lane & 15); bf[n] = *(const frag*)(sW + r * ROW_BYTES + ((ch ^ (r & 7)) << 4)); }
    };
    auto mma = [&](const frag (&af)[4], const frag (&bf)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<CT>::mma(acc[m][n], bf[n], af[m]);     // swapped operands: transposed accumulator (gemm_nt_epi.h)
    };

#ifdef MM_STAMP
    // diagnostic build (make STAMP=1): cycles per K step of {wait for the own DMA, barrier, DMA issue, fragment reads + MFMA},
    // K steps, waves, whole kernel, epilogues -- into the first generation's mm_stamps[] (tools/stamp_nt.py NT2=1)
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define NT2_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define NT2_T(x)
#endif
    issue(T, 0, 0);
    int g = 0;                                                      // ring position: slot = g & 1, continuous across tiles
    for (;;) {
        const int Tn = next_tile(T);
        int rt, ct; tile_rc(T, rt, ct);
        const int row0 = rt * TILE, col0 = ct * BN;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        EpiOperands<Epi> eops;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // own DMA of this slot has landed (the only vector-memory operations in flight); after the barrier everybody's has,
            // and nobody still reads the other slot
            NT2_T(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NT2_T(t1);
            __syncthreads();
            NT2_T(t2);
            if (kt == 0) nt_epilogue_fill_cols<Epi, WN>(ecol, epi, col0, N, tid);     // previous tile's epilogue is over; visible after the next barrier
            // DMA of the next step first, then this step's fragment reads and MFMAs.  (Interleaving the 8 DMA pieces with the 32 MFMAs
            // -- unconditional issue in one basic block + sched_group_barrier 4 : 1 -- was measured: no gain, 5 % slower at K = 1024.)
            if (kt + 1 < nk) issue(T, kt + 1, (g + 1) & 1);
            else if (Tn >= 0 && !Epi::LDS_STREAM) issue(Tn, 0, (g + 1) & 1);     // the next tile's first slot flies under this tile's epilogue
            NT2_T(t3);
            rd(f0a, f0b, g & 1, 0);
            rd(f1a, f1b, g & 1, 1);
            mma(f0a, f0b);
            mma(f1a, f1b);
#ifdef MM_STAMP
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[3][3][3]));      // the MFMAs of this step are issued before the stamp
            NT2_T(t4);
            st_acc[0] += t1 - t0; st_acc[1] += t2 - t1; st_acc[2] += t3 - t2; st_acc[3] += t4 - t3; st_acc[4] += 1;
#endif
        }
#ifdef MM_STAMP
        NT2_T(te0);
#endif
        if (nk == 1) __syncthreads();                               // the column constants were written after this tile's only barrier
        // epilogue operands (saved activation, keep mask): fetched here, not a K step early as the first generation does -- 48
        // more live registers across the last MFMAs spilled, and the co-resident workgroup covers the latency
        if constexpr (Epi::LDS_STREAM) {
            if constexpr (Epi::MODE == 2) nt2_bnbwd_epilogue<Epi, WN>(smem, ecol, red, acc, epi, row0, col0, M, N, tid, lane, wr, wc);
            else if constexpr (Epi::MODE == 3) nt2_relumask_epilogue<Epi, WN>(smem, acc, epi, row0, col0, M, N, tid, lane, wr, wc);
            else nt2_loss_epilogue<Epi, WN>(smem, ecol, red, acc, epi, row0, col0, M, N, tid, lane, wr, wc);
            if (Tn >= 0) issue(Tn, 0, g & 1);                       // the ring is free again (the epilogue ends in a barrier)
        } else {
            nt_epilogue_prefetch<Epi, 0>(eops, epi, row0, col0, M, N, BN, lane, wr, wc);
            nt_epilogue<CT, Epi, WN>(red, ecol, acc, epi, eops, row0, col0, M, N, tid, lane, wr, wc);
        }
#ifdef MM_STAMP
        NT2_T(te1);
        st_acc[5] += te1 - te0;
#endif
        if (Tn < 0) break;
        T = Tn;
    }
#ifdef MM_STAMP
    if (tid == 0 && (blockIdx.x & 15) == 3) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        for (int i = 0; i < 5; ++i) atomicAdd(&mm_stamps[i], st_acc[i]);
        atomicAdd(&mm_stamps[5], 1ull);
        atomicAdd(&mm_stamps[6], t_end - t_begin);
        atomicAdd(&mm_stamps[9], st_acc[5]);
    }
#endif
}

// Persistent grid: every CU gets its residency's worth of workgroups (2 of 4 waves, or 1 of 8), a multiple of 8 so that a
// workgroup keeps to one XCD's tile list.
template <typename Epi, int WN>
static int launch_nt2(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    typedef Nt2Lds<WN> LD;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt2_kernel<Epi, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, LD::TOTAL);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int gx = (M + TILE - 1) / TILE, gy = (N + 64 * WN - 1) / (64 * WN);
    const int ntiles = ((gx + 7) / 8) * 8 * gy;
    static const int wg_env = getenv("MMVAE_NT2_WGS") ? atoi(getenv("MMVAE_NT2_WGS")) : 0;      // A/B knob: workgroups per CU
    const int per_cu = wg_env > 0 ? wg_env : (WN == 2 ? 2 : 1);
    int grid = 256 * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((gemm_nt2_kernel<Epi, WN>), dim3(grid), dim3(128 * WN), LD::TOTAL, st,
                       (const bf16*)A, lda, (const bf16*)W, ldw, M, N, K, gx, gy, epi);
    MM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mm
