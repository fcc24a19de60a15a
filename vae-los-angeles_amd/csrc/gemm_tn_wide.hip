// Wide-tile TN GEMM for the LARGE weight gradients of the step (gfx950, bf16 MFMA):
//     dW[N,K] += P[M,N]^T * Q[M,K],   db[N] += column sums of P
// for the first encoder layers: P = BatchNorm-backward-corrected dY built from the bf16 d and y tensors, Q = the fp32 input batch
// (EncoderB.L0.dW 512 x 572, EncoderA.L0.dW 128 x 782).  Reference op: the autograd mm of nn.Linear's backward behind
// optimize_hyperparameters.py:112 (layers encoders.py:13,31).
//
// Why a second kernel next to gemm_tn.hip's 128 x 128 tiles.  With M = 65 536 batch rows and a 1 MB output, the batch is split
// over all 256 CUs and every workgroup streams its rows of P and Q once PER OUTPUT TILE: 128 x 128 tiles re-ingest P five times
// and Q four times for the 512 x 572 gradient (1.27 GB through the L2 -> CU path, which moves ~60 GB/s per CU: 83 us before any
// arithmetic), and repeat the BatchNorm correction of P and the fp32 -> bf16 conversion of Q as often.  Here a workgroup of
// 8 waves owns a 256 x 288 (or 128 x 448) tile: 2 x 2 tiles for that gradient, P re-read twice and Q twice (0.57 GB),
// 36 MFMAs per wave and batch step on 13 transposed fragments.
//
// Two forms.  gemm_tnw_dma_kernel (256 x 288 tiles, 16-byte aligned fp32 rows; also 128 x 448 with a plain bf16 P -- very wide inputs,
// where the BatchNorm correction runs as a pass of its own): raw tiles by LDS-DMA, everything else at fragment time (below).  gemm_tnw_kernel (128 x 448 tiles; EncoderA.L0: fp32 rows only 8-byte aligned): global -> VGPR -> correction /
// conversion -> LDS with one register set, double-buffered LDS, one barrier per step.  Plain bf16 x bf16 problems (the decoders'
// gradients) were measured on these tiles too and stay with gemm_tn.hip's 128 x 128 LDS-DMA form (57-59 against 58 us, larger reduce).
//
// Layout: a batch step is 32 rows (one 16x16x32 MFMA reduction).  Tiles sit row-major in LDS, rows padded to a multiple of 256 bytes,
// each 256-byte panel (128 columns) with gemm_tn.hip's XOR swizzle of its 32-byte units, so the transposed fragments come from
// ds_read_b64_tr_b16 without bank conflicts.  Partial tiles of the batch splits go to the slab workspace and are summed by
// gemm_tn.hip's tn_reduce_kernel (fixed order up to 128 splits).
#include "common.h"
#include "mmvae_hip.h"

namespace mm {

struct TnwArgs {
    const bf16* p; unsigned ldp; const bf16* py; unsigned ldpy;
    const float* mean; const float* rstd; const float* coef;
    // coef == nullptr: mmvae_bn_bwd_finalize folded in -- the constants from the f64 sums, dgamma / dbeta by the (zz == 0, tk == 0) workgroups
    const double* sum_d; const double* sum_dx; const float* gamma; float* dgamma; float* dbeta; int eval_mode;
    const void* q; unsigned ldq;
    int M, N, K, ntk, ntiles, nsplit, rps, linear;
    float* slab; float* db;
};

// the three BatchNorm-backward constants of column `col` (mmvae_bn_bwd_finalize's coef rows), stored or formed from the sums
__device__ __forceinline__ void tnw_coef(const TnwArgs& a, int col, float& k0, float& k1, float& k2) {
    if (a.coef) { k0 = a.coef[col]; k1 = a.coef[a.N + col]; k2 = a.coef[2 * a.N + col]; return; }
    k0 = a.gamma[col] * a.rstd[col];
    k1 = a.eval_mode ? 0.f : (float)(a.sum_d[col] / a.M);
    k2 = a.eval_mode ? 0.f : (float)(a.sum_dx[col] / a.M);
}
// dgamma += sum d * xhat, dbeta += sum d: once per column, by the first batch split of the first K tile of every N tile
__device__ __forceinline__ void tnw_bn_grads(const TnwArgs& a, int n0, int nt, int zz, int tk, int tid, int nthreads) {
    if (a.coef || zz != 0 || tk != 0) return;
    for (int c = tid; c < nt; c += nthreads) {
        const int col = n0 + c;
        if (col < a.N) { a.dbeta[col] += (float)a.sum_d[col]; a.dgamma[col] += (float)a.sum_dx[col]; }
    }
}

template <int WN_, int PA_, int WK_, int QB_> struct TnwCfg {
    static constexpr int WN = WN_, PA = PA_, WK = WK_, QB = QB_;
    static constexpr int NT = WN * PA * 16, KT = WK * QB * 16;
    static constexpr int PROW = (NT * 2 + 255) / 256 * 256, QROW = (KT * 2 + 255) / 256 * 256;
    static constexpr int PCH = NT / 8, QCH = KT / 8;                 // 16-byte bf16 chunks per tile row
    static constexpr int MT = 32, THREADS = 512;
    static constexpr int PI = (MT * PCH + THREADS - 1) / THREADS, QI = (MT * QCH + THREADS - 1) / THREADS;
    static constexpr int BUF = MT * (PROW + QROW);
    static constexpr int LDS = 2 * BUF + 4 * NT * 4;
    static_assert(WN * WK == 8, "8 waves");
};

__device__ __forceinline__ int tnw_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
// byte offset of 16-byte chunk ch (8 columns) of row r
__device__ __forceinline__ int tnw_chunk_off(int r, int ch, int rowb) {
    const int cin = ch & 15;
    return r * rowb + (ch >> 4) * 256 + (((((cin >> 1) ^ tnw_f(r)) << 1) | (cin & 1)) << 4);
}
__device__ __forceinline__ bf16x8 tnw_frag(const unsigned char* tile, int rowb, int colbase, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r = 8 * g + q;
    const int c32 = colbase >> 4;
    const int off0 = r * rowb + (c32 >> 3) * 256 + (((c32 & 7) ^ tnw_f(r)) << 5) + (p << 3);
    const int off1 = off0 + 4 * rowb;               // rows +4: the swizzle key is unchanged
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

#ifdef MM_STAMP
// Diagnostic build only (make STAMP=1, tools/stamp_tnw.py): cycles of {own-DMA wait, barrier, DMA issue, fragments + MFMA}
// summed over the steps of wave 0 of every 8th workgroup, steps, workgroups, whole-kernel cycles, epilogue cycles.
__device__ unsigned long long mm_stamps_tnw[12];
#define MW_T(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define MW_T(x)
#endif

template <typename QT, int VEC> struct TnwQRaw;
template <> struct TnwQRaw<bf16, 8> { bf16x8 v; };
template <int VEC> struct TnwQRaw<float, VEC> { float v[8]; };

template <class C, int PMODE, typename QT, int QVEC>
__global__ __launch_bounds__(512)
void gemm_tnw_kernel(const TnwArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* aux = (float*)(smem + 2 * C::BUF);            // [4][NT]: mean, c0, c0 c2 rstd, c0 c1 of this tile's P columns
    const int L = (int)blockIdx.x;
    int tile, zz;
    if (a.linear) { tile = L % a.ntiles; zz = L / a.ntiles; }     // fewer than 8 batch splits (very wide outputs): every XCD gets tiles
    else { const int slot = L >> 3; tile = slot % a.ntiles; zz = (slot / a.ntiles) * 8 + (L & 7); }      // all tiles of a batch split on one XCD: P / Q rows shared in its L2
    if (zz >= a.nsplit) return;
    const int tn = tile / a.ntk, tk = tile % a.ntk;
    const int n0 = tn * C::NT, k0 = tk * C::KT;
    const int M = a.M, N = a.N, K = a.K;
    const int m_begin = zz * a.rps, m_end = min(M, m_begin + a.rps);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / C::WK, wk = wid % C::WK;

    if constexpr (PMODE == 1) {
        for (int c = tid; c < C::NT; c += C::THREADS) {
            const int col = n0 + c;
            const bool ok = col < N;
            float c0 = 0.f, k1 = 0.f, k2 = 0.f;
            if (ok) tnw_coef(a, col, c0, k1, k2);
            aux[c] = ok ? a.mean[col] : 0.f; aux[C::NT + c] = c0;
            aux[2 * C::NT + c] = ok ? c0 * k2 * a.rstd[col] : 0.f; aux[3 * C::NT + c] = ok ? c0 * k1 : 0.f;
        }
        tnw_bn_grads(a, n0, C::NT, zz, tk, tid, C::THREADS);
        __syncthreads();
    }

    f32x4 acc[C::PA][C::QB];
#pragma unroll
    for (int i = 0; i < C::PA; ++i)
#pragma unroll
        for (int j = 0; j < C::QB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[C::PA];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) bsum[i] = 0.f;
    const bool do_bias = a.db != nullptr && tk == 0 && wk == 0;     // wave-uniform

    struct PRaw { bf16x8 d, y; };
    PRaw rp[C::PI];
    TnwQRaw<QT, QVEC> rq[C::QI];
    const int nt = (m_end - m_begin + C::MT - 1) / C::MT;
    const int ncap = ((N + 7) & ~7) - 8;                  // activation rows are padded to 8 elements
    const QT* qbase = (const QT*)a.q;

    // chunk c of a tile: row r = c / CH, chunk ch = c % CH; threads past the last chunk repeat it (same data to the same place)
    auto fetch = [&](int t) {
#pragma unroll
        for (int i = 0; i < C::PI; ++i) {
            const int c = min(tid + C::THREADS * i, C::MT * C::PCH - 1), r = c / C::PCH, ch = c % C::PCH;
            const unsigned row = (unsigned)min(min(m_begin + t * C::MT + r, m_end - 1), M - 1);
            const unsigned col = (unsigned)min(n0 + ch * 8, ncap);
            rp[i].d = *(const bf16x8*)(a.p + (row * a.ldp + col));
            if constexpr (PMODE == 1) rp[i].y = *(const bf16x8*)(a.py + (row * a.ldpy + col));
        }
#pragma unroll
        for (int i = 0; i < C::QI; ++i) {
            const int c = min(tid + C::THREADS * i, C::MT * C::QCH - 1), r = c / C::QCH, ch = c % C::QCH;
            const unsigned ro = (unsigned)min(min(m_begin + t * C::MT + r, m_end - 1), M - 1) * a.ldq;
            const int col = k0 + ch * 8;
            if constexpr (sizeof(QT) == 2) {
                rq[i].v = *(const bf16x8*)(qbase + (ro + (unsigned)min(col, ((K + 7) & ~7) - 8)));
            } else {
#pragma unroll
                for (int j = 0; j < 8; j += QVEC) VLoad<float, QVEC>::ld(qbase + (ro + (unsigned)min(col + j, K - QVEC)), &rq[i].v[j]);
            }
        }
    };
    // registers -> LDS.  P rows past the split become zeros (they drop out of the reduction); Q is stored unmasked: what lies
    // past K is finite and lands in output columns that are never stored, rows past the split meet the zeroed P rows.
    auto stage = [&](int t, int buf) {
        unsigned char* sP = smem + buf * C::BUF;
        unsigned char* sQ = sP + C::MT * C::PROW;
#pragma unroll
        for (int i = 0; i < C::PI; ++i) {
            const int c = min(tid + C::THREADS * i, C::MT * C::PCH - 1), r = c / C::PCH, ch = c % C::PCH;
            const bool ok = m_begin + t * C::MT + r < m_end;
            bf16x8 o;
            if constexpr (PMODE == 1) {
                const float* ax = aux + ch * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // P = c0 (d - c1 - (y - mean) rstd c2), as gemm_src.h's SrcBnBwdApply
                    const float tt = fmaf((float)rp[i].y[e] - ax[e], ax[2 * C::NT + e], ax[3 * C::NT + e]);
                    const float v = fmaf(ax[C::NT + e], (float)rp[i].d[e], -tt);
                    o[e] = (bf16)(ok ? v : 0.f);
                }
            } else {
                const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                o = ok ? rp[i].d : z;
            }
            *(bf16x8*)(sP + tnw_chunk_off(r, ch, C::PROW)) = o;
        }
#pragma unroll
        for (int i = 0; i < C::QI; ++i) {
            const int c = min(tid + C::THREADS * i, C::MT * C::QCH - 1), r = c / C::QCH, ch = c % C::QCH;
            bf16x8 o;
            if constexpr (sizeof(QT) == 2) o = rq[i].v;
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)rq[i].v[e];
            }
            *(bf16x8*)(sQ + tnw_chunk_off(r, ch, C::QROW)) = o;
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* sP = smem + buf * C::BUF;
        const unsigned char* sQ = sP + C::MT * C::PROW;
        bf16x8 af[C::PA];
#pragma unroll
        for (int i = 0; i < C::PA; ++i) af[i] = tnw_frag(sP, C::PROW, (wn * C::PA + i) * 16, lane);
#pragma unroll
        for (int j = 0; j < C::QB; ++j) {
            const bf16x8 bq = tnw_frag(sQ, C::QROW, (wk * C::QB + j) * 16, lane);
#pragma unroll
            for (int i = 0; i < C::PA; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, af[i], acc[i][j], 0, 0, 0);   // swapped: a lane holds 4 consecutive k of one n
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < C::PA; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[i] += (float)af[i][e];
        }
    };

    fetch(0);
    stage(0, 0);
    if (nt > 1) fetch(1);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        compute(t & 1);
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        if (t + 2 < nt) fetch(t + 2);
        __syncthreads();
    }

    // accumulator (i, j), lane (li = lane & 15, lg = lane >> 4): dW[n = 16 i + li][k = 16 j + 4 lg + e], e = 0..3
    float* sl = a.slab + (long)zz * N * K;
    const bool v4 = (K & 3) == 0, v2 = (K & 1) == 0;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
        const int n = n0 + (wn * C::PA + i) * 16 + (lane & 15);
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < C::QB; ++j) {
            const int k = k0 + (wk * C::QB + j) * 16 + (lane >> 4) * 4;
            if (k >= K) continue;
            float* sp = sl + (long)n * K + k;
            if (v4) *(f32x4*)sp = acc[i][j];                     // K % 4 == 0: k + 4 <= K and 16-byte aligned
            else if (v2) {
                *(f32x2*)sp = f32x2{acc[i][j][0], acc[i][j][1]};
                if (k + 2 < K) *(f32x2*)(sp + 2) = f32x2{acc[i][j][2], acc[i][j][3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < K) sp[e] = acc[i][j][e];
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < C::PA; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            const int n = n0 + (wn * C::PA + i) * 16 + lane;
            if (lane < 16 && n < N) unsafeAtomicAdd(a.db + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA form: the RAW operand tiles (bf16 d and y, fp32 or bf16 Q) are moved global -> LDS by global_load_lds_dwordx4 into a
// two-stage ring and everything else happens on the way from LDS to the MFMA: P fragments are the transposed reads of d and y
// with the BatchNorm-backward correction applied per lane (a lane of a transposed fragment holds 8 batch rows of ONE column, so
// its four column constants live in registers), fp32 Q fragments are 8 plain ds_read_b32 (rows 8g .. 8g+7 of one column)
// converted in pairs.  No VGPR staging, no LDS write pass, no second copy of the tile: the register form above spent its step
// waiting for the loads of its single register set (68 KB per step and CU at the ~25 B/clk a CU ingests) and then converting.
// Swizzles are applied on the SOURCE side of the DMA (the LDS destination of a wave-instruction is lane-linear, 1 KB):
//   bf16 tiles: gemm_tn.hip's XOR of the 32-byte units of a 256-byte panel;
//   fp32 Q tile [32][KT]: rows of KT / 16 units of 64 bytes, unit u of row r stored at (u + (r >> 3)) mod units -- the four row
//   groups of a ds_read_b32 (16 lanes x 4 bytes each) then fall into four different 16-bank groups.
// Needs: M % 32 == 0 (every step a full tile), fp32 Q rows 16-byte aligned with K % 4 == 0.
// ------------------------------------------------------------------------------------------
// One LDS-DMA wave-instruction (64 lanes x 16 bytes -> 1 KB of LDS at `lds_addr`, lane-linear) as inline assembly.  The builtin
// form makes hipcc's wait-count pass treat every later LDS read as possibly reading the DMA's destination: it puts
// s_waitcnt vmcnt(0) in front of the first ds_read behind the issue, so the tile's whole trip from HBM sits inside each step
// and a deeper ring buys nothing.  Issued from assembly the transfers are invisible to that pass and are waited for only by
// the counted s_waitcnt vmcnt(N) of the ring below (vmcnt counts them in issue order like any other vector-memory op).
__device__ __forceinline__ void tnw_dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(g) : "memory");      // m0 is written: nothing else in this kernel uses it
}
__device__ __forceinline__ unsigned tnw_lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <class C, int PMODE, typename QT, int NSTAGE>
__global__ __launch_bounds__(512)
void gemm_tnw_dma_kernel(const TnwArgs a)
{
    constexpr bool QF = sizeof(QT) == 4;
    constexpr int QROWB = QF ? C::KT * 4 : C::QROW;
    constexpr int QUNITS = C::KT / 16;                     // fp32: 64-byte units per row
    constexpr int PBYTES = C::MT * C::PROW, QBYTES = C::MT * QROWB;
    constexpr int STAGE = PBYTES * (PMODE ? 2 : 1) + QBYTES;
    constexpr int PP = PBYTES / 1024, QP = QBYTES / 1024;  // 1 KB pieces per tile
    static_assert(PBYTES % 1024 == 0 && QBYTES % 1024 == 0, "whole DMA pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;

    const int L = (int)blockIdx.x;
    int tile, zz;
    if (a.linear) { tile = L % a.ntiles; zz = L / a.ntiles; }
    else { const int slot = L >> 3; tile = slot % a.ntiles; zz = (slot / a.ntiles) * 8 + (L & 7); }
    if (zz >= a.nsplit) return;
    const int tn = tile / a.ntk, tk = tile % a.ntk;
    const int n0 = tn * C::NT, k0 = tk * C::KT;
    const int M = a.M, N = a.N, K = a.K;
    const int m_begin = zz * a.rps, m_end = min(M, m_begin + a.rps);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv / C::WK, wk = wv % C::WK;
    const int li = lane & 15, g = lane >> 4;

    // per-lane BatchNorm-backward constants of the PA fragment columns this wave multiplies
    float cm[C::PA], c0[C::PA], c2[C::PA], c1[C::PA];
    if constexpr (PMODE == 1) {
#pragma unroll
        for (int i = 0; i < C::PA; ++i) {
            const int col = n0 + (wn * C::PA + i) * 16 + li;
            const bool ok = col < N;
            float k0_ = 0.f, k1_ = 0.f, k2_ = 0.f;
            if (ok) tnw_coef(a, col, k0_, k1_, k2_);
            cm[i] = ok ? a.mean[col] : 0.f; c0[i] = k0_;
            c2[i] = ok ? k0_ * k2_ * a.rstd[col] : 0.f; c1[i] = ok ? k0_ * k1_ : 0.f;
        }
        tnw_bn_grads(a, n0, C::NT, zz, tk, tid, C::THREADS);
    }

    // DMA source offsets (elements, relative to the first row of a step) of this lane's chunk in each piece this wave issues
    constexpr int PI = (PP + 7) / 8, QI = (QP + 7) / 8;
    unsigned psrc[PI], qsrc[QI];
    const int ncap = ((N + 7) & ~7) - 8;
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int off = (wv + 8 * i) * 1024 + lane * 16;
        const int r = off / C::PROW, s_ = (off % C::PROW) >> 4, pos = s_ & 15;
        const int cin = (((pos >> 1) ^ tnw_f(r)) << 1) | (pos & 1);
        psrc[i] = (unsigned)r * a.ldp + (unsigned)min(n0 + ((s_ >> 4) * 16 + cin) * 8, ncap);      // ldp == ldpy is checked by the host
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int off = (wv + 8 * i) * 1024 + lane * 16;
        const int r = off / QROWB, s_ = (off % QROWB) >> 4;
        if constexpr (QF) {
            int u = (s_ >> 2) - ((r >> 3) & 3);
            u += u < 0 ? QUNITS : 0;
            qsrc[i] = (unsigned)r * a.ldq + (unsigned)min(k0 + (u * 4 + (s_ & 3)) * 4, K - 4);
        } else {
            const int pos = s_ & 15;
            const int cin = (((pos >> 1) ^ tnw_f(r)) << 1) | (pos & 1);
            qsrc[i] = (unsigned)r * a.ldq + (unsigned)min(k0 + ((s_ >> 4) * 16 + cin) * 8, ((K + 7) & ~7) - 8);
        }
    }
    auto issue = [&](int t, int st) {
        unsigned char* sD = smem + st * STAGE;
        unsigned char* sQ = sD + PBYTES * (PMODE ? 2 : 1);
        const long m0 = m_begin + (long)t * C::MT;
        const bf16* gd = a.p + m0 * a.ldp;
        const bf16* gy = a.py + m0 * a.ldpy;
        const QT* gq = (const QT*)a.q + m0 * a.ldq;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int p = wv + 8 * i;
            if (PP % 8 == 0 || p < PP) {
                tnw_dma16(gd + psrc[i], tnw_lds_addr(sD + p * 1024));
                if constexpr (PMODE == 1) tnw_dma16(gy + psrc[i], tnw_lds_addr(sD + PBYTES + p * 1024));
            }
        }
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int p = wv + 8 * i;
            if (QP % 8 == 0 || p < QP) tnw_dma16(gq + qsrc[i], tnw_lds_addr(sQ + p * 1024));
        }
    };

    f32x4 acc[C::PA][C::QB];
#pragma unroll
    for (int i = 0; i < C::PA; ++i)
#pragma unroll
        for (int j = 0; j < C::QB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[C::PA];
#pragma unroll
    for (int i = 0; i < C::PA; ++i) bsum[i] = 0.f;
    const bool do_bias = a.db != nullptr && tk == 0 && wk == 0;
    const int nt = (m_end - m_begin) / C::MT;

    auto compute = [&](int st) {
        const unsigned char* sD = smem + st * STAGE;
        const unsigned char* sQ = sD + PBYTES * (PMODE ? 2 : 1);
        bf16x8 af[C::PA];
#pragma unroll
        for (int i = 0; i < C::PA; ++i) {
            const bf16x8 d = tnw_frag(sD, C::PROW, (wn * C::PA + i) * 16, lane);
            if constexpr (PMODE == 1) {
                const bf16x8 y = tnw_frag(sD + PBYTES, C::PROW, (wn * C::PA + i) * 16, lane);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float tt = fmaf((float)y[e] - cm[i], c2[i], c1[i]);        // as gemm_src.h's SrcBnBwdApply
                    af[i][e] = (bf16)fmaf(c0[i], (float)d[e], -tt);
                }
            } else af[i] = d;
        }
#pragma unroll
        for (int j = 0; j < C::QB; ++j) {
            bf16x8 bq;
            if constexpr (QF) {
                int u = wk * C::QB + j + g;
                u -= u >= QUNITS ? QUNITS : 0;
                const unsigned char* base = sQ + (8 * g) * QROWB + u * 64 + li * 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) bq[e] = (bf16)(*(const float*)(base + e * QROWB));
            } else bq = tnw_frag(sQ, QROWB, (wk * C::QB + j) * 16, lane);
#pragma unroll
            for (int i = 0; i < C::PA; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq, af[i], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < C::PA; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) bsum[i] += (float)af[i][e];
        }
    };

    // NSTAGE-deep ring, NSTAGE - 1 steps of DMA ahead of the MFMAs: with two stages a step lasted as long as ONE tile's trip
    // from HBM (issue -> landed, ~2-3 us under load), not as long as its arithmetic.  vmcnt counts this wave's pieces in issue
    // order: step t has landed when at most the pieces of the NSTAGE - 2 later steps are outstanding.
    constexpr int NP_LO = (PMODE ? 2 : 1) * (PP / 8) + QP / 8;                 // pieces per step of a wave (waves < PP % 8 / QP % 8 issue one more)
    const int extra = ((PP % 8 != 0 && wv < PP % 8) ? (PMODE ? 2 : 1) : 0) + ((QP % 8 != 0 && wv < QP % 8) ? 1 : 0);
#ifdef MM_STAMP
    unsigned long long sa[4] = {0, 0, 0, 0};
    MW_T(t_begin);
#endif
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE - 1; ++s_) if (s_ < nt) issue(s_, s_);
    // The ring is unrolled so that every stage address is a compile-time offset: with a run-time stage index hipcc cannot tell
    // the LDS-DMA destination from the stage the fragment reads come from and puts s_waitcnt vmcnt(0) in front of the first
    // ds_read of a step, i.e. right behind the DMA issue -- the whole trip from HBM then sits inside every step.
    for (int t0 = 0; t0 < nt; t0 += NSTAGE) {
#pragma unroll
        for (int u = 0; u < NSTAGE; ++u) {
            const int t = t0 + u;
            if (t >= nt) break;
            MW_T(w0);
            if (NSTAGE > 2 && t + NSTAGE - 2 < nt) {             // steady state: NSTAGE - 2 later steps may still be in flight
                if (extra == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * NP_LO) : "memory");
                else if (extra == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * (NP_LO + 1)) : "memory");
                else if (extra == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * (NP_LO + 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * (NP_LO + 3)) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MW_T(w1);
            __builtin_amdgcn_s_barrier();                        // everybody's pieces of step t have landed; nobody still reads stage t - 1
            asm volatile("" ::: "memory");
            MW_T(w2);
            if (t + NSTAGE - 1 < nt) issue(t + NSTAGE - 1, (u + NSTAGE - 1) % NSTAGE);      // the stage step t - 1 was read from
            MW_T(w3);
            compute(u);
#ifdef MM_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MW_T(w4);
            sa[0] += w1 - w0; sa[1] += w2 - w1; sa[2] += w3 - w2; sa[3] += w4 - w3;
#endif
        }
    }
    MW_T(t_loop_end);

    float* sl = a.slab + (long)zz * N * K;
    const bool v4 = (K & 3) == 0, v2 = (K & 1) == 0;
#pragma unroll
    for (int i = 0; i < C::PA; ++i) {
        const int n = n0 + (wn * C::PA + i) * 16 + li;
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < C::QB; ++j) {
            const int k = k0 + (wk * C::QB + j) * 16 + g * 4;
            if (k >= K) continue;
            float* sp = sl + (long)n * K + k;
            if (v4) *(f32x4*)sp = acc[i][j];
            else if (v2) {
                *(f32x2*)sp = f32x2{acc[i][j][0], acc[i][j][1]};
                if (k + 2 < K) *(f32x2*)(sp + 2) = f32x2{acc[i][j][2], acc[i][j][3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < K) sp[e] = acc[i][j][e];
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < C::PA; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            const int n = n0 + (wn * C::PA + i) * 16 + lane;
            if (lane < 16 && n < N) unsafeAtomicAdd(a.db + n, v);
        }
    }
#ifdef MM_STAMP
    if (tid == 0 && (blockIdx.x & 7) == 3) {
        MW_T(t_end);
        for (int i = 0; i < 4; ++i) atomicAdd(&mm_stamps_tnw[i], sa[i]);
        atomicAdd(&mm_stamps_tnw[4], (unsigned long long)nt);
        atomicAdd(&mm_stamps_tnw[5], 1ull);
        atomicAdd(&mm_stamps_tnw[6], t_end - t_begin);
        atomicAdd(&mm_stamps_tnw[7], t_end - t_loop_end);
    }
#endif
}

template <class C, int PMODE, typename QT, int NSTAGE>
static int tnw_dma_launch(TnwArgs& w, hipStream_t st) {
    constexpr bool QF = sizeof(QT) == 4;
    constexpr int LDS = NSTAGE * (C::MT * C::PROW * (PMODE ? 2 : 1) + C::MT * (QF ? C::KT * 4 : C::QROW));
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tnw_dma_kernel<C, PMODE, QT, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int grid = w.linear ? w.nsplit * w.ntiles : ((w.nsplit + 7) / 8) * 8 * w.ntiles;
    hipLaunchKernelGGL((gemm_tnw_dma_kernel<C, PMODE, QT, NSTAGE>), dim3(grid), dim3(512), LDS, st, w);
    MM_CHECK_LAUNCH();
    return 0;
}

typedef TnwCfg<4, 4, 2, 9> CfgA;        // 256 x 288
typedef TnwCfg<2, 4, 4, 7> CfgB;        // 128 x 448

template <class C, int PMODE, typename QT, int QVEC>
static int tnw_launch(TnwArgs& w, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tnw_kernel<C, PMODE, QT, QVEC>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    const int grid = w.linear ? w.nsplit * w.ntiles : ((w.nsplit + 7) / 8) * 8 * w.ntiles;
    hipLaunchKernelGGL((gemm_tnw_kernel<C, PMODE, QT, QVEC>), dim3(grid), dim3(512), C::LDS, st, w);
    MM_CHECK_LAUNCH();
    return 0;
}

static int g_tn_wide_on = getenv("MMVAE_NO_TN_WIDE") ? 0 : 1;
void tn_wide_enable(int on) { g_tn_wide_on = on; }

// Returns 0 after launching the GEMM (the caller then runs the slab reduce over *nsplit_out splits), > 0 on a launch error,
// TN_WIDE_NA when the problem is not one of the wide kernel's.
int launch_tn_wide(const mmvae_gemm_tn_args* a, hipStream_t st, int* nsplit_out) {
    constexpr int NA = -100;
    if (!g_tn_wide_on || a->prec != MMVAE_PREC_BF16 || a->p_dtype != MMVAE_BF16 || a->q_prologue != MMVAE_PRO_NONE || a->nsplit > 0) return NA;
    if (a->M < 8192 || !a->slab || a->N < 128 || a->K < 256) return NA;
    const int pmode = a->p_prologue == MMVAE_PRO_BN_BWD_APPLY ? 1 : 0;
    if (a->p_prologue != MMVAE_PRO_NONE && !pmode) return NA;
    if (a->ldp % 8 || ((uintptr_t)a->p & 15)) return NA;
    const bool fin = pmode && !a->p_coef;                 // mmvae_bn_bwd_finalize folded in
    if (pmode && (!a->p_y || !a->p_mean || !a->p_rstd || a->ld_py % 8 || ((uintptr_t)a->p_y & 15) || a->N % 8)) return NA;
    if (fin && (!a->p_sum_d || !a->p_sum_dx || !a->p_gamma || !a->p_dgamma || !a->p_dbeta)) return NA;
    int qkind;                                             // 0 bf16, 4 / 2: fp32 in vectors of 4 / 2
    if (a->q_dtype == MMVAE_BF16) { if (a->ldq % 8 || ((uintptr_t)a->q & 15)) return NA; qkind = 0; }
    else if (a->ldq % 4 == 0 && a->K % 4 == 0 && ((uintptr_t)a->q & 15) == 0) qkind = 4;
    else if (a->ldq % 2 == 0 && a->K % 2 == 0 && ((uintptr_t)a->q & 7) == 0) qkind = 2;
    else return NA;
    // instantiated combinations: an fp32 Q (the input batch of the first encoder layers) with a BatchNorm-corrected bf16 P (the bench
    // widths) or a plain bf16 P (very wide inputs, where the engine applies the correction in a pass of its own: hundreds of K tiles would
    // each redo it).  Plain bf16 x bf16 problems (last decoder layers) were measured on this kernel too (3-stage ring): 57-59 us against
    // 58 us of gemm_tn.hip's 128 x 128 DMA form with two workgroups per CU, plus a larger slab reduce -- they stay there.
    if (qkind == 0) {
        // bf16 x bf16 with a very large output (the decoders' last layers at the scaled widths: 27 000 x 512): 128 x 128 tiles re-ingest
        // P four times and Q 211 times; at the bench widths gemm_tn.hip's form with two workgroups per CU is as fast (above)
        if (pmode || a->M % 32 || (long)a->N * a->K < (4L << 20)) return NA;
    } else if (!pmode && (qkind != 4 || a->M % 32)) return NA;   // plain P: LDS-DMA forms only
    auto padded = [&](int nt_, int kt_) { return (long)((a->N + nt_ - 1) / nt_ * nt_) * ((a->K + kt_ - 1) / kt_ * kt_); };
    const int cfg = qkind == 0 ? 0 : (a->N <= 128 || padded(CfgB::NT, CfgB::KT) < padded(CfgA::NT, CfgA::KT)) ? 1 : 0;       // least padded output
    // 256 x 288 tiles run the LDS-DMA form: whole 32-row steps, 16-byte aligned fp32 rows, d and y with one row stride
    if (pmode && cfg == 0 && (qkind != 4 || a->M % 32 || a->ldp != a->ld_py)) return NA;
    const int NT = cfg == 0 ? CfgA::NT : CfgB::NT, KT = cfg == 0 ? CfgA::KT : CfgB::KT;
    TnwArgs w;
    w.ntk = (a->K + KT - 1) / KT;
    w.ntiles = w.ntk * ((a->N + NT - 1) / NT);
    if (w.ntiles > (pmode ? 32 : 4096)) return NA;         // corrected P: every K tile redoes the correction -- the 128 x 128 kernel's case
    int nsplit = 256 / w.ntiles;                           // one workgroup per CU
    if (nsplit >= 8) nsplit &= ~7;                         // whole XCD rounds
    const int max_split = (a->M + 4 * 32 - 1) / (4 * 32);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int rps = (a->M + nsplit - 1) / nsplit;
    rps = (rps + 31) / 32 * 32;
    nsplit = (a->M + rps - 1) / rps;
    if ((long)nsplit * a->N * a->K > a->slab_elems) return NA;
    if (pmode && nsplit < 2) return NA;
    w.nsplit = nsplit; w.rps = rps; w.linear = nsplit < 8 ? 1 : 0;
    w.p = (const bf16*)a->p; w.ldp = (unsigned)a->ldp; w.py = (const bf16*)a->p_y; w.ldpy = (unsigned)a->ld_py;
    w.mean = a->p_mean; w.rstd = a->p_rstd; w.coef = a->p_coef;
    w.sum_d = a->p_sum_d; w.sum_dx = a->p_sum_dx; w.gamma = a->p_gamma; w.dgamma = a->p_dgamma; w.dbeta = a->p_dbeta; w.eval_mode = a->p_eval_mode;
    w.q = a->q; w.ldq = (unsigned)a->ldq; w.M = a->M; w.N = a->N; w.K = a->K; w.slab = a->slab; w.db = a->db;
    *nsplit_out = nsplit;
    if (qkind == 0) return tnw_dma_launch<CfgA, 0, bf16, 3>(w, st);
    if (!pmode) return cfg == 0 ? tnw_dma_launch<CfgA, 0, float, 3>(w, st) : tnw_dma_launch<CfgB, 0, float, 2>(w, st);
    if (cfg == 0) return tnw_dma_launch<CfgA, 1, float, 2>(w, st);
    // 128 x 448 tiles (EncoderA.L0: K = 782, fp32 rows only 8-byte aligned -- 16-byte LDS-DMA pieces from such rows delivered wrong
    // data): the register form
    return qkind == 4 ? tnw_launch<CfgB, 1, float, 4>(w, st) : tnw_launch<CfgB, 1, float, 2>(w, st);
}

}  // namespace mm

#ifdef MM_STAMP
extern "C" int mmvae_debug_stamps_tnw(unsigned long long* out12, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out12, HIP_SYMBOL(mm::mm_stamps_tnw), 12 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    if (reset) { unsigned long long z[12] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(mm::mm_stamps_tnw), z, sizeof(z)); }
    return (int)e;
}
#endif
