// Epilogues of the NT GEMM kernel (gemm_nt.hip): what happens to the f32 accumulators of a 128 x (64*WN) tile owned by
// 2*WN waves, 64x64 each, 4x4 MFMA 16x16 tiles per wave.
//
// The kernel multiplies with the operands SWAPPED (W fragment as MFMA "A", activation fragment as "B"), so a 16x16
// accumulator tile comes out transposed: lane (li = lane & 15, lg = lane >> 4) holds, for tile (m, n),
//     row  = 64*wr + 16*m + li                      (one activation row per lane)
//     cols = 64*wc + wrow(16*n + 4*lg + j)          (j = 0..3: four consecutive columns)
// where wrow() is the order in which the W rows of the wave's 64 columns were put into the MFMA tiles (EpiCols below: the
// kernel applies it to the W row it FETCHES for each LDS row, which costs nothing).  For 4-byte outputs wrow is the
// identity: a lane's 4 columns are 16 bytes and the 4 lane groups complete 64 contiguous bytes of the row.  For 2-byte
// outputs tiles (2g, 2g+1) are interleaved so that a lane owns 8 consecutive columns (16 bytes) and the 4 lane groups
// again complete 64 contiguous bytes (8-byte pieces measured 35 % slower on the store-bound layers).
// Every epilogue access is a per-lane 16-byte vector: operands are read and results stored straight from/to global
// memory with no LDS transpose, no barrier and no idle waves.  The LDS-staged form this replaces cost as many cycles per
// tile as an 8-step main loop (tools/stamp_nt.py: 12-14 k cycles per wave, 64 ds_write_b16 per lane).
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
    static constexpr bool LDS_STREAM = false; // true: the tile leaves the kernel through an LDS image and a row-coalesced pass (EpiLoss)
    static constexpr int NEED = 0;            // operand tiles to stage: 0 none, 1 = H, 2 = H + mask
    typedef OT out_t; typedef OT h_t;
    OT* C; long ldc; const float* bias; int act; int accumulate;
    const OT* H; long ldh; const uint8_t* mask; long ldm;     // unused
    double* stat1; double* stat2;
    struct Col { float b; };
    static constexpr int NCOL = 1;            // per-column constants kept in LDS (BN floats each)
    __device__ __forceinline__ bool stores() const { return true; }
    __device__ __forceinline__ bool accum() const { return accumulate != 0; }
    bool accumulate_requested() const { return accumulate != 0; }
    __device__ __forceinline__ void fill(float* e, int BN, int cl, int c, int N) const { e[cl] = (bias && c < N) ? bias[c] : 0.f; }
    __device__ __forceinline__ Col col(const float* e, int BN, int cl) const { return Col{e[cl]}; }
    // ACT >= 0: the activation is known at compile time (the epilogue dispatches on it once per tile); ACT < 0: run time
    __device__ __forceinline__ int act_code() const { return act; }
    template <int ACT>
    __device__ __forceinline__ float compute(float v, float, unsigned, const Col& cc, bool count, float& s1, float& s2) const {
        v += cc.b;
        const int a = ACT < 0 ? act : ACT;
        if (a == 1) v = fmaxf(v, 0.f);
        else if (a == 2) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));       // v_exp_f32 / v_rcp_f32: ~1 ulp each, no IEEE-division sequence
        const float r = to_f32(from_f32<OT>(v));
        if (STATS && count) { s1 += r; s2 += r * r; }
        return r;
    }
};

template <typename OT, typename HT>
struct EpiReluMask {            // dH = (H > 0) ? acc : 0
    static constexpr bool STATS = false;
    static constexpr bool LDS_STREAM = false;
    static constexpr int NEED = 1;
    typedef OT out_t; typedef HT h_t;
    OT* C; long ldc; const HT* H; long ldh; const uint8_t* mask; long ldm;
    double* stat1; double* stat2;
    struct Col {};
    static constexpr int NCOL = 0;
    __device__ __forceinline__ bool stores() const { return true; }
    __device__ __forceinline__ bool accum() const { return false; }
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ void fill(float*, int, int, int, int) const {}
    __device__ __forceinline__ Col col(const float*, int, int) const { return Col{}; }
    __device__ __forceinline__ int act_code() const { return 0; }
    template <int ACT>
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
    static constexpr bool LDS_STREAM = false;
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
    static constexpr int NCOL = 7;
    __device__ __forceinline__ void fill(float* e, int BN, int cl, int c, int N) const {
        const bool ok = c < N, p1 = ok && phase == 1;
        e[cl] = ok ? scale[c] : 0.f; e[BN + cl] = ok ? shift[c] : 0.f; e[2 * BN + cl] = ok ? mean[c] : 0.f; e[3 * BN + cl] = ok ? rstd[c] : 0.f;
        e[4 * BN + cl] = p1 ? coef[c] : 0.f; e[5 * BN + cl] = p1 ? coef[N + c] : 0.f; e[6 * BN + cl] = p1 ? coef[2 * N + c] : 0.f;
    }
    __device__ __forceinline__ Col col(const float* e, int BN, int cl) const {
        return Col{e[cl], e[BN + cl], e[2 * BN + cl], e[3 * BN + cl], e[4 * BN + cl], e[5 * BN + cl], e[6 * BN + cl]};
    }
    __device__ __forceinline__ int act_code() const { return 0; }
    template <int ACT>
    __device__ __forceinline__ float compute(float v, float y, unsigned mb, const Col& cc, bool count, float& s1, float& s2) const {
        const float keep = mask ? (mb ? inv_keep : 0.f) : 1.f;
        const float d = (y * cc.sc + cc.sh > 0.f) ? v * keep : 0.f;
        const float xh = (y - cc.mu) * cc.rs;
        if (phase != 1) { if (count) { s1 += d; s2 += d * xh; } return d; }      // 0: statistics only; 2: statistics + store d
        return cc.c0 * (d - cc.c1 - xh * cc.c2);
    }
};

// Reconstruction loss of a decoder's last layer inside its GEMM (second-generation kernel only, gemm_nt2.h): the fp32 output
// x = acc + bias is never written.  MODE 0: sum-MSE against the fp32 target T, gradient 2 (x - T) (losses.py:31 and its
// backward); MODE 1: p = sigmoid(x), sum-BCE with torch's log clamp at -100 against T, gradient w.r.t. the LOGIT x
// (losses.py:34, decoders.py:32) -- the arithmetic of vae_loss_kernel (elementwise.hip) on the values the store epilogue would
// have written.  G: bf16 gradient rows (pad columns up to the next multiple of 8 are zeroed), sum: one f64 accumulator.
// VT: widest vector the target rows allow (4: 16-byte aligned rows and N % 4 == 0; 2: 8-byte aligned, N % 2 == 0; 1).
template <int MODE_, int VT_>
struct EpiLoss {
    static constexpr bool STATS = false;
    static constexpr bool LDS_STREAM = true;
    static constexpr int NEED = 0;
    static constexpr int MODE = MODE_, VT = VT_;
    typedef float out_t; typedef float h_t;           // 4-byte output layout: a lane's accumulator (m, n)[j] is column 16 n + 4 lg + j
    bf16* G; long ldg; const float* T; long ldt; const float* bias; double* sum;
    struct Col { float b; };
    static constexpr int NCOL = 1;
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ void fill(float* e, int BN, int cl, int c, int N) const { e[cl] = (bias && c < N) ? bias[c] : 0.f; }
    __device__ __forceinline__ float term(float x, float t, float& g) const {
        if constexpr (MODE == 0) { const float d = x - t; g = 2.f * d; return d * d; }
        else {
            const float pe = __builtin_amdgcn_rcpf(1.f + __expf(-x));
            const float lp = fmaxf(fast_ln(pe), -100.f), l1p = fmaxf(fast_ln(1.f - pe), -100.f);
            const float pq = (1.f - pe) * pe, d = pe - t;
            g = d * fminf(pq * 1e12f, 1.f);            // (p - t) unless p (1 - p) < 1e-12 (torch divides by max(p (1 - p), 1e-12)): d * 1 is exact
            return -(t * lp + (1.f - t) * l1p);
        }
    }
    // Two adjacent elements at once: the additions and multiplications as packed f32 pairs (v_pk_add_f32 / v_pk_mul_f32 /
    // v_pk_fma_f32: one issue slot for two elements -- the epilogue is bound by a single wave's instruction issue), transcendentals
    // and clamps per element.  Per element the SAME operations on the same values as term(): bit-identical gradients.  Returns the two
    // loss terms (the caller adds them up as pairs).
    __device__ __forceinline__ f32x2 term2(const f32x2 x, const f32x2 t, f32x2& g) const {
        if constexpr (MODE == 0) { const f32x2 d = x - t; g = d * 2.f; return d * d; }
        else {
            const f32x2 nx = x * -1.4426950408889634f;                     // __expf(-x) = exp2(-x log2 e)
            const f32x2 s = f32x2{__builtin_amdgcn_exp2f(nx[0]), __builtin_amdgcn_exp2f(nx[1])} + 1.f;
            const f32x2 pe = {__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};
            const f32x2 om = 1.f - pe;
            f32x2 lp = f32x2{__builtin_amdgcn_logf(pe[0]), __builtin_amdgcn_logf(pe[1])} * 0.6931471805599453f;
            f32x2 l1p = f32x2{__builtin_amdgcn_logf(om[0]), __builtin_amdgcn_logf(om[1])} * 0.6931471805599453f;
            lp = f32x2{fmaxf(lp[0], -100.f), fmaxf(lp[1], -100.f)};
            l1p = f32x2{fmaxf(l1p[0], -100.f), fmaxf(l1p[1], -100.f)};
            const f32x2 pq = om * pe, d = pe - t, c = pq * 1e12f;
            g = d * f32x2{fminf(c[0], 1.f), fminf(c[1], 1.f)};
            return -__builtin_elementwise_fma(1.f - t, l1p, t * lp);
        }
    }
};

// BatchNorm+ReLU+Dropout backward around a dX contraction (EpiBnBwd phase 2: store d, accumulate sum d and sum d*xhat) in the
// row-coalesced LDS form of the second-generation kernel (gemm_nt2.h, nt2_bnbwd_epilogue): bf16 y / d, uint8 keep mask.
struct EpiBnBwdStream {
    static constexpr bool STATS = true;
    static constexpr bool LDS_STREAM = true;
    static constexpr int NEED = 0;
    static constexpr int MODE = 2;                    // 0 / 1: the loss epilogues
    typedef float out_t; typedef float h_t;           // 4-byte accumulator layout (see EpiLoss)
    bf16* C; long ldc; const bf16* Y; long ldy; const uint8_t* mask; long ldm;
    const float* scale; const float* shift; const float* mean; const float* rstd; float inv_keep;
    double* stat1; double* stat2;
    struct Col { float b; };
    static constexpr int NCOL = 4;                    // scale, shift, mean, rstd of the tile's columns
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ void fill(float* e, int BN, int cl, int c, int N) const {
        const bool ok = c < N;
        e[cl] = ok ? scale[c] : 0.f; e[BN + cl] = ok ? shift[c] : 0.f; e[2 * BN + cl] = ok ? mean[c] : 0.f; e[3 * BN + cl] = ok ? rstd[c] : 0.f;
    }
};

// ReLU backward around a dX contraction (EpiReluMask: C = (h > 0) ? acc : 0) in the row-coalesced LDS form (gemm_nt2.h): bf16 h and C.
struct EpiReluMaskStream {
    static constexpr bool STATS = false;
    static constexpr bool LDS_STREAM = true;
    static constexpr int NEED = 0;
    static constexpr int MODE = 3;
    typedef float out_t; typedef float h_t;
    bf16* C; long ldc; const bf16* H; long ldh;
    struct Col { float b; };
    static constexpr int NCOL = 0;
    bool accumulate_requested() const { return false; }
    __device__ __forceinline__ void fill(float*, int, int, int, int) const {}
};

// Column order of a wave's 64 output columns inside its 4 MFMA n-tiles.
template <bool PAIR> struct EpiCols {
    static constexpr int G = PAIR ? 8 : 4;            // consecutive columns a lane owns per group
    static constexpr int NG = 16 / G;                 // groups per lane (x 4 rows m)
    // W row (relative to the wave's 64) that goes to LDS row x = 16*n + i of the wave's W block
    static __device__ __forceinline__ int wrow(int x) {
        if constexpr (!PAIR) return x;
        const int n = x >> 4, i = x & 15;
        return 32 * (n >> 1) + 8 * (i >> 2) + 4 * (n & 1) + (i & 3);
    }
    // first column (relative to the wave's 64) of group g for lane group lg; element e of the group is accumulator
    // (n, j) = (PAIR ? 2g + (e >> 2) : g, e & 3)
    static __device__ __forceinline__ int base(int g, int lg) { return PAIR ? 32 * g + 8 * lg : 16 * g + 4 * lg; }
    static constexpr __device__ __forceinline__ int tile(int g, int e) { return PAIR ? 2 * g + (e >> 2) : g; }
};

// Epilogue operands (ReLU input / pre-BN output, keep mask) in the accumulator's own layout, fetched into registers a
// couple of K steps before the main loop ends so that their HBM latency is covered by the last MFMAs.
template <typename Epi>
struct EpiOperands {
    typedef typename Epi::h_t HT;
    typedef EpiCols<sizeof(typename Epi::out_t) == 2> EC;
    static constexpr int HD = EC::G * (int)sizeof(HT) / 4;        // dwords of H per group
    static constexpr int MD = EC::G / 4;                           // dwords of mask bytes per group
    static_assert(Epi::NEED == 0 || HD == 4, "epilogue operand groups are 16-byte vectors (H and output of the same width)");
    uint32_t h[4 * EC::NG][HD];
    uint32_t mk[4 * EC::NG][MD];
};

template <typename Epi> __device__ __forceinline__ bool epi_h_vec(const Epi& epi) {
    return Epi::NEED >= 1 && (epi.ldh * sizeof(typename Epi::h_t)) % 16 == 0 && ((uintptr_t)epi.H & 15) == 0;
}
// H rows are whole 128-byte lines and the tile's columns exist: H is then LOADED in the same full-line layout the stores
// use (8 lanes x 16 bytes per row and instruction) and swapped back into accumulator layout when it is used; loading the
// two 64-byte halves of a line from separate instructions cost +40 % HBM read traffic (FETCH_SIZE), the line having left
// the L2 before its second half was asked for.
// ncols: columns of H (and of the mask) a kernel may touch from the row's first element -- the N of the product rounded up to the
// 8-element padding every activation buffer has, NOT the leading dimension: H may be a column slice of a wider buffer, and for its
// last slice "inside the leading dimension" reaches past the row end (for the last row: past the allocation).
template <typename Epi> __device__ __forceinline__ bool epi_h_lines(const Epi& epi, int col0, int BN, int ncols) {
    return epi_h_vec(epi) && (epi.ldh * sizeof(typename Epi::h_t)) % 128 == 0 && ((uintptr_t)epi.H & 127) == 0 && col0 + BN <= ncols;
}
template <typename Epi> __device__ __forceinline__ bool epi_m_vec(const Epi& epi) {
    return Epi::NEED >= 2 && epi.mask != nullptr && epi.ldm % 8 == 0 && ((uintptr_t)epi.mask & 7) == 0;
}

// Branch-free (clamped addresses): these loads are issued between K steps, where a conditional load would cost the
// counted vmcnt waits of the remaining stage() calls (see gemm_src.h).  wr/wc: wave row/column inside the tile.
// HALF 0: rows m = 0, 1 (issued before the last K step); HALF 1: m = 2, 3 (issued when the epilogue starts, covered by
// the work on the first half) -- all 48 operand registers at once did not fit next to the accumulators and both
// fragment sets, and hipcc spilled the values it had just loaded.
template <typename Epi, int HALF>
__device__ __forceinline__ void nt_epilogue_prefetch(EpiOperands<Epi>& ops, const Epi& epi, int row0, int col0, int M, int N, int BN, int lane, int wr, int wc) {
    typedef EpiOperands<Epi> EO;
    typedef typename EO::EC EC;
    const int li = lane & 15, lg = lane >> 4;
    const int ncols = (N + 7) & ~7;
    if constexpr (Epi::NEED >= 1) {
        if (epi_h_lines(epi, col0, BN, ncols)) {
            // slot 2p holds (row 16m + (li & 7), half li >> 3), slot 2p+1 the same half of row 16m + 8 + (li & 7)
#pragma unroll
            for (int m = 2 * HALF; m < 2 * HALF + 2; ++m)
#pragma unroll
                for (int g = 0; g < EC::NG; ++g) {
                    const int r = row0 + wr * 64 + m * 16 + (g & 1) * 8 + (li & 7);
                    const int c = col0 + wc * 64 + EC::base((g & ~1) + (li >> 3), lg);
                    const uint4 t = *(const uint4*)(epi.H + (long)min(r, M - 1) * epi.ldh + c);
                    ops.h[m * EC::NG + g][0] = t.x; ops.h[m * EC::NG + g][1] = t.y; ops.h[m * EC::NG + g][2] = t.z; ops.h[m * EC::NG + g][3] = t.w;
                }
        } else if (epi_h_vec(epi)) {
#pragma unroll
            for (int m = 2 * HALF; m < 2 * HALF + 2; ++m) {
                const long ro = (long)min(row0 + wr * 64 + m * 16 + li, M - 1) * epi.ldh;
#pragma unroll
                for (int g = 0; g < EC::NG; ++g) {
                    const int c = col0 + wc * 64 + EC::base(g, lg);
                    const uint4 t = *(const uint4*)(epi.H + ro + (c + EC::G <= ncols ? c : 0));
                    ops.h[m * EC::NG + g][0] = t.x; ops.h[m * EC::NG + g][1] = t.y; ops.h[m * EC::NG + g][2] = t.z; ops.h[m * EC::NG + g][3] = t.w;
                }
            }
        }
    }
    if constexpr (Epi::NEED >= 2) {
        if (epi_m_vec(epi)) {
#pragma unroll
            for (int m = 2 * HALF; m < 2 * HALF + 2; ++m) {
                const long ro = (long)min(row0 + wr * 64 + m * 16 + li, M - 1) * epi.ldm;
#pragma unroll
                for (int g = 0; g < EC::NG; ++g) {
                    const int c = col0 + wc * 64 + EC::base(g, lg);
                    const uint8_t* q = epi.mask + ro + (c + EC::G <= ncols ? c : 0);
                    if constexpr (EO::MD == 2) { const uint2 t = *(const uint2*)q; ops.mk[m * EC::NG + g][0] = t.x; ops.mk[m * EC::NG + g][1] = t.y; }
                    else ops.mk[m * EC::NG + g][0] = *(const uint32_t*)q;
                }
            }
        }
    }
}

// Per-column constants of the epilogue (bias; BatchNorm scale/shift/mean/rstd/coefficients) go to LDS once per tile, at
// kernel start (the kernel's first barrier orders it): held in registers they cost up to 56 VGPRs of an epilogue that
// already carries 64 accumulators and 48 prefetched operand registers, and spilled.  ecol: NCOL * 64*WN floats.
template <typename Epi, int WN>
__device__ __forceinline__ void nt_epilogue_fill_cols(float* ecol, const Epi& epi, int col0, int N, int tid) {
    constexpr int BN = 64 * WN;
    if (Epi::NCOL > 0 && tid < BN) epi.fill(ecol, BN, tid, col0 + tid, N);
}

// The per-element work of the epilogue, specialised on what is uniform over the tile so that the 128 elements of a lane
// run without scalar branches: ACT (activation, -1 = run time), ACCUM (C += ...), VEC (operands prefetched as vectors;
// otherwise per-element guarded loads: odd leading dimensions, f32 precision mode corner cases).  The generic form cost
// ~5000 instructions per lane (8-10 k cycles per tile, as much as 5 K steps); this one ~600.
template <typename Epi, int ACT, bool ACCUM, bool VEC>
__device__ __forceinline__ void nt_epilogue_body(const float* ecol, f32x4 (&acc)[4][4], const Epi& epi, const EpiOperands<Epi>& ops,
                                                 float* red, bool want_stats, int BN, int row0, int col0, int M, int N, int lane, int wr, int wc,
                                                 int tile_rows = TILE)
{
    typedef typename Epi::out_t OT;
    typedef typename Epi::h_t HT;
    typedef EpiOperands<Epi> EO;
    typedef typename EO::EC EC;
    constexpr int G = EC::G, NG = EC::NG;
    constexpr int EPCO = 16 / (int)sizeof(OT);
    const int li = lane & 15, lg = lane >> 4;
    const int n_store = (int)min((long)((N + EPCO - 1) / EPCO * EPCO), epi.ldc);      // pad columns of internal buffers get zeros
    const bool c16 = (epi.ldc * sizeof(OT)) % 16 == 0 && ((uintptr_t)epi.C & 15) == 0;     // whole-group vector stores
    const bool c8 = (epi.ldc * sizeof(OT)) % 8 == 0 && ((uintptr_t)epi.C & 7) == 0;
    const bool has_mask = Epi::NEED >= 2 && epi.mask != nullptr;
    // Groups 2p and 2p+1 are the two 64-byte halves of one 128-byte line of the row.  When the whole tile is interior and
    // rows are 128-byte aligned, lanes li and li^8 swap one half each so that every store instruction writes 8 rows x one
    // FULL line (8 lanes x 16 bytes); storing the halves from separate instructions cost +55 % HBM write traffic
    // (rocprofv3 WRITE_SIZE: 104 MB for a 67 MB output) -- the L2 wrote the partial lines back twice.
    const bool full_lines = !ACCUM && epi.stores() && row0 + tile_rows <= M && col0 + BN <= n_store &&
                            (epi.ldc * sizeof(OT)) % 128 == 0 && ((uintptr_t)epi.C & 127) == 0;
    const unsigned low = (li & 8) ? 0u : 0xffffffffu;           // all-ones on lanes li < 8
    const bool h_lines = VEC && Epi::NEED >= 1 && epi_h_lines(epi, col0, BN, (N + 7) & ~7);
#pragma unroll
    for (int p = 0; p < NG / 2; ++p) {
        float s1[2][G], s2[2][G];                               // column partial sums of the two groups, over the lane's 4 rows
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < G; ++e) { s1[h][e] = 0.f; s2[h][e] = 0.f; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = row0 + wr * 64 + m * 16 + li;
            const bool rok = row < M;
            uint32_t pk[2][4];                                  // the two 16-byte pieces of this lane, packed for the store
            uint32_t hw[2][4];                                  // H of this lane's two groups, accumulator layout
            if constexpr (Epi::NEED >= 1 && VEC) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t l0 = ops.h[m * NG + 2 * p][q], l1 = ops.h[m * NG + 2 * p + 1][q];
                    hw[0][q] = l0; hw[1][q] = l1;
                    if (h_lines) {                              // loaded in full-line layout: swap back (see epi_h_lines)
                        const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)((l1 & low) | (l0 & ~low)), 0x128, 0xf, 0xf, true);   // row_ror:8 == lane li ^ 8: one VALU op (
                                                                                                                                   // __shfl_xor is a ds_bpermute round trip)
                        hw[0][q] = (l0 & low) | (got & ~low);
                        hw[1][q] = (got & low) | (l1 & ~low);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = 2 * p + h;
                const int cw = wc * 64 + EC::base(g, lg), c0 = col0 + cw;
                float hv[G]; unsigned mb[G]; float o[G];
#pragma unroll
                for (int e = 0; e < G; ++e) { hv[e] = 0.f; mb[e] = 1u; o[e] = 0.f; }
                if constexpr (Epi::NEED >= 1) {
                    if constexpr (VEC) {
#pragma unroll
                        for (int e = 0; e < G; ++e) {
                            if constexpr (sizeof(HT) == 2) {
                                const uint32_t w = hw[h][e >> 1];
                                hv[e] = __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
                            } else hv[e] = __uint_as_float(hw[h][e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < G; ++e) if (rok && c0 + e < N) hv[e] = to_f32(epi.H[(long)row * epi.ldh + c0 + e]);
                    }
                }
                if constexpr (Epi::NEED >= 2) {
                    if constexpr (VEC) {
                        if (has_mask) {
#pragma unroll
                            for (int e = 0; e < G; ++e) mb[e] = (ops.mk[m * NG + g][e >> 2] >> (8 * (e & 3))) & 0xffu;
                        }
                    } else if (has_mask) {
#pragma unroll
                        for (int e = 0; e < G; ++e) if (rok && c0 + e < N) mb[e] = epi.mask[(long)row * epi.ldm + c0 + e];
                    }
                }
                OT* gp = epi.C + (long)row * epi.ldc + c0;
                if constexpr (ACCUM) {                           // C += ...: read the old values first
#pragma unroll
                    for (int e = 0; e < G; ++e) if (rok && c0 + e < N) o[e] = to_f32(gp[e]);
                }
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    const int n = EC::tile(g, e), j = e & 3;
                    const bool ok = rok && (c0 + e < N);
                    const typename Epi::Col cc = epi.col(ecol, BN, cw + e);        // LDS (idle here); kept out of the registers
                    float v = epi.template compute<ACT>(acc[m][n][j], hv[e], mb[e], cc, ok, s1[h][e], s2[h][e]);
                    if constexpr (ACCUM) v += o[e];
                    o[e] = (c0 + e < N) ? v : 0.f;
                }
                if constexpr (sizeof(OT) == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t lo = __float_as_uint(to_f32(from_f32<OT>(o[2 * q]))), hi = __float_as_uint(to_f32(from_f32<OT>(o[2 * q + 1])));
                        pk[h][q] = (lo >> 16) | (hi & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) pk[h][q] = __float_as_uint(o[q]);
                }
                if (!full_lines && epi.stores() && rok && c0 < n_store) {       // edge tiles, odd leading dimensions, C += ...
                    const bool whole = c0 + G <= n_store;
                    if (whole && c16) *(uint4*)gp = uint4{pk[h][0], pk[h][1], pk[h][2], pk[h][3]};
                    else if (sizeof(OT) == 4 && whole && c8) { ((uint2*)gp)[0] = uint2{pk[h][0], pk[h][1]}; ((uint2*)gp)[1] = uint2{pk[h][2], pk[h][3]}; }
                    else {
#pragma unroll
                        for (int e = 0; e < G; ++e) if (c0 + e < n_store) gp[e] = from_f32<OT>(o[e]);
                    }
                }
            }
            if (full_lines) {
                // lanes li < 8 give away their second half and get row li+8's first half; lanes li >= 8 the other way round
                uint32_t st0[4], st1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t send = (pk[1][q] & low) | (pk[0][q] & ~low);
                    const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x128, 0xf, 0xf, true);             // row_ror:8 == lane li ^ 8
                    st0[q] = (pk[0][q] & low) | (got & ~low);        // rows m*16 + (li & 7):     own first half | row li-8's second half
                    st1[q] = (got & low) | (pk[1][q] & ~low);        // rows m*16 + 8 + (li & 7): row li+8's first half | own second half
                }
                const int rbase = row0 + wr * 64 + m * 16 + (li & 7);
                const int cd = col0 + wc * 64 + EC::base(2 * p + (li >> 3), lg);
                *(uint4*)(epi.C + (long)rbase * epi.ldc + cd) = uint4{st0[0], st0[1], st0[2], st0[3]};
                *(uint4*)(epi.C + (long)(rbase + 8) * epi.ldc + cd) = uint4{st1[0], st1[1], st1[2], st1[3]};
            }
        }
#ifndef MM_NO_STAT_REDUCE
        if (Epi::STATS && want_stats) {
            // Column sums of each group over the 16 lanes li of a lane group: reduce-scatter butterfly -- each step a lane
            // keeps the half of its values selected by one bit of li and adds the partner's copy of that half.
            // v index = which*G + e; the bits of li, from 8 down, select which, then e.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cw = wc * 64 + EC::base(2 * p + h, lg);
                float v[2 * G];
#pragma unroll
                for (int e = 0; e < G; ++e) { v[e] = s1[h][e]; v[G + e] = s2[h][e]; }
                int bit = 8;
#pragma unroll
                for (int half = G; half >= 1; half >>= 1, bit >>= 1) {
                    // bit-select, NOT `up ? v[i + half] : v[i]`: LLVM folds a select of two array elements into one dynamically
                    // indexed access, i.e. a 16-deep v_cmp / v_cndmask chain per value (20 k cycles per tile, measured)
                    const unsigned up = (li & bit) ? 0xffffffffu : 0u;
#pragma unroll
                    for (int i = 0; i < half; ++i) {
                        const unsigned lo = __float_as_uint(v[i]), hi = __float_as_uint(v[i + half]);
                        const float keep = __uint_as_float((hi & up) | (lo & ~up));
                        const float send = __uint_as_float((lo & up) | (hi & ~up));
                        v[i] = keep + __shfl_xor(send, bit, 64);
                    }
                }
                if constexpr (G == 4) v[0] += __shfl_xor(v[0], 1, 64);      // 8 values, 16 lanes: the last bit is a plain add
                const int which = li >> 3;
                const int e = G == 8 ? (li & 7) : ((li >> 1) & 3);
                if (G == 8 || (li & 1) == 0) red[(wr * 2 + which) * BN + cw + e] = v[0];
            }
        }
#endif
    }
}

// red: 1 KiB * WN of LDS (column partial sums of the 2 wave rows), only touched by STATS epilogues.
template <typename CT, typename Epi, int WN = 2>
__device__ __forceinline__ void nt_epilogue(float* red, const float* ecol, f32x4 (&acc)[4][4], const Epi& epi, EpiOperands<Epi>& ops,
                                            int row0, int col0, int M, int N, int tid, int lane, int wr, int wc)
{
    constexpr int BN = 64 * WN;
    typedef EpiOperands<Epi> EO;
    typedef typename EO::EC EC;
    const bool want_stats = Epi::STATS && (epi.stat1 != nullptr || epi.stat2 != nullptr);
    // operands arrive as prefetched vectors unless a leading dimension / base address rules 16-byte accesses out
    const bool vec = (Epi::NEED < 1 || epi_h_vec(epi)) && (Epi::NEED < 2 || epi.mask == nullptr || epi_m_vec(epi));
    if (epi.accum()) {
        nt_epilogue_body<Epi, -1, true, false>(ecol, acc, epi, ops, red, want_stats, BN, row0, col0, M, N, lane, wr, wc);
    } else if (!vec) {
        nt_epilogue_body<Epi, -1, false, false>(ecol, acc, epi, ops, red, want_stats, BN, row0, col0, M, N, lane, wr, wc);
    } else {
        nt_epilogue_prefetch<Epi, 1>(ops, epi, row0, col0, M, N, BN, lane, wr, wc);
        const int a = epi.act_code();
        if (a == 0) nt_epilogue_body<Epi, 0, false, true>(ecol, acc, epi, ops, red, want_stats, BN, row0, col0, M, N, lane, wr, wc);
        else if (a == 1) nt_epilogue_body<Epi, 1, false, true>(ecol, acc, epi, ops, red, want_stats, BN, row0, col0, M, N, lane, wr, wc);
        else nt_epilogue_body<Epi, 2, false, true>(ecol, acc, epi, ops, red, want_stats, BN, row0, col0, M, N, lane, wr, wc);
    }
    if (want_stats) {
        __syncthreads();
#ifndef MM_NO_STAT_ATOMICS
        if (tid < BN && col0 + tid < N) {
            if (epi.stat1) unsafeAtomicAdd(epi.stat1 + col0 + tid, (double)(red[0 * BN + tid] + red[2 * BN + tid]));
            if (epi.stat2) unsafeAtomicAdd(epi.stat2 + col0 + tid, (double)(red[1 * BN + tid] + red[3 * BN + tid]));
        }
#endif
    }
}

}  // namespace mm
