// NT GEMM, wave-specialised form (gfx950):  C[M,N] = epilogue( A[M,K] * W[N,K]^T )
//
// Replaces, on the MultiModalVAE training path: the forward first layers of EncoderA / EncoderB -- nn.Linear(782,128) and
// nn.Linear(572,512) on the caller's fp32 batch, reference src/models/encoders.py:13,31 -- with the BatchNorm1d batch statistics
// of encoders.py:14,32 as column sums of the stored outputs; and (bf16 A) the plain-store Linear forwards of the decoders,
// src/models/decoders.py:13,27,29.
//
// What the earlier generations (gemm_nt.hip, gemm_nt2.h) spend a K step on (cycle stamps): every wave first
// ISSUES the next step's operand traffic (663 of 1850 cycles) and then multiplies -- issue and MFMA are serial inside a wave, and
// with 9-13 K steps per tile a workgroup also pays its load prologue and its store epilogue per tile.  Here the waves of ONE
// persistent workgroup per CU (12 waves for 128 x 256 tiles, 8 for 128 x 128) have fixed roles:
//   * 4 producer waves, each 32 rows of the A tile and a quarter of the W tile per K step: A global -> VGPR -> (fp32 -> bf16) ->
//     LDS from THREE register sets (loads issued from inline assembly, scalar base + one lane offset per load, the set's counted
//     s_waitcnt vmcnt(N) written by hand), fully coalesced -- one wave-instruction reads 4 rows x 256 contiguous bytes; W (prepared
//     bf16 weights, L2-resident) L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, scalar base + ONE lane-constant VGPR offset);
//   * 8 / 4 MFMA consumer waves of 64 x 64 (2 x 4 / 2 x 2): LDS fragment reads + MFMA + the epilogue, nothing else.  Two consumer
//     waves per SIMD for the wide tile: the epilogue is bound by one wave's instruction issue (an instruction per 4-5 cycles), and
//     eight waves share it where four took 15 000 cycles per tile.
// (tile, K step) is one continuous sequence of steps across the workgroup's tiles: the producers run into the next tile while the
// consumers are in the epilogue, which runs behind the barrier that frees the tile's last ring slot.  ONE s_barrier per K step,
// executed by all waves: step g+2 is produced between barriers g and g+1 while step g is multiplied (rings of 3 slots).
// Measured (tools/stamp_ntp.py, tools/abl_ntp.sh): the K steps run at the rate the CU's memory path takes
// 16-byte accesses in (21-25 bytes per clock and CU from L2; issue of a vector-memory instruction then blocks for 60-220 cycles),
// which is what bounds this kernel, not HBM and not the MFMA pipe.
// The LDS images are the first generation's (128-byte rows = one K step of 64 bf16, 16-byte chunks XOR-ed with row & 7), so are
// the swapped-operand MFMA order and the store layout of the epilogue (gemm_nt_epi.h).
// BatchNorm statistics: per-lane sums over the lane's 4 rows, a 15-add DPP reduce-scatter over the 16 lanes that hold the other rows
// of the same columns (round 3; before: a wave-private transposition through 1 KB of LDS, 16 rounds per tile), kept in TWO registers
// per lane over ALL tiles of the workgroup's column tile, then one LDS add per workgroup and one f64 atomic per column and workgroup
// (256 adders per column instead of one per row tile).
#pragma once
#include "common.h"
#include "gemm_nt_epi.h"
#include "gemm_src.h"

#ifndef NTP_PRO_WCONS
#define NTP_PRO_WCONS 1
#endif
#ifndef NTP_ABL
#define NTP_ABL 0              // timing-only ablations (tools/abl_ntp.sh): 1 = W DMA from one fixed K step, 2 = A loads from one fixed K step and row tile,
#endif                         // 3 = no MFMA, 4 = no epilogue

namespace mm {

#ifdef MM_STAMP
// diagnostic build only (make STAMP=1, tools/stamp_ntp.py): cycles per role, summed over a sample of workgroups
__device__ unsigned long long mm_ntp_stamps[24];
#define NTP_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define NTP_ACC(i, d) st_acc[i] += (d)
#else
#define NTP_T(x)
#define NTP_ACC(i, d)
#endif

constexpr int NTP_SKIP = 1 << 30;        // ntp_dispatch: "not my problem", the caller goes on to the other kernels

template <int V> struct IntC { static constexpr int value = V; };

template <int NT_, int WC_> struct NtpCfg {
    static constexpr int WR = 2, WC = WC_, NT = NT_;          // consumer waves 2 x WC, each 64 rows x 16*NT columns
    static constexpr int NH = NT / 4;                         // 64-column blocks per consumer wave
    static constexpr int BM = 64 * WR, BN = 16 * NT * WC;
    static constexpr int NCONS = WR * WC, NPA = 4, NWAVES = NCONS + NPA;
    // who issues the W tile's LDS-DMA: the consumer waves at 128 x 128 (one per SIMD, idle for two thirds of a K step: EncoderA.L0.fwd
    // 55.4 -> 51.7 us, same box), the producers at 128 x 256 (two consumers per SIMD at 166 registers: 71.6 against 75.7 us)
    static constexpr bool WCONS = WC_ == 2;
    static constexpr int A_SLOT = BM * ROW_BYTES, W_SLOT = BN * ROW_BYTES;
    static constexpr int RA = 3, RW = 3;                      // ring slots: a step is produced two barriers before it is read
    static constexpr int OFF_W = RA * A_SLOT;
    static constexpr int OFF_SCR = OFF_W + RW * W_SLOT;       // 1 KB per consumer wave (spare: the statistics no longer pass through it)
    static constexpr int OFF_STAT = OFF_SCR + NCONS * 1024;   // [WR][2][BN] floats: the workgroup's running column sums
    static constexpr int OFF_ECOL = OFF_STAT + WR * 2 * BN * 4;
    static constexpr int TOTAL = OFF_ECOL + 2 * BN * 4;       // two copies of the per-column constants (tile parity): 158 KB (128 x 256), 104 KB (128 x 128)
};

// the lane id, formed where it is used: a volatile statement is not hoisted out of the K loop, and what is not hoisted is not spilled
__device__ __forceinline__ int ntp_lane_id() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// one s_barrier of the K-step protocol: the wave's LDS traffic has drained (its ds_writes are visible, its ds_reads are back);
// vector-memory operations stay in flight across it
__device__ __forceinline__ void ntp_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS-DMA piece with a SCALAR base and a 32-bit lane offset: 64 lanes x 16 bytes -> 1 KB of LDS at lds_base + LOFF.  m0 is written and
// read inside one statement (the compiler reserves the register but keeps nothing in it here: no LDS-DMA builtin, no movrel in
// this kernel); s_nop 0: the wait state between an SALU write of m0 and the LDS-DMA that reads it.
__device__ __forceinline__ void ntp_dma16(const void* sbase, unsigned voff, unsigned lds_base, int loff) {
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" :: "s"(lds_base), "n"(loff), "v"(voff), "s"(sbase) : "memory", "scc");
}

// 16 bytes of an A row into registers (four floats or eight bf16) from inline assembly: scalar base + 32-bit lane offset.  The
// compiler does not count these loads: its own wait-count analysis drained BOTH register sets at every other K step of the
// producer loop (one of the two unrolled halves waited vmcnt(14..0) where 16 younger loads could have stayed in flight).
// The destination must not be touched before ntp_wait_set() has named it (gfx950 loads of 16 bytes only need 4-byte alignment:
// the 782-float rows of the RNA matrix are 8-byte aligned).
__device__ __forceinline__ void ntp_ld16(f32x4& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// s_waitcnt vmcnt(N) that names the 8 / 16 registers of a set as read-write: nothing that uses them can be scheduled above it
template <int N>
__device__ __forceinline__ void ntp_wait_set(f32x4 (&s)[16]) {
    asm volatile("s_waitcnt vmcnt(%16)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),
                 "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void ntp_wait_set(f32x4 (&s)[8]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "n"(N) : "memory");
}

// Sum of v[i] over the 16 lanes of a DPP row, scattered: lane li (0..15 inside its row) returns the row's total of v[li].
// Recursive halving with DPP moves (row_ror:8 = lane ^ 8; lane ^ 4 as row_ror:12 for the lanes of banks 0 / 2 and row_ror:4 for the
// others -- row_ror:n makes lane i read lane (i - n) mod 16; quad_perm for lane ^ 2 and lane ^ 1): 15 adds instead of 60.
template <int CTRL> __device__ __forceinline__ float ntp_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ntp_row16_reduce_scatter(const float (&v)[16], int li) {
    const bool b3 = li & 8, b2 = li & 4, b1 = li & 2, b0 = li & 1;
    float a[8], b[4], c[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float keep = b3 ? v[8 + j] : v[j], send = b3 ? v[j] : v[8 + j]; a[j] = keep + ntp_dpp<0x128>(send); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b2 ? a[4 + j] : a[j], send = b2 ? a[j] : a[4 + j];
        const int lo = __builtin_amdgcn_mov_dpp(__float_as_int(send), 0x124, 0xf, 0xf, true);                  // from lane - 4 (lanes with bit 2 set)
        b[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(lo, __float_as_int(send), 0x12C, 0xf, 0x5, false));   // banks 0, 2: from lane + 4
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const float keep = b1 ? b[2 + j] : b[j], send = b1 ? b[j] : b[2 + j]; c[j] = keep + ntp_dpp<0x4E>(send); }
    const float keep = b0 ? c[1] : c[0], send = b0 ? c[0] : c[1];
    return keep + ntp_dpp<0xB1>(send);
}

// Operand prologue of the producers (bf16 A only): A' = relu(A * scale[k] + shift[k]) * inv_keep * keep[row][k] -- BatchNorm-normalise +
// ReLU + Dropout of the previous layer applied on the way into LDS (reference src/models/encoders.py:32-38 in front of the second
// nn.Linear of EncoderB, :35).  The same arithmetic as SrcBnReluDrop (gemm_src.h): scale and shift pre-multiplied by inv_keep in an LDS
// table, keep bytes as floats.  MASK = false: eval mode (no dropout).  K % 64 == 0, K <= 512.
struct NtpProNone {                  // no prologue: the members only keep the (never executed) prologue code well-formed
    static constexpr bool ON = false, MASK = false;
    const float* scale = nullptr; const float* shift = nullptr; const uint8_t* mask = nullptr; long ldm = 0; float inv_keep = 1.f;
    bf16* out = nullptr; long ldo = 0;
    BnFin fin = BnFin{};
};
template <bool MASK_> struct NtpProBn {
    static constexpr bool ON = true, MASK = MASK_;
    const float* scale; const float* shift; const uint8_t* mask; long ldm; float inv_keep;
    bf16* out; long ldo;              // optional: the operand after the prologue, [M][ldo] (written by the column tile 0 workgroups)
    BnFin fin;                        // optional (sum != nullptr): mmvae_bn_finalize of the operand's producer folded in (gemm_src.h)
};
__device__ __forceinline__ void ntp_ld8(f32x2& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void ntp_wait_set(f32x4 (&s)[4], f32x2 (&m)[4]) {
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void ntp_wait_set(f32x4 (&s)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]) : "n"(N) : "memory");
}

template <typename Cfg, typename AT, typename Epi, typename Pro = NtpProNone>
__global__ __launch_bounds__(64 * Cfg::NWAVES, Cfg::NWAVES / 4)
void gemm_ntp_kernel(const AT* __restrict__ A, long lda, const bf16* __restrict__ W, long ldw, int M, int N, int K, int gx, int gy, Epi epi, Pro pro)
{
    typedef bf16 CT;
    typedef Mma<CT>::frag frag;
    typedef typename Epi::out_t OT;
    typedef EpiCols<sizeof(OT) == 2> EC;
    constexpr int NT = Cfg::NT, NH = Cfg::NH, BM = Cfg::BM, BN = Cfg::BN, WC = Cfg::WC;
    constexpr bool AF = sizeof(AT) == 4;
    constexpr bool PRO = Pro::ON, PMASK = Pro::MASK;
    static_assert(!PRO || !AF, "the operand prologue is for bf16 A");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* wgstat = (float*)(smem + Cfg::OFF_STAT);
    float* const paux = (float*)(smem + Cfg::OFF_SCR);          // PRO: [2][512] floats -- scale * inv_keep, shift * inv_keep
    float* const ecol2 = (float*)(smem + Cfg::OFF_ECOL);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = (K + 63) >> 6;
    const int ntiles = ((gx + 7) / 8) * 8 * gy;
    // tile id -> (row tile, column tile): ids that differ by 8 run on one XCD and are the column tiles of one row tile (the second
    // reader of an A tile finds it in that XCD's L2), as in gemm_nt2.h
    auto tile_rc = [&](int T, int& rt, int& ct) __attribute__((always_inline)) { const int slot = T >> 3; ct = slot % gy; rt = (slot / gy) * 8 + (T & 7); };
    // The grid is a multiple of 8, so every tile of a workgroup has the same T & 7 and its row tiles only grow: past the first row
    // tile >= gx (the padding of gx to a multiple of 8) the list is over.  No loop in here: with one, the compiler's wait-count
    // analysis of the producers' K loops turned conservative and drained every load in flight at each step.
    auto next_tile = [&](int T) __attribute__((always_inline)) {
        T += gridDim.x;
        if (T >= ntiles) return -1;
        int rt, ct; tile_rc(T, rt, ct);
        return rt < gx ? T : -1;
    };
    const int T0 = blockIdx.x;
    { int rt, ct; tile_rc(T0, rt, ct); if (rt >= gx) return; }
    int G = 0;                                                      // K steps of this workgroup over all its tiles
    for (int T = T0; T >= 0; T = next_tile(T)) G += nk;

    if constexpr (PRO) {
        if (pro.fin.sum) {
            for (int i = tid; i < K; i += 64 * Cfg::NWAVES) { float sc, sh; pro.fin.column(i, blockIdx.x == 0, sc, sh); paux[i] = sc * pro.inv_keep; paux[512 + i] = sh * pro.inv_keep; }
        } else {
            for (int i = tid; i < K; i += 64 * Cfg::NWAVES) { paux[i] = pro.scale[i] * pro.inv_keep; paux[512 + i] = pro.shift[i] * pro.inv_keep; }
        }
        __syncthreads();                                  // the producers stage their first K steps in front of the loop's first barrier
    }
    // W: the issuing waves' WPW pieces (8 LDS rows each) of the column tile per K step.  LDS row x of the tile's W block holds W row
    // (x & ~63) + EpiCols::wrow(x & 63) (the epilogue's column order), chunk (position ^ (x & 7)): a part per wave (scalar), a
    // compile-time part per piece (scalar multiply) and a lane part in ONE VGPR.  Issued by the consumer waves or by the producers (NtpCfg::WCONS).
    constexpr bool WCONS = Cfg::WCONS || (Pro::ON && NTP_PRO_WCONS);   // with the operand prologue the producers are busier still (3 400 against 1 160 cycles per K step)
    constexpr int WPW = BN / 8 / (WCONS ? Cfg::NCONS : Cfg::NPA);      // pieces per issuing wave and K step: 4 / 4 (consumers), 8 / 4 (producers)
    constexpr bool PAIRC = EC::G == 8;
    const int widx = WCONS ? wid : wid - Cfg::NCONS;                   // this wave's index among the issuing waves (the other role never issues)
    const int p0 = widx * WPW;
    const int wrow_wave = PAIRC ? (p0 >> 3) * 64 + 32 * ((p0 & 7) >> 2) : 8 * p0;
    auto wrow_piece = [](int i) constexpr { return PAIRC ? 32 * (i >> 2) + 16 * (i & 1) + 4 * ((i >> 1) & 1) : 8 * i; };
    const unsigned wlds0 = lds_addr_of(smem) + Cfg::OFF_W + p0 * 1024;
    int wT = T0, wkt = 0, wcol0, wslot = 0; { int rt, ct; tile_rc(T0, rt, ct); wcol0 = ct * BN; }
    auto issue_w = [&]() __attribute__((always_inline)) {           // DMA of the next W step into its ring slot (past the last step: the last one again)
        const int ln = ntp_lane_id(), l3 = ln >> 3;
        const unsigned wlane = ((unsigned)(PAIRC ? 8 * (l3 >> 2) + (l3 & 3) : l3) * (unsigned)ldw + (unsigned)(((ln & 7) ^ l3) * 8)) * 2u;
        const bf16* sb = W + ((long)(wcol0 + wrow_wave) * ldw + (NTP_ABL == 1 ? 0 : wkt * 64));
        const unsigned ls = wlds0 + wslot * Cfg::W_SLOT;
#pragma unroll
        for (int i = 0; i < WPW; ++i) ntp_dma16(sb + (long)wrow_piece(i) * ldw, wlane, ls, i * 1024);
        wslot = wslot == Cfg::RW - 1 ? 0 : wslot + 1;
        if (wkt + 1 < nk) { ++wkt; return; }
        const int Tn = next_tile(wT);
        if (Tn < 0) return;
        wT = Tn; wkt = 0; int rt, ct; tile_rc(Tn, rt, ct); wcol0 = ct * BN;
    };

    if (wid < Cfg::NCONS) {
        // ---------------------------------------------------------------------------------- MFMA consumers
        const int wr = wid / WC, wc = wid % WC;
        const int li = lane & 15, lg = lane >> 4;
        const bool want_stats = Epi::STATS && (epi.stat1 != nullptr || epi.stat2 != nullptr);
        for (int i = tid; i < Cfg::WR * 2 * BN; i += 64 * Cfg::NCONS) wgstat[i] = 0.f;     // ordered by the first barrier
        f32x4 acc[NH][4][4];
        // fragment addresses: row * 128 + ((chunk ^ (row & 7)) << 4); row & 7 == li & 7 for every fragment of the lane
        // (the lane's four address parts -- sw0, sw1, aoff, woff -- are formed inside every K step from the hardware lane id: kept in
        // registers they were spilled around the epilogue, and the reload's vmcnt(0) landed in front of the first fragment read of
        // EVERY step, where it also drained the W DMA in flight)
        int sw0, sw1, aoff, woff;
        // One K step = 2 fragment steps (s = 0, 1) of 16 MFMAs (4 A + 4 W fragments each)
        auto rdA = [&](frag (&af)[4], int sa, int swz_) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = *(const frag*)(smem + sa + aoff + m * 16 * ROW_BYTES + swz_);
        };
        auto rdB = [&](frag (&bf)[4], int sw, int swz_, int nb) __attribute__((always_inline)) {
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = *(const frag*)(smem + sw + woff + (nb * 4 + n) * 16 * ROW_BYTES + swz_);
        };
        auto mmaU = [&](const frag (&af)[4], const frag (&bf)[4], int nb) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (NTP_ABL == 3) asm volatile("" :: "v"(bf[n]), "v"(af[m]));
                    else Mma<CT>::mma(acc[nb][m][n], bf[n], af[m]);      // swapped operands (gemm_nt_epi.h)
                }
        };
        float st_sum1 = 0.f, st_sum2 = 0.f;          // this lane's column (see the epilogue) over the tiles of the current column tile
        auto dump_stats = [&]() __attribute__((always_inline)) {           // -> the workgroup's sums; every (wr, column) entry has ONE owner lane
            if (want_stats) {
                const int ln = ntp_lane_id(), li_ = ln & 15, lg_ = ln >> 4;
                const int cl = wc * 16 * NT + 32 * (li_ >> 3) + 8 * lg_ + (li_ & 7);
                wgstat[(wr * 2 + 0) * BN + cl] += st_sum1;
                wgstat[(wr * 2 + 1) * BN + cl] += st_sum2;
                st_sum1 = 0.f; st_sum2 = 0.f;
            }
        };
        auto flush_stats = [&](int ct) __attribute__((always_inline)) {        // the workgroup's column sums -> the f64 accumulators; all consumers are past a barrier
            if (want_stats && tid < BN) {
                const int c = ct * BN + tid;
                float v1 = 0.f, v2 = 0.f;
#pragma unroll
                for (int r = 0; r < Cfg::WR; ++r) { v1 += wgstat[(r * 2 + 0) * BN + tid]; v2 += wgstat[(r * 2 + 1) * BN + tid]; }
#pragma unroll
                for (int r = 0; r < Cfg::WR; ++r) { wgstat[(r * 2 + 0) * BN + tid] = 0.f; wgstat[(r * 2 + 1) * BN + tid] = 0.f; }
                if (c < N) {
                    if (epi.stat1) unsafeAtomicAdd(epi.stat1 + c, (double)v1);
                    if (epi.stat2) unsafeAtomicAdd(epi.stat2 + c, (double)v2);
                }
            }
        };
        // The epilogue of a tile runs AFTER the barrier that follows its last K step (and before the first step of the next tile is
        // multiplied): at that barrier the producers learn that the tile's last ring slot is free and spend the epilogue staging the
        // next step and issuing its loads.  The per-column constants therefore exist in two copies (tile parity), and a change of
        // column tile flushes the statistics one barrier later, when every consumer has left the old tile's epilogue.
        auto epilogue = [&](const int row0, const int col0, const float* ecol) __attribute__((always_inline)) {
                // lane-derived values of the epilogue are formed HERE from an opaque copy of the lane id: hoisted out of the K loop
                // (as the compiler did) they stayed live across it and the fragment addresses spilled to scratch inside the loop
                const int lane_e = ntp_lane_id();
                const int li = lane_e & 15, lg = lane_e >> 4;
                // Every tile is interior and the 2-byte output rows are whole 128-byte lines (ntp_dispatch only takes such problems -- all
                // of the forward first layers at B = 65 536; the tile kernels keep the edge cases): no bounds selects, no branches; two
                // accumulators -> one v_cvt_pk_bf16_f32 = the packed store word, expanded again (2 VALU) for the statistics of the ROUNDED
                // value; packed f32 adds / fmas; the half swap between lanes li and li ^ 8 is a DPP row_ror:8 move, not a ds_bpermute.
                auto fast = [&](auto ACTC) __attribute__((always_inline)) {
                    constexpr int ACT = decltype(ACTC)::value;
                    const bool lowl = li < 8;
                    OT* crow = epi.C + (long)(row0 + wr * 64 + (li & 7)) * epi.ldc + (col0 + wc * 16 * NT + 32 * (li >> 3) + 8 * lg);
                    const long ld16 = 16 * epi.ldc, ld8 = 8 * epi.ldc;
#pragma unroll
                    for (int hh = 0; hh < NH; ++hh) {
                        const int cwb = wc * 16 * NT + hh * 64;
                        float bq[2][8];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x4 b0 = *(const f32x4*)(ecol + cwb + 32 * h + 8 * lg), b1 = *(const f32x4*)(ecol + cwb + 32 * h + 8 * lg + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { bq[h][e] = b0[e]; bq[h][4 + e] = b1[e]; }
                        }
                        f32x2 s1[2][4], s2[2][4];              // pairs of adjacent columns: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) { s1[h][qd] = f32x2{0.f, 0.f}; s2[h][qd] = f32x2{0.f, 0.f}; }
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            uint32_t pk[2][4];
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int qd = 0; qd < 4; ++qd) {
                                    const int e = 2 * qd;
                                    const f32x4 a4 = acc[hh][m][2 * h + (e >> 2)];
                                    f32x2 x = f32x2{a4[e & 3], a4[(e & 3) + 1]} + f32x2{bq[h][e], bq[h][e + 1]};
                                    if (ACT == 1) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); }
                                    else if (ACT == 2) { x[0] = __builtin_amdgcn_rcpf(1.f + __expf(-x[0])); x[1] = __builtin_amdgcn_rcpf(1.f + __expf(-x[1])); }
                                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                                    const bf16x2 t = {(bf16)x[0], (bf16)x[1]};
                                    const uint32_t w = __builtin_bit_cast(uint32_t, t);
                                    if (Epi::STATS) {
                                        const f32x2 r = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                                        s1[h][qd] += r;
                                        s2[h][qd] = __builtin_elementwise_fma(r, r, s2[h][qd]);
                                    }
                                    pk[h][qd] = w;
                                }
                            uint32_t st0[4], st1[4];
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) {
                                const uint32_t send = lowl ? pk[1][qd] : pk[0][qd];
                                const uint32_t got = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x128, 0xf, 0xf, true);      // row_ror:8 == lane li ^ 8
                                st0[qd] = lowl ? pk[0][qd] : got;
                                st1[qd] = lowl ? got : pk[1][qd];
                            }
                            OT* cp = crow + m * ld16 + hh * 64;
                            *(uint4*)cp = uint4{st0[0], st0[1], st0[2], st0[3]};
                            *(uint4*)(cp + ld8) = uint4{st1[0], st1[1], st1[2], st1[3]};
                        }
                        if (Epi::STATS && want_stats) {
                            // value 8 h + 2 qd + c of this lane = column cwb + 32 h + 8 lg + 2 qd + c, summed over the lane's 4 rows; the 16 lanes
                            // of the DPP row (li = 0..15, one lg) hold the other rows of the same columns: lane li ends with the total of value li.
                            // Kept in two registers over ALL tiles of the column tile (dump_stats), not per tile through LDS.
                            static_assert(NH == 1, "statistics registers: one 64-column block per consumer wave");
                            float v1[16], v2[16];
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                                    for (int c = 0; c < 2; ++c) { v1[8 * h + 2 * qd + c] = s1[h][qd][c]; v2[8 * h + 2 * qd + c] = s2[h][qd][c]; }
                            st_sum1 += ntp_row16_reduce_scatter(v1, li);
                            st_sum2 += ntp_row16_reduce_scatter(v2, li);
                        }
                    }
                };
                const int act = epi.act_code();
                if (NTP_ABL == 4) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[NH - 1][3][3])); }
                else if (act == 0) fast(IntC<0>{}); else if (act == 1) fast(IntC<1>{}); else fast(IntC<2>{});
        };
        int T = T0, kt = 0, w3 = 0, a3 = 0, ct_prev = -1, ct_flush = -1, row0 = 0, col0 = 0, tpar = 0;
        int e_row0 = 0, e_col0 = 0; const float* e_ecol = ecol2; bool e_pend = false, e_dump = false;
#ifdef MM_STAMP
        unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
        NTP_T(t_begin);
#endif
        if (WCONS) { issue_w(); issue_w(); }                  // W of steps 0 and 1
        for (int g = 0; g < G; ++g) {
            NTP_T(tb0);
            // own share of W(g) has landed: everything but the WPW youngest operations (= W(g+1)); this also retires the stores of the
            // epilogue that ran one iteration ago
            if (WCONS) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WPW) : "memory");
            ntp_bar();
#ifdef MM_STAMP
            unsigned long long tb1 = __builtin_readcyclecounter();
#endif
            NTP_ACC(0, tb1 - tb0); NTP_ACC(3, 1);
            if (e_pend) {
                NTP_T(te0);
                epilogue(e_row0, e_col0, e_ecol);
                if (e_dump) dump_stats();
                e_pend = false;
#ifdef MM_STAMP
                NTP_T(te1);
                NTP_ACC(2, te1 - te0); NTP_ACC(4, 1);
                tb1 = te1;
#endif
            }
            if (kt == 1 && ct_flush >= 0) { flush_stats(ct_flush); ct_flush = -1; }     // every consumer is past the old column tile's last epilogue
            if (kt == 0) {
                int rt, ct; tile_rc(T, rt, ct);
                row0 = rt * BM; col0 = ct * BN;
                if (ct != ct_prev && ct_prev >= 0) ct_flush = ct_prev;
                ct_prev = ct;
                if (Epi::NCOL > 0 && tid < BN) epi.fill(ecol2 + tpar * BN, BN, tid, col0 + tid, N);      // this copy's last reader finished >= 1 barrier ago
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[h][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                // whatever the compiler loaded on this path (bias, scratch reloads) is complete HERE, once per tile, and known to be: left
                // pending, its wait lands in the K step's common code as a vmcnt(0) in front of the first fragment read of EVERY step
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
            }
            if (WCONS) issue_w();                                 // W(g+2) into the slot every consumer left one barrier ago; in flight across two barriers
            {
                const int ln = ntp_lane_id(), li_ = ln & 15, lg_ = ln >> 4;
                sw0 = (lg_ ^ (li_ & 7)) << 4; sw1 = ((4 + lg_) ^ (li_ & 7)) << 4;
                aoff = (wr * 64 + li_) * ROW_BYTES; woff = Cfg::OFF_W + (wc * 16 * NT + li_) * ROW_BYTES;
            }
            const int sa = a3 * Cfg::A_SLOT, sw = w3 * Cfg::W_SLOT;
            frag a0[4], a1[4], b0[4], b1[4];
            if constexpr (NH == 2) {
                rdA(a0, sa, sw0); rdB(b0, sw, sw0, 0);
                rdB(b1, sw, sw0, 1); mmaU(a0, b0, 0);
                rdA(a1, sa, sw1); rdB(b0, sw, sw1, 0); mmaU(a0, b1, 1);
                rdB(b1, sw, sw1, 1); mmaU(a1, b0, 0);
                mmaU(a1, b1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            } else if constexpr (Cfg::NWAVES == 12) {
                // two consumer waves per SIMD and 168 registers: ONE fragment set (32 registers) -- the partner wave's MFMAs cover this
                // wave's LDS round trip; with both fragment steps in registers four accumulator tiles were spilled inside the K loop
                rdA(a0, sa, sw0); rdB(b0, sw, sw0, 0); mmaU(a0, b0, 0);
                rdA(a0, sa, sw1); rdB(b0, sw, sw1, 0); mmaU(a0, b0, 0);
            } else {
                rdA(a0, sa, sw0); rdB(b0, sw, sw0, 0);
                rdA(a1, sa, sw1); rdB(b1, sw, sw1, 0); mmaU(a0, b0, 0);
                mmaU(a1, b1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
#ifdef MM_STAMP
            asm volatile("s_nop 0" :: "v"(acc[0][0][0][0]), "v"(acc[NH - 1][3][3][3]));      // the MFMAs of this step are issued before the stamp
            NTP_T(tb2);
            NTP_ACC(1, tb2 - tb1);
#endif
            w3 = (w3 == Cfg::RW - 1) ? 0 : w3 + 1;
            a3 = (a3 == Cfg::RA - 1) ? 0 : a3 + 1;
            if (++kt == nk) {
                e_pend = true; e_row0 = row0; e_col0 = col0; e_ecol = ecol2 + tpar * BN;
                tpar ^= 1;
                kt = 0;
                T = next_tile(T);
                e_dump = true;                      // the column tile's last row tile: the register sums go to the workgroup's sums
                if (T >= 0) { int rt_, ct_; tile_rc(T, rt_, ct_); e_dump = ct_ != ct_prev; }
            }
        }
        ntp_bar();
        if (e_pend) {
            NTP_T(te0);
            epilogue(e_row0, e_col0, e_ecol);
            dump_stats();
#ifdef MM_STAMP
            NTP_T(te1);
            NTP_ACC(2, te1 - te0); NTP_ACC(4, 1);
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the W pieces issued past the last step have landed: no DMA outlives the workgroup
        ntp_bar();                                   // every consumer has added its last tile's column sums
        flush_stats(ct_prev);
#ifdef MM_STAMP
        if (tid == 0 && (blockIdx.x & 15) == 3) {
            NTP_T(t_end);
            for (int i = 0; i < 5; ++i) atomicAdd(&mm_ntp_stamps[i], st_acc[i]);
            atomicAdd(&mm_ntp_stamps[5], t_end - t_begin);
            atomicAdd(&mm_ntp_stamps[6], 1ull);
        }
#endif
    } else {
        // ---------------------------------------------------------------------------------- A producers
        constexpr int EL = 16 / (int)sizeof(AT);                    // A elements per lane and load (16 bytes)
        constexpr int LPR = 64 / EL;                                // lanes per row of a K step: 16 (fp32) / 8 (bf16)
        constexpr int RPI = 64 / LPR;                               // rows per load instruction: 4 / 8
        constexpr int AI = Cfg::BM / Cfg::NPA / RPI;                // loads per wave and K step: 8 / 4
        const int pw = wid - Cfg::NCONS;
        const int prow0 = pw * (BM / Cfg::NPA);
        const int q = lane % LPR, rsub = lane / LPR;
        // fp32: lane q holds floats 4q..4q+3 of the step = half (q & 1) of LDS chunk q >> 1; bf16: lane q holds chunk q.
        // The last 16-byte piece that lies inside a row: pieces past it re-read it (they only meet zero weights).  fp32 rows with
        // K % 4 = r != 0: the piece that straddles the row end is fetched from K - 4 and rotated by 4 - r elements in stage().
        const int kmaxv = AF ? K - 4 : ((K + 7) & ~7) - 8;
        const int krem = AF ? (K & 3) : 0;
        const bool tail_lane = krem != 0 && q == ((K >> 2) & 15);
        // Per wave-instruction the issue path wants ONE instruction: a scalar base that moves with (tile, K step) and a lane offset
        // per load that only changes with the tile (a dependent chain of 13 scalar instructions per load -- 64-bit row multiply,
        // clamp, add -- was measured at ~60 cycles per load).
        // rows of load i: row0 + prow0 + RPI i + rsub; M % RPI == 0 (dispatch), so the rows of a load are inside the matrix or all past
        // its end -- then it re-reads the tile's first rows (finite data for rows that are never stored)
        struct It { int T, kt, row0; };
        It it; it.T = T0; it.kt = 0; { int rt, ct; tile_rc(T0, rt, ct); it.row0 = rt * BM; }
        unsigned voff[AI], moff[PMASK ? AI : 1];
        auto set_voff = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int rl = prow0 + RPI * i;                      // wave-uniform
                voff[i] = ((unsigned)((it.row0 + rl < M ? rl : 0) + rsub) * (unsigned)lda + (unsigned)(q * EL)) * (unsigned)sizeof(AT);
                if constexpr (PMASK) moff[i] = (unsigned)((it.row0 + rl < M ? rl : 0) + rsub) * (unsigned)pro.ldm + (unsigned)(q * EL);     // one keep byte per element
            }
        };
        set_voff();
        auto adv = [&]() __attribute__((always_inline)) {              // past the last step the iterator stays where it is (the extra loads re-read valid data)
            if (it.kt + 1 < nk) { ++it.kt; return; }
            const int Tn = next_tile(it.T);
            if (Tn < 0) return;
            it.T = Tn; it.kt = 0; int rt, ct; tile_rc(Tn, rt, ct);
            const bool edge = it.row0 + BM > M || rt * BM + BM > M;     // the offsets only differ on the matrix's last row tile
            it.row0 = rt * BM;
            if (edge) set_voff();
        };
        auto load = [&](f32x4 (&s)[AI]) __attribute__((always_inline)) {
            const unsigned ko = (unsigned)(min((NTP_ABL == 2 ? 0 : it.kt * 64) + q * EL, kmaxv) - q * EL) * (unsigned)sizeof(AT);
            const char* sbase = (const char*)A + (size_t)(NTP_ABL == 2 ? (blockIdx.x & 7) * BM : it.row0) * (size_t)lda * sizeof(AT);
#pragma unroll
            for (int i = 0; i < AI; ++i) ntp_ld16(s[i], voff[i] + ko, sbase);
        };
        auto load_mask = [&](f32x2 (&m)[PMASK ? AI : 1], int kt_, int row0_) __attribute__((always_inline)) {      // the keep bytes of the same rows and K step
            if constexpr (PMASK) {
                const char* mbase = (const char*)pro.mask + (size_t)row0_ * (size_t)pro.ldm;
#pragma unroll
                for (int i = 0; i < AI; ++i) ntp_ld8(m[i], moff[i] + (unsigned)(kt_ * 64), mbase);
            }
        };
#ifdef MM_STAMP
        unsigned long long stmp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stprev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        int skt = 0, sslot = 0, staged = 0;                          // K step (inside its tile) / ring slot / index of the step that is staged next
        int s_T = T0, s_row0, s_ct; { int rt_, ct_; tile_rc(T0, rt_, ct_); s_row0 = rt_ * BM; s_ct = ct_; }      // PRO + out: the tile that is being staged
        auto stage = [&](f32x4 (&s)[AI]) __attribute__((always_inline)) {
            unsigned char* sA = smem + sslot * Cfg::A_SLOT;
            const bool rot = krem != 0 && skt == nk - 1;              // wave-uniform
            skt = skt + 1 == nk ? 0 : skt + 1;
            sslot = sslot == Cfg::RA - 1 ? 0 : sslot + 1;
            ++staged;
            bf16x4 o[AI];
            if constexpr (AF) {
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    f32x4 v = s[i];
                    if (rot) {
                        const f32x4 t = krem == 1 ? f32x4{v[3], v[0], v[1], v[2]} : krem == 2 ? f32x4{v[2], v[3], v[0], v[1]} : f32x4{v[1], v[2], v[3], v[0]};
                        v = tail_lane ? t : v;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[i][e] = (bf16)v[e];
                }
            }
#ifdef MM_STAMP
            asm volatile("s_memtime %0" : "=s"(stmp[3]) :: "memory");
#endif
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int r = prow0 + RPI * i + rsub;
                if constexpr (AF) *(bf16x4*)(sA + r * ROW_BYTES + (((q >> 1) ^ (r & 7)) << 4) + (q & 1) * 8) = o[i];
                else *(f32x4*)(sA + r * ROW_BYTES + ((q ^ (r & 7)) << 4)) = s[i];
            }
        };
        // PRO: the lane's 8 elements of every row are columns 64 kt + 8 q .. + 7 -- one set of 8 (scale, shift) pairs per K step
        auto stage_pro = [&](f32x4 (&s)[AI], f32x2 (&ms)[PMASK ? AI : 1]) __attribute__((always_inline)) {
            unsigned char* sA = smem + sslot * Cfg::A_SLOT;
            const float* ap = paux + skt * 64 + 8 * q;
            const f32x4 sc0 = *(const f32x4*)ap, sc1 = *(const f32x4*)(ap + 4), sh0 = *(const f32x4*)(ap + 512), sh1 = *(const f32x4*)(ap + 516);
            // the tile this step belongs to (the staging position trails the load iterator by SETS steps): its rows for the copy to pro.out
            const bool keep_out = pro.out != nullptr && s_ct == 0;
            bf16* const orow = pro.out + (long)s_row0 * pro.ldo + skt * 64 + 8 * q;
            skt = skt + 1 == nk ? 0 : skt + 1;
            if (skt == 0 && s_T >= 0) { s_T = next_tile(s_T); if (s_T >= 0) { int rt_, ct_; tile_rc(s_T, rt_, ct_); s_row0 = rt_ * BM; s_ct = ct_; } }
            sslot = sslot == Cfg::RA - 1 ? 0 : sslot + 1;
            ++staged;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int r = prow0 + RPI * i + rsub;
                const uint4 u = __builtin_bit_cast(uint4, s[i]);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
                uint32_t mk[2] = {0x01010101u, 0x01010101u};
                if constexpr (PMASK) { const uint2 m2 = __builtin_bit_cast(uint2, ms[i]); mk[0] = m2.x; mk[1] = m2.y; }
                typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
                bf16x8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
                    const float sc = e < 4 ? sc0[e] : sc1[e - 4], sh = e < 4 ? sh0[e] : sh1[e - 4];
                    const float v = fmaxf(x * sc + sh, 0.f);
                    o[e] = (bf16)(v * (float)((mk[e >> 2] >> (8 * (e & 3))) & 0xffu));
                }
                *(bf16x8_t*)(sA + r * ROW_BYTES + ((q ^ (r & 7)) << 4)) = o;
                if (keep_out) *(bf16x8_t*)(orow + (long)r * pro.ldo) = o;      // M % 128 == 0 (dispatch): every row of the tile exists
            }
        };
        // One in-order queue per wave: per iteration g it takes [W(g+2) x WPW, A(g+2+SETS) x AI].  Before barrier g+1 the W of step g+1
        // (issued one iteration earlier) must have landed: everything but the 2 AI + WPW youngest operations -- which also completes every
        // A load issued before iteration g-1, so two A sets stay in flight across a barrier and a third one while its step is staged:
        // SETS = 3.  The waits are counted by hand and every operation is issued unconditionally (past the last step the iterators stay
        // on it).  Measured and dropped: the W DMA in the consumer waves with 4-6 deep A sets; odd producer waves issuing
        // before they stage; a raised producer priority (twice as slow); nt / sc1 loads of A.
        constexpr int SETS = 3;
        constexpr int PW = WCONS ? 0 : WPW;                           // W pieces in this wave's queue per K step
        constexpr int LPS = AI * (PMASK ? 2 : 1);                     // loads per set: A pieces (+ keep-byte pieces)
        constexpr int N_SET = (SETS - 1) * LPS + 2 * PW;              // operations younger than the loads of the set that is staged next
        constexpr int N_W = 2 * LPS + PW;                             // operations younger than the W pieces of the next step
#ifdef MM_STAMP
        // s_memtime at 8 points of an iteration, collected WITHOUT waiting (ntp_bar()'s lgkmcnt(0) retires them): a waiting stamp would
        // serialise the LDS writes and the issue streams it is supposed to time
        unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NTP_ST(i) asm volatile("s_memtime %0" : "=s"(stmp[i]) :: "memory")
#define NTP_STACC() for (int i_ = 0; i_ < 7; ++i_) st_acc[i_] += stprev[i_ + 1] - stprev[i_]
#define NTP_STEND() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int i_ = 0; i_ < 8; ++i_) stprev[i_] = stmp[i_]
#else
#define NTP_ST(i)
#define NTP_STACC()
#define NTP_STEND()
#endif
#define NTP_WAITW(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
        int g = 0;
        bool done = false;
        f32x4 S[SETS][AI];
        f32x2 MS[SETS][PMASK ? AI : 1];
        auto loadset = [&](int u) __attribute__((always_inline)) { const int kt_ = it.kt, r0_ = it.row0; load(S[u]); load_mask(MS[u], kt_, r0_); adv(); };
        auto waitset = [&](int u) __attribute__((always_inline)) {       // u is a constant after unrolling, as in S[u]
            if constexpr (PMASK) ntp_wait_set<N_SET>(S[u], MS[u]); else ntp_wait_set<N_SET>(S[u]);
        };
        auto stageset = [&](int u) __attribute__((always_inline)) { if constexpr (PRO) stage_pro(S[u], MS[u]); else stage(S[u]); };
#pragma unroll
        for (int u = 0; u < SETS; ++u) loadset(u);
        if (!WCONS) { issue_w(); issue_w(); }
        waitset(0); stageset(0); loadset(0);
        waitset(1); if (staged < G) stageset(1); loadset(1);
        if (!WCONS) { NTP_WAITW(N_W) }
        while (!done) {             // step g+2 comes from set (g + 2) % SETS: the index is a compile-time constant in the unrolled body
#pragma unroll
            for (int u = 0; u < SETS; ++u) {
                NTP_ST(0); ntp_bar(); NTP_STACC(); NTP_ST(1);
                if (++g == G) { done = true; break; }
                waitset((u + 2) % SETS); NTP_ST(2);
                if (staged < G) stageset((u + 2) % SETS);
                NTP_ST(4); if (!WCONS) issue_w(); NTP_ST(5);
                loadset((u + 2) % SETS); NTP_ST(6);
                if (!WCONS) { NTP_WAITW(N_W) }
                NTP_ST(7); NTP_STEND();
            }
        }
#undef NTP_WAITW
#undef NTP_ST
#undef NTP_STACC
#undef NTP_STEND
#undef NTP_ASTEP
#ifdef MM_STAMP
        if (lane == 0 && pw == 0 && (blockIdx.x & 15) == 3) for (int i = 0; i < 7; ++i) atomicAdd(&mm_ntp_stamps[8 + i], st_acc[i]);
#endif
        ntp_bar();
        ntp_bar();                                   // the consumers' last epilogue / statistics flush
    }
}

template <typename Cfg, typename AT, typename Epi, typename Pro = NtpProNone>
static int launch_ntp(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st, const Pro& pro = Pro{}) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_ntp_kernel<Cfg, AT, Epi, Pro>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::TOTAL);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int gx = (M + Cfg::BM - 1) / Cfg::BM, gy = (N + Cfg::BN - 1) / Cfg::BN;
    const int ntiles = ((gx + 7) / 8) * 8 * gy;
    int grid = 256;                                                  // one 8-wave workgroup per CU
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((gemm_ntp_kernel<Cfg, AT, Epi, Pro>), dim3(grid), dim3(64 * Cfg::NWAVES), Cfg::TOTAL, st,
                       (const AT*)A, lda, (const bf16*)W, ldw, M, N, K, gx, gy, epi, pro);
    MM_CHECK_LAUNCH();
    return 0;
}

}  // namespace mm
