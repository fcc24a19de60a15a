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
#ifdef MM_STAMP
namespace mm { __device__ unsigned long long mm_stamps[12]; }
#endif
#include "gemm_nt2.h"

namespace mm {

// ------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

// LDS: main loop 2 x { A [128][128 B] + W [64*WN][128 B] } (64 KiB at WN=2, 96 KiB at WN=4); the epilogue needs none of it
template <int WN> struct NtLds {
    static constexpr int BUF = (TILE + 64 * WN) * ROW_BYTES;      // one K step of A and W
    static constexpr int STAGE = 2 * BUF;                         // double buffered
    static constexpr int ECOL = STAGE + 4096 + 4 * 64 * WN * 4;   // + BN prologue scale/shift + column-sum scratch
    static constexpr int TOTAL = ECOL + 8 * 64 * WN * 4;           // + per-column constants of the epilogue
};

// WN = 2: 128x128 tile, 4 waves, 2 workgroups per CU.  WN = 4: 128x256 tile, 8 waves, 1 workgroup per CU -- the A tile is
// fetched once for 256 output columns, which halves the L2->CU operand ingest of the N = 256 / 512 layers.
#ifdef MM_STAMP
// Diagnostic build only (make STAMP=1 -> libmmvae_stamp.so, tools/stamp_nt.py): s_memtime stamps at the points of a K step
// where the wave has drained lgkmcnt anyway, summed per wave and added to mm_stamps[] = {reads + mma0 issue, stage (vmcnt
// wait + ds_write), barrier wait, fetch + reads + mma1 issue, K steps, waves, whole-kernel cycles summed over waves}.
#define MM_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define MM_ACC(i, d) st_acc[i] += (d)
#else
#define MM_T(x)
#define MM_ACC(i, d)
#endif

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
    float* ecol = (float*)(smem + NtLds<WN>::ECOL);

    // XCD-aware tile assignment: linear id L runs on XCD L%8 (round-robin dispatch, speed only).
    const int L = blockIdx.x;
    const int slot = L >> 3;
    const int ct = slot % gy;
    const int rt = (slot / gy) * 8 + (L & 7);
    if (rt >= gx) return;
    const int row0 = rt * TILE, col0 = ct * BN;
#ifdef MM_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, t_first = 0, t_last = 0;
    MM_T(t_begin);
#endif

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WN, wc = wid % WN;

    nt_epilogue_fill_cols<Epi, WN>(ecol, epi, col0, N, tid);       // visible after the first barrier below
    if (Src::NEEDS_AUX) { src.init(aux, tid, 0); __syncthreads(); }

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
            // the W row that lands in LDS row r: the epilogue's column order inside the wave's 64 columns (gemm_nt_epi.h)
            const int wr_ = (r & ~63) + EpiCols<sizeof(typename Epi::out_t) == 2>::wrow(r & 63);
            rb[i].v = *(const decltype(rb[i].v)*)(W + ((unsigned)(col0 + wr_) * (unsigned)ldw + (unsigned)(kt * BK + ch * EPC)));
        }
    };
    const bool a_tail = Src::PAD_TAIL && (K & 7) != 0;        // kernel-uniform
    auto stage = [&](typename Src::Raw (&ra)[A_PER], Chunk<CT> (&rb)[4], int kt, int buf) {      // registers -> LDS buffer
        unsigned char* sA = smem + buf * NtLds<WN>::BUF;
        unsigned char* sB = sA + TILE * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            Chunk<CT> o;
            // unmasked unless the rows of a 2-byte A end inside a 16-byte chunk (K % 8 != 0): the pad elements of a caller's
            // buffer may hold anything, and NaN x (zero weight padding) would reach valid outputs.  See gemm_src.h.
            if (a_tail) src.finish(ra[i], row0 + r, kt * BK + ch * EPC, o, aux);
            else src.finish_fast(ra[i], kt * BK + ch * EPC, o, aux);
            *(decltype(o.v)*)(sA + swz(r, ch)) = o.v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + NTH * i, r = c >> 3, ch = c & 7;
            *(decltype(rb[i].v)*)(sB + swz(r, ch)) = rb[i].v;
        }
    };
    // Fragment registers are double-buffered too: the ds_reads of the NEXT fragment step are issued before the MFMAs of the
    // current one (tools/ubench_lds_mfma: reads + MFMAs overlap to the MFMA floor when the reads run one block ahead; issued
    // right before their MFMAs, as the single-set loop did, every fragment step exposed the LDS round trip of all 4 waves).
    frag f0a[4], f0b[4], f1a[4], f1b[4];
    auto rd = [&](frag (&af)[4], frag (&bf)[4], int buf, int s) {           // fragment step s (0/1) of an LDS buffer
        const unsigned char* sA = smem + buf * NtLds<WN>::BUF;
        const unsigned char* sB = sA + TILE * ROW_BYTES;
        const int ch = s * 4 + (lane >> 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *(const frag*)(sA + swz(wr * 64 + m * 16 + (lane & 15), ch));
#pragma unroll
        for (int n = 0; n < 4; ++n) bf[n] = *(const frag*)(sB + swz(wc * 64 + n * 16 + (lane & 15), ch));
    };
    auto mma = [&](const frag (&af)[4], const frag (&bf)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<CT>::mma(acc[m][n], bf[n], af[m]);     // swapped: transposed accumulator layout (gemm_nt_epi.h)
    };
    // One K step (tile kt in LDS buffer `buf`, its first fragment step already in f0): ONE barrier, in the middle.
    //   reads (kt, s1) -> f1 | MFMA f0 | tile kt+1: registers -> other LDS buffer | barrier | global fetch of a later tile
    //   into the registers just staged | reads (kt+1, s0) -> f0 | MFMA f1
    // Before the barrier every wave has drained its LDS traffic (lgkmcnt(0)), so nobody still reads the buffer the next
    // step overwrites; after it the freshly staged tile is visible.
    // Memory instructions are SPREAD between the MFMAs (2 MFMA, 1 ds_write ... in the first half; 2 MFMA, 1 global load ...
    // in the second) instead of issued as two bursts: a burst of 8 wave-wide loads fills the CU's texture-address queue
    // (64 B/clk: 128 cycles for the burst), the wave blocks at issue and its MFMAs wait behind it.  tools/ubench_stage.hip:
    // 1256 -> 999 cycles per 128x128x64 step and CU for this loop skeleton (no-traffic floor 700, MFMA floor 560).
    constexpr int NMEM = A_PER + 4;                          // loads / LDS writes per lane and K step
    auto kstep = [&](auto& ra_n, auto& rb_n, int kt, int buf, bool has_next, bool do_fetch, int fetch_kt) {
        MM_T(t0);
        rd(f1a, f1b, buf, 1);
        mma(f0a, f0b);
        MM_T(t1);
#ifdef MM_STAMP
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // steady state: the register set about to be staged has landed
        MM_T(t1b);
        MM_ACC(5, t1b - t1);
#endif
        if (has_next) stage(ra_n, rb_n, kt + 1, buf ^ 1);
#ifndef MM_STAMP
#pragma unroll
        for (int j = 0; j < NMEM; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 16 / NMEM, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#endif
#ifdef MM_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        MM_T(t2);
        __syncthreads();
        MM_T(t3);
        if (do_fetch) fetch(ra_n, rb_n, fetch_kt);
        if (has_next) rd(f0a, f0b, buf ^ 1, 0);
        mma(f1a, f1b);
#ifndef MM_STAMP
#pragma unroll
        for (int j = 0; j < NMEM; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 16 / NMEM, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#endif
        MM_T(t4);
#ifdef MM_STAMP
        if (st_acc[4] == 0) t_first = t0;
        t_last = t4;
#endif
        MM_ACC(0, t1 - t0); MM_ACC(1, t2 - t1); MM_ACC(2, t3 - t2); MM_ACC(3, t4 - t3); MM_ACC(4, 1);
    };

    EpiOperands<Epi> eops;
    auto pre = [&]() { nt_epilogue_prefetch<Epi, 0>(eops, epi, row0, col0, M, N, BN, lane, wr, wc); };     // no-op unless the operands are 16-byte addressable

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
        rd(f0a, f0b, 0, 0);
        int kt = 0;
        for (; kt + 4 < nk; kt += 2) {
            kstep(ra1, rb1, kt, 0, true, true, kt + 3);
            kstep(ra0, rb0, kt + 1, 1, true, true, kt + 4);
        }
        const bool four = kt + 3 < nk;                        // 3 or 4 K steps left: kt in LDS, kt+1 / kt+2 in flight
        kstep(ra1, rb1, kt, 0, true, four, kt + 3);
        kstep(ra0, rb0, kt + 1, 1, true, false, 0);
        // epilogue operands: issued before the LAST K step, when both global register sets are dead (any earlier and the
        // 48 operand registers on top of accumulators + both fragment sets + a live set spilled, 150-300 dwords)
        if (!four) pre();
        kstep(ra1, rb1, kt + 2, 0, four, false, 0);
        if (four) { pre(); kstep(ra1, rb1, kt + 3, 1, false, false, 0); }
    } else {
        // Short K (nk <= 4: the 64 / 128 / 256-wide layers and the latent): conditional fetches, so the waits are coarse
        // anyway; ONE fragment set here -- with two, the conditional form spilled (scratch reloads wait vmcnt(0), i.e. for
        // every load in flight: the K = 256 BatchNorm-backward layer ran at 12 % of the HBM roofline).
        auto kstep1 = [&](auto& ra_n, auto& rb_n, int kt, int buf, bool has_next, bool do_fetch, int fetch_kt) {
            rd(f0a, f0b, buf, 0);
            mma(f0a, f0b);
            if (has_next) stage(ra_n, rb_n, kt + 1, buf ^ 1);
            rd(f0a, f0b, buf, 1);
            mma(f0a, f0b);
            if (do_fetch) fetch(ra_n, rb_n, fetch_kt);
            __syncthreads();
        };
        fetch(ra0, rb0, 0);
        if (nk > 1) fetch(ra1, rb1, 1);
        stage(ra0, rb0, 0, 0);
        if (nk > 2) fetch(ra0, rb0, 2);
        __syncthreads();
        if (nk == 1) pre();
        kstep1(ra1, rb1, 0, 0, nk > 1, nk > 3, 3);
        if (nk > 1) { if (nk == 2) pre(); kstep1(ra0, rb0, 1, 1, nk > 2, false, 0); }
        if (nk > 2) { if (nk == 3) pre(); kstep1(ra1, rb1, 2, 0, nk > 3, false, 0); }
        if (nk > 3) { pre(); kstep1(ra1, rb1, 3, 1, false, false, 0); }
    }

    nt_epilogue<CT, Epi, WN>(red, ecol, acc, epi, eops, row0, col0, M, N, tid, lane, wr, wc);
#ifdef MM_STAMP
    if (tid == 0 && (blockIdx.x & 15) == 3) {        // a sample of waves: same-address atomics from every wave cost more than the kernel
        MM_T(t_end);
        for (int i = 0; i < 5; ++i) atomicAdd(&mm_stamps[i], st_acc[i]);
        atomicAdd(&mm_stamps[5], 1ull);
        atomicAdd(&mm_stamps[6], t_end - t_begin);
        atomicAdd(&mm_stamps[7], st_acc[5]);
        atomicAdd(&mm_stamps[8], t_first - t_begin);
        atomicAdd(&mm_stamps[9], t_end - t_last);
    }
#endif
}

#ifdef MM_STAMP
extern "C" int mmvae_debug_stamps(unsigned long long* out8, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(mm::mm_stamps), 12 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    if (reset) { unsigned long long z[12] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(mm::mm_stamps), z, sizeof(z)); }
    return (int)e;
}
#endif

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
// kernel-generation switch (mmvae_set_tuning key 2; initial value from MMVAE_NO_NT2): tests flip it inside one process to compare
// the register-staged and the LDS-DMA generation on the same data
static int g_nt2_on = getenv("MMVAE_NO_NT2") ? 0 : 1;
void tn_wide_enable(int on);             // gemm_tn_wide.hip (mmvae_set_tuning key 4)
int ntp_dispatch(const mmvae_gemm_nt_args* a, hipStream_t st);      // gemm_ntp.hip: the wave-specialised kernel; 1 << 30 = not taken
void ntp_set(int key, int value);        // mmvae_set_tuning keys 8 (on / off), 9 (minimum M)
long g_block_bytes = 1L << 31;          // row-block size for operands of >= 4 GiB (mmvae_set_tuning key 3 sets log2; shared with gemm_tn.hip)
long g_split_bytes = 1L << 32;          // operands of at least this many bytes are processed in row blocks

static inline bool nt_wide_ok(int M, int N) {
    static const bool off = getenv("MMVAE_NO_WIDE_TILES") != nullptr;      // A/B switch
    // the prepared W has ceil128(N) rows: whole 256-column tiles only; one 8-wave workgroup per CU: at least 256 tiles (M >= 32768 at
    // N = 256; a 16 384-row block of a 512-wide layer qualifies too)
    return !off && N % 256 == 0 && ((long)M * (N / 256) >= (long)g_wide_min_m || M >= g_wide_min_m);
}

static int g_bnbwd_stream = getenv("MMVAE_NO_BNBWD_STREAM") ? 0 : 1;          // mmvae_set_tuning key 6
static int g_relu_stream = getenv("MMVAE_NO_RELU_STREAM") ? 0 : 1;            // mmvae_set_tuning key 7
template <typename T> struct IsPlainBf16 { static constexpr bool value = false; };
template <> struct IsPlainBf16<SrcPlain<bf16, bf16, 8>> { static constexpr bool value = true; };

template <typename CT, typename Src, typename Epi>
static int launch_nt(const Src& src, const void* W, long ldw, int M, int N, int K, const Epi& epi, hipStream_t st) {
    if constexpr (sizeof(CT) == 2 && IsPlainBf16<Src>::value) {
        // second-generation kernel (gemm_nt2.h): operands that go into the MFMA as they are, at least two K steps
        const bool off = !g_nt2_on;
        // ... and for epilogues without operands of their own: with a saved activation / keep mask to fetch, the epilogue's loads
        // queue behind the next tile's DMA and its stores in front of the next wait (one in-order vmcnt for everything): measured
        // 5-20 % SLOWER than the first generation there, 5-10 % faster on the plain store epilogues (tools/bench_nt2.py)
        if (!off && K > 64 && !epi.accumulate_requested() && Epi::NEED == 0) {
            static const bool narrow = getenv("MMVAE_NT2_NARROW") != nullptr;      // A/B switch: 128x128 tiles only
            if (!narrow && nt_wide_ok(M, N)) return launch_nt2<Epi, 4>(src.p, src.lda, W, ldw, M, N, K, epi, st);
            return launch_nt2<Epi, 2>(src.p, src.lda, W, ldw, M, N, K, epi, st);
        }
    }
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
        if constexpr (sizeof(CT) == 2 && IsPlainBf16<Src>::value) {
            if (g_relu_stream && g_nt2_on && a->ldh % 4 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->h & 7) == 0 && ((uintptr_t)a->c & 7) == 0) {
                EpiReluMaskStream e{(bf16*)a->c, a->ldc, (const bf16*)a->h, a->ldh};
                return launch_nt2<EpiReluMaskStream, 2>(src.p, src.lda, a->w, a->ldw, a->M, a->N, a->K, e, st);
            }
        }
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
        if constexpr (sizeof(CT) == 2 && IsPlainBf16<Src>::value) {
            // phase 2 (store d + statistics) on plain bf16 operands: row-coalesced LDS epilogue (gemm_nt2.h); a single K step is fine here,
            // this epilogue's kernel starts no DMA for the next tile before it is done
            if (g_bnbwd_stream && g_nt2_on && a->bn_phase == 2 && a->ldh % 4 == 0 && a->ldc % 4 == 0 &&
                (!a->epi_mask || (a->ld_epi_mask % 4 == 0 && ((uintptr_t)a->epi_mask & 3) == 0)) && ((uintptr_t)a->h & 7) == 0 && ((uintptr_t)a->c & 7) == 0) {
                EpiBnBwdStream e{(bf16*)a->c, a->ldc, (const bf16*)a->h, a->ldh, a->epi_mask, a->ld_epi_mask,
                                 a->bn_scale, a->bn_shift, a->bn_mean, a->bn_rstd, a->epi_inv_keep, a->stat1, a->stat2};
                return launch_nt2<EpiBnBwdStream, 2>(src.p, src.lda, a->w, a->ldw, a->M, a->N, a->K, e, st);
            }
        }
        EpiBnBwd<LP, LP> e{(LP*)a->c, a->ldc, (const LP*)a->h, a->ldh, a->epi_mask, a->ld_epi_mask,
                           a->bn_scale, a->bn_shift, a->bn_mean, a->bn_rstd, a->epi_inv_keep, a->bn_coef, a->bn_phase,
                           a->bn_phase == 1 ? nullptr : a->stat1, a->bn_phase == 1 ? nullptr : a->stat2};
        return launch_nt<CT>(src, a->w, a->ldw, a->M, a->N, a->K, e, st);
    }
    case MMVAE_EPI_LOSS_MSE:
    case MMVAE_EPI_LOSS_BCE_LOGIT: {
        // reconstruction loss inside a decoder's last GEMM: bf16 mode, plain bf16 A, second-generation kernel, 128 x 128 tiles
        if constexpr (sizeof(CT) == 2 && IsPlainBf16<Src>::value) {
            if (!a->h || !a->c || !a->stat1 || a->c_dtype != MMVAE_BF16 || a->accumulate || a->K <= 64) return MMVAE_ERR_ARG;
            if (a->ldc % 8 || ((uintptr_t)a->c & 15) || ((uintptr_t)a->h & 3)) return MMVAE_ERR_ARG;
            if ((long)a->M * a->ldh * 4 >= (1L << 40)) return MMVAE_ERR_ARG;
            const uintptr_t hp = (uintptr_t)a->h;
            const int vt = (a->ldh % 4 == 0 && a->N % 4 == 0 && (hp & 15) == 0) ? 4 : (a->ldh % 2 == 0 && a->N % 2 == 0 && (hp & 7) == 0) ? 2 : 1;
            const bool mse = a->epilogue == MMVAE_EPI_LOSS_MSE;
#define MM_LOSS_EPI(MODE, VT) { EpiLoss<MODE, VT> e{(bf16*)a->c, a->ldc, (const float*)a->h, a->ldh, a->bias, a->stat1}; \
                                return launch_nt2<EpiLoss<MODE, VT>, 2>(src.p, src.lda, a->w, a->ldw, a->M, a->N, a->K, e, st); }
            if (mse) { if (vt == 4) MM_LOSS_EPI(0, 4) if (vt == 2) MM_LOSS_EPI(0, 2) MM_LOSS_EPI(0, 1) }
            if (vt == 4) MM_LOSS_EPI(1, 4) if (vt == 2) MM_LOSS_EPI(1, 2) MM_LOSS_EPI(1, 1)
#undef MM_LOSS_EPI
        }
        return MMVAE_ERR_ARG;
    }
    }
    return MMVAE_ERR_ARG;
}

// the folded mmvae_bn_finalize of the operand's producer (mmvae_gemm_nt_args::pro_finalize), or "off"
BnFin bn_fin_of(const mmvae_gemm_nt_args* a) {
    BnFin f;
    if (!a->pro_finalize) return f;
    const mmvae_bn_finalize_args* b = (const mmvae_bn_finalize_args*)a->pro_finalize;
    f.sum = b->sum; f.sumsq = b->sumsq; f.gamma = b->gamma; f.beta = b->beta; f.eps = b->eps; f.momentum = b->momentum;
    f.running_mean = b->running_mean; f.running_var = b->running_var; f.nbt = (long long*)b->num_batches_tracked;
    f.mean = b->mean; f.rstd = b->rstd; f.scale = b->scale; f.shift = b->shift; f.M = b->M;
    return f;
}
static bool bn_fin_ok(const mmvae_gemm_nt_args* a) {
    if (!a->pro_finalize) return true;
    const mmvae_bn_finalize_args* b = (const mmvae_bn_finalize_args*)a->pro_finalize;
    return a->prologue == MMVAE_PRO_BN_RELU_DROP && b->N == a->K && b->M >= 2 && b->sum && b->sumsq && b->gamma && b->beta && b->mean && b->rstd && b->scale && b->shift;
}

template <typename CT>
static int dispatch_src(const mmvae_gemm_nt_args* a, hipStream_t st) {
    constexpr int EPC = Mma<CT>::EPC;
    if (a->prologue == MMVAE_PRO_BN_RELU_DROP) {
        // A must be the activation type of this precision mode, 16-byte aligned rows
        if ((a->a_dtype == MMVAE_BF16) != (sizeof(CT) == 2)) return MMVAE_ERR_DTYPE;
        if (a->K > 512 || a->lda % EPC || ((uintptr_t)a->a & 15) || (!a->pro_finalize && (!a->pro_scale || !a->pro_shift))) return MMVAE_ERR_ARG;
        if (a->pro_mask && (a->ld_pro_mask % 4 || ((uintptr_t)a->pro_mask & 3))) return MMVAE_ERR_ARG;
        SrcBnReluDrop<CT> s{(const CT*)a->a, a->lda, a->M, a->K, a->pro_scale, a->pro_shift, a->pro_mask, a->ld_pro_mask, a->pro_inv_keep, bn_fin_of(a)};
        return dispatch_epi<CT>(a, s, st);
    }
    if (a->a_dtype == MMVAE_BF16) {
        if constexpr (sizeof(CT) == 2) {
            if (a->lda % 8 || ((uintptr_t)a->a & 15)) return MMVAE_ERR_ARG;
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
    if (key == 2) { mm::g_nt2_on = value; return 0; }
    if (key == 3) {         // tests: force the row-block path at moderate sizes (value = log2 of the block bytes; 0 restores the default)
        if (value != 0 && (value < 17 || value > 32)) return MMVAE_ERR_ARG;
        mm::g_split_bytes = value ? 1L << value : 1L << 32;
        mm::g_block_bytes = mm::g_split_bytes / 2;
        return 0;
    }
    if (key == 4) { mm::tn_wide_enable(value); return 0; }
    if (key == 6) { mm::g_bnbwd_stream = value; return 0; }
    if (key == 7) { mm::g_relu_stream = value; return 0; }
    if (key == 8 || key == 9) { mm::ntp_set(key, value); return 0; }
    return MMVAE_ERR_ARG;
}

extern "C" int mmvae_gemm_nt(const mmvae_gemm_nt_args* a, void* stream) {
    if (!a || !a->a || !a->w || (!a->c && !(a->epilogue == MMVAE_EPI_BN_BWD && a->bn_phase == 0))) return MMVAE_ERR_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return MMVAE_ERR_ARG;
    if (a->ldw % 64 || ((uintptr_t)a->w & 15)) return MMVAE_ERR_ARG;
    // The kernels address A, W and the prologue mask with 32-bit offsets from a scalar base.  Operands of 4 GiB or more (the
    // scaled omics widths: 65 536 x 27 000 fp32 = 7 GB) are processed in row blocks -- rows are independent in this product,
    // and the column statistics of the epilogues accumulate atomically across launches.
    const long lim = 1L << 32;
    if (((long)a->N + 256) * a->ldw * 4 >= lim) return MMVAE_ERR_ARG;
    const long a_row = (long)a->lda * (a->a_dtype == MMVAE_BF16 ? 2 : 4);
    long row_bytes = a_row > (long)a->ld_pro_mask ? a_row : (long)a->ld_pro_mask;
    if (!mm::bn_fin_ok(a)) return MMVAE_ERR_ARG;
    if ((long)a->M * row_bytes >= mm::g_split_bytes) {
        if (a->pro_finalize) return MMVAE_ERR_ARG;           // every row block would update the running statistics
        long rows = mm::g_block_bytes / row_bytes;          // block < split threshold: the recursion below ends after one level
        if (rows <= 0) return MMVAE_ERR_ARG;
        const long nblk = (a->M + rows - 1) / rows;         // equal blocks (65 536 rows -> 4 x 16 384, not 3 x 19 712 + 6 400: a short
        long even = (a->M + nblk - 1) / nblk;               // last block falls below the sizes the wide-tile kernels take)
        if (even >= 256) even = (even + 255) & ~255L;
        if (even <= rows) rows = even;
        else if (rows >= 256) rows &= ~255L;
        for (long r0 = 0; r0 < a->M; r0 += rows) {
            mmvae_gemm_nt_args s = *a;
            s.M = (int32_t)((a->M - r0 < rows) ? a->M - r0 : rows);
            s.a = (const char*)a->a + r0 * a_row;
            if (a->c) s.c = (char*)a->c + r0 * a->ldc * (a->c_dtype == MMVAE_BF16 ? 2 : 4);
            // H is activation-typed, except for the loss epilogues, whose H is the fp32 target
            const bool loss_epi = a->epilogue == MMVAE_EPI_LOSS_MSE || a->epilogue == MMVAE_EPI_LOSS_BCE_LOGIT;
            const long hsz = (a->prec == MMVAE_PREC_BF16 && !loss_epi) ? 2 : 4;
            if (a->h) s.h = (const char*)a->h + r0 * a->ldh * hsz;
            if (a->pro_mask) s.pro_mask = a->pro_mask + r0 * a->ld_pro_mask;
            if (a->pro_out) s.pro_out = (char*)a->pro_out + r0 * a->ld_pro_out * 2;
            if (a->epi_mask) s.epi_mask = a->epi_mask + r0 * a->ld_epi_mask;
            const int rc = mmvae_gemm_nt(&s, stream);
            if (rc) return rc;
        }
        return 0;
    }
    hipStream_t st = (hipStream_t)stream;
    { const int rc = mm::ntp_dispatch(a, st); if (rc != (1 << 30)) return rc; }
    if (a->pro_out) return MMVAE_ERR_ARG;              // only the wave-specialised kernel writes the operand after its prologue (mmvae_hip.h)
    if (a->prec == MMVAE_PREC_BF16) return mm::dispatch_src<mm::bf16>(a, st);
    if (a->prec == MMVAE_PREC_F32) return mm::dispatch_src<float>(a, st);
    return MMVAE_ERR_ARG;
}
