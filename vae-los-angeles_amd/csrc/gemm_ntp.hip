// Dispatch of the wave-specialised NT GEMM (gemm_ntp.h): which mmvae_gemm_nt problems it takes.
#include "common.h"
#include "mmvae_hip.h"
#include "gemm_ntp.h"

namespace mm {

BnFin bn_fin_of(const mmvae_gemm_nt_args* a);               // gemm_nt.hip
static int g_ntp_on = getenv("MMVAE_NO_NTP") ? 0 : 1;        // mmvae_set_tuning key 8 (tests flip it to compare with the tile kernels)
static int g_ntp_min_m = 16384;                               // key 9: below this a persistent 256-workgroup grid has < 1 tile per CU
void ntp_set(int key, int value) { if (key == 8) g_ntp_on = value; else g_ntp_min_m = value; }

template <typename AT, typename Epi>
static int ntp_cfg(const mmvae_gemm_nt_args* a, const Epi& e, hipStream_t st) {
    if (a->N % 256 == 0) return launch_ntp<NtpCfg<4, 4>, AT, Epi>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st);
    return launch_ntp<NtpCfg<4, 2>, AT, Epi>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st);
}

template <typename AT>
static int ntp_epi(const mmvae_gemm_nt_args* a, hipStream_t st) {
    const bool stats = a->stat1 != nullptr || a->stat2 != nullptr;
    if (a->c_dtype == MMVAE_BF16) {
        // whole tiles and whole 128-byte output lines only (the kernel's epilogue has no edge handling; the tile kernels do)
        const int bn = a->N % 256 == 0 ? 256 : 128;
        if (a->M % 128 || a->N % bn || a->ldc % 64 || ((uintptr_t)a->c & 127)) return NTP_SKIP;
        if (stats) { EpiStore<bf16, true> e{(bf16*)a->c, a->ldc, a->bias, a->act, 0, nullptr, 0, nullptr, 0, a->stat1, a->stat2};
                     return ntp_cfg<AT>(a, e, st); }
        EpiStore<bf16, false> e{(bf16*)a->c, a->ldc, a->bias, a->act, 0, nullptr, 0, nullptr, 0, nullptr, nullptr};
        return ntp_cfg<AT>(a, e, st);
    }
    return NTP_SKIP;
}

// bf16 A through the producers' BatchNorm + ReLU + Dropout prologue (the hidden BN layers' forward: EncoderB's second Linear)
template <typename Pro>
static int ntp_pro(const mmvae_gemm_nt_args* a, const Pro& pro, hipStream_t st) {
    const bool stats = a->stat1 != nullptr || a->stat2 != nullptr;
    const int bn = a->N % 256 == 0 ? 256 : 128;
    if (a->c_dtype != MMVAE_BF16 || a->M % 128 || a->N % bn || a->ldc % 64 || ((uintptr_t)a->c & 127)) return NTP_SKIP;
    if (stats) {
        EpiStore<bf16, true> e{(bf16*)a->c, a->ldc, a->bias, a->act, 0, nullptr, 0, nullptr, 0, a->stat1, a->stat2};
        if (bn == 256) return launch_ntp<NtpCfg<4, 4>, bf16, EpiStore<bf16, true>, Pro>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st, pro);
        return launch_ntp<NtpCfg<4, 2>, bf16, EpiStore<bf16, true>, Pro>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st, pro);
    }
    EpiStore<bf16, false> e{(bf16*)a->c, a->ldc, a->bias, a->act, 0, nullptr, 0, nullptr, 0, nullptr, nullptr};
    if (bn == 256) return launch_ntp<NtpCfg<4, 4>, bf16, EpiStore<bf16, false>, Pro>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st, pro);
    return launch_ntp<NtpCfg<4, 2>, bf16, EpiStore<bf16, false>, Pro>(a->a, a->lda, a->w, a->ldw, a->M, a->N, a->K, e, st, pro);
}

// NTP_SKIP: not taken (the caller continues with the tile kernels); anything else is the launch status
int ntp_dispatch(const mmvae_gemm_nt_args* a, hipStream_t st) {
    if (!g_ntp_on || a->prec != MMVAE_PREC_BF16 || a->epilogue != MMVAE_EPI_STORE || a->accumulate) return NTP_SKIP;
    if (a->K <= 64 || a->M < g_ntp_min_m || a->M % 8) return NTP_SKIP;      // M % 8: see the A producers' row groups
    if (a->prologue == MMVAE_PRO_BN_RELU_DROP) {
        static const bool off = getenv("MMVAE_NO_NTP_PRO") != nullptr;      // A/B switch
        if (a->pro_out && (a->ld_pro_out % 8 || ((uintptr_t)a->pro_out & 15) || a->ld_pro_out < a->K)) return NTP_SKIP;
        if (off || a->a_dtype != MMVAE_BF16 || a->K % 64 || a->K > 512 || a->lda % 8 || ((uintptr_t)a->a & 15) || (!a->pro_finalize && (!a->pro_scale || !a->pro_shift))) return NTP_SKIP;
        if (a->pro_mask) {
            if (a->ld_pro_mask % 8 || ((uintptr_t)a->pro_mask & 7)) return NTP_SKIP;      // 8 keep bytes per lane and load
            return ntp_pro(a, NtpProBn<true>{a->pro_scale, a->pro_shift, a->pro_mask, a->ld_pro_mask, a->pro_inv_keep, (bf16*)a->pro_out, a->ld_pro_out, bn_fin_of(a)}, st);
        }
        return ntp_pro(a, NtpProBn<false>{a->pro_scale, a->pro_shift, nullptr, 0, a->pro_inv_keep, (bf16*)a->pro_out, a->ld_pro_out, bn_fin_of(a)}, st);
    }
    if (a->prologue != MMVAE_PRO_NONE) return NTP_SKIP;
    if (a->a_dtype == MMVAE_F32) {
        if (a->K < 4 || ((uintptr_t)a->a & 3)) return NTP_SKIP;
        return ntp_epi<float>(a, st);
    }
    // plain bf16 A (the decoders' hidden Linear + ReLU, decoders.py:29-30): the producers copy 16-byte chunks; 35 -> 31 us at 256 -> 512
    static const bool no_bf16 = getenv("MMVAE_NO_NTP_BF16") != nullptr;      // A/B switch
    if (!no_bf16 && a->a_dtype == MMVAE_BF16 && a->lda % 8 == 0 && ((uintptr_t)a->a & 15) == 0) return ntp_epi<bf16>(a, st);
    return NTP_SKIP;
}

}  // namespace mm

#ifdef MM_STAMP
extern "C" int mmvae_debug_ntp_stamps(unsigned long long* out24, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out24, HIP_SYMBOL(mm::mm_ntp_stamps), 24 * sizeof(unsigned long long));
    if (e != hipSuccess) return (int)e;
    if (reset) { unsigned long long z[24] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(mm::mm_ntp_stamps), z, sizeof(z)); }
    return (int)e;
}
#endif
