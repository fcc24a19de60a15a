/* mmvae_hip.h -- C ABI of libmmvae_hip.so: the MI355X (gfx950) kernels behind the
 * MultiModalVAE training hot path of marcin119a/vae-los-angeles.
 *
 * The reference has no FFI of its own: its "operator interface" for this path is the set of
 * stock PyTorch ops its modules call.  Each entry point below names the reference call sites
 * (file:line, relative to the reference repo) whose device work it replaces.  The host side
 * (the mmvae Python package under vae-los-angeles_amd/) binds these with ctypes and keeps the reference's
 * src.models / src.utils.losses class surface on top.
 *
 * Conventions
 *   - plain pointers to DEVICE memory and sizes; no framework types.  The library never
 *     allocates or frees device memory and keeps no state between calls.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); nothing
 *     synchronises, so calls are graph-capturable.
 *   - return value: 0 = ok; <0 = argument check failed (MMVAE_ERR_*); >0 = hipError_t of the launch.
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 *   - "activation type" = float in MMVAE_PREC_F32 mode, bfloat16 in MMVAE_PREC_BF16 mode.
 */
#ifndef MMVAE_HIP_H
#define MMVAE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMVAE_OK 0
#define MMVAE_ERR_ARG (-1)
#define MMVAE_ERR_DTYPE (-2)

enum { MMVAE_F32 = 0, MMVAE_BF16 = 1 };                 /* storage dtype of a buffer   */
enum { MMVAE_PREC_F32 = 0, MMVAE_PREC_BF16 = 1 };       /* MFMA operand precision      */
enum { MMVAE_PRO_NONE = 0, MMVAE_PRO_BN_RELU_DROP = 1, MMVAE_PRO_BN_BWD_APPLY = 2 };
enum { MMVAE_EPI_STORE = 0, MMVAE_EPI_RELU_MASK = 1, MMVAE_EPI_BN_BWD = 2, MMVAE_EPI_LOSS_MSE = 3, MMVAE_EPI_LOSS_BCE_LOGIT = 4 };
enum { MMVAE_ACT_NONE = 0, MMVAE_ACT_RELU = 1, MMVAE_ACT_SIGMOID = 2 };

#define MMVAE_TILE 128          /* GEMM output tile edge */

int mmvae_abi_version(void);    /* bumped on any struct change; the ctypes binding checks it */
/* Tuning knobs (tests / A-B runs): key 0 = minimum M for the 128x256-tile NT kernels (default 32768); key 2 = LDS-DMA generation of the
 * NT kernel (gemm_nt2.h) on/off; key 3 = log2 of the operand size in bytes from which row blocks are used (17..32; 0 = default 32):
 * mmvae_gemm_nt / mmvae_gemm_tn address their row operands with 32-bit offsets, so an operand of 4 GiB or more (65 536 x 27 000 fp32
 * at the scaled omics widths) is processed in row blocks of at most half that threshold inside the entry point, and key 3 lowers it so
 * that tests reach that path at moderate sizes; key 4 = wide-tile kernel for the large weight gradients (gemm_tn_wide.hip) on/off;
 * key 6 / key 7 = row-coalesced LDS form of the BatchNorm-backward / ReLU-mask dX epilogue on/off; key 8 = wave-specialised NT kernel
 * (gemm_ntp.h: producer / consumer waves) on/off, key 9 = its minimum M (default 16384).  Other keys: MMVAE_ERR_ARG. */
int mmvae_set_tuning(int32_t key, int32_t value);

/* ---------------------------------------------------------------------------------------------
 * Weight preparation: fp32 master weights -> zero-padded MFMA operand copies (compute type),
 * plain and transposed, plus concatenations (fc_mu|fc_logvar heads).  One launch for a whole
 * table of items held in device memory.
 *   dst[r][c] (r < dst_rows, c < dst_cols, leading dim dst_ld) =
 *       transpose ? src[c][r] : src[r][c]   if inside src's logical [src_rows][src_cols], else 0
 * Replaces: the implicit weight reads of every aten::addmm / mm on the path.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float* src; void* dst;
    int32_t src_rows, src_cols; int64_t src_ld;
    int32_t dst_rows, dst_cols; int64_t dst_ld;
    int32_t transpose; int32_t dst_dtype;
} mmvae_prep_item;
int mmvae_prep_weights(const mmvae_prep_item* items_dev, int32_t n_items, void* stream);

/* ---------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( prologue(A)[M,K] x W[N,K]^T )          (gemm_nt.hip)
 *   W        prepared operand [ceil128(N)][ceil64(K)] in compute type, ldw % 64 == 0
 *   A (bf16) rows are padded to a multiple of 8 elements and the pad columns must hold ZEROS (every producer in this library writes
 *            them): the LDS-DMA kernels move whole 16-byte chunks and multiply the pads with the zero padding of W -- 0 x NaN bits is NaN
 *   h, masks are [M][ld] matrices whose rows hold N elements rounded up to the padding of the activation buffers (8 elements; masks:
 *            4 bytes): a kernel reads nothing beyond that from a row's first element, so h may be a column slice of a wider buffer
 *            (the merged first layers of the decoders) without any tail padding
 *   prologue MMVAE_PRO_BN_RELU_DROP: A is the previous layer's PRE-BatchNorm output (activation
 *            type); the kernel applies relu(A*pro_scale[k]+pro_shift[k]) * keep/(1-p) on the fly
 *            (encoders.py:14-16,32-34,36-38).  pro_mask: uint8 keep mask [M][ld_pro_mask], bytes 0 or 1 (mmvae_noise), or NULL.
 *   epilogue MMVAE_EPI_STORE    : C = act(acc + bias) (+ C if accumulate); optional per-column
 *                                 (sum, sum of squares) of the stored values, accumulated into
 *                                 stat1/stat2 (BatchNorm batch statistics)
 *            MMVAE_EPI_RELU_MASK: C = acc * (h > 0)                (ReLU backward, decoders)
 *            MMVAE_EPI_BN_BWD   : d = acc * keep/(1-p) * (h*bn_scale+bn_shift > 0), h = pre-BN output
 *                                 (Dropout+ReLU backward), then BatchNorm backward in two launches:
 *                                 bn_phase 0: nothing stored, stat1/stat2 += (sum d, sum d*xhat);
 *                                 bn_phase 1: C = coef0*(d - coef1 - xhat*coef2), coef from
 *                                 mmvae_bn_bwd_finalize.  d is recomputed from the f32 accumulators, so
 *                                 the subtraction happens before the one rounding to the activation type.
 *            MMVAE_EPI_LOSS_MSE / MMVAE_EPI_LOSS_BCE_LOGIT: the reconstruction loss of a decoder's last layer inside its GEMM
 *                                 (bf16 mode, bf16 A, K > 64): x = acc + bias is not stored; *stat1 (ONE f64) += sum (x - h)^2, or
 *                                 += sum BCE(sigmoid(x), h) with the log clamp at -100 (losses.py:31,34); C (bf16, ldc % 8 == 0,
 *                                 pad columns zeroed) = 2 (x - h), or sigmoid(x) - h = the gradient w.r.t. the logit; h = fp32 target
 *                                 [M][ldh].  Same arithmetic as mmvae_vae_loss on the stored output; saves writing and re-reading it.
 * Replaces: nn.Linear forward = aten::addmm (encoders.py:13,18-19,31,35,40-41,54-55;
 *   decoders.py:13,15,27,29,31,44,46), relu/sigmoid (decoders.py:14,28,30,32), batch-norm
 *   statistics, and the dX mm of each Linear backward (optimize_hyperparameters.py:112).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t prec, M, N, K;
    const void* a; int32_t a_dtype; int64_t lda;
    int32_t prologue;
    const float* pro_scale; const float* pro_shift; const uint8_t* pro_mask; int64_t ld_pro_mask; float pro_inv_keep;
    const void* w; int64_t ldw;
    int32_t epilogue;
    void* c; int32_t c_dtype; int64_t ldc;
    const float* bias; int32_t act; int32_t accumulate;
    const void* h; int64_t ldh;
    const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd;
    const uint8_t* epi_mask; int64_t ld_epi_mask; float epi_inv_keep;
    const float* bn_coef; int32_t bn_phase;       /* MMVAE_EPI_BN_BWD: 0 = statistics, 1 = apply (recompute), 2 = statistics + store d */
    double* stat1; double* stat2;                 /* optional [N] f64 accumulators (atomic adds; zero them first) */
    /* optional, MMVAE_PRO_BN_RELU_DROP with a bf16 A only: the operand AFTER the prologue -- relu(a * scale + shift) * keep / (1 - p),
       i.e. the previous layer's post-activation (encoders.py:33-34,37-38) -- is also written here ([M][ld_pro_out] bf16, K columns) so
       that the layer's dW GEMM can read it as a plain operand.  Only the wave-specialised kernel writes it (its producer waves hold the
       values anyway): MMVAE_ERR_ARG when the problem is not one of its (M >= 16384, M % 128 == 0, N % 128 == 0, N <= 256, K % 64 == 0,
       K <= 512, bf16 C with whole 128-byte rows). */
    void* pro_out; int64_t ld_pro_out;
    /* optional, MMVAE_PRO_BN_RELU_DROP: `const mmvae_bn_finalize_args*` (host memory, read during the call).  mmvae_bn_finalize of the
       layer that produced A is folded into this launch: every workgroup forms scale / shift of A's columns from the f64 column sums
       (pro_scale / pro_shift are then ignored), and ONE workgroup writes what mmvae_bn_finalize writes -- mean, rstd, scale, shift, the
       running statistics and num_batches_tracked -- for the backward pass.  fin->N must equal K.  Not for operands of 4 GiB or more
       (MMVAE_ERR_ARG: the row blocks would each update the running statistics). */
    const void* pro_finalize;
} mmvae_gemm_nt_args;
int mmvae_gemm_nt(const mmvae_gemm_nt_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * dW[N,K] += P[M,N]^T x Q[M,K] ;  db[N] += column sums of P      (gemm_tn.hip)
 *   P = gradient w.r.t. the layer output, Q = the layer input (optionally through the same
 *   BN+ReLU+Dropout prologue as above).  dW/db are fp32 and ACCUMULATED (dW through the slab workspace or f32
 *   atomics, db with atomics): zero them first.  nsplit <= 0 lets the library choose the batch split (<= 64).
 * Replaces: the dW mm and db sum of each Linear backward (optimize_hyperparameters.py:112).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t prec, M, N, K;
    const void* p; int32_t p_dtype; int64_t ldp;
    const void* q; int32_t q_dtype; int64_t ldq;
    int32_t q_prologue;
    const float* pro_scale; const float* pro_shift; const uint8_t* pro_mask; int64_t ld_pro_mask; float pro_inv_keep;
    float* dw; int64_t lddw; float* db;
    int32_t nsplit;
    float* slab; int64_t slab_elems;   /* optional workspace: when it holds nsplit*N*K floats the splits store partial tiles there
                                          and a second launch sums them in fixed order (deterministic dW, no atomics); else atomics */
    /* optional prologue on P (MMVAE_PRO_BN_BWD_APPLY): P = coef0 * (p - coef1 - xhat * coef2), xhat = (p_y - mean) * rstd, i.e.
       mmvae_bn_bwd_apply folded into the operand load (first layers: nothing else consumes dL/dy).  p_y has p's dtype;
       p_coef is [3][N] as written by mmvae_bn_bwd_finalize. */
    int32_t p_prologue;
    const void* p_y; int64_t ld_py; const float* p_mean; const float* p_rstd; const float* p_coef;
    /* p_coef == NULL with MMVAE_PRO_BN_BWD_APPLY: mmvae_bn_bwd_finalize folded into this launch.  The kernel forms the three
       constants per column from the f64 sums of MMVAE_EPI_BN_BWD (coef = {gamma * rstd, sum_d / M, sum_dx / M}; p_eval_mode != 0:
       {gamma * rstd, 0, 0}) and ONE workgroup per column tile adds dgamma += sum_dx, dbeta += sum_d.  Only the wide-tile kernels do
       this (M >= 8192, N >= 128, K >= 256, bf16, slab given): MMVAE_ERR_ARG otherwise -- call mmvae_bn_bwd_finalize and pass p_coef. */
    const double* p_sum_d; const double* p_sum_dx; const float* p_gamma; float* p_dgamma; float* p_dbeta; int32_t p_eval_mode;
} mmvae_gemm_tn_args;
int mmvae_gemm_tn(const mmvae_gemm_tn_args* args, void* stream);
/* Up to MMVAE_TN_GROUP_MAX small-output problems (same prec; the latent / class-width layers: encoder heads, decoder first
 * layers, DecoderC) in ONE GEMM launch + ONE reduce launch: alone each is a latency chain of ~25 us for a few MB.  Every
 * problem needs a slab workspace of its own (MMVAE_TN_GROUP_SPLITS * N * K floats covers any split), lddw == K and no P prologue; operand
 * combinations: P f32 with Q through the BN+ReLU+Dropout prologue, P and Q activation-typed, P f32 with Q activation-typed.
 * Returns MMVAE_ERR_ARG when a problem does not fit: launch that one with mmvae_gemm_tn. */
#define MMVAE_TN_GROUP_MAX 8
#define MMVAE_TN_GROUP_SPLITS 256
int mmvae_gemm_tn_group(const mmvae_gemm_tn_args* args, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d, training mode (encoders.py:14,32,36): from the column sums compute
 * mean / biased var, emit scale = gamma*rstd and shift = beta - mean*scale for the consumer's
 * prologue, save mean/rstd for backward, update running stats (momentum, UNBIASED variance)
 * and num_batches_tracked.  Eval mode: coefficients from the running stats.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t M, N; const double* sum; const double* sumsq;        /* [N] each, from MMVAE_EPI_STORE stat1/stat2 */
    const float* gamma; const float* beta; float eps; float momentum;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* may be NULL */
    float* mean; float* rstd; float* scale; float* shift;
} mmvae_bn_finalize_args;
int mmvae_bn_finalize(const mmvae_bn_finalize_args* args, void* stream);
/* eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale; optionally (non-NULL) also
 * mean = running_mean and rstd = 1 / sqrt(running_var + eps), which a backward pass through an eval-mode forward needs. */
int mmvae_bn_eval_coeffs(int32_t N, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, float* scale, float* shift, float* mean, float* rstd, void* stream);

/* BatchNorm backward reductions: from the sums (sum d, sum d*xhat) of MMVAE_EPI_BN_BWD phase 0:
 *   dgamma += sum d*xhat ; dbeta += sum d ; coef[3][N] = {gamma*rstd, dbeta/M, dgamma/M}
 * eval_mode != 0 (the forward ran on the running statistics, torch's batch_norm(training=False) backward): the
 * normalisation does not depend on the batch, coef = {gamma*rstd, 0, 0}; dgamma / dbeta as above. */
typedef struct {
    int32_t M, N; const double* sum_d; const double* sum_dx;      /* [N] each, from MMVAE_EPI_BN_BWD phase 0 */
    const float* gamma; const float* rstd;
    float* dgamma; float* dbeta; float* coef;           /* coef: [3][N] */
    int32_t eval_mode;
} mmvae_bn_bwd_finalize_args;
int mmvae_bn_bwd_finalize(const mmvae_bn_bwd_finalize_args* args, void* stream);
/* d <- coef0 * (d - coef1 - xhat*coef2), xhat = (y-mean)*rstd, in place on the activation-typed buffer d written by
 * MMVAE_EPI_BN_BWD bn_phase 2 (the one-contraction form of BatchNorm backward). */
int mmvae_bn_bwd_apply(int32_t dtype, int32_t M, int32_t N, void* d, int64_t ldd, const void* y, int64_t ldy,
                       const float* mean, const float* rstd, const float* coef, void* stream);
/* mmvae_bn_bwd_finalize + mmvae_bn_bwd_apply in ONE launch: the constants are formed from the f64 sums by every thread for its own
 * columns, the threads of the first rows add dgamma += sum_dx, dbeta += sum_d.  Same argument limits as mmvae_bn_bwd_apply, and the
 * column-resident form only (256 % (N / 8) == 0 for bf16, 256 % (N / 4) == 0 for f32: every hidden width of the model); else ERR_ARG. */
int mmvae_bn_bwd_finalize_apply(int32_t dtype, int32_t M, int32_t N, void* d, int64_t ldd, const void* y, int64_t ldy,
                                const float* mean, const float* rstd, const double* sum_d, const double* sum_dx, const float* gamma,
                                float* dgamma, float* dbeta, int32_t eval_mode, void* stream);

/* ---------------------------------------------------------------------------------------------
 * EncoderC (encoders.py:57-61): Embedding + two heads == a per-class table
 *   T[S][2L] = emb[S][E] x [Wmu;Wlv]^T + [bmu;blv]      (fp32), gathered per sample later.
 * Backward: from dT[S][2L]: dEmb, dWmu, dWlv, dbmu, dblv (all accumulated).
 * ------------------------------------------------------------------------------------------- */
int mmvae_embed_table_fwd(int32_t S, int32_t E, int32_t L, const float* emb, const float* w_mu, const float* b_mu,
                          const float* w_lv, const float* b_lv, float* table, void* stream);
int mmvae_embed_table_bwd(int32_t S, int32_t E, int32_t L, const float* emb, const float* w_mu, const float* w_lv,
                          const float* d_table, int32_t table_copies, float* d_emb, float* d_w_mu, float* d_b_mu, float* d_w_lv,
                          float* d_b_lv, void* stream);      /* d_table: [table_copies][S][2L] (see mmvae_fuse_bwd_args), summed here */

/* ---------------------------------------------------------------------------------------------
 * Mean-fusion over the modalities present + reparameterisation (vae.py:65-73, 11-15):
 *   mu = mean_m mu_m ; logvar = mean_m logvar_m ; z = mu + eps * exp(0.5*logvar)
 *   heads_x: [B][2L] fp32 (mu | logvar) or NULL; table/site: EncoderC table + int64 labels or NULL.
 *   z is written in activation type with leading dim ldz (pad columns are zeroed).
 * Backward: d_heads[B][2L] = [ (g_mu + dz)/n | (g_lv + dz*eps*std/2)/n ], and the same rows
 *   scatter-added into d_table[site] when the site modality is present.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, L, n_mod;
    const float* heads_a; const float* heads_b; int64_t ld_heads;
    const float* table; const int64_t* site; int32_t S;
    const float* eps;
    float* mu; float* logvar;
    void* z; int32_t z_dtype; int64_t ldz;
} mmvae_fuse_fwd_args;
int mmvae_fuse_reparam_fwd(const mmvae_fuse_fwd_args* args, void* stream);

typedef struct {
    int32_t B, L, n_mod;
    const float* g_mu; const float* g_lv;           /* may be NULL (treated as 0) */
    const float* dz; const float* dz2; const float* dz3; int64_t lddz;   /* dL/dz per decoder (dz2, dz3 may be NULL); summed */
    const float* eps; const float* logvar;
    float* d_heads; int64_t ld_heads;
    float* d_table; const int64_t* site; int32_t S;  /* may be NULL */
    void* d_heads_lp; int64_t ld_heads_lp;   /* optional bf16 copy of d_heads ([B][>= 2L], pad columns zeroed): the A operand of the heads' dX
                                              * GEMMs in bf16 mode (they round d_heads to bf16 on load anyway), which lets them run the
                                              * LDS-DMA kernel with the row-coalesced BatchNorm-backward epilogue */
    int32_t table_copies;   /* d_table is [table_copies][S][2L], zeroed; workgroup w adds into copy w % table_copies (0 = 1).  Every
                             * workgroup ends with S x 2L atomic adds onto the same addresses: 512 workgroups on ONE copy spent 19 of
                             * the kernel's 32 us there.  mmvae_embed_table_bwd sums the copies. */
} mmvae_fuse_bwd_args;
#define MMVAE_TABLE_COPIES 8
int mmvae_fuse_reparam_bwd(const mmvae_fuse_bwd_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * vae_loss (src/utils/losses.py:8-46) and the directional losses
 * (src/utils/directional_losses.py:8-55), one pass, optional terms (NULL pointers skip a term):
 *   sums[0] += sum (recon_a-a)^2                               (losses.py:31)
 *   sums[1] += sum -[b*max(log p,-100) + (1-b)*max(log(1-p),-100)]   (losses.py:34)
 *   sums[2] += sum_i w[site_i] * nll_i                          (losses.py:39)
 *   sums[3] += -0.5 * sum(1 + lv - mu^2 - exp(lv))              (losses.py:42)
 *   sums[4] += number of labels outside [0, S)  (torch's cross_entropy device-asserts on them; here such a row is
 *              computed as class 0 and COUNTED: the host wrapper raises when the count it reads back is not 0).  A label of -100
 *              is F.cross_entropy's default ignore_index: that row adds no loss and gets a zero gradient, and is not counted
 * `sums` is double[5], zeroed by the caller; mmvae_loss_finalize turns it into the tuple of losses.py:44,46 (a last-block
 * finalisation inside the kernel was measured: ~1000 tickets on one address + the fences cost 50 us against a 5 us launch).
 * Gradients of total = s0+s1+gamma*s2+beta*s3:
 *   g_a = 2(recon_a-a) ; g_b = (p-b)/max(p(1-p),1e-12)  [grad_b_wrt_logit: times p(1-p)] ;
 *   g_c = gamma*w[y]*(softmax - onehot) ; g_mu = beta*mu ; g_lv = -0.5*beta*(1-exp(lv)).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, A, D, S, L;
    const float* recon_a; const float* a; int64_t ld_ra, ld_a;
    const float* recon_b; const float* b; int64_t ld_rb, ld_b;
    const float* logits; int64_t ld_logits; const int64_t* site; const float* class_weights;
    const float* mu; const float* logvar;
    float beta, gamma;
    double* sums;
    void* g_a; int32_t g_a_dtype; int64_t ld_ga;
    void* g_b; int32_t g_b_dtype; int64_t ld_gb; int32_t grad_b_wrt_logit;
    float* g_c; int64_t ld_gc;
    float* g_mu; float* g_lv;
    const float* beta_gamma_dev;      /* optional {beta, gamma} in device memory: overrides the by-value fields, so a captured hipGraph
                                         follows the beta warm-up (optimize_hyperparameters.py:103) without re-capture */
} mmvae_loss_args;
int mmvae_vae_loss(const mmvae_loss_args* args, void* stream);
/* out5 = {recon + gamma*class + beta*kld, recon, class, kld, labels out of range} (float) from sums[5]. */
int mmvae_loss_finalize(const double* sums, float beta, float gamma, const float* beta_gamma_dev, float* out5, void* stream);

/* out = g * p * (1-p): Sigmoid backward for gradients that arrive w.r.t. recon_b (decoders.py:32). */
int mmvae_sigmoid_bwd(int32_t M, int32_t N, const float* g, int64_t ldg, const float* p, int64_t ldpp,
                      void* out, int32_t out_dtype, int64_t ldo, void* stream);

/* x *= *scale unless *scale == 1 (loss.backward(gradient=...) support); n elements of dtype. */
int mmvae_scale_if_needed(void* x, int32_t dtype, int64_t n, const float* scale_dev, void* stream);
/* the same for up to MMVAE_SCALE_MAX tensors in ONE launch (the records travel in the kernel arguments) */
#define MMVAE_SCALE_MAX 8
typedef struct { void* x; int64_t n; int32_t dtype; int32_t pad_; } mmvae_scale_item;
int mmvae_scale_many(const mmvae_scale_item* items_host, int32_t n_items, const float* scale_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Noise: Philox4x32-10 streams keyed by (seed, offset).  Dropout keep mask (nn.Dropout(0.1),
 * encoders.py:16,34,38; replaces aten::bernoulli_) and standard normal eps
 * (torch.randn_like, vae.py:14).
 * ------------------------------------------------------------------------------------------- */
/* One launch: n_mask keep-mask bytes (P(1) = keep_prob) and n_eps standard normals.  The Philox counter starts at
 * offset + *offset_dev (offset_dev may be NULL); a device-resident offset lets a captured hipGraph draw fresh noise on
 * every replay.  The call consumes ceil(n_mask/16)*4 + ceil(n_eps/4) counter values.  advance != 0: offset_dev is an array of
 * MMVAE_CTR_COPIES identical copies of the counter (block L reads copy L; no block reads a word another one writes) and the
 * launch itself moves every copy past what it consumed -- no separate counter launch. */
#define MMVAE_CTR_COPIES 16384
int mmvae_noise(uint8_t* mask, int64_t n_mask, float keep_prob, float* eps, int64_t n_eps, uint64_t seed, uint64_t offset,
                uint64_t* offset_dev, int32_t advance, void* stream);
int mmvae_counter_add(uint64_t* counter_dev, uint64_t inc, void* stream);      /* *counter_dev += inc */

/* ---------------------------------------------------------------------------------------------
 * Minibatch assembly from a device-resident dataset: dst_t[i][:] = src_t[idx[i]][:], i < rows, for up to MMVAE_GATHER_MAX
 * row-major tensors sharing one int64 index vector (rows / strides in BYTES, multiples of 4; 8-byte words are moved when everything
 * is a multiple of 8).  Indices outside
 * [0, src_rows) are clamped.  Replaces: MultiModalDataset.__getitem__ + the DataLoader's default collate
 * (src/data/dataset.py:28-39, optimize_hyperparameters.py:55-65) -- one Python call and one torch.tensor() per SAMPLE.
 * ------------------------------------------------------------------------------------------- */
#define MMVAE_GATHER_MAX 4
typedef struct { const void* src; void* dst; int64_t src_row_stride; int64_t dst_row_stride; int32_t row_bytes; int32_t pad_; } mmvae_gather_item;
int mmvae_gather_rows(const mmvae_gather_item* items_host, int32_t n_items, const int64_t* idx_dev, int32_t rows,
                      int64_t src_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AdamW (torch.optim.AdamW, constructed by the caller: optimize_hyperparameters.py:93-97,
 * train_dna2rna.py:185-189), all tensors in one launch per 64 tensors.  `items_host` is an array in HOST memory
 * (device pointers inside); it is copied into the kernel arguments, so nothing is uploaded and the call is graph-capturable:
 *   p *= 1-lr*wd ; m = b1*m+(1-b1)g ; v = b2*v+(1-b2)g^2 ; p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps)
 * ------------------------------------------------------------------------------------------- */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; } mmvae_adamw_item;
int mmvae_adamw_step(const mmvae_adamw_item* items_host, int32_t n_items, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, int32_t maximize,
                     uint64_t* step_dev, int32_t advance, const float* lr_dev, void* stream);
/* step_dev != NULL: bias corrections are computed in the kernel from t = *step_dev + 1 (graph-capturable) and
 * bias_corr1/2 are ignored.  advance != 0 (n_items <= 64): step_dev holds MMVAE_CTR_COPIES identical copies of the count and
 * the launch increments all of them itself (see mmvae_noise); otherwise advance the counter with mmvae_counter_add.
 * lr_dev != NULL: the learning rate is read from device memory (a captured graph follows ReduceLROnPlateau,
 * train_dna2rna.py:190-195,216, without re-capture). */

#ifdef __cplusplus
}
#endif
#endif /* MMVAE_HIP_H */
