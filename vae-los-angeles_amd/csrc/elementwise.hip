// Memory-bound kernels of the MultiModalVAE training path (gfx950): weight preparation,
// BatchNorm finalisation (forward statistics / backward reductions), EncoderC table,
// mean-fusion + reparameterisation, the fused loss, Philox noise and multi-tensor AdamW.
// All of these are HBM- or latency-bound; they use 64-lane wave reductions, vector loads where
// the caller's row alignment allows, and f64 only for the final cross-block accumulations.
#include "common.h"
#include "mmvae_hip.h"

namespace mm {

// ------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------
// 32 x 32 tiles of the destination; transposed copies go through LDS so that both the fp32 reads and the bf16 writes run along rows
// (the element-per-thread form read the source of a transposed copy with a stride of one source row per lane: 21 us for 2 x 2.3 MB).
__global__ __launch_bounds__(256) void prep_weights_kernel(const mmvae_prep_item* __restrict__ items) {
    const mmvae_prep_item it = items[blockIdx.y];
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // 32 x 8 threads, 4 rows each
    const int tiles_c = (it.dst_cols + 31) / 32, tiles_r = (it.dst_rows + 31) / 32;
    auto put = [&](int r, int c, float v) {
        if (it.dst_dtype == MMVAE_BF16) ((bf16*)it.dst)[(long)r * it.dst_ld + c] = (bf16)v;
        else ((float*)it.dst)[(long)r * it.dst_ld + c] = v;
    };
    for (int t = blockIdx.x; t < tiles_r * tiles_c; t += gridDim.x) {       // block-uniform trip count
        const int tr = t / tiles_c, tc = t - tr * tiles_c;
        if (it.transpose) {                                                  // dst[r][c] = src[c][r]
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sr = tc * 32 + ty + 8 * k, sc = tr * 32 + tx;
                tile[ty + 8 * k][tx] = (sr < it.src_rows && sc < it.src_cols) ? it.src[(long)sr * it.src_ld + sc] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
                if (r < it.dst_rows && c < it.dst_cols) put(r, c, tile[tx][ty + 8 * k]);
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
                if (r < it.dst_rows && c < it.dst_cols) put(r, c, (r < it.src_rows && c < it.src_cols) ? it.src[(long)r * it.src_ld + c] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm finalisation from the f64 column sums accumulated by the GEMM epilogues.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(mmvae_bn_finalize_args a) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < a.N) {
        const double mean = a.sum[col] / a.M;
        double var = a.sumsq[col] / a.M - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = a.gamma[col] * rstd;
        a.mean[col] = (float)mean; a.rstd[col] = rstd;
        a.scale[col] = sc; a.shift[col] = a.beta[col] - (float)mean * sc;
        if (a.running_mean) {
            a.running_mean[col] = (1.f - a.momentum) * a.running_mean[col] + a.momentum * (float)mean;
            const double unbiased = var * ((double)a.M / (double)(a.M - 1));
            a.running_var[col] = (1.f - a.momentum) * a.running_var[col] + a.momentum * (float)unbiased;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
}

__global__ void bn_eval_coeffs_kernel(int N, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, float* mean, float* rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < N) {
        const float rs = 1.f / sqrtf(rv[c] + eps);
        const float sc = gamma[c] * rs;
        scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
        if (mean) mean[c] = rm[c];          // backward through an eval-mode forward: xhat = (y - running_mean) * rstd
        if (rstd) rstd[c] = rs;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(mmvae_bn_bwd_finalize_args a) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < a.N) {
        const double sd = a.sum_d[col], sdx = a.sum_dx[col];
        a.dbeta[col] += (float)sd;
        a.dgamma[col] += (float)sdx;
        a.coef[col] = a.gamma[col] * a.rstd[col];
        // eval mode (running statistics): the normalisation constants do not depend on the batch, dy = gamma * rstd * d
        a.coef[a.N + col] = a.eval_mode ? 0.f : (float)(sd / a.M);
        a.coef[2 * a.N + col] = a.eval_mode ? 0.f : (float)(sdx / a.M);
    }
}

template <typename GT, int V> __device__ __forceinline__ void store_vec(GT* p, const float* v);

// dy = c0 * (d - c1 - xhat * c2), xhat = (y - mean) * rstd, in place on d; V elements per thread, N % V == 0
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int N, T* d, long ldd, const T* y, long ldy,
                                                            const float* mean, const float* rstd, const float* coef) {
    const unsigned vpr = N / V, total = (unsigned)M * vpr;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / vpr, c = (i - r * vpr) * V;
        float dv[V], yv[V], o[V];
        VLoad<T, V>::ld(d + (long)r * ldd + c, dv);
        VLoad<T, V>::ld(y + (long)r * ldy + c, yv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (yv[e] - mean[c + e]) * rstd[c + e];
            o[e] = coef[c + e] * (dv[e] - coef[N + c + e] - xh * coef[2 * N + c + e]);
        }
        store_vec<T, V>(d + (long)r * ldd + c, o);
    }
}

// Same operation for launches whose thread count is a multiple of the vectors per row (256 % (N / V) == 0 for every hidden
// width of the model): a thread keeps ONE column group for all its rows, so the five per-column constants live in registers
// instead of being re-loaded per vector (40 loads per 16 bytes of d), there is no index division, and U row vectors are in
// flight per thread (the in-place store would otherwise serialise the loop on one load pair).
// FIN: the constants come from the f64 sums (mmvae_bn_bwd_finalize folded in); the threads of the first row group add dgamma / dbeta
struct BnBwdFin { const double* sum_d; const double* sum_dx; const float* gamma; float* dgamma; float* dbeta; int eval_mode; };
template <typename T, int V, int U, bool FIN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_cols_kernel(int M, int N, T* d, long ldd, const T* y, long ldy,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ coef, const BnBwdFin fin) {
    const unsigned vpr = N / V, tid = blockIdx.x * 256u + threadIdx.x;
    const unsigned r0 = tid / vpr, c = (tid - r0 * vpr) * V, dr = gridDim.x * 256u / vpr;
    float mu[V], rs[V], c0[V], c1[V], c2[V];
#pragma unroll
    for (int q = 0; q < V; q += 4) {                                         // V = 4 or 8, N % V == 0
        VLoad<float, 4>::ld(mean + c + q, mu + q); VLoad<float, 4>::ld(rstd + c + q, rs + q);
        if constexpr (!FIN) { VLoad<float, 4>::ld(coef + c + q, c0 + q); VLoad<float, 4>::ld(coef + N + c + q, c1 + q); VLoad<float, 4>::ld(coef + 2 * N + c + q, c2 + q); }
    }
    if constexpr (FIN) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const double sd = fin.sum_d[c + e], sdx = fin.sum_dx[c + e];
            c0[e] = fin.gamma[c + e] * rs[e];
            c1[e] = fin.eval_mode ? 0.f : (float)(sd / M);
            c2[e] = fin.eval_mode ? 0.f : (float)(sdx / M);
            if (r0 == 0) { fin.dbeta[c + e] += (float)sd; fin.dgamma[c + e] += (float)sdx; }      // one thread per column
        }
    }
    for (unsigned r = r0; r < (unsigned)M; r += U * dr) {
        float dv[U][V], yv[U][V];
        unsigned rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = min(r + u * dr, (unsigned)M - 1);                       // clamped: the loads always issue
            VLoad<T, V>::ld(d + (long)rr[u] * ldd + c, dv[u]);
            VLoad<T, V>::ld(y + (long)rr[u] * ldy + c, yv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * dr >= (unsigned)M) break;
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float xh = (yv[u][e] - mu[e]) * rs[e];
                o[e] = c0[e] * (dv[u][e] - c1[e] - xh * c2[e]);
            }
            store_vec<T, V>(d + (long)rr[u] * ldd + c, o);
        }
    }
}

// ------------------------------------------------------------------------------------------
// EncoderC table
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_table_fwd_kernel(int S, int E, int L, const float* emb, const float* w_mu,
                                                               const float* b_mu, const float* w_lv, const float* b_lv, float* table) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S * 2 * L; i += gridDim.x * blockDim.x) {
        const int s = i / (2 * L), j = i % (2 * L);
        const float* w = j < L ? w_mu + (long)j * E : w_lv + (long)(j - L) * E;
        float acc = j < L ? b_mu[j] : b_lv[j - L];
        for (int e = 0; e < E; ++e) acc += emb[(long)s * E + e] * w[e];
        table[i] = acc;
    }
}

// All operands are a few KB: every workgroup first copies them into LDS in ONE round of independent loads (the table gradient summed
// over its scatter copies in fixed order, the embedding, both head weights), then every thread forms its output element from LDS.
// (Round 2's form walked w_mu / w_lv / emb in global memory inside the dot products: 20-24 dependent rounds of L2 latency, 22 us for
// a few thousand multiply-adds.)
__global__ __launch_bounds__(256) void embed_table_bwd_kernel(int S, int E, int L, const float* emb, const float* w_mu,
                                                               const float* w_lv, const float* dTc, int copies, float* d_emb, float* d_w_mu,
                                                               float* d_b_mu, float* d_w_lv, float* d_b_lv) {
    const int L2 = 2 * L;
    extern __shared__ float sm[];
    float* dT = sm;                                      // [S][2L]: the copies of the table gradient summed (fixed order)
    float* sE = dT + S * L2;                             // [S][E]
    float* sW = sE + S * E;                              // [2L][E]: w_mu rows, then w_lv rows
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    // one index space [dEmb | dWcat | db]: the three products run side by side
    const int n0 = S * E, n1 = n0 + L2 * E, n2 = n1 + L2;
    auto dst_of = [&](int i) -> float* {
        if (i < n0) return d_emb + i;
        if (i < n1) { const int k = i - n0, j = k / E, e = k - j * E; return j < L ? d_w_mu + (long)j * E + e : d_w_lv + (long)(j - L) * E + e; }
        const int j = i - n1;
        return j < L ? d_b_mu + j : d_b_lv + (j - L);
    };
    // the gradient element this thread adds to is requested together with the operands: behind the products it was one more
    // dependent round trip of a launch that is nothing but round trips
    float* const dst0 = t0 < n2 ? dst_of(t0) : nullptr;
    const float old0 = dst0 ? *dst0 : 0.f;
    for (int i = threadIdx.x; i < S * L2; i += blockDim.x) {
        float c8[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) c8[c] = c < copies ? dTc[(long)c * S * L2 + i] : 0.f;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) v += c8[c];
        for (int c = 8; c < copies; ++c) v += dTc[(long)c * S * L2 + i];
        dT[i] = v;
    }
    for (int i = threadIdx.x; i < S * E; i += blockDim.x) sE[i] = emb[i];
    for (int i = threadIdx.x; i < L2 * E; i += blockDim.x) sW[i] = i < L * E ? w_mu[i] : w_lv[i - L * E];
    __syncthreads();
    for (int i = t0; i < n2; i += nt) {
        float acc = 0.f;
        if (i < n0) {                                    // dEmb = dT x Wcat
            const int s = i / E, e = i - s * E;
#pragma unroll 4
            for (int j = 0; j < L2; ++j) acc += dT[s * L2 + j] * sW[j * E + e];
        } else if (i < n1) {                             // dWcat = dT^T x emb
            const int k = i - n0, j = k / E, e = k - j * E;
#pragma unroll 4
            for (int s = 0; s < S; ++s) acc += dT[s * L2 + j] * sE[s * E + e];
        } else {
            const int j = i - n1;
#pragma unroll 4
            for (int s = 0; s < S; ++s) acc += dT[s * L2 + j];
        }
        if (i == t0) *dst0 = old0 + acc;
        else { float* d = dst_of(i); *d += acc; }
    }
}

// ------------------------------------------------------------------------------------------
// fusion + reparameterisation
// ------------------------------------------------------------------------------------------
template <typename ZT>
__global__ __launch_bounds__(256) void fuse_fwd_kernel(mmvae_fuse_fwd_args a) {
    const long total = (long)a.B * a.ldz;
    const float inv_n = 1.f / (float)a.n_mod;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.ldz), l = (int)(i % a.ldz);
        ZT* zp = (ZT*)a.z + i;
        if (l >= a.L) { *zp = from_f32<ZT>(0.f); continue; }
        float mu = 0.f, lv = 0.f;
        if (a.heads_a) { mu += a.heads_a[(long)b * a.ld_heads + l]; lv += a.heads_a[(long)b * a.ld_heads + a.L + l]; }
        if (a.heads_b) { mu += a.heads_b[(long)b * a.ld_heads + l]; lv += a.heads_b[(long)b * a.ld_heads + a.L + l]; }
        if (a.table) {
            // torch's Embedding device-asserts on an index outside [0, S); here the row is poisoned instead (NaN in mu / logvar /
            // z and everything downstream) and no memory outside the table is touched
            const long s = a.site[b];
            const bool ok = s >= 0 && s < a.S;
            const long sc = ok ? s : 0;
            const float bad = ok ? 0.f : __builtin_nanf("");
            mu += a.table[sc * 2 * a.L + l] + bad; lv += a.table[sc * 2 * a.L + a.L + l] + bad;
        }
        if (a.n_mod > 1) { mu *= inv_n; lv *= inv_n; }
        a.mu[(long)b * a.L + l] = mu; a.logvar[(long)b * a.L + l] = lv;
        *zp = from_f32<ZT>(mu + a.eps[(long)b * a.L + l] * expf(0.5f * lv));
    }
}

__global__ __launch_bounds__(256) void fuse_bwd_kernel(mmvae_fuse_bwd_args a, int rows_per_block, int use_lds) {
    extern __shared__ float sT[];                 // [S][2L] when use_lds
    const int L2 = 2 * a.L;
    if (use_lds) { for (int i = threadIdx.x; i < a.S * L2; i += blockDim.x) sT[i] = 0.f; __syncthreads(); }
    const int b0 = blockIdx.x * rows_per_block, b1 = min(a.B, b0 + rows_per_block);
    float* dtab = a.d_table ? a.d_table + (long)(blockIdx.x % (a.table_copies > 0 ? a.table_copies : 1)) * a.S * L2 : nullptr;
    const float inv_n = 1.f / (float)a.n_mod;
    // U elements per thread are loaded before any is processed (stores may alias loads for the compiler): with one element
    // per iteration a thread had ~7 dependent-latency loads in flight at a time and the launch was latency-bound (43 us for 40 MB)
    constexpr int U = 4;
    const long iend = (long)b1 * a.L;
    for (long i0 = (long)b0 * a.L + threadIdx.x; i0 < iend; i0 += (long)U * blockDim.x) {
        float dz[U], gm[U], gl[U], ep[U], lv[U];
        int bb[U], ll[U];
        long sidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = min(i0 + (long)u * blockDim.x, iend - 1);        // clamped: the loads always issue
            const int b = (int)(i / a.L), l = (int)(i - (long)b * a.L);
            bb[u] = b; ll[u] = l;
            dz[u] = a.dz[(long)b * a.lddz + l];
            if (a.dz2) dz[u] += a.dz2[(long)b * a.lddz + l];
            if (a.dz3) dz[u] += a.dz3[(long)b * a.lddz + l];
            gm[u] = a.g_mu ? a.g_mu[i] : 0.f; gl[u] = a.g_lv ? a.g_lv[i] : 0.f;
            ep[u] = a.eps[i]; lv[u] = a.logvar[i];
            sidx[u] = a.d_table ? a.site[b] : 0;
            if (sidx[u] < 0 || sidx[u] >= a.S) sidx[u] = -1;                  // out of range: no scatter (the forward poisoned the row)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + (long)u * blockDim.x >= iend) break;
            const int b = bb[u], l = ll[u];
            float dmu = gm[u] + dz[u];
            float dlv = gl[u] + dz[u] * ep[u] * expf(0.5f * lv[u]) * 0.5f;
            if (a.n_mod > 1) { dmu *= inv_n; dlv *= inv_n; }
            a.d_heads[(long)b * a.ld_heads + l] = dmu;
            a.d_heads[(long)b * a.ld_heads + a.L + l] = dlv;
            if (a.d_heads_lp) {
                bf16* lp = (bf16*)a.d_heads_lp + (long)b * a.ld_heads_lp;
                lp[l] = (bf16)dmu; lp[a.L + l] = (bf16)dlv;
                if (l == 0) for (int e = 2 * a.L; e < a.ld_heads_lp; ++e) lp[e] = (bf16)0.f;       // pad columns
            }
            if (a.d_table && sidx[u] >= 0) {
                const long sx = sidx[u];
                if (use_lds) { atomicAdd(&sT[sx * L2 + l], dmu); atomicAdd(&sT[sx * L2 + a.L + l], dlv); }
                else { unsafeAtomicAdd(&dtab[sx * L2 + l], dmu); unsafeAtomicAdd(&dtab[sx * L2 + a.L + l], dlv); }
            }
        }
    }
    if (use_lds && a.d_table) {
        __syncthreads();
        for (int i = threadIdx.x; i < a.S * L2; i += blockDim.x) if (sT[i] != 0.f) unsafeAtomicAdd(&dtab[i], sT[i]);
    }
}

// ------------------------------------------------------------------------------------------
// fused loss
// ------------------------------------------------------------------------------------------
template <typename GT, int V>
__device__ __forceinline__ void store_vec(GT* p, const float* v) {
    GT o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = from_f32<GT>(v[e]);
    if constexpr (sizeof(GT) * V == 16) *(f32x4*)p = *(f32x4*)o;
    else if constexpr (sizeof(GT) * V == 8) *(f32x2*)p = *(f32x2*)o;
    else if constexpr (sizeof(GT) * V == 4) *(float*)p = *(float*)o;
    else for (int e = 0; e < V; ++e) p[e] = o[e];
}

// Flat index i = r * vpr + cv walks the [B][W/V] vector grid with a grid stride.  The element offsets of the three matrices
// (prediction, target, gradient; row pitches ld[k]) are advanced incrementally: an integer division per vector made these parts
// VALU-bound (~40 lane-ops per 8 bytes), and even r * ld[k] per vector is six quarter-rate 64-bit multiply-adds that sit
// between a wave's loop iterations, i.e. in series with its memory latency.  step / fix are wave-uniform.
struct RowWalk {
    unsigned cv, dc, vpr;
    long o[3], step[3], fix[3];
    __device__ __forceinline__ RowWalk(unsigned i0, unsigned stride, unsigned vpr_, int V, long ld0, long ld1, long ld2) : vpr(vpr_) {
        const unsigned r = i0 / vpr, dr = stride / vpr;
        cv = i0 - r * vpr; dc = stride - dr * vpr;
        const long ld[3] = {ld0, ld1, ld2};
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = r * ld[k] + (long)cv * V; step[k] = dr * ld[k] + (long)dc * V; fix[k] = ld[k] - (long)vpr * V; }
    }
    __device__ __forceinline__ void next() {
        cv += dc;
        const bool wrap = cv >= vpr;
        if (wrap) cv -= vpr;
#pragma unroll
        for (int k = 0; k < 3; ++k) o[k] += step[k] + (wrap ? fix[k] : 0L);
    }
};

// U vectors per thread are loaded BEFORE any is processed: the stores of one iteration may alias the loads of the next as far
// as the compiler knows, so a one-vector loop keeps only two loads in flight per thread -- 8 MB chip-wide, which at ~2 us
// of loaded latency is the 3.7 TB/s this kernel was stuck at.
constexpr int LOSS_U = 4;

// One streaming pass over a prediction / target pair of [B][W] fp32 matrices in V-wide vectors; `f(p, t, g)` returns the loss
// term of one element and its gradient.  The hot loop has no per-vector conditions: iterations in which all U vectors of the
// thread exist run unchecked, the (< U) vectors left are taken one at a time, and the pad columns of the gradient rows (the GEMM
// operand contract: zeros up to the next multiple of 8) are written by a separate row loop.
template <typename GT, int V, bool HAS_G, typename F>
__device__ __forceinline__ float stream_part(const float* __restrict__ pred, long ld_p, const float* __restrict__ tgt, long ld_t,
                                             GT* g_out, long ld_g, int B, int W, long tid0, long stride, F f) {
    float acc = 0.f;
    const unsigned vpr = (unsigned)(W / V);
    const long total = (long)B * vpr;                            // < 2^32 checked on the host
    RowWalk w((unsigned)tid0, (unsigned)stride, vpr, V, ld_p, ld_t, ld_g);
    auto one = [&](const float (&x)[V], const float (&t)[V], long og) {
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc += f(x[e], t[e], g[e]);
        if constexpr (HAS_G) store_vec<GT, V>(g_out + og, g);
    };
    long i = tid0;
    for (; i + (LOSS_U - 1) * stride < total; i += LOSS_U * stride) {
        float x[LOSS_U][V], t[LOSS_U][V];
        long og[LOSS_U];
#pragma unroll
        for (int u = 0; u < LOSS_U; ++u) {
            VLoad<float, V>::ld(pred + w.o[0], x[u]);
            VLoad<float, V>::ld(tgt + w.o[1], t[u]);
            og[u] = w.o[2];
            w.next();
        }
#pragma unroll
        for (int u = 0; u < LOSS_U; ++u) one(x[u], t[u], og[u]);
    }
    for (; i < total; i += stride) {
        float x[V], t[V];
        VLoad<float, V>::ld(pred + w.o[0], x);
        VLoad<float, V>::ld(tgt + w.o[1], t);
        const long og = w.o[2];
        w.next();
        one(x, t, og);
    }
    if constexpr (HAS_G) {
        const int pad_end = min((int)ld_g, (W + 7) & ~7);
        if (pad_end > W)
            for (long r = tid0; r < B; r += stride)
                for (int e = W; e < pad_end; ++e) g_out[r * ld_g + e] = from_f32<GT>(0.f);
    }
    return acc;
}

template <typename GT, int V>
__device__ __forceinline__ float mse_part(const mmvae_loss_args& a, long tid0, long stride) {
    auto f = [](float x, float t, float& g) { const float d = x - t; g = 2.f * d; return d * d; };
    if (a.g_a) return stream_part<GT, V, true>(a.recon_a, a.ld_ra, a.a, a.ld_a, (GT*)a.g_a, a.ld_ga, a.B, a.A, tid0, stride, f);
    return stream_part<GT, V, false>(a.recon_a, a.ld_ra, a.a, a.ld_a, (GT*)nullptr, 0, a.B, a.A, tid0, stride, f);
}

template <typename GT, int V, bool WRT_LOGIT>
__device__ __forceinline__ float bce_part_g(const mmvae_loss_args& a, long tid0, long stride) {
    auto f = [](float pe, float te, float& g) {
        const float lp = fmaxf(fast_ln(pe), -100.f), l1p = fmaxf(fast_ln(1.f - pe), -100.f);   // v_log_f32: 1 ulp, clamp as torch
        const float pq = (1.f - pe) * pe, d = pe - te;
        // torch: grad_p = (p - t) / max(p (1 - p), 1e-12); w.r.t. the logit that times p (1 - p): exactly (p - t) unless clamped
        if constexpr (WRT_LOGIT) g = d * fminf(pq * 1e12f, 1.f);             // as EpiLoss::term (gemm_nt_epi.h): d * 1 is exact
        else g = d * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f));
        return -(te * lp + (1.f - te) * l1p);
    };
    if (a.g_b) return stream_part<GT, V, true>(a.recon_b, a.ld_rb, a.b, a.ld_b, (GT*)a.g_b, a.ld_gb, a.B, a.D, tid0, stride, f);
    return stream_part<GT, V, false>(a.recon_b, a.ld_rb, a.b, a.ld_b, (GT*)nullptr, 0, a.B, a.D, tid0, stride, f);
}

template <typename GT, int V>
__device__ __forceinline__ float bce_part(const mmvae_loss_args& a, long tid0, long stride) {
    return a.grad_b_wrt_logit ? bce_part_g<GT, V, true>(a, tid0, stride) : bce_part_g<GT, V, false>(a, tid0, stride);
}

// TAIL: the class + KL terms alone (the reconstruction terms run inside the decoder GEMMs).  The launch is then a few dependent
// HBM round trips and nothing else, so everything a thread will need is requested up front: the KL operands (KL_U elements)
// BEFORE the class rows are walked, and CE_U rows per half wave at a time -- at B = 65 536 on 512 workgroups that is one round
// trip for the KL term and two for the class term instead of three + four one after the other (26 -> 14 us).
// ce_vec (S % 4 == 0, S <= 32, 16-byte aligned rows): the class term as ONE ROW PER THREAD -- S / 4 16-byte loads, max, one exp
// per logit kept in registers, S / 4 16-byte gradient stores -- instead of a row per half wave, whose two 5-step butterflies + the
// label pick are eleven DEPENDENT ds_bpermute round trips per row (~1 500 cycles; 17 of this launch's 25 us at B = 65 536).
template <typename GT, int VA, int VD, bool TAIL = false>
__global__ __launch_bounds__(256) void vae_loss_kernel(mmvae_loss_args a, int ce_vec) {
    if (a.beta_gamma_dev) { a.beta = a.beta_gamma_dev[0]; a.gamma = a.beta_gamma_dev[1]; }     // hyper-parameters a captured graph can change
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    float n_bad = 0.f;
    if (!TAIL && a.recon_a) s[0] = mse_part<GT, VA>(a, tid0, stride);
    if (!TAIL && a.recon_b) s[1] = bce_part<GT, VD>(a, tid0, stride);
    constexpr int KL_U = TAIL ? 10 : 4;
    const long total = a.mu ? (long)a.B * a.L : 0;
    float mu0[KL_U], lv0[KL_U];
    if (TAIL && a.mu) {
#pragma unroll
        for (int u = 0; u < KL_U; ++u) { const long i = min(tid0 + u * stride, total - 1); mu0[u] = a.mu[i]; lv0[u] = a.logvar[i]; }
    }
    if (a.logits && ce_vec) {
        const int nv = a.S >> 2;
        for (long r = tid0; r < a.B; r += stride) {
            const f32x4* lp = (const f32x4*)(a.logits + r * a.ld_logits);
            f32x4 x[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = c < nv ? lp[c] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            long y = a.site[r];
            const bool ign = y == -100;                      // ignore_index, as below
            const bool bad = !ign && (y < 0 || y >= a.S);
            if (bad || ign) y = 0;
            const float w = ign ? 0.f : (a.class_weights ? a.class_weights[y] : 1.f);
            float m = -INFINITY, xy = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) { m = fmaxf(m, x[c][j]); xy = (c * 4 + j == (int)y) ? x[c][j] : xy; }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float e = c < nv ? expf(x[c][j] - m) : 0.f; x[c][j] = e; se += e; }
            s[2] += w * (m + logf(se) - xy);
            if (bad) n_bad += 1.f;
            if (a.g_c) {
                f32x4* gp = (f32x4*)(a.g_c + r * a.ld_gc);
                const float gw = a.gamma * w;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c >= nv) break;
                    f32x4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = gw * (x[c][j] / se - (c * 4 + j == (int)y ? 1.f : 0.f));
                    gp[c] = g;
                }
            }
        }
    } else if (a.logits && a.S <= 32) {
        // Class term, one row per HALF wave (lane j of the half holds logit j): the row is one coalesced 4 S-byte read, max and
        // sum are 5-step shuffles, ONE exp per logit serves the loss and the gradient.  (One thread per row -- 24 strided loads
        // and two exp per logit in a serial loop -- took 25 us of the 35 this kernel needs once the reconstruction terms run
        // inside the decoder GEMMs.)
        const int lane_ = threadIdx.x & 63, sub = lane_ & 31, half = lane_ >> 5;
        const long wave0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2, wstride = (long)gridDim.x * (blockDim.x >> 6) * 2;
        // CE_U rows per half wave are requested before the first is reduced: the loop is a chain of dependent HBM round trips
        // otherwise (8 per wave at B = 65 536 and 1024 workgroups: 20 of this kernel's 34 us)
        constexpr int CE_U = TAIL ? 8 : 4;
        for (long r0 = wave0; r0 < a.B; r0 += wstride * CE_U) {    // wave-uniform trip count
            float xs[CE_U]; long ys[CE_U];
#pragma unroll
            for (int u = 0; u < CE_U; ++u) {
                const long r = r0 + u * wstride + half;
                const bool rv = r < a.B;
                xs[u] = (rv && sub < a.S) ? a.logits[r * a.ld_logits + sub] : -INFINITY;
                ys[u] = rv ? a.site[r] : 0;
            }
#pragma unroll
            for (int u = 0; u < CE_U; ++u) {
                const long r = r0 + u * wstride + half;
                const bool rv = r < a.B, ev = rv && sub < a.S;
                const float x = xs[u];
                float m = x;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const float e = ev ? expf(x - m) : 0.f;
                float se = e;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
                long y = ys[u];
                const bool ign = y == -100;                  // F.cross_entropy's default ignore_index (losses.py:39): no loss, no gradient
                const bool bad = !ign && (y < 0 || y >= a.S);
                if (bad || ign) y = 0;
                const float xy = __shfl(x, (lane_ & 32) + (int)y, 64);
                const float w = ign ? 0.f : ((rv && a.class_weights) ? a.class_weights[y] : 1.f);
                if (sub == 0 && rv) { s[2] += w * (m + logf(se) - xy); if (bad) n_bad += 1.f; }
                if (a.g_c && ev) a.g_c[r * a.ld_gc + sub] = a.gamma * w * (e / se - (sub == (int)y ? 1.f : 0.f));
            }
        }
    } else if (a.logits) {
        for (long r = tid0; r < a.B; r += stride) {
            const float* lg = a.logits + r * a.ld_logits;
            long y = a.site[r];
            const bool ign = y == -100;                      // ignore_index, as above
            if (ign) y = 0;
            if (y < 0 || y >= a.S) { y = 0; n_bad += 1.f; }          // torch device-asserts; counted in sums[4] / out[4], see mmvae_hip.h
            float m = -INFINITY;
            for (int j = 0; j < a.S; ++j) m = fmaxf(m, lg[j]);
            float se = 0.f;
            for (int j = 0; j < a.S; ++j) se += expf(lg[j] - m);
            const float lse = m + logf(se);
            const float w = ign ? 0.f : (a.class_weights ? a.class_weights[y] : 1.f);
            s[2] += w * (lse - lg[y]);
            if (a.g_c) {
                float* g = a.g_c + r * a.ld_gc;
                for (int j = 0; j < a.S; ++j) g[j] = a.gamma * w * (expf(lg[j] - lse) - (j == y ? 1.f : 0.f));
            }
        }
    }
    if (a.mu) {
        // KL_U elements per thread are loaded before any is processed: the gradient stores may alias the next loads as far as the
        // compiler knows, and one element per iteration made this a chain of dependent round trips (13 us for 21 MB)
        for (long i0 = tid0; i0 < total; i0 += stride * KL_U) {
            float mu[KL_U], lv[KL_U];
            if (TAIL && i0 == tid0) {
#pragma unroll
                for (int u = 0; u < KL_U; ++u) { mu[u] = mu0[u]; lv[u] = lv0[u]; }
            } else {
#pragma unroll
                for (int u = 0; u < KL_U; ++u) { const long i = min(i0 + u * stride, total - 1); mu[u] = a.mu[i]; lv[u] = a.logvar[i]; }
            }
#pragma unroll
            for (int u = 0; u < KL_U; ++u) {
                const long i = i0 + u * stride;
                if (i >= total) break;
                const float ex = expf(lv[u]);
                s[3] += -0.5f * (1.f + lv[u] - mu[u] * mu[u] - ex);
                if (a.g_mu) a.g_mu[i] = a.beta * mu[u];
                if (a.g_lv) a.g_lv[i] = -0.5f * a.beta * (1.f - ex);
            }
        }
    }
    __shared__ float red[4][5];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float v = wave_sum(s[k]); if (lane == 0) red[wid][k] = v; }
    { const float v = wave_sum(n_bad); if (lane == 0) red[wid][4] = v; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const double v = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
        if (v != 0.0) unsafeAtomicAdd(a.sums + threadIdx.x, v);
    }
}

// out = {total, recon, class, kld, labels out of range} as the reference returns them (losses.py:44,46); stand-alone form of the
// tail of vae_loss_kernel (callers that keep `sums` to themselves)
__global__ void loss_finalize_kernel(const double* sums, float beta, float gamma, const float* beta_gamma_dev, float* out) {
    if (beta_gamma_dev) { beta = beta_gamma_dev[0]; gamma = beta_gamma_dev[1]; }
    const double recon = sums[0] + sums[1];
    out[0] = (float)(recon + (double)gamma * sums[2] + (double)beta * sums[3]);
    out[1] = (float)recon; out[2] = (float)sums[2]; out[3] = (float)sums[3]; out[4] = (float)sums[4];
}

template <typename OT>
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(int M, int N, const float* g, long ldg, const float* p, long ldp,
                                                           OT* out, long ldo) {
    // the pad columns of the output rows (up to the next multiple of 8) are written as zeros: the GEMM operand contract -- the LDS-DMA
    // kernels multiply them with zero weights, and 0 x (uninitialised NaN bits) is NaN
    const int Np = (int)min((long)((N + 7) & ~7), ldo);
    const long total = (long)M * Np;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / Np), c = (int)(i % Np);
        if (c >= N) { out[(long)r * ldo + c] = from_f32<OT>(0.f); continue; }
        const float pv = p[(long)r * ldp + c];
        out[(long)r * ldo + c] = from_f32<OT>(g[(long)r * ldg + c] * pv * (1.f - pv));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_if_needed_kernel(T* x, long n, const float* scale) {
    const float s = *scale;
    if (s == 1.f) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = from_f32<T>(to_f32(x[i]) * s);
}

// ------------------------------------------------------------------------------------------
// noise
// ------------------------------------------------------------------------------------------
// `offset_dev` (optional) is a device-resident running offset added to `offset`: it lets a captured hipGraph draw
// fresh noise on every replay (mmvae_counter_add advances it at the end of the step).
// 16 keep decisions (bytes) per quad from TWO Philox calls: each 32-bit word decides two bytes by its 16-bit halves against
// thresh >> 16 (keep probability quantised to 1/65536: 0.9 -> 0.899994).  One word per byte made this launch VALU-bound
// (~150 integer lane-ops per call, 15 M calls for the 59 MB of masks of one step: 42 us); counters q*4+2, q*4+3 stay unused.
__device__ __forceinline__ void mask_quad(uint8_t* mask, long n, long q, uint32_t thresh, uint64_t seed, uint64_t offset) {
        uint32_t w[4];
        const uint32_t t16 = thresh == 0xFFFFFFFFu ? 0x10000u : thresh >> 16;      // keep probability 1: every 16-bit value passes
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint32_t r[4];
            Philox::gen(seed, offset + (uint64_t)q * 4 + k, 0x4D41534Bull /* "MASK" */, r);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t a = r[2 * h], b = r[2 * h + 1];
                w[2 * k + h] = ((a & 0xffffu) < t16 ? 1u : 0u) | ((a >> 16) < t16 ? 0x100u : 0u) |
                               ((b & 0xffffu) < t16 ? 0x10000u : 0u) | ((b >> 16) < t16 ? 0x1000000u : 0u);
            }
        }
        if (q * 16 + 16 <= n) { uint4 v = {w[0], w[1], w[2], w[3]}; *(uint4*)(mask + q * 16) = v; }
        else for (long i = q * 16; i < n; ++i) mask[i] = (uint8_t)((w[(i - q * 16) >> 2] >> (8 * ((i - q * 16) & 3))) & 0xff);
}

__device__ __forceinline__ void randn_quad(float* out, long n, long q, uint64_t seed, uint64_t offset) {
        uint32_t r[4];
        Philox::gen(seed, offset + (uint64_t)q, 0x4E4F524Dull /* "NORM" */, r);
        float z[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {               // Box-Muller on (0,1] x [0,1)
            const float u1 = ((float)(r[2 * k] >> 8) + 1.0f) * (1.0f / 16777216.0f);
            const float u2 = (float)(r[2 * k + 1] >> 8) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[2 * k] = rad * cs; z[2 * k + 1] = rad * sn;
        }
        if (q * 4 + 4 <= n && (((uintptr_t)out & 15) == 0)) *(f32x4*)(out + q * 4) = f32x4{z[0], z[1], z[2], z[3]};
        else for (int k = 0; k < 4; ++k) if (q * 4 + k < n) out[q * 4 + k] = z[k];
}

// one launch for all dropout masks (one contiguous uint8 buffer) and eps of a forward pass
// Device-resident counters that a launch both reads and advances (Philox offset, Adam step count) are kept as MMVAE_CTR_COPIES
// identical copies: block L reads copy L and, when it is done, rewrites the copies L, L + nblocks, ... with the advanced value.
// No block ever reads a word another block writes, so there is no ordering to enforce, no ticket and no extra launch
// (a 1-thread counter launch costs ~4 us + a kernel boundary; a last-block ticket costs ~45 ns per block on ONE address).
__device__ __forceinline__ uint64_t ctr_read(const uint64_t* copies, unsigned L) { return copies[L % MMVAE_CTR_COPIES]; }
__device__ __forceinline__ void ctr_advance(uint64_t* copies, unsigned L, unsigned nblocks, uint64_t value) {
    __syncthreads();                                  // every thread of the block holds the old value
    if (threadIdx.x == 0 && threadIdx.y == 0)
        for (unsigned e = L; e < MMVAE_CTR_COPIES; e += nblocks) copies[e] = value;
}

__global__ __launch_bounds__(256) void noise_kernel(uint8_t* mask, long n_mask, float* eps, long n_eps, uint32_t thresh,
                                                    uint64_t seed, uint64_t offset, uint64_t* offset_dev, uint64_t advance, int copies) {
    const uint64_t base = offset_dev ? (copies ? ctr_read(offset_dev, blockIdx.x) : *offset_dev) : 0;
    offset += base;
    const long nqm = (n_mask + 15) / 16, nqe = (n_eps + 3) / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nqm + nqe; q += (long)gridDim.x * blockDim.x) {
        if (q < nqm) mask_quad(mask, n_mask, q, thresh, seed, offset);
        else randn_quad(eps, n_eps, q - nqm, seed, offset + (uint64_t)nqm * 4);
    }
    if (copies) ctr_advance(offset_dev, blockIdx.x, gridDim.x, base + advance);
}

__global__ void counter_add_kernel(uint64_t* ctr, uint64_t inc) { *ctr += inc; }

// ------------------------------------------------------------------------------------------
// minibatch assembly: out_t[i][:] = src_t[idx[i]][:] for up to MMVAE_GATHER_MAX row-major tensors that share the index vector
// (the RNA matrix, the DNA matrix and the site labels of one shuffled minibatch).  One wave per (tensor, row), 8-byte words (4-byte
// words when a width or address is only 4-byte aligned).
// ------------------------------------------------------------------------------------------
struct GatherBatch { mmvae_gather_item it[MMVAE_GATHER_MAX]; int n; };
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherBatch b, const int64_t* __restrict__ idx, int rows, long src_rows) {
    const int lane = threadIdx.x & 63;
    const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long w = wave0; w < (long)rows * b.n; w += nwaves) {
        const int t = (int)(w / rows), i = (int)(w - (long)t * rows);
        const mmvae_gather_item it = b.it[t];
        long r = idx[i];
        r = r < 0 ? 0 : (r >= src_rows ? src_rows - 1 : r);            // memory-safe on a bad index (torch would device-assert)
        const char* sp = (const char*)it.src + r * it.src_row_stride;
        char* dp = (char*)it.dst + (long)i * it.dst_row_stride;
        if (((it.row_bytes | it.src_row_stride | it.dst_row_stride | (long)(uintptr_t)it.src | (long)(uintptr_t)it.dst) & 7) == 0) {      // uniform per item
            const uint2* s = (const uint2*)sp; uint2* d = (uint2*)dp;
            for (int k = lane; k < (it.row_bytes >> 3); k += 64) d[k] = s[k];
        } else {                                                       // fp32 rows of an odd width (INPUT_DIM_* overrides): 4-byte words
            const uint32_t* s = (const uint32_t*)sp; uint32_t* d = (uint32_t*)dp;
            for (int k = lane; k < (it.row_bytes >> 2); k += 64) d[k] = s[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// AdamW, all tensors in one launch
// ------------------------------------------------------------------------------------------
// Blocks are dealt out in proportion to the tensors' sizes (first[i] .. first[i+1] belong to tensor i): a (blocks x tensors) grid gave
// every one of the 39 tensors 256 blocks -- 10 000 blocks, most of them on 128-element biases, each paying a counter read, two powf
// and a counter write (29 us for 30 MB of traffic).
struct AdamWBatch { mmvae_adamw_item items[64]; int first[65]; int n; };      // passed BY VALUE (2.8 KiB of kernel arguments): no table
                                                        // upload, so the launch is hipGraph-capturable even when gradients move
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWBatch batch, float lr, float b1, float b2,
                                                     float eps, float wd, float bc1, float rsqrt_bc2, int maximize,
                                                     uint64_t* step_dev, int copies, const float* lr_dev) {
    if (lr_dev) lr = *lr_dev;                       // a captured graph follows the LR scheduler without re-capture
    int ti = 0;
    for (int i = 1; i < batch.n; ++i) if ((int)blockIdx.x >= batch.first[i]) ti = i;     // block ranges are ascending
    const mmvae_adamw_item it = batch.items[ti];
    const int lb = (int)blockIdx.x - batch.first[ti], nb = batch.first[ti + 1] - batch.first[ti];
    const unsigned L = blockIdx.x;
    uint64_t steps_done = 0;
    if (step_dev) {                                 // graph-capturable form: step count lives on the device
        steps_done = copies ? ctr_read(step_dev, L) : *step_dev;
        const float t = (float)(steps_done + 1);
        bc1 = 1.f - powf(b1, t);
        rsqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
    }
    const float step = lr / bc1;
    // U elements per thread are requested before any is updated: the stores of one element may alias the loads of the next as far
    // as the compiler knows, so one element per iteration was a chain of 4-5 dependent round trips (14 us for 42 MB)
    constexpr int U = 4;
    const long stride = (long)nb * blockDim.x;
    for (long i0 = (long)lb * blockDim.x + threadIdx.x; i0 < it.n; i0 += U * stride) {
        float g[U], p[U], m[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = min(i0 + u * stride, (long)it.n - 1);
            g[u] = it.g[i]; p[u] = it.p[i]; m[u] = it.m[i]; v[u] = it.v[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i >= it.n) break;
            const float gg = maximize ? -g[u] : g[u];
            float pp = p[u] * (1.f - lr * wd);
            const float mm_ = b1 * m[u] + (1.f - b1) * gg;
            const float vv = b2 * v[u] + (1.f - b2) * gg * gg;
            pp -= step * mm_ / (sqrtf(vv) * rsqrt_bc2 + eps);
            it.p[i] = pp; it.m[i] = mm_; it.v[i] = vv;
        }
    }
    if (copies) ctr_advance(step_dev, L, gridDim.x, steps_done + 1);
}

static inline int grid_for(long items, int per_block = 256, int cap = 2048) {
    long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace mm

using namespace mm;

extern "C" int mmvae_abi_version(void) { return 19; }

extern "C" int mmvae_prep_weights(const mmvae_prep_item* items_dev, int32_t n_items, void* stream) {
    if (!items_dev || n_items <= 0) return MMVAE_ERR_ARG;
    // workgroups per item: 48 -> 14.9 us, 96 -> 10.6, 160 -> 8.8, 320 -> 9.0 (the largest item has 288 tiles; every tile is a load -> store round trip)
    hipLaunchKernelGGL(prep_weights_kernel, dim3(160, n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_bn_finalize(const mmvae_bn_finalize_args* a, void* stream) {
    if (!a || !a->sum || !a->sumsq || !a->gamma || !a->beta || !a->mean || !a->rstd || !a->scale || !a->shift) return MMVAE_ERR_ARG;
    if (a->M < 2 || a->N <= 0) return MMVAE_ERR_ARG;                          /* BatchNorm1d needs > 1 row in training */
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((a->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_bn_eval_coeffs(int32_t N, const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float eps, float* scale, float* shift, float* mean, float* rstd, void* stream) {
    if (N <= 0 || !gamma || !beta || !rm || !rv || !scale || !shift) return MMVAE_ERR_ARG;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, gamma, beta, rm, rv, eps, scale, shift, mean, rstd);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_bn_bwd_finalize(const mmvae_bn_bwd_finalize_args* a, void* stream) {
    if (!a || !a->sum_d || !a->sum_dx || !a->gamma || !a->rstd || !a->dgamma || !a->dbeta || !a->coef) return MMVAE_ERR_ARG;
    if (a->M <= 0 || a->N <= 0) return MMVAE_ERR_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_bn_bwd_apply(int32_t dtype, int32_t M, int32_t N, void* d, int64_t ldd, const void* y, int64_t ldy,
                                  const float* mean, const float* rstd, const float* coef, void* stream) {
    if (M <= 0 || N <= 0 || !d || !y || !mean || !rstd || !coef || (long)M * N >= (1L << 32)) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMVAE_BF16) {
        if (N % 8 || ldd % 8 || ldy % 8 || ((uintptr_t)d & 15) || ((uintptr_t)y & 15)) return MMVAE_ERR_ARG;
        if (256 % (N / 8) == 0)
            hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<bf16, 8, 4>), dim3(grid_for((long)M * N / 8, 256 * 4, 1024)), dim3(256), 0, st, M, N, (bf16*)d, ldd, (const bf16*)y, ldy, mean, rstd, coef, BnBwdFin{});
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, 8>), dim3(grid_for((long)M * N / 8, 256, 4096)), dim3(256), 0, st, M, N, (bf16*)d, ldd, (const bf16*)y, ldy, mean, rstd, coef);
    } else {
        if (N % 4 || ldd % 4 || ldy % 4 || ((uintptr_t)d & 15) || ((uintptr_t)y & 15)) return MMVAE_ERR_ARG;
        if (256 % (N / 4) == 0)
            hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<float, 4, 4>), dim3(grid_for((long)M * N / 4, 256 * 4, 1024)), dim3(256), 0, st, M, N, (float*)d, ldd, (const float*)y, ldy, mean, rstd, coef, BnBwdFin{});
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3(grid_for((long)M * N / 4, 256, 4096)), dim3(256), 0, st, M, N, (float*)d, ldd, (const float*)y, ldy, mean, rstd, coef);
    }
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_bn_bwd_finalize_apply(int32_t dtype, int32_t M, int32_t N, void* d, int64_t ldd, const void* y, int64_t ldy,
                                           const float* mean, const float* rstd, const double* sum_d, const double* sum_dx, const float* gamma,
                                           float* dgamma, float* dbeta, int32_t eval_mode, void* stream) {
    if (M <= 0 || N <= 0 || !d || !y || !mean || !rstd || !sum_d || !sum_dx || !gamma || !dgamma || !dbeta || (long)M * N >= (1L << 32)) return MMVAE_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const BnBwdFin fin{sum_d, sum_dx, gamma, dgamma, dbeta, eval_mode};
    if (dtype == MMVAE_BF16) {
        if (N % 8 || ldd % 8 || ldy % 8 || ((uintptr_t)d & 15) || ((uintptr_t)y & 15) || 256 % (N / 8) != 0) return MMVAE_ERR_ARG;
        hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<bf16, 8, 4, true>), dim3(grid_for((long)M * N / 8, 256 * 4, 1024)), dim3(256), 0, st, M, N, (bf16*)d, ldd, (const bf16*)y, ldy, mean, rstd, nullptr, fin);
    } else {
        if (N % 4 || ldd % 4 || ldy % 4 || ((uintptr_t)d & 15) || ((uintptr_t)y & 15) || 256 % (N / 4) != 0) return MMVAE_ERR_ARG;
        hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<float, 4, 4, true>), dim3(grid_for((long)M * N / 4, 256 * 4, 1024)), dim3(256), 0, st, M, N, (float*)d, ldd, (const float*)y, ldy, mean, rstd, nullptr, fin);
    }
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_embed_table_fwd(int32_t S, int32_t E, int32_t L, const float* emb, const float* w_mu, const float* b_mu,
                                     const float* w_lv, const float* b_lv, float* table, void* stream) {
    if (S <= 0 || E <= 0 || L <= 0 || !emb || !w_mu || !b_mu || !w_lv || !b_lv || !table) return MMVAE_ERR_ARG;
    hipLaunchKernelGGL(embed_table_fwd_kernel, dim3((S * 2 * L + 63) / 64), dim3(64), 0, (hipStream_t)stream, S, E, L, emb, w_mu, b_mu, w_lv, b_lv, table);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_embed_table_bwd(int32_t S, int32_t E, int32_t L, const float* emb, const float* w_mu, const float* w_lv,
                                     const float* d_table, int32_t table_copies, float* d_emb, float* d_w_mu, float* d_b_mu, float* d_w_lv,
                                     float* d_b_lv, void* stream) {
    if (S <= 0 || E <= 0 || L <= 0 || !emb || !w_mu || !w_lv || !d_table || !d_emb || !d_w_mu || !d_b_mu || !d_w_lv || !d_b_lv) return MMVAE_ERR_ARG;
    if (table_copies < 1) table_copies = 1;
    const size_t lds = ((size_t)S * 2 * L + (size_t)S * E + (size_t)2 * L * E) * sizeof(float);
    if (lds > 64 * 1024) return MMVAE_ERR_ARG;                    // 60 KB at latent 128, embed 32, 24 sites
    const int work = S * E + 2 * L * E + 2 * L;          // one output element per thread
    hipLaunchKernelGGL(embed_table_bwd_kernel, dim3((work + 255) / 256), dim3(256), lds, (hipStream_t)stream, S, E, L, emb, w_mu, w_lv, d_table, table_copies, d_emb, d_w_mu, d_b_mu, d_w_lv, d_b_lv);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_fuse_reparam_fwd(const mmvae_fuse_fwd_args* a, void* stream) {
    if (!a || a->B <= 0 || a->L <= 0 || !a->eps || !a->mu || !a->logvar || !a->z || a->ldz < a->L) return MMVAE_ERR_ARG;
    const int present = (a->heads_a != nullptr) + (a->heads_b != nullptr) + (a->table != nullptr);
    if (present == 0 || present != a->n_mod) return MMVAE_ERR_ARG;
    if (a->table && (!a->site || a->S <= 0)) return MMVAE_ERR_ARG;
    const int grid = grid_for((long)a->B * a->ldz);
    if (a->z_dtype == MMVAE_BF16) hipLaunchKernelGGL(fuse_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(fuse_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_fuse_reparam_bwd(const mmvae_fuse_bwd_args* a, void* stream) {
    if (!a || a->B <= 0 || a->L <= 0 || a->n_mod <= 0 || !a->dz || !a->eps || !a->logvar || !a->d_heads) return MMVAE_ERR_ARG;
    if (a->d_table && (!a->site || a->S <= 0)) return MMVAE_ERR_ARG;
    const int rows_per_block = 128;          // 512 workgroups at B = 65 536: two per CU
    const int grid = (a->B + rows_per_block - 1) / rows_per_block;
    const size_t lds = a->d_table ? (size_t)a->S * 2 * a->L * sizeof(float) : 0;
    const int use_lds = lds > 0 && lds <= 48 * 1024;
    hipLaunchKernelGGL(fuse_bwd_kernel, dim3(grid), dim3(256), use_lds ? lds : 0, (hipStream_t)stream, *a, rows_per_block, use_lds);
    MM_CHECK_LAUNCH();
    return 0;
}

template <typename GT>
static int launch_loss(const mmvae_loss_args* a, hipStream_t st) {
    auto al = [](const void* p, int64_t ld, int v, size_t es) { return p == nullptr || (ld % v == 0 && ((uintptr_t)p % (v * es)) == 0); };
    int va = 1, vd = 1;
    for (int v : {4, 2}) {
        if (va == 1 && a->recon_a && a->A % v == 0 && al(a->recon_a, a->ld_ra, v, 4) && al(a->a, a->ld_a, v, 4) && al(a->g_a, a->ld_ga, v, sizeof(GT))) va = v;
        if (vd == 1 && a->recon_b && a->D % v == 0 && al(a->recon_b, a->ld_rb, v, 4) && al(a->b, a->ld_b, v, 4) && al(a->g_b, a->ld_gb, v, sizeof(GT))) vd = v;
    }
    // vectors of the stream parts + the per-row class term + the latent elements: without the reconstruction parts (their loss runs
    // inside the decoder GEMMs) the class / KL terms alone must still fill the chip (a row per thread, not four rows on 64 workgroups)
    const long work = (long)a->B * ((a->recon_a ? a->A / va : 0) + (a->recon_b ? a->D / vd : 0) + (a->logits ? 4 * a->S : 0) + (a->mu ? a->L : 0) + 1);
    static const int wg_cap = getenv("MMVAE_LOSS_WG") ? atoi(getenv("MMVAE_LOSS_WG")) * 256 : 1024;
    int grid = mm::grid_for(work, 256 * 4, wg_cap);
    // class / KL terms alone (reconstruction terms inside the decoder GEMMs): every workgroup ends in f64 atomics on the same few
    // addresses; with the row and element loops unrolled 512 workgroups are enough to cover the latency and halve that tail
    if (!a->recon_a && !a->recon_b && grid > 512) grid = 512;
#define MM_LOSS(VA, VD) hipLaunchKernelGGL((vae_loss_kernel<GT, VA, VD>), dim3(grid), dim3(256), 0, st, *a, ce_vec ? 1 : 0)
    static const int tail_env = getenv("MMVAE_LOSS_TAIL") ? atoi(getenv("MMVAE_LOSS_TAIL")) : 2;          // A/B switch: 0 general form, 1 without ce_vec
    const bool ce_vec = tail_env == 2 && a->logits && a->S <= 32 && a->S % 4 == 0 && a->ld_logits % 4 == 0 && ((uintptr_t)a->logits & 15) == 0 &&
                        (!a->g_c || (a->ld_gc % 4 == 0 && ((uintptr_t)a->g_c & 15) == 0));
    if (tail_env && !a->recon_a && !a->recon_b) {
        // every workgroup ends in f64 atomics on the same addresses (~13 ns each, one after the other): a row per thread, no more
        if (ce_vec) grid = (int)std::min<long>(512, (a->B + 255) / 256);
        hipLaunchKernelGGL((vae_loss_kernel<GT, 1, 1, true>), dim3(grid), dim3(256), 0, st, *a, ce_vec ? 1 : 0);
    } else
    if (va == 4 && vd == 4) MM_LOSS(4, 4); else if (va == 4 && vd == 2) MM_LOSS(4, 2); else if (va == 4) MM_LOSS(4, 1);
    else if (va == 2 && vd == 4) MM_LOSS(2, 4); else if (va == 2 && vd == 2) MM_LOSS(2, 2); else if (va == 2) MM_LOSS(2, 1);
    else if (vd == 4) MM_LOSS(1, 4); else if (vd == 2) MM_LOSS(1, 2); else MM_LOSS(1, 1);
#undef MM_LOSS
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_vae_loss(const mmvae_loss_args* a, void* stream) {
    if (!a || a->B <= 0 || !a->sums) return MMVAE_ERR_ARG;
    if (a->recon_a && (!a->a || a->A <= 0)) return MMVAE_ERR_ARG;
    if (a->recon_b && (!a->b || a->D <= 0)) return MMVAE_ERR_ARG;
    if (a->logits && (!a->site || a->S <= 0)) return MMVAE_ERR_ARG;
    if (a->mu && (!a->logvar || a->L <= 0)) return MMVAE_ERR_ARG;
    if ((long)a->B * (a->A > a->D ? a->A : a->D) >= (1L << 32)) return MMVAE_ERR_ARG;
    const int gdt = a->g_a ? a->g_a_dtype : a->g_b_dtype;
    if (a->g_a && a->g_b && a->g_a_dtype != a->g_b_dtype) return MMVAE_ERR_DTYPE;
    if (gdt == MMVAE_BF16) return launch_loss<bf16>(a, (hipStream_t)stream);
    return launch_loss<float>(a, (hipStream_t)stream);
}

extern "C" int mmvae_loss_finalize(const double* sums, float beta, float gamma, const float* beta_gamma_dev, float* out4, void* stream) {
    if (!sums || !out4) return MMVAE_ERR_ARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums, beta, gamma, beta_gamma_dev, out4);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_sigmoid_bwd(int32_t M, int32_t N, const float* g, int64_t ldg, const float* p, int64_t ldpp,
                                 void* out, int32_t out_dtype, int64_t ldo, void* stream) {
    if (M <= 0 || N <= 0 || !g || !p || !out) return MMVAE_ERR_ARG;
    const int grid = grid_for((long)M * N);
    if (out_dtype == MMVAE_BF16) hipLaunchKernelGGL(sigmoid_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, N, g, ldg, p, ldpp, (bf16*)out, ldo);
    else hipLaunchKernelGGL(sigmoid_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, M, N, g, ldg, p, ldpp, (float*)out, ldo);
    MM_CHECK_LAUNCH();
    return 0;
}

struct ScaleBatch { mmvae_scale_item it[MMVAE_SCALE_MAX]; int count; };
__global__ __launch_bounds__(256) void scale_many_kernel(ScaleBatch b, const float* scale) {
    const float s = *scale;
    if (s == 1.f) return;
    for (int k = 0; k < b.count; ++k) {
        const long n = b.it[k].n;
        if (b.it[k].dtype == MMVAE_BF16) {
            bf16* x = (bf16*)b.it[k].x;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = from_f32<bf16>(to_f32(x[i]) * s);
        } else {
            float* x = (float*)b.it[k].x;
            for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
        }
    }
}

extern "C" int mmvae_scale_many(const mmvae_scale_item* items, int32_t n_items, const float* scale_dev, void* stream) {
    if (!items || n_items <= 0 || n_items > MMVAE_SCALE_MAX || !scale_dev) return MMVAE_ERR_ARG;
    ScaleBatch b; b.count = n_items;
    long nmax = 0;
    for (int k = 0; k < n_items; ++k) {
        if (!items[k].x || items[k].n <= 0 || (items[k].dtype != MMVAE_BF16 && items[k].dtype != MMVAE_F32)) return MMVAE_ERR_ARG;
        b.it[k] = items[k];
        if (items[k].n > nmax) nmax = items[k].n;
    }
    hipLaunchKernelGGL(scale_many_kernel, dim3(grid_for(nmax)), dim3(256), 0, (hipStream_t)stream, b, scale_dev);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_scale_if_needed(void* x, int32_t dtype, int64_t n, const float* scale_dev, void* stream) {
    if (!x || n <= 0 || !scale_dev) return MMVAE_ERR_ARG;
    const int grid = grid_for(n);
    if (dtype == MMVAE_BF16) hipLaunchKernelGGL(scale_if_needed_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16*)x, n, scale_dev);
    else hipLaunchKernelGGL(scale_if_needed_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)x, n, scale_dev);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_noise(uint8_t* mask, int64_t n_mask, float keep_prob, float* eps, int64_t n_eps, uint64_t seed,
                           uint64_t offset, uint64_t* offset_dev, int32_t advance, void* stream) {
    if ((n_mask > 0 && (!mask || ((uintptr_t)mask & 15))) || (n_eps > 0 && !eps) || n_mask < 0 || n_eps < 0 || n_mask + n_eps == 0 ||
        keep_prob < 0.f || keep_prob > 1.f || (advance && !offset_dev)) return MMVAE_ERR_ARG;
    const uint64_t used = (uint64_t)((n_mask + 15) / 16 * 4 + (n_eps + 3) / 4);       // counter values this call consumes
    const double t = (double)keep_prob * 4294967296.0;
    const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    hipLaunchKernelGGL(noise_kernel, dim3(grid_for((n_mask + 15) / 16 + (n_eps + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       mask, n_mask, eps, n_eps, thresh, seed, offset, offset_dev, used, advance ? 1 : 0);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_gather_rows(const mmvae_gather_item* items_host, int32_t n_items, const int64_t* idx_dev, int32_t rows,
                                 int64_t src_rows, void* stream) {
    if (!items_host || n_items <= 0 || n_items > MMVAE_GATHER_MAX || !idx_dev || rows <= 0 || src_rows <= 0) return MMVAE_ERR_ARG;
    GatherBatch b; b.n = n_items;
    for (int k = 0; k < n_items; ++k) {
        const mmvae_gather_item& it = items_host[k];
        if (!it.src || !it.dst || it.row_bytes <= 0 || it.row_bytes % 4 || it.src_row_stride % 4 || it.dst_row_stride % 4 ||
            ((uintptr_t)it.src & 3) || ((uintptr_t)it.dst & 3)) return MMVAE_ERR_ARG;
        b.it[k] = it;
    }
    const long waves = (long)rows * n_items;
    int grid = (int)((waves + 3) / 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, b, idx_dev, rows, (long)src_rows);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_counter_add(uint64_t* counter_dev, uint64_t inc, void* stream) {
    if (!counter_dev) return MMVAE_ERR_ARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, inc);
    MM_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmvae_adamw_step(const mmvae_adamw_item* items_host, int32_t n_items, float lr, float beta1,
                                float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, int32_t maximize,
                                uint64_t* step_dev, int32_t advance, const float* lr_dev, void* stream) {
    if (!items_host || n_items <= 0 || (!step_dev && (bias_corr1 <= 0.f || bias_corr2 <= 0.f))) return MMVAE_ERR_ARG;
    if (advance && (!step_dev || n_items > 64)) return MMVAE_ERR_ARG;       // one launch = one tick of the counter
    if (step_dev) { bias_corr1 = 1.f; bias_corr2 = 1.f; }
    for (int base = 0; base < n_items; base += 64) {
        AdamWBatch batch;
        const int n = n_items - base < 64 ? n_items - base : 64;
        batch.n = n; batch.first[0] = 0;
        for (int i = 0; i < n; ++i) {
            batch.items[i] = items_host[base + i];
            if (!batch.items[i].p || !batch.items[i].g || !batch.items[i].m || !batch.items[i].v || batch.items[i].n <= 0) return MMVAE_ERR_ARG;
            batch.first[i + 1] = batch.first[i] + grid_for(batch.items[i].n, 256 * 4, 256);       // 4 elements per thread, <= 256 blocks per tensor
        }
        const int gx = batch.first[n];
        if (advance && gx > MMVAE_CTR_COPIES) return MMVAE_ERR_ARG;
        hipLaunchKernelGGL(adamw_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, batch, lr, beta1, beta2, eps,
                           weight_decay, bias_corr1, 1.0f / sqrtf(bias_corr2), maximize, step_dev, advance ? 1 : 0, lr_dev);
        MM_CHECK_LAUNCH();
    }
    return 0;
}
