// extern "C" entry points (include/getmi355.h) -> internal launchers.
#include "common.h"
#include "../../include/getmi355.h"

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; int M, N, K, lda, ldb; void* out0; void* out1; int ldo; const float* bias;
    const float* resid; const float* rowscale; int rows_per_sample; const bf16_t* aux; int ldaux;
    const uint8_t* mask; const float* mask_token; int regions_per_sample; float* colsum; int row0; int prescaled; int blocked;
};
int getmi_gemm_tn_colsum_rows(int M, int N);
int getmi_gemm_tn_colsum_rows_max(int M, int N);
int getmi_gemm_tn_colsum_dense();
int getmi_gemm_tn_blocked_ok_impl(int M, int N, int K);
int getmi_cu_scope_enter(int all_cus);
void getmi_cu_scope_leave(int token);
struct CuScope {     // one CU-budget snapshot per C-ABI call (see gemm.hip)
    int token;
    explicit CuScope(bool all_cus) : token(getmi_cu_scope_enter(all_cus ? 1 : 0)) {}
    ~CuScope() { getmi_cu_scope_leave(token); }
};
int getmi_colreduce_impl(const float* partial, float* out0, float* out1, float* out2, int R, int C, int accumulate,
                         hipStream_t st);
int getmi_gemm_tn_impl(int epi, const GemmArgs& a, hipStream_t st);
int getmi_wgrad_nsplit(int M, int N1, int N2);
size_t getmi_wgrad_pair_workspace_bytes(int M, int N1a, int N2a, int N1b, int N2b);
int getmi_gemm_wgrad_pair_impl(const bf16_t*, const bf16_t*, int, int, int, int, float*, int, const bf16_t*, const bf16_t*, int,
                               int, int, int, float*, int, int, float*, size_t, hipStream_t);
int getmi_set_cu_limit_impl(int n);
int getmi_cu_count_impl();
int getmi_gemm_wgrad_impl(const bf16_t* X, const bf16_t* Y, int M, int N1, int N2, int ldx, int ldy, float* out,
                          int accumulate, float* slabs, size_t slab_bytes, hipStream_t st);
int getmi_ln_fwd_impl(const float*, const float*, const float*, bf16_t*, float*, float*, int, int, float, hipStream_t);
size_t getmi_ln_bwd_workspace_bytes(int M, int D);
int getmi_ln_bwd_impl(const bf16_t*, const float*, const float*, const float*, const float*, const void*, int, const float*,
                      int, void*, int, bf16_t*, float*, float*, float*, int, int, int, void*, size_t, hipStream_t);
int getmi_attn_fwd_impl(const bf16_t*, bf16_t*, bf16_t*, float*, int, int, int, float, const float*, hipStream_t);
size_t getmi_attn_bwd_workspace_bytes(int B, int T, int H);
int getmi_attn_bwd_impl(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, const float*, bf16_t*, float*, float*, int, int, int,
                        float, const float*, void*, size_t, hipStream_t);
int getmi_cast_pad_impl(const float*, bf16_t*, size_t, int, int, hipStream_t);
int getmi_cast_transpose_impl(const float*, bf16_t*, int, int, hipStream_t);
int getmi_qkv_bias_impl(const float*, const float*, float*, int, hipStream_t);
int getmi_fill_cls_impl(float*, const float*, int, int, int, hipStream_t);
size_t getmi_embed_bwd_workspace_bytes_impl(int B, int N, int D);
int getmi_embed_bwd_prep_impl(const float*, const uint8_t*, bf16_t*, float*, float*, float*, int, int, int, int, void*,
                              size_t, hipStream_t);
int getmi_head_fwd_impl(const bf16_t*, const float*, const float*, float*, float*, int, int, int, int, hipStream_t);
size_t getmi_head_bwd_workspace_bytes_impl(int B, int N, int D, int OUT);
int getmi_head_bwd_impl(const float*, const float*, const bf16_t*, const float*, bf16_t*, float*, int, int, int, int, int,
                        void*, size_t, hipStream_t);
size_t getmi_loss_workspace_bytes_impl(void);
int getmi_loss_fwd_impl(int, const float*, const float*, float*, float*, size_t, int, int, int, int, float, float, void*,
                        size_t, hipStream_t);
int getmi_scale_by_scalar_impl(const float*, const float*, float, float*, size_t, hipStream_t);
int getmi_gather_rows_impl(const bf16_t*, const int64_t*, bf16_t*, int, int, int, hipStream_t);
int getmi_mask_compact_impl(const uint8_t*, int, int, int64_t*, int64_t*, int*, hipStream_t);
size_t getmi_colsum_workspace_bytes_impl(int N);
int getmi_colsum_bf16_impl(const bf16_t*, float*, int, int, int, int, void*, size_t, hipStream_t);

size_t getmi_scale_cast_workspace_bytes_impl(int M, int D);
int getmi_scale_cast_impl(const float*, const float*, int, bf16_t*, float*, int, int, int, void*, size_t, hipStream_t);
size_t getmi_scale_cast_noalloc_max_bytes_impl(void);
int getmi_scale_cast_noalloc_impl(const float*, const float*, int, bf16_t*, float*, int, int, int, hipStream_t);

struct AdamGroup { float lr, weight_decay, beta1, beta2, eps, step_size, inv_sqrt_bc2, pad; };
int getmi_adamw_step_impl(const void* jobs, int nflat, int ntile, const AdamGroup* groups, int ngroups, hipStream_t st);

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define BF(p) reinterpret_cast<const bf16_t*>(p)
#define BFW(p) reinterpret_cast<bf16_t*>(p)

extern "C" {

int getmi_cast_pad_f32_bf16(const float* src, void* dst, int64_t R, int C, int Cp, void* stream) {
    return getmi_cast_pad_impl(src, BFW(dst), (size_t)R, C, Cp, ST(stream));
}
int getmi_cast_transpose_f32_bf16(const float* src, void* dst, int R, int C, void* stream) {
    return getmi_cast_transpose_impl(src, BFW(dst), R, C, ST(stream));
}
int getmi_qkv_bias(const float* q_bias, const float* v_bias, float* out, int D, void* stream) {
    return getmi_qkv_bias_impl(q_bias, v_bias, out, D, ST(stream));
}
int getmi_gemm_tn_blocked_ok(int M, int N, int K) { return getmi_gemm_tn_blocked_ok_impl(M, N, K); }
size_t getmi_gemm_tn_workspace_bytes(int M, int N) { return (size_t)getmi_gemm_tn_colsum_rows_max(M, N) * N * sizeof(float); }
int getmi_gemm_tn(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, void* out0, void* out1,
                  int ldo, const float* bias, const float* resid, const float* rowscale, int rows_per_sample,
                  const void* aux, int ldaux, const uint8_t* mask, const float* mask_token, int regions_per_sample,
                  float* dbias, void* workspace, size_t workspace_bytes, int flags, void* stream) {
    CuScope scope(false);      // tile choice, partial-row count and launch all see one CU budget
    GemmArgs a{BF(A), BF(B), M, N, K, lda, ldb, out0, out1, ldo, bias, resid, rowscale, rows_per_sample, BF(aux), ldaux,
               mask, mask_token, regions_per_sample, nullptr, 0, (flags & 1) ? 1 : 0, (flags & 2) ? 1 : 0};
    if (dbias) {
        GETMI_CHECK_ARG(epi == 3 || epi == 7, "gemm_tn: dbias (fused column sums) is only produced by the DGELU / DMUL epilogues");
        // partial rows for ANY tile height the launcher may pick (rows a variant does not write stay zero and add nothing)
        const size_t need = getmi_gemm_tn_workspace_bytes(M, N);
        GETMI_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0),
                        "gemm_tn: dbias needs a workspace of getmi_gemm_tn_workspace_bytes(M, N)");
        a.colsum = reinterpret_cast<float*>(workspace);
        // kernels write two partial rows per (tile, wave-row): every entry of the rows this launch uses is written (the
        // staged epilogue zero-fills the row it has no sums for), so the table is cleared only for the diagnostic generation
        if (!getmi_gemm_tn_colsum_dense() && hipMemsetAsync(workspace, 0, need, ST(stream)) != hipSuccess) {
            getmi_set_error("gemm_tn: workspace memset failed");
            return GETMI_ERR_LAUNCH;
        }
    }
    if (int rc = getmi_gemm_tn_impl(epi, a, ST(stream))) return rc;
    if (dbias) {
        const int rows = getmi_gemm_tn_colsum_dense() ? getmi_gemm_tn_colsum_rows(M, N) : getmi_gemm_tn_colsum_rows_max(M, N);
        return getmi_colreduce_impl(a.colsum, dbias, nullptr, nullptr, rows, N, 0, ST(stream));
    }
    return 0;
}
size_t getmi_gemm_wgrad_workspace_bytes(int M, int N1, int N2) {
    CuScope scope(true);       // sized for all CUs: an upper bound under any getmi_set_cu_limit
    return (size_t)getmi_wgrad_nsplit(M, N1, N2) * N1 * N2 * sizeof(float);
}
int getmi_gemm_wgrad(const void* X, const void* Y, int M, int N1, int N2, int ldx, int ldy, float* out, int accumulate,
                     void* workspace, size_t workspace_bytes, void* stream) {
    CuScope scope(false);
    return getmi_gemm_wgrad_impl(BF(X), BF(Y), M, N1, N2, ldx, ldy, out, accumulate, reinterpret_cast<float*>(workspace),
                                 workspace_bytes, ST(stream));
}
int getmi_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int M,
                        int D, float eps, void* stream) {
    return getmi_ln_fwd_impl(x, gamma, beta, BFW(y), mean, rstd, M, D, eps, ST(stream));
}
size_t getmi_layernorm_bwd_workspace_bytes(int M, int D) { return getmi_ln_bwd_workspace_bytes(M, D); }
int getmi_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, const float* rowscale, int rows_per_sample, float* dx, void* dz, float* dgamma,
                        float* dbeta, float* dbias_prev, int accumulate, int M, int D, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return getmi_ln_bwd_impl(BF(dy), x, mean, rstd, gamma, dres, 0, rowscale, rows_per_sample, dx, 0, BFW(dz), dgamma, dbeta,
                             dbias_prev, accumulate, M, D, workspace, workspace_bytes, ST(stream));
}
int getmi_layernorm_bwd_stream(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const void* dres, int dres_is_bf16, const float* rowscale, int rows_per_sample, void* dx,
                               int dx_is_bf16, void* dz, float* dgamma, float* dbeta, float* dbias_prev, int accumulate, int M,
                               int D, void* workspace, size_t workspace_bytes, void* stream) {
    return getmi_ln_bwd_impl(BF(dy), x, mean, rstd, gamma, dres, dres_is_bf16 != 0, rowscale, rows_per_sample, dx,
                             dx_is_bf16 != 0, BFW(dz), dgamma, dbeta, dbias_prev, accumulate, M, D, workspace, workspace_bytes,
                             ST(stream));
}
int getmi_attention_fwd(const void* qkv, void* out, void* out_lo, float* lse, int B, int T, int H, float scale,
                        const float* sample_scale, void* stream) {
    return getmi_attn_fwd_impl(BF(qkv), BFW(out), BFW(out_lo), lse, B, T, H, scale, sample_scale, ST(stream));
}
size_t getmi_attention_bwd_workspace_bytes(int B, int T, int H) { return getmi_attn_bwd_workspace_bytes(B, T, H); }
int getmi_attention_bwd(const void* qkv, const void* out, const void* out_lo, const void* dout, const float* lse,
                        void* dqkv, float* dq_bias, float* dv_bias, int B, int T, int H, float scale,
                        const float* sample_scale, void* workspace, size_t workspace_bytes, void* stream) {
    return getmi_attn_bwd_impl(BF(qkv), BF(out), BF(out_lo), BF(dout), lse, BFW(dqkv), dq_bias, dv_bias, B, T, H, scale,
                               sample_scale, workspace, workspace_bytes, ST(stream));
}
int getmi_fill_cls(float* x0, const float* cls_token, int B, int T, int D, void* stream) {
    return getmi_fill_cls_impl(x0, cls_token, B, T, D, ST(stream));
}
int getmi_embed_bwd_prep(const float* dx0, const uint8_t* mask, void* dz, float* dcls, float* dmask_token, float* dbias,
                         int accumulate, int B, int N, int D, void* workspace, size_t workspace_bytes, void* stream) {
    return getmi_embed_bwd_prep_impl(dx0, mask, BFW(dz), dcls, dmask_token, dbias, accumulate, B, N, D, workspace,
                                     workspace_bytes, ST(stream));
}
int getmi_head_fwd(const void* hf, const float* W, const float* bias, float* logits, float* pred, int B, int N, int D,
                   int out_dim, void* stream) {
    return getmi_head_fwd_impl(BF(hf), W, bias, logits, pred, B, N, D, out_dim, ST(stream));
}
int getmi_head_bwd(const float* dpred, const float* logits, const void* hf, const float* W, void* dh, float* dwb,
                   int accumulate, int B, int N, int D, int out_dim, void* workspace, size_t workspace_bytes,
                   void* stream) {
    return getmi_head_bwd_impl(dpred, logits, BF(hf), W, BFW(dh), dwb, accumulate, B, N, D, out_dim, workspace,
                               workspace_bytes, ST(stream));
}
int getmi_loss_fwd(int kind, const float* pred, const float* obs, float* loss, float* grad_unit, int64_t R, int C, int ldp,
                   int ldo, int ldg, float weight, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    return getmi_loss_fwd_impl(kind, pred, obs, loss, grad_unit, (size_t)R, C, ldp, ldo, ldg, weight, eps, workspace,
                               workspace_bytes, ST(stream));
}
int getmi_scale_by_scalar(const float* in, const float* scalar_dev, float weight, float* out, int64_t n, void* stream) {
    return getmi_scale_by_scalar_impl(in, scalar_dev, weight, out, (size_t)n, ST(stream));
}
int getmi_gather_rows_bf16(const void* src, const int64_t* idx, void* dst, int n, int D, int scatter, void* stream) {
    return getmi_gather_rows_impl(BF(src), idx, BFW(dst), n, D, scatter, ST(stream));
}
int getmi_mask_compact(const uint8_t* mask, int B, int N, int64_t* token_rows, int64_t* region_rows, int* count,
                       void* stream) {
    return getmi_mask_compact_impl(mask, B, N, token_rows, region_rows, count, ST(stream));
}
int getmi_colsum_bf16(const void* X, float* out, int accumulate, int M, int N, int ld, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return getmi_colsum_bf16_impl(BF(X), out, accumulate, M, N, ld, workspace, workspace_bytes, ST(stream));
}

size_t getmi_embed_bwd_workspace_bytes(int B, int N, int D) { return getmi_embed_bwd_workspace_bytes_impl(B, N, D); }
size_t getmi_head_bwd_workspace_bytes(int B, int N, int D, int out_dim) {
    return getmi_head_bwd_workspace_bytes_impl(B, N, D, out_dim);
}
size_t getmi_loss_workspace_bytes(void) { return getmi_loss_workspace_bytes_impl(); }
size_t getmi_colsum_workspace_bytes(int N) { return getmi_colsum_workspace_bytes_impl(N); }

size_t getmi_scale_cast_workspace_bytes(int M, int D) { return getmi_scale_cast_workspace_bytes_impl(M, D); }
size_t getmi_scale_cast_noalloc_max_bytes(void) { return getmi_scale_cast_noalloc_max_bytes_impl(); }
int getmi_scale_cast_noalloc(const float* dx, const float* rowscale, int rows_per_sample, void* dz, float* dbias,
                             int accumulate, int M, int D, void* stream) {
    return getmi_scale_cast_noalloc_impl(dx, rowscale, rows_per_sample, BFW(dz), dbias, accumulate, M, D, ST(stream));
}
int getmi_scale_cast(const float* dx, const float* rowscale, int rows_per_sample, void* dz, float* dbias, int accumulate,
                     int M, int D, void* workspace, size_t workspace_bytes, void* stream) {
    return getmi_scale_cast_impl(dx, rowscale, rows_per_sample, BFW(dz), dbias, accumulate, M, D, workspace,
                                 workspace_bytes, ST(stream));
}

size_t getmi_gemm_wgrad_pair_workspace_bytes(int M, int N1a, int N2a, int N1b, int N2b) {
    CuScope scope(true);
    return getmi_wgrad_pair_workspace_bytes(M, N1a, N2a, N1b, N2b);
}
int getmi_gemm_wgrad_pair(const void* Xa, const void* Ya, int N1a, int N2a, int ldxa, int ldya, float* outa, int accumulate_a,
                          const void* Xb, const void* Yb, int N1b, int N2b, int ldxb, int ldyb, float* outb, int accumulate_b,
                          int M, void* workspace, size_t workspace_bytes, void* stream) {
    CuScope scope(false);
    return getmi_gemm_wgrad_pair_impl(BF(Xa), BF(Ya), N1a, N2a, ldxa, ldya, outa, accumulate_a, BF(Xb), BF(Yb), N1b, N2b, ldxb,
                                      ldyb, outb, accumulate_b, M, reinterpret_cast<float*>(workspace), workspace_bytes,
                                      ST(stream));
}
int getmi_set_cu_limit(int n) { return getmi_set_cu_limit_impl(n); }
int getmi_cu_count(void) { return getmi_cu_count_impl(); }

int getmi_adamw_step(const void* jobs, int n_flat_jobs, int n_tile_jobs, const getmi_adamw_group* groups, int n_groups,
                     void* stream) {
    static_assert(sizeof(getmi_adamw_group) == sizeof(AdamGroup), "group layout");
    return getmi_adamw_step_impl(jobs, n_flat_jobs, n_tile_jobs, reinterpret_cast<const AdamGroup*>(groups), n_groups,
                                 ST(stream));
}
}  // extern "C"
