/* getmi355 -- C ABI of the MI355X (gfx950) kernels behind the GET region-transformer hot path.
 *
 * The reference (GET-Foundation/get_model) has no native code and therefore no FFI for this path: the
 * "interface each entry point replaces" is the eager PyTorch call site in the reference's Python, cited
 * per function below (paths relative to the reference's get_model/).  Host code (get_model_amd/, Python)
 * binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (torch tensors' data_ptr()).
 *  - bf16 buffers are passed as void* (raw 16-bit words); fp32 as float*.
 *  - stream is a hipStream_t passed as void* (0 = default stream).  Calls only ENQUEUE work and return.
 *  - return 0 on success; non-zero on error (1 = bad argument, 2 = launch failure); message via
 *    getmi_last_error() (thread-local).  No exceptions, no exit().
 *  - the library never allocates device memory: scratch comes in through (workspace, workspace_bytes),
 *    sized by the matching *_workspace_bytes() query.  Workspaces must be 256-byte aligned.
 *  - re-entrant.  Global state: one-time kernel attribute set-up and environment switches (idempotent), the atomic CU
 *    limit of getmi_set_cu_limit (read once per call), and the per-GPU scratch address of getmi_scale_cast_noalloc.
 */
#ifndef GETMI355_H
#define GETMI355_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* getmi_last_error(void);
int getmi_version(void);          /* 100 = 1.0.0 */
const char* getmi_arch(void);     /* "gfx950" */

/* ---- packing -------------------------------------------------------------------------------- */
/* fp32 [R,C] -> bf16 [R,Cp], zero-padding columns C..Cp-1 (Cp % 4 == 0).  Replaces the implicit fp32
 * operand use of nn.Linear (model/modules.py:101-103, model/transformer.py:57,97,172,175). */
int getmi_cast_pad_f32_bf16(const float* src, void* dst, int64_t R, int C, int Cp, void* stream);
/* fp32 [R,C] -> bf16 [C,R]: transposed weight copy consumed by the dgrad GEMMs (autograd of F.linear). */
int getmi_cast_transpose_f32_bf16(const float* src, void* dst, int R, int C, void* stream);
/* out[3D] = [q_bias, 0, v_bias]   (model/transformer.py:50-56 torch.cat) */
int getmi_qkv_bias(const float* q_bias, const float* v_bias, float* out, int D, void* stream);

/* ---- GEMM ----------------------------------------------------------------------------------- */
/* C[M,N] = A[M,K] * B[N,K]^T, bf16 operands, fp32 accumulate, fused epilogue `epi`:
 *   0 BF16        out0(bf16)[M,ldo] = acc (+bias)                       qkv: transformer.py:57
 *   1 BIAS_GELU   out0(bf16)=u=acc+bias ; out1(bf16)=gelu_erf(u)        fc1+act: transformer.py:172-173
 *   2 BIAS_RESID  out0(f32) = resid + rowscale[row/rows_per_sample]*(acc+bias)
 *                                                                       proj/fc2 + DropPath + residual:
 *                                                                       transformer.py:97,175,247-248
 *   3 DGELU       out0(bf16) = acc * gelu'(aux)                         autograd of :173
 *   4 EMBED       out0(f32)[b*(R+1)+1+r] = mask[m] ? mask_token : acc+bias  (R = regions_per_sample)
 *                                                                       modules.py:101-103, model.py:520-525
 *   5 F32         out0(f32) = acc (+bias)                               head_mask: model.py:528
 *   6 BIAS_GELUG  out0(bf16)=gelu'(u), u=acc+bias ; out1(bf16)=gelu_erf(u)   training form of 1: saves the derivative
 *   7 DMUL        out0(bf16) = acc * aux                                autograd of :173 with aux = gelu'(u) from 6
 * K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0, 16-byte aligned pointers.  Unused pointers may be NULL; for epilogues 1
 * and 6 out0 may be NULL too (inference: only gelu(u) is stored). */
int getmi_gemm_tn(int epi, const void* A, const void* B, int M, int N, int K, int lda, int ldb, void* out0,
                  void* out1, int ldo, const float* bias, const float* resid, const float* rowscale,
                  int rows_per_sample, const void* aux, int ldaux, const uint8_t* mask, const float* mask_token,
                  int regions_per_sample, float* dbias, void* workspace, size_t workspace_bytes, int flags,
                  void* stream);
/* rowscale with epilogue 6: both outputs are multiplied by rowscale[row/rows_per_sample] (the DropPath factor of
 * transformer.py:248 folded into the saved activations, so that no scaled copy of the gradient is needed in backward).
 * flags bit 0 with epilogue 2: the A operand already carries that factor: out0 = resid + acc + rowscale*bias.
 * flags bit 1 with epilogue 6 (out0) and 7 (aux): that tensor is in the BLOCKED layout, a private hand-over format between
 * the two epilogues (the forward of fc1 writes gelu'(u), only the backward of fc2 reads it): for row block rb = row/16
 * and column block cb = col/64 the 2-KiB block ((rb*(N/64) + cb)) holds [p][lane][8 bf16] with lane = 16*c + r and the
 * eight values = columns 64cb + 32p + 4c .. +3 and 64cb + 32p + 16 + 4c .. +3 of row 16rb + r -- the MFMA accumulator
 * layout, so it is stored and re-loaded as whole 1-KiB pieces without the LDS transposition of the row-major form.  The
 * buffer holds ceil(M/16)*16 rows of N; ldo / ldaux must equal N.  Allowed only where getmi_gemm_tn_blocked_ok says so. */
int getmi_gemm_tn_blocked_ok(int M, int N, int K);
/* DGELU / DMUL only: dbias[N] (optional) = column sums of out0 before rounding = bias gradient of the Linear whose
 * pre-activation (or its GELU derivative) is `aux` (fc1.bias); needs workspace of getmi_gemm_tn_workspace_bytes(M, N). */
size_t getmi_gemm_tn_workspace_bytes(int M, int N);
/* out[N1,N2] (+)= X[M,N1]^T * Y[M,N2]  (weight gradient; autograd of F.linear).  bf16 operands, fp32 out. */
size_t getmi_gemm_wgrad_workspace_bytes(int M, int N1, int N2);
int getmi_gemm_wgrad(const void* X, const void* Y, int M, int N1, int N2, int ldx, int ldy, float* out,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm (model/transformer.py:198,221,243,248,344,364; eps from :315) ------------------ */
int getmi_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* mean,
                        float* rstd, int M, int D, float eps, void* stream);
size_t getmi_layernorm_bwd_workspace_bytes(int M, int D);
/* dx = dres + LN'(dy); dz = bf16(rowscale*dx) (optional); dgamma, dbeta, dbias_prev = column sums. */
int getmi_layernorm_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* dres, const float* rowscale, int rows_per_sample,
                        float* dx, void* dz_bf16, float* dgamma, float* dbeta, float* dbias_prev, int accumulate,
                        int M, int D, void* workspace, size_t workspace_bytes, void* stream);

/* The same with the residual-gradient stream in bf16 on either side (autograd of `x + branch(LN(x))`,
 * model/transformer.py:247-248, where PyTorch keeps that gradient in fp32): dres / dx are bf16 when the matching flag is
 * non-zero, fp32 otherwise (both zero == getmi_layernorm_bwd).  With dx bf16 and rowscale null, dz may be null while
 * dbias_prev is wanted: dz would equal dx bit for bit, so the caller uses dx.  12 instead of 16 bytes per element. */
int getmi_layernorm_bwd_stream(const void* dy_bf16, const float* x, const float* mean, const float* rstd,
                               const float* gamma, const void* dres, int dres_is_bf16, const float* rowscale,
                               int rows_per_sample, void* dx, int dx_is_bf16, void* dz_bf16, float* dgamma, float* dbeta,
                               float* dbias_prev, int accumulate, int M, int D, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- attention (model/transformer.py:58-96), packed qkv[B,T,3,H,64] bf16, head_dim 64 ---------- */
/* out_lo (may be null): bf16(O - float(out)), what the bf16 rounding of the output dropped.  The backward forms
 * delta = rowsum(dO * O) from out + out_lo: with the rounded output alone every row of dS misses sum 0 by
 * dO.(O - bf16(O)), which only the q_bias gradient (a cancellation sum over all of dS) can see. */
/* sample_scale (may be null): [B] fp32, the per-sample DropPath factor of the attention branch (keep / keep_prob,
 * model/transformer.py:247) folded into the output: out = sample_scale[b] * softmax(QK^T)V, and samples whose factor is 0
 * are skipped (zero rows) in forward and backward.  getmi_attention_bwd then takes `out` as produced and dout = the
 * gradient w.r.t. that scaled output. */
int getmi_attention_fwd(const void* qkv, void* out, void* out_lo, float* lse, int B, int T, int H, float scale,
                        const float* sample_scale, void* stream);
size_t getmi_attention_bwd_workspace_bytes(int B, int T, int H);
/* dq_bias / dv_bias (each [H*64] fp32, may be null): the gradients of Attention.q_bias / v_bias
 * (model/transformer.py:52-57) = column sums over all tokens of the Q and V thirds of d(qkv), taken in fp32
 * (dq_bias = scale * sum_key (sum_q dS[q][key]) K[key] from the fp32 dS; dv_bias from the dV accumulators).
 * out_lo: the forward's residual output, or null. */
int getmi_attention_bwd(const void* qkv, const void* out, const void* out_lo, const void* dout, const float* lse,
                        void* dqkv, float* dq_bias, float* dv_bias, int B, int T, int H, float scale,
                        const float* sample_scale, void* workspace, size_t workspace_bytes, void* stream);

/* ---- region embed glue (model/model.py:520-525,571-573) ---------------------------------------- */
int getmi_fill_cls(float* x0, const float* cls_token, int B, int T, int D, void* stream);
size_t getmi_embed_bwd_workspace_bytes(int B, int N, int D);
int getmi_embed_bwd_prep(const float* dx0, const uint8_t* mask, void* dz_bf16, float* dcls, float* dmask_token,
                         float* dbias, int accumulate, int B, int N, int D, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- expression head + softplus (model/model.py:575-576, model/modules.py:255-258) -------------- */
int getmi_head_fwd(const void* hf_bf16, const float* W, const float* bias, float* logits, float* pred, int B,
                   int N, int D, int out_dim, void* stream);
size_t getmi_head_bwd_workspace_bytes(int B, int N, int D, int out_dim);
int getmi_head_bwd(const float* dpred, const float* logits, const void* hf_bf16, const float* W, void* dh_bf16,
                   float* dwb, int accumulate, int B, int N, int D, int out_dim, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- losses (model/model.py:144-149 with torch.nn.PoissonNLLLoss / torch.nn.MSELoss, mean) ------ */
size_t getmi_loss_workspace_bytes(void);
/* kind 0 = Poisson NLL (log_input=False, full=False), 1 = MSE.  loss[0] = weight*mean(...);
 * grad_unit = d(mean loss)/d(pred) (unweighted). */
int getmi_loss_fwd(int kind, const float* pred, const float* obs, float* loss, float* grad_unit, int64_t R,
                   int C, int ldp, int ldo, int ldg, float weight, float eps, void* workspace,
                   size_t workspace_bytes, void* stream);
int getmi_scale_by_scalar(const float* in, const float* scalar_dev, float weight, float* out, int64_t n,
                          void* stream);

/* ---- gradient entering a residual branch: dz = bf16(rowscale[row/rows_per_sample]*dx), dbias = colsum(dz)
 *      (autograd of `x + drop_path(branch)`, model/transformer.py:247-248) ----------------------------- */
size_t getmi_scale_cast_workspace_bytes(int M, int D);
int getmi_scale_cast(const float* dx, const float* rowscale, int rows_per_sample, void* dz_bf16, float* dbias,
                     int accumulate, int M, int D, void* workspace, size_t workspace_bytes, void* stream);

/* No counterpart in the reference.  getmi_scale_cast for callers that CANNOT allocate or hand over a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same kernels, same results, the
 * column-sum partials live in a 3 MiB device buffer built into the library (one copy per GPU, held until the process
 * ends).  Calls no hipMalloc / hipFree or other allocating routine.  Shapes whose getmi_scale_cast_workspace_bytes(M, D)
 * exceeds getmi_scale_cast_noalloc_max_bytes() are rejected (e.g. D = 1024 with M > 3072).  Every call on a GPU
 * shares that GPU's buffer: calls that overlap on one GPU (two streams, two threads) race.
 * Set-up, once per device and OUTSIDE any stream capture: getmi_scale_cast_noalloc_max_bytes() (with that device
 * current) or one eager getmi_scale_cast_noalloc call -- the first of them on a device looks up the buffer's address
 * (hipGetSymbolAddress: may load the code object, not a capturable operation).  Every later call is two kernel
 * launches and nothing else, and may be captured. */
size_t getmi_scale_cast_noalloc_max_bytes(void);
int getmi_scale_cast_noalloc(const float* dx, const float* rowscale, int rows_per_sample, void* dz_bf16, float* dbias,
                             int accumulate, int M, int D, void* stream);

/* ---- masked-row gather/scatter (model/model.py:527,535), column sums (bias gradients) ----------- */
int getmi_gather_rows_bf16(const void* src, const int64_t* idx, void* dst, int n, int D, int scatter,
                           void* stream);
/* Row lists of the masked regions of mask[B,N] (uint8, non-zero = masked) in ascending order, i.e. the row order of
 * the boolean indexing `x[:, 1:][mask]` / `x_original[mask]` (model/model.py:527,535): token_rows[i] = b*(N+1)+1+r
 * (rows of the [B*(N+1), D] token matrix), region_rows[i] = b*N+r; count[0] = number of masked regions.  All device
 * memory, sized B*N; nothing is read back by the library (the caller reads `count` when it needs the shape). */
int getmi_mask_compact(const uint8_t* mask, int B, int N, int64_t* token_rows, int64_t* region_rows, int* count,
                       void* stream);
size_t getmi_colsum_workspace_bytes(int N);
int getmi_colsum_bf16(const void* X, float* out, int accumulate, int M, int N, int ld, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Two weight gradients over the SAME M rows in one launch (e.g. attn.proj and attn.qkv of one block: 9 + 27 output
 * tiles share the CUs like one 36-tile problem); same semantics as two getmi_gemm_wgrad calls. */
size_t getmi_gemm_wgrad_pair_workspace_bytes(int M, int N1a, int N2a, int N1b, int N2b);
int getmi_gemm_wgrad_pair(const void* Xa, const void* Ya, int N1a, int N2a, int ldxa, int ldya, float* outa, int accumulate_a,
                          const void* Xb, const void* Yb, int N1b, int N2b, int ldxb, int ldyb, float* outb, int accumulate_b,
                          int M, void* workspace, size_t workspace_bytes, void* stream);

/* ---- sharing the GPU with communication kernels.  No counterpart in the reference (its GEMMs come from vendor
 *      libraries).  The persistent GEMM and the weight-gradient GEMM launch one workgroup per CU, each taking the
 *      whole CU; while RCCL's channel workgroups are resident (gradient all-reduce overlapped with backward) some
 *      CUs cannot host them and the displaced workgroups would run as a second round.  getmi_set_cu_limit(n) makes
 *      those kernels size their grids for n CUs (0 = all CUs again) and returns the previous limit; process-wide and
 *      atomic.  Every entry point reads the limit ONCE when it is called, so a change from another thread never splits
 *      one call; the *_workspace_bytes queries always answer for all CUs (an upper bound under any limit), so a
 *      workspace sized before a change is still large enough after it.  getmi_cu_count() = the CU count the next call
 *      will assume. */
int getmi_set_cu_limit(int n);
int getmi_cu_count(void);

/* ---- fused multi-tensor AdamW: replaces torch.optim.AdamW as the reference configures it
 *      (get_model/optim.py:96-130, config/finetune_tutorial.yaml:29-35) plus the per-forward weight casts of
 *      autocast.  `jobs` is a DEVICE table of n_flat_jobs + n_tile_jobs records of 8 x int64
 *      {p, g, m, v, w_bf16 [rows,cols] or 0, w_bf16_t [cols,rows] or 0, a, b}:
 *        flat job: a = first element, b = count (<= 4096) | group << 40
 *        tile job (64 x 64 tile of a 2-D weight, rows % 64 == cols % 64 == 0): a = r0 | c0 << 32,
 *                  b = cols | rows << 20 | group << 40
 *      `groups` is a HOST array (<= 64): step_size = lr / (1 - beta1^t), inv_sqrt_bc2 = 1 / sqrt(1 - beta2^t). */
typedef struct {
    float lr, weight_decay, beta1, beta2, eps, step_size, inv_sqrt_bc2, reserved;
} getmi_adamw_group;
int getmi_adamw_step(const void* jobs, int n_flat_jobs, int n_tile_jobs, const getmi_adamw_group* groups, int n_groups,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
