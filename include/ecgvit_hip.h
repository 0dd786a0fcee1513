/*
 * ecgvit_hip.h -- C-ABI of libecgvit_hip.so: the MI355X (gfx950) kernels behind the ECG-ViT train step.
 *
 * The reference (StefanHeng/ECG-Representation-Learning) has NO FFI / operator registry for this path:
 * its boundary is the Python class surface `EcgVitConfig` / `EcgVit.forward` (ecg_transformer/models/
 * ecg_vit.py:26-149) and the train-step body (ecg_transformer/models/train.py:268-283), and every
 * numeric op below is one the reference reaches through third-party `vit-pytorch==0.33.2` -> `torch.nn`
 * (reference call sites cited per entry point).  This header is therefore the build's own design for
 * what sits UNDER that Python surface; `INTEGRATION.md` shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless named `h_*`.
 *   - the caller owns every buffer (PyTorch caching allocator in the shipped host code); kernels never
 *     allocate, never synchronise, keep no global state, and are stream-ordered on `stream`
 *     (a `hipStream_t` passed as `void*`; NULL = the legacy default stream).
 *   - return 0 on success; ECGVIT_EINVAL for an unsupported shape/argument (nothing launched);
 *     ECGVIT_ELAUNCH if hipGetLastError() reported a launch failure.  Nothing throws across the ABI.
 *   - `dtype` selects the ACTIVATION element type: ECGVIT_F32 (parity path, exact-f32 MFMA / VALU) or
 *     ECGVIT_BF16 (throughput path, bf16 MFMA with f32 accumulate).  Parameters, gradients, optimiser
 *     state, LayerNorm statistics, logits and losses are always f32.
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef ECGVIT_HIP_H
#define ECGVIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECGVIT_OK 0
#define ECGVIT_EINVAL 1
#define ECGVIT_ELAUNCH 2

#define ECGVIT_F32 0
#define ECGVIT_BF16 1
#define ECGVIT_FP8_E4M3 2 /* OCP e4m3fn, 1 byte: GEMM operands only (ecgvit_gemm A/B, ecgvit_fp8_*)   */
#define ECGVIT_BF8_E5M2 3 /* OCP e5m2,   1 byte: the A operand of input-gradient products            */

/* library / build identification: "ecgvit-hip gfx950 <abi-version>" */
const char *ecgvit_version(void);
int ecgvit_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM: C[M,N] = epilogue( alpha * op(A) . op(B) )            (f32 accumulate)
 * replaces: nn.Linear forward/backward reached from vit_pytorch Attention.to_qkv / to_out / FeedForward.net
 * (reference ecg_vit.py:141 -> ViT.forward) and, in the f32 parity path, the batched QK^T / PV products.
 * ------------------------------------------------------------------------------------------------ */
#define ECGVIT_GEMM_NT 0 /* A[M,K] row-major, B[N,K] row-major :  C = A . B^T   (Linear forward)      */
#define ECGVIT_GEMM_NN 1 /* A[M,K] row-major, B[K,N] row-major :  C = A . B     (Linear input grad)   */
#define ECGVIT_GEMM_TN 2 /* A[K,M] row-major, B[K,N] row-major :  C = A^T . B   (Linear weight grad)  */

/* epilogue flags; applied in this order to v = alpha * acc */
#define ECGVIT_EPI_BIAS 1       /* v += bias[n]                                    (f32 bias)        */
#define ECGVIT_EPI_GELU 2       /* aux[m,n] = v ; v = gelu_erf(v)                  (exact erf GELU)  */
#define ECGVIT_EPI_GELU_BWD 4   /* v *= gelu_erf'(aux[m,n])                                          */
#define ECGVIT_EPI_RESIDUAL 8   /* v += residual[m,n]                                                */
#define ECGVIT_EPI_ACCUM 16     /* v += C[m,n]   (read-modify-write of the output)                   */
#define ECGVIT_EPI_DROPOUT 32   /* v = keep(seed, m*N+n) ? v / (1-p') : 0 ; applied after GELU / GELU_BWD,
                                   before RESIDUAL (the mask is a pure function of (seed, element)).  bf16 outputs: one hash per four
                                   consecutive elements, 8 bits each: p' = round(256 p) / 256 (0 < p < 1/512: ECGVIT_EINVAL); f32 outputs:
                                   one hash per pair, 16 bits each: p' = p.  The same rule holds for every dropout_p of this header
                                   (ecgvit_embed_finish / _bwd, ecgvit_layernorm_bwd_fused, ecgvit_dropout_apply): by element type        */
#define ECGVIT_EPI_GELU_GRAD_AUX 128 /* modifies EPI_GELU: aux[m,n] = gelu_erf'(v) * (the EPI_DROPOUT multiplier of this element, if any)
                                   instead of v -- everything the backward of `dropout(gelu(.))` needs, so that the input-gradient GEMM
                                   of the next Linear finishes with EPI_MUL_AUX alone (no erf, no mask hash in the backward)    */
#define ECGVIT_EPI_MUL_AUX 256  /* v *= aux[m,n]                                                                              */
#define ECGVIT_EPI_QUANT_OUT 512 /* additionally q8_out[m,n] = saturate(C[m,n] as stored / *q8_scale) in q8_format (ECGVIT_FP8_E4M3 | ECGVIT_BF8_E5M2),
                                   *q8_amax = max(*q8_amax, max |C| as stored): the 8-bit copy the next Linear's product consumes, written
                                   by the producer instead of by a separate quantise pass (8-bit A.B^T launches only)                        */
#define ECGVIT_EPI_NO_OUT 1024  /* with ECGVIT_EPI_QUANT_OUT, on the two FFN-wide emitting bodies of the 8-bit A.B^T kernel (BIAS|GELU|GELU_GRAD_AUX[|DROPOUT] and
                                   MUL_AUX|COLSUM): C is NOT written (C may be NULL) -- q8_out, *q8_amax, aux and colsum_out are exactly what the same
                                   call without the flag produces (of the bf16-rounded values C would have held).  For a consumer chain that reads
                                   only the 8-bit copy: 128 KiB less to store per 256 x 256 tile (ABI 5)                                       */
#define ECGVIT_EPI_AUX8 2048    /* modifies GELU_GRAD_AUX (bf16 products) and MUL_AUX: the saved tensor gelu'(v) x dropout multiplier is stored / read as e4m3 BYTES
                                   [M, ldaux] (ldaux in bytes) instead of bf16 (ABI 6): the tensor is private to the FFN-up forward and the FFN-down input gradient,
                                   790 MB per layer at 128 512 x 3072 whose HBM stream costs each of the two launches ~85 us (tools/gemm_ab.py --aux-ld0); values
                                   in [-0.14, 1.13] / (1 - p): three mantissa bits, relative error <= 2^-4 per element, unbiased.  Large A.B^T kernel only
                                   (ecgvit_gemm_kernel() == ECGVIT_KERNEL_GEMM_NT for the same descriptor), flag sets BIAS|GELU|GELU_GRAD_AUX[|DROPOUT] and
                                   MUL_AUX|COLSUM (8-bit operands: also with QUANT_OUT [|NO_OUT]); anything else: ECGVIT_EINVAL                                                                          */
#define ECGVIT_EPI_COLSUM 64    /* additionally colsum_out[n] = sum_m C[m,n] (of the values as stored): the bias gradient of
                                   the Linear whose output gradient this GEMM produces. Needs `workspace` of at least
                                   max(ecgvit_colsum_workspace(M,N), 8*ceil(M/256)*N) bytes. Deterministic two-stage sum. */

typedef struct ecgvit_gemm_desc {
    int32_t layout;    /* ECGVIT_GEMM_*                                             */
    int32_t dtype;     /* element type of A and B: ECGVIT_F32 | ECGVIT_BF16; or ECGVIT_FP8_E4M3 | ECGVIT_BF8_E5M2 = the 8-bit
                          format of A with B in e4m3: ECGVIT_GEMM_NT (K % 128 == 0, lda/ldb % 16 == 0, bf16 output), or ECGVIT_GEMM_TN
                          = weight gradients dW = dY8^T . X8 (M, N % 256 == 0, K >= 4096, lda/ldb % 16 == 0, f32 output, bias / accumulate
                          epilogues only; scale_a * scale_b is applied to the sum) */
    int32_t out_dtype; /* element type of C, aux, residual                          */
    int32_t epilogue;  /* OR of ECGVIT_EPI_*                                        */
    int32_t M, N, K;
    int32_t batch1, batch2; /* batched problems: z = z1 * batch2 + z2 (both >= 1)   */
    const void *A; int64_t lda, strideA1, strideA2;
    const void *B; int64_t ldb, strideB1, strideB2;
    void *C;       int64_t ldc, strideC1, strideC2;
    const float *bias;                    /* [N] f32                                 */
    const void *residual; int64_t ldr;    /* [M,N] out_dtype (not batched)           */
    void *aux;            int64_t ldaux;  /* [M,N] out_dtype (not batched)           */
    float alpha;
    float dropout_p;      /* in [0,1)                                                */
    uint64_t dropout_seed;
    void *workspace;      /* optional split-K slabs (bf16 TN); see ecgvit_gemm_workspace */
    int64_t workspace_bytes;
    float *colsum_out;    /* [N] f32, with ECGVIT_EPI_COLSUM */
    int32_t tiles_per_workgroup; /* large A.B^T products (gemm_nt launches) only. 0: persistent launch, one workgroup per CU walks a
                             static share of the output tiles (fastest when the launch owns the GPU). k > 0: ceil(tiles / k) workgroups
                             of about k tiles each, handed out by the hardware dispatcher as CUs free up -- use it when other kernels
                             (RCCL collectives overlapped with the backward pass) hold CUs, where a static share would leave the
                             workgroups that start late a full share behind. Results are bit-identical either way. Weight-gradient
                             (ECGVIT_GEMM_TN) products ignore the field: their K-slicing and sum order never depend on it. */
    void *q8_out; int64_t ldq8; const float *q8_scale; float *q8_amax; int32_t q8_format; /* with ECGVIT_EPI_QUANT_OUT */
    const float *scale_a, *scale_b; /* optional device scalars multiplied into alpha: the per-tensor scales of 8-bit operands
                             (x ~= q * scale), read by the kernel -- no host round trip between the quantise pass and the product */
} ecgvit_gemm_desc;

int ecgvit_gemm(const ecgvit_gemm_desc *d, void *stream);
/* which kernel family ecgvit_gemm would launch for this descriptor (nothing is launched, pointers are not dereferenced but must be
 * the real ones: alignment decides eligibility).  Lets a profiler attribute a call to a kernel symbol without restating the dispatch. */
#define ECGVIT_KERNEL_NONE 0        /* the call would return ECGVIT_EINVAL                                   */
#define ECGVIT_KERNEL_GEMM_F32 1    /* gemm_f32_kernel: exact-f32 MFMA parity path                           */
#define ECGVIT_KERNEL_GEMM_BF16 2   /* gemm_bf16_kernel: small / ragged bf16 products                        */
#define ECGVIT_KERNEL_GEMM_NT 3     /* gemm_nt_kernel / gemm_nt_kernel_4w (its four-wave body: plain K >= 1536, bias + residual K >= 768): persistent 256x256x64 A.B^T (bf16 or 8-bit operands) */
#define ECGVIT_KERNEL_GEMM_WGRAD 4  /* gemm_wgrad_kernel_4w / gemm_wgrad8_kernel_4w: streaming split-K weight gradients (four-wave bodies) */
int ecgvit_gemm_kernel(const ecgvit_gemm_desc *d);
/* bytes of workspace with which the call would use its preferred split-K factor (0 = none needed) */
int64_t ecgvit_gemm_workspace(const ecgvit_gemm_desc *d);

/* ------------------------------------------------------------------------------------------------
 * fp8 operand path (BASELINE.json configs[4]; nothing in the reference: its live trainer is f32, models/train.py:197).
 * Segment form: table[2s] = first element, table[2s+1] = element count of segment s inside x / y (multiples of 8), scale / amax
 * indexed by s; table == NULL with nseg == 1 means the single segment [0, count).  `count` = the longest segment.
 * ------------------------------------------------------------------------------------------------ */
/* amax[s] = max(amax[s], max |x|) over bf16 x */
int ecgvit_fp8_amax(const void *x, const int64_t *table, int nseg, int64_t count, float *amax, void *stream);
/* y = saturate(x / scale[s]) in `format` (ECGVIT_FP8_E4M3 | ECGVIT_BF8_E5M2), one byte per element at the same element offsets;
 * amax_next (optional) accumulates max |x| per segment for the next step's scale (delayed scaling) */
int ecgvit_fp8_quantize(const void *x, void *y, const int64_t *table, int nseg, int64_t count, int format, const float *scale,
                        float *amax_next, void *stream);
/* scale[i] = amax[i] / FORMAT_MAX where amax[i] > 0 (else kept; 1.0 if never set); amax[i] = 0.  formats: per-entry, or NULL = format_all */
int ecgvit_fp8_scale_update(float *scale, float *amax, int n, const int32_t *formats, int format_all, void *stream);

/* ------------------------------------------------------------------------------------------------
 * patch embedding front end.  replaces: einops Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)')
 * inside vit_pytorch ViT.to_patch_embedding (reference ecg_vit.py:141, shape probe :277).
 * ------------------------------------------------------------------------------------------------ */
/* patches[(b*n + p) * ld + j*C + c] = x[b][c][p*P + j]; columns [C*P, ld) are zero-filled. x is f32. */
int ecgvit_patch_gather(const float *x, void *patches, int B, int C, int L, int P, int64_t ld, int dtype, void *stream);
/* Same gather with the reference's input transforms fused into the load (next row f2; preprocess/transform.py:18-35 Normalize,
 * :140-154 TimeEndPad, :175-185 TimeOut; wired at preprocess/ptb_dataset.py:132-149): x_raw is (B, C, L_raw) f32,
 * value(b,c,s) = s < L_raw and s not in [timeout_start[b], +timeout_len[b]) ? (x_raw - mean[c]) * inv_std[c] : 0, for s < L = n*P.
 * timeout_start / timeout_len: int32 [B] or both NULL (eval: no TimeOut). */
int ecgvit_patch_gather_transform(const float *x_raw, void *patches, int B, int C, int L_raw, int L, int P, int64_t ld,
                                  const float *mean, const float *inv_std, const int32_t *timeout_start,
                                  const int32_t *timeout_len, int dtype, void *stream);
/* X[b*N + 0] = cls + pos[0];  X[b*N + 1 + p] = tok[b*n + p] + pos[1 + p]   (N = n + 1; ViT.forward: cat CLS, += pos)
 * optional embedding dropout (p = emb_dropout_p, mask = f(seed, element index in X)). cls/pos are f32. */
int ecgvit_embed_finish(const void *tok, const float *cls, const float *pos, void *X, int B, int n, int d,
                        float dropout_p, uint64_t seed, int dtype, void *stream);
/* backward of embed_finish: dtok[b*n+p] = dX[b*N+1+p] ; dpos[t] = sum_b dX[b*N+t] ; dcls = sum_b dX[b*N] (f32 grads, overwritten) */
int ecgvit_embed_bwd(const void *dX, void *dtok, float *dcls, float *dpos, int B, int n, int d,
                     float dropout_p, uint64_t seed, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (eps 1e-5, biased variance, affine).  replaces: vit_pytorch PreNorm.norm / mlp_head[0].
 * ------------------------------------------------------------------------------------------------ */
int ecgvit_layernorm_fwd(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                         int64_t rows, int d, float eps, int dtype, void *stream);
/* fp8 operand path: the same forward (bf16, d in 64 * {4, 8, 12, 16, 24, 32}) that also writes y8 = saturate(y / *q8_scale) in e4m3 and
 * accumulates *q8_amax = max(*q8_amax, max |y|): the 8-bit operand of the next Linear's product, without a quantise pass over y.
 * y may be NULL (ABI 5): only y8 / mean / rstd are written -- for a step in which every consumer of y reads the 8-bit copy */
int ecgvit_layernorm_fwd_q8(const void *x, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
                            int64_t rows, int d, float eps, void *y8, const float *q8_scale, float *q8_amax, void *stream);
/* dx = (dres ? dres : 0) + LN'(dy) ; dgamma/dbeta are OVERWRITTEN with the full reduction over rows.
 * `partial` is caller workspace of ecgvit_layernorm_bwd_workspace(rows, d) bytes. */
int64_t ecgvit_layernorm_bwd_workspace(int64_t rows, int d);
int ecgvit_layernorm_bwd(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd,
                         const void *dres, void *dx, float *dgamma, float *dbeta, void *partial,
                         int64_t rows, int d, int dtype, void *stream);

/* Same, plus what the next backward stage wants from dx while it is still in registers: dxm = dx * dropout_mask(seed, i)
 * (only written when dropout_p > 0) and dcolsum[c] = sum_rows (dropout_p > 0 ? dxm : dx) -- the gradient of the bias of the
 * `dropout(Linear + bias) + residual` site that produced x. */
int ecgvit_layernorm_bwd_fused(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd,
                               const void *dres, void *dx, float *dgamma, float *dbeta, void *partial, int64_t rows, int d,
                               void *dxm, float *dcolsum, float dropout_p, uint64_t seed, int dtype, void *stream);

/* fp8 operand path: the same fused backward (bf16, d in 64 * {4, 8, 12, 16, 24, 32}) that also writes g8 = saturate(v / *q8_scale) in e5m2 for
 * v = the gradient the next stage consumes (dxm when dropout_p > 0, else dx; as stored) and accumulates *q8_amax = max(*q8_amax, max |v|):
 * the 8-bit A operand of that stage's input-gradient product, without a quantise pass over the gradient.
 * dxm may be NULL even with dropout_p > 0 (ABI 5): the masked gradient is then written as g8 only (dcolsum and g8 are unchanged) */
int ecgvit_layernorm_bwd_fused_q8(const void *dy, const void *x, const float *gamma, const float *mean, const float *rstd,
                                  const void *dres, void *dx, float *dgamma, float *dbeta, void *partial, int64_t rows, int d,
                                  void *dxm, float *dcolsum, float dropout_p, uint64_t seed, void *g8, const float *q8_scale,
                                  float *q8_amax, void *stream);

/* out[i] = in[i] * keep(seed, i) / (1-p): re-applies an epilogue dropout mask (element index = m*N+n, contiguous [M,N])
 * to the incoming gradient of a `dropout(acc + bias) + residual` site.  in == out allowed. */
int ecgvit_dropout_apply(const void *in, void *out, int64_t count, float dropout_p, uint64_t seed, int dtype, void *stream);

/* out[n] = sum_m in[m,n]  (bias gradients).  `partial`: ecgvit_colsum_workspace(M,N) bytes. */
int64_t ecgvit_colsum_workspace(int64_t M, int N);
int ecgvit_colsum(const void *in, int64_t ld, float *out, void *partial, int64_t M, int N, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-head self-attention core, fused (bf16 path).  replaces vit_pytorch Attention.forward between
 * to_qkv and to_out: split heads, dots = q k^T * dh^-0.5, softmax, (dropout), attn v, merge heads.
 * qkv: [B*N, 3*h*dh] (columns [q | k | v], head-major inside each) ; out: [B*N, h*dh] ; lse: [B,h,N] f32.
 * bf16 path requires dh == 64 and N <= 512 (online softmax over 32-key tiles; the backward takes 256 < N <= 512 as two key windows).
 * Probability dropout of the fused kernels: one 8-bit hash per 4 consecutive keys, so the probability APPLIED is
 * round(256 p) / 256 (p = 0.1 -> 26/256 = 0.1016), kept values rescaled by the exact 256 / (256 - round(256 p)); 0 < p < 1/512 cannot be
 * represented and is rejected (ECGVIT_EINVAL) rather than silently rounded to no dropout.
 * ------------------------------------------------------------------------------------------------ */
int ecgvit_attention_fwd(const void *qkv, void *out, float *lse, int B, int N, int h, int dh, float scale,
                         float dropout_p, uint64_t seed, int dtype, void *stream);
/* fp8 operand path: the same forward that also writes out8 = saturate(out as stored / *q8_scale) in e4m3 (same [B*N, h*dh] layout, one byte
 * per element, 16-byte aligned) and accumulates *q8_amax = max(*q8_amax, max |out|): the 8-bit operand of the out-projection's product, without a quantise pass */
int ecgvit_attention_fwd_q8(const void *qkv, void *out, float *lse, int B, int N, int h, int dh, float scale, float dropout_p,
                            uint64_t seed, void *out8, const float *q8_scale, float *q8_amax, void *stream);
int ecgvit_attention_bwd(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                         int B, int N, int h, int dh, float scale, float dropout_p, uint64_t seed, int dtype,
                         void *stream);
/* fp8 operand path (128 < N <= 512 only: ECGVIT_EINVAL otherwise, the caller then quantises dqkv itself): the same backward that also writes
 * dqkv8 = saturate(dqkv as stored / *q8_scale) in e5m2 (same [B*N, 3*h*dh] layout, one byte per element) and accumulates
 * *q8_amax = max(*q8_amax, max |dqkv|): the 8-bit operand of the QKV projection's two backward products, without a quantise pass */
int ecgvit_attention_bwd_q8(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                            int B, int N, int h, int dh, float scale, float dropout_p, uint64_t seed, void *dqkv8,
                            const float *q8_scale, float *q8_amax, void *stream);
/* export of the fused path's post-softmax probabilities (next row f3; what vit_pytorch's Recorder hooks, reference ecg_vit.py:176-180):
 * probs[B,h,N,N] f32 = exp(scale * q k^T - lse), from the qkv / lse a fused forward left behind. bf16 path only (the f32 path
 * materialises the scores anyway); visualisation-time, not tuned. B*h <= 65535. */
int ecgvit_attention_probs(const void *qkv, const float *lse, float *probs, int B, int N, int h, int dh, float scale, int dtype,
                           void *stream);
/* f32 parity path pieces (scores materialised; the GEMMs are ecgvit_gemm batched calls):
 * in-place row softmax of S[rows, ld] over the first N columns; optional export is the buffer itself. */
int ecgvit_softmax_rows(float *S, int64_t rows, int N, int64_t ld, void *stream);
/* dS = P * (dP - rowsum(P*dP)) * scale, written over dP */
int ecgvit_softmax_bwd_rows(const float *P, float *dP, int64_t rows, int N, int64_t ld, float scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * classification head + loss.  replaces: x[:,0] -> mlp_head (LayerNorm, Linear(d,K)) and
 * nn.BCEWithLogitsLoss (reference ecg_vit.py:118, :144-148).
 * ------------------------------------------------------------------------------------------------ */
/* logits[b,c] = LN(X[b*N+0]) . W[c,:] + bias[c] ; saves xhat [B,d] f32 and rstd [B] for backward */
int ecgvit_head_fwd(const void *X, int N, const float *gamma, const float *beta, const float *W, const float *bias,
                    float *logits, float *xhat, float *rstd, int B, int d, int K, float eps, int dtype, void *stream);
/* elementwise l = w * (max(z,0) - z*y + log1p(exp(-|z|))) ; loss_elem [B*K] always written;
 * loss_mean (1 f32) = mean(l) if non-NULL.  weight may be NULL.  Deterministic single-pass reduction. */
int ecgvit_bce_fwd(const float *logits, const float *labels, const float *weight, float *loss_elem, float *loss_mean,
                   int64_t count, void *stream);
/* dlogits = upstream * w * (sigmoid(z) - y) ; upstream = *gscalar * gscale (gelem NULL) or gelem[i] * gscale */
int ecgvit_bce_bwd(const float *logits, const float *labels, const float *weight, const float *gscalar,
                   const float *gelem, float gscale, float *dlogits, int64_t count, void *stream);
/* backward of head_fwd: dW,dbias,dgamma,dbeta overwritten; dX [B*N, d] is ZERO-FILLED then CLS rows written */
int ecgvit_head_bwd(const float *dlogits, const float *xhat, const float *rstd, const float *gamma, const float *beta,
                    const float *W, float *dW, float *dbias, float *dgamma, float *dbeta, void *dX, int N, int B, int d,
                    int K, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * optimiser: global-norm clip + AdamW/Adam over FLAT f32 buffers.  replaces train.py:281-282
 * (nn.utils.clip_grad_norm_(max_norm=1.0, error_if_nonfinite=True) + torch.optim.AdamW.step).
 * ------------------------------------------------------------------------------------------------ */
int64_t ecgvit_sumsq_workspace(int64_t count);
/* out[0] = sum(g^2) (f32, deterministic two-stage) ; `partial` = ecgvit_sumsq_workspace(count) bytes */
int ecgvit_sumsq(const float *g, int64_t count, float *out, void *partial, void *stream);
/* norm = |grad_scale| * sqrt(sumsq[0]) ; coef = min(1, max_norm / (norm + 1e-6)) (coef = 1 if max_norm <= 0);
 * g' = g * grad_scale * coef ; AdamW: p *= 1 - lr*wd ; Adam: g' += wd * p ; m,v update ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 * norm_out[0] = norm (pre-clip, of grad_scale-scaled grads), norm_out[1] = 1.0 if norm is finite else 0.0;
 * when the norm is non-finite NOTHING is updated (caller raises, as error_if_nonfinite=True does).
 * p_lowp: optional bf16 shadow copy of p, refreshed in the same pass. */
int ecgvit_adamw_step(float *p, const float *g, float *m, float *v, void *p_lowp, int64_t count,
                      const float *sumsq, float grad_scale, float max_norm, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, int decoupled, float *norm_out, void *stream);
/* g *= min(1, max_norm/(norm+1e-6)) in place (torch-optimizer interop path); norm_out as above */
int ecgvit_clip_scale(float *g, int64_t count, const float *sumsq, float max_norm, float *norm_out, void *stream);
int ecgvit_cast_f32_to_bf16(const float *src, void *dst, int64_t count, void *stream);
int ecgvit_cast_bf16_to_f32(const void *src, float *dst, int64_t count, void *stream);
/* Transposed bf16 shadows of the Linear weights: with W^T at hand the input-gradient product dX = dY . W (nn.Linear backward) runs
 * on the forward (A . B^T) kernel.  src / dst: flat bf16 buffers with identical layouts; table (DEVICE, int64 [nmat][4]) =
 * {element offset, rows, cols, index of the matrix's first 64x64 tile}; dst + offset receives the cols x rows transpose.
 * ntiles = total number of 64x64 tiles over all matrices. */
int ecgvit_transpose_bf16_batched(const void *src, void *dst, const int64_t *table, int nmat, int64_t ntiles, void *stream);

/* ------------------------------------------------------------------------------------------------
 * masked pre-train objective (build's own definition; absent from the reference)
 * ------------------------------------------------------------------------------------------------ */
/* tok[b*n + idx[b,k]] = mask_token for k < m (idx int32, distinct per record) then X = tok + pos[1..n] (no CLS row).
 * flag_ws: caller scratch of B*n bytes. */
int ecgvit_mask_embed_finish(const void *tok, const float *mask_token, const float *pos, const int32_t *idx, void *X,
                             void *flag_ws, int B, int n, int m, int d, int dtype, void *stream);
/* backward: dtok = dX on un-masked rows (0 elsewhere), dmasked = dX on masked rows (0 elsewhere; its column sum is the
 * mask-token gradient), dpos[1+p] = sum_b dX[b*n+p], dpos[0] = 0.  flag_ws as written by ecgvit_mask_embed_finish. */
int ecgvit_mask_embed_bwd(const void *dX, const void *flag_ws, void *dtok, void *dmasked, float *dpos, int B, int n, int d,
                          int dtype, void *stream);
/* gather rows: out[b*m + k] = in[b*n + idx[b,k]] */
int ecgvit_gather_rows(const void *in, const int32_t *idx, void *out, int B, int n, int m, int64_t width, int64_t ld_in,
                       int64_t ld_out, int dtype, void *stream);
/* scatter-add rows (distinct idx per record => plain stores into a zero-filled buffer) */
int ecgvit_scatter_rows(const void *in, const int32_t *idx, void *out, int B, int n, int m, int64_t width,
                        int64_t ld_in, int64_t ld_out, int dtype, void *stream);
/* L1 reconstruction loss: loss[0] = mean |pred - target| ; dpred = upstream * sign(pred - target) / count.
 * partial: >= 1024 floats of scratch (per-block partial sums of the deterministic two-stage reduction). */
int ecgvit_l1_loss_fwd_bwd(const void *pred, const void *target, float *loss, void *dpred, const float *gscalar, float *partial,
                           int64_t rows, int width, int64_t ld, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------------
 * evaluation metrics (next row f1).  replaces: get_accuracy (ecg_transformer/util/train.py:12-56), called on every train
 * step (models/train.py:289) and after each eval pass (models/train.py:371) -- there a D2H copy of the logits + sklearn.
 * scores / labels: (B, K) f32 with row pitches ld_*; scores are probabilities, or logits when from_logits != 0 (then the f32
 * sigmoid the reference applies first, models/train.py:369, is evaluated in the kernel).  counts: uint64 [4 + 2K], overwritten:
 *   [0..3] tp, tn, fp, fn of (prob >= 0.5) over all B*K;  [4 + c] positives of class c;
 *   [4 + K + c] sum over (positive i, negative j) of 2*[p_i > p_j] + [p_i == p_j]  (AUROC_c = that / (2 P_c N_c)); 0 unless with_auc.
 * ------------------------------------------------------------------------------------------------ */
int ecgvit_eval_counts(const float *scores, int64_t ld_scores, const float *labels, int64_t ld_labels, int64_t B, int K, int from_logits,
                       int with_auc, uint64_t *counts, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ECGVIT_HIP_H */
