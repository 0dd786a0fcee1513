/* Entry points of the TOOLS library only (ecg-representation-learning_amd/csrc/build/libecgvit_hip_tools.so = the product objects plus the
 * -DECGVIT_TOOLS builds of the kernel files; `make -C ecg-representation-learning_amd/csrc tools`).  Nothing here is part of the product
 * C-ABI (include/ecgvit_hip.h) or of the shipped library: probes that pin hardware fragment layouts for the tests, a second independent
 * implementation of the attention backward, cycle stamps and A/B switches of tools/*.py. */
#ifndef ECGVIT_HIP_TOOLS_H
#define ECGVIT_HIP_TOOLS_H
#include "../include/ecgvit_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* exact-integer dump of what each lane receives from the LDS fragment helpers and of the MFMA C layout (tests/test_gpu_ops.py) */
int ecgvit_probe_mfma_layout(float *out /* [4][64][16] */, void *stream);

/* The attention backward on the one-(record, head)-per-workgroup kernel (N <= 256; what ecgvit_attention_bwd itself runs for N <= 128):
 * an independent implementation of the same function, so that tests can hold the persistent kernel against it. */
int ecgvit_attention_bwd_oneitem(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv,
                                 int B, int N, int h, int dh, float scale, float dropout_p, uint64_t seed, int dtype,
                                 void *stream);

/* -1 (default): the shipped eight-wave staggered persistent kernel; 0 .. 5: its lockstep / priority variants (0 / 1: lockstep, 2: what
 * round 3 shipped, 3 / 4 / 5: priority variants).  tools/attn_variants.py, tools/attn_ab.py.  (-2 selected round 4's four-wave experiment
 * while it was part of the tools build: tools/experiments/.) */
int ecgvit_tools_attn_variant(int v);
/* forward: -1 (default) the product's dispatch; 0: always the one-item-per-workgroup forward; 1: always the streamed forward (round 6) */
int ecgvit_tools_attn_fwd_variant(int v);
/* device buffer of 768 x 128 (+ per-phase records) uint64 that the eight-wave persistent backward fills with cycle stamps; NULL = off */
int ecgvit_debug_attn_stamps(void *buf);

/* GEMM A/B and stamps (tools/gemm_ab.py, tools/nt_stamps.py, tools/contention.py, tools/wgrad_ab.py) */
int ecgvit_tools_gemm(const ecgvit_gemm_desc *d, void *stream, int kernel, int raster_g, int diag);
/* LayerNorm-fold pricing (round 6; diag bit 1024 of ecgvit_tools_gemm, kernel 2): the QKV-forward (epilogue BIAS, non-temporal stores) and FFN-up-forward
 * bodies with v = v a[m] + (b[m] g[n] + bias[n]) in front of their epilogue; a, b: f32 [M]; g: f32 [N] */
int ecgvit_tools_rowaffine(const float *row_a, const float *row_b, const float *col_g);
int ecgvit_tools_occupy(int n_cus, unsigned long long cycles, unsigned int *done, void *stream);
int ecgvit_tools_nt_stamps(unsigned long long *h_out);
void ecgvit_tools_wgrad_body(int eight_wave);

#ifdef __cplusplus
}
#endif
#endif
