// Evaluation metrics on the device: the integer statistics behind the reference's `get_accuracy`
// (ecg_transformer/util/train.py:12-56: sklearn accuracy / balanced accuracy / classification_report recalls / per-class
// roc_auc_score), so a train or eval step never ships (B, 71) logits to the host and never calls sklearn.
// Everything the device produces is an exact integer count; the handful of divisions happen on the host in double.
#include "common.h"

// counts[0..3] = tp, tn, fp, fn over all B*K (prediction = prob >= 0.5, util/train.py:23); counts[4 + c] = positives of class c
__device__ __forceinline__ float as_prob(float s, int from_logits) { return from_logits ? 1.0f / (1.0f + expf(-s)) : s; }

__global__ __launch_bounds__(256) void eval_confusion_kernel(const float *__restrict__ scores, int64_t lds_, const float *__restrict__ labels,
                                                             int64_t ldl, int64_t B, int K, int from_logits,
                                                             unsigned long long *__restrict__ counts) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float tp = 0, tn = 0, fp = 0, fn = 0;   // < 2^24 per thread for B < 4e9: exact in f32
    for (int64_t i = threadIdx.x; i < B; i += 256) {
        const bool y = labels[i * ldl + c] != 0.f;
        const bool p = as_prob(scores[i * lds_ + c], from_logits) >= 0.5f;
        tp += (p && y), tn += (!p && !y), fp += (p && !y), fn += (!p && y);
    }
    tp = block_sum<4>(tp, red), tn = block_sum<4>(tn, red), fp = block_sum<4>(fp, red), fn = block_sum<4>(fn, red);
    if (threadIdx.x == 0) {
        atomicAdd(counts + 0, (unsigned long long)tp), atomicAdd(counts + 1, (unsigned long long)tn);
        atomicAdd(counts + 2, (unsigned long long)fp), atomicAdd(counts + 3, (unsigned long long)fn);
        counts[4 + c] = (unsigned long long)(tp + fn);
    }
}

// counts[4 + K + c] = sum over (positive i, negative j) of 2*[p_i > p_j] + [p_i == p_j]   (= 2 * P_c * N_c * AUROC_c:
// the Mann-Whitney form of the trapezoidal ROC area sklearn's roc_auc_score integrates, ties worth one half)
#define PAIR_TILE 2048
__global__ __launch_bounds__(256) void eval_pairs_kernel(const float *__restrict__ scores, int64_t lds_, const float *__restrict__ labels,
                                                         int64_t ldl, int64_t B, int K, int from_logits,
                                                         unsigned long long *__restrict__ counts) {
    __shared__ __attribute__((aligned(16))) float q[PAIR_TILE];
    __shared__ int any_pos;
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) any_pos = 0;
    __syncthreads();
    float p = 0.f;
    bool pos = false;
    if (i < B) {
        pos = labels[i * ldl + c] != 0.f;
        p = as_prob(scores[i * lds_ + c], from_logits);
    }
    if (pos) any_pos = 1;
    __syncthreads();
    if (!any_pos) return;   // rare codes: most 256-record tiles hold no positive at all
    if (!pos) p = __builtin_nanf("");   // compares false against everything
    unsigned gt = 0, eq = 0;
    for (int64_t j0 = 0; j0 < B; j0 += PAIR_TILE) {
        __syncthreads();
        for (int t = threadIdx.x; t < PAIR_TILE; t += 256) {
            const int64_t j = j0 + t;
            float v = __builtin_nanf("");   // positives and the ragged tail never count
            if (j < B && labels[j * ldl + c] == 0.f) v = as_prob(scores[j * lds_ + c], from_logits);
            q[t] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < PAIR_TILE; t += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(q + t);   // wave-uniform address: LDS broadcast
            gt += (p > v[0]) + (p > v[1]) + (p > v[2]) + (p > v[3]);
            eq += (p == v[0]) + (p == v[1]) + (p == v[2]) + (p == v[3]);
        }
    }
    unsigned long long s = 2ull * gt + eq;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(counts + 4 + K + c, s);
}

extern "C" int ecgvit_eval_counts(const float *scores, int64_t ld_scores, const float *labels, int64_t ld_labels, int64_t B, int K, int from_logits,
                                  int with_auc, uint64_t *counts, void *stream) {
    if (B <= 0 || K <= 0 || B >= (1ll << 31) || !scores || !labels || !counts) return ECGVIT_EINVAL;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(counts, 0, sizeof(uint64_t) * (4 + 2 * (size_t)K), s) != hipSuccess) return ECGVIT_ELAUNCH;
    eval_confusion_kernel<<<K, 256, 0, s>>>(scores, ld_scores, labels, ld_labels, B, K, from_logits, (unsigned long long *)counts);
    ECGVIT_CHECK_LAUNCH();
    if (with_auc) {
        eval_pairs_kernel<<<dim3((unsigned)((B + 255) / 256), K), 256, 0, s>>>(scores, ld_scores, labels, ld_labels, B, K, from_logits,
                                                                              (unsigned long long *)counts);
        ECGVIT_CHECK_LAUNCH();
    }
    return ECGVIT_OK;
}
