// Exact-f32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32): the PARITY path.
//
// Bit-for-bit a k-ordered fmaf chain per output element (no reduced-precision fast path exists on
// gfx950), so this path reproduces the reference's fp32 CPU arithmetic to accumulation-order noise.
// Generic in every dimension (any M, N, K, leading dimension, 2-level batch strides) because it also
// carries the batched QK^T / PV / backward products of the f32 attention path, where N = 41 / 251 / 501.
//
// Tile 128x128x16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 accumulators.
// LDS images are k-major ([k][m]), so a fragment read is 32 consecutive floats per half-wave
// (conflict-free ds_read_b32); operands that are K-contiguous in memory are transposed on the LDS write.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;  // +4 floats: keeps 16-B alignment for b128 writes

// Stage one operand tile into registers. KCONTIG: memory is [mn][k] (k fastest); else [k][mn] (mn fastest).
template <bool KCONTIG>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int64_t ld, int mn0, int k0, int MN, int K,
                                          bool vec_ok, f32x4 (&r)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        int mn, k;
        if (KCONTIG) { mn = mn0 + (idx >> 2); k = k0 + (idx & 3) * 4; }
        else         { k = k0 + (idx >> 5);  mn = mn0 + (idx & 31) * 4; }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KCONTIG) {
            if (mn < MN) {
                const float *p = P + (int64_t)mn * ld + k;
                if (vec_ok && k + 3 < K) v = *reinterpret_cast<const f32x4 *>(p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = p[j];
                }
            }
        } else {
            if (k < K) {
                const float *p = P + (int64_t)k * ld + mn;
                if (vec_ok && mn + 3 < MN) v = *reinterpret_cast<const f32x4 *>(p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (mn + j < MN) v[j] = p[j];
                }
            }
        }
        r[i] = v;
    }
}

template <bool KCONTIG> __device__ __forceinline__ void store_tile(float (*S)[LDT], const f32x4 (&r)[2]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = t + 256 * i;
        if (KCONTIG) {
            const int mn = idx >> 2, k = (idx & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[k + j][mn] = r[i][j];
        } else {
            const int k = idx >> 5, mn = (idx & 31) * 4;
            *reinterpret_cast<f32x4 *>(&S[k][mn]) = r[i];
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(ecgvit_gemm_desc d, EpiParams e, bool vecA, bool vecB) {
    __shared__ __attribute__((aligned(16))) float As[BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];

    const int z = blockIdx.z, z1 = z / d.batch2, z2 = z % d.batch2;
    const float *A = reinterpret_cast<const float *>(d.A) + z1 * d.strideA1 + z2 * d.strideA2;
    const float *B = reinterpret_cast<const float *>(d.B) + z1 * d.strideB1 + z2 * d.strideB2;
    float *C = reinterpret_cast<float *>(d.C) + z1 * d.strideC1 + z2 * d.strideC2;
    const int M = d.M, N = d.N, K = d.K;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[2], rb[2];
    load_tile<A_KC>(A, d.lda, m0, 0, M, K, vecA, ra);
    load_tile<B_KC>(B, d.ldb, n0, 0, N, K, vecB, rb);
    const int nk = (K + BK - 1) / BK;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous tile's reads done
        store_tile<A_KC>(As, ra);
        store_tile<B_KC>(Bs, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            load_tile<A_KC>(A, d.lda, m0, (kt + 1) * BK, M, K, vecA, ra);
            load_tile<B_KC>(B, d.ldb, n0, (kt + 1) * BK, N, K, vecB, rb);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[2 * ks + lh][wm * 64 + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[2 * ks + lh][wn * 64 + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue straight from the accumulator layout: col = lane&31 (coalesced 128-B rows), row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M && n < N) {
                    // residual / aux are not batched (batched calls never use them)
                    float v = epilogue_value<float>(acc[i][j][r], (int64_t)m, n, e);
                    float *c = C + (int64_t)m * d.ldc + n;
                    if (e.flags & ECGVIT_EPI_ACCUM) v += *c;
                    *c = v;
                }
            }
        }
}

inline bool vec_ok(const void *p, int64_t ld, int64_t s1, int64_t s2) {
    return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0) && (s1 % 4 == 0) && (s2 % 4 == 0);
}

}  // namespace

int ecgvit_gemm_f32_launch(const ecgvit_gemm_desc *d, hipStream_t s, int *route) {
    if (d->dtype != ECGVIT_F32 || d->out_dtype != ECGVIT_F32) return ECGVIT_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K < 0 || d->batch1 < 1 || d->batch2 < 1) return ECGVIT_EINVAL;
    const int64_t nz = (int64_t)d->batch1 * d->batch2;
    if (nz > 65535) return ECGVIT_EINVAL;
    if (nz > 1 && (d->epilogue & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_RESIDUAL | ECGVIT_EPI_DROPOUT)))
        return ECGVIT_EINVAL;
    if (route) {
        if (d->layout != ECGVIT_GEMM_NT && d->layout != ECGVIT_GEMM_NN && d->layout != ECGVIT_GEMM_TN) return ECGVIT_EINVAL;
        *route = ECGVIT_KERNEL_GEMM_F32;
        return ECGVIT_OK;
    }
    dim3 grid((d->N + BN - 1) / BN, (d->M + BM - 1) / BM, (unsigned)nz), block(256);
    EpiParams e = make_epi(d);
    const bool va = vec_ok(d->A, d->lda, d->strideA1, d->strideA2), vb = vec_ok(d->B, d->ldb, d->strideB1, d->strideB2);
    switch (d->layout) {
        case ECGVIT_GEMM_NT: hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, *d, e, va, vb); break;
        case ECGVIT_GEMM_NN: hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, *d, e, va, vb); break;
        case ECGVIT_GEMM_TN: hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, *d, e, va, vb); break;
        default: return ECGVIT_EINVAL;
    }
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}
