// Masked pre-train objective pieces (SimMIM-style; the build's own definition -- the reference has no masked
// objective): mask-token substitution + positional add, row gather / scatter by host-generated
// int32 indices (bit-exact integer indexing), L1 reconstruction loss.  All HBM-bound, 16 B per lane.
#include "common.h"

namespace {

// X[b*n + p] = (masked(b,p) ? mask_token : tok[b*n+p]) + pos[1 + p]; `flag` [B*n] bytes marks masked patches
template <typename T>
__global__ __launch_bounds__(256) void mask_embed_kernel(const T *__restrict__ tok, const float *__restrict__ mask_token,
                                                         const float *__restrict__ pos, const uint8_t *__restrict__ flag,
                                                         T *__restrict__ X, int64_t rows, int n, int d) {
    constexpr int VN = Vec16<T>::N;
    const int dv = d / VN;
    const int64_t total = rows * dv;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / dv;
        const int c0 = (int)(i - row * dv) * VN;
        const int p = (int)(row % n);
        Vec16<T> o;
        if (flag[row]) {
#pragma unroll
            for (int k = 0; k < VN; ++k) o.set(k, mask_token[c0 + k] + pos[(int64_t)(1 + p) * d + c0 + k]);
        } else {
            const Vec16<T> v = ld16(tok + row * d + c0);
#pragma unroll
            for (int k = 0; k < VN; ++k) o.set(k, v.get(k) + pos[(int64_t)(1 + p) * d + c0 + k]);
        }
        st16(X + row * d + c0, o);
    }
}

// backward of mask_embed: dtok = dX where NOT masked (else 0); dmasked = dX where masked (else 0) -> its column sum is the
// mask-token gradient; dpos[1 + p] = sum_b dX[b*n + p]   (dpos row 0 = CLS slot, untouched by the masked trunk: zeroed)
// One workgroup per (patch p, group of 8 column vectors): its 256 threads are 8 column vectors x 32 batch lanes, so a wave instruction moves
// 8 records' 128-B segments and n * d / 64 workgroups share the three streams (round 5; the first form -- one THREAD per (p, column vector)
// walking all B records, 94 workgroups at base -- ran at a third of the HBM rate: 313 us for 0.6 GB).  The batch sum is folded over the 32
// lanes through LDS in a fixed order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void mask_embed_bwd_kernel(const T *__restrict__ dX, const uint8_t *__restrict__ flag,
                                                             T *__restrict__ dtok, T *__restrict__ dmasked,
                                                             float *__restrict__ dpos, int B, int n, int d) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float red[32][8][VN];
    const int dv = d / VN, ngrp = (dv + 7) / 8;
    const int p = blockIdx.x / ngrp, cg = blockIdx.x - p * ngrp;
    const int cv = threadIdx.x & 7, bl = threadIdx.x >> 3;
    const int c0 = (cg * 8 + cv) * VN;
    const bool live = cg * 8 + cv < dv;
    float acc[VN];
#pragma unroll
    for (int k = 0; k < VN; ++k) acc[k] = 0.f;
    Vec16<T> zero;
#pragma unroll
    for (int k = 0; k < VN; ++k) zero.set(k, 0.f);
    if (live) {
        for (int b = bl; b < B; b += 32) {
            const int64_t row = (int64_t)b * n + p;
            const Vec16<T> v = ld16(dX + row * d + c0);
#pragma unroll
            for (int k = 0; k < VN; ++k) acc[k] += v.get(k);
            const bool mk = flag[row] != 0;
            st16(dtok + row * d + c0, mk ? zero : v);
            st16(dmasked + row * d + c0, mk ? v : zero);
        }
    }
#pragma unroll
    for (int k = 0; k < VN; ++k) red[bl][cv][k] = acc[k];
    __syncthreads();
    if (bl == 0 && live) {
#pragma unroll
        for (int k = 0; k < VN; ++k) {
            float s = 0.f;
            for (int j = 0; j < 32; ++j) s += red[j][cv][k];
            dpos[(int64_t)(1 + p) * d + c0 + k] = s;
            if (p == 0) dpos[c0 + k] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void mark_mask_kernel(const int32_t *__restrict__ idx, uint8_t *__restrict__ flag, int B, int n, int m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * m) return;
    const int b = i / m;
    const int p = idx[i];
    if (p >= 0 && p < n) flag[(int64_t)b * n + p] = 1;
}

// GATHER: out[b*m + k] = in[b*n + idx[b,k]] ; SCATTER: out[b*n + idx[b,k]] = in[b*m + k]
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void rows_by_index_kernel(const T *__restrict__ in, const int32_t *__restrict__ idx,
                                                            T *__restrict__ out, int B, int n, int m, int width, int64_t ld_in,
                                                            int64_t ld_out) {
    constexpr int VN = Vec16<T>::N;
    const int wv = width / VN;
    const int64_t total = (int64_t)B * m * wv;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / wv;
        const int c0 = (int)(i - r * wv) * VN;
        const int64_t b = r / m;
        const int64_t src = b * n + idx[r];
        if (SCATTER) st16(out + src * ld_out + c0, ld16(in + r * ld_in + c0));
        else st16(out + r * ld_out + c0, ld16(in + src * ld_in + c0));
    }
}

// L1 loss + its gradient, two deterministic stages: L1_BLOCKS blocks each own a contiguous run of elements (8 per thread and pass,
// 16-B accesses when the row pitch allows) and leave one partial sum; a single wave then adds the partials in a fixed order.
constexpr int L1_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void l1_loss_kernel(const T *__restrict__ pred, const T *__restrict__ target,
                                                      float *__restrict__ partial, T *__restrict__ dpred,
                                                      const float *__restrict__ gscalar, int64_t rows, int width, int64_t ld) {
    __shared__ float red[4];
    const float up = (gscalar ? gscalar[0] : 1.0f) / (float)(rows * width);
    const int64_t total = rows * width;
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    float s = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int64_t r = i / width;
        const int c = (int)(i - r * width);
        const float df = to_f32<T>(pred[r * ld + c]) - to_f32<T>(target[r * ld + c]);
        s += fabsf(df);
        if (dpred) dpred[r * ld + c] = from_f32<T>(df > 0.f ? up : (df < 0.f ? -up : 0.f));
    }
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ __launch_bounds__(64) void l1_finish_kernel(const float *__restrict__ partial, int n, float *__restrict__ loss, float inv_total) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * inv_total;
}

inline int grid_ew(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

}  // namespace

extern "C" {

int ecgvit_mask_embed_finish(const void *tok, const float *mask_token, const float *pos, const int32_t *idx, void *X,
                                void *flag_ws, int B, int n, int m, int d, int dtype, void *stream) {
    if (B <= 0 || n <= 0 || m < 0 || m > n || d <= 0 || d % 8 != 0 || !flag_ws) return ECGVIT_EINVAL;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(flag_ws, 0, (size_t)B * n, s) != hipSuccess) return ECGVIT_ELAUNCH;
    if (m > 0) {
        hipLaunchKernelGGL(mark_mask_kernel, dim3((B * m + 255) / 256), dim3(256), 0, s, idx, (uint8_t *)flag_ws, B, n, m);
        ECGVIT_CHECK_LAUNCH();
    }
    const int64_t rows = (int64_t)B * n;
    if (dtype == ECGVIT_F32)
        hipLaunchKernelGGL(mask_embed_kernel<float>, dim3(grid_ew(rows * d / 4)), dim3(256), 0, s, (const float *)tok, mask_token, pos, (const uint8_t *)flag_ws, (float *)X, rows, n, d);
    else if (dtype == ECGVIT_BF16)
        hipLaunchKernelGGL(mask_embed_kernel<bf16_t>, dim3(grid_ew(rows * d / 8)), dim3(256), 0, s, (const bf16_t *)tok, mask_token, pos, (const uint8_t *)flag_ws, (bf16_t *)X, rows, n, d);
    else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_mask_embed_bwd(const void *dX, const void *flag_ws, void *dtok, void *dmasked, float *dpos, int B, int n, int d,
                          int dtype, void *stream) {
    if (B <= 0 || n <= 0 || d <= 0 || d % 8 != 0 || !flag_ws) return ECGVIT_EINVAL;
    const int dv = d / (dtype == ECGVIT_F32 ? 4 : 8);
    const int grid = n * ((dv + 7) / 8);   // one workgroup per (patch, 8 column vectors)
    if (dtype == ECGVIT_F32)
        hipLaunchKernelGGL(mask_embed_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float *)dX, (const uint8_t *)flag_ws, (float *)dtok, (float *)dmasked, dpos, B, n, d);
    else if (dtype == ECGVIT_BF16)
        hipLaunchKernelGGL(mask_embed_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t *)dX, (const uint8_t *)flag_ws, (bf16_t *)dtok, (bf16_t *)dmasked, dpos, B, n, d);
    else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

static int rows_by_index(const void *in, const int32_t *idx, void *out, int B, int n, int m, int64_t width, int64_t ld_in,
                         int64_t ld_out, int dtype, void *stream, bool scatter) {
    if (B <= 0 || n <= 0 || m <= 0 || width <= 0 || width % 8 != 0 || ld_in % 8 != 0 || ld_out % 8 != 0) return ECGVIT_EINVAL;
    hipStream_t s = as_stream(stream);
    const int vn = dtype == ECGVIT_F32 ? 4 : 8;
    const int g = grid_ew((int64_t)B * m * width / vn);
    if (dtype == ECGVIT_F32) {
        if (scatter) hipLaunchKernelGGL((rows_by_index_kernel<float, true>), dim3(g), dim3(256), 0, s, (const float *)in, idx, (float *)out, B, n, m, (int)width, ld_in, ld_out);
        else hipLaunchKernelGGL((rows_by_index_kernel<float, false>), dim3(g), dim3(256), 0, s, (const float *)in, idx, (float *)out, B, n, m, (int)width, ld_in, ld_out);
    } else if (dtype == ECGVIT_BF16) {
        if (scatter) hipLaunchKernelGGL((rows_by_index_kernel<bf16_t, true>), dim3(g), dim3(256), 0, s, (const bf16_t *)in, idx, (bf16_t *)out, B, n, m, (int)width, ld_in, ld_out);
        else hipLaunchKernelGGL((rows_by_index_kernel<bf16_t, false>), dim3(g), dim3(256), 0, s, (const bf16_t *)in, idx, (bf16_t *)out, B, n, m, (int)width, ld_in, ld_out);
    } else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_gather_rows(const void *in, const int32_t *idx, void *out, int B, int n, int m, int64_t width, int64_t ld_in,
                       int64_t ld_out, int dtype, void *stream) {
    return rows_by_index(in, idx, out, B, n, m, width, ld_in, ld_out, dtype, stream, false);
}

int ecgvit_scatter_rows(const void *in, const int32_t *idx, void *out, int B, int n, int m, int64_t width, int64_t ld_in,
                        int64_t ld_out, int dtype, void *stream) {
    return rows_by_index(in, idx, out, B, n, m, width, ld_in, ld_out, dtype, stream, true);
}

int ecgvit_l1_loss_fwd_bwd(const void *pred, const void *target, float *loss, void *dpred, const float *gscalar, float *partial,
                           int64_t rows, int width, int64_t ld, int dtype, void *stream) {
    if (rows <= 0 || width <= 0 || ld < width || !partial) return ECGVIT_EINVAL;
    const int64_t total = rows * width;
    const int nb = (int)std::min<int64_t>(L1_BLOCKS, (total + 2047) / 2048);
    if (dtype == ECGVIT_F32)
        hipLaunchKernelGGL(l1_loss_kernel<float>, dim3(nb), dim3(256), 0, as_stream(stream), (const float *)pred, (const float *)target, partial, (float *)dpred, gscalar, rows, width, ld);
    else if (dtype == ECGVIT_BF16)
        hipLaunchKernelGGL(l1_loss_kernel<bf16_t>, dim3(nb), dim3(256), 0, as_stream(stream), (const bf16_t *)pred, (const bf16_t *)target, partial, (bf16_t *)dpred, gscalar, rows, width, ld);
    else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, as_stream(stream), partial, nb, loss, 1.0f / (float)total);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

}  // extern "C"
