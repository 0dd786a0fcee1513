// Classification head (CLS row -> LayerNorm -> Linear(d,K)), BCE-with-logits, and the fused
// global-norm clip + AdamW step over flat f32 buffers.  Small or HBM-bound; all f32 except the
// activation tensor X / dX (template T) and the optional bf16 shadow weights.
#include "common.h"

namespace {

// ---- head forward: one block per record ---------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T *__restrict__ X, int N, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ W,
                                                       const float *__restrict__ bias, float *__restrict__ logits,
                                                       float *__restrict__ xhat, float *__restrict__ rstd_out, int d, int K, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [d] normalised row + [4] reduction scratch
    float *xn = sm, *red = sm + d;
    const int b = blockIdx.x;
    const T *xr = X + (int64_t)b * N * d;  // CLS row = token 0 of record b
    float s = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) { const float v = to_f32<T>(xr[c]); xn[c] = v; s += v; }
    const float mu = block_sum<4>(s, red) / (float)d;
    float q = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) { const float t = xn[c] - mu; q += t * t; }
    const float rs = 1.0f / sqrtf(block_sum<4>(q, red) / (float)d + eps);
    for (int c = threadIdx.x; c < d; c += 256) {
        const float h = (xn[c] - mu) * rs;
        xhat[(int64_t)b * d + c] = h;
        xn[c] = h * gamma[c] + beta[c];
    }
    if (threadIdx.x == 0) rstd_out[b] = rs;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = w; k < K; k += 4) {
        const float *wr = W + (int64_t)k * d;
        float acc = 0.f;
        for (int c = lane; c < d; c += 64) acc += xn[c] * wr[c];
        acc = wave_sum(acc);
        if (lane == 0) logits[(int64_t)b * K + k] = acc + bias[k];
    }
}

// ---- BCE with logits ----------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bce_fwd_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                       const float *__restrict__ w, float *__restrict__ le,
                                                       float *__restrict__ lmean, int64_t count) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += 1024) {
        const float zi = z[i];
        float l = fmaxf(zi, 0.f) - zi * y[i] + log1pf(expf(-fabsf(zi)));
        if (w) l *= w[i];
        le[i] = l;
        s += l;
    }
    if (lmean) {
        const float tot = block_sum<16>(s, red);
        if (threadIdx.x == 0) lmean[0] = tot / (float)count;
    }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                      const float *__restrict__ w, const float *__restrict__ gscalar,
                                                      const float *__restrict__ gelem, float gscale, float *__restrict__ dz,
                                                      int64_t count) {
    const int64_t i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= count) return;
    const float up = (gelem ? gelem[i] : (gscalar ? gscalar[0] : 1.0f)) * gscale;
    const float sg = 1.0f / (1.0f + expf(-z[i]));
    float g = (sg - y[i]) * up;
    if (w) g *= w[i];
    dz[i] = g;
}

// ---- head backward ------------------------------------------------------------------------------
// With T[k][c] = sum_b dl[b][k] * xhat[b][c] and dbias[k] = sum_b dl[b][k]:
//   dW[k][c] = gamma[c] * T[k][c] + beta[c] * dbias[k]
//   dgamma[c] = sum_k W[k][c] * T[k][c]          (= sum_b dxn[b][c] * xhat[b][c],  dxn = dl . W)
//   dbeta[c]  = sum_k W[k][c] * dbias[k]         (= sum_b dxn[b][c])
// stage 1 (grid = K blocks) leaves T in the dW buffer; stage 2 (one thread per column) finishes all four.
__global__ __launch_bounds__(256) void head_bwd_t_kernel(const float *__restrict__ dl, const float *__restrict__ xhat,
                                                         float *__restrict__ T, float *__restrict__ dbias, int B, int d, int K) {
    // block = (class k, group of 64 columns); 64 columns x 4 batch slices, fixed-order LDS combine (deterministic).
    // (one block per class with a 512-deep sequential loop per thread took 159 us for 28 MFLOP)
    __shared__ float red[4][64];
    __shared__ float redb[4][64];
    const int groups = (d + 63) / 64;
    const int k = blockIdx.x / groups, c = (blockIdx.x - k * groups) * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, sb = 0.f;
    if (c < d) {
        int b = sl;
        for (; b + 4 < B; b += 8) {
            a0 += dl[(int64_t)b * K + k] * xhat[(int64_t)b * d + c];
            a1 += dl[(int64_t)(b + 4) * K + k] * xhat[(int64_t)(b + 4) * d + c];
        }
        for (; b < B; b += 4) a0 += dl[(int64_t)b * K + k] * xhat[(int64_t)b * d + c];
    }
    if (blockIdx.x == k * groups)   // the class's first column group also sums its logit gradients: thread t takes rows t, t+256, ...
        for (int b = threadIdx.x; b < B; b += 256) sb += dl[(int64_t)b * K + k];
    red[sl][threadIdx.x & 63] = a0 + a1;
    redb[sl][threadIdx.x & 63] = sb;
    __syncthreads();
    if (sl == 0) {
        if (c < d) T[(int64_t)k * d + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (blockIdx.x == k * groups) {
            float t = (redb[0][threadIdx.x] + redb[1][threadIdx.x]) + (redb[2][threadIdx.x] + redb[3][threadIdx.x]);
            t = wave_sum(t);
            if (threadIdx.x == 0) dbias[k] = t;
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_finish_kernel(const float *__restrict__ W, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, const float *__restrict__ dbias,
                                                              float *__restrict__ dW, float *__restrict__ dgamma,
                                                              float *__restrict__ dbeta, int d, int K) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const float g = gamma[c], bt = beta[c];
    float ag = 0.f, ab = 0.f;
    for (int k = 0; k < K; ++k) {
        const float w = W[(int64_t)k * d + c], t = dW[(int64_t)k * d + c], db = dbias[k];
        ag += w * t;
        ab += w * db;
        dW[(int64_t)k * d + c] = g * t + bt * db;
    }
    dgamma[c] = ag;
    dbeta[c] = ab;
}

// per record: dxn = dl[b] . W ; LayerNorm backward on the CLS row ; writes dX[b*N+0]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_x_kernel(const float *__restrict__ dl, const float *__restrict__ xhat,
                                                         const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                         const float *__restrict__ W, T *__restrict__ dX, int N, int d, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [K] dl row + [d] g + [4]
    float *dlr = sm, *g = sm + K, *red = sm + K + d;
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += 256) dlr[k] = dl[(int64_t)b * K + k];
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < d; c += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += dlr[k] * W[(int64_t)k * d + c];
        const float gv = acc * gamma[c];
        g[c] = gv;
        s1 += gv;
        s2 += gv * xhat[(int64_t)b * d + c];
    }
    const float c1 = block_sum<4>(s1, red) / (float)d;
    const float c2 = block_sum<4>(s2, red) / (float)d;
    const float rs = rstd[b];
    T *o = dX + (int64_t)b * N * d;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = from_f32<T>(rs * (g[c] - c1 - xhat[(int64_t)b * d + c] * c2));
}

// ---- optimiser ----------------------------------------------------------------------------------
constexpr int SUMSQ_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ g, int64_t count, float *__restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t nv = count / 4;
    const f32x4 *gv = reinterpret_cast<const f32x4 *>(g);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = gv[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0) for (int64_t i = nv * 4 + threadIdx.x; i < count; i += 256) s += g[i] * g[i];
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float *__restrict__ partial, int n, float *__restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    const float tot = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

__device__ __forceinline__ float clip_coef(const float *sumsq, float grad_scale, float max_norm, float *norm, bool *finite) {
    const float nrm = sqrtf(sumsq[0]) * fabsf(grad_scale);
    *norm = nrm;
    *finite = isfinite(nrm);
    if (max_norm <= 0.f) return 1.0f;
    return fminf(1.0f, max_norm / (nrm + 1e-6f));
}

__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, bf16_t *__restrict__ plow, int64_t count,
                                                    const float *__restrict__ sumsq, float grad_scale, float max_norm, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    int decoupled, float *__restrict__ norm_out) {
    float nrm;
    bool fin;
    const float coef = clip_coef(sumsq, grad_scale, max_norm, &nrm, &fin) * grad_scale;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) { norm_out[0] = nrm; norm_out[1] = fin ? 1.0f : 0.0f; }
    if (!fin) return;
    const float step_size = lr / bc1;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float pi = p[i], gi = g[i] * coef, mi = m[i], vi = v[i];
        if (decoupled) pi *= 1.0f - lr * wd;
        else gi += wd * pi;
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (plow) plow[i] = (bf16_t)pi;
    }
}

__global__ __launch_bounds__(256) void clip_scale_kernel(float *__restrict__ g, int64_t count, const float *__restrict__ sumsq,
                                                         float max_norm, float *__restrict__ norm_out) {
    float nrm;
    bool fin;
    const float coef = clip_coef(sumsq, 1.0f, max_norm, &nrm, &fin);
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) { norm_out[0] = nrm; norm_out[1] = fin ? 1.0f : 0.0f; }
    if (!fin) return;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) g[i] *= coef;
}

// ---- batched transpose of bf16 matrices that live at the SAME offsets in two flat buffers (shadow weights -> transposed shadows):
//      table[4i..4i+3] = {element offset, rows, cols, index of the matrix's first 64x64 tile}; dst holds cols x rows row-major.
//      One 64 x 64 tile per 256-thread block through LDS; 16-B accesses on both sides when the tile is interior.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t *__restrict__ src, bf16_t *__restrict__ dst,
                                                                const int64_t *__restrict__ table, int nmat) {
    __shared__ bf16_t tile[64][72];   // 144-B pitch: 16-B aligned rows, column reads spread over banks
    int lo = 0, hi = nmat - 1;        // last matrix whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[4 * mid + 3] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t off = table[4 * lo];
    const int rows = (int)table[4 * lo + 1], cols = (int)table[4 * lo + 2];
    const int t = (int)((int64_t)blockIdx.x - table[4 * lo + 3]);
    const int tc = (cols + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const bf16_t *S = src + off;
    bf16_t *D = dst + off;
    const bool interior = r0 + 64 <= rows && c0 + 64 <= cols && (cols % 8) == 0 && (rows % 8) == 0 && (off % 8) == 0;
    if (interior) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = p * 32 + (threadIdx.x >> 3), c = (threadIdx.x & 7) * 8;
            *reinterpret_cast<bf16x8 *>(&tile[r][c]) = *reinterpret_cast<const bf16x8 *>(S + (int64_t)(r0 + r) * cols + c0 + c);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = p * 32 + (threadIdx.x >> 3), r = (threadIdx.x & 7) * 8;   // dst row = source column c, 8 source rows r..r+7
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[r + k][c];
            *reinterpret_cast<bf16x8 *>(D + (int64_t)(c0 + c) * rows + r0 + r) = v;
        }
    } else {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            if (r0 + r < rows && c0 + c < cols) tile[r][c] = S[(int64_t)(r0 + r) * cols + c0 + c];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int c = i >> 6, r = i & 63;
            if (r0 + r < rows && c0 + c < cols) D[(int64_t)(c0 + c) * rows + r0 + r] = tile[r][c];
        }
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float *__restrict__ s, bf16_t *__restrict__ d, int64_t count) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) d[i] = (bf16_t)s[i];
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t *__restrict__ s, float *__restrict__ d, int64_t count) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) d[i] = (float)s[i];
}

inline int ew_grid(int64_t count) { return (int)std::max<int64_t>(1, std::min<int64_t>((count + 255) / 256, 4096)); }

}  // namespace

extern "C" {

int ecgvit_head_fwd(const void *X, int N, const float *gamma, const float *beta, const float *W, const float *bias, float *logits,
                    float *xhat, float *rstd, int B, int d, int K, float eps, int dtype, void *stream) {
    if (B <= 0 || d <= 0 || K <= 0 || N <= 0 || d > 8192) return ECGVIT_EINVAL;
    const size_t lds = (size_t)(d + 4) * 4;
    if (dtype == ECGVIT_F32)
        hipLaunchKernelGGL(head_fwd_kernel<float>, dim3(B), dim3(256), lds, as_stream(stream), (const float *)X, N, gamma, beta, W, bias, logits, xhat, rstd, d, K, eps);
    else if (dtype == ECGVIT_BF16)
        hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, dim3(B), dim3(256), lds, as_stream(stream), (const bf16_t *)X, N, gamma, beta, W, bias, logits, xhat, rstd, d, K, eps);
    else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_bce_fwd(const float *logits, const float *labels, const float *weight, float *loss_elem, float *loss_mean, int64_t count,
                   void *stream) {
    if (count <= 0 || !loss_elem) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, labels, weight, loss_elem, loss_mean, count);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_bce_bwd(const float *logits, const float *labels, const float *weight, const float *gscalar, const float *gelem,
                   float gscale, float *dlogits, int64_t count, void *stream) {
    if (count <= 0) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, as_stream(stream), logits, labels, weight, gscalar, gelem, gscale, dlogits, count);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_head_bwd(const float *dlogits, const float *xhat, const float *rstd, const float *gamma, const float *beta,
                    const float *W, float *dW, float *dbias, float *dgamma, float *dbeta, void *dX, int N, int B, int d, int K,
                    int dtype, void *stream) {
    if (B <= 0 || d <= 0 || K <= 0 || N <= 0 || d > 8192) return ECGVIT_EINVAL;
    if (dtype != ECGVIT_F32 && dtype != ECGVIT_BF16) return ECGVIT_EINVAL;
    hipStream_t s = as_stream(stream);
    const size_t esz = dtype == ECGVIT_F32 ? 4 : 2;
    if (hipMemsetAsync(dX, 0, (size_t)B * N * d * esz, s) != hipSuccess) return ECGVIT_ELAUNCH;
    hipLaunchKernelGGL(head_bwd_t_kernel, dim3(K * ((d + 63) / 64)), dim3(256), 0, s, dlogits, xhat, dW, dbias, B, d, K);
    ECGVIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bwd_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, s, W, gamma, beta, dbias, dW, dgamma, dbeta, d, K);
    ECGVIT_CHECK_LAUNCH();
    const size_t lds = (size_t)(K + d + 4) * 4;
    if (dtype == ECGVIT_F32)
        hipLaunchKernelGGL(head_bwd_x_kernel<float>, dim3(B), dim3(256), lds, s, dlogits, xhat, rstd, gamma, W, (float *)dX, N, d, K);
    else
        hipLaunchKernelGGL(head_bwd_x_kernel<bf16_t>, dim3(B), dim3(256), lds, s, dlogits, xhat, rstd, gamma, W, (bf16_t *)dX, N, d, K);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int64_t ecgvit_sumsq_workspace(int64_t count) { (void)count; return (int64_t)SUMSQ_BLOCKS * 4; }

int ecgvit_sumsq(const float *g, int64_t count, float *out, void *partial, void *stream) {
    if (count <= 0 || !partial || (reinterpret_cast<uintptr_t>(g) % 16) != 0) return ECGVIT_EINVAL;
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((count / 4 + 255) / 256, SUMSQ_BLOCKS));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), g, count, (float *)partial);
    ECGVIT_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float *)partial, nb, out);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_adamw_step(float *p, const float *g, float *m, float *v, void *p_lowp, int64_t count, const float *sumsq,
                      float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      int decoupled, float *norm_out, void *stream) {
    if (count <= 0 || step < 1 || !sumsq) return ECGVIT_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(count)), dim3(256), 0, as_stream(stream), p, g, m, v, (bf16_t *)p_lowp, count, sumsq,
                       grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), decoupled, norm_out);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_clip_scale(float *g, int64_t count, const float *sumsq, float max_norm, float *norm_out, void *stream) {
    if (count <= 0 || !sumsq) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(ew_grid(count)), dim3(256), 0, as_stream(stream), g, count, sumsq, max_norm, norm_out);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_transpose_bf16_batched(const void *src, void *dst, const int64_t *table, int nmat, int64_t ntiles, void *stream) {
    if (nmat <= 0 || ntiles <= 0 || ntiles > 0x7fffffff || !src || !dst || !table) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((unsigned)ntiles), dim3(256), 0, as_stream(stream), (const bf16_t *)src, (bf16_t *)dst, table, nmat);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_cast_f32_to_bf16(const float *src, void *dst, int64_t count, void *stream) {
    if (count <= 0) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(count)), dim3(256), 0, as_stream(stream), src, (bf16_t *)dst, count);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_cast_bf16_to_f32(const void *src, float *dst, int64_t count, void *stream) {
    if (count <= 0) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(count)), dim3(256), 0, as_stream(stream), (const bf16_t *)src, dst, count);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

#define ECGVIT_ABI 6
#define ECGVIT_STR2(x) #x
#define ECGVIT_STR(x) ECGVIT_STR2(x)
const char *ecgvit_version(void) { return "ecgvit-hip gfx950 abi" ECGVIT_STR(ECGVIT_ABI); }   // one constant behind both identity calls
int ecgvit_abi_version(void) { return ECGVIT_ABI; }   // 4 (round 4): the probe / stamp / one-item entry points left the product ABI (tools/ecgvit_hip_tools.h); 5: ECGVIT_EPI_NO_OUT, NULL y / dxm in the emitting LayerNorm entry points; 6 (round 5): the quad dropout mask of the 16-bit sites (dropout_p of bf16 tensors is applied as round(256 p) / 256; 0 < p < 1/512 is ECGVIT_EINVAL)

}  // extern "C"
