// fp8 operand path of the Linear GEMMs (BASELINE.json configs[4]: EcgVit-large, CDNA4 fp8 MFMA): per-tensor scaled OCP formats --
// e4m3 for activations and weights, e5m2 ("bf8") for the gradients entering the input-gradient products.  x ~= q * scale with
// scale = amax / FORMAT_MAX; activations and gradients use DELAYED scaling (the scale of step t comes from the amax the quantise
// pass of step t-1 accumulated; values beyond it saturate), weights are rescaled from their own amax every time the optimiser has
// rewritten them.  Everything here is HBM-bound elementwise work: 16-B vector loads, at most one atomic max per wave.
#include "common.h"

namespace {

constexpr float FP8_E4M3_MAX = 448.f, BF8_E5M2_MAX = 57344.f;

__device__ __forceinline__ float fmt_max(int fmt) { return fmt == ECGVIT_BF8_E5M2 ? BF8_E5M2_MAX : FP8_E4M3_MAX; }

__device__ __forceinline__ void wave_atomic_max(float *dst, float v) {
    wave_amax_publish(dst, v);
}

// segments: table[2*s] = first element (multiple of 16), table[2*s+1] = element count (multiple of 8) of segment s = blockIdx.y; a null table = one
// segment (off0, n0).  amax[s] = max(amax[s], max |x|).
__global__ __launch_bounds__(256) void fp8_amax_kernel(const bf16_t *__restrict__ x, const int64_t *__restrict__ table, int64_t off0, int64_t n0,
                                                       float *__restrict__ amax) {
    const int s = blockIdx.y;
    const int64_t off = table ? table[2 * s] : off0, n = table ? table[2 * s + 1] : n0;
    float m = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
        const Vec16<bf16_t> v = ld16(x + off + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v.get(k)));
    }
    wave_atomic_max(amax + s, m);
}

// y[i] = saturate(x[i] / scale[s]) in the 8-bit format; amax_next[s] (optional) accumulates max |x| for the next step's scale
template <int FMT>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const bf16_t *__restrict__ x, uint8_t *__restrict__ y, const int64_t *__restrict__ table,
                                                           int64_t off0, int64_t n0, const float *__restrict__ scale, float *__restrict__ amax_next) {
    const int s = blockIdx.y;
    const int64_t off = table ? table[2 * s] : off0, n = table ? table[2 * s + 1] : n0;
    const float sc = scale[s];
    const float inv = sc > 0.f ? 1.0f / sc : 0.f;
    constexpr float MX = FMT == ECGVIT_BF8_E5M2 ? BF8_E5M2_MAX : FP8_E4M3_MAX;
    float m = 0.f;
    auto q8 = [&](const Vec16<bf16_t> &v) {   // 8 bf16 -> 8 bytes
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float r = v.get(k);
            m = fmaxf(m, fabsf(r));
            f[k] = __builtin_amdgcn_fmed3f(r * inv, -MX, MX);
        }
        int w0 = 0, w1 = 0;
        if constexpr (FMT == ECGVIT_BF8_E5M2) {
            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
        } else {
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        }
        u32x2 o;
        o[0] = (uint32_t)w0; o[1] = (uint32_t)w1;
        return o;
    };
    // 16 elements per lane and step: two 16-B loads, one 16-B store (an 8-element tail, if any, goes as one 8-B store)
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += (int64_t)gridDim.x * 256 * 16) {
        if (i + 16 <= n) {
            const Vec16<bf16_t> v0 = ld16(x + off + i), v1 = ld16(x + off + i + 8);
            const u32x2 a = q8(v0), b = q8(v1);
            u32x4 o;
            o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
            *reinterpret_cast<u32x4 *>(y + off + i) = o;
        } else {
            *reinterpret_cast<u32x2 *>(y + off + i) = q8(ld16(x + off + i));
        }
    }
    if (amax_next) wave_atomic_max(amax_next + s, m);
}

__global__ void fp8_scale_update_kernel(float *__restrict__ scale, float *__restrict__ amax, int n, const int32_t *__restrict__ fmt, int fmt_all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i];
    if (a > 0.f) scale[i] = a / fmt_max(fmt ? fmt[i] : fmt_all);
    else if (!(scale[i] > 0.f)) scale[i] = 1.0f;      // never seen a non-zero value: any scale represents zeros exactly
    amax[i] = 0.f;
}

inline dim3 seg_grid(int64_t n_max, int nseg) {
    const int64_t b = (n_max / 16 + 255) / 256;
    return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(b, nseg > 1 ? 256 : 4096)), (unsigned)nseg);
}

}  // namespace

extern "C" {

int ecgvit_fp8_amax(const void *x, const int64_t *table, int nseg, int64_t count, float *amax, void *stream) {
    if (!x || !amax || nseg < 1 || count <= 0 || count % 8 || (!table && nseg != 1)) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(fp8_amax_kernel, seg_grid(count, nseg), dim3(256), 0, as_stream(stream), (const bf16_t *)x, table, (int64_t)0, count, amax);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_fp8_quantize(const void *x, void *y, const int64_t *table, int nseg, int64_t count, int format, const float *scale, float *amax_next,
                        void *stream) {
    if (!x || !y || !scale || nseg < 1 || count <= 0 || count % 8 || (!table && nseg != 1)) return ECGVIT_EINVAL;
    if (format == ECGVIT_FP8_E4M3)
        hipLaunchKernelGGL(fp8_quantize_kernel<ECGVIT_FP8_E4M3>, seg_grid(count, nseg), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (uint8_t *)y, table,
                           (int64_t)0, count, scale, amax_next);
    else if (format == ECGVIT_BF8_E5M2)
        hipLaunchKernelGGL(fp8_quantize_kernel<ECGVIT_BF8_E5M2>, seg_grid(count, nseg), dim3(256), 0, as_stream(stream), (const bf16_t *)x, (uint8_t *)y, table,
                           (int64_t)0, count, scale, amax_next);
    else return ECGVIT_EINVAL;
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_fp8_scale_update(float *scale, float *amax, int n, const int32_t *formats, int format_all, void *stream) {
    if (!scale || !amax || n < 1) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(fp8_scale_update_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), scale, amax, n, formats, format_all);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

}  // extern "C"
