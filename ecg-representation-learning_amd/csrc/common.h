// Shared device helpers for the gfx950 (CDNA4, wave64) ECG-ViT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ecgvit_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

#define ECGVIT_CHECK_LAUNCH()                                    \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) return ECGVIT_ELAUNCH;            \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- scalar conversions -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// ---- 16-byte vector access: VEC<T>::N elements per 16 B -------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T *p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v) *>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T *p, const Vec16<T> &r) {
    *reinterpret_cast<decltype(r.v) *>(p) = r.v;
}

// ---- wave64 / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// running absolute maximum of an 8-bit emitting kernel -> *dst (delayed scaling: the next step's scale).  One wave-wide maximum, then an atomic
// max on ONE address ONLY when the value would raise what is there: same-address atomics retire one per ~8 ns at the L2 channel, and a
// launch that sends one per wave and tile (49 k in the attention forward at 512 x 12 x 251, 64 k in an FFN-wide product of EcgVit-large) spent
// 300-400 us of a 200-500 us kernel on them (profiles/r04_amax_atomics.txt).  The guard reads the slot at device scope (past the CU's L1):
// a stale value can only be too SMALL, which costs an atomic, never skips one that was needed.  Non-negative floats order as integers.
// `seen` = amax_peek(dst) taken EARLIER by the caller (at the top of the kernel / of the tile's epilogue): the atomic itself is fire-and-forget,
// a load consumed on the spot would make the wave wait an L2 round trip (and, behind stores, for every store before it).
__device__ __forceinline__ unsigned int amax_peek(const float *dst) {
    return __hip_atomic_load(reinterpret_cast<const unsigned int *>(dst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_amax_publish(float *dst, float v, unsigned int seen) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0 && __float_as_uint(v) > seen) atomicMax(reinterpret_cast<unsigned int *>(dst), __float_as_uint(v));   // (v >= 0: never below a seen 0)
}
__device__ __forceinline__ void wave_amax_publish(float *dst, float v) { wave_amax_publish(dst, v, amax_peek(dst)); }
// sum over a block of NW waves; every thread gets the result. `red` = NW floats of LDS.
template <int NW> __device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// ---- exact-erf GELU (torch nn.GELU() default) ------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- counter-based dropout mask: keep(seed, element) is a pure function, recomputed wherever it is needed ----
// One 32-bit hash per PAIR of consecutive elements (2j, 2j+1), 16 random bits each (threshold = p * 2^16).  The pair counter goes
// through a Weyl step (counter * golden ratio + seed mix) and ONE xorshift-multiply-xorshift round: consecutive pairs cost one add
// for the Weyl step (the multiply is shared by a run of pairs) and 8 more ALU slots (v_mul_lo_u32 is quarter rate) -- the FFN
// epilogues and the attention kernels evaluate 4e8 masks per launch, so the hash is priced per instruction.
// The element index is a 32-BIT WRAPPING counter (callers may pass wider integers; only the low 32 bits count): masks of tensors
// beyond 2^32 elements repeat, which is harmless for dropout, and every index computation stays 32-bit VALU.
#define ECGVIT_WEYL 0x9E3779B1u
__device__ __forceinline__ uint32_t seed_mix(uint64_t seed) { return (uint32_t)seed * 0x85EBCA6Bu + (uint32_t)(seed >> 32) * 0xC2B2AE35u + 0x27D4EB2Fu; }
__device__ __forceinline__ uint32_t pair_base(uint64_t seed, uint32_t idx) { return (idx >> 1) * ECGVIT_WEYL + seed_mix(seed); }
__device__ __forceinline__ uint32_t pair_finish(uint32_t h) {
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t pair_hash(uint64_t seed, uint32_t idx) { return pair_finish(pair_base(seed, idx)); }
// both elements of the pair starting at EVEN index idx0
__device__ __forceinline__ void dropout_pair(uint64_t seed, uint32_t idx0, uint32_t thresh, float inv_keep, float &m0, float &m1) {
    const uint32_t h = pair_hash(seed, idx0);
    m0 = (h & 0xFFFFu) >= thresh ? inv_keep : 0.f;
    m1 = (h >> 16) >= thresh ? inv_keep : 0.f;
}
// ---- the bf16 path's hidden / embedding dropout (round 5): ONE hash per FOUR consecutive elements, 8 random bits each (byte b of the hash of
// quad idx / 4 belongs to element idx % 4), keep iff byte >= round(256 p) -- the form the fused attention kernels use since round 2, now for
// every 16-bit dropout site: half the hashes of the pair form, and the keep decision of a whole quad in three bit-parallel instructions.  p is
// applied as round(256 p) / 256 (0.1 -> 0.1016) and rescaled by the exact keep rate 256 / (256 - t): dropout_threshold8 / dropout_inv_keep8
// (0 < p < 1/512 would round to no dropout: the host side rejects it).  The f32 parity path keeps the pair form at the exact p (16 bits).
// quad_keepbits: bit 7 of byte b = keep(element b).  a >= t for bytes: with t < 128 it is a7 | (a_lo >= t_lo), else a7 & (a_lo >= t_lo), and
// bit 7 of (a_lo + 0x80 - t_lo) is (a_lo >= t_lo) without a carry into the next byte.  c4 = (0x80 - (t & 0x7F)) in every byte.
__device__ __forceinline__ uint32_t quad_c4(uint32_t t8) { return (0x80u - (t8 & 0x7Fu)) * 0x01010101u; }
__device__ __forceinline__ uint32_t quad_keepbits(uint32_t h, uint32_t c4, bool t_hi) {
    const uint32_t x1 = (h & 0x7F7F7F7Fu) + c4;
    return t_hi ? (x1 & h) : (x1 | h);
}
__device__ __forceinline__ uint32_t quad_base(uint64_t seed, uint32_t idx) { return (idx >> 2) * ECGVIT_WEYL + seed_mix(seed); }
// finisher of the quad form: TWO xorshift-multiply rounds ("lowbias32").  With one round (pair_finish) the top bits of a byte are not mixed enough
// for counters that advance by d / 4 per row: at thresholds >= 128, where the keep decision rests on a byte's top two bits, single columns of a
// [8192 x 3072] mask were kept 8 standard deviations off the rate (emulated on the host; two rounds: <= 4.2 over every shape / threshold / seed
// tried, tests/test_gpu_ops.py::test_dropout_mask_statistics_bf16_quad_form).  Eight instructions per FOUR elements; the pair form's six per two.
__device__ __forceinline__ uint32_t quad_finish(uint32_t h) {
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
// returns the multiplier: 0 or 1/(1-p).  Q8: the quad form (thresh = 8-bit threshold), else the pair form (16-bit threshold)
template <bool Q8> __device__ __forceinline__ float dropout_mult(uint64_t seed, uint32_t idx, uint32_t thresh, float inv_keep) {
    if constexpr (Q8) {
        const uint32_t h = quad_finish(quad_base(seed, idx));
        return ((h >> (8 * (idx & 3))) & 0xFFu) >= thresh ? inv_keep : 0.f;
    } else {
        const uint32_t h = pair_hash(seed, idx);
        const uint32_t r = (idx & 1) ? (h >> 16) : (h & 0xFFFFu);
        return r >= thresh ? inv_keep : 0.f;
    }
}
// multipliers of VN (4 or 8) consecutive elements starting at a multiple of 4: one Weyl multiply, VN/2 (pair form) or VN/4 (quad form) finishers
template <int VN, bool Q8> __device__ __forceinline__ void dropout_maskN(uint64_t seed, uint32_t idx0, uint32_t thresh, float inv_keep, float (&m)[VN]) {
    if constexpr (Q8) {
        const uint32_t base = quad_base(seed, idx0), c4 = quad_c4(thresh), ik = __float_as_uint(inv_keep);
        const bool t_hi = thresh >= 128u;
#pragma unroll
        for (int q = 0; q < VN / 4; ++q) {
            const uint32_t x = quad_keepbits(quad_finish(base + (uint32_t)q * ECGVIT_WEYL), c4, t_hi);
#pragma unroll
            for (int b = 0; b < 4; ++b)   // the keep bit spread over the dword (v_bfe_i32), ANDed into the multiplier's bits
                m[4 * q + b] = __uint_as_float(ik & (uint32_t)__builtin_amdgcn_sbfe((int)x, 8 * b + 7, 1));
        }
    } else {
        const uint32_t base = pair_base(seed, idx0);
#pragma unroll
        for (int k = 0; k < VN / 2; ++k) {
            const uint32_t h = pair_finish(base + (uint32_t)k * ECGVIT_WEYL);
            m[2 * k] = (h & 0xFFFFu) >= thresh ? inv_keep : 0.f;
            m[2 * k + 1] = (h >> 16) >= thresh ? inv_keep : 0.f;
        }
    }
}
// the bf16 epilogues' run of 8 (quad form)
__device__ __forceinline__ void dropout_mask8(uint64_t seed, uint32_t idx0, uint32_t thresh, float inv_keep, float (&m)[8]) {
    dropout_maskN<8, true>(seed, idx0, thresh, inv_keep, m);
}
// ---- attention-probability dropout: one hash per FOUR consecutive keys of a query, 8 random bits each (threshold = round(p * 256),
// as FlashAttention's kernels quantise p; the kept values are rescaled by the exact 256 / (256 - threshold), so the estimator stays
// unbiased for the probability actually applied).  The score tensor holds 4e8 elements per layer at the benchmark shape and its
// kernels are VALU-bound: the 16-bit pair form above costs 8.5 instruction slots per element there, this one 4.5-5.
//   element (bh, q, key): quad = ((bh * N + q) * ceil(N / 4) + key / 4), byte = key & 3, keep iff byte >= threshold.
__device__ __forceinline__ uint32_t quad_hash(uint32_t seedmix, uint32_t quad) { return pair_finish(quad * ECGVIT_WEYL + seedmix); }
static inline uint32_t dropout_threshold8(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 256.0 + 0.5;
    if (t > 255.0) t = 255.0;
    return (uint32_t)t;
}
static inline float dropout_inv_keep8(float p) { return 256.0f / (256.0f - (float)dropout_threshold8(p)); }

static inline uint32_t dropout_threshold(float p);
// threshold and rescale of a hidden / embedding dropout site: 16-bit element types take the quad mask (8-bit threshold; false: p would round to
// no dropout), f32 the pair mask at the exact p
static inline bool dropout_site_params(float p, bool q8, uint32_t &thresh, float &inv_keep) {
    if (p <= 0.f) { thresh = 0u; inv_keep = 1.f; return true; }
    if (q8) {
        thresh = dropout_threshold8(p);
        inv_keep = dropout_inv_keep8(p);
        return thresh != 0u;
    }
    thresh = dropout_threshold(p);
    inv_keep = 1.0f / (1.0f - p);
    return true;
}

static inline uint32_t dropout_threshold(float p) {
    if (p <= 0.f) return 0u;
    double t = (double)p * 65536.0 + 0.5;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

// ---- bf16-path GELU: erf by Abramowitz-Stegun 7.1.25 (three terms, |err| <= 2.5e-5: an absolute error of 1.3e-5 in Phi, i.e. <= 5e-5 in
//      GELU at |x| = 4 where one bf16 ulp of the result is 0.016 -- far below the rounding of the stored value; round 4 used 7.1.26's five terms
//      at 1.5e-7, two fma per element more in epilogues that are bound by vector-instruction issue), one exp + one rcp;
//      GELU' reuses the same exponential (erf(x/sqrt2) is built on e^{-x^2/2} = sqrt(2 pi) * pdf).
//      ONE formulation for every kernel (the 128^2 GEMM, the persistent GEMM, the split-K reducer): the same record must produce the
//      same bits whichever kernel its batch size selects (eval logits are batch-slice invariant: tests/test_gpu_model.py).
//      The f32 parity path does not come here (erff / expf: gelu_erf, gelu_erf_grad).
//      cdf_k = k * Phi(x), pdf_k = k * phi(x) for a wave-uniform k >= 0 passed as hk = k/2, ck = k/sqrt(2 pi): the FFN-up epilogue
//      folds the dropout rescale 1/(1-p) in here; everything else passes k = 1.  |x| is a source modifier and the exponential is
//      taken as 2^(-u^2), u = |x| sqrt(log2(e)/2): 12 VALU instructions for both parts.
#define ECGVIT_GELU_HK1 0.5f
#define ECGVIT_GELU_CK1 0.39894228040143267794f
__device__ __forceinline__ void gelu_fast_parts_scaled(float x, float hk, float ck, float &cdf, float &pdf) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.47047f * 0.70710678118654752440f, 1.0f));   // v_rcp_f32 (1 ulp)
    const float u = ax * 0.84932180028801904272f;
    const float ex = __builtin_amdgcn_exp2f(-u * u);   // = e^{-x^2/2}
    const float poly = t * (0.3480242f + t * (-0.0958798f + t * 0.7478556f));
    const float s = copysignf(fmaf(-poly, ex, 1.0f), x);   // erf(x / sqrt 2)
    cdf = fmaf(hk, s, hk);
    pdf = ck * ex;
}
__device__ __forceinline__ void gelu_fast_parts(float x, float &cdf, float &pdf) { gelu_fast_parts_scaled(x, ECGVIT_GELU_HK1, ECGVIT_GELU_CK1, cdf, pdf); }
__device__ __forceinline__ float gelu_fast(float x) { float c, p; gelu_fast_parts(x, c, p); return x * c; }
__device__ __forceinline__ void gelu_fast_both(float x, float &y, float &dy) { float c, p; gelu_fast_parts(x, c, p); y = x * c; dy = fmaf(x, p, c); }
__device__ __forceinline__ float gelu_fast_grad(float x) { float c, p; gelu_fast_parts(x, c, p); return fmaf(x, p, c); }
// y = k * gelu(x), dy = k * gelu'(x)
__device__ __forceinline__ void gelu_fast_both_scaled(float x, float hk, float ck, float &y, float &dy) {
    float c, p;
    gelu_fast_parts_scaled(x, hk, ck, c, p);
    y = x * c;
    dy = fmaf(x, p, c);
}

// The same arithmetic on TWO elements per instruction (round 5): gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two f32 per lane) in the
// four cycles of a single one, and hipcc does not form them from the scalar body (|x| as a source modifier and the literal-constant fmas have no
// packed encoding).  Written on float pairs with the constants in registers; every element sees the very operations of gelu_fast_parts_scaled in
// the very order (IEEE fma / mul per half), so the results are bit for bit the scalar ones -- 9.5 vector instructions per element instead of 14.
__device__ __forceinline__ f32x2 pk_splat(float a) { return f32x2{a, a}; }
__device__ __forceinline__ void gelu_fast_both_scaled_x2(f32x2 x, float hk, float ck, f32x2 &y, f32x2 &dy) {
    const f32x2 den = {fmaf(fabsf(x[0]), 0.47047f * 0.70710678118654752440f, 1.0f), fmaf(fabsf(x[1]), 0.47047f * 0.70710678118654752440f, 1.0f)};
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 u = x * pk_splat(0.84932180028801904272f);   // (the sign of u does not reach -u u)
    const f32x2 e = -u * u;
    const f32x2 ex = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    f32x2 poly = __builtin_elementwise_fma(t, pk_splat(0.7478556f), pk_splat(-0.0958798f));
    poly = __builtin_elementwise_fma(t, poly, pk_splat(0.3480242f));
    poly = t * poly;
    const f32x2 s1 = __builtin_elementwise_fma(-poly, ex, pk_splat(1.0f));
    const f32x2 s = {copysignf(s1[0], x[0]), copysignf(s1[1], x[1])};
    const f32x2 cdf = __builtin_elementwise_fma(pk_splat(hk), s, pk_splat(hk));
    const f32x2 pdf = pk_splat(ck) * ex;
    y = x * cdf;
    dy = __builtin_elementwise_fma(x, pdf, cdf);
}

// KEEP masks of 8 consecutive elements (a multiple of 4) for PACKED bf16 pairs (quad form): dword k = pair (2k, 2k + 1): 0xFFFF in the half of
// a kept element.  Per quad the keep bits (3 instructions), per pair one byte permute that puts the two keep bits at the halves' sign
// positions and one packed arithmetic shift that spreads them: the same keep set as dropout_maskN<8, true>, applied to packed results
// with one AND each (the rescale 1/(1-p) is folded into the values beforehand).
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void keepmask8(uint64_t seed, uint32_t idx0, uint32_t thresh, uint32_t (&m)[4]) {
    const uint32_t base = quad_base(seed, idx0), c4 = quad_c4(thresh);
    const bool t_hi = thresh >= 128u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t x = quad_keepbits(quad_finish(base + (uint32_t)q * ECGVIT_WEYL), c4, t_hi);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const uint32_t w = __builtin_amdgcn_perm(x, x, pp ? 0x030C020Cu : 0x010C000Cu);   // bytes (2pp, 2pp + 1) -> the high bytes of the two halves
            m[2 * q + pp] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, w) >> (s16x2_t){15, 15});
        }
    }
}

// ---- epilogue parameters shared by the f32 and bf16 GEMMs -------------------------------------
struct EpiParams {
    int flags;
    const float *bias;
    const void *residual;
    int64_t ldr;
    void *aux;
    int64_t ldaux;
    float alpha;
    uint64_t seed;
    uint32_t drop_thresh;
    float inv_keep;
    int N;  // logical N for the dropout element index m*N+n
};

// apply everything except the final store / ACCUM; TO = element type of aux & residual
template <typename TO> __device__ __forceinline__ float epilogue_value(float acc, int64_t m, int n, const EpiParams &e) {
    float v = acc * e.alpha;
    if (e.flags & ECGVIT_EPI_BIAS) v += e.bias[n];
    const float mult = (e.flags & ECGVIT_EPI_DROPOUT) ? dropout_mult<sizeof(TO) == 2>(e.seed, (uint32_t)m * (uint32_t)e.N + (uint32_t)n, e.drop_thresh, e.inv_keep) : 1.f;
    if (e.flags & ECGVIT_EPI_GELU) {
        if (e.flags & ECGVIT_EPI_GELU_GRAD_AUX) {
            reinterpret_cast<TO *>(e.aux)[m * e.ldaux + n] = from_f32<TO>(gelu_erf_grad(v) * mult);
            v = gelu_erf(v);
        } else {
            reinterpret_cast<TO *>(e.aux)[m * e.ldaux + n] = from_f32<TO>(v);
            // GELU of the value as STORED (so backward, which re-reads aux, sees the same pre-activation)
            v = gelu_erf(to_f32<TO>(from_f32<TO>(v)));
        }
    }
    v *= mult;
    if (e.flags & ECGVIT_EPI_GELU_BWD) v *= gelu_erf_grad(to_f32<TO>(reinterpret_cast<const TO *>(e.aux)[m * e.ldaux + n]));
    if (e.flags & ECGVIT_EPI_MUL_AUX) v *= to_f32<TO>(reinterpret_cast<const TO *>(e.aux)[m * e.ldaux + n]);
    if (e.flags & ECGVIT_EPI_RESIDUAL) v += to_f32<TO>(reinterpret_cast<const TO *>(e.residual)[m * e.ldr + n]);
    return v;
}

static inline EpiParams make_epi(const ecgvit_gemm_desc *d) {
    EpiParams e;
    e.flags = d->epilogue;
    e.bias = d->bias;
    e.residual = d->residual;
    e.ldr = d->ldr;
    e.aux = d->aux;
    e.ldaux = d->ldaux;
    e.alpha = d->alpha;
    e.seed = d->dropout_seed;
    // 16-bit outputs: the quad mask (8-bit threshold, exact rescale of round(256 p) / 256); f32 outputs (parity path): the pair mask at the exact p
    const bool q8 = d->out_dtype == ECGVIT_BF16;
    e.drop_thresh = q8 ? dropout_threshold8(d->dropout_p) : dropout_threshold(d->dropout_p);
    e.inv_keep = d->dropout_p > 0.f ? (q8 ? dropout_inv_keep8(d->dropout_p) : 1.0f / (1.0f - d->dropout_p)) : 1.0f;
    e.N = d->N;
    return e;
}

// internal launchers (defined in gemm_f32.hip / gemm_bf16.hip).  route != nullptr: launch nothing, report the kernel family
// (ECGVIT_KERNEL_*) the same arguments would run on -- one dispatch, whether it is executed or asked about (ecgvit_gemm_kernel)
int ecgvit_gemm_f32_launch(const ecgvit_gemm_desc *d, hipStream_t s, int *route = nullptr);
int ecgvit_gemm_bf16_launch(const ecgvit_gemm_desc *d, hipStream_t s, int *route = nullptr);
