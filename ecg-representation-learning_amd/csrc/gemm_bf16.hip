// bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, f32 accumulate) for the Linear layers of the ECG-ViT step:
// forward (NT), input-gradient (NN) and weight-gradient (TN, split-K) with fused epilogues.
//
// Block = 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 accumulators of
// 32x32 (64 acc VGPRs).  Operands are register-staged (16 B / lane global loads issued one K-tile ahead,
// written to LDS after the MFMA block) into a double-buffered LDS ring, one barrier per K-tile.
//
// LDS images
//   K-contiguous operand  ([rows][64 k], 128-B rows): 16-B chunk index XOR ((row>>1)&7) so the 16 lanes of a
//       ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row (conflict-free fragment reads).
//   MN-contiguous operand ([64 k][128 mn], as stored for NN's B and TN's A,B): 320-B row stride and
//       ds_read_b64_tr_b16 transposed reads (4 k-rows x 16 columns per 16-lane group) -- the hardware
//       transpose makes the k-strided fragment without a transposing store; 320 B = 80 banks puts the 4 rows
//       of one 32-lane half on disjoint bank quarters.
// Epilogue: accumulators -> LDS (f32, [128][132]) -> 16-B coalesced rows with bias / GELU / GELU' / dropout /
//   residual applied in f32, so C, aux and residual traffic is full-line.
// Block order: 1-D grid remapped so that the 8 XCDs each own a contiguous run of tiles, N fastest: the blocks
//   sharing an activation row-panel run on one XCD's L2; the (small) weight matrix streams from L2/MALL.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KC_TILE_BYTES = BM * BK * 2;   // 16 KiB
constexpr int MN_ROW_BYTES = 320;            // 256 B of data + 64 B pad
constexpr int MN_TILE_BYTES = BK * MN_ROW_BYTES;  // 20 KiB
constexpr int CS_LD = BN + 4;
constexpr int CS_BYTES = BM * CS_LD * 4;     // 67584

template <bool KC> struct TileBytes { static constexpr int value = KC ? KC_TILE_BYTES : MN_TILE_BYTES; };

__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- global -> registers (4 x 16 B per thread per operand per K-tile) ---------------------------------
template <bool KC>
__device__ __forceinline__ void g_load(const bf16_t *__restrict__ P, int64_t ld, int mn0, int k0, int MN, int K, u32x4 (&r)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = t + 256 * i;
        if (KC) {
            int row = mn0 + (c >> 3);
            row = row < MN ? row : MN - 1;  // clamp: rows beyond MN only feed outputs that are never stored
            const int k = k0 + (c & 7) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};     // K tail (K % 64 != 0, e.g. the 240-wide patch rows) is zero-filled
            if (k < K) v = *reinterpret_cast<const u32x4 *>(P + (int64_t)row * ld + k);
            r[i] = v;
        } else {
            const int k = k0 + (c >> 4), mn = mn0 + (c & 15) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (k < K && mn < MN) v = *reinterpret_cast<const u32x4 *>(P + (int64_t)k * ld + mn);
            r[i] = v;
        }
    }
}

template <bool KC> __device__ __forceinline__ void s_store(char *S, const u32x4 (&r)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = t + 256 * i;
        if (KC) *reinterpret_cast<u32x4 *>(S + kc_off(c >> 3, c & 7)) = r[i];
        else *reinterpret_cast<u32x4 *>(S + (c >> 4) * MN_ROW_BYTES + (c & 15) * 16) = r[i];
    }
}

// ---- LDS -> MFMA fragment: element j of lane (r = lane&31, h = lane>>5) = X[mn = base + r][k = 16*ks + 8h + j] ----
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <bool KC> __device__ __forceinline__ bf16x8 frag_load(const char *S, int mn_base, int ks, int lane) {
    if constexpr (KC) {
        const int row = mn_base + (lane & 31);
        return *reinterpret_cast<const bf16x8 *>(S + kc_off(row, ks * 2 + (lane >> 5)));
    } else {
        const int g = lane >> 4, i = lane & 15;
        const int col = mn_base + (g & 1) * 16 + (i & 3) * 4;
        const int k = ks * 16 + (g >> 1) * 8 + (i >> 2);
        const char *p = S + k * MN_ROW_BYTES + col * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p + 4 * MN_ROW_BYTES));
        const u32x2 ul = __builtin_bit_cast(u32x2, lo), uh = __builtin_bit_cast(u32x2, hi);
        u32x4 u;
        u[0] = ul[0]; u[1] = ul[1]; u[2] = uh[0]; u[3] = uh[1];
        return __builtin_bit_cast(bf16x8, u);
    }
}

// bijective XCD remap of a 1-D block id (cdna guide T1): XCD x gets a contiguous run of tile ids
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

struct SplitK {
    int splits;         // >= 1
    int k_per_split;    // multiple of BK
    float *slabs;       // [splits][M][N] f32 when splits > 1
};

template <bool A_KC, bool B_KC, typename TO>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(ecgvit_gemm_desc d, EpiParams e, SplitK sk, int tiles_m, int tiles_n) {
    constexpr int SA = TileBytes<A_KC>::value, SB = TileBytes<B_KC>::value;
    constexpr int STAGE = SA + SB;
    constexpr int LDS_BYTES = (2 * STAGE > CS_BYTES) ? 2 * STAGE : CS_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int ntile = tiles_m * tiles_n;
    const int split = blockIdx.x / ntile;
    const int tid = xcd_remap(blockIdx.x - split * ntile, ntile);
    const int tm = tid / tiles_n, tn = tid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = d.M, N = d.N;
    const int kbeg = split * sk.k_per_split;
    const int kend = min(d.K, kbeg + sk.k_per_split);

    const bf16_t *A = reinterpret_cast<const bf16_t *>(d.A);
    const bf16_t *B = reinterpret_cast<const bf16_t *>(d.B);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rb[4];
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        g_load<A_KC>(A, d.lda, m0, kbeg, M, kend, ra);
        g_load<B_KC>(B, d.ldb, n0, kbeg, N, kend, rb);
        s_store<A_KC>(smem, ra);
        s_store<B_KC>(smem + SA, rb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char *sa = smem + (kt & 1) * STAGE;
        const char *sb = sa + SA;
        if (kt + 1 < nk) {
            g_load<A_KC>(A, d.lda, m0, kbeg + (kt + 1) * BK, M, kend, ra);
            g_load<B_KC>(B, d.ldb, n0, kbeg + (kt + 1) * BK, N, kend, rb);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag_load<A_KC>(sa, wm * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag_load<B_KC>(sb, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            char *na = smem + ((kt + 1) & 1) * STAGE;
            s_store<A_KC>(na, ra);
            s_store<B_KC>(na + SA, rb);
        }
        __syncthreads();
    }

    // ---- epilogue: acc -> LDS f32 -> coalesced 16-B rows -------------------------------------------
    float *Cs = reinterpret_cast<float *>(smem);
    {
        const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CS_LD + wn * 64 + j * 32 + lr] = acc[i][j][r];
    }
    __syncthreads();
    const int cc = (threadIdx.x & 15) * 8;
    const int n = n0 + cc;
#pragma unroll 2
    for (int p = 0; p < 8; ++p) {
        const int rr = p * 16 + (threadIdx.x >> 4);
        const int64_t m = m0 + rr;
        if (m >= M || n >= N) continue;
        float v[8];
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(&Cs[rr * CS_LD + cc]);
        const f32x4 c1 = *reinterpret_cast<const f32x4 *>(&Cs[rr * CS_LD + cc + 4]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = c0[k]; v[4 + k] = c1[k]; }
        if (sk.splits > 1) {  // raw partial sums; the epilogue runs in the split-K reduce kernel
            float *o = sk.slabs + ((int64_t)split * M + m) * N + n;
            *reinterpret_cast<f32x4 *>(o) = c0;
            *reinterpret_cast<f32x4 *>(o + 4) = c1;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= e.alpha;
        if (e.flags & ECGVIT_EPI_BIAS) {
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(e.bias + n), b1 = *reinterpret_cast<const f32x4 *>(e.bias + n + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
        }
        if constexpr (sizeof(TO) == 2) {
            float mult[8];
            const bool drop = e.flags & ECGVIT_EPI_DROPOUT;
            if (drop) dropout_mask8(e.seed, (uint32_t)m * (uint32_t)e.N + (uint32_t)n, e.drop_thresh, e.inv_keep, mult);
            if (e.flags & ECGVIT_EPI_GELU) {
                Vec16<bf16_t> sav;
                if (e.flags & ECGVIT_EPI_GELU_GRAD_AUX) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float dy;
                        gelu_fast_both(v[k], v[k], dy);
                        sav.set(k, drop ? dy * mult[k] : dy);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) sav.set(k, v[k]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = gelu_fast(sav.get(k));
                }
                st16(reinterpret_cast<bf16_t *>(e.aux) + m * e.ldaux + n, sav);
            }
            if (drop) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= mult[k];
            }
            if (e.flags & ECGVIT_EPI_GELU_BWD) {
                const Vec16<bf16_t> pre = ld16(reinterpret_cast<const bf16_t *>(e.aux) + m * e.ldaux + n);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= gelu_fast_grad(pre.get(k));
            }
            if (e.flags & ECGVIT_EPI_MUL_AUX) {
                const Vec16<bf16_t> a = ld16(reinterpret_cast<const bf16_t *>(e.aux) + m * e.ldaux + n);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= a.get(k);
            }
            if (e.flags & ECGVIT_EPI_RESIDUAL) {
                const Vec16<bf16_t> res = ld16(reinterpret_cast<const bf16_t *>(e.residual) + m * e.ldr + n);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += res.get(k);
            }
            bf16_t *o = reinterpret_cast<bf16_t *>(d.C) + m * d.ldc + n;
            if (e.flags & ECGVIT_EPI_ACCUM) {
                const Vec16<bf16_t> old = ld16(o);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += old.get(k);
            }
            Vec16<bf16_t> out;
#pragma unroll
            for (int k = 0; k < 8; ++k) out.set(k, v[k]);
            st16(o, out);
        } else {
            // f32 output (weight gradients): bias / alpha / accum only
            float *o = reinterpret_cast<float *>(d.C) + m * d.ldc + n;
            if (e.flags & ECGVIT_EPI_ACCUM) {
                const f32x4 o0 = *reinterpret_cast<const f32x4 *>(o), o1 = *reinterpret_cast<const f32x4 *>(o + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] += o0[k]; v[4 + k] += o1[k]; }
            }
            f32x4 w0, w1;
#pragma unroll
            for (int k = 0; k < 4; ++k) { w0[k] = v[k]; w1[k] = v[4 + k]; }
            *reinterpret_cast<f32x4 *>(o) = w0;
            *reinterpret_cast<f32x4 *>(o + 4) = w1;
        }
    }
}

// out[m,n] = alpha * sum_s slab[s][m][n] (+ bias[n]) (+ out[m,n])   -- split-K combine, deterministic order
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slabs, int splits, int64_t MN, int N,
                                                            TO *__restrict__ C, int64_t ldc, EpiParams e) {
    const int64_t nv = MN / 4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        f32x4 s = *reinterpret_cast<const f32x4 *>(slabs + i * 4);
        for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4 *>(slabs + (int64_t)k * MN + i * 4);
        const int64_t m = (i * 4) / N;
        const int n = (int)((i * 4) - m * N);
        TO *o = C + m * ldc + n;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = s[k] * e.alpha;
            if (e.flags & ECGVIT_EPI_BIAS) v += e.bias[n + k];
            if (e.flags & ECGVIT_EPI_ACCUM) v += to_f32<TO>(o[k]);
            o[k] = from_f32<TO>(v);
        }
    }
}

inline int choose_splits(const ecgvit_gemm_desc *d, int ntile) {
    if (d->layout != ECGVIT_GEMM_TN) return 1;
    const int ksteps = (d->K + BK - 1) / BK;
    int s = (768 + ntile - 1) / ntile;  // aim for ~3 waves of blocks over 256 CUs
    s = std::min(s, std::max(1, ksteps / 8));  // keep >= 8 K-steps per split
    return std::max(1, std::min(s, 64));
}

}  // namespace

// large weight-gradient products (gemm_wgrad.hip) and large A . B^T products (gemm_nt.hip)
bool ecgvit_gemm_wgrad_applicable(const ecgvit_gemm_desc *d);
int64_t ecgvit_gemm_wgrad_workspace(const ecgvit_gemm_desc *d);
int ecgvit_gemm_wgrad_launch(const ecgvit_gemm_desc *d, hipStream_t s);
bool ecgvit_gemm_nt_applicable(const ecgvit_gemm_desc *d);
int ecgvit_gemm_nt_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag);
int ecgvit_gemm_nt4w_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag);   // gemm_nt.hip: the four-wave body

extern "C" int64_t ecgvit_gemm_workspace(const ecgvit_gemm_desc *d) {
    if ((d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2) && d->layout == ECGVIT_GEMM_TN) return ecgvit_gemm_wgrad_workspace(d);
    if (d->dtype != ECGVIT_BF16 || d->layout != ECGVIT_GEMM_TN) return 0;
    const int ntile = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
    const int s = choose_splits(d, ntile);
    const int64_t v1 = s > 1 ? (int64_t)s * d->M * d->N * 4 : 0;
    return std::max(v1, ecgvit_gemm_wgrad_workspace(d));
}

int ecgvit_gemm_bf16_launch(const ecgvit_gemm_desc *d, hipStream_t s, int *route) {
    if (d->dtype != ECGVIT_BF16) return ECGVIT_EINVAL;
    if (d->out_dtype != ECGVIT_BF16 && d->out_dtype != ECGVIT_F32) return ECGVIT_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch1 != 1 || d->batch2 != 1) return ECGVIT_EINVAL;
    if (d->N % 8 != 0 || d->lda % 8 != 0 || d->ldb % 8 != 0 || d->ldc % 8 != 0) return ECGVIT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) | reinterpret_cast<uintptr_t>(d->C)) % 16) return ECGVIT_EINVAL;
    const bool a_kc = d->layout != ECGVIT_GEMM_TN, b_kc = d->layout == ECGVIT_GEMM_NT;
    if ((a_kc || b_kc) && d->K % 8 != 0) return ECGVIT_EINVAL;      // K-contiguous operands move 8-element chunks
    if (!a_kc && d->M % 8 != 0) return ECGVIT_EINVAL;
    if (d->out_dtype == ECGVIT_F32 &&
        (d->epilogue & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_RESIDUAL | ECGVIT_EPI_DROPOUT)))
        return ECGVIT_EINVAL;
    if ((d->epilogue & ECGVIT_EPI_BIAS) && (reinterpret_cast<uintptr_t>(d->bias) % 16)) return ECGVIT_EINVAL;
    if ((d->epilogue & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_MUL_AUX)) && (!d->aux || d->ldaux % 8 || reinterpret_cast<uintptr_t>(d->aux) % 16)) return ECGVIT_EINVAL;
    if ((d->epilogue & ECGVIT_EPI_RESIDUAL) && (!d->residual || d->ldr % 8 || reinterpret_cast<uintptr_t>(d->residual) % 16)) return ECGVIT_EINVAL;

    if (d->layout != ECGVIT_GEMM_NT && d->layout != ECGVIT_GEMM_NN && d->layout != ECGVIT_GEMM_TN) return ECGVIT_EINVAL;
    if (ecgvit_gemm_nt_applicable(d)) {
        if (route) { *route = ECGVIT_KERNEL_GEMM_NT; return ECGVIT_OK; }
        return ecgvit_gemm_nt_launch(d, s, 0, 0);
    }
    if (ecgvit_gemm_wgrad_applicable(d)) {
        if (route) { *route = ECGVIT_KERNEL_GEMM_WGRAD; return ECGVIT_OK; }
        return ecgvit_gemm_wgrad_launch(d, s);
    }
    if (route) { *route = ECGVIT_KERNEL_GEMM_BF16; return ECGVIT_OK; }
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    SplitK sk;
    sk.splits = 1;
    sk.slabs = nullptr;
    sk.k_per_split = ((d->K + BK - 1) / BK) * BK;
    if (d->workspace && d->layout == ECGVIT_GEMM_TN) {
        int sp = choose_splits(d, ntile);
        while (sp > 1 && (int64_t)sp * d->M * d->N * 4 > d->workspace_bytes) --sp;
        if (sp > 1 && ((int64_t)d->M * d->N) % 4 == 0 &&
            !(d->epilogue & ~(ECGVIT_EPI_BIAS | ECGVIT_EPI_ACCUM))) {
            const int ksteps = (d->K + BK - 1) / BK;
            sk.splits = sp;
            sk.k_per_split = ((ksteps + sp - 1) / sp) * BK;
            sk.slabs = reinterpret_cast<float *>(d->workspace);
        }
    }
    EpiParams e = make_epi(d);
    dim3 grid((unsigned)(ntile * sk.splits)), block(256);
#define LAUNCH(AK, BKC, TO) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKC, TO>), grid, block, 0, s, *d, e, sk, tiles_m, tiles_n)
    const bool obf = d->out_dtype == ECGVIT_BF16;
    switch (d->layout) {
        case ECGVIT_GEMM_NT: if (obf) LAUNCH(true, true, bf16_t); else LAUNCH(true, true, float); break;
        case ECGVIT_GEMM_NN: if (obf) LAUNCH(true, false, bf16_t); else LAUNCH(true, false, float); break;
        case ECGVIT_GEMM_TN: if (obf) LAUNCH(false, false, bf16_t); else LAUNCH(false, false, float); break;
        default: return ECGVIT_EINVAL;
    }
#undef LAUNCH
    ECGVIT_CHECK_LAUNCH();
    if (sk.splits > 1) {
        const int64_t MN = (int64_t)d->M * d->N;
        const int g = (int)std::min<int64_t>((MN / 4 + 255) / 256, 2048);
        if (obf) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(g), dim3(256), 0, s, sk.slabs, sk.splits, MN, d->N, (bf16_t *)d->C, d->ldc, e);
        else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(g), dim3(256), 0, s, sk.slabs, sk.splits, MN, d->N, (float *)d->C, d->ldc, e);
        ECGVIT_CHECK_LAUNCH();
    }
    return ECGVIT_OK;
}

// the one dispatch of ecgvit_gemm: executed (route == nullptr) or only asked about (route receives ECGVIT_KERNEL_*)
static int gemm_dispatch(const ecgvit_gemm_desc *d, void *stream, int *route) {
    if (!d) return ECGVIT_EINVAL;
    if ((d->epilogue & ECGVIT_EPI_AUX8) && !ecgvit_gemm_nt_applicable(d)) return ECGVIT_EINVAL;   // the e4m3 saved tensor exists on the large A.B^T kernel only
    if ((d->epilogue & ECGVIT_EPI_DROPOUT) && d->dropout_p > 0.f && d->out_dtype == ECGVIT_BF16 && dropout_threshold8(d->dropout_p) == 0u)
        return ECGVIT_EINVAL;   // 16-bit outputs draw 8 bits per element: 0 < p < 1/512 would silently round to no dropout
    if (d->epilogue & ECGVIT_EPI_NO_OUT) {   // no-output form: the 8-bit A . B^T kernel's emitting FFN-wide bodies only (ecgvit_gemm_nt_applicable holds the list)
        const bool f8 = d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2;
        if (!f8 || d->layout != ECGVIT_GEMM_NT || !(d->epilogue & ECGVIT_EPI_QUANT_OUT) || !d->A || !d->B || !d->aux ||
            (reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) | reinterpret_cast<uintptr_t>(d->C)) % 16)
            return ECGVIT_EINVAL;
    }
    if (d->epilogue & ECGVIT_EPI_COLSUM) {
        if (!d->colsum_out || !d->workspace || d->batch1 != 1 || d->batch2 != 1) return ECGVIT_EINVAL;
        if (d->dtype == ECGVIT_BF16 && ecgvit_gemm_nt_applicable(d)) return ecgvit_gemm_bf16_launch(d, as_stream(stream), route);   // fused column sums
        if ((d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2) && ecgvit_gemm_nt_applicable(d)) {
            if (route) { *route = ECGVIT_KERNEL_GEMM_NT; return ECGVIT_OK; }
            return ecgvit_gemm_nt_launch(d, as_stream(stream), 0, 0);
        }
        // generic path: plain GEMM, then the stand-alone column-sum kernel over the stored output
        if (d->workspace_bytes < ecgvit_colsum_workspace(d->M, d->N)) return ECGVIT_EINVAL;
        ecgvit_gemm_desc g = *d;
        g.epilogue &= ~ECGVIT_EPI_COLSUM;
        const int rc = gemm_dispatch(&g, stream, route);
        if (rc != ECGVIT_OK || route) return rc;
        return ecgvit_colsum(d->C, d->ldc, d->colsum_out, d->workspace, d->M, d->N, d->out_dtype, stream);
    }
    if (d->dtype == ECGVIT_F32) return ecgvit_gemm_f32_launch(d, as_stream(stream), route);
    if (d->dtype == ECGVIT_BF16) return ecgvit_gemm_bf16_launch(d, as_stream(stream), route);
    if ((d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2) && d->layout == ECGVIT_GEMM_TN) {
        // 8-bit weight gradients dW = dY8^T . X8 (A in `dtype`, B e4m3, f32 output): the streaming split-K kernel only
        if (!d->C || reinterpret_cast<uintptr_t>(d->C) % 16 || d->ldc % 4 || !ecgvit_gemm_wgrad_applicable(d)) return ECGVIT_EINVAL;
        if (route) { *route = ECGVIT_KERNEL_GEMM_WGRAD; return ECGVIT_OK; }
        return ecgvit_gemm_wgrad_launch(d, as_stream(stream));
    }
    if (d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2) {   // 8-bit operands: the large A . B^T kernel only (no small-shape fallback)
        if (!d->A || !d->B || (!d->C && !(d->epilogue & ECGVIT_EPI_NO_OUT)) ||
            (reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B) | reinterpret_cast<uintptr_t>(d->C)) % 16)
            return ECGVIT_EINVAL;
        if ((d->epilogue & ECGVIT_EPI_BIAS) && (!d->bias || reinterpret_cast<uintptr_t>(d->bias) % 16)) return ECGVIT_EINVAL;
        if ((d->epilogue & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_MUL_AUX)) && (!d->aux || d->ldaux % 8 || reinterpret_cast<uintptr_t>(d->aux) % 16)) return ECGVIT_EINVAL;
        if ((d->epilogue & ECGVIT_EPI_RESIDUAL) && (!d->residual || d->ldr % 8 || reinterpret_cast<uintptr_t>(d->residual) % 16)) return ECGVIT_EINVAL;
        if (d->ldc % 8 != 0 || !ecgvit_gemm_nt_applicable(d)) return ECGVIT_EINVAL;
        if (route) { *route = ECGVIT_KERNEL_GEMM_NT; return ECGVIT_OK; }
        return ecgvit_gemm_nt_launch(d, as_stream(stream), 0, 0);
    }
    return ECGVIT_EINVAL;
}

extern "C" int ecgvit_gemm(const ecgvit_gemm_desc *d, void *stream) { return gemm_dispatch(d, stream, nullptr); }

extern "C" int ecgvit_gemm_kernel(const ecgvit_gemm_desc *d) {
    int route = ECGVIT_KERNEL_NONE;
    return gemm_dispatch(d, nullptr, &route) == ECGVIT_OK ? route : ECGVIT_KERNEL_NONE;
}

#ifdef ECGVIT_TOOLS
// tools build only (libecgvit_hip_tools.so): one A . B^T call on gemm_nt_kernel with column groups of raster_g n-tiles (0 = the
// built-in order) and diag bits (1 = stamped instantiation, 2 = its output stores dropped): tools/gemm_ab.py, tools/nt_stamps.py
extern "C" int ecgvit_tools_gemm(const ecgvit_gemm_desc *d, void *stream, int kernel, int raster_g, int diag) {
    if (!d) return ECGVIT_EINVAL;
    if (kernel == 2) return ecgvit_gemm_nt_applicable(d) ? ecgvit_gemm_nt_launch(d, as_stream(stream), raster_g, diag) : ECGVIT_EINVAL;
    if (kernel == 3) return ecgvit_gemm_nt_applicable(d) ? ecgvit_gemm_nt4w_launch(d, as_stream(stream), raster_g, diag) : ECGVIT_EINVAL;
    return ecgvit_gemm(d, stream);
}
#endif
