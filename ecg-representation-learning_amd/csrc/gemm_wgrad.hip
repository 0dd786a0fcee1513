// Weight-gradient products dW = dY^T . X for the Linear layers (C = A^T . B, both operands k-major: A[K, M], B[K, N] row-major with
// K = token rows): 256 x 256 x 64 block tile, 8 waves (2 x 4, wave tile 128 x 64), v_mfma_f32_32x32x16_bf16, split-K over the token
// rows with f32 slabs + a deterministic reduce.
//
// gemm_wgrad_kernel: one (K-slice, tile) item per block; operands stream HBM -> LDS by LDS-DMA (`buffer_load ... lds`, no VGPR
// staging): A tiles ([64 k][256 m], 32 KiB) two K-tiles ahead in a 3-slot ring, B tiles one ahead in 2 slots (5 x 32 KiB = all of
// LDS); a K-tile = four 16-deep phases (12 transposed `ds_read_b64_tr_b16` fragment reads + 2 DMA pieces | 8 MFMAs), ONE counted
// vmcnt(4) per K-tile, waves 4-7 one barrier behind waves 0-3 (ping-pong per SIMD).  K-slices are pinned to XCDs (slice counts that
// are multiples of 8) or laid out XCD-contiguously, so a slice's operand rows live in one L2.
// LDS image of a k-major operand: [64 k][256 mn] (512-B rows), 16-B chunk ^= (k&3)<<2 (applied to the per-lane SOURCE address: the
// DMA destination is wave-uniform base + lane*16) -- the 4 k-rows of one transposed half-wave read land on the 4 distinct 64-B
// quarters of the bank row.  Rows beyond a slice's end fall outside the buffer descriptor and read as zero.
// The transposed reads are inline asm: with an LDS-DMA in flight hipcc's waitcnt pass drains vmcnt(0) in front of the builtin.
// Also here: the column-sum reducer shared with gemm_nt.hip's EPI_COLSUM.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = 32768;
constexpr int CS_LD = 68, CS_WAVE_BYTES = 64 * CS_LD * 4;  // 17408: per-wave epilogue patch (64 rows x 64 f32 + pad)
constexpr int LDS_BYTES = 163840;                          // all of LDS: 3 A + 2 B tiles; the epilogue patches need 139264

typedef __attribute__((address_space(3))) void *lptr_t;

// ---- fast DMA path: buffer_load ... lds with a per-lane byte offset computed ONCE per output tile and the K advance in
// the scalar offset -> zero VALU per piece (the generic path above spends ~15 VALU ops + a 64-bit address per piece).
// Out-of-range rows fall beyond the descriptor's num_records and read as zero (hardware bounds check), so no clamping.
// Valid when no piece can straddle a row end: K % 64 == 0 for K-contiguous operands, MN % 256 == 0 for k-major ones.
typedef int v4i32 __attribute__((ext_vector_type(4)));
struct FastOp {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[4];
};
template <bool KC>
__device__ __forceinline__ FastOp fast_setup(const bf16_t *P, int64_t ld, int mn0, int MN, int kend, int wave, int lane) {
    FastOp f;
    // k-major operands: rows >= kend (the split's end) must read as zero -> shrink the descriptor to kend rows
    const uint32_t bytes = KC ? (uint32_t)((int64_t)MN * ld * 2) : (uint32_t)((int64_t)kend * ld * 2);
    f.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P, 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wave * 4 + i;
        if constexpr (KC) {
            const int r = j * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            f.voff[i] = (int)(((int64_t)(mn0 + r) * ld + c * 8) * 2);
        } else {
            const int kr = j * 2 + (lane >> 5);
            const int c = (lane & 31) ^ ((kr & 3) << 2);
            f.voff[i] = (int)(((int64_t)kr * ld + mn0 + c * 8) * 2);
        }
    }
    return f;
}
template <bool KC, int I0 = 0, int I1 = 4>
__device__ __forceinline__ void fast_dma(const FastOp &f, int64_t ld, int k0, char *tile, int wave) {
    const int soff = KC ? k0 * 2 : (int)((int64_t)k0 * ld * 2);
#pragma unroll
    for (int i = I0; i < I1; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(f.rsrc, (lptr_t)(tile + (wave * 4 + i) * 1024), 16, f.voff[i], soff, 0, 0);
}

// fragment: element j of lane (r = lane&31, h = lane>>5) = X[mn = base + r][k = 16*ks + 8h + j]
template <bool KC> __device__ __forceinline__ bf16x8 frag(const char *tile, int mn_base, int ks, int lane) {
    if constexpr (KC) {
        const int row = mn_base + (lane & 31);
        return *reinterpret_cast<const bf16x8 *>(tile + row * 128 + (((ks * 2 + (lane >> 5)) ^ ((row >> 1) & 7)) << 4));
    } else {
        const int g = lane >> 4, i = lane & 15;
        const int colb = (mn_base + (g & 1) * 16 + (i & 3) * 4) * 2;
        const int k = ks * 16 + (g >> 1) * 8 + (i >> 2);  // k and k+4 share (k&3)
        const char *p = tile + k * 512 + (colb ^ ((k & 3) << 6));
        // Inline asm, not the builtin: with an LDS-DMA in flight hipcc's waitcnt pass cannot prove that the builtin's read does not
        // alias the DMA's destination and drains vmcnt(0) in front of it -- which serialised every K-tile's prefetch of the k-major
        // layouts (found in the ISA: `s_waitcnt vmcnt(0)` right after the 8 `buffer_load ... lds` of a phase).  Every caller already
        // orders these reads by hand (s_waitcnt lgkmcnt(0) + sched_barrier before the consuming MFMAs / the barrier).
        const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(pa) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(pa) : "memory");
        const u32x2 ul = __builtin_bit_cast(u32x2, lo), uh = __builtin_bit_cast(u32x2, hi);
        u32x4 u;
        u[0] = ul[0]; u[1] = ul[1]; u[2] = uh[0]; u[3] = uh[1];
        return __builtin_bit_cast(bf16x8, u);
    }
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

struct SplitK2 {
    int splits, k_per_split;
    float *slabs;
};

// Finish one 8-column piece of an output row: split-K slab store, or alpha / bias / GELU (+aux) / dropout / GELU' / residual /
// accumulate, then a 16-B (bf16) or 2 x 16-B (f32) store.  `cs` accumulates the column sums of what is stored (EPI_COLSUM).
template <typename TO>
__device__ __forceinline__ void epi_row8(const f32x4 c0, const f32x4 c1, int64_t m, int n, const ecgvit_gemm_desc &d,
                                         const EpiParams &e, const SplitK2 &sk, int split, float (&cs)[8]) {
    const int M = d.M, N = d.N;
    if (m >= M || n >= N) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = c0[k]; v[4 + k] = c1[k]; }
    if (sk.splits > 1) {
        float *o = sk.slabs + ((int64_t)split * M + m) * N + n;
        *reinterpret_cast<f32x4 *>(o) = c0;
        *reinterpret_cast<f32x4 *>(o + 4) = c1;
        return;
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
            if (e.flags & ECGVIT_EPI_GELU_GRAD_AUX) {   // aux = gelu'(v) * dropout multiplier: all the backward of this site needs
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
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] += out.get(k);
    } else {
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

__device__ __forceinline__ void epi_colsum_flush(float (&cs)[8], const ecgvit_gemm_desc &d, const EpiParams &e, int m0, int n, int wm,
                                                 int lane) {
    if (!(e.flags & ECGVIT_EPI_COLSUM)) return;
    // lanes with equal (lane & 7) hold the same 8 columns: fold the 8 row groups, then one partial row per (tile row, wm)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cs[k] += __shfl_xor(cs[k], 8, 64);
        cs[k] += __shfl_xor(cs[k], 16, 64);
        cs[k] += __shfl_xor(cs[k], 32, 64);
    }
    if (lane < 8 && n < d.N) {
        float *pr = reinterpret_cast<float *>(d.workspace) + ((int64_t)(m0 / BM) * 2 + wm) * d.N + n;
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[k] = cs[k];
    }
}

// Drain one wave's 128 x 64 accumulator block through its private LDS patch (two 64-row passes) and store 128-B row segments.
template <typename TO>
__device__ __forceinline__ void epilogue_store(f32x16 (&acc)[4][2], char *smem, const ecgvit_gemm_desc &d, const EpiParams &e,
                                               const SplitK2 &sk, int split, int m0, int n0, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float *Cs = reinterpret_cast<float *>(smem + wave * CS_WAVE_BYTES);
    const int lr = lane & 31, lh = lane >> 5;
    const int cc = (lane & 7) * 8;
    const int n = n0 + wn * 64 + cc;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * CS_LD + j * 32 + lr] = acc[hh * 2 + ii][j][r];
        // same-wave LDS ops execute in order: the reads below see the writes above
#pragma unroll 1
        for (int p = 0; p < 8; ++p) {
            const int rr = p * 8 + (lane >> 3);
            const f32x4 c0 = *reinterpret_cast<const f32x4 *>(&Cs[rr * CS_LD + cc]);
            const f32x4 c1 = *reinterpret_cast<const f32x4 *>(&Cs[rr * CS_LD + cc + 4]);
            epi_row8<TO>(c0, c1, (int64_t)m0 + wm * 128 + hh * 64 + rr, n, d, e, sk, split, cs);
        }
    }
    epi_colsum_flush(cs, d, e, m0, n, wm, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient kernel ("TQ": both operands k-major, split-K): the Q kernel's streaming discipline on the k-major images.
// One (K-slice, tile) item per block; A tiles ([64 k][256 m], 32 KiB) two K-tiles ahead in a 3-slot ring, B tiles one ahead in
// 2 slots (5 x 32 KiB = all of LDS); a K-tile = four 16-deep phases (12 transposed reads + 2 DMA pieces | 8 MFMAs of 32x32x16),
// one counted vmcnt(4) per K-tile, waves 4-7 one barrier behind waves 0-3.  The whole-tile schedule it replaces drained
// vmcnt(0) once per K-tile and wave half (64 KiB in flight at most): both operands of this layout are pure HBM streams.
template <typename TO>
__global__ __launch_bounds__(512, 2) void gemm_wgrad_kernel(ecgvit_gemm_desc d, EpiParams e, SplitK2 sk, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int ntile = tiles_m * tiles_n;
    int split, tid;
    if (sk.splits > 1 && (sk.splits & 7) == 0) {
        const int r = sk.splits >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        split = x + 8 * (q % r);
        tid = q / r;
    } else if (sk.splits > 1) {
        const int gid = xcd_remap(blockIdx.x, ntile * sk.splits);
        split = gid / ntile;
        tid = gid - split * ntile;
    } else {
        split = 0;
        tid = xcd_remap(blockIdx.x, ntile);
    }
    const int tm = tid / tiles_n, tn = tid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * sk.k_per_split;
    const int kend = min(d.K, kbeg + sk.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const bf16_t *A = reinterpret_cast<const bf16_t *>(d.A);
    const bf16_t *B = reinterpret_cast<const bf16_t *>(d.B);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wm == 1;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const FastOp fa = fast_setup<false>(A, d.lda, m0, d.M, kend, wave, lane);
    const FastOp fb = fast_setup<false>(B, d.ldb, n0, d.N, kend, wave, lane);
    char *const ringA = smem, *const ringB = smem + 3 * TILE_BYTES;
    // prologue: A(0), B(0), A(1)
    fast_dma<false>(fa, d.lda, kbeg, ringA, wave);
    fast_dma<false>(fb, d.ldb, kbeg, ringB, wave);
    if (nk > 1) {
        fast_dma<false>(fa, d.lda, kbeg + BK, ringA + TILE_BYTES, wave);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();

    int ga = 0, gb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char *sa = ringA + ga * TILE_BYTES;
        const char *sb = ringB + gb * TILE_BYTES;
        char *nA = ringA + (ga == 0 ? 2 : ga - 1) * TILE_BYTES;   // slot of K-tile kt+2
        char *nB = ringB + (gb ^ 1) * TILE_BYTES;                 // slot of K-tile kt+1
        const bool b_ok = kt + 1 < nk, a_ok = kt + 2 < nk;
        const int kB = kbeg + (kt + 1) * BK, kA = kbeg + (kt + 2) * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = frag<false>(sb, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag<false>(sa, wm * 128 + i * 32, ks, lane);
            if (ks == 0) { if (b_ok) fast_dma<false, 0, 2>(fb, d.ldb, kB, nB, wave); }
            else if (ks == 1) { if (b_ok) fast_dma<false, 2, 4>(fb, d.ldb, kB, nB, wave); }
            else if (ks == 2) { if (a_ok) fast_dma<false, 0, 2>(fa, d.lda, kA, nA, wave); }
            else {
                // the K-tile's one counted wait: everything but A(kt+2) (4 pieces per wave) has landed
                if (a_ok) { fast_dma<false, 2, 4>(fa, d.lda, kA, nA, wave); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // WAR by construction: the leading group refills this K-tile's B slot in its NEXT phase (phase 0 of kt+1), which runs
                // while the trailing group is still in this phase's MFMA half -- so this phase's reads retire BEFORE the barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        ga = ga == 2 ? 0 : ga + 1;
        gb ^= 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();   // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_store<TO>(acc, smem, d, e, sk, split, m0, n0, wave, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// gemm_wgrad_kernel_4w: the same items, images, ring, counted waits and epilogue with FOUR waves -- one per SIMD, each owning a 128 x 128
// block of the tile in 256 accumulator registers (4 x 4 tiles of v_mfma_f32_32x32x16_bf16).  A 16-deep k-step is ONE instruction stream
// of 16 MFMAs (512 cycles) with the next k-step's 16 transposed fragment reads and four DMA pieces placed between them in a fixed
// order; 64 instead of 96 fragment reads per k-step and CU, ONE workgroup barrier per K-tile instead of eight (it stands between
// k-steps 2 and 3: behind it K-tile kt+1 is visible and nobody reads K-tile kt from LDS any more, so B(kt+2) and A(kt+3) may go out).
// Same MFMA, same K order per accumulator: results are bit-identical to gemm_wgrad_kernel's.
template <int... X, typename F>
__device__ __forceinline__ void w4_static_for_impl(std::integer_sequence<int, X...>, F &&f) { (f(std::integral_constant<int, X>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void w4_static_for(F &&f) { w4_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

template <typename TO>
__global__ __launch_bounds__(256, 1) void gemm_wgrad_kernel_4w(ecgvit_gemm_desc d, EpiParams e, SplitK2 sk, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int ntile = tiles_m * tiles_n;
    int split, tid;
    if (sk.splits > 1 && (sk.splits & 7) == 0) {
        const int r = sk.splits >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        split = x + 8 * (q % r);
        tid = q / r;
    } else if (sk.splits > 1) {
        const int gid = xcd_remap(blockIdx.x, ntile * sk.splits);
        split = gid / ntile;
        tid = gid - split * ntile;
    } else {
        split = 0;
        tid = xcd_remap(blockIdx.x, ntile);
    }
    const int tm = tid / tiles_n, tn = tid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * sk.k_per_split;
    const int kend = min(d.K, kbeg + sk.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lda2 = (int)d.lda * 2, ldb2 = (int)d.ldb * 2;

    // DMA: eight 1-KiB pieces of a 32-KiB tile per wave; piece 8*wave + i = k-rows 16*wave + 2*i + {0, 1}: pieces i and i + 4 share the
    // per-lane offset (same row swizzle), the 8 rows between them go into the scalar offset.  Rows >= kend read as zero (descriptor);
    // pieces of K-tiles past the slice's end are sent through an EMPTY descriptor instead of around a branch.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, (uint32_t)((int64_t)kend * lda2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.B), 0, (uint32_t)((int64_t)kend * ldb2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, 0u, 0x00020000);
    int voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = 16 * wave + 2 * i + (lane >> 5);
        const int c = (lane & 31) ^ ((kr & 3) << 2);
        voA[i] = kr * lda2 + (m0 + c * 8) * 2;
        voB[i] = kr * ldb2 + (n0 + c * 8) * 2;
    }
#define W4_PIECE(RS, VO, LD2, SLOT, K0, I) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lptr_t)(smem + (SLOT) * TILE_BYTES + (8 * wave + (I)) * 1024), 16, VO[(I) & 3], (K0) * (LD2) + ((I) >> 2) * 8 * (LD2), 0, 0)
    // fragment reads: lane (g = lane >> 4, i = lane & 15) of the 32 x 16 fragment of 32-column tile t, k-step ks reads 8 bytes at
    // k = 16 ks + 8 (g >> 1) + (i >> 2) [and k + 4], columns 32 t + 16 (g & 1) + 4 (i & 3) ..: per lane one offset per tile t (the image's row
    // swizzle folds into the tile index), k-step and the k + 4 half are immediates
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    uint32_t foA[4], foB[4];
    {
        const int g = lane >> 4, i = lane & 15;
        const int base = ((g >> 1) * 8 + (i >> 2)) * 512 + ((g & 1) * 16 + (i & 3) * 4) * 2, sw = (i >> 2) & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            foA[t] = sm + base + wm * 256 + ((t ^ sw) << 6);
            foB[t] = sm + base + wn * 256 + ((t ^ sw) << 6);
        }
    }
    // two fragment buffers (this k-step's and the next one's): lo = k .. k+3 half, hi = k+4 .. half of a lane's 8 k
    bf16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
#define W4_READ(DST_LO, DST_HI, ADDR, KS)                                                                                    \
    do {                                                                                                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST_LO) : "v"(ADDR), "n"((KS) * 8192) : "memory");         \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST_HI) : "v"(ADDR), "n"((KS) * 8192 + 2048) : "memory");  \
    } while (0)
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)

    f32x16 acc[2][4][2];   // [column half][row tile][column tile of the half]: the epilogue drains two 128 x 64 halves
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // prologue: A(0), B(0), A(1); behind the barrier the first half of B(1) and the first k-step's fragments
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsA, voA, lda2, 0, kbeg, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsB, voB, ldb2, 3, kbeg, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsA, voA, lda2, 1, kbeg + BK, i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) W4_PIECE(rsB, voB, ldb2, 4, kbeg + BK, i);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t ab = foA[t], bb = foB[t] + 3 * TILE_BYTES;
        W4_READ(blo[0][t], bhi[0][t], bb, 0);
        W4_READ(alo[0][t], ahi[0][t], ab, 0);
    }

    int ga = 0, gb = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int ga1 = ga == 2 ? 0 : ga + 1, ga2 = ga == 0 ? 2 : ga - 1, gb1 = gb ^ 1;
        const int kB1 = kbeg + (kt + 1) * BK, kA2 = kbeg + (kt + 2) * BK;
        const bool ok1 = kt + 1 < nk, ok2 = kt + 2 < nk;
        const __amdgpu_buffer_rsrc_t rB1 = ok1 ? rsB : rs0, rA2 = ok2 ? rsA : rs0, rB2 = ok2 ? rsB : rs0;
        // one k-step: 16 MFMAs from buffer CUR; the next k-step's 16 reads into buffer CUR ^ 1 (tile bases NA / NB, k-step NKS) behind
        // MFMAs 0-11 (two each behind the first four: all B fragments, then one A half per MFMA) and four DMA pieces behind MFMAs 2, 6, 10, 14
#define W4_KSTEP(CUR, NA, NB, NKS, RS, VO, LD2, SLOT, K0, P0)                                                                \
    do {                                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
        W4_FENCE();                                                                                                          \
        w4_static_for<16>([&](auto xc_) __attribute__((always_inline)) {                                                     \
            constexpr int x_ = decltype(xc_)::value, i_ = x_ >> 2, j_ = x_ & 3;                                              \
            {                                                                                                                \
                const u32x2 al_ = __builtin_bit_cast(u32x2, alo[CUR][i_]), ah_ = __builtin_bit_cast(u32x2, ahi[CUR][i_]);    \
                const u32x2 bl_ = __builtin_bit_cast(u32x2, blo[CUR][j_]), bh_ = __builtin_bit_cast(u32x2, bhi[CUR][j_]);    \
                const u32x4 au_ = {al_[0], al_[1], ah_[0], ah_[1]}, bu_ = {bl_[0], bl_[1], bh_[0], bh_[1]};                   \
                acc[j_ >> 1][i_][j_ & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, au_), __builtin_bit_cast(bf16x8, bu_), \
                                                                                  acc[j_ >> 1][i_][j_ & 1], 0, 0, 0);        \
            }                                                                                                                \
            W4_FENCE();                                                                                                      \
            if constexpr (x_ < 4) {                                                                                          \
                W4_READ(blo[(CUR) ^ 1][x_], bhi[(CUR) ^ 1][x_], (NB)[x_], NKS);                                              \
                W4_FENCE();                                                                                                  \
            } else if constexpr (x_ < 12) {                                                                                  \
                constexpr int t_ = (x_ - 4) >> 1;                                                                            \
                if constexpr ((x_ & 1) == 0) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[(CUR) ^ 1][t_]) : "v"((NA)[t_]), "n"((NKS) * 8192) : "memory"); \
                else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[(CUR) ^ 1][t_]) : "v"((NA)[t_]), "n"((NKS) * 8192 + 2048) : "memory"); \
                W4_FENCE();                                                                                                  \
            }                                                                                                                \
            if constexpr ((x_ & 3) == 2) {                                                                                   \
                W4_PIECE(RS, VO, LD2, SLOT, K0, (P0) + (x_ >> 2));                                                           \
                W4_FENCE();                                                                                                  \
            }                                                                                                                \
        });                                                                                                                  \
    } while (0)
        uint32_t ca[4], cb[4], na[4], nb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ca[t] = foA[t] + ga * TILE_BYTES; cb[t] = foB[t] + (3 + gb) * TILE_BYTES;
            na[t] = foA[t] + ga1 * TILE_BYTES; nb[t] = foB[t] + (3 + gb1) * TILE_BYTES;
        }
        W4_KSTEP(0, ca, cb, 1, rB1, voB, ldb2, 3 + gb1, kB1, 4);    // second half of B(kt+1)
        W4_KSTEP(1, ca, cb, 2, rA2, voA, lda2, ga2, kA2, 0);        // A(kt+2)
        W4_KSTEP(0, ca, cb, 3, rA2, voA, lda2, ga2, kA2, 4);
        // the K-tile's one counted wait and one barrier: everything but A(kt+2) has landed; this wave's reads of K-tile kt have retired
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        __builtin_amdgcn_s_barrier();
        W4_FENCE();
        W4_KSTEP(1, na, nb, 0, rB2, voB, ldb2, 3 + gb, kB1 + BK, 0);   // first half of B(kt+2), into the slot of B(kt); reads of K-tile kt+1
        ga = ga1; gb = gb1;
    }
#undef W4_KSTEP
#undef W4_READ
#undef W4_PIECE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W4_FENCE();
    __builtin_amdgcn_s_barrier();
#undef W4_FENCE
    // the epilogue's per-wave LDS patches: one per (wave, column half) -- eight in all, as with eight waves
    epilogue_store<TO>(acc[0], smem, d, e, sk, split, m0, n0, 4 * wm + 2 * wn, lane);
    epilogue_store<TO>(acc[1], smem, d, e, sk, split, m0, n0, 4 * wm + 2 * wn + 1, lane);
}


// ------------------------------------------------------------------------------------------------------------------
// The same kernel on 8-BIT operands (fp8_linear; BASELINE.json configs[4]): dW = dY8^T . X8 with dY8 in e5m2 (or e4m3) and X8 in e4m3,
// both k-major ([token rows][features], one byte per element -- the copies the forward / input-gradient products already own), on the
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the bf16 MFMA rate).  A K-tile is 128 token rows deep in
// the same 32-KiB tile ([128 k][256 mn] bytes, 256-B rows): the same DMA pieces, barriers and counted waits as above, four phases of
// 4 MFMAs (64 cycles each) per K-tile, i.e. twice the K per byte that crosses the CU's memory path.
// Fragments come from `ds_read_b64_tr_b8` (probed on the device, tools/probe_tr8.hip): in a 16-lane group source lane s supplies the
// 8 bytes at ITS address and result lane i receives byte (i & 7) of source lanes 2j + (i >> 3), j = 0..7 -- with source lane s pointing
// at k-row s >> 1, bytes 8 (s & 1) .. +7 of a 16-byte column chunk, lane i ends up with 8 consecutive k of column i.  Four such reads
// (k = 32 hb + 8 t + j) are the 32 operand bytes of one lane (row lane & 31, k-half hb = lane >> 5); A and B use the same k order.
// Image swizzle (on the per-lane SOURCE address of the DMA): 16-B chunk ^= (k & 7) << 1 -- the 8 k-rows x 2 column chunks a 32-lane
// read cycle touches land on 16 distinct 16-B slots of the 256-B bank row.
typedef int i32x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ FastOp fast_setup8(const uint8_t *P, int64_t ld, int mn0, int kend, int wave, int lane) {
    FastOp f;
    f.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P, 0, (uint32_t)((int64_t)kend * ld), 0x00020000);   // rows >= kend read as zero
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = (wave * 4 + i) * 4 + (lane >> 4);          // k-row of this lane inside the K-tile (a piece = 4 rows x 256 B)
        const int c = (lane & 15) ^ ((kr & 7) << 1);
        f.voff[i] = (int)((int64_t)kr * ld + mn0 + c * 16);
    }
    return f;
}
template <int I0 = 0, int I1 = 4>
__device__ __forceinline__ void fast_dma8(const FastOp &f, int64_t ld, int k0, char *tile, int wave) {
    const int soff = (int)((int64_t)k0 * ld);
#pragma unroll
    for (int i = I0; i < I1; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(f.rsrc, (lptr_t)(tile + (wave * 4 + i) * 1024), 16, f.voff[i], soff, 0, 0);
}
// per-lane byte offset (inside a tile image) of the lane's first transposed read for a 32-column block starting at column mn_base (a
// multiple of 32); read t of k-step ks adds 16384 ks + 2048 t
__device__ __forceinline__ uint32_t frag8_lane_off(int mn_base, int lane) {
    const int s = lane & 15, j = s >> 1;
    const int chunk = (mn_base >> 4) + ((lane >> 4) & 1);
    return (uint32_t)((32 * (lane >> 5) + j) * 256 + ((chunk ^ (j << 1)) << 4) + 8 * (s & 1));
}
template <int KS> __device__ __forceinline__ i32x8_t frag8(uint32_t lds_addr) {
    u32x2 t0, t1, t2, t3;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(t0) : "v"(lds_addr), "n"(KS * 16384) : "memory");
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(t1) : "v"(lds_addr), "n"(KS * 16384 + 2048) : "memory");
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(t2) : "v"(lds_addr), "n"(KS * 16384 + 4096) : "memory");
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(t3) : "v"(lds_addr), "n"(KS * 16384 + 6144) : "memory");
    return i32x8_t{(int)t0[0], (int)t0[1], (int)t1[0], (int)t1[1], (int)t2[0], (int)t2[1], (int)t3[0], (int)t3[1]};
}

template <int AFMT>   // format of A (= dY): 0 e4m3, 1 e5m2; B (= X) is e4m3
__global__ __launch_bounds__(512, 2) void gemm_wgrad8_kernel(ecgvit_gemm_desc d, EpiParams e, SplitK2 sk, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    constexpr int BK8 = 128;
    const int ntile = tiles_m * tiles_n;
    int split, tid;
    if (sk.splits > 1 && (sk.splits & 7) == 0) {
        const int r = sk.splits >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        split = x + 8 * (q % r);
        tid = q / r;
    } else if (sk.splits > 1) {
        const int gid = xcd_remap(blockIdx.x, ntile * sk.splits);
        split = gid / ntile;
        tid = gid - split * ntile;
    } else {
        split = 0;
        tid = xcd_remap(blockIdx.x, ntile);
    }
    const int tm = tid / tiles_n, tn = tid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * sk.k_per_split;
    const int kend = min(d.K, kbeg + sk.k_per_split);
    const int nk = (kend - kbeg + BK8 - 1) / BK8;
    const uint8_t *A = reinterpret_cast<const uint8_t *>(d.A);
    const uint8_t *B = reinterpret_cast<const uint8_t *>(d.B);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wm == 1;
    if (d.scale_a) e.alpha *= *d.scale_a;     // per-tensor scales of the 8-bit operands (device scalars); the split-K reducer applies them
    if (d.scale_b) e.alpha *= *d.scale_b;     // itself when the partial sums go through slabs

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const FastOp fa = fast_setup8(A, d.lda, m0, kend, wave, lane);
    const FastOp fb = fast_setup8(B, d.ldb, n0, kend, wave, lane);
    char *const ringA = smem, *const ringB = smem + 3 * TILE_BYTES;
    uint32_t offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = frag8_lane_off(wm * 128 + i * 32, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = frag8_lane_off(wn * 64 + j * 32, lane);
    // prologue: A(0), B(0), A(1)
    fast_dma8(fa, d.lda, kbeg, ringA, wave);
    fast_dma8(fb, d.ldb, kbeg, ringB, wave);
    if (nk > 1) {
        fast_dma8(fa, d.lda, kbeg + BK8, ringA + TILE_BYTES, wave);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();

    int ga = 0, gb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const uint32_t sa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)(ringA + ga * TILE_BYTES);
        const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)(ringB + gb * TILE_BYTES);
        char *nA = ringA + (ga == 0 ? 2 : ga - 1) * TILE_BYTES;   // slot of K-tile kt+2
        char *nB = ringB + (gb ^ 1) * TILE_BYTES;                 // slot of K-tile kt+1
        const bool b_ok = kt + 1 < nk, a_ok = kt + 2 < nk;
        const int kB = kbeg + (kt + 1) * BK8, kA = kbeg + (kt + 2) * BK8;
        i32x8_t b[2];
        // phase p: k-step p >> 1 (64 token rows), A row blocks 2 (p & 1) .. +1: 16 or 8 transposed reads + 2 DMA pieces | 4 MFMAs
#define W8_PHASE(P)                                                                                                          \
        {                                                                                                                    \
            constexpr int KS = (P) >> 1, IH = (P) & 1;                                                                       \
            i32x8_t a[2];                                                                                                    \
            if (IH == 0) { b[0] = frag8<KS>(sb + offB[0]); b[1] = frag8<KS>(sb + offB[1]); }                                 \
            a[0] = frag8<KS>(sa + offA[2 * IH]); a[1] = frag8<KS>(sa + offA[2 * IH + 1]);                                    \
            if ((P) == 0) { if (b_ok) fast_dma8<0, 2>(fb, d.ldb, kB, nB, wave); }                                            \
            else if ((P) == 1) { if (b_ok) fast_dma8<2, 4>(fb, d.ldb, kB, nB, wave); }                                       \
            else if ((P) == 2) { if (a_ok) fast_dma8<0, 2>(fa, d.lda, kA, nA, wave); }                                       \
            else {                                                                                                           \
                if (a_ok) { fast_dma8<2, 4>(fa, d.lda, kA, nA, wave); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }     \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
            }                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            __builtin_amdgcn_s_barrier();                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            __builtin_amdgcn_s_setprio(1);                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                      \
                acc[2 * IH + i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[2 * IH + i][j], AFMT, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
            __builtin_amdgcn_s_setprio(0);                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            __builtin_amdgcn_s_barrier();                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }
        W8_PHASE(0) W8_PHASE(1) W8_PHASE(2) W8_PHASE(3)
#undef W8_PHASE
        ga = ga == 2 ? 0 : ga + 1;
        gb ^= 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();   // re-align the two groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    epilogue_store<float>(acc, smem, d, e, sk, split, m0, n0, wave, lane);
}

// gemm_wgrad8_kernel_4w: the 8-bit weight-gradient kernel with FOUR waves (one per SIMD, 128 x 128 per wave in 256 accumulator registers),
// as gemm_wgrad_kernel_4w is to gemm_wgrad_kernel: a 64-deep k-step is one instruction stream of 16 MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4,
// 64 cycles each) with the next k-step's 32 transposed reads and eight DMA pieces between them; one barrier per 128-deep K-tile, between its
// two k-steps.  Bit-identical to gemm_wgrad8_kernel.
template <int AFMT>
__global__ __launch_bounds__(256, 1) void gemm_wgrad8_kernel_4w(ecgvit_gemm_desc d, EpiParams e, SplitK2 sk, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    constexpr int BK8 = 128;
    const int ntile = tiles_m * tiles_n;
    int split, tid;
    if (sk.splits > 1 && (sk.splits & 7) == 0) {
        const int r = sk.splits >> 3, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        split = x + 8 * (q % r);
        tid = q / r;
    } else if (sk.splits > 1) {
        const int gid = xcd_remap(blockIdx.x, ntile * sk.splits);
        split = gid / ntile;
        tid = gid - split * ntile;
    } else {
        split = 0;
        tid = xcd_remap(blockIdx.x, ntile);
    }
    const int tm = tid / tiles_n, tn = tid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * sk.k_per_split;
    const int kend = min(d.K, kbeg + sk.k_per_split);
    const int nk = (kend - kbeg + BK8 - 1) / BK8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lda1 = (int)d.lda, ldb1 = (int)d.ldb;   // byte pitches
    if (d.scale_a) e.alpha *= *d.scale_a;
    if (d.scale_b) e.alpha *= *d.scale_b;

    // DMA: eight 1-KiB pieces (4 k-rows x 256 B) of a 32-KiB tile per wave: piece 8*wave + i = k-rows 32*wave + 4*i + (lane >> 4); pieces i and
    // i + 4 share the per-lane offset (same swizzle: 16 rows apart), the 16 rows go into the scalar offset
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, (uint32_t)((int64_t)kend * lda1), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.B), 0, (uint32_t)((int64_t)kend * ldb1), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, 0u, 0x00020000);
    int voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kr = 32 * wave + 4 * i + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 7) << 1);
        voA[i] = kr * lda1 + m0 + c * 16;
        voB[i] = kr * ldb1 + n0 + c * 16;
    }
#define W4_PIECE(RS, VO, LD1, SLOT, K0, I) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lptr_t)(smem + (SLOT) * TILE_BYTES + (8 * wave + (I)) * 1024), 16, VO[(I) & 3], (K0) * (LD1) + ((I) >> 2) * 16 * (LD1), 0, 0)
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    uint32_t foA[4], foB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        foA[t] = sm + frag8_lane_off(wm * 128 + t * 32, lane);
        foB[t] = sm + frag8_lane_off(wn * 128 + t * 32, lane);
    }
    // two fragment buffers; fragment f: 0-3 = B column blocks, 4-7 = A row blocks; four 8-byte transposed reads each
    u32x2 fr[2][8][4];
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define W4_READ8(BUF, F, T, ADDR, KS) \
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(fr[BUF][F][T]) : "v"(ADDR), "n"((KS) * 16384 + 2048 * (T)) : "memory")

    f32x16 acc[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // prologue: A(0), B(0), A(1); behind the barrier B(1) and the first k-step's fragments
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsA, voA, lda1, 0, kbeg, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsB, voB, ldb1, 3, kbeg, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsA, voA, lda1, 1, kbeg + BK8, i);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) W4_PIECE(rsB, voB, ldb1, 4, kbeg + BK8, i);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const uint32_t ad = f < 4 ? foB[f & 3] + 3 * TILE_BYTES : foA[f & 3];
#pragma unroll
        for (int t = 0; t < 4; ++t) W4_READ8(0, f, t, ad, 0);
    }

    int ga = 0, gb = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int ga1 = ga == 2 ? 0 : ga + 1, ga2 = ga == 0 ? 2 : ga - 1, gb1 = gb ^ 1;
        const int kn = kbeg + (kt + 2) * BK8;
        const bool ok2 = kt + 2 < nk;
        const __amdgpu_buffer_rsrc_t rA2 = ok2 ? rsA : rs0, rB2 = ok2 ? rsB : rs0;
        uint32_t ca[4], cb[4], na[4], nb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ca[t] = foA[t] + ga * TILE_BYTES; cb[t] = foB[t] + (3 + gb) * TILE_BYTES;
            na[t] = foA[t] + ga1 * TILE_BYTES; nb[t] = foB[t] + (3 + gb1) * TILE_BYTES;
        }
        // one k-step: 16 MFMAs from buffer CUR; the next k-step's 32 reads into CUR ^ 1 behind MFMAs 0-11 (three each behind the first
        // eight, two behind the next four; B fragments first) and eight DMA pieces, one behind every other MFMA
#define W8_KSTEP(CUR, NA, NB, NKS, RS, VO, LD1, SLOT)                                                                         \
    do {                                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
        W4_FENCE();                                                                                                          \
        w4_static_for<16>([&](auto xc_) __attribute__((always_inline)) {                                                     \
            constexpr int x_ = decltype(xc_)::value, i_ = x_ >> 2, j_ = x_ & 3;                                              \
            {                                                                                                                \
                const i32x8_t a_ = {(int)fr[CUR][4 + i_][0][0], (int)fr[CUR][4 + i_][0][1], (int)fr[CUR][4 + i_][1][0], (int)fr[CUR][4 + i_][1][1], \
                                    (int)fr[CUR][4 + i_][2][0], (int)fr[CUR][4 + i_][2][1], (int)fr[CUR][4 + i_][3][0], (int)fr[CUR][4 + i_][3][1]}; \
                const i32x8_t b_ = {(int)fr[CUR][j_][0][0], (int)fr[CUR][j_][0][1], (int)fr[CUR][j_][1][0], (int)fr[CUR][j_][1][1],   \
                                    (int)fr[CUR][j_][2][0], (int)fr[CUR][j_][2][1], (int)fr[CUR][j_][3][0], (int)fr[CUR][j_][3][1]};   \
                acc[j_ >> 1][i_][j_ & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, acc[j_ >> 1][i_][j_ & 1], AFMT, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F); \
            }                                                                                                                \
            W4_FENCE();                                                                                                      \
            constexpr int r0_ = x_ < 8 ? 3 * x_ : 24 + 2 * (x_ - 8), nr_ = x_ < 8 ? 3 : x_ < 12 ? 2 : 0;                     \
            if constexpr (nr_ > 0) { W4_READ8((CUR) ^ 1, (r0_ >> 2), (r0_ & 3), ((r0_ >> 2) < 4 ? (NB)[(r0_ >> 2) & 3] : (NA)[(r0_ >> 2) & 3]), NKS); } \
            if constexpr (nr_ > 1) { W4_READ8((CUR) ^ 1, ((r0_ + 1) >> 2), ((r0_ + 1) & 3), (((r0_ + 1) >> 2) < 4 ? (NB)[((r0_ + 1) >> 2) & 3] : (NA)[((r0_ + 1) >> 2) & 3]), NKS); } \
            if constexpr (nr_ > 2) { W4_READ8((CUR) ^ 1, ((r0_ + 2) >> 2), ((r0_ + 2) & 3), (((r0_ + 2) >> 2) < 4 ? (NB)[((r0_ + 2) >> 2) & 3] : (NA)[((r0_ + 2) >> 2) & 3]), NKS); } \
            if constexpr (nr_ > 0) W4_FENCE();                                                                               \
            if constexpr ((x_ & 1) == 1) {                                                                                   \
                W4_PIECE(RS, VO, LD1, SLOT, kn, x_ >> 1);                                                                    \
                W4_FENCE();                                                                                                  \
            }                                                                                                                \
        });                                                                                                                  \
    } while (0)
        W8_KSTEP(0, ca, cb, 1, rA2, voA, lda1, ga2);          // A(kt+2) goes out
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        W4_FENCE();
        __builtin_amdgcn_s_barrier();
        W4_FENCE();
        W8_KSTEP(1, na, nb, 0, rB2, voB, ldb1, 3 + gb);       // B(kt+2) into the slot of B(kt); reads of K-tile kt+1
        ga = ga1; gb = gb1;
    }
#undef W8_KSTEP
#undef W4_READ8
#undef W4_PIECE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    W4_FENCE();
    __builtin_amdgcn_s_barrier();
#undef W4_FENCE
    epilogue_store<float>(acc[0], smem, d, e, sk, split, m0, n0, 4 * wm + 2 * wn, lane);
    epilogue_store<float>(acc[1], smem, d, e, sk, split, m0, n0, 4 * wm + 2 * wn + 1, lane);
}

template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce2_kernel(const float *__restrict__ slabs, int splits, int64_t MN, int N,
                                                             TO *__restrict__ C, int64_t ldc, EpiParams e,
                                                             const float *__restrict__ scale_a = nullptr, const float *__restrict__ scale_b = nullptr) {
    if (scale_a) e.alpha *= *scale_a;   // per-tensor scales of 8-bit operands
    if (scale_b) e.alpha *= *scale_b;
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

__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float *__restrict__ partial, int nparts, int N, float *__restrict__ out) {
    // 64 columns x 16 row slices per block (N/64 blocks only: the parallelism comes from inside the block); deterministic tree
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        int p = slice;
        for (; p + 48 < nparts; p += 64) {
            s0 += partial[(int64_t)p * N + c];
            s1 += partial[(int64_t)(p + 16) * N + c];
            s2 += partial[(int64_t)(p + 32) * N + c];
            s3 += partial[(int64_t)(p + 48) * N + c];
        }
        for (; p < nparts; p += 16) s0 += partial[(int64_t)p * N + c];
    }
    red[slice][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < N) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 4) { t[0] += red[r][lane]; t[1] += red[r + 1][lane]; t[2] += red[r + 2][lane]; t[3] += red[r + 3][lane]; }
        out[c] = (t[0] + t[1]) + (t[2] + t[3]);
    }
}

inline int choose_splits2(const ecgvit_gemm_desc *d, int ntile) {
    if (d->layout != ECGVIT_GEMM_TN) return 1;
    const int ksteps = (d->K + BK - 1) / BK;
    // one block per CU: fill ONE round of the 256 CUs (never 2.1 rounds); a multiple of 8 slices lets each XCD own whole K-slices
    int s = 256 / ntile;
    // a multiple of 8 slices lets each XCD own whole K-slices (best L2 locality), but only if the rounding leaves < 7 % of the CUs idle:
    // 27 tiles x 8 slices = 216 blocks wastes 16 % of the chip for the whole launch, 27 x 9 = 243 (XCD-contiguous order) does not
    if (s >= 8 && (s & ~7) * ntile * 100 >= s * ntile * 93) s &= ~7;
    if (s < 1) s = 1;
    // tiles_per_workgroup (launches that share the GPU with RCCL kernels) does NOT change the slicing here.  Three times as many,
    // shorter slices would bound the tail of a block that finds its CU held, but they cost every launch: measured inside the step on
    // a 1-rank RCCL group (bench.py --single-rank-collectives, profiles/r02_dp_single_rank.txt) +2.3 ms of gemm_wgrad and +1.1 ms of
    // split-K reduce per step, against the ~8 % of the backward during which a bucket's all-reduce actually holds CUs.
    s = std::min(s, std::max(1, ksteps / 16));   // keep >= 16 K-steps per slice
    return std::max(1, std::min(s, 64));
}

}  // namespace

void ecgvit_colsum_reduce_launch(const float *partial, int nparts, int N, float *out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, s, partial, nparts, N, out);
}

// large weight-gradient products only (both extents whole tiles, a long reduction); everything else stays on gemm_bf16.hip's 128^2 kernel
bool ecgvit_gemm_wgrad_applicable(const ecgvit_gemm_desc *d) {
    const bool f8 = d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2;
    if (d->layout != ECGVIT_GEMM_TN || !(d->dtype == ECGVIT_BF16 || f8) || d->batch1 != 1 || d->batch2 != 1) return false;
    if (d->epilogue & ~(ECGVIT_EPI_BIAS | ECGVIT_EPI_ACCUM)) return false;
    if (d->K < 4096 || d->M % 256 != 0 || d->N % 256 != 0) return false;
    if (f8) {   // 8-bit operands: f32 output, 16-B aligned rows (one DMA lane = 16 bytes of a row)
        if (d->out_dtype != ECGVIT_F32 || d->lda % 16 || d->ldb % 16 || !d->A || !d->B ||
            (reinterpret_cast<uintptr_t>(d->A) | reinterpret_cast<uintptr_t>(d->B)) % 16)
            return false;
        return (int64_t)d->K * d->lda + 65536 * d->lda < (1ll << 31) && (int64_t)d->K * d->ldb + 65536 * d->ldb < (1ll << 31);
    }
    return (int64_t)d->K * d->lda * 2 + 65536 * d->lda < (1ll << 31) && (int64_t)d->K * d->ldb * 2 + 65536 * d->ldb < (1ll << 31);
}

int64_t ecgvit_gemm_wgrad_workspace(const ecgvit_gemm_desc *d) {
    if (!ecgvit_gemm_wgrad_applicable(d)) return 0;
    const int ntile = (d->M / BM) * (d->N / BN);
    const int s = choose_splits2(d, ntile);
    return s > 1 ? (int64_t)s * d->M * d->N * 4 : 0;
}

#ifdef ECGVIT_TOOLS
static int g_tools_wgrad_8w = 0;
extern "C" void ecgvit_tools_wgrad_body(int eight_wave) { g_tools_wgrad_8w = eight_wave; }   // tools/wgrad_ab.py: which body the next launches take
#endif
// argument validation is done by the caller (ecgvit_gemm_bf16_launch)
int ecgvit_gemm_wgrad_launch(const ecgvit_gemm_desc *d, hipStream_t s) {
    const int tiles_m = d->M / BM, tiles_n = d->N / BN, ntile = tiles_m * tiles_n;
    const int ksteps = (d->K + BK - 1) / BK;
    SplitK2 sk;
    sk.splits = 1;
    sk.slabs = nullptr;
    sk.k_per_split = ksteps * BK;
    if (d->workspace) {
        int sp = choose_splits2(d, ntile);
        while (sp > 1 && (int64_t)sp * d->M * d->N * 4 > d->workspace_bytes) --sp;
        if (sp > 1) {
            sk.splits = sp;
            sk.k_per_split = ((ksteps + sp - 1) / sp) * BK;
            sk.slabs = reinterpret_cast<float *>(d->workspace);
        }
    }
    const EpiParams e = make_epi(d);
    const dim3 grid((unsigned)(ntile * sk.splits));
    [[maybe_unused]] const dim3 block(512);   // the eight-wave kernels (tools build)
    const bool f8 = d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2;
    if (f8) {
        // 8-bit K-tiles are 128 token rows deep: slice boundaries on multiples of 128
        if (sk.splits > 1) sk.k_per_split = (((d->K + 127) / 128 + sk.splits - 1) / sk.splits) * 128;
        const dim3 block4(256);
#ifdef ECGVIT_TOOLS   // A/B against the eight-wave kernels (tools/wgrad_ab.py, ECGVIT_WGRAD_8W=1): the shipped library does not carry them
        static const int env8 = [] { const char *e_ = getenv("ECGVIT_WGRAD_8W"); return e_ ? atoi(e_) : 0; }();
        if (env8 || g_tools_wgrad_8w) {
            if (d->dtype == ECGVIT_BF8_E5M2) hipLaunchKernelGGL(gemm_wgrad8_kernel<1>, grid, block, 0, s, *d, e, sk, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_wgrad8_kernel<0>, grid, block, 0, s, *d, e, sk, tiles_m, tiles_n);
        } else
#endif
        if (d->dtype == ECGVIT_BF8_E5M2) hipLaunchKernelGGL(gemm_wgrad8_kernel_4w<1>, grid, block4, 0, s, *d, e, sk, tiles_m, tiles_n);
        else hipLaunchKernelGGL(gemm_wgrad8_kernel_4w<0>, grid, block4, 0, s, *d, e, sk, tiles_m, tiles_n);
    } else {
        const dim3 block4(256);
#ifdef ECGVIT_TOOLS
        static const int env8 = [] { const char *e_ = getenv("ECGVIT_WGRAD_8W"); return e_ ? atoi(e_) : 0; }();
        if (env8 || g_tools_wgrad_8w) {
            if (d->out_dtype == ECGVIT_BF16) hipLaunchKernelGGL(gemm_wgrad_kernel<bf16_t>, grid, block, 0, s, *d, e, sk, tiles_m, tiles_n);
            else hipLaunchKernelGGL(gemm_wgrad_kernel<float>, grid, block, 0, s, *d, e, sk, tiles_m, tiles_n);
        } else
#endif
        if (d->out_dtype == ECGVIT_BF16) hipLaunchKernelGGL(gemm_wgrad_kernel_4w<bf16_t>, grid, block4, 0, s, *d, e, sk, tiles_m, tiles_n);
        else hipLaunchKernelGGL(gemm_wgrad_kernel_4w<float>, grid, block4, 0, s, *d, e, sk, tiles_m, tiles_n);
    }
    ECGVIT_CHECK_LAUNCH();
    if (sk.splits > 1) {
        const int64_t MN = (int64_t)d->M * d->N;
        const int g = (int)std::min<int64_t>((MN / 4 + 255) / 256, 2048);
        if (d->out_dtype == ECGVIT_BF16) hipLaunchKernelGGL(splitk_reduce2_kernel<bf16_t>, dim3(g), dim3(256), 0, s, sk.slabs, sk.splits, MN, d->N, (bf16_t *)d->C, d->ldc, e);
        else hipLaunchKernelGGL(splitk_reduce2_kernel<float>, dim3(g), dim3(256), 0, s, sk.slabs, sk.splits, MN, d->N, (float *)d->C, d->ldc, e,
                                f8 ? d->scale_a : nullptr, f8 ? d->scale_b : nullptr);
        ECGVIT_CHECK_LAUNCH();
    }
    return ECGVIT_OK;
}
