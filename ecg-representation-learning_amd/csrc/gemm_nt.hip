// C = epilogue(A . B^T) for the large Linear shapes of the train step (forward products and, against the transposed bf16 weight
// shadows, the input-gradient products): M x N x K with both operands K-contiguous, K % 64 == 0, N % 8 == 0.
//
// gemm_nt_kernel ("R"): persistent, 256 x 256 x 64 block tile, 8 waves (2 x 4, wave tile 128 x 64), v_mfma_f32_16x16x32_bf16.
//   * main loop: four quadrants per K-tile (one 64 x 32 output quadrant each: 12 / 4 / 8 / 0 ds_read_b128, nothing read twice);
//     operands arrive as 16-KiB half-tiles by LDS-DMA (`buffer_load ... lds`) from a stream that runs CONTINUOUSLY across output
//     tiles -- activations two K-tiles ahead in a 3-slot ring per half, weights one K-tile ahead in 2 slots (10 x 16 KiB = all of
//     LDS), one counted vmcnt(4) and ONE workgroup barrier per K-tile, never a drain; waves 4-7 pass that barrier one quadrant
//     later in their program than waves 0-3 (before their register-only quadrant 4), so each SIMD's two waves alternate between
//     reading fragments and multiplying without any barrier inside the K-tile.
//   * the MFMA takes the WEIGHT fragment as its first operand and the activation fragment as its second, so an accumulator
//     register quad holds 4 consecutive output COLUMNS of one row (D row = 4*(lane>>4) + reg <-> n, D col = lane&15 <-> m).  The
//     weight fragment of n-tile j reads image rows 32*(j>>1) + 8*(i>>2) + 4*(j&1) + (i&3) (i = lane&15), which makes the 16
//     values a lane holds for one output row two runs of 8 consecutive columns (8q .. 8q+7 and 32+8q .. 32+8q+7, q = lane>>4):
//     the epilogue runs straight out of the accumulators -- bias / GELU / dropout / residual / x-aux / column sums in f32, two
//     16-B runs per lane and row, each moved across the lanes once (ds_bpermute) so that a quad stores a 64-B half-line -- with
//     no LDS round trip and no patch buffer.
//     The kernel it replaces (gemm_bf16_q_kernel) drained through per-wave LDS patches: 256 KiB of ds_write_b32 per tile at
//     64 B/clk plus the read-back, inside a 10.4k-cycle drain per tile (20 % of a K = 768 tile).
//   * LDS images (the DMA destination is wave-uniform base + lane*16, so swizzles go on the per-lane SOURCE address):
//     activations [128 rows][64 k], 16-B chunk ^= (row>>1)&7; weights chunk ^= (((row>>3)&3)<<1) | ((row>>1)&1) -- both
//     conflict-free for their ds_read_b128 fragment patterns (rows 16t + (lane&15) / the permuted rows above).
//   * epilogue flags are a template parameter for the combinations the train step uses (branch-free bodies); any other
//     combination runs the same kernel with run-time flags.
#include "common.h"
#include "nt_tiles.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 16384;
constexpr int LDS_BYTES = 163840;

typedef __attribute__((address_space(3))) void *lptr_t;
typedef long i64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

using nt_tiles::xcd_remap;
using nt_tiles::decode_tile;

#define NT_HAS(f) (((FL >= 0) ? FL : e.flags) & (f))
// PRICING EXPERIMENT (round 6, tools build only; a template flag, never a descriptor flag): LayerNorm folded into its consumer product --
// LN(x) . W^T = rstd[m] (x . (gamma o W)^T)[m, n] - (rstd mu)[m] ((gamma o W) . 1)[n] + (W . beta)[n], i.e. per output element two FMAs on a row pair
// (a[m], b[m]) and a column vector g[n] (the second column vector rides on the bias).  tools/gemm_ab.py --rowaffine times the QKV forward and the FFN-up
// forward with it against what ships; nothing in the product instantiates it.  profiles/r06_ln_fold_pricing.txt
#define ECGVIT_EPI_ROWAFFINE_X 4096
#ifdef ECGVIT_TOOLS
__device__ const float *g_ra_a = nullptr, *g_ra_b = nullptr, *g_ra_g = nullptr;   // row vectors [M], column vector [N] (ecgvit_tools_rowaffine)
#endif

// two f32 -> one dword of two bf16 (RNE, NaN-safe); written out because hipcc otherwise pairs the converts of an 8-element
// run across odd register boundaries (5 converts + 4 v_perm/v_alignbit per 16-B store instead of 4 converts)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }

// one run of 8 consecutive columns of one output row: v = alpha*acc -> bias -> GELU (+aux) -> dropout -> GELU' / x-aux -> residual
// -> accumulate -> store; same order and rounding points as gemm_bf16.hip's epilogue_value.  All global accesses are buffer
// operations: `off` is the byte offset of (m, n) in C scaled per buffer by the caller, 0x80000000 for a masked lane -- beyond
// every descriptor's num_records, so loads return zero and stores are dropped, and the body needs no exec-mask branch.
// (m, ncol, ok) are the run's coordinates in the ACCUMULATOR layout -- what the arithmetic, the dropout hash, the column sums use;
// (mm, ncm, okm) those of the run this lane moves to or from MEMORY for 16-bit buffers: the vector-memory path merges only
// ADJACENT lanes into one request, and in the accumulator layout adjacent lanes are adjacent ROWS (64 requests of 16 B per wave
// instruction: 27 ns per instruction and CU, loads and stores alike, `tools/ta_rate.hip`), so every 16-B run crosses the lanes
// once (4 ds_bpermute_b32) to a layout where the four lanes of a quad hold the four runs of one 64-B half-line (7-8 ns).
struct NtBufs {
    __amdgpu_buffer_rsrc_t c, res, aux, q8;
    int ldc2, ldr2, ldx2, ldq;   // row pitches in bytes
    float q8_inv;                // 1 / scale of the 8-bit output copy (ECGVIT_EPI_QUANT_OUT)
    int q8_bf8;                  // its format: 0 = e4m3, 1 = e5m2
    int t_out, t_in;             // ds_bpermute byte addresses: accumulator layout -> memory layout and back (see nt_epilogue)
};
constexpr uint32_t NT_OOB = 0x80000000u;

// the same lane permutation on the four dwords of a 16-B run (LDS crossbar, no LDS memory)
__device__ __forceinline__ u32x4 nt_permute(const u32x4 &x, int addr) {
    u32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)x[k]);
    return y;
}

// what follows the store of one run: column sums and the 8-bit copy, both of the values AS STORED (`out` = the packed bf16 run in the
// accumulator layout)
template <int FL>
__device__ __forceinline__ void nt_epi8_tail(const u32x4 &out, bool ok, bool okm, uint32_t mm, uint32_t ncm, const NtBufs &bf, float *cs8, float &qmax,
                                             int rt_flags = 0) {
    struct { int flags; } e{rt_flags};
    if (NT_HAS(ECGVIT_EPI_COLSUM)) {   // of the values as stored; masked lanes add nothing
#pragma unroll
        for (int k = 0; k < 4; ++k) { cs8[2 * k] += ok ? bf16_lo(out[k]) : 0.f; cs8[2 * k + 1] += ok ? bf16_hi(out[k]) : 0.f; }
    }
    if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) {   // the 8-bit copy of the values as stored (what a separate quantise pass would read back)
        float f[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { f[2 * k] = bf16_lo(out[k]); f[2 * k + 1] = bf16_hi(out[k]); }
        if (ok) {
#pragma unroll
            for (int k = 0; k < 8; ++k) qmax = fmaxf(qmax, fabsf(f[k]));
        }
        const float mx = bf.q8_bf8 ? 57344.f : 448.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = __builtin_amdgcn_fmed3f(f[k] * bf.q8_inv, -mx, mx);
        int w0 = 0, w1 = 0;
        if (bf.q8_bf8) {
            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
        } else {
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        }
        u32x2 q;
        q[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, w0); q[1] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, w1);
        __builtin_amdgcn_raw_buffer_store_b64(q, bf.q8, okm ? mm * (uint32_t)bf.ldq + ncm : NT_OOB, 0, 0);
    }
}

template <typename TO, int FL, int CAUX = 0>
__device__ __forceinline__ void nt_epi8(float (&v)[8], const float *bias8, uint32_t m, uint32_t ncol, bool ok, uint32_t mm, uint32_t ncm, bool okm,
                                        const NtBufs &bf, const EpiParams &e, const u32x4 &res, const u32x4 &auxin, float *cs8, float &qmax) {
    // light bodies: alpha was applied to the accumulators by nt_epilogue, behind ONE branch (unrolled, the compiler turns this one into
    // a select per element).  The rolled heavy bodies keep it here, where it is a real branch
    constexpr bool kLightBody = FL >= 0 && !(FL & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD));
    if constexpr (!kLightBody) {
        if (e.alpha != 1.f) {   // wave-uniform
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= e.alpha;
        }
    }
    if (NT_HAS(ECGVIT_EPI_BIAS) && !(kLightBody && (FL & ECGVIT_EPI_ROWAFFINE_X))) {   // (the light row-affine pricing body has folded the bias into its second FMA)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bias8[k];
    }
    // FFN-up forward (bias + erf-GELU + saved GELU' x mask [+ dropout]): the launch whose epilogue is VALU-bound.  The dropout rescale is
    // folded into the GELU constants and the mask is applied to the PACKED results (keep masks of the quad form: one hash per four elements,
    // one AND per result dword); the keep set is dropout_mask8's bit for bit
    constexpr bool kUp = sizeof(TO) == 2 && FL >= 0 && (FL & ECGVIT_EPI_GELU) && (FL & ECGVIT_EPI_GELU_GRAD_AUX) &&
                         !(FL & (ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_RESIDUAL | ECGVIT_EPI_ACCUM));
    if constexpr (kUp) {
        constexpr bool kDrop = (FL & ECGVIT_EPI_DROPOUT) != 0;
        const float kk = kDrop ? e.inv_keep : 1.0f;
        const float hk = ECGVIT_GELU_HK1 * kk, ck = ECGVIT_GELU_CK1 * kk;   // kk == 1: the very constants every other kernel uses
        float dy[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // two elements per vector instruction (common.h), bit for bit gelu_fast_both_scaled
            f32x2 y2, d2;
            gelu_fast_both_scaled_x2(f32x2{v[2 * k], v[2 * k + 1]}, hk, ck, y2, d2);
            v[2 * k] = y2[0]; v[2 * k + 1] = y2[1]; dy[2 * k] = d2[0]; dy[2 * k + 1] = d2[1];
        }
        constexpr bool kAux8 = (FL & ECGVIT_EPI_AUX8) != 0;
        u32x4 sav = {0u, 0u, 0u, 0u}, out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (!kAux8) sav[k] = pack_bf16x2(dy[2 * k], dy[2 * k + 1]);
            out[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        }
        [[maybe_unused]] uint32_t kmk[4] = {0u, 0u, 0u, 0u};
        if constexpr (kDrop) {
            keepmask8(e.seed, m * (uint32_t)e.N + ncol, e.drop_thresh, kmk);
#pragma unroll
            for (int k = 0; k < 4; ++k) { if constexpr (!kAux8) sav[k] &= kmk[k]; out[k] &= kmk[k]; }
        }
        if constexpr ((FL & ECGVIT_EPI_AUX8) != 0) {
            // the saved tensor as e4m3 bytes (from the UNROUNDED f32 values): 8 B per run -- half the stream, two crossbar moves instead of four
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(dy[0], dy[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(dy[2], dy[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(dy[4], dy[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(dy[6], dy[7], w1, true);
            if constexpr (kDrop) {   // byte masks from the pairs' halfword masks (sav already carries them: recover from the masked bf16 pairs' keep masks)
                w0 &= (int)__builtin_amdgcn_perm(kmk[1], kmk[0], 0x06040200u);
                w1 &= (int)__builtin_amdgcn_perm(kmk[3], kmk[2], 0x06040200u);
            }
            u32x2 q;
            q[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, w0); q[1] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, w1);
            __builtin_amdgcn_raw_buffer_store_b64(q, bf.aux, okm ? mm * (uint32_t)bf.ldx2 + ncm : NT_OOB, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(nt_permute(sav, bf.t_out), bf.aux, okm ? mm * (uint32_t)bf.ldx2 + ncm * 2 : NT_OOB, 0, 0);
        }
        if constexpr (!(FL & ECGVIT_EPI_NO_OUT))   // (NO_OUT: the consumers read the 8-bit copy only)
            __builtin_amdgcn_raw_buffer_store_b128(nt_permute(out, bf.t_out), bf.c, okm ? mm * (uint32_t)bf.ldc2 + ncm * 2 : NT_OOB, 0, CAUX);
        nt_epi8_tail<FL>(out, ok, okm, mm, ncm, bf, cs8, qmax);
        return;
    }
    if constexpr (sizeof(TO) == 2) {
        float mult[8];
        const bool drop = NT_HAS(ECGVIT_EPI_DROPOUT);
        if (drop) dropout_mask8(e.seed, m * (uint32_t)e.N + ncol, e.drop_thresh, e.inv_keep, mult);
        if (NT_HAS(ECGVIT_EPI_GELU)) {
            u32x4 sav;
            if (NT_HAS(ECGVIT_EPI_GELU_GRAD_AUX)) {
                float dy[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    gelu_fast_both(v[k], v[k], dy[k]);
                    if (drop) dy[k] *= mult[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) sav[k] = pack_bf16x2(dy[2 * k], dy[2 * k + 1]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) sav[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] = gelu_fast(bf16_lo(sav[k])); v[2 * k + 1] = gelu_fast(bf16_hi(sav[k])); }
            }
            __builtin_amdgcn_raw_buffer_store_b128(nt_permute(sav, bf.t_out), bf.aux, okm ? mm * (uint32_t)bf.ldx2 + ncm * 2 : NT_OOB, 0, 0);
        }
        if (drop) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= mult[k];
        }
        if (NT_HAS(ECGVIT_EPI_GELU_BWD)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] *= gelu_fast_grad(bf16_lo(auxin[k])); v[2 * k + 1] *= gelu_fast_grad(bf16_hi(auxin[k])); }
        }
        if (NT_HAS(ECGVIT_EPI_MUL_AUX)) {
            if constexpr (FL >= 0 && (FL & ECGVIT_EPI_AUX8)) {   // e4m3 bytes: auxin[0], auxin[1] hold the run's 8 values
                const int a0 = (int)auxin[0], a1 = (int)auxin[1];
                v[0] *= __builtin_amdgcn_cvt_f32_fp8(a0, 0); v[1] *= __builtin_amdgcn_cvt_f32_fp8(a0, 1);
                v[2] *= __builtin_amdgcn_cvt_f32_fp8(a0, 2); v[3] *= __builtin_amdgcn_cvt_f32_fp8(a0, 3);
                v[4] *= __builtin_amdgcn_cvt_f32_fp8(a1, 0); v[5] *= __builtin_amdgcn_cvt_f32_fp8(a1, 1);
                v[6] *= __builtin_amdgcn_cvt_f32_fp8(a1, 2); v[7] *= __builtin_amdgcn_cvt_f32_fp8(a1, 3);
                // (two bytes per conversion + packed multiplies, and the column sums as one select per run + packed fmas -- a third of this body's vector
                // instructions -- were measured on top of the early row request: bit-identical, no change of the launch or the step: not kept)
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] *= bf16_lo(auxin[k]); v[2 * k + 1] *= bf16_hi(auxin[k]); }
            }
        }
        if (NT_HAS(ECGVIT_EPI_RESIDUAL)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_lo(res[k]); v[2 * k + 1] += bf16_hi(res[k]); }
        }
        const uint32_t off = okm ? mm * (uint32_t)bf.ldc2 + ncm * 2 : NT_OOB;
        if (NT_HAS(ECGVIT_EPI_ACCUM)) {
            const u32x4 old = nt_permute(__builtin_amdgcn_raw_buffer_load_b128(bf.c, off, 0, 0), bf.t_in);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += bf16_lo(old[k]); v[2 * k + 1] += bf16_hi(old[k]); }
        }
        u32x4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        if constexpr (!(FL >= 0 && (FL & ECGVIT_EPI_NO_OUT)))
            __builtin_amdgcn_raw_buffer_store_b128(nt_permute(out, bf.t_out), bf.c, off, 0, CAUX);   // CAUX: cache policy of the output stores (2 = nt)
        nt_epi8_tail<FL>(out, ok, okm, mm, ncm, bf, cs8, qmax, e.flags);
    } else {   // f32 outputs: 32 B per lane stay in the accumulator layout (no train-step launch takes this branch)
        const uint32_t off = ok ? m * (uint32_t)bf.ldc2 + ncol * 4 : NT_OOB;
        if (NT_HAS(ECGVIT_EPI_ACCUM)) {
            const u32x4 o0 = __builtin_amdgcn_raw_buffer_load_b128(bf.c, off, 0, 0), o1 = __builtin_amdgcn_raw_buffer_load_b128(bf.c, off, 16, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] += __builtin_bit_cast(float, o0[k]); v[4 + k] += __builtin_bit_cast(float, o1[k]); }
        }
        u32x4 w0, w1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { w0[k] = __builtin_bit_cast(uint32_t, v[k]); w1[k] = __builtin_bit_cast(uint32_t, v[4 + k]); }
        __builtin_amdgcn_raw_buffer_store_b128(w0, bf.c, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(w1, bf.c, off, 16, 0);
    }
}

// Drain one wave's 128 x 64 accumulator block straight from registers.  acc[i][j][r] = C[row 16i + (lane&15)]
// [col 32*(j>>1) + 8*(lane>>4) + 4*(j&1) + r] of the wave tile.
// `issue_next` puts the next tile's first operand pieces in flight (the kernel's prefetch, see the call site).  It runs AFTER the epilogue's
// own up-front loads (bias, residual / aux rows) and BEFORE its first store: vmcnt retires in issue order, so pieces issued ahead of
// those loads would have to land (an HBM round trip, with the matrix pipe idle) before the first row of the epilogue could start,
// and pieces issued behind the stores would hold the next main loop until the stores are acknowledged.
// ROLL: run a light body through the rolled loop as well (row loads one step ahead instead of all 16 up front): the four-wave kernel's
// x-aux body -- with 256 accumulators live, 64 registers of aux rows on top of them are spilled
// the saved-tensor rows of one wave tile as e4m3 bytes (ECGVIT_EPI_AUX8): 16 runs x 8 B in the memory layout -- what ld_aux of nt_epilogue loads.
// gemm_nt_kernel requests them at the START of the tile's main loop (round 5: 32 registers -- the bf16 form's 64 did not fit) instead of in front
// of the epilogue's first row, where every tile paid their HBM round trip with the matrix pipe idle.
__device__ __forceinline__ void nt_aux8_request(u32x2 (&x)[16], const NtBufs &bf, int N, int m0, int n0, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t mrowm = (uint32_t)(m0 + wm * 128 + (lane >> 2));
    const int nbm = n0 + wn * 64 + 8 * (lane & 3);
    const bool nokm0 = nbm < N, nokm1 = nbm + 32 < N;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
            x[2 * i + h] = __builtin_amdgcn_raw_buffer_load_b64(bf.aux, (h ? nokm1 : nokm0) ? (mrowm + 16 * i) * (uint32_t)bf.ldx2 + (nbm + 32 * h) : NT_OOB, 0, 0);
}

template <typename TO, int FL, int CAUX = 0, bool ROLL = false, typename IssueNext>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[8][4], const ecgvit_gemm_desc &d, const EpiParams &e, const NtBufs &bf, int m0, int n0,
                                            int wave, int lane, IssueNext &&issue_next, const u32x2 *auxpre = nullptr) {
    const int wm = wave >> 2, wn = wave & 3;
    const int c = lane & 15, q = lane >> 4;
    const int M = d.M, N = d.N;
    const int nb = n0 + wn * 64 + 8 * q;
    const bool nok0 = nb < N, nok1 = nb + 32 < N;
    const int nl0 = nok0 ? nb : 0, nl1 = nok1 ? nb + 32 : 0;   // in-range columns for the bias loads of masked lanes
    float bias[16];
    if (NT_HAS(ECGVIT_EPI_BIAS)) {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(e.bias + nl0), b1 = *reinterpret_cast<const f32x4 *>(e.bias + nl0 + 4);
        const f32x4 b2 = *reinterpret_cast<const f32x4 *>(e.bias + nl1), b3 = *reinterpret_cast<const f32x4 *>(e.bias + nl1 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { bias[k] = b0[k]; bias[4 + k] = b1[k]; bias[8 + k] = b2[k]; bias[12 + k] = b3[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) bias[k] = 0.f;
    }
    constexpr bool kRA = FL >= 0 && (FL & ECGVIT_EPI_ROWAFFINE_X) != 0;
    [[maybe_unused]] float rag[16];
    [[maybe_unused]] const float *ra_a = nullptr, *ra_b = nullptr;
#ifdef ECGVIT_TOOLS
    if constexpr (kRA) {
        ra_a = g_ra_a; ra_b = g_ra_b;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g_ra_g + nl0), g1 = *reinterpret_cast<const f32x4 *>(g_ra_g + nl0 + 4);
        const f32x4 g2 = *reinterpret_cast<const f32x4 *>(g_ra_g + nl1), g3 = *reinterpret_cast<const f32x4 *>(g_ra_g + nl1 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { rag[k] = g0[k]; rag[4 + k] = g1[k]; rag[8 + k] = g2[k]; rag[12 + k] = g3[k]; }
    }
#endif
    float cs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) cs[k] = 0.f;
    float qmax = 0.f;
    unsigned int amax_seen = 0u;
    if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) amax_seen = amax_peek(d.q8_amax);   // with the epilogue's up-front loads; consumed behind its last store
    if (FL >= 0 && !(FL & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD)) && e.alpha != 1.f) {
        // light bodies: a REAL wave-uniform branch (the empty asm keeps the compiler from if-converting it): as a select per element it cost every
        // launch with a light body 1.5 VALU instructions per element, alpha == 1 included -- a third of the plain epilogue
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= e.alpha;
    }
    const uint32_t mrow = (uint32_t)(m0 + wm * 128 + c);
    // memory layout of a 16-row x 32-column half block: lane t moves row t>>2, run t&3 (accumulator layout: row s&15, run s>>4)
    const uint32_t mrowm = (uint32_t)(m0 + wm * 128 + (lane >> 2));
    const int nbm = n0 + wn * 64 + 8 * (lane & 3);
    const bool nokm0 = nbm < N, nokm1 = nbm + 32 < N;
    constexpr bool kBf = sizeof(TO) == 2;
    constexpr bool kLight = FL >= 0 && !(FL & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD)) && !ROLL;
    const bool want_res = kBf && NT_HAS(ECGVIT_EPI_RESIDUAL), want_aux = kBf && NT_HAS(ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_GELU_BWD);
    // rows >= M fall beyond num_records by themselves; masked columns are forced there
    auto ld_res = [&](int i, int h) {
        return want_res ? __builtin_amdgcn_raw_buffer_load_b128(bf.res, (h ? nokm1 : nokm0) ? (mrowm + 16 * i) * (uint32_t)bf.ldr2 + (nbm + 32 * h) * 2 : NT_OOB, 0, 0) : u32x4{};
    };
    constexpr bool kAux8 = FL >= 0 && (FL & ECGVIT_EPI_AUX8) != 0;
    auto ld_aux = [&](int i, int h) {
        if constexpr (kAux8) {   // e4m3 bytes: 8 B per run
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(bf.aux, (h ? nokm1 : nokm0) ? (mrowm + 16 * i) * (uint32_t)bf.ldx2 + (nbm + 32 * h) : NT_OOB, 0, 0);
            return u32x4{w[0], w[1], 0u, 0u};
        } else {
            return want_aux ? __builtin_amdgcn_raw_buffer_load_b128(bf.aux, (h ? nokm1 : nokm0) ? (mrowm + 16 * i) * (uint32_t)bf.ldx2 + (nbm + 32 * h) * 2 : NT_OOB, 0, 0) : u32x4{};
        }
    };
    // loaded runs arrive in the memory layout; to the accumulator layout when they are consumed
    auto to_acc = [&](const u32x4 &x, bool want) { return want ? nt_permute(x, bf.t_in) : x; };
    auto to_acc_aux = [&](const u32x4 &x, bool want) {
        if constexpr (kAux8) {
            if (!want) return x;
            return u32x4{(uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_in, (int)x[0]), (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_in, (int)x[1]), 0u, 0u};
        } else {
            return want ? nt_permute(x, bf.t_in) : x;
        }
    };
    if constexpr (kLight) {
        // light bodies: every row load of the tile is issued up front (the 64 fragment registers are free now), then the 8 row steps
        // run fully unrolled on the accumulators in place; stores are fire-and-forget
        u32x4 R[8][2], X[8][2];
        [[maybe_unused]] float RA[8], RB[8];
        if constexpr (kRA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int mc = min((int)mrow + 16 * i, M - 1); RA[i] = ra_a[mc]; RB[i] = ra_b[mc]; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                R[i][h] = ld_res(i, h);
                if (auxpre) X[i][h] = u32x4{auxpre[2 * i + h][0], auxpre[2 * i + h][1], 0u, 0u};   // (requested with the tile's first K-tile: nt_aux8_request)
                else X[i][h] = ld_aux(i, h);
            }
        if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) asm volatile("" : "+v"(amax_seen));   // (waited for here, with the row loads: not behind the pieces)
        issue_next();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v0[8], v1[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = acc[i][0][r]; v0[4 + r] = acc[i][1][r]; v1[r] = acc[i][2][r]; v1[4 + r] = acc[i][3][r]; }
            if constexpr (kRA) {   // two FMAs per element: v rstd[m] + ((rstd mu)[m] g[n] + c[n]) (c = the bias vector)
#pragma unroll
                for (int k = 0; k < 8; ++k) { v0[k] = fmaf(v0[k], RA[i], fmaf(RB[i], rag[k], bias[k])); v1[k] = fmaf(v1[k], RA[i], fmaf(RB[i], rag[8 + k], bias[8 + k])); }
            }
            const uint32_t m = mrow + 16 * i, mm = mrowm + 16 * i;
            const bool mok = (int)m < M, mokm = (int)mm < M;
            nt_epi8<TO, FL, CAUX>(v0, bias, m, nb, mok && nok0, mm, nbm, mokm && nokm0, bf, e, to_acc(R[i][0], want_res), to_acc_aux(X[i][0], want_aux), cs, qmax);
            nt_epi8<TO, FL, CAUX>(v1, bias + 8, m, nb + 32, mok && nok1, mm, nbm + 32, mokm && nokm1, bf, e, to_acc(R[i][1], want_res), to_acc_aux(X[i][1], want_aux),
                            cs + 8, qmax);
        }
    } else {
        // heavy bodies must exist ONCE in the instruction stream (I-cache): rolled loop, only the accumulator pick is a switch;
        // row loads one step ahead
        u32x4 nr0 = ld_res(0, 0), nr1 = ld_res(0, 1), na0 = ld_aux(0, 0), na1 = ld_aux(0, 1);
        [[maybe_unused]] float nra = 0.f, nrb = 0.f;
        if constexpr (kRA) { const int mc = min((int)mrow, M - 1); nra = ra_a[mc]; nrb = ra_b[mc]; }
        // hipcc's wait-count model does not see LDS-DMA: behind the pieces it would wait `vmcnt(0)` for the bias (all of it is needed by
        // the first row), i.e. for the pieces too.  Consume the bias here, while only the epilogue's own loads are in flight
        if (NT_HAS(ECGVIT_EPI_BIAS)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(bias[k]));
        }
        if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) asm volatile("" : "+v"(amax_seen));
        issue_next();
#pragma unroll 1
        for (int i = 0; i < 8; ++i) {
            const u32x4 r0 = nr0, r1 = nr1, a0 = na0, a1 = na1;
            [[maybe_unused]] const float cra = nra, crb = nrb;
            if (i < 7) { nr0 = ld_res(i + 1, 0); nr1 = ld_res(i + 1, 1); na0 = ld_aux(i + 1, 0); na1 = ld_aux(i + 1, 1); }
            if constexpr (kRA) { if (i < 7) { const int mc = min((int)mrow + 16 * (i + 1), M - 1); nra = ra_a[mc]; nrb = ra_b[mc]; } }
            float v0[8], v1[8];
#define NT_PICK(I)                                                                                       \
    case I:                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                  \
            v0[r] = acc[I][0][r]; v0[4 + r] = acc[I][1][r]; v1[r] = acc[I][2][r]; v1[4 + r] = acc[I][3][r]; \
        }                                                                                                \
        break;
            switch (i) { NT_PICK(0) NT_PICK(1) NT_PICK(2) NT_PICK(3) NT_PICK(4) NT_PICK(5) NT_PICK(6) default: NT_PICK(7) }
#undef NT_PICK
            if constexpr (kRA) {   // (the bias add stays where it is in the heavy body: here fma + multiply = the same two instructions per element)
#pragma unroll
                for (int k = 0; k < 8; ++k) { v0[k] = fmaf(v0[k], cra, crb * rag[k]); v1[k] = fmaf(v1[k], cra, crb * rag[8 + k]); }
            }
            const uint32_t m = mrow + 16 * i, mm = mrowm + 16 * i;
            const bool mok = (int)m < M, mokm = (int)mm < M;
            nt_epi8<TO, FL, CAUX>(v0, bias, m, nb, mok && nok0, mm, nbm, mokm && nokm0, bf, e, to_acc(r0, want_res), to_acc_aux(a0, want_aux), cs, qmax);
            nt_epi8<TO, FL, CAUX>(v1, bias + 8, m, nb + 32, mok && nok1, mm, nbm + 32, mokm && nokm1, bf, e, to_acc(r1, want_res), to_acc_aux(a1, want_aux), cs + 8, qmax);
        }
    }
    if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) {   // at most one atomic max per wave and tile, only when it raises the slot (common.h: wave_amax_publish)
        wave_amax_publish(d.q8_amax, qmax, amax_seen);
    }
    if (NT_HAS(ECGVIT_EPI_COLSUM)) {
        // lanes with equal (lane >> 4) hold the same 16 columns: fold the 16 rows, one partial row per (tile row, wm)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cs[k] += __shfl_xor(cs[k], 1, 64);
            cs[k] += __shfl_xor(cs[k], 2, 64);
            cs[k] += __shfl_xor(cs[k], 4, 64);
            cs[k] += __shfl_xor(cs[k], 8, 64);
        }
        if (c == 0) {
            float *pr = reinterpret_cast<float *>(d.workspace) + ((int64_t)(m0 / BM) * 2 + wm) * N;
            if (nok0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) pr[nb + k] = cs[k];
            }
            if (nok1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) pr[nb + 32 + k] = cs[8 + k];
            }
        }
    }
}

#ifdef ECGVIT_TOOLS
// diagnostics (tools build only): per block {s_memtime, s_memrealtime} at start and end, cycles summed over main loops and epilogues
__device__ unsigned long long g_nt_stamps[256 * 8];
// XCD rendezvous experiment (ablate bit 16): one arrival counter per XCD (own cache line), zeroed by the launcher
__device__ unsigned int g_nt_rdv[8 * 32];
#define NT_STAMP_T() (STAMP ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define NT_STAMP_T() 0ull
#endif

template <typename TO, int FL, bool STAMP = false, int OPS = 0, int CAUX = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(ecgvit_gemm_desc d, EpiParams e, int tiles_m, int tiles_n, int ngroup, int nitems, int ablate) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    const int M = d.M, N = d.N;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool late = wm == 1;
    // OPS: 0 = bf16 operands; 1 / 3 = e4m3 x e4m3, 2 / 4 = e4m3 weights x e5m2 activations (input gradients), on the plain fp8 MFMA
    // (bf16 rate) / on the block-scaled MFMA with unit scales (twice the rate: what ships).  An 8-bit K-tile is 128 deep:
    // the same 128-B image rows, DMA pieces and fragment reads, twice the MFMAs per byte that crosses the CU's memory path
    constexpr int ES = OPS ? 1 : 2;
    const int nk = d.K / (128 / ES);
    const int lda2 = (int)d.lda * ES, ldb2 = (int)d.ldb * ES;   // row pitches in bytes
    if (d.scale_a) e.alpha *= *d.scale_a;                     // per-tensor scales of 8-bit operands (device scalars)
    if (d.scale_b) e.alpha *= *d.scale_b;
    const int ntile = tiles_m * tiles_n;

    int it = blockIdx.x;
    if (it >= nitems) return;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, (uint32_t)((int64_t)M * lda2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.B), 0, (uint32_t)((int64_t)N * ldb2), 0x00020000);
    NtBufs bf;
    bf.ldc2 = (int)d.ldc * (int)sizeof(TO); bf.ldr2 = (int)e.ldr * 2; bf.ldx2 = (int)e.ldaux * ((FL >= 0 && (FL & ECGVIT_EPI_AUX8)) ? 1 : 2);   // (AUX8: the saved tensor is bytes)
    bf.c = __builtin_amdgcn_make_buffer_rsrc(d.C, 0, ((STAMP && (ablate & 1)) || !d.C) ? 0u : (uint32_t)((int64_t)M * bf.ldc2), 0x00020000);   // ablate 1 (diagnostics): stores dropped
    bf.res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(e.residual), 0, e.residual ? (uint32_t)((int64_t)M * bf.ldr2) : 0u, 0x00020000);
    bf.aux = __builtin_amdgcn_make_buffer_rsrc(e.aux, 0, e.aux ? (uint32_t)((int64_t)M * bf.ldx2) : 0u, 0x00020000);
    bf.ldq = (int)d.ldq8;
    bf.q8 = __builtin_amdgcn_make_buffer_rsrc(d.q8_out, 0, d.q8_out ? (uint32_t)((int64_t)M * bf.ldq) : 0u, 0x00020000);
    bf.q8_bf8 = d.q8_format == ECGVIT_BF8_E5M2;
    bf.t_out = ((lane >> 2) + 16 * (lane & 3)) << 2;   // memory-layout lane t takes its run from accumulator-layout lane (t>>2) + 16 (t&3)
    bf.t_in = (4 * (lane & 15) + (lane >> 4)) << 2;    // and back
    bf.q8_inv = 0.f;
    if (NT_HAS(ECGVIT_EPI_QUANT_OUT)) { const float qs = *d.q8_scale; bf.q8_inv = qs > 0.f ? 1.0f / qs : 0.f; }
    // this wave's two DMA pieces of a half-tile: rows 16*wave + {0..7}, {8..15}; LDS chunk p of row r holds source chunk p ^ f(r)
    const int r0 = 16 * wave + (lane >> 3), r1 = r0 + 8, p = lane & 7;
    const int voA0 = r0 * lda2 + ((p ^ ((r0 >> 1) & 7)) << 4), voA1 = r1 * lda2 + ((p ^ ((r1 >> 1) & 7)) << 4);
    const int fw0 = (((r0 >> 3) & 3) << 1) | ((r0 >> 1) & 1), fw1 = (((r1 >> 3) & 3) << 1) | ((r1 >> 1) & 1);
    const int voB0 = r0 * ldb2 + ((p ^ fw0) << 4), voB1 = r1 * ldb2 + ((p ^ fw1) << 4);
    // fragment read offsets inside a half-tile image; the second 32-deep k-step (s = 1) is ^ 64
    const int fr = lane & 15, fq = lane >> 4;
    const int loff = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);                 // activations: row 16t + fr, tile t adds 2048
    const int rl = 8 * (fr >> 2) + (fr & 3);
    const int boff = (wn & 1) * 8192 + rl * 128 + ((fq ^ (((fr >> 2) << 1) | ((fr >> 1) & 1))) << 4);   // weights: n-tile j adds 512*(j&1) + 4096*(j>>1)

#define R_DMA_A(h, ring, soff)                                                                                                   \
    do {                                                                                                                         \
        if (STAMP && (ablate & 2) && dma_off) break;                                                                             \
        char *dst_ = smem + (3 * (h) + (ring)) * HALF_BYTES + wave * 2048;                                                       \
        const int so_ = (soff) + (h) * 128 * lda2;                                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)dst_, 16, voA0, so_, 0, 0);                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dst_ + 1024), 16, voA1, so_, 0, 0);                               \
    } while (0)
#define R_DMA_B(h, ring, soff)                                                                                                   \
    do {                                                                                                                         \
        if (STAMP && (ablate & 2) && dma_off) break;                                                                             \
        char *dst_ = smem + (6 + 2 * (h) + (ring)) * HALF_BYTES + wave * 2048;                                                   \
        const int so_ = (soff) + (h) * 128 * ldb2;                                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)dst_, 16, voB0, so_, 0, 0);                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dst_ + 1024), 16, voB1, so_, 0, 0);                               \
    } while (0)
    // one fragment pair -> accumulator.  8-bit operands: the 16 fragment bytes are two 8-byte k-groups (k = 16c + 8t + j for chunk c, half
    // t: a permutation of the 128 k's of a row that both operands share), one MFMA each
#define R_MMA(ACC, WF, XF)                                                                                                       \
    do {                                                                                                                         \
        if constexpr (OPS == 0) {                                                                                                \
            ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF, XF, ACC, 0, 0, 0);                                                 \
        } else {                                                                                                                 \
            const i64x2 w_ = __builtin_bit_cast(i64x2, WF), x_ = __builtin_bit_cast(i64x2, XF);                                   \
            if constexpr (OPS == 1) {                                                                                            \
                ACC = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w_[0], x_[0], ACC, 0, 0, 0);                                    \
                ACC = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(w_[1], x_[1], ACC, 0, 0, 0);                                    \
            } else {                                                                                                             \
                ACC = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(w_[0], x_[0], ACC, 0, 0, 0);                                    \
                ACC = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(w_[1], x_[1], ACC, 0, 0, 0);                                    \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    // one 64 x 32 output quadrant of the wave tile x one K-tile.  OPS 3 / 4: the block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with every
    // block scale 2^0 (E8M0 0x7F) -- plain e4m3 x e4m3 / e5m2 products at twice the bf16 MFMA rate; a lane's 32 operand bytes are its
    // two 16-B fragment reads (chunks q and 4+q of the 128-B row: again one k-permutation shared by both operands)
#define R_QUAD(IO, JO, BF)                                                                                                       \
    do {                                                                                                                         \
        if constexpr (OPS <= 2) {                                                                                                \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i)                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) R_MMA(acc[IO + i][JO + j], BF[j][s], a[i][s]);                     \
        } else {                                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) {                        \
                const i32x4 w0_ = __builtin_bit_cast(i32x4, BF[j][0]), w1_ = __builtin_bit_cast(i32x4, BF[j][1]);                \
                const i32x4 x0_ = __builtin_bit_cast(i32x4, a[i][0]), x1_ = __builtin_bit_cast(i32x4, a[i][1]);                  \
                const i32x8 w_ = {w0_[0], w0_[1], w0_[2], w0_[3], w1_[0], w1_[1], w1_[2], w1_[3]};                               \
                const i32x8 x_ = {x0_[0], x0_[1], x0_[2], x0_[3], x1_[0], x1_[1], x1_[2], x1_[3]};                               \
                acc[IO + i][JO + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w_, x_, acc[IO + i][JO + j], 0, OPS == 4 ? 1 : 0, 0,   \
                                                                                       0x7F7F7F7F, 0, 0x7F7F7F7F);               \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    // a quadrant = its fragment reads (and up to two DMA pieces) and 16 MFMAs, scheduled by the compiler: its counted lgkmcnt waits
    // let a cluster's first MFMAs start before the last fragment has arrived, and it may hoist the next quadrant's reads into the
    // cluster.  Explicit `lgkmcnt(0)` + sched barriers + s_setprio around every cluster measured 1.5 % slower per launch inside the step
    // (profiles/r02_nt_compiler_scheduled.txt).  No workgroup barrier here: the K-tile's one barrier is R_BAR, fenced on both sides.
    // The 8-bit instantiations (half as many, longer MFMAs per cluster) keep the explicit form: it measured 1.3 % faster per launch there.
#define R_PHASE_SYNC_A()                                       \
    do {                                                       \
        if constexpr (OPS != 0) {                              \
            __builtin_amdgcn_sched_barrier(0);                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
            __builtin_amdgcn_sched_barrier(0);                 \
            __builtin_amdgcn_s_setprio(1);                     \
        }                                                      \
    } while (0)
#define R_PHASE_SYNC_B()                       \
    do {                                       \
        if constexpr (OPS != 0) {              \
            __builtin_amdgcn_s_setprio(0);     \
            __builtin_amdgcn_sched_barrier(0); \
        }                                      \
    } while (0)

    [[maybe_unused]] bool dma_off = false;   // diagnostics (stamped builds): ablate 2 = no DMA pieces after the prologue, 4 = no counted waits
    int cm0, cn0, nm0, nn0;
    decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
    nm0 = cm0; nn0 = cn0;
    // producer cursors: byte offset of (tile row, k) in the scalar offset; per-lane offsets never change
    int a_it = it, a_kt = 0, a_base = cm0 * lda2;
    int b_it = it, b_kt = 0, b_base = cn0 * ldb2;
    bool a_ok = true, b_ok = true;
#define R_ADV_A()                                                                                         \
    do {                                                                                                  \
        if (STAMP && (ablate & 8) && dma_off) break;                                                      \
        if (++a_kt == nk) {                                                                               \
            a_kt = 0;                                                                                     \
            a_it += (int)gridDim.x;                                                                       \
            if (a_it < nitems) { decode_tile(a_it, ntile, tiles_m, tiles_n, ngroup, nm0, nn0); a_base = nm0 * lda2; } \
            else a_ok = false;                                                                            \
        }                                                                                                 \
    } while (0)
#define R_ADV_B()                                          \
    do {                                                   \
        if (STAMP && (ablate & 8) && dma_off) break;       \
        if (++b_kt == nk) {                                \
            b_kt = 0;                                      \
            b_it += (int)gridDim.x;                        \
            if (b_it < nitems) b_base = nn0 * ldb2;        \
            else b_ok = false;                             \
        }                                                  \
    } while (0)

    // ---- prologue: A(0), B(0), A(1); the trailing group also the first half of B(1), which it otherwise issues in quadrant 4 of the
    // K-tile before (see the loop)
    R_DMA_A(0, 0, a_base); R_DMA_A(1, 0, a_base); R_ADV_A();
    R_DMA_B(0, 0, b_base); R_DMA_B(1, 0, b_base); R_ADV_B();
    R_DMA_A(0, 1, a_base + a_kt * (BK * 2)); R_DMA_A(1, 1, a_base + a_kt * (BK * 2)); R_ADV_A();
    if (late) {
        R_DMA_B(0, 1, b_base + b_kt * (BK * 2));
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    dma_off = true;
    int ga = 0, gb = 0;   // ring slots of the K-tile being multiplied
    bool pre = false;     // the coming K-tile's DMA pieces were issued ahead of the previous tile's epilogue
    // VMEM instructions the epilogue leaves in flight at least: its output stores (masked lanes still issue)
    constexpr int NST = (sizeof(TO) == 2 ? 16 : 32) + ((FL >= 0 && (FL & ECGVIT_EPI_GELU)) ? 16 : 0);
    // the x-aux input gradient with the e4m3 saved tensor (bf16 operands: the 8-bit instantiations have no 32 registers to spare): its 16 row loads
    // per wave tile go out ahead of the tile's first K-tile -- the youngest operations in flight at that K-tile's barrier, whose counted wait lets them
    // pass; the second K-tile's wait covers them: two K-tiles of flight time.  The same for the residual rows of the four-wave kernel (64 registers
    // per column half, or 32 for its first four row steps) spills in its main loop AND in its epilogue: not built in
    constexpr bool kAuxPre = OPS == 0 && sizeof(TO) == 2 && FL >= 0 && (FL & ECGVIT_EPI_MUL_AUX) && (FL & ECGVIT_EPI_AUX8) && !(FL & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD));
    constexpr int NAUXPRE = kAuxPre ? 16 : 0;
    [[maybe_unused]] unsigned long long st_t0 = 0, st_r0 = 0, st_main = 0, st_epi = 0, st_ntile = 0;
#ifdef ECGVIT_TOOLS
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    [[maybe_unused]] unsigned int rdv_cum = 0;
    [[maybe_unused]] int rdv_round = 0;
    for (;;) {
#ifdef ECGVIT_TOOLS
        if constexpr (STAMP) {
            if (ablate & 16) {
                // experiment: the workgroups of an XCD (blockIdx & 7) start every tile round together -- their K-tiles then
                // march through the shared activation / weight panels in step, and a panel slice fetched by one is an L2 hit for the others
                const int x = blockIdx.x & 7, left = nitems - rdv_round * (int)gridDim.x;   // items of this round: blocks b < left take one
                const int nb = left >= (int)gridDim.x ? (int)gridDim.x / 8 : (left > x ? (left - x + 7) / 8 : 0);
                rdv_cum += (unsigned int)nb;
                if (threadIdx.x == 0) {
                    atomicAdd(&g_nt_rdv[x * 32], 1u);
                    int spins = 0;
                    while (__hip_atomic_load(&g_nt_rdv[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rdv_cum && ++spins < 20000) __builtin_amdgcn_s_sleep(1);
                }
                ++rdv_round;
                __builtin_amdgcn_s_barrier();
            }
        }
#endif
        [[maybe_unused]] const unsigned long long st_a = NT_STAMP_T();
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        [[maybe_unused]] u32x2 auxpre[16];
        if constexpr (kAuxPre) nt_aux8_request(auxpre, bf, N, cm0, cn0, wave, lane);
        // the K loop exists twice, once per wave group: which group a wave belongs to never changes, and as a run-time condition it cost
        // six taken branches per K-tile
        auto kloop = [&](auto late_c) __attribute__((always_inline)) {
        constexpr bool LATE = decltype(late_c)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const int sa = (3 * wm + ga) * HALF_BYTES + loff;          // byte offsets into smem (kept integral: LDS address space)
            const int sb = (6 + 2 * (wn >> 1) + gb) * HALF_BYTES + boff;
            const int ga2 = ga == 0 ? 2 : ga - 1;   // (g + 2) % 3
            const int gb1 = gb ^ 1;
            bf16x8 a[4][2], b0[2][2], b1[2][2];
            const bool pre_k = pre;
            pre = false;
            const bool a_iss = a_ok && !pre_k;      // A(kt+2) goes out during this K-tile (both halves or neither: a_ok changes after the second)
            // ONE workgroup barrier per K-tile (R_BAR): the leading group (waves 0-3) passes it after quadrant 4, the trailing group
            // (waves 4-7, same SIMDs) before its quadrant 4 -- one quadrant behind, so that on every SIMD one wave multiplies while
            // the other reads fragments.  Behind the barrier K-tile kt+1 is visible to everyone (each wave waited for its own pieces:
            // everything but the four of A(kt+2)) and nobody reads K-tile kt from LDS any more (quadrant 4 runs from registers), so
            // the pieces issued after it -- B(kt+2) into B(kt)'s slot, A(kt+3) into A(kt)'s -- are safe.  Each wave issues them two per
            // quadrant in the order B half 0, B half 1, A half 0, A half 1 counted from ITS barrier: the leading group in quadrants
            // 1-4 of the next K-tile, the trailing group in quadrant 4 of this one and 1-3 of the next.
#define R_BAR()                                                                  \
    do {                                                                         \
        if (STAMP && (ablate & 6)) { }                                           \
        else if (a_iss) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         \
        else if (pre_k) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NST + NAUXPRE) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                       \
        __builtin_amdgcn_s_barrier();                                            \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
            // ---------------- quadrant 1: rows 0-63 x n-tiles 0,1
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) b0[j][s] = *reinterpret_cast<const bf16x8 *>(smem + ((sb + j * 512) ^ (s * 64)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) a[i][s] = *reinterpret_cast<const bf16x8 *>(smem + ((sa + i * 2048) ^ (s * 64)));
            if (!pre_k && b_ok) {
                if constexpr (!LATE) R_DMA_B(0, gb1, b_base + b_kt * (BK * 2));
                else { R_DMA_B(1, gb1, b_base + b_kt * (BK * 2)); R_ADV_B(); }
            }
            R_PHASE_SYNC_A();
            R_QUAD(0, 0, b0);
            R_PHASE_SYNC_B();
            // ---------------- quadrant 2: rows 0-63 x n-tiles 2,3
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 2; ++s) b1[j][s] = *reinterpret_cast<const bf16x8 *>(smem + ((sb + 4096 + j * 512) ^ (s * 64)));
            if constexpr (!LATE) {
                if (!pre_k && b_ok) { R_DMA_B(1, gb1, b_base + b_kt * (BK * 2)); R_ADV_B(); }
            } else {
                if (a_iss) R_DMA_A(0, ga2, a_base + a_kt * (BK * 2));
            }
            R_PHASE_SYNC_A();
            R_QUAD(0, 2, b1);
            R_PHASE_SYNC_B();
            // ---------------- quadrant 3: rows 64-127 x n-tiles 2,3
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int s = 0; s < 2; ++s) a[i][s] = *reinterpret_cast<const bf16x8 *>(smem + ((sa + (4 + i) * 2048) ^ (s * 64)));
            if (a_iss) {
                if constexpr (!LATE) R_DMA_A(0, ga2, a_base + a_kt * (BK * 2));
                else { R_DMA_A(1, ga2, a_base + a_kt * (BK * 2)); R_ADV_A(); }
            }
            R_PHASE_SYNC_A();
            R_QUAD(4, 2, b1);
            R_PHASE_SYNC_B();
            // ---------------- quadrant 4: rows 64-127 x n-tiles 0,1 (no LDS reads)
            if constexpr (LATE) {
                R_BAR();
                if (b_ok) R_DMA_B(0, gb, b_base + b_kt * (BK * 2));   // first half of B(kt+2), into the slot of B(kt)
            } else {
                if (a_iss) { R_DMA_A(1, ga2, a_base + a_kt * (BK * 2)); R_ADV_A(); }
            }
            if constexpr (OPS != 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); }
            R_QUAD(4, 0, b0);
            R_PHASE_SYNC_B();
            if constexpr (!LATE) R_BAR();
            ga = ga == 2 ? 0 : ga + 1;
            gb ^= 1;
        }
        };
        if (late) kloop(std::true_type{});
        else kloop(std::false_type{});
#undef R_BAR
        // ---------------- output tile done; the epilogue runs straight from the accumulators (no LDS memory involved, so the operand
        // stream of the next tile stays in flight underneath and no barrier surrounds it).
        const int next_it = it + (int)gridDim.x;
        const bool has_next = next_it < nitems;
        // Issue what the next tile's K-tile 0 would issue -- B(1), A(2) -- NOW, ahead of the epilogue's stores: vmcnt retires in
        // issue order, so a counted wait that sits BEHIND the stores stalls the next main loop until they are acknowledged; this
        // way the first wait that covers them comes two K-tiles later.  (The trailing group has sent B(1)'s first half already.)
        // (Issued from inside the epilogue, behind its up-front row loads: see nt_epilogue.)
        auto issue_next = [&]() __attribute__((always_inline)) {
            if (a_ok && b_ok && has_next) {
                if (!late) R_DMA_B(0, gb ^ 1, b_base + b_kt * (BK * 2));
                R_DMA_B(1, gb ^ 1, b_base + b_kt * (BK * 2)); R_ADV_B();
                const int gaf = ga == 0 ? 2 : ga - 1;
                R_DMA_A(0, gaf, a_base + a_kt * (BK * 2)); R_DMA_A(1, gaf, a_base + a_kt * (BK * 2)); R_ADV_A();
                pre = true;
            }
        };
        [[maybe_unused]] const unsigned long long st_b = NT_STAMP_T();
        if constexpr (kAuxPre) nt_epilogue<TO, FL, CAUX>(acc, d, e, bf, cm0, cn0, wave, lane, issue_next, auxpre);
        else nt_epilogue<TO, FL, CAUX>(acc, d, e, bf, cm0, cn0, wave, lane, issue_next);
        if constexpr (STAMP) { st_main += st_b - st_a; st_epi += NT_STAMP_T() - st_b; ++st_ntile; }
        if (!has_next) break;
        it = next_it;
        if (a_it == it) { cm0 = nm0; cn0 = nn0; }                     // the producer cursor already decoded this item
        else decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
    }
#ifdef ECGVIT_TOOLS
    if constexpr (STAMP) {
        if (threadIdx.x == 0) {
            unsigned long long *o = g_nt_stamps + blockIdx.x * 8;
            o[0] = st_t0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
            o[4] = st_main; o[5] = st_epi; o[6] = st_ntile; o[7] = (unsigned long long)nk;
        }
    }
#endif
#undef R_DMA_A
#undef R_DMA_B
#undef R_PHASE_SYNC_A
#undef R_MMA
#undef R_QUAD
#undef R_PHASE_SYNC_B
#undef R_ADV_A
#undef R_ADV_B
}

// ---------------------------------------------------------------------------------------------------------------------------------
// gemm_nt_kernel_4w: the same tile, LDS images, ring, tile walk and epilogue with FOUR waves -- one per SIMD, each owning a 128 x 128
// block of the tile in 256 accumulator registers (the kernel's 512 registers per lane = one wave per SIMD).  Every A fragment feeds
// eight MFMAs instead of four (128 KB of fragment reads per K-tile and CU instead of 192), and ONE instruction stream per SIMD carries
// the 64 MFMAs of a 32-deep k-step with the next k-step's 16 fragment reads and eight DMA pieces placed between them, in a fixed
// order (the way the vendor library's own 256x256x64 kernel is written; profiles/r03_gemm_4w.txt).  Its main loop costs ~2,500 cycles
// per K-tile where the eight-wave kernel's costs ~2,870 (2,048 = MFMA alone), but its epilogue runs on ONE wave per SIMD -- half the
// VALU issue rate of two -- so it serves the launches whose time is main loop: plain products with K >= 1536 (the QKV and FFN-up
// input gradients).  Results are bit-identical to gemm_nt_kernel's (same MFMA, same K order, same epilogue code).
template <int... X, typename F>
__device__ __forceinline__ void q4_static_for_impl(std::integer_sequence<int, X...>, F &&f) { (f(std::integral_constant<int, X>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void q4_static_for(F &&f) { q4_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
// where in a k-step's 64 MFMAs the 16 fragment reads and 8 DMA pieces go (index of the read / piece issued behind MFMA x, or -1): one
// read per four MFMAs, one piece per eight.  (Measured against two reads + one piece per eight MFMAs: equal; against the pieces
// front-loaded one per four MFMAs: 0-3 % slower.)
__device__ constexpr int q4_read_at(int x) { return (x & 3) == 1 ? x >> 2 : -1; }
__device__ constexpr int q4_dma_at(int x) { return (x & 7) == 3 ? x >> 3 : -1; }
template <typename TO, int FL, int CAUX = 0, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void gemm_nt_kernel_4w(ecgvit_gemm_desc d, EpiParams e, int tiles_m, int tiles_n, int ngroup, int nitems, int ablate) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    e.alpha = 1.f;   // (the launcher takes alpha == 1 only: the epilogue's scaling branch folds away)
    const int M = d.M, N = d.N;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = d.K / BK;
    const int lda2 = (int)d.lda * 2, ldb2 = (int)d.ldb * 2;
    const int ntile = tiles_m * tiles_n;
    int it = blockIdx.x;
    if (it >= nitems) return;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.A), 0, (uint32_t)((int64_t)M * lda2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(d.B), 0, (uint32_t)((int64_t)N * ldb2), 0x00020000);
    NtBufs bf;
    bf.ldc2 = (int)d.ldc * (int)sizeof(TO); bf.ldr2 = (int)e.ldr * 2; bf.ldx2 = (int)e.ldaux * 2;
    bf.c = __builtin_amdgcn_make_buffer_rsrc(d.C, 0, (STAMP && (ablate & 1)) ? 0u : (uint32_t)((int64_t)M * bf.ldc2), 0x00020000);   // ablate 1: stores dropped
    bf.res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(e.residual), 0, e.residual ? (uint32_t)((int64_t)M * bf.ldr2) : 0u, 0x00020000);
    bf.aux = __builtin_amdgcn_make_buffer_rsrc(e.aux, 0, e.aux ? (uint32_t)((int64_t)M * bf.ldx2) : 0u, 0x00020000);
    bf.ldq = 0; bf.q8 = bf.aux; bf.q8_bf8 = false; bf.q8_inv = 0.f;
    bf.t_out = ((lane >> 2) + 16 * (lane & 3)) << 2;
    bf.t_in = (4 * (lane & 15) + (lane >> 4)) << 2;
    // this wave's four DMA pieces of a half-tile: rows 32*wave + 8*i + (lane >> 3)
    // (recomputed behind every epilogue from an opaque copy of the lane id: kept live across the epilogue they are spilled, and hipcc then
    // guards every DMA piece of the main loop with a vmcnt wait for the reload)
    int voA[4], voB[4];
    auto calc_vo = [&]() __attribute__((always_inline)) {
        int ln;   // the lane id, from the hardware each time (a live copy across the main loop is what gets spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wave + 8 * i + (ln >> 3), p = ln & 7;
            voA[i] = r * lda2 + ((p ^ ((r >> 1) & 7)) << 4);
            const int fw = (((r >> 3) & 3) << 1) | ((r >> 1) & 1);
            voB[i] = r * ldb2 + ((p ^ fw) << 4);
        }
    };
    calc_vo();
    const int fr = lane & 15, fq = lane >> 4;
    const int loff = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
    const int rl = 8 * (fr >> 2) + (fr & 3);
    const int boff = rl * 128 + ((fq ^ (((fr >> 2) << 1) | ((fr >> 1) & 1))) << 4);   // n-tile j of column group g: + 8192 g + 512 (j&1) + 4096 (j>>1)

#define Q_DMA_A(h, ring, soff)                                                                                \
    do {                                                                                                      \
        char *dst_ = smem + (3 * (h) + (ring)) * HALF_BYTES + wave * 4096;                                    \
        const int so_ = (soff) + (h) * 128 * lda2;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dst_ + 1024 * i_), 16, voA[i_], so_, 0, 0); \
    } while (0)
#define Q_DMA_B(h, ring, soff)                                                                                \
    do {                                                                                                      \
        char *dst_ = smem + (6 + 2 * (h) + (ring)) * HALF_BYTES + wave * 4096;                                \
        const int so_ = (soff) + (h) * 128 * ldb2;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dst_ + 1024 * i_), 16, voB[i_], so_, 0, 0); \
    } while (0)

    int cm0, cn0, nm0, nn0;
    decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
    nm0 = cm0; nn0 = cn0;
    // producer cursors; past the end of this workgroup's share they stay where they are (harmless re-reads into free slots): the
    // pieces are issued unconditionally so that a k-step stays ONE scheduling region
    int a_it = it, a_kt = 0, a_base = cm0 * lda2;
    int b_it = it, b_kt = 0, b_base = cn0 * ldb2;
#define Q_ADV_A()                                                                                             \
    do {                                                                                                      \
        if (++a_kt == nk) {                                                                                   \
            a_kt = 0;                                                                                         \
            if (a_it + (int)gridDim.x < nitems) { a_it += (int)gridDim.x; decode_tile(a_it, ntile, tiles_m, tiles_n, ngroup, nm0, nn0); a_base = nm0 * lda2; } \
        }                                                                                                     \
    } while (0)
#define Q_ADV_B()                                                                                             \
    do {                                                                                                      \
        if (++b_kt == nk) {                                                                                   \
            b_kt = 0;                                                                                         \
            if (b_it + (int)gridDim.x < nitems) { b_it += (int)gridDim.x; b_base = nn0 * ldb2; }              \
        }                                                                                                     \
    } while (0)
    // fragments of one 32-deep k-step: 8 n-tiles of B (reads 0-7), 8 m-tiles of A (reads 8-15)
#define Q_READ1(AF, BFR, SA, SB, S, R)                                                                        \
    do {                                                                                                      \
        if ((R) < 8) BFR[(R) & 7] = *reinterpret_cast<const bf16x8 *>(smem + (((SB) + 8192 * (((R) & 7) >> 2) + 512 * ((R) & 1) + 4096 * (((R) >> 1) & 1)) ^ ((S) * 64))); \
        else AF[(R) & 7] = *reinterpret_cast<const bf16x8 *>(smem + (((SA) + ((R) & 7) * 2048) ^ ((S) * 64))); \
    } while (0)
#define Q_READ(AF, BFR, GA, GB, S)                                                                            \
    do {                                                                                                      \
        const int sa_ = (3 * wm + (GA)) * HALF_BYTES + loff, sb_ = (6 + 2 * wn + (GB)) * HALF_BYTES + boff;   \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) Q_READ1(AF, BFR, sa_, sb_, S, r_);                  \
    } while (0)
#define Q_MMA1(AF, BFR, X)                                                                                    \
    acc[((X) & 7) >> 2][(X) >> 3][(X) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFR[(X) & 7], AF[(X) >> 3], acc[((X) & 7) >> 2][(X) >> 3][(X) & 3], 0, 0, 0)
#define Q_FENCE() __builtin_amdgcn_sched_barrier(0)
    // ONE k-step in issue order: 64 MFMAs from (AF, BFR); between them the 16 fragment reads of the next k-step into (AN, BN) and, with
    // DMA, eight pieces (RS / VO / pitch LD, half-tiles 0 and 1 into LDS at D0 / D1, source offset SO).  Fenced so that the order stays.
#define Q_KSTEP(AF, BFR, AN, BN, GA, GB, S, RD, DMA, RS, VO, LD, D0, D1, SO)                                      \
    do {                                                                                                      \
        const int sa_ = (3 * wm + (GA)) * HALF_BYTES + loff, sb_ = (6 + 2 * wn + (GB)) * HALF_BYTES + boff;   \
        q4_static_for<64>([&](auto xc_) __attribute__((always_inline)) {                                      \
            constexpr int x_ = decltype(xc_)::value, r_ = q4_read_at(x_), p_ = q4_dma_at(x_);                 \
            Q_MMA1(AF, BFR, x_);                                                                              \
            if constexpr ((RD) && r_ >= 0) { Q_FENCE(); Q_READ1(AN, BN, sa_, sb_, S, r_); Q_FENCE(); }        \
            if constexpr ((DMA) && p_ >= 0) {                                                                 \
                Q_FENCE();                                                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lptr_t)((p_ < 4 ? (D0) : (D1)) + 1024 * (p_ & 3)), 16, VO[p_ & 3], \
                                                         (SO) + (p_ < 4 ? 0 : 128 * (LD)), 0, 0);              \
                Q_FENCE();                                                                                    \
            }                                                                                                 \
        });                                                                                                   \
        Q_FENCE();                                                                                            \
    } while (0)

    bf16x8 a0[8], b0[8];
    // prologue: A(0), B(0), A(1); behind the barrier the first k-step's fragments, B(1) and A(2)
    Q_DMA_A(0, 0, a_base); Q_DMA_A(1, 0, a_base); Q_ADV_A();
    Q_DMA_B(0, 0, b_base); Q_DMA_B(1, 0, b_base); Q_ADV_B();
    Q_DMA_A(0, 1, a_base + a_kt * (BK * 2)); Q_DMA_A(1, 1, a_base + a_kt * (BK * 2)); Q_ADV_A();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Q_READ(a0, b0, 0, 0, 0);
    Q_DMA_B(0, 1, b_base + b_kt * (BK * 2)); Q_DMA_B(1, 1, b_base + b_kt * (BK * 2)); Q_ADV_B();
    Q_DMA_A(0, 2, a_base + a_kt * (BK * 2)); Q_DMA_A(1, 2, a_base + a_kt * (BK * 2)); Q_ADV_A();
    int ga = 0, gb = 0;
    bool first = true;
    // vector-memory instructions the epilogue issues BEHIND the pieces it sends ahead: the output stores of both column halves and the
    // second half's row loads
    constexpr int NST = 2 * (sizeof(TO) == 2 ? 16 : 32) + ((FL & ECGVIT_EPI_RESIDUAL) ? 16 : 0) + ((FL & ECGVIT_EPI_MUL_AUX) ? 16 : 0);
    static_assert(FL >= 0 && !(FL & (ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_BWD | ECGVIT_EPI_ACCUM | ECGVIT_EPI_QUANT_OUT)) && 8 + NST <= 63, "light bodies only");
    constexpr bool kRoll = (FL & ECGVIT_EPI_MUL_AUX) != 0;
    [[maybe_unused]] unsigned long long st_t0 = 0, st_r0 = 0, st_main = 0, st_epi = 0, st_ntile = 0;
#ifdef ECGVIT_TOOLS
    if constexpr (STAMP) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (;;) {
        [[maybe_unused]] const unsigned long long st_a = NT_STAMP_T();
        f32x4 acc[2][8][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[g][i][j][r] = 0.f;
        // one K-tile; its first and last instance of a tile are separate straight-line copies (no control flow merges 256 accumulators)
        auto ktile = [&](auto first_c, auto last_c) __attribute__((always_inline)) {
            constexpr bool KFIRST = decltype(first_c)::value, KLAST = decltype(last_c)::value;
            bf16x8 a1[8], b1[8];
            const int ga1 = ga == 2 ? 0 : ga + 1, ga2 = ga == 0 ? 2 : ga - 1, gb1 = gb ^ 1;
            // ---- k-step 0 of K-tile kt (fragments already in a0 / b0); reads of k-step 1; A(kt+2) goes out (kt = 0: already in flight)
            Q_FENCE();
            if constexpr (!KFIRST) {
                Q_KSTEP(a0, b0, a1, b1, ga, gb, 1, true, true, rsA, voA, lda2, smem + ga2 * HALF_BYTES + wave * 4096, smem + (3 + ga2) * HALF_BYTES + wave * 4096,
                        a_base + a_kt * (BK * 2));
                Q_ADV_A();
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                Q_KSTEP(a0, b0, a1, b1, ga, gb, 1, true, false, rsA, voA, lda2, smem, smem, 0);
                if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NST) : "memory");
            }
            Q_FENCE();
            __builtin_amdgcn_s_barrier();
            Q_FENCE();
            // ---- k-step 1; K-tile kt+1 is visible and nobody reads K-tile kt from LDS any more: reads of (kt+1, k-step 0), B(kt+2) goes out
            // (the last k-step of a tile reads nothing: the next tile's first fragments are fetched behind the epilogue, which needs the registers)
            Q_KSTEP(a1, b1, a0, b0, ga1, gb1, 0, !KLAST, true, rsB, voB, ldb2, smem + (6 + gb) * HALF_BYTES + wave * 4096, smem + (8 + gb) * HALF_BYTES + wave * 4096,
                    b_base + b_kt * (BK * 2));
            Q_ADV_B();
            ga = ga1; gb = gb1;
        };
        ktile(std::true_type{}, std::false_type{});
#pragma unroll 1
        for (int kt = 1; kt < nk - 1; ++kt) ktile(std::false_type{}, std::false_type{});
        ktile(std::false_type{}, std::true_type{});
        first = false;
        const int next_it = it + (int)gridDim.x;
        const bool has_next = next_it < nitems;
        // the next tile's A(2) goes out from inside the epilogue of the first column half: behind that half's own row loads (residual),
        // ahead of every store (vmcnt retires in issue order; see nt_epilogue)
        auto issue_next = [&]() __attribute__((always_inline)) {
            const int gaf = ga == 0 ? 2 : ga - 1;
            Q_DMA_A(0, gaf, a_base + a_kt * (BK * 2)); Q_DMA_A(1, gaf, a_base + a_kt * (BK * 2)); Q_ADV_A();
        };
        auto none = [&]() __attribute__((always_inline)) {};
        [[maybe_unused]] const unsigned long long st_b = NT_STAMP_T();
        if (STAMP && (ablate & 2)) {   // ablate 2: no epilogue at all (the accumulators are consumed by one dummy store)
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum += acc[g][i][j][0];
            if (sum == 12345.678f) reinterpret_cast<float *>(d.C)[lane] = sum;
        } else {
        // (the epilogue's lane-derived offsets from a fresh lane id, per tile: hoisted out of the tile loop they are
        // spilled, and the reload's `s_waitcnt vmcnt(0)` would wait for the pieces just issued)
        int eln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(eln));
        bf.t_out = ((eln >> 2) + 16 * (eln & 3)) << 2;
        nt_epilogue<TO, FL, CAUX, kRoll>(acc[0], d, e, bf, cm0, cn0, 4 * wm + 2 * wn, eln, issue_next);
        nt_epilogue<TO, FL, CAUX, kRoll>(acc[1], d, e, bf, cm0, cn0, 4 * wm + 2 * wn + 1, eln, none);
        }
        if constexpr (STAMP) { st_main += st_b - st_a; st_epi += NT_STAMP_T() - st_b; ++st_ntile; }
        if (!has_next) break;
        it = next_it;
        decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
        calc_vo();
        Q_READ(a0, b0, ga, gb, 0);
    }
    // nothing of this wave's may still be on its way into LDS when the workgroup's allocation is released (the cursors keep issuing
    // pieces past the end of the share)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ECGVIT_TOOLS
    if constexpr (STAMP) {
        if (threadIdx.x == 0) {
            unsigned long long *o = g_nt_stamps + blockIdx.x * 8;
            o[0] = st_t0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
            o[4] = st_main; o[5] = st_epi; o[6] = st_ntile; o[7] = (unsigned long long)nk;
        }
    }
#endif
#undef Q_DMA_A
#undef Q_DMA_B
#undef Q_ADV_A
#undef Q_ADV_B
#undef Q_READ
#undef Q_READ1
#undef Q_MMA1
#undef Q_FENCE
#undef Q_KSTEP
}


}  // namespace

static int nt_default_group(int tiles_n) { return tiles_n <= 8 ? tiles_n : 6; }   // n-tiles per column group of the built-in tile walk (ecgvit_gemm_nt_launch)

bool ecgvit_gemm_nt_applicable(const ecgvit_gemm_desc *d) {
    const bool f8 = d->dtype == ECGVIT_FP8_E4M3 || d->dtype == ECGVIT_BF8_E5M2;
    if (d->layout != ECGVIT_GEMM_NT || !(d->dtype == ECGVIT_BF16 || f8)) return false;
    if (d->batch1 != 1 || d->batch2 != 1) return false;
    if (d->M < 2048 || d->N < 128 || d->N % 8 != 0) return false;
    if (f8 ? (d->K % 128 != 0 || d->K < 384 || d->lda % 16 != 0 || d->ldb % 16 != 0 || d->out_dtype != ECGVIT_BF16) : (d->K % 64 != 0 || d->K < 192)) return false;
    const int es = f8 ? 1 : 2;
    if ((int64_t)d->M * d->lda * es + 65536 * d->lda >= (1ll << 31) || (int64_t)d->N * d->ldb * es + 65536 * d->ldb >= (1ll << 31)) return false;
    const int64_t esz = d->out_dtype == ECGVIT_BF16 ? 2 : 4, rows = (int64_t)d->M + 256;   // epilogue offsets are 32-bit byte offsets
    if (rows * d->ldc * esz >= (1ll << 31) || rows * d->ldr * 2 >= (1ll << 31) || rows * d->ldaux * 2 >= (1ll << 31)) return false;
    if (d->epilogue & ECGVIT_EPI_NO_OUT) {   // only the emitting FFN-wide bodies of the 8-bit kernel have a no-output form
        const int fl = d->epilogue & ~(ECGVIT_EPI_NO_OUT | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_AUX8);
        const int up = ECGVIT_EPI_BIAS | ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_GRAD_AUX | ECGVIT_EPI_QUANT_OUT, dh = ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_COLSUM | ECGVIT_EPI_QUANT_OUT;
        if (!((d->dtype == ECGVIT_FP8_E4M3 && fl == up) || (d->dtype == ECGVIT_BF8_E5M2 && (d->epilogue & ~ECGVIT_EPI_AUX8) == (dh | ECGVIT_EPI_NO_OUT)))) return false;
    } else if (!d->C) {
        return false;
    }
    if (d->epilogue & ECGVIT_EPI_AUX8) {   // the e4m3 saved tensor: the two FFN-wide bodies (bf16 operands, or 8-bit operands with their emitting forms)
        const int fl = d->epilogue & ~(ECGVIT_EPI_AUX8 | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT);
        const int up = ECGVIT_EPI_BIAS | ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_GRAD_AUX, dh = ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_COLSUM;
        if (d->out_dtype != ECGVIT_BF16 || !d->aux || d->ldaux % 8 || reinterpret_cast<uintptr_t>(d->aux) % 8) return false;
        if (d->dtype == ECGVIT_BF16 && (d->epilogue & (ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT))) return false;
        const bool is_up = fl == up && d->dtype != ECGVIT_BF8_E5M2, is_dh = fl == dh && !(d->epilogue & ECGVIT_EPI_DROPOUT) && d->dtype != ECGVIT_FP8_E4M3;
        if (!(is_up || is_dh)) return false;
    }
    if (d->epilogue & ECGVIT_EPI_QUANT_OUT) {
        if (!f8 || !d->q8_out || !d->q8_scale || !d->q8_amax || d->ldq8 % 8 || reinterpret_cast<uintptr_t>(d->q8_out) % 8 ||
            (d->q8_format != ECGVIT_FP8_E4M3 && d->q8_format != ECGVIT_BF8_E5M2) || rows * d->ldq8 >= (1ll << 31))
            return false;
    }
    if (d->epilogue & ECGVIT_EPI_COLSUM) {
        if (d->out_dtype != ECGVIT_BF16 || !d->workspace || !d->colsum_out ||
            d->workspace_bytes < (int64_t)8 * ((d->M + BM - 1) / BM) * d->N)
            return false;
    }
    return true;
}

void ecgvit_colsum_reduce_launch(const float *partial, int nparts, int N, float *out, hipStream_t s);   // gemm_wgrad.hip

int ecgvit_gemm_nt4w_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag);
// argument validation is done by the caller (ecgvit_gemm_bf16_launch); raster_g <= 0 selects the built-in choice
int ecgvit_gemm_nt_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag) {
#ifdef ECGVIT_TOOLS
    {   // tools build only: whole-step A/B of the diag bits (bench.py under ECGVIT_HIP_LIB=libecgvit_hip_tools.so)
        static const int env_diag = [] { const char *e = getenv("ECGVIT_NT_DIAG"); return e ? atoi(e) : 0; }();
        static const int env_g = [] { const char *e = getenv("ECGVIT_NT_G"); return e ? atoi(e) : 0; }();
        diag |= env_diag;
        if (raster_g == 0) raster_g = env_g;
    }
#endif
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    // Built-in tile walk: column groups of 6 n-tiles (m-major inside a group).  Measured inside the train step against the plain
    // n-fastest order (tools/pmc_step_raster.sh, tools/ab_bench.sh ECGVIT_NT_G): the same step time (+-0.02 %) with 13 % fewer bytes
    // fetched from beyond L2 per launch (1.08 -> 0.94 GB); groups of 3 fetch 0.97 GB at -0.1 %, groups of 4 cost 0.6 % of the step.
    // Up to 8 n-tiles (N <= 2048: the FFN-wide products of EcgVit-small) stay ONE group -- a 6 + 2 split costs that step 0.9 % (round 4).
    const int G = raster_g > 0 ? std::min(raster_g, tiles_n) : nt_default_group(tiles_n);
    const EpiParams e = make_epi(d);
    // persistent (one workgroup per CU, static shares) unless the caller asks for dispatcher-balanced chunks of ~k tiles
    const int tpw = d->tiles_per_workgroup;
    const dim3 grid((unsigned)(tpw > 0 ? std::max(std::min(ntile, 256), (ntile + tpw - 1) / tpw) : std::min(ntile, 256))), block(512);
    const int fl = d->epilogue;
    constexpr int F_LIN = ECGVIT_EPI_BIAS | ECGVIT_EPI_RESIDUAL;
    constexpr int F_UP = ECGVIT_EPI_BIAS | ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_GRAD_AUX;
    constexpr int F_DH = ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_COLSUM;
#define NT_LAUNCH(TO, FL) hipLaunchKernelGGL((gemm_nt_kernel<TO, FL>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, 0)
#ifdef ECGVIT_TOOLS
    if ((diag & 1) && d->out_dtype == ECGVIT_BF16) {   // stamped diagnostic instantiations; diag & 2: output stores dropped
        const int ab = (diag >> 1) & 31;   // ablate bits: 1 stores dropped, 2 no DMA after the prologue, 4 no counted waits, 8 operand cursors frozen, 16 XCD rendezvous per tile round
        if (ab & 16) {
            void *p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_nt_rdv)) != hipSuccess || hipMemsetAsync(p, 0, sizeof(unsigned int) * 8 * 32, s) != hipSuccess) return ECGVIT_ELAUNCH;
        }
        if (fl == 0) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 0, true>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, ab);
        else if (fl == (F_UP | ECGVIT_EPI_DROPOUT)) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, F_UP | ECGVIT_EPI_DROPOUT, true>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, ab);
        else if (fl == F_DH) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, F_DH, true>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, ab);
        else return ECGVIT_EINVAL;
        ECGVIT_CHECK_LAUNCH();
        return ECGVIT_OK;
    }
#endif
#define NT_LAUNCH8(FL, OPS) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, FL, false, OPS>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, 0)
#ifdef ECGVIT_TOOLS
    if ((diag & 1024) && d->dtype == ECGVIT_BF16 && d->out_dtype == ECGVIT_BF16) {   // LayerNorm-fold pricing (ECGVIT_EPI_ROWAFFINE_X): QKV forward / FFN-up forward bodies
        constexpr int RA = ECGVIT_EPI_ROWAFFINE_X;
        if (fl == ECGVIT_EPI_BIAS) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, ECGVIT_EPI_BIAS | RA, false, 0, 2>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, 0);
        else if (fl == (F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_AUX8)) NT_LAUNCH(bf16_t, F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_AUX8 | RA);
        else return ECGVIT_EINVAL;
        ECGVIT_CHECK_LAUNCH();
        return ECGVIT_OK;
    }
    if ((diag & 64) && fl == 0 && d->dtype == ECGVIT_FP8_E4M3) { NT_LAUNCH8(0, 1); ECGVIT_CHECK_LAUNCH(); return ECGVIT_OK; }   // plain fp8 MFMA (A/B)
#endif
    // plain 8-bit products whose bf16 output does not fit the 256-MB Infinity Cache (EcgVit-large: the QKV forward's 788 MB) store it non-temporally, as
    // the bf16 QKV forward does since round 3: written through L2 the output evicts the operand panels the tile's neighbours are about to re-read
    // (tools/fp8_nt_ab.py at 256 x 501 token rows, default -> non-temporal: QKV forward K = 1024, 752 MB: 403.8 -> 350.5 us; the 250-MB outputs: K = 1024
    // 145.5 -> 133.8, K = 3072 323.0 -> 338.1, K = 4096 411.8 -> 422.9: a long main loop re-reads its panels from L2 often enough to want the cache's help)
    [[maybe_unused]] bool nt8 = (int64_t)d->M * d->N * 2 > (320ll << 20) || ((int64_t)d->M * d->N * 2 > (240ll << 20) && d->K <= 1024);
#ifdef ECGVIT_TOOLS
    if (diag & 256) nt8 = false;   // A/B: bit 256 = default-policy stores, 512 = non-temporal stores, whatever the size
    if (diag & 512) nt8 = true;
#endif
#ifdef ECGVIT_AB_NO_NT8
    nt8 = false;   // (A/B builds only: tools/ab_bench.sh --hip-lib)
#endif
#ifdef ECGVIT_AB_NT8_BIG_ONLY
    nt8 = (int64_t)d->M * d->N * 2 > (320ll << 20);
#endif
#define NT_LAUNCH8_NT(OPS) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 0, false, OPS, 2>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, 0)
    if (d->dtype == ECGVIT_FP8_E4M3) {          // forward products: e4m3 activations x e4m3 weights
        switch (fl) {
            case 0: if (nt8) NT_LAUNCH8_NT(3); else NT_LAUNCH8(0, 3); break;
            case F_LIN: NT_LAUNCH8(F_LIN, 3); break;
            case F_LIN | ECGVIT_EPI_DROPOUT: NT_LAUNCH8(F_LIN | ECGVIT_EPI_DROPOUT, 3); break;
            case F_UP: NT_LAUNCH8(F_UP, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT, 3); break;
            case F_UP | ECGVIT_EPI_QUANT_OUT: NT_LAUNCH8(F_UP | ECGVIT_EPI_QUANT_OUT, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT, 3); break;
            case F_UP | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT: NT_LAUNCH8(F_UP | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT, 3); break;
#define A8 ECGVIT_EPI_AUX8
            case F_UP | A8: NT_LAUNCH8(F_UP | A8, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT | A8: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT | A8, 3); break;
            case F_UP | ECGVIT_EPI_QUANT_OUT | A8: NT_LAUNCH8(F_UP | ECGVIT_EPI_QUANT_OUT | A8, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | A8: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | A8, 3); break;
            case F_UP | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8: NT_LAUNCH8(F_UP | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8, 3); break;
            case F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8: NT_LAUNCH8(F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8, 3); break;
            default: if (fl & A8) return ECGVIT_EINVAL; NT_LAUNCH8(-1, 3); break;
        }
    } else if (d->dtype == ECGVIT_BF8_E5M2) {   // input-gradient products: e5m2 gradients x e4m3 transposed weights
        switch (fl) {
            case 0: if (nt8) NT_LAUNCH8_NT(4); else NT_LAUNCH8(0, 4); break;
            case F_DH: NT_LAUNCH8(F_DH, 4); break;
            case F_DH | ECGVIT_EPI_QUANT_OUT: NT_LAUNCH8(F_DH | ECGVIT_EPI_QUANT_OUT, 4); break;
            case F_DH | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT: NT_LAUNCH8(F_DH | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT, 4); break;
            case F_DH | A8: NT_LAUNCH8(F_DH | A8, 4); break;
            case F_DH | ECGVIT_EPI_QUANT_OUT | A8: NT_LAUNCH8(F_DH | ECGVIT_EPI_QUANT_OUT | A8, 4); break;
            case F_DH | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8: NT_LAUNCH8(F_DH | ECGVIT_EPI_QUANT_OUT | ECGVIT_EPI_NO_OUT | A8, 4); break;
#undef A8
            default: if (fl & ECGVIT_EPI_AUX8) return ECGVIT_EINVAL; NT_LAUNCH8(-1, 4); break;
        }
    } else if (d->out_dtype == ECGVIT_BF16) {
        if (fl == 0) {
            // plain products.  K >= 1536 (the QKV and FFN-up input gradients): the four-wave body (alpha 1 only).
            // Outputs that do not fit the 256 MB Infinity Cache (QKV forward: 592 MB) are stored non-temporally: written through L2 they
            // evict the operand panels the tile's neighbours are about to re-read (main loop 3,020 -> 2,620 cycles per K-tile, launch
            // -6...-10 %); smaller outputs (197 MB) are absorbed by the cache and nt costs them 2-3 % (profiles/r03_gemm_4w.txt)
            const bool big_out = (int64_t)d->M * d->N * 2 > (256ll << 20);
#ifdef ECGVIT_TOOLS
            // A/B: ECGVIT_NT_NO4W (whole step) or diag bits 128 / 256 / 512 (one launch): 1 = eight-wave body everywhere, 2 = no nt stores, 4 = nt stores everywhere
            static const int env_no4w = [] { const char *e_ = getenv("ECGVIT_NT_NO4W"); return e_ ? atoi(e_) : 0; }();
            const int no4w = env_no4w | ((diag >> 7) & 7);
            const bool use4w = !(no4w & 1), use_nt = (big_out && !(no4w & 2)) || (no4w & 4);
#else
            const bool use4w = true, use_nt = big_out;
#endif
            if (use4w && d->K >= 1536 && e.alpha == 1.f && !d->scale_a && !d->scale_b) return ecgvit_gemm_nt4w_launch(d, s, raster_g, use_nt ? 2 : 0);
            if (use_nt) hipLaunchKernelGGL((gemm_nt_kernel<bf16_t, 0, false, 0, 2>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, 0);
            else NT_LAUNCH(bf16_t, 0);
            ECGVIT_CHECK_LAUNCH();
            return ECGVIT_OK;
        }
        // the two residual launches (bias + residual [+ dropout]: attn-out and FFN-down forward) with K >= 768: the four-wave body as well --
        // its shorter main loop outweighs the one-wave epilogue (launch -2 % at K = 768, -3 % at K = 3072; step +0.2...0.3 %)
        {
#ifdef ECGVIT_TOOLS
            static const int env_no4w = [] { const char *e_ = getenv("ECGVIT_NT_NO4W"); return e_ ? atoi(e_) : 0; }();
            const bool use4w = !((env_no4w | (diag >> 7)) & 1);
#else
            const bool use4w = true;
#endif
            if (use4w && (fl & ~ECGVIT_EPI_DROPOUT) == F_LIN && d->K >= 768 && e.alpha == 1.f && !d->scale_a && !d->scale_b)
                return ecgvit_gemm_nt4w_launch(d, s, raster_g, 0);
        }
        switch (fl) {
            case ECGVIT_EPI_BIAS: NT_LAUNCH(bf16_t, ECGVIT_EPI_BIAS); break;   // the masked objective's pixel head
            case F_LIN: NT_LAUNCH(bf16_t, F_LIN); break;
            case F_LIN | ECGVIT_EPI_DROPOUT: NT_LAUNCH(bf16_t, F_LIN | ECGVIT_EPI_DROPOUT); break;
            case F_UP: NT_LAUNCH(bf16_t, F_UP); break;
            case F_UP | ECGVIT_EPI_DROPOUT: NT_LAUNCH(bf16_t, F_UP | ECGVIT_EPI_DROPOUT); break;
            case F_DH: NT_LAUNCH(bf16_t, F_DH); break;
            case F_UP | ECGVIT_EPI_AUX8: NT_LAUNCH(bf16_t, F_UP | ECGVIT_EPI_AUX8); break;
            case F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_AUX8: NT_LAUNCH(bf16_t, F_UP | ECGVIT_EPI_DROPOUT | ECGVIT_EPI_AUX8); break;
            case F_DH | ECGVIT_EPI_AUX8: NT_LAUNCH(bf16_t, F_DH | ECGVIT_EPI_AUX8); break;
            default: if (fl & ECGVIT_EPI_AUX8) return ECGVIT_EINVAL; NT_LAUNCH(bf16_t, -1); break;
        }
    } else {
        if (fl == 0) NT_LAUNCH(float, 0);
        else NT_LAUNCH(float, -1);
    }
#undef NT_LAUNCH
#undef NT_LAUNCH8
#undef NT_LAUNCH8_NT
    ECGVIT_CHECK_LAUNCH();
    if (d->epilogue & ECGVIT_EPI_COLSUM) {
        ecgvit_colsum_reduce_launch((const float *)d->workspace, 2 * tiles_m, d->N, d->colsum_out, s);
        ECGVIT_CHECK_LAUNCH();
    }
    return ECGVIT_OK;
}

// the four-wave body (bf16 products, plain or bias + residual [+ dropout], alpha 1; persistent grid or dispatcher-balanced chunks); diag: 2 = non-temporal output stores (plain); tools build: 1 = stamped
// instantiation with ablate bits (diag >> 2: 1 stores dropped, 2 no epilogue)
int ecgvit_gemm_nt4w_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag) {
    const EpiParams e = make_epi(d);
    constexpr int F_LIN = ECGVIT_EPI_BIAS | ECGVIT_EPI_RESIDUAL;
    const int fl = d->epilogue;
    if (d->dtype != ECGVIT_BF16 || d->out_dtype != ECGVIT_BF16 || d->K < 192 || e.alpha != 1.f || d->scale_a || d->scale_b) return ECGVIT_EINVAL;
    // (the FFN-down input gradient's body -- x aux, column sums -- measured 1,044 us on this body against 696: 288 B of spills, one wave's VALU)
    constexpr int F_DH = ECGVIT_EPI_MUL_AUX | ECGVIT_EPI_COLSUM;
    if (fl != 0 && fl != F_LIN && fl != (F_LIN | ECGVIT_EPI_DROPOUT) && fl != F_DH) return ECGVIT_EINVAL;
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, ntile = tiles_m * tiles_n;
    const int G = raster_g > 0 ? std::min(raster_g, tiles_n) : nt_default_group(tiles_n);
    const int tpw = d->tiles_per_workgroup;   // > 0: dispatcher-balanced chunks of ~tpw tiles, as in ecgvit_gemm_nt_launch
    const dim3 grid((unsigned)(tpw > 0 ? std::max(std::min(ntile, 256), (ntile + tpw - 1) / tpw) : std::min(ntile, 256))), block(256);
#define NT4W_GO(FL, CAUX, ST, AB) hipLaunchKernelGGL((gemm_nt_kernel_4w<bf16_t, FL, CAUX, ST>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, AB)
#ifdef ECGVIT_TOOLS
    if ((diag & 1) && fl == 0) {
        if (diag & 2) NT4W_GO(0, 2, true, (diag >> 2) & 3);
        else NT4W_GO(0, 0, true, (diag >> 2) & 3);
        ECGVIT_CHECK_LAUNCH();
        return ECGVIT_OK;
    }
#endif
    if (fl == F_LIN) NT4W_GO(F_LIN, 0, false, 0);
    else if (fl == (F_LIN | ECGVIT_EPI_DROPOUT)) NT4W_GO(F_LIN | ECGVIT_EPI_DROPOUT, 0, false, 0);
    else if (fl == F_DH) NT4W_GO(F_DH, 0, false, 0);
    else if (diag & 2) NT4W_GO(0, 2, false, 0);
    else NT4W_GO(0, 0, false, 0);
#undef NT4W_GO
    ECGVIT_CHECK_LAUNCH();
    if (fl & ECGVIT_EPI_COLSUM) {
        ecgvit_colsum_reduce_launch((const float *)d->workspace, 2 * tiles_m, d->N, d->colsum_out, s);
        ECGVIT_CHECK_LAUNCH();
    }
    return ECGVIT_OK;
}

#ifdef ECGVIT_TOOLS
extern "C" int ecgvit_tools_rowaffine(const float *row_a, const float *row_b, const float *col_g) {   // operands of the LayerNorm-fold pricing bodies
    return (hipMemcpyToSymbol(HIP_SYMBOL(g_ra_a), &row_a, sizeof(row_a)) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_ra_b), &row_b, sizeof(row_b)) == hipSuccess &&
            hipMemcpyToSymbol(HIP_SYMBOL(g_ra_g), &col_g, sizeof(col_g)) == hipSuccess) ? ECGVIT_OK : ECGVIT_ELAUNCH;
}
// stand-in for a collective's kernel: n workgroups that each hold a whole CU (all of its LDS) for `cycles` shader cycles
__global__ __launch_bounds__(64) void tools_occupy_kernel(unsigned long long cycles, unsigned int *done) {
    __shared__ char hold[LDS_BYTES];
    hold[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && hold[0]) atomicAdd(done, 1u);
}
extern "C" int ecgvit_tools_occupy(int n_cus, unsigned long long cycles, unsigned int *done, void *stream) {
    hipLaunchKernelGGL(tools_occupy_kernel, dim3(n_cus), dim3(64), 0, as_stream(stream), cycles, done);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

extern "C" int ecgvit_tools_nt_stamps(unsigned long long *h_out) {   // host buffer of 256*8 words
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_nt_stamps), sizeof(unsigned long long) * 256 * 8) == hipSuccess ? ECGVIT_OK : ECGVIT_ELAUNCH;
}
#endif
