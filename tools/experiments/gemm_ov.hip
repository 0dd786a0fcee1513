// gemm_ov_kernel: the A . B^T product of the two FFN-wide launches whose EPILOGUE, not their main loop, sets their time -- the FFN-up forward
// (bias + erf-GELU + saved GELU' x mask + dropout) and the FFN-down input gradient (x saved tensor + column sums) -- with the epilogue of
// output tile T spread INTO the main loop of tile T+1, so that the vector pipe works in the shadow of the matrix pipe instead of after it.
//
//   * four waves, one per SIMD (512 registers per lane), 256 x 256 x 64 block tile, 128 x 128 per wave as 4 x 4 blocks of
//     v_mfma_f32_32x32x16_bf16 in 256 accumulator registers (AGPRs); the same LDS images, rings (activations 3 slots per half, weights 2) and
//     continuous operand stream as gemm_nt_kernel_4w, here with ONE image swizzle for both operands (16-B chunk ^= (row >> 1) & 7: conflict-free
//     for the 32-row fragment reads of either operand) and the workgroup's one barrier per K-tile between its third and fourth 16-deep slice.
//   * the MFMA takes the WEIGHT fragment first; lane rho of the weight fragment reads weight row pi(rho) = rho with bits 2 and 3 swapped, which
//     makes the 16 accumulator registers of a lane (one output row m = lane & 31) two runs of 8 consecutive columns: 8Q.. and 16 + 8Q..
//     (Q = lane >> 5).  A UNIT of epilogue work = one run of one block = 8 elements per lane = one 16-B store per output tensor.
//   * at a tile boundary (the first slice of the next tile, whose MFMAs take srcC = 0) the 256 accumulators are packed into a 128-register
//     STASH -- f16 pairs for the GELU body (v_cvt_pkrtz_f16_f32: 11 significant bits, saturating; the body's arithmetic is packed f16, two
//     elements per instruction, because ONE wave per SIMD issues a vector instruction every ~5.7 cycles whatever its type: instruction COUNT
//     is the cost, profiles/r04_valu_rate.txt), bf16 pairs for the gradient body (f16 would underflow: gradients carry 1 / (B K)) -- and the
//     32 units of the stash are worked off in fixed positions between the MFMAs of the next tile's K-tiles (UPK = ceil(32 / nk) unit slots per
//     K-tile; every slot issues its vector-memory instructions whether it holds a unit or not, so every counted vmcnt wait is a constant).
//     The FFN-up bias enters through the matrix pipe: 16 MFMAs per tile multiply a (bias_hi, bias_lo) bf16 pair by a fragment of ones.
//   * numerics: the pre-activation is rounded ONCE more than in the eight-wave body (to f16 / bf16 in the stash) before the epilogue
//     arithmetic -- what an unfused bf16 pipeline (Linear output in bf16, then GELU) does as well; erf by Abramowitz-Stegun 7.1.25
//     (|err| <= 2.5e-5, three terms) on packed f16 with the small tail Phi(-|x|) selected, never formed by cancellation; products x * Phi,
//     Phi + x * phi in f32 (v_fma_mix_f32).  Tests hold the body against the f32 formulation at bf16 resolution (tests/test_gpu_ops.py).
//     Dropout at this site is a PRIVATE mask (it reaches the backward pass only through the saved tensor): one hash per four consecutive
//     elements, 7-bit thresholds (p applied as round(128 p) / 128, rescaled by the exact keep rate).
#include "common.h"
#include "nt_tiles.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

using nt_tiles::decode_tile;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 16384;
constexpr int LDS_BYTES = 163840;
constexpr uint32_t OV_OOB = 0x80000000u;
constexpr int MODE_DH = 0, MODE_UP = 1;

typedef __attribute__((address_space(3))) void *lptr_t;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int... X, typename F>
__device__ __forceinline__ void ov_static_for_impl(std::integer_sequence<int, X...>, F &&f) { (f(std::integral_constant<int, X>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ov_static_for(F &&f) { ov_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ uint32_t ov_pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float ov_bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float ov_bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t ov_f16x2(float v) {
    const _Float16 h = (_Float16)v;
    const f16x2 p = {h, h};
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ f16x2 H2(uint32_t w) { return __builtin_bit_cast(f16x2, w); }
__device__ __forceinline__ uint32_t U2(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// ---- schedule of one K-tile: 64 MFMA gaps (slice S = gap >> 4, block X = gap & 15: i = X >> 2, j = X & 3) ------------------------------------
//   fragment reads (each as late as its latency allows: a lone wave has 256 registers for stash + fragments + unit):
//     X = 3, 7      activation blocks 2, 3 of THIS slice (used from X = 8 / 12)
//     X = 9 .. 12   weight blocks 0 .. 3 of the NEXT slice (second register set), X = 13, 14 activation blocks 0, 1 of the next slice
//   the workgroup's barrier follows gap 55 (slice 3, X = 7): every read of K-tile t has been issued (fast waves send A(t+3) into A(t)'s slot
//   from the next K-tile's first gaps on), K-tile t+1 is visible for the reads of gaps 57 .. 62, B(t+2) may overwrite B(t)
//   DMA pieces: A(t+2) -- 8 pieces -- at gaps 2, 9, .., 51; B(t+2) at gaps 56 .. 63
__device__ constexpr int OV_BAR_GAP = 55;
__device__ constexpr int ov_dmaA_at(int g) { return (g <= 51 && g % 7 == 2) ? g / 7 : -1; }
__device__ constexpr int ov_dmaB_at(int g) { return g >= 56 ? g - 56 : -1; }

// unit slots of a K-tile with UPK slots: slot q owns gaps [q * 64 / UPK, (q + 1) * 64 / UPK); its NA atoms are spread evenly over them
__device__ constexpr int ov_slot_lo(int UPK, int q) { return q * 64 / UPK; }
__device__ constexpr int ov_slot_of(int UPK, int g) { int q = 0; for (int k = 1; k < UPK; ++k) if (g >= ov_slot_lo(UPK, k)) q = k; return q; }
__device__ constexpr int ov_atom_lo(int UPK, int NA, int g) {   // first atom of gap g inside its slot
    const int q = ov_slot_of(UPK, g), lo = ov_slot_lo(UPK, q), hi = ov_slot_lo(UPK, q + 1), n = hi - lo;
    return (g - lo) * NA / n;
}
__device__ constexpr int ov_atom_hi(int UPK, int NA, int g) {
    const int q = ov_slot_of(UPK, g), lo = ov_slot_lo(UPK, q), hi = ov_slot_lo(UPK, q + 1), n = hi - lo;
    return (g + 1 - lo) * NA / n;
}

// gap in which atom `a` of slot q runs
__device__ constexpr int ov_atom_gap(int UPK, int NA, int q, int a) {
    int gg = ov_slot_lo(UPK, q);
    for (int x = ov_slot_lo(UPK, q); x < ov_slot_lo(UPK, q + 1); ++x)
        if (ov_atom_lo(UPK, NA, x) <= a && a < ov_atom_hi(UPK, NA, x)) gg = x;
    return gg;
}
// vector-memory instructions the slots issue up to the barrier gap (atoms a0, a0 + 1, .. of every slot are its VMEM instructions)
__device__ constexpr int ov_vmem_upto_bar(int UPK, int NA, int a0, int n) {
    int c = 0;
    for (int q = 0; q < UPK; ++q)
        for (int w = 0; w < n; ++w) c += ov_atom_gap(UPK, NA, q, a0 + w) <= OV_BAR_GAP ? 1 : 0;
    return c;
}

// ---- the two unit bodies ---------------------------------------------------------------------------------------------------------------------
struct OvBufs {
    __amdgpu_buffer_rsrc_t c, aux, part;   // output, saved tensor, column-sum partials
    int ldc2, ldx2;                        // row pitches in bytes
    int t_out, t_in;                       // ds_bpermute byte addresses: accumulator layout -> memory layout (lane t: row t >> 1, run t & 1) and back
    int M, N;
};

// FFN-up: constants of the packed-f16 GELU (k = dropout rescale, folded in)
struct UpConst {
    uint32_t P2, ONE, C1, A1, A2, A3, K, CK;   // f16 pairs
    uint32_t seedmix, c4;                      // dropout: mixed seed, (0x80 - t7) in every byte
};

// per-unit registers of the FFN-up body
struct UpState {
    uint32_t X[4], T[4], EX[4], CDF[4], Yb[4], Db[4], KM[4];
    uint32_t h0, h1;
    u32x4 ym, dm;
    uint32_t offc, offx;
};
constexpr int UP_NATOM = 44;

// atom A of the FFN-up unit (8 elements = 4 f16 pairs).  The sequence 0 .. UP_NATOM-1 is a correct sequential program; the caller spreads it over
// the MFMA gaps of the unit's slot.  Atoms 0 .. 31 are the GELU arithmetic STAGE-major (atom = 4 stage + pair): neighbouring atoms work on
// different pairs, so dependent instructions (and the transcendental unit's issue hazards) are four atoms apart instead of back to back.
template <int A, bool DROP>
__device__ __forceinline__ void up_atom(UpState &s, const UpConst &k, const OvBufs &bf, uint32_t row_l, uint32_t col_l, uint32_t row_m, uint32_t col_m) {
    if constexpr (A < 32) {
        constexpr int st = A >> 2, p = A & 3;
        if constexpr (st == 0) {          // t's denominator 1 + p |x| / sqrt 2
            const uint32_t ax = s.X[p] & 0x7FFF7FFFu;
            s.T[p] = U2(H2(ax) * H2(k.P2) + H2(k.ONE));
        } else if constexpr (st == 1) {
            const uint32_t den = s.T[p];
            uint32_t t;
            asm("v_rcp_f16_e32 %0, %1" : "=v"(t) : "v"(den));
            asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(t) : "v"(den));
            s.T[p] = t;
        } else if constexpr (st == 2) {   // the exponent: -x^2 log2(e) / 2
            const f16x2 u = H2(s.X[p]) * H2(k.C1);
            s.EX[p] = U2(u * -u);
        } else if constexpr (st == 3) {
            const uint32_t arg = s.EX[p];
            uint32_t ex;
            asm("v_exp_f16_e32 %0, %1" : "=v"(ex) : "v"(arg));
            asm("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(ex) : "v"(arg));
            s.EX[p] = ex;
        } else if constexpr (st == 4) {   // the three-term polynomial in t (coefficients carry k / 2)
            const f16x2 t = H2(s.T[p]);
            f16x2 pl = t * H2(k.A3) + H2(k.A2);
            pl = pl * t + H2(k.A1);
            s.T[p] = U2(pl * t);
        } else if constexpr (st == 5) {   // Q = k Phi(-|x|): the small tail, formed without cancellation; k Phi(x) = x < 0 ? Q : k - Q
            const f16x2 q = H2(s.T[p]) * H2(s.EX[p]);
            const uint32_t kq = U2(H2(k.K) - q);
            const uint32_t sg = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, s.X[p]) >> (s16x2){15, 15});   // 0xFFFF where x < 0
            s.CDF[p] = (sg & U2(q)) | (~sg & kq);   // v_bfi_b32
        } else if constexpr (st == 6) {   // y = x k Phi(x), in f32 from the f16 factors
            float y0, y1;
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(y0) : "v"(s.X[p]), "v"(s.CDF[p]));
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(y1) : "v"(s.X[p]), "v"(s.CDF[p]));
            s.Yb[p] = ov_pack_bf16x2(y0, y1);
        } else {                          // dy = k (Phi(x) + x phi(x))
            const uint32_t pdf = U2(H2(s.EX[p]) * H2(k.CK));
            float d0, d1;
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,1]" : "=v"(d0) : "v"(s.X[p]), "v"(pdf), "v"(s.CDF[p]));
            asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1]" : "=v"(d1) : "v"(s.X[p]), "v"(pdf), "v"(s.CDF[p]));
            s.Db[p] = ov_pack_bf16x2(d0, d1);
        }
    } else if constexpr (A == 32) {
        if constexpr (DROP) {   // one hash per FOUR consecutive elements: quads (row N + col) / 4 and the next
            const uint32_t quad = (row_l * (uint32_t)bf.N + col_l) >> 2;
            const uint32_t b = quad * ECGVIT_WEYL + k.seedmix;
            s.h0 = pair_finish(b);
            s.h1 = b + ECGVIT_WEYL;
        }
    } else if constexpr (A == 33) {
        if constexpr (DROP) {
            s.h1 = pair_finish(s.h1);
            // keep bit = bit 7 of every byte of (r7 + 0x80 - t7): set iff the 7-bit draw r7 >= t7
            s.h0 = (s.h0 & 0x7F7F7F7Fu) + k.c4;
            s.h1 = (s.h1 & 0x7F7F7F7Fu) + k.c4;
        }
    } else if constexpr (A == 34) {
        if constexpr (DROP) {   // keep masks of the four pairs: byte b's bit 7 spread over halfword b
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t h = p < 2 ? s.h0 : s.h1;
                const uint32_t m = __builtin_amdgcn_perm(h, h, (p & 1) ? 0x030C020Cu : 0x010C000Cu);
                s.KM[p] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, m) >> (s16x2){15, 15});
            }
        }
    } else if constexpr (A == 35) {
        if constexpr (DROP) {
#pragma unroll
            for (int p = 0; p < 4; ++p) { s.Yb[p] &= s.KM[p]; s.Db[p] &= s.KM[p]; }
        }
        s.offc = row_m * (uint32_t)bf.ldc2 + col_m * 2;
        s.offx = row_m * (uint32_t)bf.ldx2 + col_m * 2;
    } else if constexpr (A >= 36 && A < 40) {   // one crossbar move per atom: the LDS pipe takes ~24 cycles per ds_bpermute from a lone wave
        constexpr int p = A - 36;
        s.ym[p] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, (int)s.Yb[p]);
    } else if constexpr (A >= 40 && A < 44) {
        constexpr int p = A - 40;
        s.dm[p] = (uint32_t)__builtin_amdgcn_ds_bpermute(bf.t_out, (int)s.Db[p]);
    }
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------------
template <int MODE, int UPK, bool DROP, bool NOEPI = false>   // NOEPI (diagnostics): the main loop alone -- no unit arithmetic; the slots still issue their out-of-bounds stores
__global__ __launch_bounds__(256, 1) void gemm_ov_kernel(ecgvit_gemm_desc d, EpiParams e, int tiles_m, int tiles_n, int ngroup, int nitems, int nfull) {
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
    static_assert(MODE == MODE_UP, "the gradient body is not built yet");
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
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const uint64_t pbias = reinterpret_cast<uint64_t>(e.bias);
    const i32x4_t rsBias = {(int)(uint32_t)pbias, (int)((pbias >> 32) & 0xFFFFu), e.bias ? N * 4 : 0, 0x00020000};
    OvBufs bf;
    bf.M = M; bf.N = N;
    bf.ldc2 = (int)d.ldc * 2; bf.ldx2 = (int)e.ldaux * 2;
    bf.c = __builtin_amdgcn_make_buffer_rsrc(d.C, 0, (uint32_t)((int64_t)M * bf.ldc2), 0x00020000);
    bf.aux = __builtin_amdgcn_make_buffer_rsrc(e.aux, 0, e.aux ? (uint32_t)((int64_t)M * bf.ldx2) : 0u, 0x00020000);
    bf.part = bf.aux;
    bf.t_out = ((lane >> 1) + 32 * (lane & 1)) << 2;    // memory-layout lane t takes its run from accumulator-layout lane (t >> 1) + 32 (t & 1)
    bf.t_in = (2 * (lane & 31) + (lane >> 5)) << 2;     // and back
    const int g = lane & 31, Q = lane >> 5;
    const uint32_t qmask = Q ? 0u : 0xFFFFFFFFu;

    UpConst kc;
    {   // wave-uniform: in SGPRs (one constant-bus operand per packed instruction; A2 stays a vector register: its fma takes A3 as well)
        const float kk = DROP ? e.inv_keep : 1.0f;
        auto sg = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        kc.P2 = sg(ov_f16x2(0.47047f * 0.70710678f)); kc.ONE = sg(ov_f16x2(1.0f)); kc.C1 = sg(ov_f16x2(0.84932180f));
        kc.A1 = sg(ov_f16x2(0.5f * 0.3480242f * kk)); kc.A2 = ov_f16x2(0.5f * -0.0958798f * kk); kc.A3 = sg(ov_f16x2(0.5f * 0.7478556f * kk));
        kc.K = sg(ov_f16x2(kk)); kc.CK = sg(ov_f16x2(0.39894228f * kk));
        kc.seedmix = sg(seed_mix(e.seed));
        kc.c4 = sg((0x80u - (e.drop_thresh & 0x7Fu)) * 0x01010101u);   // drop_thresh: the 7-bit threshold (launcher: round(128 p))
    }

    // this wave's four DMA pieces of a half-tile: rows 32 wave + 8 i + (lane >> 3); LDS chunk p of row r holds source chunk p ^ ((r >> 1) & 7)
    // (lda == ldb: both operands are [rows][K] with pitch K -- checked by the launcher -- so ONE set of per-lane offsets serves both streams)
    int voA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * wave + 8 * i + (lane >> 3), p = lane & 7;
        voA[i] = r * lda2 + ((p ^ ((r >> 1) & 7)) << 4);
    }
    // fragment read offsets inside a half-tile image (block b adds 4096 b, slice s is ^ (s << 5)): activations row g, weights row pi(g)
    const int pg = (g & 0x13) | ((g & 4) << 1) | ((g & 8) >> 1);
    const int roffA = g * 128 + ((Q ^ ((g >> 1) & 7)) << 4);
    const int roffB = pg * 128 + ((Q ^ ((pg >> 1) & 7)) << 4);

#define O_DMA_A(h, ring, soff)                                                                                \
    do {                                                                                                      \
        char *dst_ = smem + (3 * (h) + (ring)) * HALF_BYTES + wave * 4096;                                    \
        const int so_ = (soff) + (h) * 128 * lda2;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(dst_ + 1024 * i_), 16, voA[i_], so_, 0, 0); \
    } while (0)
#define O_DMA_B(h, ring, soff)                                                                                \
    do {                                                                                                      \
        char *dst_ = smem + (6 + 2 * (h) + (ring)) * HALF_BYTES + wave * 4096;                                \
        const int so_ = (soff) + (h) * 128 * ldb2;                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(dst_ + 1024 * i_), 16, voA[i_], so_, 0, 0); \
    } while (0)
#define O_FENCE() __builtin_amdgcn_sched_barrier(0)

    int cm0, cn0, nm0, nn0;
    decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
    nm0 = cm0; nn0 = cn0;
    // producer cursors of the continuous operand stream; past the end of this workgroup's share they stay where they are (harmless re-reads)
    int a_it = it, a_kt = 0, a_base = cm0 * lda2;
    int b_it = it, b_kt = 0, b_base = cn0 * ldb2;
#define O_ADV_A()                                                                                             \
    do {                                                                                                      \
        if (++a_kt == nk) {                                                                                   \
            a_kt = 0;                                                                                         \
            if (a_it + (int)gridDim.x < nitems) { a_it += (int)gridDim.x; decode_tile(a_it, ntile, tiles_m, tiles_n, ngroup, nm0, nn0); a_base = nm0 * lda2; } \
        }                                                                                                     \
    } while (0)
#define O_ADV_B()                                                                                             \
    do {                                                                                                      \
        if (++b_kt == nk) {                                                                                   \
            b_kt = 0;                                                                                         \
            if (b_it + (int)gridDim.x < nitems) { b_it += (int)gridDim.x; b_base = nn0 * ldb2; }              \
        }                                                                                                     \
    } while (0)

    // ---- prologue: A(0), B(0), A(1), B(1); the first slice's fragments behind the barrier
    O_DMA_A(0, 0, a_base); O_DMA_A(1, 0, a_base); O_ADV_A();
    O_DMA_B(0, 0, b_base); O_DMA_B(1, 0, b_base); O_ADV_B();
    O_DMA_A(0, 1, a_base + a_kt * (BK * 2)); O_DMA_A(1, 1, a_base + a_kt * (BK * 2)); O_ADV_A();
    O_DMA_B(0, 1, b_base + b_kt * (BK * 2)); O_DMA_B(1, 1, b_base + b_kt * (BK * 2)); O_ADV_B();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 af[4], bw[2][4];   // fragments: weight blocks in two sets (slice S multiplies set S & 1), activation blocks reloaded in place
    f32x16 acc[4][4];
    // the stash: unit U = 8 j + 4 h + i (block (i, j), run h) holds its four pair registers in sq[0 .. 3][U].  Four 32-register vectors: the unit
    // of a slot is a run-time index (ubase + slot), read with the VGPR index mode (s_set_gpr_idx_on + v_mov) -- a switch over 32 cases splits
    // every live range of the main loop around it, an indexed local array lives in scratch memory
    typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));
    u32x32 sq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 32; ++u) sq[k][u] = 0u;
    {   // the first slice's weight blocks and activation blocks 0, 1 (what gaps 57 .. 62 of a K-tile read for the next one)
        const int sa = (3 * wm) * HALF_BYTES + roffA, sb = (6 + 2 * wn) * HALF_BYTES + roffB;
#pragma unroll
        for (int b = 0; b < 4; ++b) bw[0][b] = *reinterpret_cast<const bf16x8 *>(smem + sb + 4096 * b);
        af[0] = *reinterpret_cast<const bf16x8 *>(smem + sa);
        af[1] = *reinterpret_cast<const bf16x8 *>(smem + sa + 4096);
        af[2] = af[0]; af[3] = af[0];
    }
    int ga = 0, gb = 0;          // ring slots of the K-tile being multiplied
    int pm0 = 0, pn0 = 0;        // origin of the tile the stash belongs to
    bool have = false;           // the stash holds a tile (false until the first tile of this workgroup is done)
    UpState us;
    int ubase = 0;               // first unit of the coming K-tile

    // VMEM instructions of one K-tile: 16 pieces + 2 stores per slot (issued by empty slots as well, out of bounds).  The wait in front of the
    // barrier (behind gap 47) must cover B(t+1)'s last piece, issued at gap 63 of the K-tile before: behind it came A(t+2)'s 8 pieces and the
    // stores of the slots' store gaps <= 47 (none may sit behind the piece of gap 63 inside that gap: checked below)
    constexpr int NA = UP_NATOM + 3;   // atoms of a slot: 0 = stash fetch, 1 .. UP_NATOM = up_atom<0 ..>, then the two stores
    constexpr int NBAR = 8 + ov_vmem_upto_bar(UPK, NA, UP_NATOM + 1, 2);
    static_assert(NBAR <= 63, "vmcnt range");

    // the unit of slot q in this K-tile (U = ubase + q; active iff the stash holds a tile and q < NU)
    auto unit_coords = [&](int U, uint32_t &row_l, uint32_t &col_l, uint32_t &row_m, uint32_t &col_m) __attribute__((always_inline)) {
        const int j = U >> 3, h = (U >> 2) & 1, i = U & 3;
        const int rb = pm0 + 128 * wm + 32 * i, cb = pn0 + 128 * wn + 32 * j + 16 * h;
        row_l = (uint32_t)(rb + g); col_l = (uint32_t)(cb + 8 * Q);
        row_m = (uint32_t)(rb + (lane >> 1)); col_m = (uint32_t)(cb + 8 * (lane & 1));
    };
    auto fetch = [&](int U) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) us.X[k] = sq[k][U];
    };

    // one K-tile of the continuous stream.  KFIRST: first K-tile of an output tile -- its first slice packs the finished accumulators into the
    // stash block by block and restarts them (srcC = 0); NU: slots of this K-tile that hold a unit
    auto ktile = [&](auto first_c, auto nu_c) __attribute__((always_inline)) {
        constexpr bool KFIRST = decltype(first_c)::value;
        constexpr int NU = decltype(nu_c)::value;
        const int ga1 = ga == 2 ? 0 : ga + 1, ga2 = ga == 0 ? 2 : ga - 1, gb1 = gb ^ 1;
        const int sa = (3 * wm + ga) * HALF_BYTES + roffA, sb = (6 + 2 * wn + gb) * HALF_BYTES + roffB;
        const int sa1 = (3 * wm + ga1) * HALF_BYTES + roffA, sb1 = (6 + 2 * wn + gb1) * HALF_BYTES + roffB;
        char *dA0 = smem + ga2 * HALF_BYTES + wave * 4096, *dA1 = smem + (3 + ga2) * HALF_BYTES + wave * 4096;
        char *dB0 = smem + (6 + gb) * HALF_BYTES + wave * 4096, *dB1 = smem + (8 + gb) * HALF_BYTES + wave * 4096;
        const int soA = a_base + a_kt * (BK * 2), soB = b_base + b_kt * (BK * 2);
        uint32_t bl[4] = {0u, 0u, 0u, 0u};
        if constexpr (KFIRST) {   // this tile's bias values (one per weight-fragment row and n-block), consumed by the bias MFMAs behind gap 63
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t off = (uint32_t)(cn0 + 128 * wn + 32 * j + pg) * 4u;
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(bl[j]) : "v"(off), "s"(rsBias) : "memory");
            }
        }
        uint32_t row_l = 0, col_l = 0, row_m = 0, col_m = 0;
        bool act = false;
        O_FENCE();
        ov_static_for<64>([&](auto gc) __attribute__((always_inline)) {
            constexpr int G = decltype(gc)::value, S = G >> 4, X = G & 15, I = X >> 2, J = X & 3;
            constexpr int SET = S & 1, NSET = SET ^ 1;
            if constexpr (KFIRST && S == 0) {   // a new output tile: srcC = 0 (no accumulator is carried across the tile boundary: see the pack below)
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[SET][J], af[I], z, 0, 0, 0);
            } else {
                acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[SET][J], af[I], acc[I][J], 0, 0, 0);
            }
            O_FENCE();
            {   // fragment reads of this gap (see the schedule above); slice 3 reads its successor from K-tile t+1 (behind the barrier)
                constexpr int NS = (S + 1) & 3;
                const int base_a = S == 3 ? sa1 : sa, base_b = S == 3 ? sb1 : sb;
                if constexpr (X == 3 || X == 7) {
                    constexpr int B = X == 3 ? 2 : 3;
                    af[B] = *reinterpret_cast<const bf16x8 *>(smem + ((sa + 4096 * B) ^ (S << 5)));
                    O_FENCE();
                } else if constexpr (X >= 9 && X <= 12) {
                    bw[NSET][X - 9] = *reinterpret_cast<const bf16x8 *>(smem + ((base_b + 4096 * (X - 9)) ^ (NS << 5)));
                    O_FENCE();
                } else if constexpr (X == 13 || X == 14) {
                    af[X - 13] = *reinterpret_cast<const bf16x8 *>(smem + ((base_a + 4096 * (X - 13)) ^ (NS << 5)));
                    O_FENCE();
                }
            }
            // ---- the slot's atoms of this gap (ahead of the gap's DMA piece: the piece of gap 63 must be the K-tile's last VMEM instruction)
            constexpr int SQ = ov_slot_of(UPK, G), ALO = ov_atom_lo(UPK, NA, G), AHI = ov_atom_hi(UPK, NA, G);
            ov_static_for<AHI - ALO>([&](auto ac) __attribute__((always_inline)) {
                constexpr int AT = ALO + decltype(ac)::value;
                if constexpr (AT == 0) {
                    act = have && SQ < NU;
                    if constexpr (SQ < NU) {
                        const int U = ubase + SQ;
                        unit_coords(U, row_l, col_l, row_m, col_m);
                        fetch(U);
                    }
                } else if constexpr (AT <= UP_NATOM) {
                    if constexpr (SQ < NU) up_atom<AT - 1, DROP>(us, kc, bf, row_l, col_l, row_m, col_m);
                } else if constexpr (AT == UP_NATOM + 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(us.ym, bf.c, act ? us.offc : OV_OOB, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(us.dm, bf.aux, act ? us.offx : OV_OOB, 0, 0);
                }
            });
            O_FENCE();
            constexpr int PA = ov_dmaA_at(G), PB = ov_dmaB_at(G);
            if constexpr (PA >= 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)((PA < 4 ? dA0 : dA1) + 1024 * (PA & 3)), 16, voA[PA & 3], soA + (PA < 4 ? 0 : 128 * lda2), 0, 0);
                O_FENCE();
            }
            if constexpr (PB >= 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)((PB < 4 ? dB0 : dB1) + 1024 * (PB & 3)), 16, voA[PB & 3], soB + (PB < 4 ? 0 : 128 * ldb2), 0, 0);
                O_FENCE();
            }
            if constexpr (G == OV_BAR_GAP) {   // K-tile t+1 has landed (this wave's pieces; the barrier publishes everyone's), every read of K-tile t is out
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBAR) : "memory");
                O_FENCE();
                __builtin_amdgcn_s_barrier();
                O_FENCE();
            }
        });
        O_ADV_A();
        O_ADV_B();
        if constexpr (KFIRST) {
            // the bias, through the matrix pipe: weight-side fragment = (bias_hi, bias_lo) bf16 in k = 0, 1 of lanes Q = 0, activation-side = ones
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3])::"memory");
            uint32_t o0, oz;   // (built here, opaquely: as a loop invariant the fragment is spilled and its reload drains the operand stream)
            asm volatile("v_and_b32 %0, 0x3f803f80, %1" : "=v"(o0) : "v"(qmask));
            asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
            const u32x4 of = {o0, oz, oz, oz};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = __builtin_bit_cast(float, bl[j]);
                const float bh = ov_bf16_lo(ov_pack_bf16x2(b, 0.f));
                const u32x4 wf = {ov_pack_bf16x2(bh, b - bh) & qmask, oz, oz, oz};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, of), acc[i][j], 0, 0, 0);
            }
            O_FENCE();
        }
        ubase += NU;
        ga = ga1; gb = gb1;
    };

    int qm0 = 0, qn0 = 0;
    for (;;) {
        // ---- one output tile: nk K-tiles; the stash (the tile before) is worked off in their unit slots: nfull K-tiles hold UPK units, the rest UPK - 1
        ubase = 0;
        pm0 = qm0; pn0 = qn0;
        constexpr int NUF = NOEPI ? 0 : UPK, NUS = NOEPI ? 0 : UPK - 1;
        ktile(std::true_type{}, std::integral_constant<int, NUF>{});
        // (two loops in sequence, not one loop with a branch: a diamond inside the loop makes hipcc copy accumulator blocks between its arms)
#pragma unroll 1
        for (int kt = 1; kt < nfull; ++kt) ktile(std::false_type{}, std::integral_constant<int, NUF>{});
#pragma unroll 1
        for (int kt = nfull; kt < nk; ++kt) ktile(std::false_type{}, std::integral_constant<int, NUS>{});
        // ---- tile done: its 256 accumulators -> the stash (f16 pairs, round toward zero: saturates instead of overflowing).  In ONE piece, behind
        // the tile's last MFMA: packed block by block under the next tile's first slice it would hide 16 MFMAs (~1 % of a tile), but every
        // accumulator would then be live across the tile boundary next to its successor, and hipcc spills them by the dozen
        O_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2)
                        sq[k2][8 * j + 4 * h + i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(acc[i][j][8 * h + 2 * k2], acc[i][j][8 * h + 2 * k2 + 1]));
        O_FENCE();
        qm0 = cm0; qn0 = cn0; have = true;
        const int next_it = it + (int)gridDim.x;
        if (next_it >= nitems) break;
        it = next_it;
        decode_tile(it, ntile, tiles_m, tiles_n, ngroup, cm0, cn0);
    }
    // ---- the last tile of this workgroup: no main loop left to hide its epilogue in
    pm0 = qm0; pn0 = qn0;
#pragma unroll 1
    for (int U = 0; U < 32; ++U) {
        uint32_t row_l, col_l, row_m, col_m;
        unit_coords(U, row_l, col_l, row_m, col_m);
        fetch(U);
        ov_static_for<UP_NATOM>([&](auto ac) __attribute__((always_inline)) { up_atom<decltype(ac)::value, DROP>(us, kc, bf, row_l, col_l, row_m, col_m); });
        __builtin_amdgcn_raw_buffer_store_b128(us.ym, bf.c, us.offc, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(us.dm, bf.aux, us.offx, 0, 0);
    }
    // nothing of this wave's may still be on its way into LDS when the workgroup's allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef O_DMA_A
#undef O_DMA_B
#undef O_ADV_A
#undef O_ADV_B
#undef O_FENCE
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------------
// host side.  The caller has run ecgvit_gemm_nt_applicable(); this body takes the FFN-up forward's flag set on whole 256-column tiles
bool ecgvit_gemm_ov_applicable(const ecgvit_gemm_desc *d) {
    constexpr int F_UP = ECGVIT_EPI_BIAS | ECGVIT_EPI_GELU | ECGVIT_EPI_GELU_GRAD_AUX;
    if (d->dtype != ECGVIT_BF16 || d->out_dtype != ECGVIT_BF16 || d->layout != ECGVIT_GEMM_NT) return false;
    if ((d->epilogue & ~ECGVIT_EPI_DROPOUT) != F_UP) return false;
    if (d->N % BN != 0 || d->K % BK != 0 || d->K < 8 * BK || d->M < 2048 || d->lda != d->ldb) return false;
    if (d->alpha != 1.f || d->scale_a || d->scale_b || !d->C || !d->aux || !d->bias) return false;
    if ((d->epilogue & ECGVIT_EPI_DROPOUT) && d->dropout_p > 0.f) {
        const int t7 = (int)(d->dropout_p * 128.0 + 0.5);
        if (t7 < 1 || t7 > 127) return false;   // p below 1/256 or above 127/128: the eight-wave body's 16-bit mask
    }
    return true;
}

int ecgvit_gemm_ov_launch(const ecgvit_gemm_desc *d, hipStream_t s, int raster_g, int diag) {
    if (!ecgvit_gemm_ov_applicable(d)) return ECGVIT_EINVAL;
    EpiParams e = make_epi(d);
    const bool drop = (d->epilogue & ECGVIT_EPI_DROPOUT) && d->dropout_p > 0.f;
    if (drop) {   // this site's private mask: 7-bit thresholds, exact rescale of the rate actually applied
        const int t7 = (int)(d->dropout_p * 128.0 + 0.5);
        e.drop_thresh = (uint32_t)t7;
        e.inv_keep = 128.0f / (128.0f - (float)t7);
    }
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = d->N / BN, ntile = tiles_m * tiles_n;
    const int G = raster_g > 0 ? std::min(raster_g, tiles_n) : (tiles_n <= 8 ? tiles_n : 6);
    const int nk = d->K / BK;
    const int upk = (32 + nk - 1) / nk;            // unit slots per K-tile: 2 (K >= 1024), 3 (K = 704 .. 960), 4 (K = 512 .. 640)
    const int nfull = nk - (nk * upk - 32);        // K-tiles whose slots all hold a unit; the others leave their last slot empty
    const int tpw = d->tiles_per_workgroup;
    const dim3 grid((unsigned)(tpw > 0 ? std::max(std::min(ntile, 256), (ntile + tpw - 1) / tpw) : std::min(ntile, 256))), block(256);
#define OV_GO(UPK, DR) hipLaunchKernelGGL((gemm_ov_kernel<MODE_UP, UPK, DR>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, nfull)
#ifdef ECGVIT_TOOLS
    if ((diag & 1) && upk == 3) { hipLaunchKernelGGL((gemm_ov_kernel<MODE_UP, 3, true, true>), grid, block, 0, s, *d, e, tiles_m, tiles_n, G, ntile, nfull); ECGVIT_CHECK_LAUNCH(); return ECGVIT_OK; }
#endif
    if (upk == 2) { if (drop) OV_GO(2, true); else OV_GO(2, false); }
    else if (upk == 3) { if (drop) OV_GO(3, true); else OV_GO(3, false); }
    else if (upk == 4) { if (drop) OV_GO(4, true); else OV_GO(4, false); }
    else return ECGVIT_EINVAL;
#undef OV_GO
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}
