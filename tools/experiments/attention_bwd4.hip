// Fused attention backward, persistent, FOUR waves -- one per SIMD -- each owning TWO 32-key tiles (64 keys) of a 256-key window.
//
// Why this shape (profiles/r04_valu_rate.txt): the kernel is bound by vector-instruction ISSUE (softmax / dropout / dS arithmetic: a
// wave alone on its SIMD issues one plain vector instruction per 5.7 cycles, two waves sharing a SIMD 4.35 cycles per instruction
// between them; an MFMA costs the stream ~5 cycles of issue and 32 of matrix pipe).  The eight-wave kernel it replaces ran two waves
// per SIMD through the same phases between the same barriers (13.5 vector instructions per score element, 19 % of its time at the
// workgroup barrier, seven lgkmcnt(0) drains per block).  Here
//   * ONE instruction stream per SIMD carries, in a fixed order, the vector work of one tile and -- in its gaps -- the MFMAs of the
//     neighbouring pipeline stages: while the wave does the softmax arithmetic V(j, t0) of query block j on its first key tile, the
//     matrix pipe runs S / dP of (j, t1), the dV / dK products of (j-1, t1) and half of the dQ tile of block j-1; during V(j, t1):
//     S / dP of (j+1, t0), dV / dK of (j, t0), the other half of dQ(j-1).  Every quarter of a phase = one group of four queries:
//     12 fragment reads issued up front, then [hash | 4 x element | pack] with one MFMA behind each piece, counted lgkmcnt waits;
//   * 9.5 vector instructions per score element instead of 13.5: log2(1/(1-p)) folded into the stored -LSE (the exponential returns
//     p / (1-p) directly), delta pre-divided by the keep scale, `scale` moved out of the element path into the dK flush and the dQ
//     tile (exact for the power-of-two dh^-1/2), the drop decision as ONE byte-select compare + ONE select on a hash word whose
//     bytes were transposed inside the lane quad (2 DPP moves + 2 byte permutes per FOUR elements) so that each element's byte sits
//     at a compile-time position;
//   * one barrier per query block (dS crosses the waves for dQ through the [key][query] image, double-buffered), met by four
//     symmetric waves; Q / dO arrive as 32-query slabs through a five-slot ring DMA'd three blocks ahead across item boundaries, O
//     (needed only for delta) through its own three-slot ring; every wave computes delta for the eight rows it DMA'd itself;
//   * K^T for the dQ product comes from the double-buffered K image (next item's K prefetched during block 1); the S / dP operands
//     K, V of the wave's 64 keys stay in registers for the whole item.
// Math, LDS images, dropout function (common.h: quad_hash) and results as the one-item kernel in attention.hip (tests hold both
// against an fp64 reference and against each other).  N <= 512 keys run as two launches over 256-key windows (`k0`, ACCUM) as before.
#include "attn_common.h"

namespace {

typedef __attribute__((address_space(3))) void *lds_void_p4;

#ifdef ECGVIT_TOOLS
// tools build: cycle stamps of the second item of every workgroup (tools/attn_bwd4_stamps.py): [workgroup][wave][16 iterations][8] uint64;
// per iteration 0 top, 1 stream fed, 2 phase 0 done, 3 phase 1 done, 4 dQ stored, 5 wait done, 6 barrier passed; slot 7 of iterations 0..5 =
// item start, first S / dP done, loop done, tail done, flush done, next item ready.  Every stamp drains lgkmcnt: read SHARES, not lengths.
__device__ unsigned long long *g_bwd4_stamps = nullptr;
#define B4_STAMP(J, I)                                                                                              \
    do {                                                                                                            \
        if (stamp_on && (J) < 16) {                                                                                 \
            unsigned long long t_;                                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
            u32x2 tv_;                                                                                              \
            tv_[0] = (uint32_t)t_; tv_[1] = (uint32_t)(t_ >> 32);                                                   \
            const uint32_t ta_ = lds_addr_of(stamp_lds) + (uint32_t)((wave * 128 + (J) * 8 + (I)) * 8);             \
            B4_W64(ta_, tv_, 0);                                                                                    \
        }                                                                                                           \
    } while (0)
#else
#define B4_STAMP(J, I) do { } while (0)
#endif

#define B4_R128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define B4_R128A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(off) : "memory")
#define B4_RTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define B4_W64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define B4_W32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define B4_SB() __builtin_amdgcn_sched_barrier(0)
#define B4_LGKM(n)                                                \
    do {                                                          \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
#define B4_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// keeps a register tuple allocated up to this point (an asm MFMA's operand registers must not be handed to a later LDS request while the MFMA may still read them)
#define B4_KEEP(x) asm volatile("" ::"v"(x))
// S / dP products as inline asm: their accumulators must be ARCHITECTURAL registers (the softmax arithmetic reads every element; hipcc gives a
// kernel with a 512-register budget AGPR accumulators and copies each element out with v_accvgpr_read: +64 vector instructions per block); the
// K / V fragments (B operand) live in AGPRs for the whole item.  Hazards are the author's (nothing inside asm is padded): the first vector
// read of an accumulator comes a whole phase (> 40 instructions) after the MFMA that wrote it.
// If hipcc has to evict a K / V fragment from the AGPRs it copies it back with v_accvgpr_write DIRECTLY in front of the statement: two wait
// states between such a write and an MFMA that reads it are the author's to provide (s_nop 1; measured wrong dP / dS without it).
#define B4_MFMA_ACC(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define B4_MFMA_NEW(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b))   /* early clobber: the destination must not overlap the A operand */
// the dQ tiles' 16x16x32 products as inline asm too: as builtins hipcc SINKS them (six quarters' worth collected in front of their first use,
// their twelve operand fragments held in 72 registers meanwhile); asm volatile statements stay where they are written
#ifndef B4_ASM_DQ
#define B4_ASM_DQ 1
#endif
#if B4_ASM_DQ
#define B4_MFMA16_ACC(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define B4_MFMA16_NEW(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#else
#define B4_MFMA16_ACC(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)
#define B4_MFMA16_NEW(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
#endif
// drop decision of element K of a group: byte K of the transposed hash word against the threshold, into an SGPR pair (an SDWA compare needs two
// wait states before a vector instruction reads its mask: all four compares of a group are issued in the hash piece, the selects come later)
#define B4_CMP(mask, X, K) asm volatile("v_cmp_ge_u32_sdwa %0, %1, %2 src0_sel:BYTE_" #K " src1_sel:DWORD" : "=s"(mask) : "v"(X), "s"(thresh))
#define B4_SEL(dst, p, mask) asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(dst) : "v"(p), "s"(mask))

__device__ __forceinline__ bf16x8 b4_join(bf16x4 a, bf16x4 b) { return join_halves(a, b); }
__device__ __forceinline__ bf16x8 b4_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    u32x4 u;
    u[0] = a; u[1] = b; u[2] = c; u[3] = d;
    return __builtin_bit_cast(bf16x8, u);
}

// Q8 (fp8_linear; bit 0: dK / dV, bit 1: dQ): additionally dqkv8 = saturate(dqkv as stored / *q8_scale) in e5m2 and *q8_amax = max |dqkv|
template <bool DROP, bool ACCUM, int Q8>
__global__ __launch_bounds__(256) void attn_bwd4_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out,
                                                         const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                         bf16_t *__restrict__ dqkv, int N, int h, float scale, uint64_t seed,
                                                         uint32_t thresh, float inv_keep, int nitems, int k0,
                                                         uint8_t *__restrict__ dqkv8, const float *__restrict__ q8_scale,
                                                         float *__restrict__ q8_amax) {
    constexpr int IMG = 32768, QDS = 8192, OS = 4096, DSB = 16384, NQ = 512;
    constexpr int NQD = 5, NOR = 3;
    constexpr int SQ = (Q8 & 2) ? 2 : 1;     // stores of one dQ tile
    constexpr int SF = (Q8 & 1) ? 2 : 1;     // stores per flushed dK / dV row piece
#ifdef ECGVIT_TOOLS
    constexpr int STAMP_BYTES = 4096;
#else
    constexpr int STAMP_BYTES = 0;
#endif
    __shared__ __attribute__((aligned(1024))) char smem[2 * IMG + NQD * QDS + NOR * OS + 2 * DSB + 2 * NQ * 4 + 2 * 32 * 4 + 64 + STAMP_BYTES];
    char *const Kimg0 = smem, *const qd0 = smem + 2 * IMG, *const o0 = qd0 + NQD * QDS, *const dSimg = o0 + NOR * OS;
    float *const lse_s = reinterpret_cast<float *>(dSimg + 2 * DSB), *const delta_s = lse_s + 2 * NQ;
#ifdef ECGVIT_TOOLS
    char *const stamp_lds = reinterpret_cast<char *>(delta_s + 2 * 32 + 16);
    unsigned long long *const stamp_out = g_bwd4_stamps;
    bool stamp_on = false;
    int item_no = 0;
#endif

    [[maybe_unused]] float q8_inv = 0.f, qmax = 0.f;
    if constexpr (Q8 != 0) { const float sc = *q8_scale; q8_inv = sc > 0.f ? 1.0f / sc : 0.f; }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int key0 = wave * 64 + lr;          // this lane's key of tile 0 inside the window (tile 1: + 32)
    const int d = h * 64, d3 = 3 * d;
    const int nqb = (N + 31) >> 5;
    const float c = scale * 1.44269504088896340736f;
    const float log2_ik = DROP ? __builtin_log2f(inv_keep) : 0.f;
    const float inv_ik = 1.0f / inv_keep;     // delta' = delta / inv_keep

    // ---- lane constants of the LDS addressing
    const RowOff ro = make_row_off(lane);
    const TrOff to = make_tr_off(lane);
    const int dq_g = lane >> 4, dq_i = lane & 15;
    const int dq_key = 4 * dq_g + (dq_i >> 2);
    const int qt = wave & 1, dhc0 = (wave >> 1) * 2;      // dQ tiles of this wave: queries 16 qt.., dh chunks dhc0, dhc0 + 1
    const uint32_t dq_a = (uint32_t)(dq_key * 64 + (((qt * 4 + (dq_i & 3)) ^ dsw(dq_key)) << 3));
    const uint32_t dq_b0 = (uint32_t)img_off(dq_key, (dhc0 * 16 + (dq_i & 3) * 4) * 2);
    const uint32_t dq_b1 = (uint32_t)img_off(dq_key, ((dhc0 + 1) * 16 + (dq_i & 3) * 4) * 2);
    // dS^T image row of my key (tile 0; tile 1 = + 32 rows = + 2048 B: dsw looks at key bits 1..3 only), one address per group of 4 queries
    uint32_t dsw_off[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) dsw_off[g4] = (uint32_t)(key0 * 64 + (((2 * g4 + lh) ^ dsw(key0)) << 3));
    // delta: my row (8 rows per wave: the rows this wave DMAs), 16-B chunk
    const int drow = wave * 8 + (lane >> 3);
    const uint32_t d_off = (uint32_t)img_off(drow, (lane & 7) * 16);
    // byte-permute selectors of the in-quad transpose of the hash words
    const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u;
    const uint32_t sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;

    // ---- per-lane DMA source offsets (bytes): one piece = 8 image rows; the image swizzle is applied to the SOURCE chunk
    const int prow = wave * 8 + (lane >> 3);
    const int pchunk = ((lane & 7) ^ swz3(prow)) * 16;
    const int vo_q = prow * d3 * 2 + pchunk, vo_d = prow * d * 2 + pchunk;
    const int vo_k = prow * d3 * 2 + pchunk;   // K piece i of this wave: rows (wave + 4 i) * 8 ..: + i * 32 rows (same swizzle: 32 rows keep bits 1..3)
    const uint32_t bytes_q = (uint32_t)(((int64_t)(N - 1) * d3 + 64) * 2), bytes_d = (uint32_t)(((int64_t)(N - 1) * d + 64) * 2);
    const uint32_t bytes_k = (uint32_t)(((int64_t)(N - k0 - 1) * d3 + 64) * 2);

    struct Item { const bf16_t *q, *o, *dO; int bh, b, hd; uint32_t live; };
    auto make_item = [&](int it) {
        Item x;
        x.live = it < nitems ? 1u : 0u;
        const int itc = x.live ? it : 0;
        x.bh = itc; x.b = itc / h; x.hd = itc - x.b * h;
        x.q = qkv + (int64_t)x.b * N * d3 + x.hd * 64;
        x.o = out + (int64_t)x.b * N * d + x.hd * 64;
        x.dO = dout + (int64_t)x.b * N * d + x.hd * 64;
        return x;
    };
    // slab qb of item x -> Q | dO ring slot sq, O ring slot so.  Always three pieces per wave (an item past the end gets EMPTY descriptors:
    // the hardware drops the loads, the instruction count -- what the counted waits depend on -- stays)
    auto dma_slab = [&](const Item &x, int qb, int sq, int so) {
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)x.q, 0, x.live ? bytes_q : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)x.dO, 0, x.live ? bytes_d : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)x.o, 0, x.live ? bytes_d : 0u, 0x00020000);
        char *dst = qd0 + sq * QDS + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_p4)dst, 16, vo_q, qb * 32 * d3 * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_p4)(dst + 4096), 16, vo_d, qb * 32 * d * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rO, (lds_void_p4)(o0 + so * OS + wave * 1024), 16, vo_d, qb * 32 * d * 2, 0, 0);
    };
    auto dma_k = [&](const Item &x, char *img) {   // 8 pieces per wave
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(x.q + d + (int64_t)k0 * d3), 0, x.live ? bytes_k : 0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_p4)(img + (wave + 4 * i) * 1024), 16, vo_k, i * 32 * d3 * 2, 0, 0);
    };
    // V fragments of the wave's 64 keys (B operand of dP): 8 inline-asm loads (hipcc would wait vmcnt(0) for a builtin load behind DMA pieces)
    auto load_v = [&](const Item &x, u32x4 (&v)[2][4]) {
        const uintptr_t pa = reinterpret_cast<uintptr_t>(x.q + 2 * d + (int64_t)k0 * d3);
        const i32x4_t rv = i32x4_t{(int)(uint32_t)pa, (int)((pa >> 32) & 0xFFFFu), (int)(x.live ? bytes_k : 0u), 0x00020000};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint32_t off = (uint32_t)(((key0 + 32 * t) * d3 + ks * 16 + 8 * lh) * 2);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=a"(v[t][ks]) : "v"(off), "s"(rv) : "memory");
            }
    };
    auto load_lse = [&](const Item &x, float (&l)[2]) {   // 2 inline-asm loads: queries tid, tid + 256
        const uintptr_t pa = reinterpret_cast<uintptr_t>(lse + (int64_t)x.bh * N);
        const i32x4_t rl = i32x4_t{(int)(uint32_t)pa, (int)((pa >> 32) & 0xFFFFu), (int)(x.live ? (uint32_t)N * 4u : 0u), 0x00020000};
        const uint32_t o0_ = threadIdx.x * 4u, o1_ = threadIdx.x * 4u + 1024u;
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(l[0]) : "v"(o0_), "s"(rl) : "memory");
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(l[1]) : "v"(o1_), "s"(rl) : "memory");
    };
    auto store_lse = [&](int par_, const float (&l)[2]) {   // -(LSE log2 e) + log2(1/(1-p)): the exponential then returns p / (1-p)
        lse_s[par_ * NQ + threadIdx.x] = log2_ik - l[0] * 1.44269504088896340736f;
        lse_s[par_ * NQ + 256 + threadIdx.x] = log2_ik - l[1] * 1.44269504088896340736f;
    };
    // delta' of the 8 slab rows this wave DMA'd (Q|dO slot sq, O slot so) -> delta_s[buf][32]
    auto slab_delta = [&](int sq, int so, int buf) {
        u32x4 a, o;
        const uint32_t aa = lds_addr_of(qd0 + sq * QDS + 4096) + d_off, ao = lds_addr_of(o0 + so * OS) + d_off;
        B4_R128(a, aa, 0);
        B4_R128(o, ao, 0);
        B4_LGKM(0);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc = fmaf(__builtin_bit_cast(float, a[k] << 16), __builtin_bit_cast(float, o[k] << 16), acc);
            acc = fmaf(__builtin_bit_cast(float, a[k] & 0xFFFF0000u), __builtin_bit_cast(float, o[k] & 0xFFFF0000u), acc);
        }
        // 8-lane row reduction with DPP row shifts: lane 7 of each group of 8 ends with the sum
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x114, 0xF, 0xF, true));   // row_shr:4
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x112, 0xF, 0xF, true));   // row_shr:2
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x111, 0xF, 0xF, true));   // row_shr:1
        if ((lane & 7) == 7) {
            const uint32_t da = lds_addr_of(reinterpret_cast<const char *>(delta_s + buf * 32 + drow));
            const float v = acc * inv_ik;
            B4_W32(da, v, 0);
        }
    };

    int it = blockIdx.x;
    if (it >= nitems) return;
    Item cur = make_item(it), nxt = make_item(it + (int)gridDim.x);
    u32x4 kf[2][4], vf[2][4];
    float lse_n[2];
    // ---- prologue of the workgroup's first item: K image, V fragments, LSE row, slabs 0..2; the Q|dO slot "before" slab 0 is zeroed
    //      (the first block's dV / dK products of "block -1" multiply it with zero probabilities: it must hold finite numbers)
    dma_k(cur, Kimg0);
    load_v(cur, vf);
    load_lse(cur, lse_n);
    dma_slab(cur, 0, 0, 0);
    dma_slab(cur, 1, 1, 1);     // (N > 128: at least five query blocks)
    dma_slab(cur, 2, 2, 2);
    for (int i = threadIdx.x; i < QDS / 16; i += 256) reinterpret_cast<u32x4 *>(qd0 + (NQD - 1) * QDS)[i] = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    store_lse(0, lse_n);
    slab_delta(0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ring slots, kept incrementally (no modulo arithmetic in the loop): Q|dO ring of 5 -- slab j-1, j, j+1, j+3 --, O ring of 3 -- slab j (= where
    // slab j+3 goes), j+1 --, delta / dS parity, K / LSE buffer
    int gqm = NQD - 1, gq = 0, gqp = 1, gq3 = 3, go = 0, gop = 1, gj = 0, par = 0;
    for (;;) {
        const int next_it = it + (int)gridDim.x;
        const bool has_next = next_it < nitems;
#ifdef ECGVIT_TOOLS
        stamp_on = stamp_out != nullptr && item_no == 1;
#endif
        B4_STAMP(0, 7);
        const char *Kimg = Kimg0 + par * IMG;
        const uint32_t lse_c = lds_addr_of(reinterpret_cast<const char *>(lse_s + par * NQ)) + (uint32_t)lh * 16u;
        // ---- item start: K fragments of my 64 keys from the image, accumulators
        {
            const uint32_t ka = lds_addr_of(Kimg) + (uint32_t)wave * 8192u;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) B4_R128A(kf[t][ks], ka + (uint32_t)ro.ks[ks], t * 4096);
        }
        f32x16 dKt[2][2], dVt[2][2], s[2], dp[2];
        uint32_t Pk[2][8], Dk[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { dKt[t][dt][r] = 0.f; dVt[t][dt][r] = 0.f; }
#pragma unroll
            for (int m = 0; m < 8; ++m) { Pk[t][m] = 0u; Dk[t][m] = 0u; }
        }
        // dropout hash: lane (quad position lq) hashes query 4 lh + lq of each group of four, key quad (key >> 2); per (tile, group) offsets
        const uint32_t qpitch = (uint32_t)((N + 3) >> 2);
        const uint32_t hstep = qpitch * ECGVIT_WEYL;
        uint32_t hq = seed_mix(seed) + (((uint32_t)cur.bh * (uint32_t)N + (uint32_t)(4 * lh + (lane & 3))) * qpitch + (uint32_t)((key0 + k0) >> 2)) * ECGVIT_WEYL;
        [[maybe_unused]] i32x4_t rdq_words;
        {
            const uintptr_t pa = reinterpret_cast<uintptr_t>(dqkv + (int64_t)cur.b * N * d3 + cur.hd * 64);
            rdq_words = i32x4_t{(int)(uint32_t)pa, (int)((pa >> 32) & 0xFFFFu), (int)bytes_q, 0x00020000};
        }
        const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv + (int64_t)cur.b * N * d3 + cur.hd * 64), 0, bytes_q, 0x00020000);
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rdq8 =
            __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv8 + (int64_t)cur.b * N * d3 + cur.hd * 64), 0, (Q8 & 2) ? bytes_q / 2 : 0u, 0x00020000);
        B4_LGKM(0);
        // ---- S / dP of (block 0, tile 0): the pipeline's first stage
        {
            const uint32_t qa = lds_addr_of(qd0 + gq * QDS);
            u32x4 fq[4], fd[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { B4_R128(fq[ks], qa + (uint32_t)ro.ks[ks], 0); B4_R128(fd[ks], qa + (uint32_t)ro.ks[ks], 4096); }
            B4_LGKM(0);
            B4_MFMA_NEW(s[0], fq[0], kf[0][0]);
            B4_MFMA_NEW(dp[0], fd[0], vf[0][0]);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                B4_MFMA_ACC(s[0], fq[ks], kf[0][ks]);
                B4_MFMA_ACC(dp[0], fd[ks], vf[0][ks]);
            }
            s[1] = s[0];      // (defined values for the compiler; overwritten by the first phase before any use)
            dp[1] = dp[0];
        }
        // the fragment reads ROTATE through the quarters (each MFMA's operands are requested three pieces ahead of it): the first quarter's
        // S / dP operands (block 0, tile 1, k-step 0) are requested here
        f32x4 nlE, dlE, nlO = {0.f, 0.f, 0.f, 0.f}, dlO = {0.f, 0.f, 0.f, 0.f};   // -LSE' / delta' of the even / odd quarters' query groups
        u32x4 raS0, raD0, raS1, raD1;      // S / dP operand rows (Q, dO) of the even / odd quarters
        {
            const uint32_t aA = lds_addr_of(qd0 + gq * QDS) + (uint32_t)ro.ks[0];
            B4_R128(raS0, aA, 0);
            B4_R128(raD0, aA, 4096);
            raS1 = u32x4{0u, 0u, 0u, 0u};       // (defined values for the compiler: requested by the first quarter before any use)
            raD1 = u32x4{0u, 0u, 0u, 0u};
        }
        B4_STAMP(1, 7);

        f32x4 accq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        u32x2 dq_req[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};
        // dQ tile of block qb (both dh chunks): scale, [+ first window's value], bf16, store; rows >= N and "block -1" fall outside the descriptor
        auto dq_store = [&](int qb) __attribute__((always_inline)) {
            const int q = qb * 32 + qt * 16 + dq_i;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x4 v;
                if constexpr (ACCUM) {
                    const bf16x4 o = __builtin_bit_cast(bf16x4, dq_req[i]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(accq[i][r] * scale + (float)o[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(accq[i][r] * scale);
                }
                const int eoff = q * d3 + (dhc0 + i) * 16 + 4 * dq_g;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rdq, eoff * 2, 0, 0);
                if constexpr ((Q8 & 2) != 0) {
                    float f[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = (float)v[r];
                        qmax = fmaxf(qmax, (q >= 0 && q < N) ? fabsf(f[r]) : 0.f);
                        f[r] = __builtin_amdgcn_fmed3f(f[r] * q8_inv, -57344.f, 57344.f);
                    }
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w, false);
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w, true);
                    __builtin_amdgcn_raw_buffer_store_b32(w, rdq8, eoff, 0, 0);
                }
            }
        };
        auto dq_request = [&](int qb) __attribute__((always_inline)) {   // ACCUM: the first window's dQ values of block qb (2 loads)
            if constexpr (ACCUM) {
                const int q = qb * 32 + qt * 16 + dq_i;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t off = (uint32_t)((q * d3 + (dhc0 + i) * 16 + 4 * dq_g) * 2);
                    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dq_req[i]) : "v"(off), "s"(rdq_words) : "memory");
                }
            }
        };

        [[maybe_unused]] uint32_t hmine = pair_finish(hq);   // hash word of (block 0, tile 0, group 0); every quarter hashes its successor's
        for (int j = 0; j < nqb; ++j) {
            B4_STAMP(j, 0);
            // ---- LDS requests of the iteration's top, oldest first: -LSE' and delta' of the block's 32 queries (4 + 4 reads, shared by both tiles)
            const uint32_t dl_c = lds_addr_of(reinterpret_cast<const char *>(delta_s + (gj & 1) * 32)) + (uint32_t)lh * 16u;
            const uint32_t lq_c = lse_c + (uint32_t)j * 128u;
            // (-LSE', delta' of a group of four queries are requested one quarter ahead into the register set of the quarter's parity; the block's
            //  first group here: delta' of this block became visible with the previous barrier)
            B4_R128(nlE, lq_c, 0);
            B4_R128(dlE, dl_c, 0);
            // ---- feed the stream (vector-memory operations of this iteration, in this order: the counted waits below depend on it)
            //   [ACCUM: 2 dQ requests of block j-1] | 3 slab pieces (slab j+3) | [j == 1: 8 K pieces + 2 LSE words of the next item]
            dq_request(j - 1);
            if (j + 3 < nqb) dma_slab(cur, j + 3, gq3, go);     // O ring of 3: slab j+3 reuses slab j's slot (its delta was computed in iteration j-1)
            else dma_slab(nxt, j + 3 - nqb, gq3, go);
            if (j == 1) {
                dma_k(nxt, Kimg0 + (par ^ 1) * IMG);
                load_lse(nxt, lse_n);
            }
            const uint32_t qdA0 = lds_addr_of(qd0 + gq * QDS), qdB0 = lds_addr_of(qd0 + gqm * QDS);
            const uint32_t qdA1 = lds_addr_of(qd0 + gqp * QDS), qdB1 = qdA0;
            const uint32_t dsb_w = lds_addr_of(dSimg + (gj & 1) * DSB);             // dS of block j goes here
            const uint32_t dsb_r = lds_addr_of(dSimg + ((gj + 1) & 1) * DSB);       // dS of block j-1 is read from here
            const uint32_t sa = dsb_r + dq_a, ka0 = lds_addr_of(Kimg) + dq_b0, ka1 = lds_addr_of(Kimg) + dq_b1;
            B4_STAMP(j, 1);
            const uint32_t hqb = hq;          // hash base of this block's first query group
            hq += 32u * hstep;

// hash offset (from the block's base hqb) of the quarter FOLLOWING (PH, Q): group Q+1 of the same tile, group 0 of tile 1 after (0, 3), and
// group 0 / tile 0 of the NEXT block after (1, 3)
#define B4_NEXT_HASH(PH, Q) ((Q) < 3 ? (uint32_t)(8 * ((Q) + 1)) * hstep + (uint32_t)(8 * (PH)) * ECGVIT_WEYL : ((PH) == 0 ? (uint32_t)8 * ECGVIT_WEYL : 32u * hstep))
// One quarter of a phase: the vector work of query group Q of tile TV (phase PH = TV) in six pieces -- hash | element 0..3 | pack --, ONE MFMA of
// tile TM = 1 - PH behind each piece: M1 S, M2 dP (k-step Q), M3 dV, M4 dK (sub-step SS, dh half DT of block j-1 / j), M5, M6 the two dQ tiles (key
// step ST of block j-1).  The operand reads rotate: each piece opens with the requests of the MFMA three pieces on (M1 / M2 of the NEXT quarter
// behind the last two), so LDS requests are spread through the vector stream instead of queueing up at a quarter's head, and every wait is
// counted: lgkmcnt(n) with n = the requests issued behind the operands needed.  T = LDS operations issued between the previous quarter and this one.
#define B4_QUARTER(PH, Q, T)                                                                                                       \
    {                                                                                                                              \
        constexpr int TV = PH, TM = 1 - PH, SS = (Q) >> 1, DT = (Q) & 1, ST = 4 * (PH) + (Q), NK = ((Q) + 1) & 3;                  \
        constexpr bool FIRST = (PH) == 0 && (Q) == 0;                                                                              \
        const uint32_t qdB = (PH) == 0 ? qdB0 : qdB1;                                                                              \
        const uint32_t aAn = (((PH) == 0 && (Q) < 3) ? qdA0 : qdA1) + (uint32_t)ro.ks[NK];   /* S / dP operand rows of the NEXT quarter */ \
        const uint32_t aBl = qdB + (uint32_t)to.lo[DT], aBh = qdB + (uint32_t)to.hi[DT];                                           \
        bf16x4 rb0, rb1, rb2, rb3, rc0, rc1, rc2, rc3, rc4, rc5;                                                                   \
        /* this quarter's S / dP operands sit in the register set of its parity, the next quarter's are requested into the other one: no   \
           compiler-visible copy of a register an LDS request may still be writing */                                                \
        u32x4 &ra0o = ((Q) & 1) ? raS1 : raS0, &ra1o = ((Q) & 1) ? raD1 : raD0;                                                     \
        u32x4 &ra0n = ((Q) & 1) ? raS0 : raS1, &ra1n = ((Q) & 1) ? raD0 : raD1;                                                     \
        f32x4 &nlq = ((Q) & 1) ? nlO : nlE, &dlq = ((Q) & 1) ? dlO : dlE, &nln = ((Q) & 1) ? nlE : nlO, &dln = ((Q) & 1) ? dlE : dlO; \
        constexpr int NL = ((PH) == 1 && (Q) == 3) ? 0 : 2;   /* requests of the next quarter's -LSE' / delta' (none across the barrier) */ \
        /* requests, by piece: H: dV, dK operands (4) | E0: dQ tile 0 (4) | E1: dQ tile 1 (2) | E2, E3: the next quarter's S, dP operands */ \
        B4_RTR(rb0, aBl, SS * 2048 + 4096);                                                                                        \
        B4_RTR(rb1, aBh, SS * 2048 + 4096);                                                                                        \
        B4_RTR(rb2, aBl, SS * 2048);                                                                                               \
        B4_RTR(rb3, aBh, SS * 2048);                                                                                               \
        B4_SB();                                                                                                                   \
        B4_LGKM(FIRST ? 4 : 7 + (T));      /* -LSE' (requested a quarter ago; FIRST: at the iteration's top) has landed: the first exponential reads it */ \
        [[maybe_unused]] uint64_t km0 = 0, km1 = 0, km2 = 0, km3 = 0;                                                              \
        if constexpr (DROP) {                                                                                                      \
            /* in-quad byte transpose of the word hashed one quarter ago; the NEXT quarter's hash fills the DPP wait states */        \
            const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hmine, 0xB1, 0xF, 0xF, true);                        \
            const uint32_t t1 = __builtin_amdgcn_perm(nb, hmine, sel1);                                                            \
            hmine = pair_finish(hqb + B4_NEXT_HASH(PH, Q));                                                                        \
            const uint32_t nb2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t1, 0x4E, 0xF, 0xF, true);                          \
            const uint32_t X = __builtin_amdgcn_perm(nb2, t1, sel2);                                                               \
            B4_CMP(km0, X, 0);                                                                                                     \
            B4_CMP(km1, X, 1);                                                                                                     \
            B4_CMP(km2, X, 2);                                                                                                     \
            B4_CMP(km3, X, 3);                                                                                                     \
        }                                                                                                                          \
        /* the group's first exponential goes with the hash piece (its result is used one MFMA later) */                              \
        float pe0 = B4_EXP(TV, Q, 0), pe1, pe2, pe3, pd0, pd1, pd2, pd3, u0, u1, u2, u3, ds0, ds1, ds2, ds3;                        \
        B4_SB();                                                                                                                   \
        B4_LGKM(FIRST ? 4 : 6 + (T));                                                                                              \
        if constexpr ((Q) == 0) B4_MFMA_NEW(s[TM], ra0o, kf[TM][Q]); else B4_MFMA_ACC(s[TM], ra0o, kf[TM][Q]);                     \
        B4_SB();                                                                                                                   \
        /* piece k: exponential of element k+1, u of element k, dS of element k-1, and LAST the (asm) select of element k */         \
        B4_RTR(rc0, sa, ST * 2048);                                                                                                \
        B4_RTR(rc1, sa, ST * 2048 + 1024);                                                                                         \
        B4_RTR(rc2, ka0, ST * 4096);                                                                                               \
        B4_RTR(rc3, ka0, ST * 4096 + 2048);                                                                                        \
        pe1 = B4_EXP(TV, Q, 1);                                                                                                    \
        u0 = pe0 * -dlq[0];                                                                                                      \
        B4_SB();                                                                                                                   \
        B4_PICK(pd0, pe0, km0);                                                                                                    \
        B4_SB();                                                                                                                   \
        B4_LGKM(FIRST ? 8 : 9 + (T));                                                                                              \
        if constexpr ((Q) == 0) B4_MFMA_NEW(dp[TM], ra1o, vf[TM][Q]); else B4_MFMA_ACC(dp[TM], ra1o, vf[TM][Q]);                   \
        B4_SB();                                                                                                                   \
        B4_RTR(rc4, ka1, ST * 4096);                                                                                               \
        B4_RTR(rc5, ka1, ST * 4096 + 2048);                                                                                        \
        if constexpr (NL != 0) {                                                                                                   \
            B4_R128(nln, lq_c, (((Q) + 1) & 3) * 32);                                                                              \
            B4_R128(dln, dl_c, (((Q) + 1) & 3) * 32);                                                                              \
        }                                                                                                                          \
        pe2 = B4_EXP(TV, Q, 2);                                                                                                    \
        u1 = pe1 * -dlq[1];                                                                                                      \
        ds0 = fmaf(pd0, dp[TV][4 * (Q) + 0], u0);                                                                                  \
        B4_SB();                                                                                                                   \
        B4_PICK(pd1, pe1, km1);                                                                                                    \
        B4_SB();                                                                                                                   \
        B4_LGKM(8 + NL);                                                                                                           \
        dVt[TM][DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4_join(rb0, rb1), b4_frag(Pk[TM][4 * SS], Pk[TM][4 * SS + 1], Pk[TM][4 * SS + 2], Pk[TM][4 * SS + 3]), dVt[TM][DT], 0, 0, 0); \
        B4_SB();                                                                                                                   \
        B4_R128(ra0n, aAn, 0);                                                                                                     \
        pe3 = B4_EXP(TV, Q, 3);                                                                                                    \
        u2 = pe2 * -dlq[2];                                                                                                      \
        ds1 = fmaf(pd1, dp[TV][4 * (Q) + 1], u1);                                                                                  \
        B4_SB();                                                                                                                   \
        B4_PICK(pd2, pe2, km2);                                                                                                    \
        B4_SB();                                                                                                                   \
        B4_LGKM(7 + NL);                                                                                                           \
        dKt[TM][DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4_join(rb2, rb3), b4_frag(Dk[TM][4 * SS], Dk[TM][4 * SS + 1], Dk[TM][4 * SS + 2], Dk[TM][4 * SS + 3]), dKt[TM][DT], 0, 0, 0); \
        B4_SB();                                                                                                                   \
        B4_R128(ra1n, aAn, 4096);                                                                                                  \
        u3 = pe3 * -dlq[3];                                                                                                      \
        ds2 = fmaf(pd2, dp[TV][4 * (Q) + 2], u2);                                                                                  \
        const uint32_t wa = dsb_w + dsw_off[Q];                                                                                    \
        B4_SB();                                                                                                                   \
        B4_PICK(pd3, pe3, km3);                                                                                                    \
        B4_SB();                                                                                                                   \
        B4_LGKM(4 + NL);                                                                                                           \
        const bf16x8 fc_s = b4_join(rc0, rc1), fc_k0 = b4_join(rc2, rc3);                                                          \
        if constexpr (ST == 0) B4_MFMA16_NEW(accq[0], fc_k0, fc_s); else B4_MFMA16_ACC(accq[0], fc_k0, fc_s);                      \
        B4_SB();                                                                                                                   \
        {                                                                                                                          \
            ds3 = fmaf(pd3, dp[TV][4 * (Q) + 3], u3);                                                                              \
            const uint32_t p0 = cvt_pk_bf16(pd0, pd1), p1 = cvt_pk_bf16(pd2, pd3);                                                 \
            const uint32_t d0 = cvt_pk_bf16(ds0, ds1), d1 = cvt_pk_bf16(ds2, ds3);                                                 \
            u32x2 dw;                                                                                                              \
            dw[0] = d0; dw[1] = d1;                                                                                                \
            B4_W64(wa, dw, TV * 2048);                                                                                             \
            /* tile TV's packed values: this phase's dV / dK products read tile TM's, the previous block's values of TV are spent */  \
            Pk[TV][2 * (Q)] = p0; Pk[TV][2 * (Q) + 1] = p1; Dk[TV][2 * (Q)] = d0; Dk[TV][2 * (Q) + 1] = d1;                        \
        }                                                                                                                          \
        B4_SB();                                                                                                                   \
        B4_LGKM(3 + NL);                                                                                                           \
        const bf16x8 fc_k1 = b4_join(rc4, rc5);                                                                                    \
        if constexpr (ST == 0) B4_MFMA16_NEW(accq[1], fc_k1, fc_s); else B4_MFMA16_ACC(accq[1], fc_k1, fc_s);                      \
        B4_SB();                                                                                                                   \
    }

// one score element (query row r = 4 Q + K of tile T): p' = p / (1-p_drop) straight from the exponential, P_dropped = keep ? p' : 0,
// dS' = P_dropped dP - p' delta'
#define B4_EXP(T, Q, K) __builtin_amdgcn_exp2f(fmaf(s[T][4 * (Q) + (K)], c, nlq[K]))
#define B4_PICK(dst, p, mask)                      \
    do {                                           \
        if constexpr (DROP) B4_SEL(dst, p, mask);  \
        else dst = p;                              \
    } while (0)

            // ---- phase 0: V(j, tile 0) | S / dP (j, tile 1), dV / dK (j-1, tile 1), dQ(j-1) key steps 0..3
            B4_QUARTER(0, 0, 0)
            B4_QUARTER(0, 1, 0)
            B4_QUARTER(0, 2, 0)
            B4_QUARTER(0, 3, 0)
            B4_STAMP(j, 2);
            // ---- phase 1: V(j, tile 1) | S / dP (j+1, tile 0), dV / dK (j, tile 0), dQ(j-1) key steps 4..7; delta' of the 8 rows of slab j+1 this
            //      wave DMA'd itself (landed since the previous iteration's wait; published by this iteration's barrier): requested ahead of the
            //      phase, evaluated between its quarters 1 and 2
            u32x4 da_, do_;
            {
                const uint32_t aa = lds_addr_of(qd0 + gqp * QDS + 4096) + d_off, ao = lds_addr_of(o0 + gop * OS) + d_off;
                B4_R128(da_, aa, 0);
                B4_R128(do_, ao, 0);
            }
            B4_QUARTER(1, 0, 2)
            B4_QUARTER(1, 1, 0)
            {
                B4_LGKM(4);   // (the two delta operand rows were requested ahead of the phase: long landed)
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc = fmaf(__builtin_bit_cast(float, da_[k] << 16), __builtin_bit_cast(float, do_[k] << 16), acc);
                    acc = fmaf(__builtin_bit_cast(float, da_[k] & 0xFFFF0000u), __builtin_bit_cast(float, do_[k] & 0xFFFF0000u), acc);
                }
                // 8-lane row reduction with DPP row shifts: lane 7 of each group of 8 ends with the sum; the other lanes write a scratch word
                acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x114, 0xF, 0xF, true));   // row_shr:4
                acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x112, 0xF, 0xF, true));   // row_shr:2
                acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x111, 0xF, 0xF, true));   // row_shr:1
                const uint32_t da = (lane & 7) == 7 ? lds_addr_of(reinterpret_cast<const char *>(delta_s + ((gj + 1) & 1) * 32 + drow))
                                                    : lds_addr_of(reinterpret_cast<const char *>(delta_s + 64 + wave));
                const float v = acc * inv_ik;
                B4_W32(da, v, 0);
            }
            B4_SB();
            B4_QUARTER(1, 2, 1)
            B4_QUARTER(1, 3, 0)
            B4_STAMP(j, 3);
            // ---- dQ of block j-1 leaves (ACCUM: its request is older than this iteration's 3 pieces [+ 10 at j == 1])
            if constexpr (ACCUM) { if (j == 1) B4_VM(13); else B4_VM(3); }
            asm volatile("s_nop 15" ::: "memory");   // the last (asm) dQ product's result: wait states before the vector instructions below read it
            dq_store(j - 1);
            B4_STAMP(j, 4);
            // ---- the block's one wait + barrier.  Must have landed: my pieces of slab j+2 (issued one iteration ago).  Allowed in flight
            // (younger): this iteration's requests / pieces / stores and the previous iteration's stores (+ prefetches of iterations 1, 2).
            __builtin_amdgcn_sched_barrier(0);
            {
                constexpr int A2 = ACCUM ? 2 : 0;
                if (j == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A2 + 3 + 10 + 4 * SQ) : "memory");
                else if (j == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A2 + 3 + 10 + 4 * SQ) : "memory");
                else if (j == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A2 + 3 + 2 * SQ) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A2 + 3 + 4 * SQ) : "memory");
            }
            B4_STAMP(j, 5);
            __builtin_amdgcn_s_barrier();
            B4_STAMP(j, 6);
            __builtin_amdgcn_sched_barrier(0);
            gqm = gq; gq = gqp; gqp = gqp == NQD - 1 ? 0 : gqp + 1; gq3 = gq3 == NQD - 1 ? 0 : gq3 + 1;
            go = gop; gop = gop == NOR - 1 ? 0 : gop + 1;
            gj ^= 1;
        }
        B4_LGKM(0);     // (the operand reads the last quarter requested for a following iteration)
        B4_STAMP(2, 7);
        // ---- tail of the item: dV / dK of (last block, tile 1) and the last block's dQ tile (their operands are complete behind the barrier)
        {
            const uint32_t qdB = lds_addr_of(qd0 + gqm * QDS);
            const uint32_t dsb_r = lds_addr_of(dSimg + ((gj + 1) & 1) * DSB);
            const uint32_t sa = dsb_r + dq_a, ka0 = lds_addr_of(Kimg) + dq_b0, ka1 = lds_addr_of(Kimg) + dq_b1;
            if (has_next) load_v(nxt, vf);      // the next item's V fragments travel during the tail and the flush
            dq_request(nqb - 1);
            accq[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            accq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#define B4_TAILB(SS, DT)                                                                                                           \
    {                                                                                                                              \
        bf16x4 rb0, rb1, rb2, rb3;                                                                                                 \
        const uint32_t aBl = qdB + (uint32_t)to.lo[DT], aBh = qdB + (uint32_t)to.hi[DT];                                           \
        B4_RTR(rb0, aBl, SS * 2048 + 4096);                                                                                        \
        B4_RTR(rb1, aBh, SS * 2048 + 4096);                                                                                        \
        B4_RTR(rb2, aBl, SS * 2048);                                                                                               \
        B4_RTR(rb3, aBh, SS * 2048);                                                                                               \
        B4_LGKM(0);                                                                                                                \
        dVt[1][DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4_join(rb0, rb1), b4_frag(Pk[1][4 * SS], Pk[1][4 * SS + 1], Pk[1][4 * SS + 2], Pk[1][4 * SS + 3]), dVt[1][DT], 0, 0, 0); \
        dKt[1][DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b4_join(rb2, rb3), b4_frag(Dk[1][4 * SS], Dk[1][4 * SS + 1], Dk[1][4 * SS + 2], Dk[1][4 * SS + 3]), dKt[1][DT], 0, 0, 0); \
    }
#define B4_TAILC(ST)                                                                                                               \
    {                                                                                                                              \
        bf16x4 rc0, rc1, rc2, rc3, rc4, rc5;                                                                                       \
        B4_RTR(rc0, sa, ST * 2048);                                                                                                \
        B4_RTR(rc1, sa, ST * 2048 + 1024);                                                                                         \
        B4_RTR(rc2, ka0, ST * 4096);                                                                                               \
        B4_RTR(rc3, ka0, ST * 4096 + 2048);                                                                                        \
        B4_RTR(rc4, ka1, ST * 4096);                                                                                               \
        B4_RTR(rc5, ka1, ST * 4096 + 2048);                                                                                        \
        B4_LGKM(0);                                                                                                                \
        accq[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b4_join(rc2, rc3), b4_join(rc0, rc1), accq[0], 0, 0, 0);                 \
        accq[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b4_join(rc4, rc5), b4_join(rc0, rc1), accq[1], 0, 0, 0);                 \
    }
            B4_TAILB(0, 0) B4_TAILC(0) B4_TAILC(1)
            B4_TAILB(0, 1) B4_TAILC(2) B4_TAILC(3)
            B4_TAILB(1, 0) B4_TAILC(4) B4_TAILC(5)
            B4_TAILB(1, 1) B4_TAILC(6) B4_TAILC(7)
#undef B4_TAILB
#undef B4_TAILC
            if constexpr (ACCUM) B4_VM(0);
            dq_store(nqb - 1);
        }
        B4_STAMP(3, 7);
        // ---- item done: dK^T / dV^T (dh on rows, key on the lane) -> this wave's [key][dh] rows in a private 4-KiB patch -> 128-B row stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // every wave is done reading both dS buffers (the patches live there)
        {
            const uint32_t patch = lds_addr_of(dSimg + wave * 4096);
            // one descriptor for the dK and the dV section of a row (dV = + d elements): it ends behind the LAST key's dV piece, rows >= N start beyond it
            const __amdgpu_buffer_rsrc_t rk_ = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv + ((int64_t)cur.b * N + k0) * d3 + d + cur.hd * 64), 0, bytes_k + (uint32_t)d * 2u, 0x00020000);
            [[maybe_unused]] const __amdgpu_buffer_rsrc_t rk8_ =
                __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv8 + ((int64_t)cur.b * N + k0) * d3 + d + cur.hd * 64), 0, (Q8 & 1) ? bytes_k / 2 + (uint32_t)d : 0u, 0x00020000);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int which = 0; which < 2; ++which) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            u32x2 a;
                            if (which == 0) {
                                a[0] = cvt_pk_bf16(dKt[t][dt][4 * g4] * scale, dKt[t][dt][4 * g4 + 1] * scale);
                                a[1] = cvt_pk_bf16(dKt[t][dt][4 * g4 + 2] * scale, dKt[t][dt][4 * g4 + 3] * scale);
                            } else {
                                a[0] = cvt_pk_bf16(dVt[t][dt][4 * g4], dVt[t][dt][4 * g4 + 1]);
                                a[1] = cvt_pk_bf16(dVt[t][dt][4 * g4 + 2], dVt[t][dt][4 * g4 + 3]);
                            }
                            const uint32_t wa = patch + (uint32_t)img_off(lr, (dt * 32 + 8 * g4 + 4 * lh) * 2);
                            B4_W64(wa, a, 0);
                        }
                    // same-wave LDS operations execute in order: read the rows back (8 rows x 128 B per instruction) and store them
                    u32x4 val[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t ra = patch + (uint32_t)img_off(i * 8 + (lane >> 3), (lane & 7) * 16);
                        B4_R128(val[i], ra, 0);
                    }
                    B4_LGKM(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = wave * 64 + t * 32 + i * 8 + (lane >> 3), ch = lane & 7;
                        __builtin_amdgcn_raw_buffer_store_b128(val[i], rk_, (key * d3 + which * d + ch * 8) * 2, 0, 0);   // keys >= N: dropped
                        if constexpr ((Q8 & 1) != 0) {
                            float f[8];
#pragma unroll
                            for (int k = 0; k < 4; ++k) { f[2 * k] = __builtin_bit_cast(float, val[i][k] << 16); f[2 * k + 1] = __builtin_bit_cast(float, val[i][k] & 0xFFFF0000u); }
                            const bool kin = key + k0 < N;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                qmax = fmaxf(qmax, kin ? fabsf(f[k]) : 0.f);
                                f[k] = __builtin_amdgcn_fmed3f(f[k] * q8_inv, -57344.f, 57344.f);
                            }
                            int w0 = 0, w1 = 0;
                            w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w0, true);
                            w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4], f[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6], f[7], w1, true);
                            u32x2 q2;
                            q2[0] = (uint32_t)w0; q2[1] = (uint32_t)w1;
                            __builtin_amdgcn_raw_buffer_store_b64(q2, rk8_, key * d3 + which * d + ch * 8, 0, 0);
                        }
                    }
                }
        }
        B4_STAMP(4, 7);
#ifdef ECGVIT_TOOLS
        if (stamp_on) {   // this item's record leaves (plain stores: the boundary wait below only gets stricter)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int k = lane; k < 128; k += 64)
                stamp_out[((int64_t)blockIdx.x * 4 + wave) * 128 + k] = reinterpret_cast<const unsigned long long *>(stamp_lds)[wave * 128 + k];
        }
        ++item_no;
#endif
        if (!has_next) break;
        // ---- switch to the next item: its K image / LSE row were fetched during block 1, its V fragments during the tail.  Everything
        // older than the flush stores has landed after this wait (V fragments, LSE words, K pieces, slabs 0..2 of the next item)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * SF) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        store_lse(par ^ 1, lse_n);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // patches drained, K image and LSE row of the next item visible to every wave
        par ^= 1;
        it = next_it;
        cur = nxt;
        nxt = make_item(it + (int)gridDim.x);
    }
    if constexpr (Q8 != 0) {
        wave_amax_publish(q8_amax, qmax);
    }
#undef B4_QUARTER
#undef B4_EXP
#undef B4_PICK
#undef B4_NEXT_HASH
}

}  // namespace

#ifdef ECGVIT_TOOLS
extern "C" int ecgvit_tools_bwd4_stamps(void *buf) {   // device buffer of 768 x 4 x 128 uint64 (NULL = off): tools/attn_bwd4_stamps.py
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd4_stamps), &buf, sizeof(buf)) == hipSuccess ? ECGVIT_OK : ECGVIT_ELAUNCH;
}
#endif

// launcher used by attention.hip (N > 128; both key windows).  Arguments validated by the caller.
int ecgvit_attention_bwd4_launch(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h, float scale,
                                 uint32_t th, float ik, uint64_t seed, hipStream_t stream, void *dqkv8, const float *q8_scale, float *q8_amax) {
    const int nitems = B * h;
    // three workgroups' worth of items per CU slot: the hardware dispatcher hands them out as CUs free up (a launch that shares the GPU with a
    // collective's kernels is not left with late workgroups a full static share behind)
    const dim3 pg((unsigned)(nitems < 768 ? nitems : 768));
#define B4_ARGS(K0) pg, dim3(256), 0, stream, (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, N, h, scale, seed, th, ik, nitems, K0, (uint8_t *)dqkv8, q8_scale, q8_amax
#define B4_GO(DR, AC, K0, Q) hipLaunchKernelGGL((attn_bwd4_kernel<DR, AC, Q>), B4_ARGS(K0))
    if (dqkv8) {
        // one window: dK / dV / dQ all final in this launch; two windows: the first emits its dK / dV, the second its dK / dV and the final dQ
        if (N <= 256) { if (th) B4_GO(true, false, 0, 3); else B4_GO(false, false, 0, 3); }
        else { if (th) B4_GO(true, false, 0, 1); else B4_GO(false, false, 0, 1); }
        if (hipGetLastError() != hipSuccess) return ECGVIT_ELAUNCH;
        if (N > 256) { if (th) B4_GO(true, true, 256, 3); else B4_GO(false, true, 256, 3); }
    } else {
        if (th) B4_GO(true, false, 0, 0); else B4_GO(false, false, 0, 0);
        if (hipGetLastError() != hipSuccess) return ECGVIT_ELAUNCH;
        if (N > 256) { if (th) B4_GO(true, true, 256, 0); else B4_GO(false, true, 256, 0); }
    }
#undef B4_GO
#undef B4_ARGS
    return hipGetLastError() == hipSuccess ? ECGVIT_OK : ECGVIT_ELAUNCH;
}
