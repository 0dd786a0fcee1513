// Fused multi-head self-attention core for the bf16 path (dh = 64, N <= 256 tokens): scores never touch HBM.
//
// One workgroup owns one (record, head): its whole K/V (<= 256 x 64 bf16 = 32 KiB each) sits in LDS, so softmax
// is single-pass (no online rescale) and the backward needs no atomics -- dQ, dK, dV of a head are all produced
// inside its workgroup.
//
// forward  (4 waves, one 32-query block per wave per pass):
//   S^T = K . Q^T  (key on the accumulator ROW, query on the LANE)  -> row max / sum are in-lane + one xor-32 shuffle
//   O^T = V^T . P^T : the S^T accumulator, packed to bf16, is directly the B operand (k order = accumulator row order);
//                     V^T fragments come from the row-major V image through ds_read_b64_tr_b16.
// backward (one wave per 32-key tile; loop over 32-query blocks):
//   S = Q K^T and dP = dO V^T with the KEY on the lane, so P and dS (bf16) are directly the B operands of
//   dV^T += dO^T . P and dK^T += Q^T . dS (A operands = transposed reads of the dO / Q images);
//   dS crosses LDS once ([key][query] image) for dQ = dS . K, computed as 16x16x32 tiles, one per wave.
//
// LDS image for every [row][64 x bf16] tile (128-B rows):  16-B chunk index ^= bitrev3((row>>1)&7)
//   -> ds_read_b128 row reads (MFMA K-contiguous operand) hit 16 distinct slots per 16-lane group, and
//   -> ds_read_b64_tr_b16 reads of 4 consecutive rows x 64 B land in the 4 different 64-B quarters of the bank row.
#include "attn_common.h"
#include <cstdlib>

namespace {

// =====================================================================================================
// forward: online softmax over 32-key tiles (running max / sum per query, O rescaled when the max moves)
// =====================================================================================================
// Q8 (fp8_linear): additionally out8 = saturate(out as stored / *q8_scale) in e4m3 -- the A operand of the out-projection's 8-bit
// product, written here instead of by a quantise pass over `out` -- and *q8_amax = max(*q8_amax, max |out|) for the next step's scale
// SPLIT (records of more than 256 tokens, e.g. patch 10 -> 501): one workgroup per (record, head, 256-query half); the keys pass through
// the SAME 64 KiB of images in 256-key windows (the online softmax carries m, l and O across them), so two workgroups still share a CU --
// with all 501 keys resident (128 KiB) a CU held one workgroup, two waves per SIMD, and this VALU-bound kernel ran at 2/3 of its rate.
#ifdef ECGVIT_TOOLS
__device__ unsigned long long *g_attn_stamps = nullptr;   // diagnostics (ecgvit_debug_attn_stamps, tools build): per-block cycle stamps, else null
#endif
// `make tools TOOLS_EXTRA=-DECGVIT_ATTN_ABL=n` (tools/attn_ablate.sh): timing diagnostics with WRONG results -- one phase of the persistent backward removed
// per build (1: dQ product, 2: dV / dK products, 3: the vector arithmetic, 4: S / dP products, 5: dS -> LDS, 6: the slab / K stream, 7: the dQ stores), and of the
// forward (8: the softmax / dropout arithmetic, 9: the Q.K^T products, 10: the P.V products, 11: the K / V image loads, 12: the output stores)
#if defined(ECGVIT_TOOLS) && defined(ECGVIT_ATTN_ABL)
#define ATTN_ABL(n) (((ECGVIT_ATTN_ABL) >> (n)) & 1)   // a bit mask: bit n removes phase n
#else
#define ATTN_ABL(n) false
#endif
// ---- the vector phase of one 32-key x 32-query score tile, shared by both forward kernels (so that they agree bit for bit): lazy running maximum,
// p = exp2(s c - m c) and the row sum two elements per instruction (v_pk_fma_f32 / v_pk_add_f32: the same IEEE operations per element; the sum runs
// as two interleaved partial sums), packed to bf16, dropout applied to the PACKED pairs -- per key quad (one hash word, 8 bits per key) the keep bits
// in three bit-parallel instructions (the >= / < 128 threshold forms are ONE three-input bit operation on a uniform mask), per pair one byte permute,
// one packed arithmetic shift, one AND.  ~9 vector instructions per score element instead of ~11 (round 6).
template <bool DROP>
__device__ __forceinline__ void attn_fwd_tile_vec(f32x16 &sc, float &m, float &l, f32x16 (&o)[2], float c, uint32_t hb, uint32_t c4, uint32_t thi_mask,
                                                  u32x4 (&pk)[2]) {
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));   // (NOT v_permlane32_swap(mx, mx): with one value as both operands hipcc treats the two results as equal)
    // LAZY running maximum: the reference m of a query moves only when a tile's maximum exceeds it by more than 2^8 in the exponent, so a tile's
    // probabilities are at most 256 relative to it (bf16 / f32 keep the same relative precision there; the sums stay far inside f32) and the common
    // tile has no exponential of alpha and no rescale of the 32 output accumulators.  The exact maximum is not needed: out = sum(p v) / sum(p) and
    // LSE = m scale + log(sum p) hold for any reference m.  Wave-uniform branch; always taken on the first tile (m = -inf).
    if (__any((mx - m) * c > 8.0f)) {
        const float mn = fmaxf(m, mx);                 // finite from tile 0 on (key 0 is always valid)
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);   // raw v_exp_f32; m = -inf on the first tile -> 0; lanes that do not move: 1
        m = mn;
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
#ifndef ATTN_VEC_FORM
#define ATTN_VEC_FORM 0   // 0 ships.  Measured at 256 x 16 x 501 (streamed kernel) / 512 x 12 x 251 (one-item kernel) against round 5 on one device (profiles/r06_attn_fwd_stream.txt): form 0 (scalar fma / exp / add, dropout as four selects per hash word) 476.7 / 198.3 us; 1 (dropout on the packed pairs: keep bits in three bit-parallel instructions, permute + packed shift + AND per pair) 485.6 / 201.3; 2 (1 + v_pk_fma_f32 / v_pk_add_f32 for the exponent argument and the row sum) 498.8 / 200.5 -- fewer instructions, slower: packed f32 forms issue at half rate beside the MFMAs (profiles/r04_valu_rate.txt) and the select form overlaps better
#endif
#if ATTN_VEC_FORM >= 2
    const f32x2 c2 = {c, c}, nmc = {-m * c, -m * c};
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{sc[r], sc[r + 1]}, c2, nmc);   // argument <= 8
        const f32x2 p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        sc[r] = p[0]; sc[r + 1] = p[1];
        ls2 += p;
    }
    l += ls2[0] + ls2[1];                           // per-half partial sums; halves are combined after the loop
#else
    const float mc = m * c;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sc[r], c, -mc));   // one fma per score (the kernel is VALU-bound); argument <= 8
        sc[r] = p;
        ls += p;
    }
    l += ls;
#endif
#if ATTN_VEC_FORM == 0
    if constexpr (DROP) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t hh = pair_finish(hb + (uint32_t)(2 * g) * ECGVIT_WEYL);
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[4 * g + k] = ((hh >> (8 * k)) & 0xFFu) >= (256u - (c4 & 0xFFu) - 128u + (thi_mask & 128u)) ? sc[4 * g + k] : 0.f;
        }
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) pk[ss] = __builtin_bit_cast(u32x4, pack8(sc, ss));
#else
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) pk[ss] = __builtin_bit_cast(u32x4, pack8(sc, ss));
    if constexpr (DROP) {  // dropout on the probabilities (the normaliser keeps the un-dropped sum: softmax -> Dropout); the 1/(1-p) rescale is folded
        // into the final normalisation.  My 16 keys are 4 quads: kt*32 + 8g + 4*lh + {0..3}; dword j of pk[ss] = keys (2 (j & 1), + 1) of quad 2 ss + (j >> 1)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t hh = pair_finish(hb + (uint32_t)(2 * g) * ECGVIT_WEYL);
            const uint32_t x1 = (hh & 0x7F7F7F7Fu) + c4;
            const uint32_t x = (x1 & hh & thi_mask) | ((x1 | hh) & ~thi_mask);      // bit 7 of byte k: keep(key k) -- quad_keepbits for either threshold range
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const uint32_t wv = __builtin_amdgcn_perm(x, x, pp ? 0x030C020Cu : 0x010C000Cu);
                pk[g >> 1][2 * (g & 1) + pp] &= __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, wv) >> (s16x2_t){15, 15});
            }
        }
    }
#endif
}

template <bool DROP, bool Q8 = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 4) void attn_fwd_bf16_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out,
                                                               float *__restrict__ lse, int N, int h, float scale,
                                                               uint64_t seed, uint32_t thresh, float inv_keep,
                                                               uint8_t *__restrict__ out8 = nullptr, const float *__restrict__ q8_scale = nullptr,
                                                               float *__restrict__ q8_amax = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    [[maybe_unused]] unsigned int amax_seen = 0u;
    if constexpr (Q8) amax_seen = amax_peek(q8_amax);   // (consumed behind the last store: common.h, wave_amax_publish)
    const int nkt = (N + 31) >> 5, NK = SPLIT ? 256 : nkt * 32;
    char *Kimg = smem, *Vimg = smem + NK * 128;
    const int nqh = SPLIT ? (nkt + 7) >> 3 : 1;                         // 256-query halves per (record, head)
    const int bh = SPLIT ? blockIdx.x / nqh : blockIdx.x, qh = SPLIT ? blockIdx.x - bh * nqh : 0;
    const int b = bh / h, hd = bh - b * h;
    const int d = h * 64;
    const int64_t d3 = 3 * (int64_t)d;
    const bf16_t *base = qkv + (int64_t)b * N * d3 + hd * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // One workgroup per work item, two per CU.  Round 5 measured where such a workgroup's 16.7 us go (tools/attn_fwd_timeline.py: 4.5 us waiting for its
    // images, 7.0 in products and softmax, 0.8 storing, 4.3 of empty slot until the next one runs) and built both persistent forms: 2 x CUs workgroups
    // walking the items behind one barrier each (241 against 203 us at 512 x 12 x 251: the two workgroups of a CU fall into step, load together, compute
    // together) and ONE 16-wave workgroup per CU whose two 8-wave groups alternate by construction, one computing while the other loads (218 against 194:
    // two waves per SIMD cannot hide the LDS / MFMA / exp latencies that four do).  The dispatcher's refill keeps the phases mixed: kept.
    // profiles/r05_attn_ablation.txt
#ifdef ECGVIT_TOOLS
    // tools/attn_fwd_timeline.py: per workgroup {start, images landed, last product done, stores issued} on the 100-MHz clock, and the hardware id
    unsigned long long *fst = (g_attn_stamps && threadIdx.x == 0) ? g_attn_stamps + (int64_t)blockIdx.x * 8 : nullptr;
    if (fst) { fst[0] = __builtin_amdgcn_s_memrealtime(); fst[4] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
#else
    unsigned long long *const fst = nullptr;
#endif
    if constexpr (!SPLIT) {
        // K and V images by LDS-DMA: all 64 one-KiB pieces of the item in flight at once, no VGPR round trip and no ds_write pass
        if constexpr (!ATTN_ABL(11)) {
        dma_image<8>(Kimg, base + d, d3, N, NK, wave, lane);
        dma_image<8>(Vimg, base + 2 * d, d3, N, NK, wave, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (fst) fst[1] = __builtin_amdgcn_s_memrealtime();
    }

    const int lr = lane & 31, lh = lane >> 5;
    const float c = scale * 1.44269504088896340736f;
    const RowOff ro = make_row_off(lane);
    const TrOff to = make_tr_off(lane);
    // SPLIT: exactly one query block per wave (waves past the last block keep the barriers company)
    for (int qb = SPLIT ? qh * 8 + wave : wave; qb < (SPLIT ? qh * 8 + wave + 1 : nkt); qb += 8) {   // 8 waves: one 32-query block each per pass (all of N <= 256 in one pass)
        const bool live = !SPLIT || qb < nkt;
        const int q = live ? qb * 32 + lr : N;
        const int qc = q < N ? q : N - 1;
        bf16x8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(base + (int64_t)qc * d3 + ks * 16 + 8 * lh);
        const uint32_t rowquad = ((uint32_t)bh * (uint32_t)N + (uint32_t)qc) * (uint32_t)((N + 3) >> 2);   // first key quad of this query's row (4 keys share one hash)
        const uint32_t smix = seed_mix(seed);

        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m = -INFINITY, l = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            if constexpr (SPLIT) {
                if ((kt & 7) == 0) {   // next 256-key window: everyone is done with the previous one, then its K / V rows replace it
                    __syncthreads();
                    const int k0 = kt * 32, nv = min(256, N - k0);
                    dma_image<8>(Kimg, base + d + (int64_t)k0 * d3, d3, nv, ((nv + 31) >> 5) << 5, wave, lane);
                    dma_image<8>(Vimg, base + 2 * d + (int64_t)k0 * d3, d3, nv, ((nv + 31) >> 5) << 5, wave, lane);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                if (!live) continue;
            }
            const int ktl = SPLIT ? kt & 7 : kt;   // tile inside the resident images
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            if constexpr (ATTN_ABL(9)) { asm volatile("" : "+v"(s)); } else
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(Kimg + ktl * 4096, ro.ks[ks]), qf[ks], s, 0, 0, 0);
            if (kt == nkt - 1) {  // only the last tile can hold padded keys
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= N) s[r] = -INFINITY;
                }
            }
            u32x4 pk[2];
            if constexpr (!ATTN_ABL(8)) {
                attn_fwd_tile_vec<DROP>(s, m, l, o, c, (rowquad + (uint32_t)(kt * 8 + lh)) * ECGVIT_WEYL + smix, quad_c4(thresh), thresh >= 128u ? ~0u : 0u, pk);
            } else {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) pk[ss] = __builtin_bit_cast(u32x4, pack8(s, ss));
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pk[ss]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    if constexpr (ATTN_ABL(10)) { asm volatile("" : "+v"(o[dt]) : "v"(pf)); continue; }
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_c(Vimg + ktl * 4096 + ss * 2048, to.lo[dt], to.hi[dt]), pf, o[dt], 0, 0, 0);
                }
            }
        }
        l += __shfl_xor(l, 32, 64);
        if (fst) fst[2] = __builtin_amdgcn_s_memrealtime();
        [[maybe_unused]] float qmax = 0.f;
        // output: lane (lr, lh) holds columns dt*32 + 8*g4 + 4*lh + {0..3} of query row lr -- 8-B runs interleaved with its partner lane's (lr, lh^1).
        // One v_permlane32_swap per dword hands each lane of the pair BOTH halves of two g4 groups: 16-B stores of contiguous bytes (4 per wave and
        // dt instead of 16 of 8 B; the 8-bit copy: ONE 16-B store per dt instead of four of 4 B)
        {
            const float inv = inv_keep / l;   // inv_keep = 1 without dropout
            [[maybe_unused]] float q8_inv = 0.f;
            if constexpr (Q8) { const float sc = *q8_scale; q8_inv = sc > 0.f ? 1.0f / sc : 0.f; }
            const int qr = q < N ? q : 0;
            bf16_t *orow = out + ((int64_t)b * N + qr) * d + hd * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                // column groups (j, j + 2) at a time: (a, b) -> a' = {lower half: a of lh 0, upper half: b of lh 0}, b' = {lower: a of lh 1, upper: b of lh 1},
                // i.e. lane lh ends up with group 2*lh + j complete: [lh 0: dwords 0, 1 | lh 1: dwords 0, 1] = one 16-B store
                [[maybe_unused]] u32x4 w8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint32_t D[2][2];
                    [[maybe_unused]] uint32_t W8[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g4 = j + 2 * gg;
                        bf16x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[dt][4 * g4 + k] * inv);
                        const u32x2 vv = __builtin_bit_cast(u32x2, v);
                        D[gg][0] = vv[0]; D[gg][1] = vv[1];
                        if constexpr (Q8) {
                            float f[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                f[k] = (float)v[k];                       // the value as stored
                                qmax = fmaxf(qmax, q < N ? fabsf(f[k]) : 0.f);
                                f[k] = __builtin_amdgcn_fmed3f(f[k] * q8_inv, -448.f, 448.f);
                            }
                            int w = 0;
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
                            W8[gg] = (uint32_t)w;
                        }
                    }
                    u32x4 st;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(D[0][k], D[1][k], false, false);
                        st[k] = sw[0]; st[2 + k] = sw[1];
                    }
                    if (q < N && !(ATTN_ABL(12) && st[0] != 0x12345u)) *reinterpret_cast<u32x4 *>(orow + dt * 32 + 16 * lh + 8 * j) = st;
                    if constexpr (Q8) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(W8[0], W8[1], false, false);
                        w8[2 * j] = sw[0]; w8[2 * j + 1] = sw[1];
                    }
                }
                if constexpr (Q8) {
                    if (q < N) *reinterpret_cast<u32x4 *>(out8 + ((int64_t)b * N + qr) * d + hd * 64 + dt * 32 + 16 * lh) = w8;
                }
            }
            if (q < N && lh == 0) lse[(int64_t)bh * N + q] = m * scale + logf(l);
        }
        if (fst) fst[3] = __builtin_amdgcn_s_memrealtime();
        if constexpr (Q8) {
            wave_amax_publish(q8_amax, qmax, amax_seen);
        }
    }
}

// =====================================================================================================
// forward, STREAMED (round 6): ONE persistent 16-wave workgroup per CU; the K / V rows arrive as a continuous stream of 64-KiB windows through a
// two-slot ring of LDS-DMA images that runs across (record, head) items -- every wave computes window t while the pieces of window t + 1 are in
// flight, four waves per SIMD compute all the time, nothing is relaunched and nothing drains.
//   GROUPS = 2 (N <= 256): waves 0-7 / 8-15 own two items side by side, a window = 128 keys of each (K 16 KiB + V 16 KiB per item);
//   GROUPS = 1 (256 < N <= 512): the 16 waves are the 16 query blocks of ONE item, a window = 256 of its keys (K 32 KiB + V 32 KiB) -- the K / V
//                rows of an item are read ONCE (the split one-item form read them once per 256-query half: 1.6 x the algorithmic bytes at 501 tokens).
// The online softmax of the one-item kernel already carries (m, l, O) across key tiles, so a window boundary changes no arithmetic: outputs, LSE and
// dropout masks are BIT-IDENTICAL to attn_fwd_bf16_kernel (tests/test_gpu_ops.py holds the two against each other).
// One barrier per window: [my pieces of window t landed: counted vmcnt] -> barrier (window t complete, everyone done with window t - 1: its slot
// is free) -> issue the pieces of window t + 1 -> compute window t.  vmcnt retires in issue order, and every wave issues a FIXED number of vector-memory
// operations behind the pieces of an item's first window (the next Q fragments -- requested behind the item's last Q.K^T product, into the registers it
// frees -- and NST output stores, all bounds-checked buffer operations: rows >= N are dropped by the descriptor, never branched around), so the wait for
// those pieces is vmcnt(NST): the previous item's stores stay in flight under the next item's first window (a one-item kernel drains them with its
// sixteen waves idle: profiles/r05_attn_ablation.txt).  V^T fragments are inline-asm transposed reads (hipcc drains vmcnt(0) in front of the builtin
// form when an LDS-DMA is in flight: attn_common.h), requested at the top of a tile and consumed behind its softmax.
// =====================================================================================================
// MODE 1: 16 waves, one item, 256-key windows (N > 256).  MODE 2: 16 waves, two items side by side, 128-key windows (N <= 256).  MODE 3: 8 waves, one item,
// 128-key windows, TWO workgroups per CU with a 2 x 32 KiB ring each (N <= 256): a workgroup's barrier joins its own eight waves only
template <bool DROP, bool Q8, int MODE>
__global__ __launch_bounds__(MODE == 3 ? 512 : 1024, 4) void attn_fwd_stream_kernel(const bf16_t *__restrict__ qkv, bf16_t *__restrict__ out, float *__restrict__ lse, int N, int h,
                                                               float scale, uint64_t seed, uint32_t thresh, float inv_keep, int nitems,
                                                               uint8_t *__restrict__ out8 = nullptr, const float *__restrict__ q8_scale = nullptr,
                                                               float *__restrict__ q8_amax = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 slots x 64 KiB
    constexpr int GROUPS = MODE == 2 ? 2 : 1, NWV = MODE == 3 ? 8 : 16;
    constexpr int WG_ = NWV / GROUPS;     // waves = 32-query blocks per item
    constexpr int WK = MODE == 1 ? 256 : 128;   // keys per window
    constexpr int SLOT = GROUPS * WK * 256;     // bytes of a ring slot
    constexpr int TPW = WK / 32;          // key tiles per window
    constexpr int GB = WK * 256;          // bytes of one group's K + V images inside a slot
    constexpr int NST = Q8 ? 7 : 5;       // vector-memory operations a wave issues behind its Q request: 4 output stores of 16 B (+ 2 of the 8-bit copy) + the LSE store
    [[maybe_unused]] float qmax = 0.f;      // running |out| maximum of this wave over all its items (8-bit emitting form)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = GROUPS == 2 ? wave >> 3 : 0, wq = GROUPS == 2 ? wave & 7 : wave;
    const int nkt = (N + 31) >> 5, nw = (nkt + TPW - 1) / TPW;
    const int nsuper = (nitems + GROUPS - 1) / GROUPS;
    const int d = h * 64;
    const int d3 = 3 * d;
    const int lr = lane & 31, lh = lane >> 5;
    const float c = scale * 1.44269504088896340736f;
    const RowOff ro = make_row_off(lane);
    const TrOff to = make_tr_off(lane);
    const uint32_t smix = seed_mix(seed);
    [[maybe_unused]] const uint32_t c4 = quad_c4(thresh);
    [[maybe_unused]] const uint32_t thi_mask = thresh >= 128u ? ~0u : 0u;
    const uint32_t bytes_q = (uint32_t)(((int64_t)(N - 1) * d3 + 64) * 2), bytes_o = (uint32_t)(((int64_t)(N - 1) * d + 64) * 2);
    const int q = wq * 32 + lr;                       // my query of the item (rows >= N: loads return 0, stores are dropped)
    const int qoff = (q * d3 + 8 * lh) * 2;

    // pieces of window w of super-item s -> slot; my group's item only (an item past the end: nothing is issued, its waves keep the barriers company)
    auto issue = [&](int s, int w, int slot) __attribute__((always_inline)) {
        const int item = s * GROUPS + group;
        if (item >= nitems) return;
        const int b = item / h, hd = item - b * h;
        const bf16_t *base = qkv + (int64_t)b * N * d3 + hd * 64;
        const int k0 = w * WK, nv = min(WK, N - k0), rp = ((nv + 31) >> 5) << 5;
        char *Kimg = smem + slot * SLOT + group * GB;
        dma_image<WG_>(Kimg, base + d + (int64_t)k0 * d3, d3, nv, rp, wq, lane);
        dma_image<WG_>(Kimg + WK * 128, base + 2 * d + (int64_t)k0 * d3, d3, nv, rp, wq, lane);
    };
    // the item's Q fragments as inline-asm buffer loads: hipcc's waitcnt pass would guard every use of a loop-carried load with vmcnt(0) -- in EVERY
    // tile, draining the next window's pieces in front of the window they are supposed to land under; issued this way the pass does not see them, and the
    // counted wait at the item's first window (vmcnt(NST): they are older than the stores) orders them by hand
    auto load_q = [&](int s, bf16x8 (&qf)[4]) __attribute__((always_inline)) {
        const int item = s * GROUPS + group;
        if (item >= nitems) return;
        const int b = item / h, hd = item - b * h;
        const uint64_t pa = (uint64_t)(uintptr_t)(qkv + (int64_t)b * N * d3 + hd * 64);
        const i32x4_t rq = i32x4_t{(int)(uint32_t)pa, (int)((pa >> 32) & 0xFFFFu), (int)bytes_q, 0x00020000};
        asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen offset:32\n\t"
                     "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:64\n\tbuffer_load_dwordx4 %3, %4, %5, 0 offen offset:96"
                     : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]) : "v"(qoff), "s"(rq) : "memory");
    };

    int s = blockIdx.x;
    if (s >= nsuper) return;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = bf16x8{};
    issue(s, 0, 0);
    load_q(s, qf);
    int t = 0;                                        // windows this workgroup has started: slot = t & 1
    for (; s < nsuper; s += gridDim.x) {
        const int item = s * GROUPS + group;
        const bool live = item < nitems;
        const int bh = live ? item : 0;
        const int b = bh / h, hd = bh - b * h;
        const int snext = s + (int)gridDim.x;
        const uint32_t rowquad = ((uint32_t)bh * (uint32_t)N + (uint32_t)(q < N ? q : N - 1)) * (uint32_t)((N + 3) >> 2);
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        float m = -INFINITY, l = 0.f;
        for (int w = 0; w < nw; ++w, ++t) {
            // my pieces of window t have landed (an item's first window: behind them stand my Q request -- landed too, it is older than the stores --
            // and the previous item's NST stores, which stay in flight); then the barrier
            if (w == 0 && t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // (LDS-DMA writes and LDS reads only: no fence -- __syncthreads() would wait for the stores in flight)
            if (w + 1 < nw) issue(s, w + 1, (t + 1) & 1);
            else if (snext < nsuper) issue(snext, 0, (t + 1) & 1);
            if (!live) continue;
            const char *Kimg = smem + (t & 1) * SLOT + group * GB, *Vimg = Kimg + WK * 128;
            const int ntile = min(TPW, nkt - w * TPW);
            for (int ktl = 0; ktl < ntile; ++ktl) {
                const int kt = w * TPW + ktl;
                // V^T fragments of this tile: requested now, consumed behind the softmax (8 transposed reads, 16 registers; the emitting dropout
                // instantiation has 8 registers less to spare: its second half is requested behind the softmax)
                constexpr bool VLATE = Q8 && DROP;
                bf16x4 vt[2][2][2];
                const uint32_t va = lds_addr_of(Vimg + ktl * 4096);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    vt[0][dt][0] = tr_read_asm_o<0>(va + to.lo[dt]);
                    vt[0][dt][1] = tr_read_asm_o<0>(va + to.hi[dt]);
                }
                if constexpr (!VLATE) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        vt[1][dt][0] = tr_read_asm_o<2048>(va + to.lo[dt]);
                        vt[1][dt][1] = tr_read_asm_o<2048>(va + to.hi[dt]);
                    }
                }
                f32x16 sc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(Kimg + ktl * 4096, ro.ks[ks]), qf[ks], sc, 0, 0, 0);
                if (kt == nkt - 1) {  // the item's last tile: padded keys, and the Q fragments are dead -- request the next item's into their registers
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key >= N) sc[r] = -INFINITY;
                    }
                    if (snext < nsuper) load_q(snext, qf);
                }
                u32x4 pk[2];
                attn_fwd_tile_vec<DROP>(sc, m, l, o, c, (rowquad + (uint32_t)(kt * 8 + lh)) * ECGVIT_WEYL + smix, c4, thi_mask, pk);
                // the V^T fragments have landed (the wait names them: nothing that reads them may be scheduled above it)
                if constexpr (VLATE) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        vt[1][dt][0] = tr_read_asm_o<2048>(va + to.lo[dt]);
                        vt[1][dt][1] = tr_read_asm_o<2048>(va + to.hi[dt]);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]),
                             "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]) :: "memory");
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pk[ss]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_halves(vt[ss][dt][0], vt[ss][dt][1]), pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (!live) continue;
        l += __shfl_xor(l, 32, 64);
        // output rows: the one-item kernel's 16-B runs (one v_permlane32_swap per dword), as bounds-checked buffer stores
        {
            const float inv = inv_keep / l;
            [[maybe_unused]] float q8_inv = 0.f;
            if constexpr (Q8) { const float sc8 = *q8_scale; q8_inv = sc8 > 0.f ? 1.0f / sc8 : 0.f; }
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (int64_t)b * N * d + hd * 64), 0, bytes_o, 0x00020000);
            const int ooff = (q * d + 16 * lh) * 2;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                [[maybe_unused]] u32x4 w8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint32_t D[2][2];
                    [[maybe_unused]] uint32_t W8[2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g4 = j + 2 * gg;
                        bf16x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = (bf16_t)(o[dt][4 * g4 + k] * inv);
                        const u32x2 vv = __builtin_bit_cast(u32x2, v);
                        D[gg][0] = vv[0]; D[gg][1] = vv[1];
                        if constexpr (Q8) {
                            float f[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                f[k] = (float)v[k];
                                qmax = fmaxf(qmax, q < N ? fabsf(f[k]) : 0.f);
                                f[k] = __builtin_amdgcn_fmed3f(f[k] * q8_inv, -448.f, 448.f);
                            }
                            int wv = 0;
                            wv = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], wv, false);
                            wv = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], wv, true);
                            W8[gg] = (uint32_t)wv;
                        }
                    }
                    u32x4 st;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(D[0][k], D[1][k], false, false);
                        st[k] = sw[0]; st[2 + k] = sw[1];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(st, rout, ooff + (dt * 32 + 8 * j) * 2, 0, 0);
                    if constexpr (Q8) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(W8[0], W8[1], false, false);
                        w8[2 * j] = sw[0]; w8[2 * j + 1] = sw[1];
                    }
                }
                if constexpr (Q8) {
                    const __amdgpu_buffer_rsrc_t rout8 = __builtin_amdgcn_make_buffer_rsrc((void *)(out8 + (int64_t)b * N * d + hd * 64), 0, bytes_o / 2, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(w8, rout8, q * d + 16 * lh + dt * 32, 0, 0);
                }
            }
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)(lse + (int64_t)bh * N), 0, (uint32_t)N * 4u, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m * scale + logf(l)), rl, lh == 0 ? q * 4 : 0x7FFFFFF0, 0, 0);
        }
    }
    // ONE publish per wave and launch, of the running maximum over all its items, against a FRESH look at the slot: a value peeked at the top of a
    // persistent kernel is stale for every item but the first -- inside the train step the slot is zeroed by the step's scale update, every wave of every
    // item then sent its atomic (65 k same-address atomics per launch at 256 x 16 x 501: +280 us per layer, found by the whole-line A/B of round 6)
    if constexpr (Q8) wave_amax_publish(q8_amax, qmax);
}

// =====================================================================================================
// backward
// =====================================================================================================
template <int NKT, bool DROP>
__global__ __launch_bounds__(NKT * 64) void attn_bwd_bf16_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out,
                                                                 const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                                 bf16_t *__restrict__ dqkv, int N, int h, float scale,
                                                                 uint64_t seed, uint32_t thresh, float inv_keep, int ablate) {
    constexpr int NK = NKT * 32, NT = NKT * 64;
    constexpr int IMG = NK * 128, DSB = NK * 64;
    __shared__ __attribute__((aligned(16))) char smem[3 * IMG + 2 * DSB + 2 * NK * 4];
    char *Qimg = smem, *dOimg = smem + IMG, *Kimg = smem + 2 * IMG, *dSimg = smem + 3 * IMG;
    float *lse_s = reinterpret_cast<float *>(smem + 3 * IMG + 2 * DSB), *delta_s = lse_s + NK;

    const int bh = blockIdx.x, b = bh / h, hd = bh - b * h;
    const int d = h * 64;
    const int64_t d3 = 3 * (int64_t)d;
    const bf16_t *base = qkv + (int64_t)b * N * d3 + hd * 64;
    const bf16_t *dobase = dout + (int64_t)b * N * d + hd * 64;
    const bf16_t *obase = out + (int64_t)b * N * d + hd * 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int mykey = wave * 32 + lr;
    // ---- prologue: EVERYTHING this block reads is put in flight before the first wait (one block per CU: nothing else hides
    //      the HBM latency): three images by LDS-DMA, V fragments / dO,O rows (for delta) / LSE by ordinary loads
    dma_image<NKT>(Qimg, base, d3, N, NK, wave, lane);
    dma_image<NKT>(Kimg, base + d, d3, N, NK, wave, lane);
    dma_image<NKT>(dOimg, dobase, d, N, NK, wave, lane);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 v4 = {0u, 0u, 0u, 0u};
        if (mykey < N) v4 = *reinterpret_cast<const u32x4 *>(base + 2 * d + (int64_t)mykey * d3 + ks * 16 + 8 * lh);
        vf[ks] = __builtin_bit_cast(bf16x8, v4);
    }
    {   // delta[q] = sum_dh dO[q][dh] * O[q][dh] ; 8 lanes per row, 16 B each; all 4 passes' loads issued back to back
        const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
        Vec16<bf16_t> va[4], vo[4];
        float ls[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * (NT / 8) + row;
            const int rc = r < N ? r : N - 1;
            va[ps] = ld16(dobase + (int64_t)rc * d + part * 8);
            vo[ps] = ld16(obase + (int64_t)rc * d + part * 8);
            ls[ps] = lse[(int64_t)bh * N + rc];
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * (NT / 8) + row;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += va[ps].get(k) * vo[ps].get(k);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0) {
                delta_s[r] = r < N ? acc : 0.f;
                lse_s[r] = r < N ? ls[ps] * 1.44269504088896340736f : 0.f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {   // K fragments (B operand of S = Q K^T) straight from the staged image
        const RowOff rk = make_row_off(lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = row_frag_c(Kimg + wave * 4096, rk.ks[ks]);
    }

    f32x16 dKt[2], dVt[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKt[dt][r] = 0.f; dVt[dt][r] = 0.f; }
    const float c = scale * 1.44269504088896340736f;
    const int nqb = (N + 31) >> 5;
    const RowOff ro = make_row_off(lane);
    const TrOff to = make_tr_off(lane);
    // dQ phase: lane-constant parts of the transposed reads (rows 4g + (i>>2) (+16), see the index permutation below)
    const int dq_g = lane >> 4, dq_i = lane & 15;
    const int dq_key = 4 * dq_g + (dq_i >> 2);
    const int dq_a[2] = {dq_key * 64 + (((0 * 4 + (dq_i & 3)) ^ dsw(dq_key)) << 3), dq_key * 64 + (((1 * 4 + (dq_i & 3)) ^ dsw(dq_key)) << 3)};
    int dq_b[4];
#pragma unroll
    for (int dhc = 0; dhc < 4; ++dhc) dq_b[dhc] = img_off(dq_key, (dhc * 16 + (dq_i & 3) * 4) * 2);

    for (int qb = (ablate & 4) ? nqb : 0; qb < nqb; ++qb) {
        const char *Qrow = Qimg + qb * 4096, *dOrow = dOimg + qb * 4096;
        // dropout: the forward kernel's function -- quad = (bh*N + q) * ceil(N/4) + key/4, byte = key & 3.  The 4 keys of a quad sit on 4
        // adjacent lanes: lane j of the quad hashes query j of each group of four and the others take it by a quad_perm broadcast
        const uint32_t qpitch = (uint32_t)((N + 3) >> 2);
        const uint32_t hstep = qpitch * ECGVIT_WEYL;   // one query down
        const uint32_t hq0 = seed_mix(seed) + (((uint32_t)bh * (uint32_t)N + (uint32_t)(qb * 32)) * qpitch + (uint32_t)(mykey >> 2)) * ECGVIT_WEYL;
        const uint32_t bsh = (uint32_t)(mykey & 3) * 8u, lq = (uint32_t)(lane & 3);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(Qrow, ro.ks[ks]), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(dOrow, ro.ks[ks]), vf[ks], dp, 0, 0, 0);
        }
        // rows of s/dp = queries qb*32 + (r&3) + 8*(r>>2) + 4*lh ; column = my key
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 l4 = *reinterpret_cast<const f32x4 *>(&lse_s[qb * 32 + 8 * g4 + 4 * lh]);
            const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&delta_s[qb * 32 + 8 * g4 + 4 * lh]);
            uint32_t hk[4];
            if constexpr (DROP) {
                const uint32_t mine = pair_finish(hq0 + ((uint32_t)(8 * g4 + 4 * lh) + lq) * hstep);   // query k = lane & 3 of this group
                hk[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x00, 0xF, 0xF, true);   // quad_perm:[0,0,0,0]
                hk[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
                hk[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
                hk[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xFF, 0xF, 0xF, true);   // [3,3,3,3]
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * g4 + k;
                float p = __builtin_amdgcn_exp2f(s[r] * c - l4[k]);
                float g = dp[r];
                if constexpr (DROP) {
                    const float mlt = ((hk[k] >> bsh) & 0xFFu) >= thresh ? inv_keep : 0.f;
                    g *= mlt;
                    s[r] = p * mlt;  // dropped probabilities feed dV
                } else {
                    s[r] = p;
                }
                dp[r] = p * (g - d4[k]) * scale;  // dS, in place
            }
        }
        if (!(ablate & 2))
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 pf = pack8(s, ss), dsf = pack8(dp, ss);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dVt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_c(dOrow + ss * 2048, to.lo[dt], to.hi[dt]), pf, dVt[dt], 0, 0, 0);
                dKt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_c(Qrow + ss * 2048, to.lo[dt], to.hi[dt]), dsf, dKt[dt], 0, 0, 0);
            }
        }
        // dS^T image [key][32 queries]: 64-B rows = 8 slots of 8 B; slot ^= dsw(key) (key bits 1,2,3 -> slot bits 0,2,1) makes
        // the 16-lane ds_write_b64 groups and the 8-row transposed reads below both conflict-free
        char *dsb = dSimg + (qb & 1) * DSB;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (bf16_t)dp[4 * g4 + k];
            const int slot = 2 * g4 + lh;   // queries 8*g4 + 4*lh .. +3
            *reinterpret_cast<bf16x4 *>(dsb + mykey * 64 + ((slot ^ dsw(mykey)) << 3)) = v;
        }
        // LDS-only barrier: a full __syncthreads() would also drain the dQ global stores of the previous block
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // dQ[32 x 64] = dS[32 x NK] . K[NK x 64] as 8 tiles of 16x16 (qt = tile&1, dhc = tile>>1).  The contraction index of
        // the 16x16x32 MFMA is permuted (k = 8g + j  <->  key = 32*st + 4g + (j&3) + 16*(j>>2)) so that each half-wave's
        // transposed read covers 8 CONSECUTIVE key rows of the dS and K images -- conflict-free on both.
        for (int tile = (ablate & 1) ? 8 : wave; tile < 8; tile += NKT) {
            const int qt = tile & 1, dhc = tile >> 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int g = lane >> 4, i = lane & 15;
            const int aoff = dq_a[qt], boff = dq_b[dhc];   // (tile is wave-uniform: these are selects, not scratch)
#pragma unroll 2
            for (int st = 0; st < NKT; ++st) {
                const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(dsb + st * 2048 + aoff));
                const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(dsb + st * 2048 + 1024 + aoff));
                const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(Kimg + st * 4096 + boff));
                const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(Kimg + st * 4096 + 2048 + boff));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_halves(b0, b1), join_halves(a0, a1), acc, 0, 0, 0);   // dQ^T tile: D[dh][q]
            }
            // lane (i = query, g -> 4 consecutive dh): one 8-byte store; the four g-groups of a query form 32 contiguous bytes
            const int q = qb * 32 + qt * 16 + i;
            if (q < N) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)acc[r];
                *reinterpret_cast<bf16x4 *>(dqkv + ((int64_t)b * N + q) * d3 + hd * 64 + dhc * 16 + 4 * g) = v;
            }
        }
    }
    // ---- epilogue: dK^T / dV^T (dh on rows, key on the lane) -> [key][dh] rows in the now idle Q / dO images -> 128-B row stores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 a, v;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = (bf16_t)dKt[dt][4 * g4 + k]; v[k] = (bf16_t)dVt[dt][4 * g4 + k]; }
            const int off = img_off(mykey, (dt * 32 + 8 * g4 + 4 * lh) * 2);
            *reinterpret_cast<bf16x4 *>(Qimg + off) = a;
            *reinterpret_cast<bf16x4 *>(dOimg + off) = v;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < NK * 8; c += NT) {
        const int row = c >> 3, ch = c & 7;
        if (row < N) {
            bf16_t *dst = dqkv + ((int64_t)b * N + row) * d3 + d + hd * 64 + ch * 8;
            *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(Qimg + img_off(row, ch * 16));
            *reinterpret_cast<u32x4 *>(dst + d) = *reinterpret_cast<const u32x4 *>(dOimg + img_off(row, ch * 16));
        }
    }
}

// =====================================================================================================
// backward, persistent (128 < N <= 256): one 512-thread block per CU walks (record, head) items.
// The one-item-per-block kernel above spends 42 % of its time with the matrix cores idle -- every block first waits for
// 190 KiB of inputs (a CU ingests ~11 B/cycle from a cold start) and ends with 64 KiB of dK / dV stores, and its 130 KiB of LDS
// leave no room for a second block to hide either.  Here the inputs are a STREAM that never drains:
//   * Q / dO / O arrive as 32-query slabs (3 x 4 KiB) through a four-slot ring, DMA'd three query blocks ahead of their use --
//     across the item boundary, so the next (record, head)'s first slabs are already in LDS when the current one finishes;
//   * the K image is double-buffered (the next item's K, its V fragments and LSE row are fetched during query block 1);
//   * delta = rowsum(dO * O) is computed per slab, cooperatively, one query block ahead;
//   * ONE counted wait (vmcnt) and ONE barrier per query block: a wave waits only for its own DMA pieces of slab j+2 while the
//     youngest operations (the slab just issued, the previous dQ store) stay in flight;
//   * dK / dV leave through per-wave 4-KiB patches in the dS buffers; their stores are still in flight when the next item starts.
// Math, fragment layouts, dropout indexing and the dQ tile scheme are those of the kernel above.

// Records longer than 256 tokens (N <= 512, e.g. patch 10 -> 501) run as TWO launches, one per half of the keys: `k0` is the first key
// of this launch's 256-key window, queries always run over all of N; the second launch adds its dQ to the first one's (ACCUM).
// PRIO: static wave priority for the whole kernel (no per-phase flips): 0 none, 1 waves 4-7 raised, 2 waves 0-3 raised
// Q8 (fp8_linear; bit 0: dK / dV, bit 1: dQ): additionally dqkv8 = saturate(dqkv as stored / *q8_scale) in e5m2 (same [B*N, 3*h*dh] layout, one
// byte per element) -- the A operand of the QKV projection's two backward products, written here instead of by a quantise pass over
// dqkv -- and *q8_amax = max(*q8_amax, max |dqkv|).  With two key windows the first launch emits its dK / dV only (dQ is final in the second).
template <bool DROP, bool ACCUM, bool STAGGER = true, int PRIO = 1, int Q8 = 0>
__global__ __launch_bounds__(512) void attn_bwd_pers_kernel(const bf16_t *__restrict__ qkv, const bf16_t *__restrict__ out,
                                                            const bf16_t *__restrict__ dout, const float *__restrict__ lse,
                                                            bf16_t *__restrict__ dqkv, int N, int h, float scale, uint64_t seed,
                                                            uint32_t thresh, float inv_keep, int nitems, int k0,
                                                            uint8_t *__restrict__ dqkv8 = nullptr, const float *__restrict__ q8_scale = nullptr,
                                                            float *__restrict__ q8_amax = nullptr) {
    constexpr int SQ = (Q8 & 2) ? 2 : 1;     // stores of one dQ tile (C phase)
    constexpr int SF = (Q8 & 1) ? 16 : 8;    // dK / dV stores of one item's flush
    [[maybe_unused]] float q8_inv = 0.f, qmax = 0.f;
    if constexpr (Q8 != 0) { const float sc = *q8_scale; q8_inv = sc > 0.f ? 1.0f / sc : 0.f; }
    constexpr int NQ = 512, IMG = 32768, DSB = 16384, SLAB = 12288;
    __shared__ __attribute__((aligned(1024))) char smem[2 * IMG + 4 * SLAB + 2 * DSB + 2 * NQ * 4 + 2 * 32 * 4];
    char *const Kimg0 = smem, *const slab0 = smem + 2 * IMG, *const dSimg = slab0 + 4 * SLAB;
    float *const lse_s = reinterpret_cast<float *>(dSimg + 2 * DSB), *const delta_s = lse_s + 2 * NQ;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int mykey = wave * 32 + lr;
    const bool late = wave >= 4;
    const int d = h * 64;
    const int d3 = 3 * d;
    const int nqb = (N + 31) >> 5;
    const float c = scale * 1.44269504088896340736f;
    const float log2_ik = DROP ? __builtin_log2f(inv_keep) : 0.f;
    const float inv_ik = 1.0f / inv_keep;
    // byte-permute selectors of the in-quad transpose of the dropout hash words (ph_V)
    [[maybe_unused]] const uint32_t sel1 = (threadIdx.x & 1) ? 0x03070105u : 0x06020400u;
    [[maybe_unused]] const uint32_t sel2 = (threadIdx.x & 2) ? 0x03020706u : 0x05040100u;
    const RowOff ro = make_row_off(lane);
    const TrOff to = make_tr_off(lane);
    const int dq_g = lane >> 4, dq_i = lane & 15;
    const int dq_key = 4 * dq_g + (dq_i >> 2);
    const int dq_a[2] = {dq_key * 64 + (((0 * 4 + (dq_i & 3)) ^ dsw(dq_key)) << 3), dq_key * 64 + (((1 * 4 + (dq_i & 3)) ^ dsw(dq_key)) << 3)};
    int dq_b[4];
#pragma unroll
    for (int dhc = 0; dhc < 4; ++dhc) dq_b[dhc] = img_off(dq_key, (dhc * 16 + (dq_i & 3) * 4) * 2);

    // per-lane DMA source offsets (bytes): a piece = 8 image rows; the image swizzle is applied to the SOURCE chunk
    const int prow = (wave & 3) * 8 + (lane >> 3);                          // slab piece row (0..31)
    const int pchunk = ((lane & 7) ^ swz3(prow)) * 16;
    const int vo_q = prow * d3 * 2 + pchunk, vo_d = prow * d * 2 + pchunk;  // Q (row pitch 3d) ; dO / O (row pitch d)
    int vo_k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        vo_k[i] = row * d3 * 2 + (((lane & 7) ^ swz3(row)) * 16);
    }
    const uint32_t bytes_q = (uint32_t)(((int64_t)(N - 1) * d3 + 64) * 2), bytes_d = (uint32_t)(((int64_t)(N - 1) * d + 64) * 2);
    const uint32_t bytes_k = (uint32_t)(((int64_t)(N - k0 - 1) * d3 + 64) * 2);   // K / V / dK / dV rows of this launch's key window (keys >= N: out of bounds)

    struct Item { const bf16_t *q, *o, *dO; int bh, b, hd; };
    auto make_item = [&](int it) {
        Item x;
        x.bh = it; x.b = it / h; x.hd = it - x.b * h;
        x.q = qkv + (int64_t)x.b * N * d3 + x.hd * 64;
        x.o = out + (int64_t)x.b * N * d + x.hd * 64;
        x.dO = dout + (int64_t)x.b * N * d + x.hd * 64;
        return x;
    };
    // slab qb of item x -> ring slot: waves 0-3 move one Q piece and one O piece each, waves 4-7 one dO piece each
    auto dma_slab = [&](const Item &x, int qb, int slot) {
        char *dst = slab0 + slot * SLAB + (wave & 3) * 1024;
        if (!late) {
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)x.q, 0, bytes_q, 0x00020000);
            const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void *)x.o, 0, bytes_d, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_p)dst, 16, vo_q, qb * 32 * d3 * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rO, (lds_void_p)(dst + 8192), 16, vo_d, qb * 32 * d * 2, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)x.dO, 0, bytes_d, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_p)(dst + 4096), 16, vo_d, qb * 32 * d * 2, 0, 0);
        }
    };
    auto dma_k = [&](const Item &x, char *img) {
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(x.q + d + (int64_t)k0 * d3), 0, bytes_k, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_p)(img + (wave + 8 * i) * 1024), 16, vo_k[i], 0, 0, 0);
    };
    // Every global access below is a bounds-checked BUFFER operation on a per-item descriptor (rows >= N read zero / are dropped by
    // the hardware): no lane predicate, so each wave issues the same number of memory instructions whatever N is -- the counted
    // waits in the loop depend on it.
    auto load_v = [&](const Item &x, bf16x8 (&v)[4]) {
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(x.q + 2 * d + (int64_t)k0 * d3), 0, bytes_k, 0x00020000);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            v[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rv, (mykey * d3 + ks * 16 + 8 * lh) * 2, 0, 0));
    };
    auto load_lse = [&](const Item &x) {
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)(lse + (int64_t)x.bh * N), 0, (uint32_t)N * 4u, 0x00020000);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (int)threadIdx.x * 4, 0, 0));   // scaled by log2(e) when stored
    };
    // delta of the slab in ring slot `slot` -> delta_s[buf][32]: wave w owns rows 4w..4w+3, 16 lanes per row, 4 elements per lane
    auto slab_delta = [&](int slot, int buf) {
        const int row = wave * 4 + (lane >> 4), e = (lane & 15) * 4;
        const char *sb = slab0 + slot * SLAB;
        const bf16x4 a = *reinterpret_cast<const bf16x4 *>(sb + 4096 + img_off(row, e * 2));
        const bf16x4 o = *reinterpret_cast<const bf16x4 *>(sb + 8192 + img_off(row, e * 2));
        float acc = (float)a[0] * (float)o[0] + (float)a[1] * (float)o[1] + (float)a[2] * (float)o[2] + (float)a[3] * (float)o[3];
        // 16-lane row reduction with DPP row shifts (no LDS crossbar): lane 15 of each row ends with the sum
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x118, 0xF, 0xF, true));   // row_shr:8
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x114, 0xF, 0xF, true));   // row_shr:4
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x112, 0xF, 0xF, true));   // row_shr:2
        acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x111, 0xF, 0xF, true));   // row_shr:1
        if ((lane & 15) == 15) delta_s[buf * 32 + row] = -(acc * inv_ik);   // -delta', delta' = delta (1 - p_drop): the probabilities below carry 1 / (1 - p_drop)
    };

    int it = blockIdx.x;
    if (it >= nitems) return;
    Item cur = make_item(it), nxt = cur;
    bf16x8 vf[4], vfn[4];
    float lse_n = 0.f;
    // ---- prologue of the block's first item: K, V, LSE, slabs 0..2
    dma_k(cur, Kimg0);
    load_v(cur, vf);
    {
        const float l0 = load_lse(cur);
        dma_slab(cur, 0, 0);
        dma_slab(cur, 1, 1);
        dma_slab(cur, 2, 2);
        lse_s[threadIdx.x] = log2_ik - l0 * 1.44269504088896340736f;   // -LSE' = log2(1 / (1 - p_drop)) - LSE log2 e: the exponential returns p / (1 - p_drop)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    slab_delta(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // the second-dispatched half of the workgroup loses VALU arbitration to the older half in every block (priority, then age): one
    // static priority raise for it, no per-phase flips (guide: two waves per SIMD, item 4)
    if (PRIO == 1 && late) __builtin_amdgcn_s_setprio(1);
    if (PRIO == 2 && !late) __builtin_amdgcn_s_setprio(1);
#ifdef ECGVIT_TOOLS
    unsigned long long *stamps = g_attn_stamps ? g_attn_stamps + (int64_t)blockIdx.x * 128 : nullptr;
#else
    unsigned long long *const stamps = nullptr;   // (every stamp below folds away)
#endif
#if defined(ECGVIT_TOOLS) && defined(ECGVIT_ATTN_PHASE_STAMPS)
    // `make tools TOOLS_EXTRA=-DECGVIT_ATTN_PHASE_STAMPS` only (tools/attn_phase_stamps.py): per-PHASE stamps of the second item, one record per wave
    // group (lane 0 of waves 0 and 4), behind the 768 block records of the buffer: [768 + block][group][query block][phase 0..7 = start, issue, A, V,
    // B+W, C, wait, barrier].  Not in the default tools build: the per-lane stamp conditions cost the staggered schedule 13 % (570 against 500 us)
    unsigned long long *pstamps = (g_attn_stamps && (lane == 0) && (wave == 0 || wave == 4)) ? g_attn_stamps + (768 + (int64_t)blockIdx.x) * 128 + (wave >> 2) * 64 : nullptr;
#define PH_STAMP(QB, IDX) do { if (pstamps && item_no == 1 && (QB) < 8) pstamps[(QB) * 8 + (IDX)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PH_STAMP(QB, IDX) do { } while (0)
#endif
    int item_no = 0;
    int slot = 0, par = 0, jj = 0;   // ring slot of the current slab, K / LSE buffer of the current item, running slab counter (delta / dS parity)
    for (;;) {
        const int next_it = it + (int)gridDim.x;
        const bool has_next = next_it < nitems;
        if (has_next) nxt = make_item(next_it);
        const char *Kimg = Kimg0 + par * IMG;
        const float *lse_c = lse_s + par * NQ;
        f32x16 dKt[2], dVt[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[dt][r] = 0.f; dVt[dt][r] = 0.f; }

        if (stamps && threadIdx.x == 0 && item_no < 4) stamps[item_no * 32] = __builtin_amdgcn_s_memtime();
        // ---- one query block = phases  issue | A: S, dP (8 MFMA) | V: softmax / dropout / dS arithmetic (VALU) | B: dV, dK (8 MFMA) |
        // W: dS -> LDS | wait + barrier | C: dQ tile (8 small MFMA) + store.  Block j lives in ring slot (slot0 + j) & 3 and uses
        // delta / dS buffer (jj0 + j) & 1.
        const int slot0 = slot, jj0 = jj;
        [[maybe_unused]] i32x4_t rdq_words;   // ACCUM: the dQ descriptor of this item as four dwords (operand of the inline-asm load)
        {
            const uintptr_t pa = reinterpret_cast<uintptr_t>(dqkv + (int64_t)cur.b * N * d3 + cur.hd * 64);
            rdq_words = i32x4_t{(int)(uint32_t)pa, (int)((pa >> 32) & 0xFFFFu), (int)bytes_q, 0x00020000};
        }
        f32x16 s, dp;            // scores / dP accumulators of the block whose A phase ran last
        uint32_t Pk[8], Dk[8];   // bf16 pairs (2m, 2m+1) of P (dropped) and dS: MFMA B operands AND the dS^T image rows
        // ACCUM (second key window): the first window's dQ values this wave adds to are requested at the TOP of the block, by inline asm --
        // a builtin load behind LDS-DMA pieces is waited for with vmcnt(0) by hipcc (it does not model the pieces), draining the stream
        // once per block (+15 % kernel time).  Older than everything else the block issues, the request is retired by the block's one
        // counted wait; the value is consumed behind that block's barrier (C phase).
        // (The trailing group consumes block j's value one block later than it requests block j+1's: two register pairs, moved, never indexed.)
        u32x2 dq_prev = {0u, 0u}, dq_req = {0u, 0u};
        auto ph_issue = [&](int qb) __attribute__((always_inline)) {
            if constexpr (ACCUM) {
                const int qt = wave & 1, dhc = wave >> 1;
                const uint32_t off = (uint32_t)(((qb * 32 + qt * 16 + dq_i) * d3 + dhc * 16 + 4 * dq_g) * 2);
                dq_prev = dq_req;   // block qb-1's value (landed: the previous block's counted wait covered it)
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dq_req) : "v"(off), "s"(rdq_words) : "memory");
            }
            const int sl = (slot0 + qb) & 3;
            // ---- feed the stream: next item's K / V / LSE once, slab qb+3 (of this item or the first slabs of the next)
            if (has_next && qb == 1) {
                dma_k(nxt, Kimg0 + (par ^ 1) * IMG);
                lse_n = load_lse(nxt);
            }
            if constexpr (!ATTN_ABL(6)) {
            if (qb + 3 < nqb) dma_slab(cur, qb + 3, (sl + 3) & 3);
            else if (has_next) dma_slab(nxt, qb + 3 - nqb, (sl + 3) & 3);
            }
            // delta of the NEXT slab (visible since the previous barrier), published by this iteration's barrier
            slab_delta((sl + 1) & 3, (jj0 + qb + 1) & 1);
        };
        auto ph_A = [&](int qb) __attribute__((always_inline)) {
            const char *Qrow = slab0 + ((slot0 + qb) & 3) * SLAB, *dOrow = Qrow + 4096;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            if constexpr (ATTN_ABL(4)) { asm volatile("" : "+v"(s), "+v"(dp)); return; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {   // K fragments of this wave's 32 keys come from the image every time (4 reads): 16 VGPRs less
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(Qrow, ro.ks[ks]), row_frag_c(Kimg + wave * 4096, ro.ks[ks]), s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_c(dOrow, ro.ks[ks]), vf[ks], dp, 0, 0, 0);
            }
        };
        // Vector work of a block: 9.5 instructions per score element (13.5 until round 3; the kernel is bound by vector-instruction issue,
        // profiles/r04_valu_rate.txt): the stored -LSE' carries log2(1 / (1 - p_drop)), so the exponential returns p' = p / (1 - p_drop) at once;
        // P_dropped = keep ? p' : 0 is ONE byte-select compare + ONE select, on a hash word whose bytes were transposed inside the lane quad (the
        // lane of key byte b hashes query b of the group: 2 DPP moves + 2 byte permutes per FOUR elements put query k's byte at position k);
        // dS' = P_dropped dP - p' delta' with delta' = delta (1 - p_drop); the factor `scale` is applied to the dQ tile and the dK flush instead
        // of every element (exact for the power-of-two dh^-1/2).
        auto ph_V = [&](int qb) __attribute__((always_inline)) {
            if constexpr (ATTN_ABL(3)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) { Pk[m] = cvt_pk_bf16(s[2 * m], s[2 * m + 1]); Dk[m] = cvt_pk_bf16(dp[2 * m], dp[2 * m + 1]); }
                return;
            }
            const float *delta_c = delta_s + ((jj0 + qb) & 1) * 32;
            const uint32_t qpitch = (uint32_t)((N + 3) >> 2);
            const uint32_t hstep = qpitch * ECGVIT_WEYL;
            const uint32_t lq = (uint32_t)(lane & 3);
            const uint32_t hq0 = seed_mix(seed) + (((uint32_t)cur.bh * (uint32_t)N + (uint32_t)(qb * 32 + 4 * lh) + lq) * qpitch + (uint32_t)((mykey + k0) >> 2)) * ECGVIT_WEYL;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = *reinterpret_cast<const f32x4 *>(&lse_c[qb * 32 + 8 * g4 + 4 * lh]);
                const f32x4 d4 = *reinterpret_cast<const f32x4 *>(&delta_c[8 * g4 + 4 * lh]);
                [[maybe_unused]] uint32_t X = 0u;
                if constexpr (DROP) {
                    const uint32_t mine = pair_finish(hq0 + (uint32_t)(8 * g4) * hstep);   // query (lane & 3) of this group, my key quad
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
                    const uint32_t t1 = __builtin_amdgcn_perm(nb, mine, sel1);
                    const uint32_t nb2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t1, 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
                    X = __builtin_amdgcn_perm(nb2, t1, sel2);                                                          // byte k = query k's byte of MY key
                }
                // (two elements per vector instruction where the operation has a packed f32 form -- the exponent's fma, p delta, the dS fma: the
                // same IEEE operations per element, 8 instead of 9.5 per element; round 5)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int r = 4 * g4 + 2 * k2;
                    const f32x2 a = __builtin_elementwise_fma(f32x2{s[r], s[r + 1]}, f32x2{c, c}, f32x2{l4[2 * k2], l4[2 * k2 + 1]});
                    const f32x2 p = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                    f32x2 pd = p;
                    if constexpr (DROP) {
                        pd[0] = ((X >> (16 * k2)) & 0xFFu) >= thresh ? p[0] : 0.f;
                        pd[1] = ((X >> (16 * k2 + 8)) & 0xFFu) >= thresh ? p[1] : 0.f;
                    }
                    const f32x2 u = p * f32x2{d4[2 * k2], d4[2 * k2 + 1]};                        // -p' delta' (the row holds -delta')
                    const f32x2 ds = __builtin_elementwise_fma(pd, f32x2{dp[r], dp[r + 1]}, u);    // dS / scale
                    s[r] = pd[0]; s[r + 1] = pd[1];   // dropped probabilities feed dV
                    dp[r] = ds[0]; dp[r + 1] = ds[1];
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) { Pk[m] = cvt_pk_bf16(s[2 * m], s[2 * m + 1]); Dk[m] = cvt_pk_bf16(dp[2 * m], dp[2 * m + 1]); }
        };
        auto ph_B = [&](int qb) __attribute__((always_inline)) {
            if constexpr (ATTN_ABL(2)) { asm volatile("" : "+v"(dVt[0]), "+v"(dKt[0]) : "v"(Pk[0]), "v"(Dk[0])); return; }
            const char *Qrow = slab0 + ((slot0 + qb) & 3) * SLAB, *dOrow = Qrow + 4096;
            uint32_t qa[4], da[4];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                qa[2 * dt] = lds_addr_of(Qrow) + to.lo[dt]; qa[2 * dt + 1] = lds_addr_of(Qrow) + to.hi[dt];
                da[2 * dt] = lds_addr_of(dOrow) + to.lo[dt]; da[2 * dt + 1] = lds_addr_of(dOrow) + to.hi[dt];
            }
            // both sub-steps' transposed fragments are requested up front (16 reads; the score / dP accumulators are dead here, their registers
            // hold them): the first group of MFMAs waits for the first eight only (counted lgkmcnt), the second group's reads land under it --
            // one exposed LDS round trip per block instead of two (round 4)
#define TRQ(SS, T)                                                                                                          \
            T[0] = tr_read_asm_o<SS * 2048>(da[0]); T[1] = tr_read_asm_o<SS * 2048>(da[1]);                                 \
            T[2] = tr_read_asm_o<SS * 2048>(qa[0]); T[3] = tr_read_asm_o<SS * 2048>(qa[1]);                                 \
            T[4] = tr_read_asm_o<SS * 2048>(da[2]); T[5] = tr_read_asm_o<SS * 2048>(da[3]);                                 \
            T[6] = tr_read_asm_o<SS * 2048>(qa[2]); T[7] = tr_read_asm_o<SS * 2048>(qa[3]);
#define TRM(SS, T, CNT)                                                                                                     \
            {                                                                                                               \
                u32x4 pu, du;                                                                                               \
                pu[0] = Pk[4 * SS]; pu[1] = Pk[4 * SS + 1]; pu[2] = Pk[4 * SS + 2]; pu[3] = Pk[4 * SS + 3];                 \
                du[0] = Dk[4 * SS]; du[1] = Dk[4 * SS + 1]; du[2] = Dk[4 * SS + 2]; du[3] = Dk[4 * SS + 3];                 \
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pu), dsf = __builtin_bit_cast(bf16x8, du);                     \
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                          \
                dVt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_halves(T[0], T[1]), pf, dVt[0], 0, 0, 0);             \
                dKt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_halves(T[2], T[3]), dsf, dKt[0], 0, 0, 0);            \
                dVt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_halves(T[4], T[5]), pf, dVt[1], 0, 0, 0);             \
                dKt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(join_halves(T[6], T[7]), dsf, dKt[1], 0, 0, 0);            \
            }
            bf16x4 tA[8], tB[8];
            TRQ(0, tA)
            TRQ(1, tB)
            TRM(0, tA, 8)
            TRM(1, tB, 0)
#undef TRQ
#undef TRM
        };
        auto ph_W = [&](int qb) __attribute__((always_inline)) {
            if constexpr (ATTN_ABL(5)) return;
            char *dsb = dSimg + ((jj0 + qb) & 1) * DSB;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 v;
                v[0] = Dk[2 * g4]; v[1] = Dk[2 * g4 + 1];
                const int slt = 2 * g4 + lh;
                *reinterpret_cast<u32x2 *>(dsb + mykey * 64 + ((slt ^ dsw(mykey)) << 3)) = v;
            }
        };
        // ---- the block's one wait + barrier.  Needed: my DMA pieces of slab qb+2 (issued one block ago).  Still allowed in
        // flight, youngest first: this block's slab pieces (2 on waves 0-3, 1 on waves 4-7), the K / LSE prefetch of query
        // block 1 (4 + 1), one dQ store (the previous block's) -- or, on an item's first query block, the 8 dK / dV stores + last dQ store.
        auto ph_waitbar = [&](int qb) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            if (stamps && threadIdx.x == 0 && item_no < 4 && qb < 8) stamps[item_no * 32 + 1 + qb] = __builtin_amdgcn_s_memtime();
            // allowed in flight (youngest first), with SQ = stores of a dQ tile and SF = dK / dV stores of an item's flush:
            //   leading:  this block's 2 slab pieces [+ K / LSE prefetch 5 at qb == 1] + the previous dQ tile's SQ stores [qb == 0: + SF + SQ of the previous item]
            //   trailing: 1 slab piece [+ 5] + SQ [qb == 0: + SF + SQ]
            //   ACCUM: this block's dQ request is OLDER than its pieces and must have landed: leading 2 [+ 5]; trailing 1 + SQ [+ 5]; on an
            //   item's first block it is YOUNGER than the previous item's stores: 2 / 1
#define ATTN_WAIT(N_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory")
            if (!has_next) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            } else if (late) {
                // (lockstep order: the trailing waves' dQ stores of the previous block are OLDER than this block's request)
                constexpr int SQL = (ACCUM && !STAGGER) ? 0 : SQ;
                if (qb == 0) { if (ACCUM) ATTN_WAIT(1); else ATTN_WAIT(1 + SF + SQ); }
                else if (qb == 1) ATTN_WAIT(6 + SQL);
                else ATTN_WAIT(1 + SQL);
            } else if (!ACCUM) {
                if (qb == 0) ATTN_WAIT(2 + SF + SQ);
                else if (qb == 1) ATTN_WAIT(7 + SQ);
                else ATTN_WAIT(2 + SQ);
            } else {
                if (qb == 0) ATTN_WAIT(2);
                else if (qb == 1) ATTN_WAIT(7);
                else ATTN_WAIT(2);
            }
#undef ATTN_WAIT
            if (stamps && threadIdx.x == 0 && item_no < 4 && qb < 8) stamps[item_no * 32 + 9 + qb] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (stamps && threadIdx.x == 0 && item_no < 4 && qb < 8) stamps[item_no * 32 + 17 + qb] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            // the next item's V fragments start travelling behind the item's last barrier
            if (has_next && qb == nqb - 1) load_v(nxt, vfn);
        };
        // ---- dQ tile of this wave (one 16 x 16 tile: qt = wave&1, dhc = wave>>1), then its store
        auto ph_C = [&](int qb, bool use_prev = false) __attribute__((always_inline)) {
            const char *dsb = dSimg + ((jj0 + qb) & 1) * DSB;
            const int qt = wave & 1, dhc = wave >> 1;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int aoff = dq_a[qt], boff = dq_b[dhc];
            const uint32_t sa = lds_addr_of(dsb) + aoff, ka = lds_addr_of(Kimg) + boff;
            // the four key-step groups are software-pipelined by one: group h+1's eight transposed reads are in flight while group h's two
            // MFMAs run (counted lgkmcnt; after the barrier 48+ registers of the block's arithmetic are dead): one exposed LDS round trip
            // per dQ tile instead of four (round 4)
#define DQR(HF, T)                                                                                                           \
            T[0] = tr_read_asm_o<(2 * HF) * 2048>(sa); T[1] = tr_read_asm_o<(2 * HF) * 2048 + 1024>(sa);                     \
            T[2] = tr_read_asm_o<(2 * HF) * 4096>(ka); T[3] = tr_read_asm_o<(2 * HF) * 4096 + 2048>(ka);                     \
            T[4] = tr_read_asm_o<(2 * HF + 1) * 2048>(sa); T[5] = tr_read_asm_o<(2 * HF + 1) * 2048 + 1024>(sa);             \
            T[6] = tr_read_asm_o<(2 * HF + 1) * 4096>(ka); T[7] = tr_read_asm_o<(2 * HF + 1) * 4096 + 2048>(ka);
#define DQM(T, CNT)                                                                                                          \
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_halves(T[2], T[3]), join_halves(T[0], T[1]), acc, 0, 0, 0);   \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join_halves(T[6], T[7]), join_halves(T[4], T[5]), acc, 0, 0, 0);   \
            __builtin_amdgcn_sched_barrier(0);
            bf16x4 tA[8], tB[8];
            if constexpr (!ATTN_ABL(1)) {
            DQR(0, tA)
            DQR(1, tB)
            DQM(tA, 8)
            DQR(2, tA)
            DQM(tB, 8)
            DQR(3, tB)
            DQM(tA, 8)
            DQM(tB, 0)
            }
#undef DQR
#undef DQM
            const int q = qb * 32 + qt * 16 + dq_i;
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[r] * scale);   // (dS is carried without `scale`: ph_V)
            const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv + (int64_t)cur.b * N * d3 + cur.hd * 64), 0, bytes_q, 0x00020000);
            if constexpr (ACCUM) {   // second key window: dQ += (the first launch's dQ, bf16; requested in ph_issue of this block)
                // leading group and the item's last block: C(qb) runs behind the barrier of the block that requested it (dq_req);
                // trailing group otherwise: one block later, after the next request went out (dq_prev)
                const bf16x4 o = __builtin_bit_cast(bf16x4, use_prev ? dq_prev : dq_req);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[r] * scale + (float)o[r]);
            }
            if constexpr (!ATTN_ABL(7))
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rdq, (q * d3 + dhc * 16 + 4 * dq_g) * 2, 0, 0);   // rows >= N: dropped
            if constexpr ((Q8 & 2) != 0) {
                float f[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[r] = (float)v[r];                            // the value as stored
                    qmax = fmaxf(qmax, q < N ? fabsf(f[r]) : 0.f);
                    f[r] = __builtin_amdgcn_fmed3f(f[r] * q8_inv, -57344.f, 57344.f);
                }
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], w, false);
                w = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], w, true);
                const __amdgpu_buffer_rsrc_t rdq8 = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv8 + (int64_t)cur.b * N * d3 + cur.hd * 64), 0, bytes_q / 2, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(w, rdq8, q * d3 + dhc * 16 + 4 * dq_g, 0, 0);
            }
        };
        // The two waves of a SIMD (w and w + 4) would run the same phases between the same barriers -- MFMA phases together, VALU
        // phases together (22 % MFMA-busy, 43 % of wave time waiting).  Waves 4-7 run HALF A BLOCK LATE instead: between two
        // barriers they do V, B, W of block j, then the dQ tile of block j-1 (its dS buffer is not rewritten before block j+1)
        // and the S / dP products of block j+1 (its slab is visible since the barrier before), so that their VALU phase
        // meets the leading group's MFMA phases (C, A) and their MFMA phases (B, C, A) meet its VALU phase.  Results are unchanged bit
        // for bit: every block's arithmetic is the same instruction sequence, only its placement relative to the barriers moves.
        if (!STAGGER || !late) {
            for (int qb = 0; qb < nqb; ++qb) {
                PH_STAMP(qb, 0);
                ph_issue(qb);
                PH_STAMP(qb, 1);
                ph_A(qb);
                PH_STAMP(qb, 2);
                ph_V(qb);
                PH_STAMP(qb, 3);
                ph_B(qb);
                ph_W(qb);   // (W ahead of B, the trailing group's order, was measured here too: 504.4 against 503.1 us)
                PH_STAMP(qb, 4);
                PH_STAMP(qb, 5);
                ph_waitbar(qb);
                PH_STAMP(qb, 7);
                ph_C(qb);
            }
        } else {
            // (first and last block peeled: a conditionally executed A phase would merge two versions of the 32 accumulator registers)
            ph_issue(0);
            ph_A(0);
            ph_V(0);
            ph_W(0);
            ph_B(0);
            ph_A(1);
            ph_waitbar(0);
            for (int qb = 1; qb < nqb - 1; ++qb) {
                PH_STAMP(qb, 0);
                ph_issue(qb);
                PH_STAMP(qb, 1);
                ph_V(qb);
                PH_STAMP(qb, 3);
                ph_W(qb);
                ph_B(qb);
                PH_STAMP(qb, 4);
                ph_C(qb - 1, true);
                PH_STAMP(qb, 5);
                ph_A(qb + 1);
                PH_STAMP(qb, 2);
                ph_waitbar(qb);
                PH_STAMP(qb, 7);
            }
            ph_issue(nqb - 1);
            ph_V(nqb - 1);
            ph_W(nqb - 1);
            ph_B(nqb - 1);
            ph_C(nqb - 2, true);
            ph_waitbar(nqb - 1);
            ph_C(nqb - 1);
        }
#undef PH_STAMP
        slot = (slot0 + nqb) & 3;
        jj = jj0 + nqb;
        if (stamps && threadIdx.x == 0 && item_no < 4) stamps[item_no * 32 + 25] = __builtin_amdgcn_s_memtime();
        // ---- item done: dK^T / dV^T (dh on rows, key on the lane) -> this wave's 32 [key][dh] rows in a private 4-KiB patch -> 128-B rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // every wave is done reading both dS buffers
        {
            char *patch = dSimg + wave * 4096;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        bf16x4 a;
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = (bf16_t)(which == 0 ? dKt[dt][4 * g4 + k] * scale : dVt[dt][4 * g4 + k]);
                        *reinterpret_cast<bf16x4 *>(patch + img_off(lr, (dt * 32 + 8 * g4 + 4 * lh) * 2)) = a;
                    }
                // same-wave LDS operations execute in order: read the rows back (8 rows x 128 B per instruction) and store them
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3), ch = lane & 7;
                    const int key = wave * 32 + row;
                    const u32x4 val = *reinterpret_cast<const u32x4 *>(patch + img_off(row, ch * 16));
                    const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv + ((int64_t)cur.b * N + k0) * d3 + (1 + which) * d + cur.hd * 64), 0, bytes_k, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(val, rkv, (key * d3 + ch * 8) * 2, 0, 0);   // keys >= N: dropped
                    if constexpr ((Q8 & 1) != 0) {
                        float f[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { f[2 * k] = __builtin_bit_cast(float, val[k] << 16); f[2 * k + 1] = __builtin_bit_cast(float, val[k] & 0xFFFF0000u); }
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
                        const __amdgpu_buffer_rsrc_t rkv8 = __builtin_amdgcn_make_buffer_rsrc((void *)(dqkv8 + ((int64_t)cur.b * N + k0) * d3 + (1 + which) * d + cur.hd * 64), 0, bytes_k / 2, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b64(q2, rkv8, key * d3 + ch * 8, 0, 0);
                    }
                }
            }
        }
        if (stamps && threadIdx.x == 0 && item_no < 4) stamps[item_no * 32 + 26] = __builtin_amdgcn_s_memtime();
        ++item_no;
        if (!has_next) break;
        // ---- switch to the next item: its K image and LSE row were fetched during query block 1, its V fragments during the drain
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[ks] = vfn[ks];
        lse_s[(par ^ 1) * NQ + threadIdx.x] = log2_ik - lse_n * 1.44269504088896340736f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // patches drained (the next dS write may reuse them); LSE row visible
        par ^= 1;
        it = next_it;
        cur = nxt;
    }
    if constexpr (Q8 != 0) {   // at most one atomic max per wave and launch (common.h: wave_amax_publish)
        wave_amax_publish(q8_amax, qmax);
    }
}

#ifdef ECGVIT_TOOLS
// probe: exact-integer dump of what each lane receives from the fragment helpers (tests pin the layouts with it)
__global__ __launch_bounds__(64) void probe_kernel(float *out) {
    __shared__ __attribute__((aligned(16))) char img[64 * 128];
    const int lane = threadIdx.x;
    // image X[row][col] = row * 64 + col  (exact in bf16 only up to 256, so use row*2 + col/32 style codes per probe)
    for (int c = lane; c < 64 * 64; c += 64) {
        const int row = c >> 6, col = c & 63;
        *reinterpret_cast<bf16_t *>(img + img_off(row, col * 2)) = (bf16_t)(float)(row * 2 + (col >> 5));  // <= 127: exact
    }
    __syncthreads();
    const bf16x8 rf = row_frag(img, 32, 1, lane);          // rows 32.., k-step 1 (cols 16..31 -> col>>5 = 0)
    const bf16x8 tf = tr_frag32(img, 16, 32, lane);        // rows 16.., cols 32.. (col>>5 = 1)
    for (int j = 0; j < 8; ++j) {
        out[(0 * 64 + lane) * 16 + j] = (float)rf[j];
        out[(1 * 64 + lane) * 16 + j] = (float)tf[j];
    }
    // second image pass: code = col (0..63) to pin the column each lane gets
    __syncthreads();
    for (int c = lane; c < 64 * 64; c += 64) {
        const int row = c >> 6, col = c & 63;
        *reinterpret_cast<bf16_t *>(img + img_off(row, col * 2)) = (bf16_t)(float)col;
    }
    __syncthreads();
    const bf16x8 rf2 = row_frag(img, 32, 1, lane);
    const bf16x8 tf2 = tr_frag32(img, 16, 32, lane);
    for (int j = 0; j < 8; ++j) {
        out[(0 * 64 + lane) * 16 + 8 + j] = (float)rf2[j];
        out[(1 * 64 + lane) * 16 + 8 + j] = (float)tf2[j];
    }
    // MFMA C layout: A[i][k] = (k == 0) ? i : 0 ; B[k][j] = (k == 0) ? 1 : 0 (+ second product coding the column)
    bf16x8 a, bcol, ones;
    for (int j = 0; j < 8; ++j) { a[j] = (bf16_t)0.f; bcol[j] = (bf16_t)0.f; ones[j] = (bf16_t)0.f; }
    if ((lane >> 5) == 0) { a[0] = (bf16_t)(float)(lane & 31); ones[0] = (bf16_t)1.f; bcol[0] = (bf16_t)(float)(lane & 31); }
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    const f32x16 rowcode = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, z, 0, 0, 0);   // D[i][j] = i
    const f32x16 colcode = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bcol, z, 0, 0, 0);  // D[i][j] = j
    for (int r = 0; r < 16; ++r) {
        out[(2 * 64 + lane) * 16 + r] = rowcode[r];
        out[(3 * 64 + lane) * 16 + r] = colcode[r];
    }
}
#endif   // ECGVIT_TOOLS (probe)


// ---- export of the post-softmax probabilities of the fused path (next row f3): P[bh][q][k] = exp(scale * q.k - lse[bh][q]), rebuilt from
//      the qkv and log-sum-exp the fused forward already keeps for backward. Visualisation-time only (vit_pytorch Recorder, reference
//      ecg_vit.py:176-180), so a plain VALU kernel: one wave per query row, lanes over keys, f32 accumulation of the bf16 products.
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16_t *__restrict__ qkv, const float *__restrict__ lse, float *__restrict__ probs,
                                                         int N, int h, int dh, float scale) {
    const int bh = blockIdx.y, b = bh / h, head = bh % h;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= N) return;
    const int64_t ld = 3 * (int64_t)h * dh;
    const bf16_t *qrow = qkv + ((int64_t)b * N + q) * ld + head * dh;
    const float l = lse[(int64_t)bh * N + q];
    for (int k = lane; k < N; k += 64) {
        const bf16_t *krow = qkv + ((int64_t)b * N + k) * ld + (int64_t)h * dh + head * dh;
        float acc = 0.f;
        for (int e = 0; e < dh; e += 8) {
            const Vec16<bf16_t> a = ld16(qrow + e), c = ld16(krow + e);
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = fmaf(a.get(t), c.get(t), acc);
        }
        probs[((int64_t)bh * N + q) * N + k] = expf(acc * scale - l);
    }
}

}  // namespace

#ifdef ECGVIT_TOOLS
static int g_tools_attn_variant = -1;
extern "C" int ecgvit_tools_attn_variant(int v) { g_tools_attn_variant = v; return ECGVIT_OK; }
static int g_tools_attn_fwd_variant = -1;   // -1: the product's dispatch; 0: always the one-item forward; 1: always the streamed forward
extern "C" int ecgvit_tools_attn_fwd_variant(int v) { g_tools_attn_fwd_variant = v; return ECGVIT_OK; }
#endif

extern "C" {

#ifdef ECGVIT_TOOLS   // declared in tools/ecgvit_hip_tools.h, exported by build/libecgvit_hip_tools.so only
int ecgvit_debug_attn_stamps(void *buf) {   // diagnostics: 768 workgroups x 128 uint64 cycle stamps written by the eight-wave persistent backward; NULL = off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &buf, sizeof(buf)) == hipSuccess ? ECGVIT_OK : ECGVIT_ELAUNCH;
}
#endif

int ecgvit_attention_probs(const void *qkv, const float *lse, float *probs, int B, int N, int h, int dh, float scale, int dtype, void *stream) {
    if (dtype != ECGVIT_BF16 || dh % 8 || B <= 0 || N <= 0 || h <= 0 || (int64_t)B * h > 65535) return ECGVIT_EINVAL;
    hipLaunchKernelGGL(attn_probs_kernel, dim3((N + 3) / 4, B * h), dim3(256), 0, as_stream(stream), (const bf16_t *)qkv, lse, probs, N, h, dh, scale);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

static int attention_fwd_launch(const void *qkv, void *out, float *lse, int B, int N, int h, int dh, float scale, float dropout_p,
                                uint64_t seed, int dtype, void *stream, void *out8, const float *q8_scale, float *q8_amax) {
    if (dtype != ECGVIT_BF16 || dh != 64 || N < 1 || N > 512 || B < 1 || h < 1) return ECGVIT_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out)) % 16) return ECGVIT_EINVAL;
    if (out8 && (!q8_scale || !q8_amax || reinterpret_cast<uintptr_t>(out8) % 16)) return ECGVIT_EINVAL;   // (16-B stores)
    if (dropout_p > 0.f && dropout_threshold8(dropout_p) == 0) return ECGVIT_EINVAL;   // p < 1/512 would silently round to no dropout
    const uint32_t th = dropout_threshold8(dropout_p);
    const float ik = dropout_inv_keep8(dropout_p);
    // the STREAMED form (one persistent 16-wave workgroup per CU, K / V windows through a two-slot ring) for records of more than 256 tokens once there
    // is an item per CU (256 x 16 x 501: 494 against 533 us, 8-bit emitting 523 against 575; profiles/r06_attn_fwd_stream.txt); below that the one-item
    // kernel, whose B*h workgroups spread over more CUs.  Up to 256 tokens the one-item kernel stays (512 x 12 x 251: 215 against 236 us streamed -- with two
    // items side by side a window is four key tiles, and the sixteen waves meet at a barrier every four tiles)
    {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return ECGVIT_ELAUNCH;
            n_cu = v;
        }
        // (MODE 2 / 3 -- the two forms for up to 256 tokens, both slower than the one-item kernel there -- exist in the TOOLS build only: tools/attn_fwd_ab.py,
        // tests/test_gpu_ops.py hold them bit for bit against the one-item kernel)
        int mode = 1;
#ifdef ECGVIT_TOOLS
        if (N <= 256) mode = g_tools_attn_fwd_variant == 2 ? 3 : 2;
#endif
        const int groups = mode == 2 ? 2 : 1, nsuper = (B * h + groups - 1) / groups;
        bool stream_form = N > 256 && nsuper >= n_cu && (int64_t)N * 3 * h * 64 * 2 < (1ll << 31);
#ifdef ECGVIT_TOOLS
        if (g_tools_attn_fwd_variant == 0) stream_form = false;
        if (g_tools_attn_fwd_variant >= 1) stream_form = (int64_t)N * 3 * h * 64 * 2 < (1ll << 31);
#endif
#ifdef ECGVIT_AB_NO_STREAM
        stream_form = false;   // (A/B builds only)
#endif
        if (stream_form) {
            static bool sattr = false;
            if (!sattr) {
#define SATTR(DR, Q, G) if (hipFuncSetAttribute((const void *)attn_fwd_stream_kernel<DR, Q, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return ECGVIT_ELAUNCH
                SATTR(true, false, 1); SATTR(false, false, 1); SATTR(true, true, 1); SATTR(false, true, 1);
#ifdef ECGVIT_TOOLS
                SATTR(true, false, 2); SATTR(false, false, 2); SATTR(true, true, 2); SATTR(false, true, 2);
                SATTR(true, false, 3); SATTR(false, false, 3); SATTR(true, true, 3); SATTR(false, true, 3);
#endif
#undef SATTR
                sattr = true;
            }
            const int slots = mode == 3 ? 2 * n_cu : n_cu;
            const dim3 sg((unsigned)(nsuper < slots ? nsuper : slots)), sb(mode == 3 ? 512 : 1024);
            const size_t slds = mode == 3 ? 64 * 1024 : 128 * 1024;
#define SFWD(DR, Q, G) hipLaunchKernelGGL((attn_fwd_stream_kernel<DR, Q, G>), sg, sb, slds, as_stream(stream), (const bf16_t *)qkv, (bf16_t *)out, lse, N, h, scale, seed, th, ik, B * h, (uint8_t *)out8, q8_scale, q8_amax)
#ifdef ECGVIT_TOOLS
#define SFWD2(DR, Q) do { if (mode == 1) SFWD(DR, Q, 1); else if (mode == 2) SFWD(DR, Q, 2); else SFWD(DR, Q, 3); } while (0)
#else
#define SFWD2(DR, Q) SFWD(DR, Q, 1)
#endif
            if (out8) { if (th) SFWD2(true, true); else SFWD2(false, true); }
            else { if (th) SFWD2(true, false); else SFWD2(false, false); }
#undef SFWD2
#undef SFWD
            ECGVIT_CHECK_LAUNCH();
            return ECGVIT_OK;
        }
    }
    const bool split = N > 256;   // two 256-key windows through 64 KiB of images, one workgroup per 256-query half
    dim3 grid((unsigned)(B * h * (split ? (N + 255) / 256 : 1)));
    const size_t lds = split ? (size_t)256 * 128 * 2 : (size_t)((N + 31) / 32) * 32 * 128 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)attn_fwd_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ECGVIT_ELAUNCH;
        if (hipFuncSetAttribute((const void *)attn_fwd_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ECGVIT_ELAUNCH;
        if (hipFuncSetAttribute((const void *)attn_fwd_bf16_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ECGVIT_ELAUNCH;
        if (hipFuncSetAttribute((const void *)attn_fwd_bf16_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return ECGVIT_ELAUNCH;
        attr_set = true;
    }
#define FWD(DR, Q, SP) hipLaunchKernelGGL((attn_fwd_bf16_kernel<DR, Q, SP>), grid, dim3(512), lds, as_stream(stream), (const bf16_t *)qkv, (bf16_t *)out, lse, N, h, scale, seed, th, ik, (uint8_t *)out8, q8_scale, q8_amax)
    if (split) {
        if (out8) { if (th) FWD(true, true, true); else FWD(false, true, true); }
        else { if (th) FWD(true, false, true); else FWD(false, false, true); }
    } else if (out8) { if (th) FWD(true, true, false); else FWD(false, true, false); }
    else { if (th) FWD(true, false, false); else FWD(false, false, false); }
#undef FWD
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

int ecgvit_attention_fwd(const void *qkv, void *out, float *lse, int B, int N, int h, int dh, float scale, float dropout_p,
                         uint64_t seed, int dtype, void *stream) {
    return attention_fwd_launch(qkv, out, lse, B, N, h, dh, scale, dropout_p, seed, dtype, stream, nullptr, nullptr, nullptr);
}

int ecgvit_attention_fwd_q8(const void *qkv, void *out, float *lse, int B, int N, int h, int dh, float scale, float dropout_p,
                            uint64_t seed, void *out8, const float *q8_scale, float *q8_amax, void *stream) {
    if (!out8) return ECGVIT_EINVAL;
    return attention_fwd_launch(qkv, out, lse, B, N, h, dh, scale, dropout_p, seed, ECGVIT_BF16, stream, out8, q8_scale, q8_amax);
}

static int attention_bwd_args_ok(const void *qkv, const void *out, const void *dout, void *dqkv, int B, int N, int h, int dh, int dtype) {
    if (dtype != ECGVIT_BF16 || dh != 64 || N < 1 || N > 512 || B < 1 || h < 1) return 0;
    return (reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dqkv)) % 16 == 0;
}

// one (record, head) item per workgroup, all keys of the item on its waves: N <= 256 only.  The shipped backward for short
// sequences (N <= 128) and the independent implementation the tests hold the persistent kernel against (exported by the tools library).
static int attention_bwd_oneitem(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h,
                                 int dh, float scale, float dropout_p, uint64_t seed, int dtype, void *stream) {
    if (!attention_bwd_args_ok(qkv, out, dout, dqkv, B, N, h, dh, dtype) || N > 256) return ECGVIT_EINVAL;
    if (dropout_p > 0.f && dropout_threshold8(dropout_p) == 0) return ECGVIT_EINVAL;   // p < 1/512 would silently round to no dropout
    const uint32_t th = dropout_threshold8(dropout_p);
    const float ik = dropout_inv_keep8(dropout_p);
    dim3 grid((unsigned)(B * h));
#define BWD(NKT, DR) hipLaunchKernelGGL((attn_bwd_bf16_kernel<NKT, DR>), grid, dim3(NKT * 64), 0, as_stream(stream), (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, N, h, scale, seed, th, ik, 0)
    if (N <= 128) { if (th) BWD(4, true); else BWD(4, false); }
    else { if (th) BWD(8, true); else BWD(8, false); }
#undef BWD
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}

static int attention_bwd_launch(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h,
                                int dh, float scale, float dropout_p, uint64_t seed, int dtype, void *stream, void *dqkv8, const float *q8_scale,
                                float *q8_amax) {
    if (!attention_bwd_args_ok(qkv, out, dout, dqkv, B, N, h, dh, dtype)) return ECGVIT_EINVAL;
    if (N <= 128 || (int64_t)N * 3 * h * 64 * 2 >= (1ll << 31)) {   // short sequences / 32-bit buffer offsets exhausted
        if (dqkv8) return ECGVIT_EINVAL;   // the one-item kernel has no 8-bit emission: the caller quantises dqkv itself
        return attention_bwd_oneitem(qkv, out, dout, lse, dqkv, B, N, h, dh, scale, dropout_p, seed, dtype, stream);
    }
    if (dqkv8 && (!q8_scale || !q8_amax || reinterpret_cast<uintptr_t>(dqkv8) % 8)) return ECGVIT_EINVAL;
    if (dropout_p > 0.f && dropout_threshold8(dropout_p) == 0) return ECGVIT_EINVAL;   // p < 1/512 would silently round to no dropout
    const uint32_t th = dropout_threshold8(dropout_p);
    const float ik = dropout_inv_keep8(dropout_p);
    const int nitems = B * h;
    // three workgroups' worth of items per CU slot: the hardware dispatcher hands them out as CUs free up, so a launch that shares the
    // GPU with a collective's kernels is not left with late workgroups a full static share behind (one per CU measured the same alone)
    const dim3 pg((unsigned)(nitems < 768 ? nitems : 768));
#define PERS_ARGS(K0) pg, dim3(512), 0, as_stream(stream), (const bf16_t *)qkv, (const bf16_t *)out, (const bf16_t *)dout, lse, (bf16_t *)dqkv, N, h, scale, seed, th, ik, nitems, K0
#ifndef ATTN_BWD_STAGGER
#define ATTN_BWD_STAGGER true
#endif
#ifndef ATTN_BWD_PRIO
#define ATTN_BWD_PRIO 1
#endif
#define PERS(DR, AC, K0) hipLaunchKernelGGL((attn_bwd_pers_kernel<DR, AC, ATTN_BWD_STAGGER, ATTN_BWD_PRIO>), PERS_ARGS(K0))
// (the emitting variants run the staggered schedule as well since round 4's vector diet -- 239-255 VGPRs, no spills -- except the second key
// window under dropout with all three conversions, which would spill 4 registers: scratch traffic would join the counted vmcnt waits, so that
// one keeps the lockstep schedule; round 6 took ten loop-invariant registers out of the kernel -- the K-image offsets recomputed per item, the dQ offset
// tables replaced by the wave's one offset -- and it still spilled those four (the peak is inside the block's vector phase, not in what lives across it),
// while the other instantiations got 1.6-2 % SLOWER with the different allocation: taken out again)
#define PERS8(DR, AC, K0, Q) hipLaunchKernelGGL((attn_bwd_pers_kernel<DR, AC, ATTN_BWD_STAGGER && !(DR && AC && Q == 3), ATTN_BWD_PRIO, Q>), PERS_ARGS(K0), (uint8_t *)dqkv8, q8_scale, q8_amax)
#ifdef ECGVIT_TOOLS
    if (g_tools_attn_variant >= 0 && th && N <= 256 && !dqkv8) {   // tools build: A/B of the stagger / priority variants (tools/attn_variants.py)
        switch (g_tools_attn_variant) {
            case 0: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, false, 1>), PERS_ARGS(0)); break;   // lockstep
            case 1: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, false, 0>), PERS_ARGS(0)); break;
            case 2: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, true, 1>), PERS_ARGS(0)); break;    // what ships
            case 3: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, true, 0>), PERS_ARGS(0)); break;
            case 4: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, true, 2>), PERS_ARGS(0)); break;
            default: hipLaunchKernelGGL((attn_bwd_pers_kernel<true, false, false, 2>), PERS_ARGS(0)); break;
        }
        ECGVIT_CHECK_LAUNCH();
        return ECGVIT_OK;
    }
#endif
    if (dqkv8) {
        // one window: dK / dV / dQ all final in this launch; two windows: the first emits its dK / dV, the second its dK / dV and the final dQ
        if (N <= 256) { if (th) PERS8(true, false, 0, 3); else PERS8(false, false, 0, 3); }
        else { if (th) PERS8(true, false, 0, 1); else PERS8(false, false, 0, 1); }
        ECGVIT_CHECK_LAUNCH();
        if (N > 256) {
            if (th) PERS8(true, true, 256, 3); else PERS8(false, true, 256, 3);
            ECGVIT_CHECK_LAUNCH();
        }
        return ECGVIT_OK;
    }
    if (th) PERS(true, false, 0); else PERS(false, false, 0);
    ECGVIT_CHECK_LAUNCH();
    if (N > 256) {   // second window of keys; its dQ accumulates on the first launch's (stream order)
        if (th) PERS(true, true, 256); else PERS(false, true, 256);
        ECGVIT_CHECK_LAUNCH();
    }
#undef PERS
#undef PERS8
#undef PERS_ARGS
    return ECGVIT_OK;
}

int ecgvit_attention_bwd(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h,
                         int dh, float scale, float dropout_p, uint64_t seed, int dtype, void *stream) {
    return attention_bwd_launch(qkv, out, dout, lse, dqkv, B, N, h, dh, scale, dropout_p, seed, dtype, stream, nullptr, nullptr, nullptr);
}

int ecgvit_attention_bwd_q8(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h,
                            int dh, float scale, float dropout_p, uint64_t seed, void *dqkv8, const float *q8_scale, float *q8_amax, void *stream) {
    if (!dqkv8) return ECGVIT_EINVAL;
    return attention_bwd_launch(qkv, out, dout, lse, dqkv, B, N, h, dh, scale, dropout_p, seed, ECGVIT_BF16, stream, dqkv8, q8_scale, q8_amax);
}

#ifdef ECGVIT_TOOLS   // declared in tools/ecgvit_hip_tools.h, exported by build/libecgvit_hip_tools.so only
int ecgvit_probe_mfma_layout(float *out, void *stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), out);
    ECGVIT_CHECK_LAUNCH();
    return ECGVIT_OK;
}
// the one-item-per-workgroup backward as an entry point of its own (N <= 256): the independent implementation tests hold the persistent kernels against
int ecgvit_attention_bwd_oneitem(const void *qkv, const void *out, const void *dout, const float *lse, void *dqkv, int B, int N, int h,
                                 int dh, float scale, float dropout_p, uint64_t seed, int dtype, void *stream) {
    return attention_bwd_oneitem(qkv, out, dout, lse, dqkv, B, N, h, dh, scale, dropout_p, seed, dtype, stream);
}
#endif

}  // extern "C"
