// Per-CU rate of the vector-memory path for L2-resident data, one 512-thread (or 256-thread) workgroup per CU:
//   mode 0: LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), what gemm_nt_kernel stages its operands with
//   mode 1: buffer_load_dwordx4 into VGPRs (register staging), same addresses
//   mode 2: mode 1 + ds_write_b128 of every loaded register quad (the full register-staged fill)
//   mode 3: buffer_store_dwordx4, a wave-instruction covering 16 rows x 64 B (gemm_nt_kernel's epilogue shape)
//   mode 4: buffer_store_dwordx4, a wave-instruction covering 8 rows x 128 B (whole lines)
//   mode 5: buffer_store_dwordx4, 1 KiB contiguous per wave-instruction
//   mode 6: buffer_store_dwordx4, 8 rows x 128 B with the lanes of a line 8 apart (lane = 16 q + 8 half + row: an MFMA accumulator
//           layout after a row_ror:8 exchange of the two half-line runs)
//   mode 7 / 8: buffer_load_dwordx4 -> VGPR, 16 rows x 64 B / 8 rows x 128 B per wave-instruction (epilogue operand reads)
//   mode 32 / 33: bare v_mfma_f32_16x16x32_bf16 / 32x32x16 loops on random operands (same MACs): FLOP/s at the power cap
//   mode 16 + bits: the GEMM K-tile's instruction mix without its dependencies, per wave and sweep: bit 0 = 8 LDS-DMA pieces,
//           bit 1 = 24 ds_read_b128, bit 2 = 64 v_mfma_f32_16x16x32_bf16, bit 3 = one s_barrier per sweep; LDS and register operands random
// Every workgroup works on its own 64-KiB window (256 windows = 16 MiB: L2-resident on 8 x 4 MiB), `iters` sweeps of 64 wave-instructions.
// Build + run (GPU box):  hipcc -O3 -w --offload-arch=gfx950 tools/ta_rate.hip -o gpurun_out/ta_rate && gpurun_out/ta_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void *lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void ta_rate_kernel(char *buf, int iters, unsigned long long *cycles, unsigned *sink) {
    __shared__ __attribute__((aligned(1024))) char smem[65536];
    constexpr int NW = THREADS / 64;
    constexpr int PER_WAVE = 64 / NW;   // wave-instructions per sweep and wave: 64 KiB per workgroup and sweep
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *win = buf + (size_t)blockIdx.x * 65536;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(win, 0, 65536u, 0x00020000);
    // per-lane byte offset inside a 1-KiB-per-instruction piece; rows are 128 B apart in the window (K-contiguous operand rows of a 64-deep bf16 K-tile)
    int vo;
    if (MODE == 3) vo = (lane & 15) * 128 + (lane >> 4) * 16;               // 16 rows x 64 B (second half-line by the next instruction)
    else if (MODE == 4 || MODE == 8) vo = (lane >> 3) * 128 + (lane & 7) * 16;   // 8 rows x 128 B
    else if (MODE == 6) vo = (lane & 7) * 128 + ((lane >> 3) & 1) * 64 + (lane >> 4) * 16;
    else if (MODE == 7) vo = (lane & 15) * 128 + (lane >> 4) * 16;
    else if (MODE == 9 || MODE == 10) vo = (lane >> 2) * 128 + (lane & 3) * 16;   // 16 rows x 64 B, the four lanes of a half-line adjacent
    else vo = lane * 16;                                                   // 1 KiB contiguous (= 8 rows x 128 B of a dense image)
    u32x4 acc = {0u, 0u, 0u, 0u};
    [[maybe_unused]] f32x4 macc[8] = {};
    [[maybe_unused]] f32x16 macc16[2] = {};
    [[maybe_unused]] f32x4 bacc[(MODE >= 50 && MODE < 60) ? 8 : 1][4] = {};
    [[maybe_unused]] f32x4 wacc[(MODE == 60 || MODE == 61) ? 8 : 1][MODE == 61 ? 8 : 4] = {};
    [[maybe_unused]] bf16x8 rnd[4];
    {   // pseudo-random bf16 operands in [-2, 2): exponent bits from a small set, random mantissas
        unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u + 12345u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h ^= h << 13; h ^= h >> 17; h ^= h << 5;
                w[k] = (h & 0x807F807Fu) | 0x3F803F80u;   // sign + 7 mantissa bits random, exponent of 1.0
            }
            rnd[q] = __builtin_bit_cast(bf16x8, w);
        }
    }
    if (MODE >= 16) {   // random bf16 contents for the fragment reads (the matrix pipe's power depends on its operands)
        unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ 0x9E3779B9u;
        for (int i = threadIdx.x; i < 16384; i += THREADS) {
            h ^= h << 13; h ^= h >> 17; h ^= h << 5;
            reinterpret_cast<unsigned *>(smem)[i] = (h & 0x807F807Fu) | 0x3F803F80u;
        }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                const int piece = wave * PER_WAVE + p;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + piece * 1024), 16, vo, piece * 1024, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE > 8 ? 8 : PER_WAVE / 2) : "memory");
        } else if constexpr (MODE == 1 || MODE == 2) {
            u32x4 r[PER_WAVE];
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) r[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, ((wave * PER_WAVE + p + it) & 63) * 1024, 0);
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                if constexpr (MODE == 2) *reinterpret_cast<u32x4 *>(smem + (wave * PER_WAVE + p) * 1024 + lane * 16) = r[p];
                else acc ^= r[p];
            }
        } else if constexpr (MODE == 11) {
            // 64 ds_bpermute_b32 per wave and sweep (what a transposition of 64 accumulator dwords per lane costs on the LDS crossbar)
            const int src = ((lane >> 3) + 8 * ((lane >> 2) & 1) + 16 * (lane & 3)) * 4;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(acc[k] + p));
            }
        } else if constexpr (MODE == 12) {
            // full-line stores fed by 4 bpermutes + 4 packs + 12 DPP-class moves per pair, as an accumulator-layout epilogue would
            const int src = ((lane >> 3) + 8 * ((lane >> 2) & 1) + 16 * (lane & 3)) * 4;
            u32x4 v = {(unsigned)it, (unsigned)lane, 3u, 4u};
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v[k], (int)v[(k + 1) & 3], 0x128, 0xF, 0xC, false);   // row_ror:8, banks 2,3
                    t += (unsigned)__builtin_amdgcn_update_dpp((int)t, (int)v[(k + 2) & 3], 0x128, 0xF, 0x3, false);
                    o[k] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)t);
                }
                v = o;
                __builtin_amdgcn_raw_buffer_store_b128(o, rs, (lane >> 3) * 128 + (lane & 7) * 16, (wave * PER_WAVE + p) * 1024, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE > 8 ? 8 : PER_WAVE / 2) : "memory");
        } else if constexpr (MODE == 7 || MODE == 8 || MODE == 10) {
            u32x4 r[PER_WAVE];
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                const int piece = (wave * PER_WAVE + p + it) & 63;
                r[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, MODE != 8 ? (piece >> 1) * 2048 + (piece & 1) * 64 : piece * 1024, 0);
            }
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) acc ^= r[p];
        } else if constexpr (MODE == 32 || MODE == 33) {
            // bare MFMA loops on random register operands (the power the matrix pipe draws depends on the data): 64 x 16x16x32 or
            // 32 x 32x32x16 per wave and sweep = the same 2^20 MACs
            if constexpr (MODE == 32) {
#pragma unroll
                for (int k = 0; k < 64; ++k) macc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rnd[k & 3], rnd[(k + 1) & 3], macc[k & 7], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) macc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rnd[k & 3], rnd[(k + 1) & 3], macc16[k & 1], 0, 0, 0);
            }
        } else if constexpr (MODE == 60 || MODE == 61) {
            // one K-tile of a 256 x 256 x 64 block tile per sweep, with its operand pieces (64 KiB by LDS-DMA from the L2-resident window),
            // compiler-scheduled, random operands:  60 = 8 waves x (128 x 64) wave tiles (192 KiB of fragment reads per K-tile),
            //                                        61 = 4 waves x (128 x 128) wave tiles (128 KiB), 256 accumulator registers per lane
            constexpr int NJ = MODE == 60 ? 4 : 8, NP = MODE == 60 ? 8 : 16;
            const int fr = lane & 15, fq = lane >> 4;
            const int base = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int piece = (wave * NP + p + it) & 63;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (wave * NP + p) * 1024), 16, vo, piece * 1024, 0, 0);
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                bf16x8 a[8], b[NJ];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(smem + (((wave & 1) * 16384 + i * 2048 + base) ^ (ss * 64)));
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(smem + ((32768 + ((wave >> 1) & 1) * 16384 + j * 2048 + base) ^ (ss * 64)));
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], wacc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP / 2) : "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (MODE >= 50 && MODE <= 53) {
            // gemm_nt_kernel's K-tile without its DMA: 128 accumulators, quadrants of 16 MFMAs, fragment reads 12 / 4 / 8 / 0.
            //   50: compiler-scheduled   51: + explicit lgkmcnt(0), sched barriers and s_setprio around each MFMA cluster
            //   52: 51 + one barrier per K-tile, waves 4-7 passing it before quadrant 4 (the shipped arrangement)   53: 52 without s_setprio
            constexpr bool EXPL = MODE >= 51, BAR = MODE >= 52, PRIO = MODE == 51 || MODE == 52;
            const int fr = lane & 15, fq = lane >> 4;
            const int base = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
            const int sa = (wave >> 2) * 16384 + base, sb = 32768 + ((wave & 3) >> 1) * 16384 + ((wave & 1) * 8192) + base;
            bf16x8 a[4][2], b0[2][2], b1[2][2];
            const bool latew = wave >= 4;
#define T_SYNC_A() do { if (EXPL) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); if (PRIO) __builtin_amdgcn_s_setprio(1); } } while (0)
#define T_SYNC_B() do { if (EXPL) { if (PRIO) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define T_QUAD(IO, JO, BF) do { _Pragma("unroll") for (int ss = 0; ss < 2; ++ss) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
                bacc[IO + i][JO + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ss], a[i][ss], bacc[IO + i][JO + j], 0, 0, 0); } while (0)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) b0[j][ss] = *reinterpret_cast<const bf16x8 *>(smem + ((sb + j * 512) ^ (ss * 64)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) a[i][ss] = *reinterpret_cast<const bf16x8 *>(smem + ((sa + i * 2048) ^ (ss * 64)));
            T_SYNC_A(); T_QUAD(0, 0, b0); T_SYNC_B();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) b1[j][ss] = *reinterpret_cast<const bf16x8 *>(smem + ((sb + 4096 + j * 512) ^ (ss * 64)));
            T_SYNC_A(); T_QUAD(0, 2, b1); T_SYNC_B();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) a[i][ss] = *reinterpret_cast<const bf16x8 *>(smem + ((sa + (4 + i) * 2048) ^ (ss * 64)));
            T_SYNC_A(); T_QUAD(4, 2, b1); T_SYNC_B();
            if (BAR && latew) __builtin_amdgcn_s_barrier();
            if (EXPL) { __builtin_amdgcn_sched_barrier(0); if (PRIO) __builtin_amdgcn_s_setprio(1); }
            T_QUAD(4, 0, b0); T_SYNC_B();
            if (BAR && !latew) __builtin_amdgcn_s_barrier();
#undef T_SYNC_A
#undef T_SYNC_B
#undef T_QUAD
        } else if constexpr (MODE == 40 || MODE == 41) {
            // the GEMM's issue pattern: two pieces per quadrant; the sweep's counted wait covers the pieces of ITS first two quadrants
            // (mode 40: what a one-K-tile-ahead operand gets) or only pieces of the sweep before (mode 41: two K-tiles ahead)
            bf16x8 f[6];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int fr = lane & 15, fq = lane >> 4;
                    f[k] = *reinterpret_cast<const bf16x8 *>(smem + ((wave * 4 + g) & 31) * 2048 + fr * 128 + (((fq + 2 * k) & 7) ^ ((fr >> 1) & 7)) * 16);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int piece = (wave * 8 + g * 2 + p + it) & 63;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (wave * 8 + g * 2 + p) * 1024), 16, vo, piece * 1024, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) macc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[k % 6], f[(k + 1) % 6], macc[k & 7], 0, 0, 0);
            }
            if constexpr (MODE == 40) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if constexpr (MODE >= 16) {
            static_assert(MODE < 16 || THREADS == 512, "mix modes: 8 waves");
            if constexpr (MODE & 1) {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int piece = (wave * 8 + p + it) & 63;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + (wave * 8 + p) * 1024), 16, vo, piece * 1024, 0, 0);
                }
            }
            bf16x8 f[6];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (MODE & 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int fr = lane & 15, fq = lane >> 4;
                        f[k] = *reinterpret_cast<const bf16x8 *>(smem + ((wave * 4 + g) & 31) * 2048 + fr * 128 + (((fq + 2 * k) & 7) ^ ((fr >> 1) & 7)) * 16);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) f[k] = rnd[k & 3];
                }
                if constexpr (MODE & 4) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) macc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[k % 6], f[(k + 1) % 6], macc[k & 7], 0, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc ^= __builtin_bit_cast(u32x4, f[k]);
                }
            }
            if constexpr (MODE & 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if constexpr ((MODE & 8) != 0) __builtin_amdgcn_s_barrier();   // + the K-tile's workgroup barrier
        } else {
            const u32x4 v = {(unsigned)it, (unsigned)lane, 3u, 4u};
#pragma unroll
            for (int p = 0; p < PER_WAVE; ++p) {
                const int piece = wave * PER_WAVE + p;
                // MODE 3: pieces 2q and 2q+1 are the two half-lines of the same 16 rows
                const int so = (MODE == 3 || MODE == 9) ? (piece >> 1) * 2048 + (piece & 1) * 64 : piece * 1024;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, vo, so, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE > 8 ? 8 : PER_WAVE / 2) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (MODE == 2) acc[0] ^= *reinterpret_cast<unsigned *>(smem + threadIdx.x * 4);
    if (MODE >= 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[0] ^= __builtin_bit_cast(u32x4, macc[k])[k & 3];
        acc[1] ^= __builtin_bit_cast(unsigned, macc16[0][3]) ^ __builtin_bit_cast(unsigned, macc16[1][5]);
        if constexpr (MODE >= 50 && MODE < 60) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[(i + j) & 3] ^= __builtin_bit_cast(u32x4, bacc[i][j])[(i * j) & 3];
        }
        if constexpr (MODE == 60 || MODE == 61) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < (MODE == 61 ? 8 : 4); ++j) acc[(i + j) & 3] ^= __builtin_bit_cast(u32x4, wacc[i][j])[(i * j) & 3];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE, int THREADS>
static void run(const char *name, char *buf, unsigned long long *dcyc, unsigned *sink, int nwg) {
    const int iters = MODE >= 32 ? 200000 : (MODE == 20 || MODE == 22 || MODE == 23 || MODE == 21 || MODE == 31 || MODE == 28 || MODE == 40 || MODE == 41) ? 100000 : 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    ta_rate_kernel<MODE, THREADS><<<nwg, THREADS>>>(buf, 50, dcyc, sink);
    hipEventRecord(e0);
    ta_rate_kernel<MODE, THREADS><<<nwg, THREADS>>>(buf, iters, dcyc, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(nwg);
    hipMemcpy(c.data(), dcyc, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double med = (double)c[nwg / 2];
    // s_memtime ticks at 100 MHz on this part; wall time gives cycles through the measured clock instead
    const double us = ms * 1e3;
    const double bytes = (double)iters * 65536.0;
    printf("%-44s threads %3d wgs %3d: %8.1f us  %6.1f GB/s per CU  %7.2f TB/s chip   %6.1f ns per wave-instruction per CU (memtime ticks med %.0f)\n", name, THREADS, nwg, us,
           bytes / us * 1e-3, bytes * nwg / us * 1e-6, us * 1e3 / (iters * 64.0), med);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    char *buf; unsigned long long *dcyc; unsigned *sink;
    hipMalloc(&buf, 256u * 65536u);
    hipMemset(buf, 1, 256u * 65536u);
    hipMalloc(&dcyc, 256 * sizeof(unsigned long long));
    hipMalloc(&sink, 4);
    for (int nwg : {256, 32}) {
        printf("---- %d workgroups\n", nwg);
        run<0, 512>("LDS-DMA pieces", buf, dcyc, sink, nwg);
        run<0, 256>("LDS-DMA pieces", buf, dcyc, sink, nwg);
        run<1, 512>("buffer_load_dwordx4 -> VGPR", buf, dcyc, sink, nwg);
        run<1, 256>("buffer_load_dwordx4 -> VGPR", buf, dcyc, sink, nwg);
        run<2, 512>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", buf, dcyc, sink, nwg);
        run<2, 256>("buffer_load_dwordx4 -> VGPR -> ds_write_b128", buf, dcyc, sink, nwg);
        run<3, 512>("store dwordx4, 16 rows x 64 B", buf, dcyc, sink, nwg);
        run<4, 512>("store dwordx4, 8 rows x 128 B", buf, dcyc, sink, nwg);
        run<5, 512>("store dwordx4, 1 KiB contiguous", buf, dcyc, sink, nwg);
        run<5, 256>("store dwordx4, 1 KiB contiguous", buf, dcyc, sink, nwg);
        run<6, 512>("store dwordx4, 8 x 128 B, line lanes 8 apart", buf, dcyc, sink, nwg);
        run<7, 512>("load dwordx4 -> VGPR, 16 rows x 64 B", buf, dcyc, sink, nwg);
        run<8, 512>("load dwordx4 -> VGPR, 8 rows x 128 B", buf, dcyc, sink, nwg);
        run<9, 512>("store dwordx4, 16 x 64 B, quads adjacent", buf, dcyc, sink, nwg);
        run<10, 512>("load dwordx4, 16 x 64 B, quads adjacent", buf, dcyc, sink, nwg);
        run<11, 512>("64 ds_bpermute_b32 per wave and sweep", buf, dcyc, sink, nwg);
        run<12, 512>("store 8 x 128 B after bpermute+DPP transposition", buf, dcyc, sink, nwg);
        run<32, 512>("bare MFMA 16x16x32 bf16, random operands (2^20 MACs per wave and sweep)", buf, dcyc, sink, nwg);
        run<33, 512>("bare MFMA 32x32x16 bf16, random operands (2^20 MACs per wave and sweep)", buf, dcyc, sink, nwg);
        run<32, 256>("bare MFMA 16x16x32 bf16, random operands, 4 waves", buf, dcyc, sink, nwg);
        run<33, 256>("bare MFMA 32x32x16 bf16, random operands, 4 waves", buf, dcyc, sink, nwg);
        run<17, 512>("mix: DMA", buf, dcyc, sink, nwg);
        run<18, 512>("mix: ds_read", buf, dcyc, sink, nwg);
        run<20, 512>("mix: MFMA", buf, dcyc, sink, nwg);
        run<19, 512>("mix: DMA + ds_read", buf, dcyc, sink, nwg);
        run<21, 512>("mix: DMA + MFMA", buf, dcyc, sink, nwg);
        run<22, 512>("mix: ds_read + MFMA", buf, dcyc, sink, nwg);
        run<23, 512>("mix: DMA + ds_read + MFMA", buf, dcyc, sink, nwg);
        run<31, 512>("mix: DMA + ds_read + MFMA + one s_barrier per sweep", buf, dcyc, sink, nwg);
        run<28, 512>("mix: MFMA + one s_barrier per sweep", buf, dcyc, sink, nwg);
        run<60, 512>("256x256x64 K-tile, 8 waves x 128x64, DMA + reads + MFMA + barrier", buf, dcyc, sink, nwg);
        run<61, 256>("256x256x64 K-tile, 4 waves x 128x128, DMA + reads + MFMA + barrier", buf, dcyc, sink, nwg);
        run<50, 512>("K-tile skeleton (128 accumulators, 12/4/8/0 reads), compiler-scheduled", buf, dcyc, sink, nwg);
        run<51, 512>("K-tile skeleton + explicit waits, sched barriers, s_setprio", buf, dcyc, sink, nwg);
        run<52, 512>("K-tile skeleton + those + one barrier per K-tile (waves 4-7 before quadrant 4)", buf, dcyc, sink, nwg);
        run<53, 512>("K-tile skeleton as 52 without s_setprio", buf, dcyc, sink, nwg);
        run<40, 512>("mix: all, 2 pieces per quadrant, wait covers this sweep's first 4 pieces", buf, dcyc, sink, nwg);
        run<41, 512>("mix: all, 2 pieces per quadrant, wait covers the previous sweep's pieces", buf, dcyc, sink, nwg);
    }
    return 0;
}
