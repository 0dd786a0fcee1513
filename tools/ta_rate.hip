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
//   mode 16 + bits: the GEMM K-tile's instruction mix without its dependencies, per wave and sweep: bit 0 = 8 LDS-DMA pieces,
//           bit 1 = 24 ds_read_b128, bit 2 = 64 v_mfma_f32_16x16x32_bf16
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
                    for (int k = 0; k < 6; ++k) f[k] = __builtin_bit_cast(bf16x8, acc);
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
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE, int THREADS>
static void run(const char *name, char *buf, unsigned long long *dcyc, unsigned *sink, int nwg) {
    const int iters = 2000;
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
        run<17, 512>("mix: DMA", buf, dcyc, sink, nwg);
        run<18, 512>("mix: ds_read", buf, dcyc, sink, nwg);
        run<20, 512>("mix: MFMA", buf, dcyc, sink, nwg);
        run<19, 512>("mix: DMA + ds_read", buf, dcyc, sink, nwg);
        run<21, 512>("mix: DMA + MFMA", buf, dcyc, sink, nwg);
        run<22, 512>("mix: ds_read + MFMA", buf, dcyc, sink, nwg);
        run<23, 512>("mix: DMA + ds_read + MFMA", buf, dcyc, sink, nwg);
    }
    return 0;
}
