// Issue cost of single vector instructions on gfx950, as one wave's stream sees it: alone on its SIMD (256-thread workgroup, one
// per CU) or with a partner wave running the same stream (512 threads), by itself or placed in the gaps between back-to-back
// v_mfma_f32_32x32x16_bf16 (4 or 8 per gap).  The attention and GELU-epilogue bodies are VALU-issue-bound: their instruction
// budgets are priced with this table (profiles/r04_valu_rate.txt).
// Every test is 64 copies of one instruction on 8 independent registers (a dependent use comes 8 instructions later).
// Build + run (GPU box):  hipcc -O3 -w --offload-arch=gfx950 tools/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define MFMA "v_mfma_f32_32x32x16_bf16 %[acc], %[a], %[b], %[acc]\n"
#define OPS                                                                                                                     \
    [r0] "+v"(r[0]), [r1] "+v"(r[1]), [r2] "+v"(r[2]), [r3] "+v"(r[3]), [r4] "+v"(r[4]), [r5] "+v"(r[5]), [r6] "+v"(r[6]),       \
        [r7] "+v"(r[7]), [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [acc] "+v"(acc)                      \
        : [x] "v"(x), [y] "v"(y), [a] "v"(a), [b] "v"(b), [s] "s"(sx), [px] "v"(px), [m] "s"(m64), [la] "v"(la)                                  \
        : "vcc", "memory"

// S4A / S4B: four instructions each on registers r0-r3 / r4-r7 (p0-p3: 64-bit register pairs for packed f32)
#define TEST(ID, NAME, S4A, S4B)                                                                                                \
    struct T##ID {                                                                                                              \
        static constexpr const char *name = NAME;                                                                               \
        template <int variant> static __device__ __forceinline__ void run( uint32_t (&r)[8], u32x2 (&p)[4], uint32_t x, uint32_t y, f32x16 &acc, \
                                                   bf16x8 a, bf16x8 b, uint32_t sx, u32x2 px, unsigned long long m64, uint32_t la) {                                 \
            if constexpr (variant == 0) asm volatile(".rept 8\n" S4A S4B ".endr\ns_waitcnt lgkmcnt(0)\n" : OPS);                                                \
            else if constexpr (variant == 1) asm volatile(".rept 8\n" MFMA S4A MFMA S4B ".endr\ns_waitcnt lgkmcnt(0)\n" : OPS);                                 \
            else if constexpr (variant == 2) asm volatile(".rept 8\n" MFMA S4A S4B ".endr\ns_waitcnt lgkmcnt(0)\n" : OPS);                                      \
            else asm volatile(".rept 16\n" MFMA ".endr\ns_waitcnt lgkmcnt(0)\n" : OPS);                                                               \
        }                                                                                                                       \
    };

#define I4(FMT_A, FMT_B, FMT_C, FMT_D) FMT_A "\n" FMT_B "\n" FMT_C "\n" FMT_D "\n"

TEST(0, "v_fma_f32", I4("v_fma_f32 %[r0], %[r0], %[x], %[y]", "v_fma_f32 %[r1], %[r1], %[x], %[y]", "v_fma_f32 %[r2], %[r2], %[x], %[y]", "v_fma_f32 %[r3], %[r3], %[x], %[y]"),
     I4("v_fma_f32 %[r4], %[r4], %[x], %[y]", "v_fma_f32 %[r5], %[r5], %[x], %[y]", "v_fma_f32 %[r6], %[r6], %[x], %[y]", "v_fma_f32 %[r7], %[r7], %[x], %[y]"))
TEST(1, "v_mul_f32", I4("v_mul_f32 %[r0], %[r0], %[x]", "v_mul_f32 %[r1], %[r1], %[x]", "v_mul_f32 %[r2], %[r2], %[x]", "v_mul_f32 %[r3], %[r3], %[x]"),
     I4("v_mul_f32 %[r4], %[r4], %[x]", "v_mul_f32 %[r5], %[r5], %[x]", "v_mul_f32 %[r6], %[r6], %[x]", "v_mul_f32 %[r7], %[r7], %[x]"))
TEST(2, "v_exp_f32", I4("v_exp_f32 %[r0], %[r0]", "v_exp_f32 %[r1], %[r1]", "v_exp_f32 %[r2], %[r2]", "v_exp_f32 %[r3], %[r3]"),
     I4("v_exp_f32 %[r4], %[r4]", "v_exp_f32 %[r5], %[r5]", "v_exp_f32 %[r6], %[r6]", "v_exp_f32 %[r7], %[r7]"))
TEST(3, "v_rcp_f32", I4("v_rcp_f32 %[r0], %[r0]", "v_rcp_f32 %[r1], %[r1]", "v_rcp_f32 %[r2], %[r2]", "v_rcp_f32 %[r3], %[r3]"),
     I4("v_rcp_f32 %[r4], %[r4]", "v_rcp_f32 %[r5], %[r5]", "v_rcp_f32 %[r6], %[r6]", "v_rcp_f32 %[r7], %[r7]"))
TEST(4, "v_pk_mul_f32", I4("v_pk_mul_f32 %[p0], %[p0], %[px]", "v_pk_mul_f32 %[p1], %[p1], %[px]", "v_pk_mul_f32 %[p2], %[p2], %[px]", "v_pk_mul_f32 %[p3], %[p3], %[px]"),
     I4("v_pk_mul_f32 %[p0], %[p0], %[px]", "v_pk_mul_f32 %[p1], %[p1], %[px]", "v_pk_mul_f32 %[p2], %[p2], %[px]", "v_pk_mul_f32 %[p3], %[p3], %[px]"))
TEST(5, "v_pk_fma_f32", I4("v_pk_fma_f32 %[p0], %[p0], %[px], %[px]", "v_pk_fma_f32 %[p1], %[p1], %[px], %[px]", "v_pk_fma_f32 %[p2], %[p2], %[px], %[px]", "v_pk_fma_f32 %[p3], %[p3], %[px], %[px]"),
     I4("v_pk_fma_f32 %[p0], %[p0], %[px], %[px]", "v_pk_fma_f32 %[p1], %[p1], %[px], %[px]", "v_pk_fma_f32 %[p2], %[p2], %[px], %[px]", "v_pk_fma_f32 %[p3], %[p3], %[px], %[px]"))
TEST(6, "v_cmp_ge_u32 vcc", I4("v_cmp_ge_u32 vcc, %[r0], %[x]", "v_cmp_ge_u32 vcc, %[r1], %[x]", "v_cmp_ge_u32 vcc, %[r2], %[x]", "v_cmp_ge_u32 vcc, %[r3], %[x]"),
     I4("v_cmp_ge_u32 vcc, %[r4], %[x]", "v_cmp_ge_u32 vcc, %[r5], %[x]", "v_cmp_ge_u32 vcc, %[r6], %[x]", "v_cmp_ge_u32 vcc, %[r7], %[x]"))
TEST(7, "v_cmp_ge_u32_sdwa BYTE_n", I4("v_cmp_ge_u32_sdwa vcc, %[r0], %[x] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ge_u32_sdwa vcc, %[r1], %[x] src0_sel:BYTE_1 src1_sel:DWORD",
                                        "v_cmp_ge_u32_sdwa vcc, %[r2], %[x] src0_sel:BYTE_2 src1_sel:DWORD", "v_cmp_ge_u32_sdwa vcc, %[r3], %[x] src0_sel:BYTE_3 src1_sel:DWORD"),
     I4("v_cmp_ge_u32_sdwa vcc, %[r4], %[x] src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_ge_u32_sdwa vcc, %[r5], %[x] src0_sel:BYTE_1 src1_sel:DWORD",
        "v_cmp_ge_u32_sdwa vcc, %[r6], %[x] src0_sel:BYTE_2 src1_sel:DWORD", "v_cmp_ge_u32_sdwa vcc, %[r7], %[x] src0_sel:BYTE_3 src1_sel:DWORD"))
TEST(8, "v_cmp + v_cndmask pair (vcc)", I4("v_cmp_ge_u32 vcc, %[r0], %[x]", "v_cndmask_b32 %[r1], %[r1], %[y], vcc", "v_cmp_ge_u32 vcc, %[r2], %[x]", "v_cndmask_b32 %[r3], %[r3], %[y], vcc"),
     I4("v_cmp_ge_u32 vcc, %[r4], %[x]", "v_cndmask_b32 %[r5], %[r5], %[y], vcc", "v_cmp_ge_u32 vcc, %[r6], %[x]", "v_cndmask_b32 %[r7], %[r7], %[y], vcc"))
TEST(9, "v_cndmask_b32 (sgpr pair)", I4("v_cndmask_b32 %[r0], %[r0], %[y], %[m]", "v_cndmask_b32 %[r1], %[r1], %[y], %[m]", "v_cndmask_b32 %[r2], %[r2], %[y], %[m]", "v_cndmask_b32 %[r3], %[r3], %[y], %[m]"),
     I4("v_cndmask_b32 %[r4], %[r4], %[y], %[m]", "v_cndmask_b32 %[r5], %[r5], %[y], %[m]", "v_cndmask_b32 %[r6], %[r6], %[y], %[m]", "v_cndmask_b32 %[r7], %[r7], %[y], %[m]"))
TEST(10, "v_bfe_u32", I4("v_bfe_u32 %[r0], %[r0], %[x], 8", "v_bfe_u32 %[r1], %[r1], %[x], 8", "v_bfe_u32 %[r2], %[r2], %[x], 8", "v_bfe_u32 %[r3], %[r3], %[x], 8"),
     I4("v_bfe_u32 %[r4], %[r4], %[x], 8", "v_bfe_u32 %[r5], %[r5], %[x], 8", "v_bfe_u32 %[r6], %[r6], %[x], 8", "v_bfe_u32 %[r7], %[r7], %[x], 8"))
TEST(11, "v_mul_lo_u32", I4("v_mul_lo_u32 %[r0], %[r0], %[x]", "v_mul_lo_u32 %[r1], %[r1], %[x]", "v_mul_lo_u32 %[r2], %[r2], %[x]", "v_mul_lo_u32 %[r3], %[r3], %[x]"),
     I4("v_mul_lo_u32 %[r4], %[r4], %[x]", "v_mul_lo_u32 %[r5], %[r5], %[x]", "v_mul_lo_u32 %[r6], %[r6], %[x]", "v_mul_lo_u32 %[r7], %[r7], %[x]"))
TEST(12, "v_mad_u32_u24", I4("v_mad_u32_u24 %[r0], %[r0], %[x], %[y]", "v_mad_u32_u24 %[r1], %[r1], %[x], %[y]", "v_mad_u32_u24 %[r2], %[r2], %[x], %[y]", "v_mad_u32_u24 %[r3], %[r3], %[x], %[y]"),
     I4("v_mad_u32_u24 %[r4], %[r4], %[x], %[y]", "v_mad_u32_u24 %[r5], %[r5], %[x], %[y]", "v_mad_u32_u24 %[r6], %[r6], %[x], %[y]", "v_mad_u32_u24 %[r7], %[r7], %[x], %[y]"))
TEST(13, "v_xor_b32", I4("v_xor_b32 %[r0], %[r0], %[x]", "v_xor_b32 %[r1], %[r1], %[x]", "v_xor_b32 %[r2], %[r2], %[x]", "v_xor_b32 %[r3], %[r3], %[x]"),
     I4("v_xor_b32 %[r4], %[r4], %[x]", "v_xor_b32 %[r5], %[r5], %[x]", "v_xor_b32 %[r6], %[r6], %[x]", "v_xor_b32 %[r7], %[r7], %[x]"))
TEST(14, "v_xad_u32", I4("v_xad_u32 %[r0], %[r0], %[x], %[y]", "v_xad_u32 %[r1], %[r1], %[x], %[y]", "v_xad_u32 %[r2], %[r2], %[x], %[y]", "v_xad_u32 %[r3], %[r3], %[x], %[y]"),
     I4("v_xad_u32 %[r4], %[r4], %[x], %[y]", "v_xad_u32 %[r5], %[r5], %[x], %[y]", "v_xad_u32 %[r6], %[r6], %[x], %[y]", "v_xad_u32 %[r7], %[r7], %[x], %[y]"))
TEST(15, "v_perm_b32", I4("v_perm_b32 %[r0], %[r0], %[x], %[y]", "v_perm_b32 %[r1], %[r1], %[x], %[y]", "v_perm_b32 %[r2], %[r2], %[x], %[y]", "v_perm_b32 %[r3], %[r3], %[x], %[y]"),
     I4("v_perm_b32 %[r4], %[r4], %[x], %[y]", "v_perm_b32 %[r5], %[r5], %[x], %[y]", "v_perm_b32 %[r6], %[r6], %[x], %[y]", "v_perm_b32 %[r7], %[r7], %[x], %[y]"))
TEST(16, "v_mov_b32_dpp quad_perm", I4("v_mov_b32_dpp %[r0], %[r1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %[r1], %[r2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf",
                                       "v_mov_b32_dpp %[r2], %[r3] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %[r3], %[r4] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"),
     I4("v_mov_b32_dpp %[r4], %[r5] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %[r5], %[r6] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf",
        "v_mov_b32_dpp %[r6], %[r7] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %[r7], %[r0] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"))
TEST(17, "v_cvt_pk_bf16_f32", I4("v_cvt_pk_bf16_f32 %[r0], %[r0], %[x]", "v_cvt_pk_bf16_f32 %[r1], %[r1], %[x]", "v_cvt_pk_bf16_f32 %[r2], %[r2], %[x]", "v_cvt_pk_bf16_f32 %[r3], %[r3], %[x]"),
     I4("v_cvt_pk_bf16_f32 %[r4], %[r4], %[x]", "v_cvt_pk_bf16_f32 %[r5], %[r5], %[x]", "v_cvt_pk_bf16_f32 %[r6], %[r6], %[x]", "v_cvt_pk_bf16_f32 %[r7], %[r7], %[x]"))
TEST(18, "v_and_b32_sdwa BYTE_n", I4("v_and_b32_sdwa %[r0], %[r0], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %[r1], %[r1], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD",
                                     "v_and_b32_sdwa %[r2], %[r2], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "v_and_b32_sdwa %[r3], %[r3], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD"),
     I4("v_and_b32_sdwa %[r4], %[r4], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %[r5], %[r5], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD",
        "v_and_b32_sdwa %[r6], %[r6], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD", "v_and_b32_sdwa %[r7], %[r7], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD"))
TEST(19, "v_pk_mul_f16", I4("v_pk_mul_f16 %[r0], %[r0], %[x]", "v_pk_mul_f16 %[r1], %[r1], %[x]", "v_pk_mul_f16 %[r2], %[r2], %[x]", "v_pk_mul_f16 %[r3], %[r3], %[x]"),
     I4("v_pk_mul_f16 %[r4], %[r4], %[x]", "v_pk_mul_f16 %[r5], %[r5], %[x]", "v_pk_mul_f16 %[r6], %[r6], %[x]", "v_pk_mul_f16 %[r7], %[r7], %[x]"))
TEST(20, "v_med3_f32", I4("v_med3_f32 %[r0], %[r0], %[x], %[y]", "v_med3_f32 %[r1], %[r1], %[x], %[y]", "v_med3_f32 %[r2], %[r2], %[x], %[y]", "v_med3_f32 %[r3], %[r3], %[x], %[y]"),
     I4("v_med3_f32 %[r4], %[r4], %[x], %[y]", "v_med3_f32 %[r5], %[r5], %[x], %[y]", "v_med3_f32 %[r6], %[r6], %[x], %[y]", "v_med3_f32 %[r7], %[r7], %[x], %[y]"))
TEST(21, "v_mul_u32_u24", I4("v_mul_u32_u24 %[r0], %[r0], %[x]", "v_mul_u32_u24 %[r1], %[r1], %[x]", "v_mul_u32_u24 %[r2], %[r2], %[x]", "v_mul_u32_u24 %[r3], %[r3], %[x]"),
     I4("v_mul_u32_u24 %[r4], %[r4], %[x]", "v_mul_u32_u24 %[r5], %[r5], %[x]", "v_mul_u32_u24 %[r6], %[r6], %[x]", "v_mul_u32_u24 %[r7], %[r7], %[x]"))
TEST(22, "v_add_f32_dpp quad_perm", I4("v_add_f32_dpp %[r0], %[r1], %[r0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %[r1], %[r2], %[r1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
                                       "v_add_f32_dpp %[r2], %[r3], %[r2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %[r3], %[r4], %[r3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
     I4("v_add_f32_dpp %[r4], %[r5], %[r4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %[r5], %[r6], %[r5] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
        "v_add_f32_dpp %[r6], %[r7], %[r6] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %[r7], %[r0], %[r7] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"))
TEST(23, "v_lshrrev_b32", I4("v_lshrrev_b32 %[r0], 15, %[r0]", "v_lshrrev_b32 %[r1], 15, %[r1]", "v_lshrrev_b32 %[r2], 15, %[r2]", "v_lshrrev_b32 %[r3], 15, %[r3]"),
     I4("v_lshrrev_b32 %[r4], 15, %[r4]", "v_lshrrev_b32 %[r5], 15, %[r5]", "v_lshrrev_b32 %[r6], 15, %[r6]", "v_lshrrev_b32 %[r7], 15, %[r7]"))
TEST(24, "v_max_f32", I4("v_max_f32 %[r0], %[r0], %[x]", "v_max_f32 %[r1], %[r1], %[x]", "v_max_f32 %[r2], %[r2], %[x]", "v_max_f32 %[r3], %[r3], %[x]"),
     I4("v_max_f32 %[r4], %[r4], %[x]", "v_max_f32 %[r5], %[r5], %[x]", "v_max_f32 %[r6], %[r6], %[x]", "v_max_f32 %[r7], %[r7], %[x]"))
TEST(25, "v_pk_add_u16", I4("v_pk_add_u16 %[r0], %[r0], %[x]", "v_pk_add_u16 %[r1], %[r1], %[x]", "v_pk_add_u16 %[r2], %[r2], %[x]", "v_pk_add_u16 %[r3], %[r3], %[x]"),
     I4("v_pk_add_u16 %[r4], %[r4], %[x]", "v_pk_add_u16 %[r5], %[r5], %[x]", "v_pk_add_u16 %[r6], %[r6], %[x]", "v_pk_add_u16 %[r7], %[r7], %[x]"))
TEST(26, "v_cndmask_b32 (vcc)", I4("v_cndmask_b32 %[r0], %[r0], %[y], vcc", "v_cndmask_b32 %[r1], %[r1], %[y], vcc", "v_cndmask_b32 %[r2], %[r2], %[y], vcc", "v_cndmask_b32 %[r3], %[r3], %[y], vcc"),
     I4("v_cndmask_b32 %[r4], %[r4], %[y], vcc", "v_cndmask_b32 %[r5], %[r5], %[y], vcc", "v_cndmask_b32 %[r6], %[r6], %[y], vcc", "v_cndmask_b32 %[r7], %[r7], %[y], vcc"))
TEST(27, "v_and_b32 (mask apply)", I4("v_and_b32 %[r0], %[r0], %[x]", "v_and_b32 %[r1], %[r1], %[x]", "v_and_b32 %[r2], %[r2], %[x]", "v_and_b32 %[r3], %[r3], %[x]"),
     I4("v_and_b32 %[r4], %[r4], %[x]", "v_and_b32 %[r5], %[r5], %[x]", "v_and_b32 %[r6], %[r6], %[x]", "v_and_b32 %[r7], %[r7], %[x]"))
TEST(28, "v_bfi_b32", I4("v_bfi_b32 %[r0], %[x], %[r0], %[y]", "v_bfi_b32 %[r1], %[x], %[r1], %[y]", "v_bfi_b32 %[r2], %[x], %[r2], %[y]", "v_bfi_b32 %[r3], %[x], %[r3], %[y]"),
     I4("v_bfi_b32 %[r4], %[x], %[r4], %[y]", "v_bfi_b32 %[r5], %[x], %[r5], %[y]", "v_bfi_b32 %[r6], %[x], %[r6], %[y]", "v_bfi_b32 %[r7], %[x], %[r7], %[y]"))
TEST(29, "v_mul_f32_sdwa (bf16 hi half src)", I4("v_mul_f32_sdwa %[r0], %[r0], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v_mul_f32_sdwa %[r1], %[r1], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
                                                 "v_mul_f32_sdwa %[r2], %[r2], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v_mul_f32_sdwa %[r3], %[r3], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
     I4("v_mul_f32_sdwa %[r4], %[r4], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v_mul_f32_sdwa %[r5], %[r5], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1",
        "v_mul_f32_sdwa %[r6], %[r6], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v_mul_f32_sdwa %[r7], %[r7], %[x] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"))

TEST(30, "ds_read_b64", I4("ds_read_b64 %[p0], %[la]", "ds_read_b64 %[p1], %[la] offset:512", "ds_read_b64 %[p2], %[la] offset:2048", "ds_read_b64 %[p3], %[la] offset:2560"),
     I4("ds_read_b64 %[p0], %[la] offset:4096", "ds_read_b64 %[p1], %[la] offset:4608", "ds_read_b64 %[p2], %[la] offset:6144", "ds_read_b64 %[p3], %[la] offset:6656"))
TEST(31, "ds_read_b64_tr_b16", I4("ds_read_b64_tr_b16 %[p0], %[la]", "ds_read_b64_tr_b16 %[p1], %[la] offset:512", "ds_read_b64_tr_b16 %[p2], %[la] offset:2048", "ds_read_b64_tr_b16 %[p3], %[la] offset:2560"),
     I4("ds_read_b64_tr_b16 %[p0], %[la] offset:4096", "ds_read_b64_tr_b16 %[p1], %[la] offset:4608", "ds_read_b64_tr_b16 %[p2], %[la] offset:6144", "ds_read_b64_tr_b16 %[p3], %[la] offset:6656"))
TEST(32, "ds_write_b64", I4("ds_write_b64 %[la], %[p0]", "ds_write_b64 %[la], %[p1] offset:512", "ds_write_b64 %[la], %[p2] offset:2048", "ds_write_b64 %[la], %[p3] offset:2560"),
     I4("ds_write_b64 %[la], %[p0] offset:4096", "ds_write_b64 %[la], %[p1] offset:4608", "ds_write_b64 %[la], %[p2] offset:6144", "ds_write_b64 %[la], %[p3] offset:6656"))
TEST(33, "ds_bpermute_b32", I4("ds_bpermute_b32 %[r0], %[la], %[r0]", "ds_bpermute_b32 %[r1], %[la], %[r1]", "ds_bpermute_b32 %[r2], %[la], %[r2]", "ds_bpermute_b32 %[r3], %[la], %[r3]"),
     I4("ds_bpermute_b32 %[r4], %[la], %[r4]", "ds_bpermute_b32 %[r5], %[la], %[r5]", "ds_bpermute_b32 %[r6], %[la], %[r6]", "ds_bpermute_b32 %[r7], %[la], %[r7]"))
TEST(34, "v_permlane32_swap", I4("v_permlane32_swap_b32 %[r0], %[r1]", "v_permlane32_swap_b32 %[r2], %[r3]", "v_permlane32_swap_b32 %[r4], %[r5]", "v_permlane32_swap_b32 %[r6], %[r7]"),
     I4("v_permlane32_swap_b32 %[r0], %[r2]", "v_permlane32_swap_b32 %[r1], %[r3]", "v_permlane32_swap_b32 %[r4], %[r6]", "v_permlane32_swap_b32 %[r5], %[r7]"))

template <typename T, int THREADS, int VARIANT>
__global__ __launch_bounds__(THREADS) void rate_kernel(int iters, unsigned long long *cycles, unsigned *sink) {
    uint32_t r[8];
    u32x2 p[4];
    unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u + 12345u);
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
    for (int i = 0; i < 8; ++i) r[i] = (rnd() & 0x007FFFFFu) | 0x3F000000u;   // floats in [0.5, 1)
    for (int i = 0; i < 4; ++i) { p[i][0] = (rnd() & 0x007FFFFFu) | 0x3F000000u; p[i][1] = (rnd() & 0x007FFFFFu) | 0x3F000000u; }
    const uint32_t x = (rnd() & 0x007FFFFFu) | 0x3F000000u, y = (rnd() & 0x007FFFFFu) | 0x3E000000u;
    u32x2 px; px[0] = x; px[1] = y;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((int)(rnd() & 255) - 128) * (__bf16)0.01f; b[i] = (__bf16)(float)((int)(rnd() & 255) - 128) * (__bf16)0.01f; }
    f32x16 acc = {};
    const uint32_t sx = __builtin_amdgcn_readfirstlane(x);
    const unsigned long long m64 = __builtin_amdgcn_readfirstlane(x) | 0x5555000000000000ull;
    __shared__ __attribute__((aligned(16))) char smem[16384];
    for (int i = threadIdx.x; i < 4096; i += THREADS) reinterpret_cast<uint32_t *>(smem)[i] = rnd();
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 1024;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) T::template run<VARIANT>(r, p, x, y, acc, a, b, sx, px, m64, la);
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    for (int i = 0; i < 4; ++i) s ^= p[i][0] ^ p[i][1];
    for (int i = 0; i < 16; ++i) s ^= __float_as_uint(acc[i]);
    if (s == 0x12345678u) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename T, int THREADS, int VARIANT>
static double run_one(int iters, unsigned long long *d_cycles, unsigned *d_sink) {
    const int nblk = 256, nw = THREADS / 64;
    hipLaunchKernelGGL((rate_kernel<T, THREADS, VARIANT>), dim3(nblk), dim3(THREADS), 0, 0, iters, d_cycles, d_sink);   // warm-up
    hipLaunchKernelGGL((rate_kernel<T, THREADS, VARIANT>), dim3(nblk), dim3(THREADS), 0, 0, iters, d_cycles, d_sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(nblk * nw);
    hipMemcpy(c.data(), d_cycles, c.size() * sizeof(c[0]), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    return (double)c[c.size() / 2] / iters;   // median wave: cycles per body
}

template <typename T>
static void report(unsigned long long *d_cycles, unsigned *d_sink, double mf1, double mf2) {
    const int iters = 200;
    // variant 0: 64 instructions; 1: 16 x (MFMA + 4); 2: 8 x (MFMA + 8)
    const double a1 = run_one<T, 256, 0>(iters, d_cycles, d_sink), a2 = run_one<T, 512, 0>(iters, d_cycles, d_sink);
    const double m41 = run_one<T, 256, 1>(iters, d_cycles, d_sink), m42 = run_one<T, 512, 1>(iters, d_cycles, d_sink);
    const double m81 = run_one<T, 256, 2>(iters, d_cycles, d_sink), m82 = run_one<T, 512, 2>(iters, d_cycles, d_sink);
    printf("%-36s | alone: %5.2f (1w) %5.2f (2w) cyc/instr | gap of 4: %6.1f (1w) %6.1f (2w) | gap of 8: %6.1f (1w) %6.1f (2w) cyc per MFMA+gap  [bare MFMA %.1f / %.1f]\n",
           T::name, a1 / 64, a2 / 64, m41 / 16, m42 / 16, m81 / 8, m82 / 8, mf1, mf2);
}

int main() {
    unsigned long long *d_cycles;
    unsigned *d_sink;
    hipMalloc(&d_cycles, 256 * 8 * sizeof(unsigned long long));
    hipMalloc(&d_sink, 64);
    const double mf1 = run_one<T0, 256, 3>(200, d_cycles, d_sink) / 16, mf2 = run_one<T0, 512, 3>(200, d_cycles, d_sink) / 16;
    printf("one wave's stream; 1w = alone on its SIMD (256-thread workgroup per CU), 2w = with a partner running the same stream (512 threads); median wave of 256 CUs\n");
    printf("bare v_mfma_f32_32x32x16_bf16 chain: %.1f (1w) %.1f (2w) cycles per MFMA\n", mf1, mf2);
#define R(ID) report<T##ID>(d_cycles, d_sink, mf1, mf2);
    R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23) R(24) R(25) R(26) R(27) R(28) R(29) R(30) R(31) R(32) R(33) R(34)
    return 0;
}
