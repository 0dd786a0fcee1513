// Shared device helpers of the fused attention kernels (attention.hip: forward, one-item backward; attention_bwd4.hip: the four-wave
// persistent backward): the swizzled [row][64 x bf16] LDS image, its row / transposed fragment reads, the dS^T image slot swizzle.
//
// LDS image for every [row][64 x bf16] tile (128-B rows):  16-B chunk index ^= bitrev3((row>>1)&7)
//   -> ds_read_b128 row reads (MFMA K-contiguous operand) hit 16 distinct slots per 16-lane group, and
//   -> ds_read_b64_tr_b16 reads of 4 consecutive rows x 64 B land in the 4 different 64-B quarters of the bank row.
#pragma once
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int swz3(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x & 2) | (x >> 2);
}
// slot swizzle of the 64-B-row dS^T image: key bits (1,2,3) -> slot bits (0,2,1)
__device__ __forceinline__ int dsw(int key) { return ((key >> 1) & 1) | (((key >> 2) & 1) << 2) | (((key >> 3) & 1) << 1); }
__device__ __forceinline__ int img_off(int row, int byte) { return row * 128 + ((((byte >> 4) ^ swz3(row)) << 4) | (byte & 15)); }

// stage `rows_pad` rows x 128 B from global (row stride `ld` elements, rows >= nvalid zero-filled) into an image
template <int NT>
__device__ __forceinline__ void stage_image(char *img, const bf16_t *__restrict__ g, int64_t ld, int nvalid, int rows_pad) {
    for (int c = threadIdx.x; c < rows_pad * 8; c += NT) {
        const int row = c >> 3, ch = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nvalid) v = *reinterpret_cast<const u32x4 *>(g + (int64_t)row * ld + ch * 8);
        *reinterpret_cast<u32x4 *>(img + img_off(row, ch * 16)) = v;
    }
}

// Same image, staged by LDS-DMA (`buffer_load ... lds`): asynchronous, no VGPR round trip, all pieces of all images in flight
// at once.  One piece = 8 image rows (1 KiB); the swizzle is applied to the per-lane SOURCE chunk; rows >= nvalid fall beyond
// the descriptor's num_records and read as zero (hardware bounds check).  Caller waits (vmcnt(0)) and barriers.
typedef __attribute__((address_space(3))) void *lds_void_p;
template <int NW>
__device__ __forceinline__ void dma_image(char *img, const bf16_t *g, int64_t ld, int nvalid, int rows_pad, int wave, int lane) {
    const uint32_t bytes = (uint32_t)(((int64_t)(nvalid - 1) * ld + 64) * 2);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, bytes, 0x00020000);
    const int npiece = rows_pad >> 3;
    for (int j = wave; j < npiece; j += NW) {
        const int row = j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz3(row);
        const int voff = (int)(((int64_t)row * ld + chunk * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_p)(img + j * 1024), 16, voff, 0, 0, 0);
    }
}

// A/B operand fragment of a 32x32x16 MFMA whose k runs along the image's 64 columns: X[row0 + (lane&31)][16*ks + 8*(lane>>5) + j]
__device__ __forceinline__ bf16x8 row_frag(const char *img, int row0, int ks, int lane) {
    return *reinterpret_cast<const bf16x8 *>(img + img_off(row0 + (lane & 31), (ks * 16 + 8 * (lane >> 5)) * 2));
}

// fragment whose k runs along the image ROWS in the "accumulator order" of a 32x32 tile:
//   element j of lane (r = lane&31, h = lane>>5)  =  X[row0 + 8*(j>>2) + 4h + (j&3)][col0 + r]
// (two transposed reads of 4 rows x 16 columns per 16-lane group)
__device__ __forceinline__ bf16x8 tr_frag32(const char *img, int row0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int row = row0 + 4 * (g >> 1) + (i >> 2);
    const int colb = (col0 + (g & 1) * 16 + (i & 3) * 4) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + img_off(row, colb)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img + img_off(row + 8, colb)));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
}

// ---- lane-constant address parts.  Every fragment read below starts at a row that is a multiple of 16, and the image
// swizzle only looks at row bits 1..3, so the swizzled byte offset splits into (uniform row0 * 128) + a per-lane constant
// computed ONCE per kernel: the inner loops then spend one v_add per base instead of ~12 VALU ops per read.
struct RowOff { int ks[4]; };        // row_frag: lane row (lane&31), k-step ks
struct TrOff { int lo[2], hi[2]; };  // tr_frag32: column block dt = 0/1 (32 columns each), first / second (rows + 8) read
__device__ __forceinline__ RowOff make_row_off(int lane) {
    RowOff r;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r.ks[ks] = img_off(lane & 31, (ks * 16 + 8 * (lane >> 5)) * 2);
    return r;
}
__device__ __forceinline__ TrOff make_tr_off(int lane) {
    TrOff t;
    const int g = lane >> 4, i = lane & 15;
    const int row = 4 * (g >> 1) + (i >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int colb = (dt * 32 + (g & 1) * 16 + (i & 3) * 4) * 2;
        t.lo[dt] = img_off(row, colb);
        t.hi[dt] = img_off(row + 8, colb);
    }
    return t;
}
__device__ __forceinline__ bf16x8 row_frag_c(const char *img_row0, int off) { return *reinterpret_cast<const bf16x8 *>(img_row0 + off); }
__device__ __forceinline__ bf16x8 join_halves(bf16x4 a, bf16x4 b) {
    // two 8-byte halves -> one 16-byte fragment without per-element moves
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    u32x4 u;
    u[0] = ua[0]; u[1] = ua[1]; u[2] = ub[0]; u[3] = ub[1];
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 tr_frag_c(const char *img_row0, int lo, int hi) {
    const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img_row0 + lo));
    const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(img_row0 + hi));
    return join_halves(a, b);
}

// pack accumulator registers 8*ss .. 8*ss+7 into the bf16 B-operand fragment: explicit PAIRS (one v_cvt_pk_bf16_f32 per two values;
// element-wise casts compile to one convert per value plus a v_perm per pair)
__device__ __forceinline__ bf16x8 pack8(const f32x16 &x, int ss) {
    typedef float f32x2_p __attribute__((ext_vector_type(2)));
    u32x4 u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2_p v; v[0] = x[8 * ss + 2 * j]; v[1] = x[8 * ss + 2 * j + 1];
        u[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    }
    return __builtin_bit_cast(bf16x8, u);
}

// Transposed LDS read issued as inline asm: hipcc's waitcnt pass cannot see through the builtin whether an LDS-DMA still in flight
// aliases the read and drains vmcnt(0) in front of it -- fatal for a kernel whose operand stream is never supposed to drain.  The
// asm form is invisible to that pass; the caller orders it by hand (s_waitcnt lgkmcnt + sched_barrier before the consumer).
__device__ __forceinline__ bf16x4 tr_read_asm(uint32_t lds_addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ bf16x4 tr_read_asm_o(uint32_t lds_addr) {   // immediate offset: no address VALU
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
// two f32 -> one dword of two bf16 (v_cvt_pk_bf16_f32): explicit pairs, so every accumulator value is converted exactly once
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    f32x2_t v; v[0] = a; v[1] = b;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint32_t lds_addr_of(const char *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }

}  // namespace
