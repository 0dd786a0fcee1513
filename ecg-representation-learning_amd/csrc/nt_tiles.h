// Tile walk of the persistent A . B^T kernels (gemm_nt.hip: the eight- and four-wave bodies; tools/experiments/gemm_ov.hip: round 5's four-wave body
// with the deferred epilogue): 256 x 256 output tiles, dealt XCD-contiguously over column groups of G n-tiles.
#pragma once
#include "common.h"

namespace nt_tiles {
constexpr int BM = 256, BN = 256;

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// item -> tile origin.  Items are dealt XCD-contiguously (blocks b and b + 8 share an XCD, so XCD x walks one contiguous run of
// tile ids); tile ids run over column groups of G n-tiles, m-major inside a group: the 32 tiles an XCD works on at a time share
// ~32/G activation panels and G weight panels.
__device__ __forceinline__ void decode_tile(int it, int ntile, int tiles_m, int tiles_n, int G, int &m0, int &n0) {
    const int tid = xcd_remap(it, ntile);
    const int full = G * tiles_m;
    const int ng = (tiles_n + G - 1) / G;
    int g = tid / full;
    g = g < ng - 1 ? g : ng - 1;
    const int rem = tid - g * full;
    const int w = (g == ng - 1) ? tiles_n - g * G : G;
    const int tm = rem / w, tn = g * G + (rem - tm * w);
    m0 = tm * BM;
    n0 = tn * BN;
}

}  // namespace nt_tiles
