// probe: which hardware registers tell two co-resident 256-thread / 80-KiB workgroups of one CU apart?  Each workgroup records HW_ID (reg 4),
// LDS_ALLOC (reg 6), XCC_ID (reg 20) and two timestamps.  hipcc --offload-arch=gfx950 tools/probe_hwid.hip -o tools/probe_hwid.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(uint32_t *out) {
    __shared__ char smem[81920];
    smem[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t hw, lds, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = lds; out[blockIdx.x * 4 + 2] = xcc; out[blockIdx.x * 4 + 3] = smem[5];
    }
    // stay resident long enough that all 512 workgroups coexist
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(32);
}
int main() {
    uint32_t *d; static uint32_t h[512 * 4];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::map<uint64_t, std::vector<int>> cu;
    for (int b = 0; b < 512; ++b) {
        const uint32_t hw = h[b * 4], xcc = h[b * 4 + 2] & 0xF;
        const uint32_t cu_id = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("distinct (xcc, se, sh, cu) = %zu\n", cu.size());
    int shown = 0;
    for (auto &kv : cu) {
        if (shown++ < 12) {
            printf("cu %05llx:", (unsigned long long)kv.first);
            for (int b : kv.second) printf("  block %3d hw %08x tg %u wave %u simd %u lds %08x", b, h[b * 4], (h[b * 4] >> 16) & 0xF, h[b * 4] & 0xF, (h[b * 4] >> 4) & 3, h[b * 4 + 1]);
            printf("\n");
        }
    }
    int two = 0; for (auto &kv : cu) two += kv.second.size() == 2;
    printf("CUs with exactly two workgroups: %d\n", two);
    return 0;
}
