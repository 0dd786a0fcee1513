// probe: what does ds_read_b64_tr_b8 return?  LDS byte a holds (a & 0xFF); lane l reads 8 bytes at address 8*l.  Output: for each lane the 8
// source byte addresses it received -> (source lane, source byte) pairs.   hipcc --offload-arch=gfx950 tools/probe_tr8.hip -o /tmp/probe_tr8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t *out) {
    __shared__ __attribute__((aligned(16))) unsigned char m[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) m[i] = (unsigned char)(i & 0xFF);
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)m + threadIdx.x * 8;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 2] = v[0];
    out[threadIdx.x * 2 + 1] = v[1];
}
int main() {
    uint32_t *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int b = 0; b < 8; ++b) {
            int src = (h[l * 2 + b / 4] >> (8 * (b % 4))) & 0xFF;   // byte address mod 256
            // addresses of lanes 32..63 wrap (>= 256): report mod 256 plus the lane's own 256-B half
            printf(" (L%2d,b%d)", (src / 8) + (l >= 32 ? 32 : 0), src % 8);
        }
        printf("\n");
    }
    return 0;
}
