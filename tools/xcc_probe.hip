// Workgroup -> XCD placement probe (HW_REG_XCC_ID per block) for three workgroup shapes; evidence behind the XCD-local tile orders of
// kernels_ws.hip / kernels_tail.hip (speed only, never correctness).  hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o tools/_bin/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out) {
    extern __shared__ char lds[];
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
    if (threadIdx.x == 1) lds[0] = 1;
    // keep the block alive a little so that all blocks are co-resident
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000) {}
}
int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int threads = cfg == 0 ? 256 : 512, lds = cfg == 2 ? 158 * 1024 : 1024, grid = 256;
        unsigned *d; hipMalloc(&d, grid * 4);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, d);
        std::vector<unsigned> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
        int same = 0; for (int b = 0; b + 8 < grid; ++b) same += (h[b] & 0xf) == (h[b + 8] & 0xf);
        int rr = 0; for (int b = 0; b < grid; ++b) rr += (h[b] & 0xf) == ((h[0] + b) & 7);
        printf("threads %d lds %d: first 16 xcc ids:", threads, lds);
        for (int b = 0; b < 16; ++b) printf(" %u", h[b] & 0xf);
        printf(" | blocks b, b+8 on the same XCD: %d of %d; xcc == (xcc0 + b) %% 8: %d of %d\n", same, grid - 8, rr, grid);
        hipFree(d);
    }
    return 0;
}
