// Micro-benchmark: sustained f32 MFMA rate on this device (sanity ceiling for the roofline).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16x16x4, 1: 32x32x2, 2: 16x16x4 + one ds_read_b128 per 4 MFMA
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if (MODE == 1) {
        f32x16 c0 = {0}, c1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3];
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        const f32x4 *lp = reinterpret_cast<const f32x4 *>(lds) + (threadIdx.x & 63);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f32x4 bv = {b, b, b, b};
                if (MODE == 2) bv = lp[(u * 64 + i * 8) & 1023 & ~63 | 0];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[2], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[3], c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

template <int MODE> void run(const char *name, int wg_per_cu, float *d) {
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_wave = (double)iters * (MODE == 1 ? 16 : 32);
    const double flop = mfma_per_wave * (MODE == 1 ? 4096.0 : 2048.0) * grid * 4;
    printf("%-28s %d WG/CU: %.2f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) { run<0>("mfma_f32_16x16x4", w, d); run<1>("mfma_f32_32x32x2", w, d); run<2>("16x16x4 + ds_read_b128/4", w, d); }
    return 0;
}
