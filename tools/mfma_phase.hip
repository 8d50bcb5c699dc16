// Micro-benchmark mirroring conv_mfma_kernel's MFMA phase (MB=32, PBW=5, KSTEPS=8, 9 taps):
// operands from LDS one tap ahead, no global traffic, no barriers.  Isolates the inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N, int I = 0, class F> __device__ __forceinline__ void unroll(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll<N, I + 1>(f); } }

template <int PBW, int PIN>
__global__ __launch_bounds__(256) void k(float *out, int chunks, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < lds_floats; i += 256) lds[i] = (float)((i * 7) & 15) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 5, pl = lane & 31;
    constexpr int XS = 20, IW = 28, KS2 = 9, KSTEPS = 8;
    const float *xs = lds, *ws = lds + 392 * XS;
    int lbase[PBW];
    for (int pb = 0; pb < PBW; ++pb) { int q = (pb * 2) * 32 + pl; q %= 312; lbase[pb] = ((q / 26) * IW + q % 26) * XS + KSTEPS * g; }
    f32x16 acc[PBW];
    for (int pb = 0; pb < PBW; ++pb) for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
    const float *wbase = ws + lane * KSTEPS;
    for (int ch = 0; ch < chunks; ++ch) {
        float av[2][KSTEPS], bv[2][PBW][KSTEPS];
        auto load = [&](auto tc, auto sc) {
            constexpr int T = decltype(tc)::value, S = decltype(sc)::value;
            *(f32x4 *)&av[S][0] = *(const f32x4 *)(wbase + T * 64 * KSTEPS);
            *(f32x4 *)&av[S][4] = *(const f32x4 *)(wbase + T * 64 * KSTEPS + 4);
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb) {
                *(f32x4 *)&bv[S][pb][0] = *(const f32x4 *)(xs + lbase[pb] + ((T / 3) * IW + T % 3) * XS);
                *(f32x4 *)&bv[S][pb][4] = *(const f32x4 *)(xs + lbase[pb] + ((T / 3) * IW + T % 3) * XS + 4);
            }
        };
        load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        unroll<KS2>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (t + 1 < KS2) load(std::integral_constant<int, t + 1>{}, std::integral_constant<int, (t + 1) & 1>{});
            if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
                for (int pb = 0; pb < PBW; ++pb)
                    acc[pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][s], bv[t & 1][pb][s], acc[pb], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        });
    }
    float s = 0;
    for (int pb = 0; pb < PBW; ++pb) for (int r = 0; r < 16; ++r) s += acc[pb][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PBW, int PIN> void run(int wg_per_cu, float *d) {
    const int chunks = 64, grid = 256 * wg_per_cu, lds_floats = 392 * 20 + 9 * 64 * 8 * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kern = k<PBW, PIN>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_floats * 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_floats * 4, 0, d, 1, lds_floats);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_floats * 4, 0, d, chunks, lds_floats);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)chunks * 9 * 8 * PBW * 4096.0 * grid * 4;
    printf("PBW %d pin %d  %d WG/CU: %.3f ms  %.1f TFLOP/s (%.0f%%)\n", PBW, PIN, wg_per_cu, ms, flop / (ms * 1e-3) / 1e12,
           100 * flop / (ms * 1e-3) / 157.3e12);
}

int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2}) { run<5, 0>(w, d); run<5, 1>(w, d); run<3, 1>(w, d); }
    return 0;
}
