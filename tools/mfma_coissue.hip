// Micro-benchmark: how do VALU / LDS instructions of one wave share a SIMD with the MFMA stream of
// another wave?  One workgroup of 512 threads per CU: waves 0-3 ("consumers", one per SIMD) issue
// NM independent MFMAs per iteration, waves 4-7 ("producers") issue NV VALU instructions (or LDS
// reads) per iteration; no barriers inside the loop.  Each role stamps its own duration with
// s_memtime, so the table shows how long each role takes alone and together.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

// MODE: 0 = 16x16x4 f32 (32 cycles), 1 = 32x32x2 f32 (64 cycles)
// VK:   0 = independent v_fma_f32 x8 accumulators, 1 = dependent v_fma_f32 chain, 2 = v_pk_fma_f32 independent,
//       3 = ds_read_b128 (LDS)
template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(unsigned long long *stamps, float *sink, int iters, int run_m, int run_v, int prio) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)(i & 7);
    __syncthreads();
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) stamps[4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(63492);   // HW_ID: wave slot [3:0], SIMD [5:4]
    unsigned long long t0, t1;
    if (!producer) {
        if (!run_m) return;
        const float av = (float)(threadIdx.x & 3), bv = 1.0f;
        if constexpr (MODE == 0) {
            f32x4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            STAMP(t0);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
            }
            STAMP(t1);
            float s = 0;
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
            sink[blockIdx.x * 512 + threadIdx.x] = s;
        } else {
            f32x16 acc[4];
            for (int i = 0; i < 4; ++i)
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            STAMP(t0);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
            STAMP(t1);
            float s = 0;
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
            sink[blockIdx.x * 512 + threadIdx.x] = s;
        }
        if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
    } else {
        if (!run_v) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x = (float)threadIdx.x * 1e-3f;
        float s = 0;
        STAMP(t0);
        if constexpr (VK == 0) {
            float a[8];
            for (int i = 0; i < 8; ++i) a[i] = x + i;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(x));
            }
            for (int i = 0; i < 8; ++i) s += a[i];
        } else if constexpr (VK == 1) {
            float a = x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(x));
            }
            s = a;
        } else if constexpr (VK == 2) {
            f32x2 a[8];
            for (int i = 0; i < 8; ++i) a[i] = f32x2{x + i, x - i};
            const f32x2 xx = {x, x};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(xx));
            }
            for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
        } else {
            const float *p = lds + (threadIdx.x & 63) * 4;
            f32x4 a[8];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i]) : "v"((unsigned)(size_t)p), "i"(i * 1024));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            for (int i = 0; i < 8; ++i) s += a[i][0];
        }
        STAMP(t1);
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 256 && blockIdx.x == 0) stamps[1] = t1 - t0;
    }
}

// Same wave: NV independent VALU instructions after every MFMA (NV = 0, 1, 2, 4, 8).
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_same(unsigned long long *stamps, float *sink, int iters) {
    const float av = (float)(threadIdx.x & 3), bv = 1.0f;
    float x = (float)threadIdx.x * 1e-3f, a[8];
    for (int i = 0; i < 8; ++i) a[i] = x + i;
    unsigned long long t0, t1;
    f32x4 acc4[8];
    f32x16 acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == 0) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc4[i], 0, 0, 0);
            else                     acc16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc16[i & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(i + v) & 7]) : "v"(x));
        }
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + acc4[i][0];
    for (int i = 0; i < 4; ++i) s += acc16[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
}
template <int MODE, int NV>
void run_same(unsigned long long *d_st, float *d_sink) {
    const int iters = 4000;
    unsigned long long h[2];
    hipLaunchKernelGGL((k_same<MODE, NV>), dim3(256), dim3(256), 0, 0, d_st, d_sink, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost);
    printf("same wave, %s + %d v_fma_f32 after each MFMA: %.1f cycles per MFMA\n", MODE == 0 ? "16x16x4" : "32x32x2", NV, h[0] / (8.0 * iters));
}

template <int MODE, int VK>
void run(const char *name, unsigned long long *d_st, float *d_sink) {
    const int iters = 2000;
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int rm = cfg != 1, rv = cfg != 0, prio = cfg == 3;
        unsigned long long z[2] = {0, 0}, h[2];
        hipMemcpy(d_st, z, 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, d_st, d_sink, iters, rm, rv, prio);
        hipDeviceSynchronize();
        hipMemcpy(h, d_st, 16, hipMemcpyDeviceToHost);
        const double nm = 32.0 * iters, nv = 32.0 * iters;
        if (cfg == 2) { unsigned long long hh[12]; hipMemcpy(hh, d_st, 96, hipMemcpyDeviceToHost); printf("   waves 0..7: SIMD"); for (int w = 0; w < 8; ++w) printf(" %llu", (hh[4 + w] >> 4) & 3); printf("  slot"); for (int w = 0; w < 8; ++w) printf(" %llu", hh[4 + w] & 15); printf("\n"); }
        printf("%-34s %-22s mfma %7.1f cyc/inst   other %7.1f cyc/inst\n", name,
               cfg == 0 ? "mfma alone" : cfg == 1 ? "other alone" : cfg == 2 ? "together" : "together, other prio 3",
               rm ? h[0] / nm : 0.0, rv ? h[1] / nv : 0.0);
    }
}

int main() {
    unsigned long long *d_st; float *d_sink;
    hipMalloc(&d_st, 256); hipMalloc(&d_sink, 256 * 512 * 4);
    run<0, 0>("16x16x4 + indep v_fma_f32", d_st, d_sink);
    run<0, 1>("16x16x4 + dependent v_fma_f32", d_st, d_sink);
    run<0, 2>("16x16x4 + indep v_pk_fma_f32", d_st, d_sink);
    run<0, 3>("16x16x4 + ds_read_b128", d_st, d_sink);
    run<1, 0>("32x32x2 + indep v_fma_f32", d_st, d_sink);
    run<1, 1>("32x32x2 + dependent v_fma_f32", d_st, d_sink);
    run<1, 2>("32x32x2 + indep v_pk_fma_f32", d_st, d_sink);
    run<1, 3>("32x32x2 + ds_read_b128", d_st, d_sink);
    run_same<0, 0>(d_st, d_sink); run_same<0, 1>(d_st, d_sink); run_same<0, 2>(d_st, d_sink); run_same<0, 4>(d_st, d_sink); run_same<0, 8>(d_st, d_sink);
    run_same<1, 0>(d_st, d_sink); run_same<1, 1>(d_st, d_sink); run_same<1, 2>(d_st, d_sink); run_same<1, 4>(d_st, d_sink); run_same<1, 8>(d_st, d_sink);
    run_same<1, 16>(d_st, d_sink);
    return 0;
}
