// Micro-benchmark for the next design step: do bf16 MFMAs (dense matrix cores) share their time with the vector ALU the way
// fp32 MFMAs do (r01: fp32 MFMA and VALU serialise on a SIMD -- the fp32 "matrix" rate equals the fp32 vector rate)?
//   A. one wave: NV independent v_fma_f32 after every v_mfma_f32_32x32x16_bf16 (cycles per MFMA for NV = 0, 1, 2, 4, 8, 16)
//   B. two waves per SIMD: wave 0-3 a dense MFMA stream, waves 4-7 a VALU stream (or the other way round): each alone, then together.
//      Each role stamps its start BEFORE its first vector instruction: a starved wave sits on that instruction, and a stamp taken
//      after the set-up code measures the loop only after the other role has finished (which reads as 'perfect overlap').
//   D. r01's kernel shape with the stamp in either place, to show exactly that
//   C. cost of splitting fp32 into three bf16 pieces (the operand preparation of an fp32-exact 'bf16x3' product), VALU cycles per element
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_coissue.hip -o /tmp/mfma_bf16 && /tmp/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

template <int NV, bool F32>
__global__ __launch_bounds__(256) void k_same(unsigned long long *stamps, float *sink, int iters) {
    const u32x4 ab = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    float x = (float)threadIdx.x * 1e-3f, a[8];
    for (int i = 0; i < 8; ++i) a[i] = x + i;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (F32) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, x, acc[i & 3], 0, 0, 0);
            else acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, ab), acc[i & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[(i + v) & 7]) : "v"(x));
        }
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
}

template <bool F32, int VREP, bool PK>
__global__ __launch_bounds__(512) void k_two(unsigned long long *stamps, float *sink, int iters, int run_m, int run_v, int valu_first) {
    if (valu_first & 2) __syncthreads();                 // bit 1: a workgroup barrier before the roles start (as every real kernel has)
    const bool first_half = __builtin_amdgcn_readfirstlane(threadIdx.x) < 256;
    const bool mfma_role = (valu_first & 1) ? !first_half : first_half;
    unsigned long long t0, t1;
    if (mfma_role) {
        if (!run_m) return;
        STAMP(t0);
        const u32x4 ab = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        const float x = (float)threadIdx.x * 1e-3f;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (F32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, x, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ab), __builtin_bit_cast(bf16x8, ab), acc[i], 0, 0, 0);
                }
        }
        STAMP(t1);
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
    } else {
        if (!run_v) return;
        STAMP(t0);                                   // BEFORE the role's first vector instruction: a starved wave is stuck on exactly that one
        const float x = (float)threadIdx.x * 1e-3f;
        f32x2 a[8];
        for (int i = 0; i < 8; ++i) a[i] = f32x2{x + i, x - i};
        const f32x2 xx = {x, x};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < VREP; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if constexpr (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(xx));
                    else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i][0]) : "v"(x));
                }
        }
        STAMP(t1);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if ((threadIdx.x & 255) == 0 && blockIdx.x == 0) stamps[1] = t1 - t0;
    }
}


// r01's arrangement, with knobs to find what decides between "VALU wave starves" and "VALU wave runs freely":
// LDSK: touch LDS + barrier first; NM: MFMAs per iteration; acc count fixed at 4
template <bool LDSK, int NM, bool EARLY = false>
__global__ __launch_bounds__(512) void k_r1(unsigned long long *stamps, float *sink, int iters, int run_m, int run_v, int prio = 0) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    if (LDSK) { for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)(i & 7); __syncthreads(); }
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) stamps[4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(63492);   // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]
    unsigned long long t0, t1;
    if (!producer) {
        if (!run_m) return;
        const float av = (float)(threadIdx.x & 3), bv = LDSK ? lds[threadIdx.x & 7] : 1.0f;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        STAMP(t0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < NM / 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
        }
        STAMP(t1);
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
    } else {
        if (!run_v) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x = (float)threadIdx.x * 1e-3f;
        if (EARLY) STAMP(t0);
        f32x2 a[8];
        for (int i = 0; i < 8; ++i) a[i] = f32x2{x + i, x - i};
        const f32x2 xx = {x, x};
        if (!EARLY) STAMP(t0);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(xx));
        }
        STAMP(t1);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 256 && blockIdx.x == 0) stamps[1] = t1 - t0;
    }
}

// C: exact three-way split of fp32 into bf16 pieces by truncation: x = h + m + l, each with <= 8 significant bits
__global__ __launch_bounds__(256) void k_split(const float *in, unsigned *out, int n_per_thread, unsigned long long *stamps) {
    const float *p = in + (size_t)(blockIdx.x * 256 + threadIdx.x) * n_per_thread;
    unsigned *o = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * n_per_thread * 2;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int i = 0; i < n_per_thread; i += 2) {
        const float x0 = p[i], x1 = p[i + 1];
        const unsigned h0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u, h1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
        const float r0 = x0 - __builtin_bit_cast(float, h0), r1 = x1 - __builtin_bit_cast(float, h1);
        const unsigned m0 = __builtin_bit_cast(unsigned, r0) & 0xffff0000u, m1 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const float q0 = r0 - __builtin_bit_cast(float, m0), q1 = r1 - __builtin_bit_cast(float, m1);
        const unsigned l0 = __builtin_bit_cast(unsigned, q0), l1 = __builtin_bit_cast(unsigned, q1);
        o[2 * i + 0] = (h0 >> 16) | h1;                 // packed bf16 pairs of the high, middle and low pieces
        o[2 * i + 1] = (m0 >> 16) | m1;
        o[2 * i + 2] = (l0 >> 16) | (l1 & 0xffff0000u);
    }
    STAMP(t1);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = t1 - t0;
}

template <int NV, bool F32> double run_same(unsigned long long *d_st, float *d_sink, int iters) {
    hipLaunchKernelGGL((k_same<NV, F32>), dim3(256), dim3(256), 0, 0, d_st, d_sink, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k_same<NV, F32>), dim3(256), dim3(256), 0, 0, d_st, d_sink, iters);
    unsigned long long h[4];
    hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost);
    return (double)h[0] / (iters * 8.0);
}

int main() {
    unsigned long long *d_st; float *d_sink;
    hipMalloc(&d_st, 256); hipMalloc(&d_sink, 4 * 512 * 512);
    const int iters = 2000;
    // s_memtime ticks at 100 MHz: convert with the measured solo MFMA time (bf16 32x32x16 = 8 passes = 32 cycles at full rate)
    printf("A. one wave, NV v_fma_f32 after every MFMA: ticks (10 ns) per MFMA\n");
    printf("   bf16 32x32x16: NV=0 %.3f  1 %.3f  2 %.3f  4 %.3f  8 %.3f  16 %.3f\n", run_same<0, false>(d_st, d_sink, iters), run_same<1, false>(d_st, d_sink, iters),
           run_same<2, false>(d_st, d_sink, iters), run_same<4, false>(d_st, d_sink, iters), run_same<8, false>(d_st, d_sink, iters), run_same<16, false>(d_st, d_sink, iters));
    printf("   f32  32x32x2 : NV=0 %.3f  1 %.3f  2 %.3f  4 %.3f  8 %.3f  16 %.3f\n", run_same<0, true>(d_st, d_sink, iters), run_same<1, true>(d_st, d_sink, iters),
           run_same<2, true>(d_st, d_sink, iters), run_same<4, true>(d_st, d_sink, iters), run_same<8, true>(d_st, d_sink, iters), run_same<16, true>(d_st, d_sink, iters));
    auto report = [&](const char *what) {
        unsigned long long h[4];
        hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost);
        printf("      %-34s mfma role %8.0f  valu role %8.0f\n", what, (double)h[0], (double)h[1]);
    };
#define TWO(F32, VREP, PK, RM, RV, VF, WHAT) { unsigned long long z[4] = {0, 0, 0, 0}; for (int rep = 0; rep < 2; ++rep) { hipMemcpy(d_st, z, 32, hipMemcpyHostToDevice); \
        hipLaunchKernelGGL((k_two<F32, VREP, PK>), dim3(256), dim3(512), 0, 0, d_st, d_sink, iters, RM, RV, VF); hipDeviceSynchronize(); } report(WHAT); }
#define SERIES(F32, VREP, PK, NAME) { printf("B. %s, %d MFMAs and %d %s per iteration, %d iterations (cycles)\n", F32 ? "f32 32x32x2" : "bf16 32x32x16", 32, 8 * VREP, NAME, iters); \
        TWO(F32, VREP, PK, 1, 0, 0, "MFMA alone") TWO(F32, VREP, PK, 0, 1, 0, "VALU alone") TWO(F32, VREP, PK, 1, 1, 0, "together, MFMA in waves 0-3") TWO(F32, VREP, PK, 1, 1, 1, "together, VALU in waves 0-3") \
        TWO(F32, VREP, PK, 1, 1, 2, "barrier first, MFMA in waves 0-3") TWO(F32, VREP, PK, 1, 1, 3, "barrier first, VALU in waves 0-3") }
    SERIES(true, 4, true, "v_pk_fma_f32") SERIES(true, 16, true, "v_pk_fma_f32") SERIES(true, 48, true, "v_pk_fma_f32")
    SERIES(true, 4, false, "v_fma_f32") SERIES(true, 16, false, "v_fma_f32") SERIES(true, 48, false, "v_fma_f32")
    SERIES(false, 4, true, "v_pk_fma_f32") SERIES(false, 16, true, "v_pk_fma_f32") SERIES(false, 48, false, "v_fma_f32")
#define R1(LDSK, NM, WHAT) { unsigned long long z[4] = {0, 0, 0, 0}; for (int rep = 0; rep < 3; ++rep) { hipMemcpy(d_st, z, 32, hipMemcpyHostToDevice); \
        hipLaunchKernelGGL((k_r1<LDSK, NM>), dim3(256), dim3(512), 0, 0, d_st, d_sink, iters, 1, 1); hipDeviceSynchronize(); report(WHAT); } }
    printf("D. r01's kernel shape (MFMA waves 0-3, v_pk_fma_f32 waves 4-7, 32 VALU per iteration), together:\n");
    R1(true, 16, "LDS+barrier, 16 MFMA/iter")
    { unsigned long long h[12]; hipMemcpy(h, d_st, 96, hipMemcpyDeviceToHost); printf("      waves 0..7 of workgroup 0: SIMD"); for (int w = 0; w < 8; ++w) printf(" %llu", (h[4 + w] >> 4) & 3); printf("  wave slot"); for (int w = 0; w < 8; ++w) printf(" %llu", h[4 + w] & 15); printf("\n"); } R1(false, 16, "no LDS, 16 MFMA/iter") R1(true, 32, "LDS+barrier, 32 MFMA/iter") R1(false, 32, "no LDS, 32 MFMA/iter")
#define R1E(WHAT) { unsigned long long z[4] = {0, 0, 0, 0}; for (int rep = 0; rep < 2; ++rep) { hipMemcpy(d_st, z, 32, hipMemcpyHostToDevice); \
        hipLaunchKernelGGL((k_r1<true, 16, true>), dim3(256), dim3(512), 0, 0, d_st, d_sink, iters, 1, 1, 0); hipDeviceSynchronize(); report(WHAT); } }
    R1E("t0 stamped before the VALU setup")
    const int npt = 512;
    float *d_in; unsigned *d_out;
    hipMalloc(&d_in, (size_t)256 * 256 * npt * 4); hipMalloc(&d_out, (size_t)256 * 256 * npt * 8);
    hipMemset(d_in, 0x3f, (size_t)256 * 256 * npt * 4);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_split, dim3(256), dim3(256), 0, 0, d_in, d_out, npt, d_st); hipDeviceSynchronize(); }
    unsigned long long h[4];
    hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost);
    printf("C. exact fp32 -> 3 x bf16 split incl. load and packed stores: %.3f ticks per element per lane (one wave per SIMD)\n", (double)h[2] / npt);
    return 0;
}
