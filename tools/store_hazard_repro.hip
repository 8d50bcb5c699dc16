// Stand-alone reproducer of the wide-store data hazard met in round 6 (profiles/r06_notes.md section 10, kernels_ws.hip store_b128_sofs).
//   hipcc --offload-arch=gfx950 -O2 tools/store_hazard_repro.hip -o tools/_bin/store_hazard_repro && tools/_bin/store_hazard_repro
//
// The victim kernel is one hand-written instruction sequence per variant (inline asm, fixed registers v[100:103], so that the compiler neither pads
// nor reorders it):
//     v100..v103 <- the values to store        (tagged with the element index, so a wrong store is recognisable)
//     buffer_store_dwordx4 v[100:103], voff, rsrc, SOFF offen
//     PAD
//     v100..v103 <- 0xDEAD0000 | j             (the "next piece" being packed into the same registers)
// Variant 0: SOFF = an SGPR, no PAD           -- what hipcc emitted for kernels_ws.hip (LLVM pads the hazard only when soffset is a constant)
// Variant 1: SOFF = an SGPR, PAD = s_nop 0    -- one wait state
// Variant 2: SOFF = an SGPR, PAD = s_waitcnt expcnt(0); s_nop 7; s_nop 7  -- what store_b128_sofs ships
// Variant 3: SOFF = 0 (constant), no PAD      -- the form hipcc DOES pad (2 wait states on gfx94x/95x); unpadded here to see the hardware
// Variants 4, 5: buffer_store_dwordx2 / x3 with an SGPR soffset and no PAD -- where the hazard starts (the memset leaves the unwritten dwords of a piece 0)
// Each variant runs alone and beside a streaming copy kernel on a second stream (memory back-pressure); the host counts 16-byte pieces that
// hold a 0xDEAD.... value.  Nothing else is shared between the two kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void victim(unsigned *out, int pieces_per_wave, int rounds) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // this wave's slab: pieces_per_wave x 64 lanes x 16 bytes; a store instruction writes 1 KB of consecutive bytes
    unsigned *base = out + (size_t)wave * pieces_per_wave * 64 * 4;
    // raw buffer descriptor in SGPRs: base, stride 0, range = the slab, flags as __builtin_amdgcn_make_buffer_rsrc(..., 0x00020000) sets them
    const unsigned long long ba = (unsigned long long)base;
    const u32x4 rsrc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ba), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ba >> 32)),
                        (unsigned)__builtin_amdgcn_readfirstlane(pieces_per_wave * 1024), 0x00020000u};
    const unsigned voff = lane * 16;
    for (int r = 0; r < rounds; ++r)
        for (int p = 0; p < pieces_per_wave; ++p) {
            const unsigned tag = ((unsigned)wave << 12 | (unsigned)p) << 2;          // element j of the piece holds (tag | j), never 0xDEAD....
            const unsigned soff = __builtin_amdgcn_readfirstlane(p * 1024);
            const unsigned junk = 0xDEAD0000u | (unsigned)(p & 0xfff);
            if constexpr (VARIANT == 4 || VARIANT == 5) {     // narrower stores, SGPR soffset, no wait state: x2 (8 bytes: no hazard documented) and x3 (12 bytes)
                asm volatile("v_or_b32 v100, %[t], 0\n\tv_or_b32 v101, %[t], 1\n\tv_or_b32 v102, %[t], 2\n\tv_or_b32 v103, %[t], 3\n\ts_nop 4\n\t"
                             ".if %[variant] == 4\n\tbuffer_store_dwordx2 v[100:101], %[vo], %[rs], %[so] offen\n\t.else\n\tbuffer_store_dwordx3 v[100:102], %[vo], %[rs], %[so] offen\n\t.endif\n\t"
                             "v_mov_b32 v100, %[j]\n\tv_mov_b32 v101, %[j]\n\tv_mov_b32 v102, %[j]\n\tv_mov_b32 v103, %[j]\n\t"
                             :: [t] "v"(tag), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff), [j] "v"(junk), [variant] "n"(VARIANT)
                             : "v100", "v101", "v102", "v103", "memory");
            } else if constexpr (VARIANT == 3) {
                const unsigned vo = voff + soff;
                asm volatile("v_or_b32 v100, %[t], 0\n\tv_or_b32 v101, %[t], 1\n\tv_or_b32 v102, %[t], 2\n\tv_or_b32 v103, %[t], 3\n\ts_nop 4\n\t"
                             "buffer_store_dwordx4 v[100:103], %[vo], %[rs], 0 offen\n\t"
                             "v_mov_b32 v100, %[j]\n\tv_mov_b32 v101, %[j]\n\tv_mov_b32 v102, %[j]\n\tv_mov_b32 v103, %[j]\n\t"
                             :: [t] "v"(tag), [vo] "v"(vo), [rs] "s"(rsrc), [j] "v"(junk) : "v100", "v101", "v102", "v103", "memory");
            } else {
                asm volatile("v_or_b32 v100, %[t], 0\n\tv_or_b32 v101, %[t], 1\n\tv_or_b32 v102, %[t], 2\n\tv_or_b32 v103, %[t], 3\n\ts_nop 4\n\t"
                             "buffer_store_dwordx4 v[100:103], %[vo], %[rs], %[so] offen\n\t"
                             ".if %[variant] == 1\n\ts_nop 0\n\t.endif\n\t"
                             ".if %[variant] == 2\n\ts_waitcnt expcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t.endif\n\t"
                             "v_mov_b32 v100, %[j]\n\tv_mov_b32 v101, %[j]\n\tv_mov_b32 v102, %[j]\n\tv_mov_b32 v103, %[j]\n\t"
                             :: [t] "v"(tag), [vo] "v"(voff), [rs] "s"(rsrc), [so] "s"(soff), [j] "v"(junk), [variant] "n"(VARIANT)
                             : "v100", "v101", "v102", "v103", "memory");
            }
        }
}

__global__ __launch_bounds__(256) void pressure(const u32x4 *src, u32x4 *dst, size_t n, int passes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < passes; ++k)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i] + (unsigned)k;
}

template <int VARIANT>
static void run(const char *what, unsigned *d_out, std::vector<unsigned> &h, int waves, int ppw, hipStream_t s0, hipStream_t s1, const u32x4 *src, u32x4 *dst, size_t ncopy) {
    for (int beside = 0; beside < 2; ++beside) {
        size_t bad_pieces = 0, bad_words = 0, pieces = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipMemsetAsync(d_out, 0, h.size() * 4, s0));
            CHECK(hipStreamSynchronize(s0));
            if (beside) hipLaunchKernelGGL(pressure, dim3(1024), dim3(256), 0, s1, src, dst, ncopy, 4);   // half the wave slots: the victim runs beside it
            hipLaunchKernelGGL(victim<VARIANT>, dim3(waves / 4), dim3(256), 0, s0, d_out, ppw, 4);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
            for (size_t q = 0; q < h.size(); q += 4) {
                int w = 0;
                for (int j = 0; j < 4; ++j) w += (h[q + j] >> 16) == 0xDEADu;
                bad_pieces += w != 0; bad_words += w; ++pieces;
            }
        }
        printf("%-58s %-28s %9zu of %zu 16-byte pieces hold the overwriting value (%zu words)\n", what, beside ? "beside a streaming copy" : "alone", bad_pieces, pieces, bad_words);
    }
}

int main() {
    const int waves = 256 * 8 * 2, ppw = 64;                              // two waves per SIMD slot of 8 per CU; 64 KB per wave
    std::vector<unsigned> h((size_t)waves * ppw * 64 * 4);
    unsigned *d_out; CHECK(hipMalloc(&d_out, h.size() * 4));
    const size_t ncopy = (size_t)1 << 26;                                 // 1 GiB read + 1 GiB written per pass
    u32x4 *src, *dst; CHECK(hipMalloc(&src, ncopy * 16)); CHECK(hipMalloc(&dst, ncopy * 16));
    CHECK(hipMemset(src, 1, ncopy * 16));
    hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    run<0>("SGPR soffset, no wait state (as hipcc emitted)", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    run<1>("SGPR soffset, s_nop 0", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    run<2>("SGPR soffset, expcnt(0) + 16 wait states (shipped)", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    run<3>("constant soffset, no wait state (hipcc would pad this)", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    run<4>("dwordx2 (8 bytes), SGPR soffset, no wait state", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    run<5>("dwordx3 (12 bytes), SGPR soffset, no wait state", d_out, h, waves, ppw, s0, s1, src, dst, ncopy);
    return 0;
}
