// Stand-alone placement test of the fused U-Net tail (kernels_tail.hip) with identity filters: logits[c] must equal the input
// channel that the chain of centre taps routes to it.  Build: hipcc --offload-arch=gfx950 -I ukbb_cardiac_amd/csrc tools/test_tail.cpp
// -L ukbb_cardiac_amd -lukbb_fcn -o tools/_bin/test_tail ; run on the GPU box with LD_LIBRARY_PATH=ukbb_cardiac_amd.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "kernels.h"
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
int main(int argc, char **argv) {
    const int N = 2, H = argc > 1 ? atoi(argv[1]) : 64, W = argc > 2 ? atoi(argv[2]) : 96, mode = argc > 3 ? atoi(argv[3]) : 0;
    // mode 0: up0_0 centre tap routes skip channel c -> c; up0_1 centre tap c -> c; logits class m = channel m
    // mode 1: up0_0 tap (kh=2,kw=2) of the UP source, up0_1 tap (0,0): tests shifts
    std::vector<float> w0(9 * 32 * 16, 0.f), w1(9 * 16 * 16, 0.f), b0(16, 0.f), b1(16, 0.f), lw(16 * 3, 0.f), lb(3, 0.f);
    const int t0 = mode == 0 ? 4 : 8, src = mode == 0 ? 0 : 16, t1 = mode == 0 ? 4 : 0;
    if (mode < 2) {
        for (int c = 0; c < 16; ++c) { w0[(t0 * 32 + src + c) * 16 + c] = 1.f; w1[(t1 * 16 + c) * 16 + c] = 1.f; }
    } else {                                                        // mode 2: sparse random +-1 filters over all taps and both sources (exact in bf16)
        unsigned rng = 12345u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (int co = 0; co < 16; ++co) {
            for (int k = 0; k < 4; ++k) w0[((rnd() % 9) * 32 + rnd() % 32) * 16 + co] = (rnd() & 1) ? 1.f : -1.f;
            for (int k = 0; k < 3; ++k) w1[((rnd() % 9) * 16 + rnd() % 16) * 16 + co] = (rnd() & 1) ? 1.f : -1.f;
            b0[co] = (float)(rnd() % 3); b1[co] = (float)(rnd() % 2);
        }
    }
    for (int m = 0; m < 3; ++m) lw[(m + 5) * 3 + m] = 1.f;          // class m = channel m + 5
    std::vector<unsigned short> in0((size_t)N * H * W * 16), in1(in0.size());
    auto val = [&](int s, int n, int y, int x, int c) {
        if (mode == 2) return (float)(((x * 7 + y * 13 + c * 5 + s * 3 + n) % 4));                     // 0..3
        return (float)((x % 16) + 16 * (y % 8) + (c == 5 ? 0 : c == 6 ? 1 : 2) + 3 * s); };   // < 256: exact in bf16
    for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int c = 0; c < 16; ++c) {
        const size_t i = (((size_t)n * H + y) * W + x) * 16 + c;
        in0[i] = bf(val(0, n, y, x, c)); in1[i] = bf(val(1, n, y, x, c));
    }
    std::vector<float> p0(9 * 64 * 4), p1(5 * 64 * 4);
    ukbb::pack_tail_weights(w0.data(), w1.data(), p0.data(), p1.data());
    void *d_in0, *d_in1, *d_p0, *d_p1, *d_b0, *d_b1, *d_lw, *d_lb, *d_lg, *d_pred;
    auto up = [&](void **d, const void *h, size_t b) { hipMalloc(d, b); hipMemcpy(*d, h, b, hipMemcpyHostToDevice); };
    up(&d_in0, in0.data(), in0.size() * 2); up(&d_in1, in1.data(), in1.size() * 2); up(&d_p0, p0.data(), p0.size() * 4); up(&d_p1, p1.data(), p1.size() * 4);
    up(&d_b0, b0.data(), 64); up(&d_b1, b1.data(), 64); up(&d_lw, lw.data(), lw.size() * 4); up(&d_lb, lb.data(), 12);
    hipMalloc(&d_lg, (size_t)N * H * W * 3 * 4); hipMalloc(&d_pred, (size_t)N * H * W * 4);
    hipMemset(d_lg, 0xff, (size_t)N * H * W * 3 * 4);
    ukbb::TailArgs a{};
    a.in0 = (const float *)d_in0; a.in1 = (const float *)d_in1; a.wA0 = (const float *)d_p0; a.wA1 = (const float *)d_p1; a.b0 = (const float *)d_b0; a.b1 = (const float *)d_b1;
    a.lg_w = (const float *)d_lw; a.lg_b = (const float *)d_lb; a.logits = (float *)d_lg; a.pred = (int32_t *)d_pred; a.N = N; a.H = H; a.W = W; a.ncls = 3;
    hipError_t e = ukbb::launch_unet_tail(a, 0);
    hipDeviceSynchronize();
    printf("launch: %s / %s\n", hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    std::vector<float> lg((size_t)N * H * W * 3);
    hipMemcpy(lg.data(), d_lg, lg.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    std::vector<float> ref;
    if (mode == 2) {                                                // CPU reference of the chain (all values small integers: exact)
        std::vector<float> mid((size_t)N * H * W * 16), o1((size_t)N * H * W * 16);
        for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < 16; ++co) {
            float acc = b0[co];
            for (int t = 0; t < 9; ++t) { const int yy = y + t / 3 - 1, xx = x + t % 3 - 1; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                for (int ci = 0; ci < 32; ++ci) { const float w = w0[(t * 32 + ci) * 16 + co]; if (w != 0.f) acc += w * val(ci >> 4, n, yy, xx, ci & 15); } }
            mid[(((size_t)n * H + y) * W + x) * 16 + co] = acc > 0.f ? acc : 0.f;
        }
        for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < 16; ++co) {
            float acc = b1[co];
            for (int t = 0; t < 9; ++t) { const int yy = y + t / 3 - 1, xx = x + t % 3 - 1; if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                for (int ci = 0; ci < 16; ++ci) { const float w = w1[(t * 16 + ci) * 16 + co]; if (w != 0.f) acc += w * mid[(((size_t)n * H + yy) * W + xx) * 16 + ci]; } }
            o1[(((size_t)n * H + y) * W + x) * 16 + co] = acc > 0.f ? acc : 0.f;
        }
        ref.resize((size_t)N * H * W * 3);
        for (size_t px = 0; px < (size_t)N * H * W; ++px) for (int m = 0; m < 3; ++m) ref[px * 3 + m] = o1[px * 16 + m + 5];
    }
    for (int n = 0; n < N; ++n) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int m = 0; m < 3; ++m) {
        // mode 0: logits = skip[y][x][m+5]; mode 1: mid[y][x] = up[y+1][x+1] (0 outside); out[y][x] = mid[y-1][x-1] = up[y][x] where (y-1,x-1) inside
        float want;
        if (mode == 2) want = ref[(((size_t)n * H + y) * W + x) * 3 + m];
        else if (mode == 0) want = val(0, n, y, x, m + 5);
        else want = (y >= 1 && x >= 1) ? val(1, n, y, x, m + 5) : 0.f;
        const float got = lg[(((size_t)n * H + y) * W + x) * 3 + m];
        if (got != want) { if (bad < 40) printf("n %d y %d x %d m %d: got %g want %g\n", n, y, x, m, got, want); ++bad; }
    }
    printf("%ld mismatches of %zu\n", bad, lg.size());
    return bad != 0;
}
