// Fused FCN head for gfx950.  Two kernels replace the full-resolution part of
// build_FCN (reference common/network.py:201-229) and the prob / pred definition
// (common/train_network.py:198-199):
//
//   same_dim_l   1x1 C_l -> 32, BN, ReLU                       (network.py:203-204)
//   up_l         transpose_upsample2d x2/x4/x8/x16             (network.py:138-167,210-211)
//   concat       -> 160 channels                               (network.py:218)
//   out0         1x1 160 -> 64, BN, ReLU ; out1 1x1 64 -> 64, BN, ReLU ; logits 1x1 64 -> n_class + bias
//   softmax, argmax
//
// Algebra used: out0's pre-activation is  W0 * concat_l(up_l(s_l)) = sum_l up_l(W0_l * s_l),
// because the bilinear "transposed conv" acts per channel and linearly, and a 1x1 conv acts
// per pixel and linearly, so they commute (borders included: both sides apply the same,
// possibly un-normalised, tap weights).  Levels 1..4 are therefore projected to 64
// channels at LOW resolution by sqg_kernel (G_l = W0_l * relu(BN(Ws_l * x_l))), and the
// full-resolution kernel only gathers the <= 2x2 taps of G_l and adds them.  This removes
// 128 of the 160 input channels of out0 at full resolution (-300 M of 603 M MAC per slice)
// and the 160-channel concat / the four upsampled maps are never materialised.
//
// Common mapping: one wave owns 32 pixels; the pixel sits on the MFMA N dimension
// (lane & 31), channels on M, so every 1x1 layer is D[cout][px] = W[cout][k] X[k][px] with
// v_mfma_f32_32x32x2_f32.  The 32x32 result has its column (pixel) on the lane and its rows
// (channels) in the 16 accumulator registers, so after bias + ReLU the registers ARE the next
// layer's B operand: k-step r supplies rows rowmap(r,0) on lanes 0-31 and rowmap(r,1) on
// lanes 32-63, and the weights are packed on the host in that k order.
#include "kernels.h"

#include <cstdio>

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace ukbb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ constexpr int rowmap(int r, int g) {
    // row of a 32x32 f32 MFMA result held in register r by lane half g
    return (r & 3) + 8 * (r >> 2) + 4 * g;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef UKBB_NO_PACKED_F32
// A/B form (r06): pairs of scalar v_fma_f32 instead of v_pk_fma_f32 (build with -fno-slp-vectorize as well, or
// hipcc packs the gather's scalar FMAs again); tools/ab_packed.sh
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{s_fma(a[0], b[0], c[0]), s_fma(a[1], b[1], c[1])}; }
#else
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
#endif
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ldg4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// Accumulator tile pre-loaded with the bias of its 16 rows: acc[4j+i] = bias[8j + 4g + i].
// Used as the C operand of the first MFMA of a chain, so the bias add costs no VALU op.
__device__ __forceinline__ f32x16 bias_tile(const float *bias, int g) {
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b = ldg4(bias + 8 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * j + i] = b[i];
    }
    return acc;
}

// ReLU as one integer max: for IEEE floats max_i32(bits(x), 0) is x for x >= +0 and +0 for anything
// with the sign bit set.  fmaxf() on an MFMA result compiles to two instructions (a canonicalising
// max and the max), and on gfx950 every VALU instruction takes issue time away from the fp32 MFMA
// stream (tools/mfma_coissue.hip).  Deliberately not inline asm: the compiler must see a VALU read of
// the accumulator to insert the MFMA -> VALU wait states.
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ void relu16(f32x16 &acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = relu1(acc[r]);
}

// D0/D1 (two Cout blocks of 32) += W[64][32 rows of X] * X, X given as an accumulator tile.
// wp: packed [cb][q4][lane][4]
__device__ __forceinline__ void chain_32to64(const float *wp, int lane, const f32x16 &X, f32x16 &D0, f32x16 &D1) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 wa = ldg4(wp + ((0 * 4 + q4) * 64 + lane) * 4);
        const f32x4 wb = ldg4(wp + ((1 * 4 + q4) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            D0 = MFMA32(wa[i], X[4 * q4 + i], D0);
            D1 = MFMA32(wb[i], X[4 * q4 + i], D1);
        }
    }
}

// D (one Cout block of 32) += W[32 couts][32 rows of X] * X.  wp: packed [q4][lane][4] of that block.
__device__ __forceinline__ void chain_32to32(const float *wp, int lane, const f32x16 &X, f32x16 &D) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 w = ldg4(wp + (q4 * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) D = MFMA32(w[i], X[4 * q4 + i], D);
    }
}

// ---------------------------------------------------------------------------
// sqg_kernel: G[px][64] = W0_l (64x32) * relu(Ws (32xCIN) * x[px] + bs)   at low resolution.
// One wave per 32 consecutive pixels of the flattened [N*H_l*W_l] map; no LDS.
// ---------------------------------------------------------------------------
template <int CIN>
__device__ __forceinline__ void sqg_body(const SqgArgs &a, int vblock, int vgrid) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, g = lane >> 5;
    const long long nblk = (a.npix + 31) >> 5;
    for (long long blk = (long long)vblock * 4 + (threadIdx.x >> 6); blk < nblk; blk += (long long)vgrid * 4) {
        const long long q = blk * 32 + p;
        const bool valid = q < a.npix;
        const float *xp = a.x + (valid ? q : a.npix - 1) * CIN + 4 * g;
        f32x16 S = bias_tile(a.b_s, g);
        // k-step (j,i) pairs channels 8j+i (lanes 0-31) and 8j+4+i (lanes 32-63).
        // Levels 3-4 are small (1-2 blocks per wave): all of the block's feature loads are issued before
        // the first MFMA so their latency is paid once, not once per group of k-steps.
        // (at most 128 channels at a time: inside sqg_multi_kernel this body must not push the register count past the
        // two-waves-per-SIMD limit, or level 1's bandwidth-bound stream loses half of its loads in flight)
        constexpr int XG = CIN / 8 < 16 ? CIN / 8 : 16;
#pragma unroll
        for (int j0 = 0; j0 < CIN / 8; j0 += XG) {
            f32x4 xv[XG];
#pragma unroll
            for (int j = 0; j < XG; ++j) xv[j] = ldg4(xp + 8 * (j0 + j));
#pragma unroll
            for (int j = 0; j < XG; ++j) {
                const f32x4 wv = ldg4(a.w_s + ((j0 + j) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) S = MFMA32(wv[i], xv[j][i], S);
            }
        }
        relu16(S);
        f32x16 G0, G1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { G0[r] = 0.f; G1[r] = 0.f; }
        chain_32to64(a.w_g, lane, S, G0, G1);
        if (valid) {
            float *o = a.out + q * 64 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v0, v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = G0[4 * j + i]; v1[i] = G1[4 * j + i]; }
                *reinterpret_cast<f32x4 *>(o + 8 * j) = v0;
                *reinterpret_cast<f32x4 *>(o + 32 + 8 * j) = v1;
            }
        }
    }
}

// Levels 1 and 2 (C_in = 32, 64) are big enough to be bandwidth-bound (level 1 of the batch-64
// workload: 82 MB in, 164 MB out against 27 us of MFMA time), so this variant keeps both weight
// matrices and the bias tile in registers and requests the features of the wave's next 32-pixel
// block before it starts the MFMA chains of the current one: loads, MFMAs and stores of different
// blocks overlap inside a wave instead of relying on occupancy alone.
template <int CIN>
__global__ __launch_bounds__(256) void sqg_kernel(const SqgArgs a) { sqg_body<CIN>(a, blockIdx.x, gridDim.x); }

// TR: the last layer is computed transposed (G^T = S^T W0^T: the same registers with the MFMA operands swapped), which puts
// the 32 output channels of a block on the lanes and the pixels on the registers: every store instruction then writes two
// contiguous 128-byte segments (one accumulator register as it stands) instead of 32 segments of 32 bytes.
// LT: the features of a block (32 pixels x CIN floats, one contiguous chunk of the NHWC map) are fetched with fully contiguous
// 1-KB wave loads and turned into the operand layout (pixel on the lane) through a wave-private LDS tile, instead of loads
// that touch 32 rows with 32 bytes each.
constexpr int SQG_LDS_WAVE = 32 * (64 + 4);
template <int CIN, bool TR = false, bool LT = false>
__device__ __forceinline__ void sqg_stream_body(const SqgArgs &a, int vblock, int vgrid, float *lds_x = nullptr) {
    constexpr int NJ = CIN / 8;
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, g = lane >> 5;
    const long long nblk = (a.npix + 31) >> 5;
    const long long step = (long long)vgrid * 4;
    f32x4 ws[NJ], wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) ws[j] = ldg4(a.w_s + (j * 64 + lane) * 4);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        wa[q4] = ldg4(a.w_g + ((0 * 4 + q4) * 64 + lane) * 4);
        wb[q4] = ldg4(a.w_g + ((1 * 4 + q4) * 64 + lane) * 4);
    }
    const f32x16 bias = bias_tile(a.b_s, g);
    constexpr int RS = CIN + 4;                         // LDS row stride (floats): b128-aligned, the strided read is conflict-free
    float *const xl = LT ? lds_x + (threadIdx.x >> 6) * SQG_LDS_WAVE : nullptr;
    const long long qlast = a.npix * (CIN / 4) - 1;     // last valid 16-byte quad of the map (rows past the end are never stored)
    auto fetch = [&](long long blk, f32x4 *dst) {
        if constexpr (LT) {
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const long long f = blk * (32 * (CIN / 4)) + t * 64 + lane;
                dst[t] = ldg4(a.x + 4 * (f < qlast ? f : qlast));
            }
        } else {
            const long long q = blk * 32 + p;
            const float *xp = a.x + (q < a.npix ? q : a.npix - 1) * CIN + 4 * g;
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[j] = ldg4(xp + 8 * j);
        }
    };
    long long blk = (long long)vblock * 4 + (threadIdx.x >> 6);
    f32x4 xn[NJ];
    if (blk < nblk) fetch(blk, xn);
    for (; blk < nblk; blk += step) {
        f32x4 xv[NJ];
        if constexpr (LT) {
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                const int f = t * 64 + lane;
                *reinterpret_cast<f32x4 *>(xl + (f / (CIN / 4)) * RS + (f % (CIN / 4)) * 4) = xn[t];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) xv[j] = *reinterpret_cast<const f32x4 *>(xl + p * RS + (2 * j + g) * 4);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) xv[j] = xn[j];
        }
        if (blk + step < nblk) fetch(blk + step, xn);
        f32x16 S = bias;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) S = MFMA32(ws[j][i], xv[j][i], S);
        relu16(S);
        f32x16 G0, G1;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (TR) {
            G0 = MFMA32(S[0], wa[0][0], zero);
            G1 = MFMA32(S[0], wb[0][0], zero);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int i = (q4 == 0 ? 1 : 0); i < 4; ++i) {
                    G0 = MFMA32(S[4 * q4 + i], wa[q4][i], G0);
                    G1 = MFMA32(S[4 * q4 + i], wb[q4][i], G1);
                }
            // register r of G0/G1: pixel 8*(r/4) + 4*g + r%4 of the block, channel p (+32)
            const long long q0 = blk * 32 + 4 * g;
            float *o = a.out + q0 * 64 + p;
            if (blk * 32 + 32 <= a.npix) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[(8 * (r / 4) + r % 4) * 64] = G0[r];
                    o[(8 * (r / 4) + r % 4) * 64 + 32] = G1[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (q0 + 8 * (r / 4) + r % 4 < a.npix) {
                        o[(8 * (r / 4) + r % 4) * 64] = G0[r];
                        o[(8 * (r / 4) + r % 4) * 64 + 32] = G1[r];
                    }
            }
            continue;
        }
        G0 = MFMA32(wa[0][0], S[0], zero);              // C = 0 as an inline constant: no accumulator clearing
        G1 = MFMA32(wb[0][0], S[0], zero);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int i = (q4 == 0 ? 1 : 0); i < 4; ++i) {
                G0 = MFMA32(wa[q4][i], S[4 * q4 + i], G0);
                G1 = MFMA32(wb[q4][i], S[4 * q4 + i], G1);
            }
        const long long q = blk * 32 + p;
        if (q < a.npix) {
            float *o = a.out + q * 64 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v0, v1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = G0[4 * j + i]; v1[i] = G1[4 * j + i]; }
                *reinterpret_cast<f32x4 *>(o + 8 * j) = v0;
                *reinterpret_cast<f32x4 *>(o + 32 + 8 * j) = v1;
            }
        }
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void sqg_stream_kernel(const SqgArgs a) { sqg_stream_body<CIN>(a, blockIdx.x, gridDim.x); }

// Levels 1-4 in ONE launch (C_in = 32, 64, 128, 256).  Levels 2-4 together are a tenth of level 1's work and as
// launches of their own were dominated by launch ramp and tail (13-25 us each for 2-9 us of work); level 1 is a pure
// HBM stream (82 MB in, 164 MB out at batch 64).  r02: every launch of this network costs ~10 us of fixed time
// (dispatch + first-load latency + tail, measured as 2 t(N=64) - t(N=128) per kernel), so the four go out together after
// level 4: the small levels' blocks come first and finish inside the first microseconds of level 1's stream.
template <bool TR, bool LT>
__global__ __launch_bounds__(256, 2) void sqg_multi_kernel(const SqgMultiArgs m) {
    __shared__ __attribute__((aligned(16))) float lds_x[LT ? 4 * SQG_LDS_WAVE : 4];
    const int b = blockIdx.x;
    const int e1 = m.nb[1], e2 = e1 + m.nb[2], e3 = e2 + m.nb[3];
    if (b < e1) sqg_stream_body<64, TR, LT>(m.lv[1], b, m.nb[1], lds_x);
    else if (b < e2) sqg_body<128>(m.lv[2], b - e1, m.nb[2]);
    else if (b < e3) sqg_body<256>(m.lv[3], b - e2, m.nb[3]);
    else sqg_stream_body<32, TR, LT>(m.lv[0], b - e3, m.nb[0], lds_x);
}

hipError_t launch_sqg_multi(const SqgArgs lv[4], hipStream_t s) {
    if (lv[0].cin != 32 || lv[1].cin != 64 || lv[2].cin != 128 || lv[3].cin != 256) return hipErrorInvalidValue;
    SqgMultiArgs m;
    int total = 0;
    for (int i = 0; i < 4; ++i) {
        m.lv[i] = lv[i];
        long long wg = ((lv[i].npix + 31) / 32 + 3) / 4;
        if (wg > 2048) wg = 2048;
        m.nb[i] = (int)wg;
        total += m.nb[i];
    }
    static const bool tr = [] { const char *e = getenv("UKBB_SQG_TR"); return e ? atoi(e) != 0 : true; }();
    static const bool lt = [] { const char *e = getenv("UKBB_SQG_LT"); return e ? atoi(e) != 0 : true; }();
    if (tr && lt) hipLaunchKernelGGL((sqg_multi_kernel<true, true>), dim3((unsigned)total), dim3(256), 0, s, m);
    else if (tr) hipLaunchKernelGGL((sqg_multi_kernel<true, false>), dim3((unsigned)total), dim3(256), 0, s, m);
    else hipLaunchKernelGGL((sqg_multi_kernel<false, false>), dim3((unsigned)total), dim3(256), 0, s, m);
    return hipGetLastError();
}

hipError_t launch_sqg(const SqgArgs &a, hipStream_t s) {
    const long long nblk = (a.npix + 31) / 32;
    long long wg = (nblk + 3) / 4;
    if (wg > 256 * 8) wg = 256 * 8;
    dim3 grid((unsigned)wg), block(256);
    switch (a.cin) {
        case 32: hipLaunchKernelGGL(sqg_stream_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(sqg_stream_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(sqg_kernel<128>, grid, block, 0, s, a); break;
        case 256: hipLaunchKernelGGL(sqg_kernel<256>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fcn_head_kernel: one workgroup = one 16x16 pixel tile (H, W are multiples of 16) =
// 8 blocks of 32 pixels (two tile rows each), two blocks per wave.
//
// LDS holds the low-resolution windows of G_1..G_4 this tile's bilinear taps touch:
// for level l (factor f = 2^l, pb = (f-1)/2) output row y reads source rows
// i1 = (y+pb)>>l and i1-1, so a 16-row tile origin y0 needs rows (y0>>l)-1 .. (y0>>l)+((15+pb)>>l):
// 9, 6, 4, 3 rows (and columns) for l = 1..4, i.e. 81+36+16+9 = 142 source pixels x 64 ch.
// Rows/columns outside the map are stored as zeros, which is exactly the reference's
// un-normalised border (the tap is dropped).  Pixel stride is 68 floats
// so the ds_read_b128 of 16 lanes with different source pixels hit distinct bank quads.
// ---------------------------------------------------------------------------
constexpr int HT = 16;                                 // tile edge
constexpr int GSTRIDE = 68;                            // floats per staged source pixel
__host__ __device__ constexpr int win_n(int l) { return l == 1 ? 9 : l == 2 ? 6 : l == 3 ? 4 : 3; }
__host__ __device__ constexpr int win_base(int l) { return l == 1 ? 0 : l == 2 ? 81 : l == 3 ? 117 : 133; }
constexpr int GPIX = 142;
constexpr int HEAD_LDS_FLOATS = GPIX * GSTRIDE;

template <int NCLS, int OCC>
__global__ __launch_bounds__(256, OCC) void fcn_head_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gl[];   // [GPIX][GSTRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, g = lane >> 5;
    const int tiles_x = a.W / HT, tiles_y = a.H / HT;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int y0 = ty * HT, x0 = tx * HT;

    // ---- stage the G windows (global NHWC -> LDS), 16 float4 per source pixel ---------
    {
        constexpr int NF4 = GPIX * 16, NIT = (NF4 + 255) / 256;
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int sp = idx >> 4, c4 = idx & 15;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (sp < GPIX) {
                const int l = sp < 81 ? 1 : sp < 117 ? 2 : sp < 133 ? 3 : 4;
                const int wn_ = win_n(l), rel = sp - win_base(l);
                const int ry = rel / wn_, rx = rel - ry * wn_;
                const int hl = a.H >> l, wl = a.W >> l;
                const int sy = (y0 >> l) - 1 + ry, sx = (x0 >> l) - 1 + rx;
                if ((unsigned)sy < (unsigned)hl && (unsigned)sx < (unsigned)wl)
                    t = ldg4(a.G[l - 1] + (((size_t)n * hl + sy) * wl + sx) * 64 + 4 * c4);
            }
            v[it] = t;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int sp = idx >> 4, c4 = idx & 15;
            if (sp < GPIX) *reinterpret_cast<f32x4 *>(gl + sp * GSTRIDE + 4 * c4) = v[it];
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int bb = 0; bb < 2; ++bb) {
        const int blk = wave * 2 + bb;                        // tile rows 2*blk, 2*blk+1
        const int yl = 2 * blk + (p >> 4), xl = p & 15;
        const int y = y0 + yl, x = x0 + xl;
        const size_t q = ((size_t)n * a.H + y) * a.W + x;

        // ---- out0 = bias + sum_l up_l(G_l) [<= 2x2 taps per level, from LDS] + W0_0[64][32] * S -------
        f32x16 P0 = bias_tile(a.b_o0, g), P1 = bias_tile(a.b_o0 + 32, g);
#pragma unroll
        for (int l = 1; l <= 4; ++l) {
            const int f = 1 << l, pb = (f - 1) >> 1;
            const float inv = 1.0f / (float)f;
            const int tyy = yl + pb, txx = xl + pb;           // tile-relative: (y0 >> l) cancels
            const int ry1 = (tyy >> l) + 1, rx1 = (txx >> l) + 1;   // window index (row 0 = source row (y0>>l)-1)
            const int jy = tyy & (f - 1), jx = txx & (f - 1);
            const float wy1 = (float)(jy + 1) * inv, wy0 = (float)(f - 1 - jy) * inv;
            const float wx1 = (float)(jx + 1) * inv, wx0 = (float)(f - 1 - jx) * inv;
            const int wn_ = win_n(l);
            const float *b11 = gl + (win_base(l) + ry1 * wn_ + rx1) * GSTRIDE + 4 * g;
            const float *b10 = b11 - GSTRIDE, *b01 = b11 - wn_ * GSTRIDE, *b00 = b01 - GSTRIDE;
            const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a00 = *reinterpret_cast<const f32x4 *>(b00 + 8 * j), a01 = *reinterpret_cast<const f32x4 *>(b01 + 8 * j);
                const f32x4 a10 = *reinterpret_cast<const f32x4 *>(b10 + 8 * j), a11 = *reinterpret_cast<const f32x4 *>(b11 + 8 * j);
                const f32x4 c00 = *reinterpret_cast<const f32x4 *>(b00 + 32 + 8 * j), c01 = *reinterpret_cast<const f32x4 *>(b01 + 32 + 8 * j);
                const f32x4 c10 = *reinterpret_cast<const f32x4 *>(b10 + 32 + 8 * j), c11 = *reinterpret_cast<const f32x4 *>(b11 + 32 + 8 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    P0[4 * j + i] = fmaf(w00, a00[i], fmaf(w01, a01[i], fmaf(w10, a10[i], fmaf(w11, a11[i], P0[4 * j + i]))));
                    P1[4 * j + i] = fmaf(w00, c00[i], fmaf(w01, c01[i], fmaf(w10, c10[i], fmaf(w11, c11[i], P1[4 * j + i]))));
                }
            }
        }
        // ---- same_dim0: S[32][px] = Ws0[32][16] * conv0[16][px] -----------------------------
        f32x16 S = bias_tile(a.b_s0, g);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 xv = ldg4(a.conv0 + q * 16 + 8 * j + 4 * g);
            const f32x4 wv = ldg4(a.w_s0 + (j * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) S = MFMA32(wv[i], xv[i], S);
        }
        relu16(S);
        chain_32to64(a.w_o0, lane, S, P0, P1);
        relu16(P0);
        relu16(P1);
        // ---- out1: Q[64][px] = b1 + W1[64][64] * X[64][px] ------------------------------------------------
        f32x16 Q0 = bias_tile(a.b_o1, g), Q1 = bias_tile(a.b_o1 + 32, g);
        chain_32to64(a.w_o1, lane, P0, Q0, Q1);               // k rows 0..31
        chain_32to64(a.w_o1 + 2 * 4 * 64 * 4, lane, P1, Q0, Q1);   // k rows 32..63
        relu16(Q0);
        relu16(Q1);
        // ---- logits on the vector ALU: each lane holds 32 of the 64 channels --------------------------
        // w_lg layout: [g][c][32] with index cb*16 + r  <->  channel cb*32 + rowmap(r, g)
        float lg[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const float *wp = a.w_lg + (g * NCLS + c) * 32;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w = ldg4(wp + 4 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i) s = fmaf(w[i], Q0[4 * j + i], s);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 w = ldg4(wp + 16 + 4 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i) s = fmaf(w[i], Q1[4 * j + i], s);
            }
            // both halves form (lower + upper) in the SAME order -> identical bits
            const float other = __shfl_xor(s, 32);
            lg[c] = (g == 0 ? s + other : other + s) + a.b_lg[c];
        }
        if (g == 0) {
            if (a.logits) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c) a.logits[q * NCLS + c] = lg[c];
            }
            float pr[NCLS];
            const int best = softmax_argmax_opt<NCLS>(lg, a.prob != nullptr, pr);
            if (a.pred) a.pred[q] = best;
            if (a.prob) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c) a.prob[q * NCLS + c] = pr[c];
            }
        }
    }
}

#ifdef UKBB_DIAG
__device__ unsigned long long g_hstamps[8];
#endif
// ---------------------------------------------------------------------------
// Producer/consumer head (768 threads, persistent over 16x16 tiles, one workgroup per CU):
//   waves 4-11 "producers": stage the G windows of upcoming tiles (global -> registers -> LDS,
//       double buffered per tile) and evaluate, for the 32-pixel block their partner wave will
//       process next, out0's pre-activation  b0 + sum_l up_l(G_l)  (all VALU + LDS work), which
//       they hand over through LDS as a ready accumulator tile (8 float4 per lane, laid out
//       [slot][lane] so both sides touch consecutive 16-byte words);
//   waves 0-3 "consumers": same_dim0 (8 MFMA) -> out0 level-0 slice accumulated ONTO the handed
//       tile (32 MFMA) -> ReLU -> out1 (64 MFMA) -> ReLU -> logits / softmax / argmax.
// A stage = one block per consumer wave (2 stages per tile); one barrier per stage.
// r01 measurements behind this split: in the single-role kernel the gather (VALU+LDS), the G
// staging and the MFMA chains simply added up (ablation: 76 + 63 + 40%..) because every wave
// ran them back to back and at most 3 waves fit per SIMD.
// ---------------------------------------------------------------------------
constexpr int PX_WAVE = 8 * 64 * 4;                    // floats of one handed-over tile (64 ch x 32 px)
// LDS map of fcn_head_pc_kernel (floats)
constexpr int L_GW = 0;                                // [2][GPIX][GSTRIDE]  G windows, double buffered per tile
constexpr int L_PX = L_GW + 2 * GPIX * GSTRIDE;        // [4][PX_WAVE]        hand-over tiles (single buffered)
// Weight part of the map: fp32 fragments, or (X3) out0's level-0 slice and out1 as three bf16 pieces per weight
template <bool X3> struct HeadLds {
    static constexpr int WS0 = L_PX + 4 * PX_WAVE;             // pack_sq(same_dim0)                                   512
    static constexpr int WO0 = WS0 + 512;                      // pack_rowmap(out0 rows 0..31) 2048 | pack_head_x3(.., 32) 3072 dwords
    static constexpr int WO1 = WO0 + (X3 ? 3072 : 2048);       // pack_rowmap(out1) x2         4096 | pack_head_x3(.., 64) 6144 dwords
    static constexpr int BS0 = WO1 + (X3 ? 6144 : 4096);       // 32
    static constexpr int BO0 = BS0 + 32;                       // 64
    static constexpr int BO1 = BO0 + 64;                       // 64
    static constexpr int WLG = BO1 + 64;                       // [2][NCLS][32] <= 384
    static constexpr int BLG = WLG + 384;                      // <= 8
    static constexpr int FLOATS = BLG + 8;
};
constexpr int HEADPC_LDS_FLOATS = HeadLds<false>::FLOATS;
constexpr int HEADPC_X3_LDS_FLOATS = HeadLds<true>::FLOATS;
static_assert(HEADPC_X3_LDS_FLOATS * 4 <= 160 * 1024, "head kernel LDS");

__device__ __forceinline__ f32x4 lds4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_n(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_n<N, I + 1>(f); }
}

__device__ __forceinline__ f32x16 bias_tile_lds(const float *bias, int g) {
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b = lds4(bias + 8 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * j + i] = b[i];
    }
    return acc;
}

__device__ __forceinline__ void chain_32to64_lds(const float *wp, int lane, const f32x16 &X, f32x16 &D0, f32x16 &D1) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 wa = lds4(wp + ((0 * 4 + q4) * 64 + lane) * 4);
        const f32x4 wb = lds4(wp + ((1 * 4 + q4) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            D0 = MFMA32(wa[i], X[4 * q4 + i], D0);
            D1 = MFMA32(wb[i], X[4 * q4 + i], D1);
        }
    }
}

// ---- X3 experiment: an fp32-exact product from bf16 pieces on the dense matrix cores ---------------------------------------
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = x - h - m (all exact in fp32); W likewise on the host.  The six partial
// products that matter (hh, hm, mh, mm, hl, lh) run as v_mfma_f32_32x32x16_bf16 with fp32 accumulation; what is dropped is
// below 2^-24 of |x||w| per product (profiles/r02_notes.md section 11).
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;       // RNE; v_cvt_pk_bf16_f32
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = pk_bf16(r0, r1);
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = pk_bf16(q0, q1);
}
__device__ __forceinline__ f32x16 mfma_b16(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, a), __builtin_bit_cast(hbf16x8, b), c, 0, 0, 0);
}

template <int NCLS, bool X3 = false>
__global__ __launch_bounds__(768) void fcn_head_pc_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using LM = HeadLds<X3>;
    constexpr int L_WS0 = LM::WS0, L_WO0 = LM::WO0, L_WO1 = LM::WO1, L_BS0 = LM::BS0, L_BO0 = LM::BO0, L_BO1 = LM::BO1, L_WLG = LM::WLG, L_BLG = LM::BLG;
    float *gw = lds + L_GW;
    float *px = lds + L_PX;
    const int tiles_x = a.W / HT, tiles_y = a.H / HT;
    const int ntiles = a.N * tiles_y * tiles_x;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstages = 2 * my_tiles;
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    const int lane = threadIdx.x & 63, p = lane & 31, g = lane >> 5;

    // ---- every weight / bias of the head lives in LDS for the lifetime of the workgroup ----
    {
        const int t = threadIdx.x;
        auto copy = [&](int dst, const float *src, int nfloat) {
            for (int i = t; i < nfloat; i += 768) lds[dst + i] = src[i];
        };
        copy(L_WS0, a.w_s0, 512);
        if constexpr (X3) { copy(L_WO0, a.w_o0x3, 3072); copy(L_WO1, a.w_o1x3, 6144); }
        else              { copy(L_WO0, a.w_o0, 2048);   copy(L_WO1, a.w_o1, 4096); }
        copy(L_BS0, a.b_s0, 32);  copy(L_BO0, a.b_o0, 64);   copy(L_BO1, a.b_o1, 64);
        copy(L_WLG, a.w_lg, 2 * NCLS * 32); copy(L_BLG, a.b_lg, NCLS);
    }

    if (producer) {
        // 8 producer waves: waves 4-7 build channels 0..31 of the hand-over tile of consumer wave pw,
        // waves 8-11 channels 32..63 (the gather is the longest per-stage job; halving it per wave keeps
        // the producers ahead of the MFMA waves).  All 512 producer threads share the window staging.
        const int tid = threadIdx.x - 256, pw = (tid >> 6) & 3, half = tid >> 8;
        // Staging index space: every iteration (512 threads = 32 source pixels x 16 float4) belongs to ONE
        // level, so level, window width and the row/column split are compile-time per iteration:
        // l=1: 81 px -> 3 iterations, l=2: 36 -> 2, l=3: 16 -> 1, l=4: 9 -> 1.
        constexpr int NIT = 7;
        u32x4 gv[NIT];
        const int sp32 = tid >> 4, c4 = tid & 15;
        // Every VALU instruction of this role takes issue time from the consumers' MFMA stream (they
        // share the SIMDs), so the staging keeps its per-tile arithmetic on the scalar unit: per-thread
        // window coordinates and byte offsets are computed once, validity is one and/compare against a
        // per-tile row|column mask, and taps outside the map are buffer loads with an out-of-range
        // offset (the hardware returns 0, border taps are dropped).
#ifndef UKBB_HEAD_RECOMPUTE_WINDOW_CONSTS
#define UKBB_HEAD_RECOMPUTE_WINDOW_CONSTS 0
#endif
        // r05 experiment (-DUKBB_HEAD_RECOMPUTE_WINDOW_CONSTS=1): the 14 per-thread window constants recomputed inside g_load (once per
        // tile) instead of living in registers across the stage loop -- removes the 2 spilled VGPRs of the 168-register budget
        unsigned tbit[NIT], goff[NIT];
        auto window_consts = [&]() {
        int sp32 = tid >> 4;                            // shadows the outer one: opaque to the optimiser in the recompute form, so that it is not hoisted back
        if (UKBB_HEAD_RECOMPUTE_WINDOW_CONSTS) asm volatile("" : "+v"(sp32));
        unroll_n<NIT>([&](auto ic) {
            constexpr int it = decltype(ic)::value;
            constexpr int l = it < 3 ? 1 : it < 5 ? 2 : it < 6 ? 3 : 4;
            constexpr int it0 = l == 1 ? 0 : l == 2 ? 3 : l == 3 ? 5 : 6;
            constexpr int wn_ = win_n(l);
            const int rel = (it - it0) * 32 + sp32;
            const int ry = rel / wn_, rx = rel - ry * wn_;
            tbit[it] = rel < wn_ * wn_ ? (1u << ry) | (1u << (16 + rx)) : 0x80000000u;   // bit 31 never set in a tile mask
            goff[it] = (unsigned)((ry * (a.W >> l) + rx) * 64 + 4 * c4) * 4u;
        });
        };
        if (!UKBB_HEAD_RECOMPUTE_WINDOW_CONSTS) window_consts();
        auto g_load = [&](int k) {                      // G windows of this workgroup's k-th tile -> registers
            if (UKBB_HEAD_RECOMPUTE_WINDOW_CONSTS) window_consts();
            int bid = blockIdx.x + k * gridDim.x;
            const int tx = bid % tiles_x; bid /= tiles_x;
            const int ty = bid % tiles_y;
            const int n = bid / tiles_y;
            const int y0 = ty * HT, x0 = tx * HT;
            __amdgpu_buffer_rsrc_t rs[4];
            unsigned cm[4];
#pragma unroll
            for (int l = 1; l <= 4; ++l) {
                const int hl = a.H >> l, wl = a.W >> l, wn_ = win_n(l);
                const int sy0 = (y0 >> l) - 1, sx0 = (x0 >> l) - 1;     // window origin in the level-l map
                const int ylo = sy0 < 0 ? -sy0 : 0, yhi = hl - sy0 < wn_ ? hl - sy0 : wn_;
                const int xlo = sx0 < 0 ? -sx0 : 0, xhi = wl - sx0 < wn_ ? wl - sx0 : wn_;
                cm[l - 1] = (((1u << yhi) - 1u) & ~((1u << ylo) - 1u)) | ((((1u << xhi) - 1u) & ~((1u << xlo) - 1u)) << 16);
                const float *base = a.G[l - 1] + ((long long)(n * hl + sy0) * wl + sx0) * 64;   // may precede the map; masked lanes never use it
                rs[l - 1] = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
            }
            unroll_n<NIT>([&](auto ic) {
                constexpr int it = decltype(ic)::value;
                constexpr int l = it < 3 ? 1 : it < 5 ? 2 : it < 6 ? 3 : 4;
                const unsigned vo = (cm[l - 1] & tbit[it]) == tbit[it] ? goff[it] : 0x80000000u;
                gv[it] = __builtin_amdgcn_raw_buffer_load_b128(rs[l - 1], vo, 0, 0);
            });
        };
        float *const gdst = gw + sp32 * GSTRIDE + 4 * c4;
        auto g_store = [&](int b) {
            float *dst = gdst + b * GPIX * GSTRIDE;
            unroll_n<NIT>([&](auto ic) {
                constexpr int it = decltype(ic)::value;
                constexpr int l = it < 3 ? 1 : it < 5 ? 2 : it < 6 ? 3 : 4;
                constexpr int it0 = l == 1 ? 0 : l == 2 ? 3 : l == 3 ? 5 : 6;
                const int rel = (it - it0) * 32 + sp32;
                if (rel < win_n(l) * win_n(l))
                    *reinterpret_cast<u32x4 *>(dst + (win_base(l) + (it - it0) * 32) * GSTRIDE) = gv[it];
            });
        };
        // Gather constants of this thread, for both block parities (a wave alternates between the two
        // 32-pixel blocks 2*pw and 2*pw+1 of a tile): LDS offset of the top-left tap and the four tap
        // weights per level.  Computed once; the stage loop is unrolled by four so the block parity and
        // the window buffer are compile-time and every LDS address is base register + immediate.
        int tap0[2][4];
        float tw[2][4][4];
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int blk = pw * 2 + par;
            const int yl = 2 * blk + (p >> 4), xl = p & 15;
#pragma unroll
            for (int l = 1; l <= 4; ++l) {
                const int f = 1 << l, pb = (f - 1) >> 1;
                const float inv = 1.0f / (float)f;
                const int tyy = yl + pb, txx = xl + pb;
                const int ry1 = (tyy >> l) + 1, rx1 = (txx >> l) + 1;
                const int jy = tyy & (f - 1), jx = txx & (f - 1);
                const float wy1 = (float)(jy + 1) * inv, wy0 = (float)(f - 1 - jy) * inv;
                const float wx1 = (float)(jx + 1) * inv, wx0 = (float)(f - 1 - jx) * inv;
                tap0[par][l - 1] = (win_base(l) + (ry1 - 1) * win_n(l) + (rx1 - 1)) * GSTRIDE + 4 * g + 32 * half;
                tw[par][l - 1][0] = wy0 * wx0; tw[par][l - 1][1] = wy0 * wx1;
                tw[par][l - 1][2] = wy1 * wx0; tw[par][l - 1][3] = wy1 * wx1;
            }
        }
        f32x16 P;
        auto gather = [&](auto parc, auto bufc) {       // block 2*pw + PAR of the tile staged in window buffer BUF -> P (32 channels)
            constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value;
            P = bias_tile_lds(lds + L_BO0 + 32 * half, g);
#ifdef UKBB_DIAG
            if (a.diag & 1) return;                 // ablation: no gather (hand over the bias tile only)
#endif
#pragma unroll
            for (int l = 1; l <= 4; ++l) {
                const int wn_ = win_n(l);
                const float *b00 = gw + BUF * GPIX * GSTRIDE + tap0[PAR][l - 1];
                const float *b01 = b00 + GSTRIDE, *b10 = b00 + wn_ * GSTRIDE, *b11 = b10 + GSTRIDE;
                const float w00 = tw[PAR][l - 1][0], w01 = tw[PAR][l - 1][1], w10 = tw[PAR][l - 1][2], w11 = tw[PAR][l - 1][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a00 = lds4(b00 + 8 * j), a01 = lds4(b01 + 8 * j), a10 = lds4(b10 + 8 * j), a11 = lds4(b11 + 8 * j);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        P[4 * j + i] = fmaf(w00, a00[i], fmaf(w01, a01[i], fmaf(w10, a10[i], fmaf(w11, a11[i], P[4 * j + i]))));
                }
            }
        };
        auto hand_over = [&]() {                        // P -> px[pw]  ([slot][lane] float4 image)
            float *dst = px + pw * PX_WAVE + (4 * half) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = P[4 * j + i];
                *reinterpret_cast<f32x4 *>(dst + j * 256) = v;
            }
        };
        // prologue: windows of tiles 0 and 1 into LDS
        if (my_tiles > 0) { g_load(0); g_store(0); }
        if (my_tiles > 1) { g_load(1); g_store(1); }
        __syncthreads();                                // barrier X: weights, GW[0], GW[1] visible
        auto stage = [&](auto parc, auto bufc, int s) {
            constexpr int PAR = decltype(parc)::value, BUF = decltype(bufc)::value;
            gather(parc, bufc);                         // LDS windows -> registers (consumers still busy with s-1)
            hand_over();                                // px was released at barrier B of stage s-1
            __syncthreads();                            // barrier A_s: px ready
            __syncthreads();                            // barrier B_s: px consumed
            if constexpr (PAR == 1) {                   // tile k = s>>1 fully gathered: its window buffer is free
                // Fetch the windows of tile k+2 into it now, synchronously: the producers are far ahead of the
                // MFMA waves (they would only wait at barrier A), and a prefetch held in registers across the
                // next two gathers pushed the kernel into scratch spills.
                const int k = s >> 1;
                if (k + 2 < my_tiles) { g_load(k + 2); g_store(BUF); }
            }
        };
        constexpr std::integral_constant<int, 0> I0{};
        constexpr std::integral_constant<int, 1> I1{};
#pragma unroll 1
        for (int s = 0; s < nstages; s += 4) {          // nstages is even
            stage(I0, I0, s);
            stage(I1, I0, s + 1);
            if (s + 2 < nstages) { stage(I0, I1, s + 2); stage(I1, I1, s + 3); }
        }
    } else {
        const int wave = threadIdx.x >> 6;
        const float *w_s0 = lds + L_WS0, *w_o0 = lds + L_WO0, *w_o1 = lds + L_WO1, *w_lg = lds + L_WLG;
        auto pixel_of = [&](int s) -> size_t {
            int bid = blockIdx.x + (s >> 1) * gridDim.x;
            const int tx = bid % tiles_x; bid /= tiles_x;
            const int ty = bid % tiles_y;
            const int n = bid / tiles_y;
            const int blk = wave * 2 + (s & 1);
            const int y = ty * HT + 2 * blk + (p >> 4), x = tx * HT + (p & 15);
            return ((size_t)n * a.H + y) * a.W + x;
        };
        f32x4 xv0 = {0.f, 0.f, 0.f, 0.f}, xv1 = xv0;      // conv0 features of the NEXT block, prefetched
        if (nstages > 0) {
            const size_t q0 = pixel_of(0);
            xv0 = ldg4(a.conv0 + q0 * 16 + 4 * g);
            xv1 = ldg4(a.conv0 + q0 * 16 + 8 + 4 * g);
        }
        __syncthreads();                                // barrier X
#ifdef UKBB_DIAG
        // diagnostic build, UKBB_HEAD_STAMPS=1: where an MFMA wave's stage goes -- wait at barrier A (the producers' hand-off), the LDS
        // read of the handed tile, wait at barrier B, everything else (MFMA chain + own vector work)
        unsigned long long hs_a = 0, hs_r = 0, hs_b = 0, hs_t0 = __builtin_amdgcn_s_memtime();
#define UKBB_HS(var, prev) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - prev; prev = t_; }
#else
#define UKBB_HS(var, prev)
#endif
#pragma unroll 1
        for (int s = 0; s < nstages; ++s) {
            const size_t q = pixel_of(s);
            const f32x4 x0 = xv0, x1 = xv1;
#ifdef UKBB_DIAG
            unsigned long long hs_p = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();                            // barrier A_s: px[wave] ready
            UKBB_HS(hs_a, hs_p)
            f32x16 P0, P1;
            {
                const float *src = px + wave * PX_WAVE + lane * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v0 = lds4(src + j * 256);
                    const f32x4 v1 = lds4(src + (4 + j) * 256);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { P0[4 * j + i] = v0[i]; P1[4 * j + i] = v1[i]; }
                }
            }
            UKBB_HS(hs_r, hs_p)
            __syncthreads();                            // barrier B_s: px may be overwritten
            UKBB_HS(hs_b, hs_p)
            if (s + 1 < nstages) {                      // features of the next block: a whole stage to arrive
                const size_t qn = pixel_of(s + 1);
                xv0 = ldg4(a.conv0 + qn * 16 + 4 * g);
                xv1 = ldg4(a.conv0 + qn * 16 + 8 + 4 * g);
            }
            // ---- same_dim0 ----
            f32x16 S = bias_tile_lds(lds + L_BS0, g);
            {
                const f32x4 wv0 = lds4(w_s0 + lane * 4), wv1 = lds4(w_s0 + (64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) S = MFMA32(wv0[i], x0[i], S);
#pragma unroll
                for (int i = 0; i < 4; ++i) S = MFMA32(wv1[i], x1[i], S);
            }
            relu16(S);
#ifdef UKBB_DIAG
            if (a.diag & 2) { relu16(P0); relu16(P1); if (a.pred) a.pred[q] = (int)P0[0]; continue; }   // ablation: no out0/out1/logits
#endif
            // ---- out0: handed tile (bias + upsampled levels 1..4) + W0_0 * S ----
            if constexpr (X3) {                         // out0's level-0 slice from bf16 pieces: K = 32 -> two chunks of 16
                u32x4 sh[2], sm[2], sl[2];
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        unsigned th, tm, tl;
                        split_pair(S[8 * kc + 2 * d], S[8 * kc + 2 * d + 1], th, tm, tl);
                        sh[kc][d] = th; sm[kc][d] = tm; sl[kc][d] = tl;
                    }
                const float *wx0 = lds + L_WO0 + lane * 4;
                auto A0 = [&](int t, int cb, int kc) { return __builtin_bit_cast(u32x4, lds4(wx0 + ((t * 2 + cb) * 2 + kc) * 256)); };
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const u32x4 ah0 = A0(0, 0, kc), ah1 = A0(0, 1, kc), am0 = A0(1, 0, kc), am1 = A0(1, 1, kc), al0 = A0(2, 0, kc), al1 = A0(2, 1, kc);
                    P0 = mfma_b16(al0, sh[kc], P0); P1 = mfma_b16(al1, sh[kc], P1);
                    P0 = mfma_b16(ah0, sl[kc], P0); P1 = mfma_b16(ah1, sl[kc], P1);
                    P0 = mfma_b16(am0, sm[kc], P0); P1 = mfma_b16(am1, sm[kc], P1);
                    P0 = mfma_b16(am0, sh[kc], P0); P1 = mfma_b16(am1, sh[kc], P1);
                    P0 = mfma_b16(ah0, sm[kc], P0); P1 = mfma_b16(ah1, sm[kc], P1);
                    P0 = mfma_b16(ah0, sh[kc], P0); P1 = mfma_b16(ah1, sh[kc], P1);
                }
            } else {
                chain_32to64_lds(w_o0, lane, S, P0, P1);
            }
            relu16(P0);
            relu16(P1);
            // ---- out1 ----
            f32x16 Q0 = bias_tile_lds(lds + L_BO1, g), Q1 = bias_tile_lds(lds + L_BO1 + 32, g);
            if constexpr (X3) {
                // B operands: lane (pixel p, half g) supplies k slots e = 0..7 of chunk kc = registers 8 (kc & 1) + e of P0 (kc < 2) / P1.
                // The split of chunk kc+1 (~60 vector instructions) is interleaved with the 12 MFMAs of chunk kc: a bf16 MFMA leaves
                // about six vector-issue slots free inside the issuing wave (tools/mfma_bf16_coissue.hip).
                u32x4 bh[4], bm[4], bl[4];
                auto split = [&](auto kcc) {
                    constexpr int kc = decltype(kcc)::value;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        constexpr int r0 = 8 * (kc & 1);
                        const float x0 = kc < 2 ? P0[r0 + 2 * d] : P1[r0 + 2 * d], x1 = kc < 2 ? P0[r0 + 2 * d + 1] : P1[r0 + 2 * d + 1];
                        unsigned th, tm, tl;
                        split_pair(x0, x1, th, tm, tl);
                        bh[kc][d] = th; bm[kc][d] = tm; bl[kc][d] = tl;
                    }
                };
                const float *wx = lds + L_WO1 + lane * 4;
                auto A = [&](int t, int cb, int kc) { return __builtin_bit_cast(u32x4, lds4(wx + ((t * 2 + cb) * 4 + kc) * 256)); };
                split(std::integral_constant<int, 0>{});
                unroll_n<4>([&](auto kcc) {
                    constexpr int kc = decltype(kcc)::value;
                    const u32x4 ah0 = A(0, 0, kc), ah1 = A(0, 1, kc), am0 = A(1, 0, kc), am1 = A(1, 1, kc), al0 = A(2, 0, kc), al1 = A(2, 1, kc);
                    if constexpr (kc < 3) split(std::integral_constant<int, kc + 1>{});
                    Q0 = mfma_b16(al0, bh[kc], Q0); Q1 = mfma_b16(al1, bh[kc], Q1);      // small terms first
                    Q0 = mfma_b16(ah0, bl[kc], Q0); Q1 = mfma_b16(ah1, bl[kc], Q1);
                    Q0 = mfma_b16(am0, bm[kc], Q0); Q1 = mfma_b16(am1, bm[kc], Q1);
                    Q0 = mfma_b16(am0, bh[kc], Q0); Q1 = mfma_b16(am1, bh[kc], Q1);
                    Q0 = mfma_b16(ah0, bm[kc], Q0); Q1 = mfma_b16(ah1, bm[kc], Q1);
                    Q0 = mfma_b16(ah0, bh[kc], Q0); Q1 = mfma_b16(ah1, bh[kc], Q1);
                    if constexpr (kc < 3) {
#pragma unroll
                        for (int i = 0; i < 12; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five vector instructions of the next chunk's split
                        }
                    }
                });
            } else {
                chain_32to64_lds(w_o1, lane, P0, Q0, Q1);
                chain_32to64_lds(w_o1 + 2 * 4 * 64 * 4, lane, P1, Q0, Q1);
            }
            relu16(Q0);
            relu16(Q1);
#ifdef UKBB_DIAG
            if (a.diag & 4) { if (a.pred) a.pred[q] = (int)(Q0[0] + Q1[0]); continue; }   // ablation: no logits / softmax VALU
#endif
            // ---- logits / softmax / argmax ----
            // Packed over channel pairs (weights and activations are both consecutive registers), so
            // each class costs 16 v_pk_fma_f32 + 1 add and no operand shuffling.
            float lg[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const float *wp = w_lg + (g * NCLS + c) * 32;
                f32x2 acc2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w = lds4(wp + 4 * j);
                    acc2 = pk_fma(f32x2{w[0], w[1]}, f32x2{Q0[4 * j], Q0[4 * j + 1]}, acc2);
                    acc2 = pk_fma(f32x2{w[2], w[3]}, f32x2{Q0[4 * j + 2], Q0[4 * j + 3]}, acc2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w = lds4(wp + 16 + 4 * j);
                    acc2 = pk_fma(f32x2{w[0], w[1]}, f32x2{Q1[4 * j], Q1[4 * j + 1]}, acc2);
                    acc2 = pk_fma(f32x2{w[2], w[3]}, f32x2{Q1[4 * j + 2], Q1[4 * j + 3]}, acc2);
                }
                const float acc = acc2[0] + acc2[1];
                const float other = __shfl_xor(acc, 32);
                lg[c] = (g == 0 ? acc + other : other + acc) + lds[L_BLG + c];
            }
            if (g == 0) {
                if (a.logits) {
#pragma unroll
                    for (int c = 0; c < NCLS; ++c) a.logits[q * NCLS + c] = lg[c];
                }
                float pr[NCLS];
                const int best = softmax_argmax_opt<NCLS>(lg, a.prob != nullptr, pr);
                if (a.pred) a.pred[q] = best;
                if (a.prob) {
#pragma unroll
                    for (int c = 0; c < NCLS; ++c) a.prob[q * NCLS + c] = pr[c];
                }
            }
        }
#ifdef UKBB_DIAG
        if (lane == 0 && (a.diag & 64)) {
            atomicAdd(g_hstamps + 0, hs_a); atomicAdd(g_hstamps + 1, hs_r); atomicAdd(g_hstamps + 2, hs_b);
            atomicAdd(g_hstamps + 3, __builtin_amdgcn_s_memtime() - hs_t0); atomicAdd(g_hstamps + 4, (unsigned long long)nstages);
        }
#endif
    }
}

static hipError_t launch_head_pc(const HeadArgs &a, hipStream_t s) {
    const int n_cu = device_cu_count();
    const int ntiles = a.N * (a.H / HT) * (a.W / HT);
    dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(768);
    static const bool x3_env = [] { const char *e = getenv("UKBB_HEAD_X3"); return e && atoi(e) != 0; }();   // A/B knob
    if ((a.x3 || x3_env) && a.w_o1x3 && a.w_o0x3) {     // UKBB_PREC_F32X3
        const size_t ldsx = HEADPC_X3_LDS_FLOATS * sizeof(float);
#define UKBB_HEADX3_CASE(NC)                                                                          \
    case NC: {                                                                                       \
        auto k = fcn_head_pc_kernel<NC, true>;                                                       \
        static OncePerDevice lds_ok;                                                                 \
        {                                                                                            \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), (int)ldsx);    \
            if (e != hipSuccess) return e;                                                           \
        }                                                                                            \
        hipLaunchKernelGGL(k, grid, block, ldsx, s, a);                                              \
        break;                                                                                       \
    }
        switch (a.n_class) {
            UKBB_HEADX3_CASE(2) UKBB_HEADX3_CASE(3) UKBB_HEADX3_CASE(4) UKBB_HEADX3_CASE(5) UKBB_HEADX3_CASE(6)
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const size_t lds = HEADPC_LDS_FLOATS * sizeof(float);
#define UKBB_HEADPC_CASE(NC)                                                                          \
    case NC: {                                                                                       \
        auto k = fcn_head_pc_kernel<NC>;                                                             \
        static OncePerDevice lds_ok;                                                                 \
        {                                                                                            \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), (int)lds);    \
            if (e != hipSuccess) return e;                                                           \
        }                                                                                            \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                               \
        break;                                                                                       \
    }
    switch (a.n_class) {
        UKBB_HEADPC_CASE(2) UKBB_HEADPC_CASE(3) UKBB_HEADPC_CASE(4) UKBB_HEADPC_CASE(5) UKBB_HEADPC_CASE(6)
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int OCC>
static hipError_t launch_head_occ(const HeadArgs &a, hipStream_t s) {
    dim3 grid((unsigned)(a.N * (a.H / HT) * (a.W / HT))), block(256);
    const size_t lds = HEAD_LDS_FLOATS * sizeof(float);
    switch (a.n_class) {
        case 2: hipLaunchKernelGGL((fcn_head_kernel<2, OCC>), grid, block, lds, s, a); break;
        case 3: hipLaunchKernelGGL((fcn_head_kernel<3, OCC>), grid, block, lds, s, a); break;
        case 4: hipLaunchKernelGGL((fcn_head_kernel<4, OCC>), grid, block, lds, s, a); break;
        case 5: hipLaunchKernelGGL((fcn_head_kernel<5, OCC>), grid, block, lds, s, a); break;
        case 6: hipLaunchKernelGGL((fcn_head_kernel<6, OCC>), grid, block, lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_head(const HeadArgs &a_in, hipStream_t s) {
    HeadArgs a = a_in;
#ifdef UKBB_DIAG
    { const char *e = getenv("UKBB_HEAD_DIAG"); a.diag = e ? atoi(e) : 0; }
#endif
    if ((a.H % HT) || (a.W % HT)) return hipErrorInvalidValue;
    static const int use_pc = [] { const char *e = getenv("UKBB_HEAD_PC"); return e ? atoi(e) : 1; }();   // A/B knob
#ifdef UKBB_DIAG
    if (use_pc && getenv("UKBB_HEAD_STAMPS")) {        // the 6th stamped launch reports the MFMA waves' stage budget
        static int shots = 0;
        unsigned long long z[8] = {0};
        a.diag |= 64;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hstamps), z, 64);
        const hipError_t e = launch_head_pc(a, s);
        (void)hipStreamSynchronize(s);
        if (++shots == 6) {
            unsigned long long h[8];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hstamps), 64);
            const double st = (double)h[4];
            if (st > 0) fprintf(stderr, "[head stamps] per MFMA wave and stage (32-pixel block): wait at barrier A (hand-off) %.0f, read of the handed tile %.0f, wait at barrier B %.0f, "
                                        "MFMA chain + own vector work %.0f; total %.0f cycles\n", h[0] / st, h[1] / st, h[2] / st, (h[3] - h[0] - h[1] - h[2]) / st, h[3] / st);
        }
        return e;
    }
#endif
    if (use_pc) return launch_head_pc(a, s);
    const HeadArgs &b = a;
    static const int occ = [] { const char *e = getenv("UKBB_HEAD_OCC"); return e ? atoi(e) : 3; }();   // tuning knob
    if (occ == 2) return launch_head_occ<2>(b, s);
    if (occ == 4) return launch_head_occ<4>(b, s);
    return launch_head_occ<3>(b, s);
}

// ---- host-side weight packers (k order documented at the top) ---------------
void pack_sq(const float *w, int cin, float *dst) {
    // W is [cin][32].  dst[j][lane][i] = W[ci = 8j + 4g + i][co = m],  lane = (g<<5)|m
    for (int j = 0; j < cin / 8; ++j)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i)
                dst[(j * 64 + lane) * 4 + i] = w[(8 * j + 4 * (lane >> 5) + i) * 32 + (lane & 31)];
}

void pack_rowmap_32x64(const float *w, int ld, float *dst) {
    // W is 32 rows (k) x 64 cols (cout), row stride ld.  dst[cb][q4][lane][i]:
    //   k = rowmap(4*q4 + i, g), co = 32*cb + m   (B operand = a 32-row accumulator tile)
    for (int cb = 0; cb < 2; ++cb)
        for (int q4 = 0; q4 < 4; ++q4)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int g = lane >> 5, m = lane & 31;
                    dst[(((cb * 4 + q4) * 64 + lane) * 4) + i] = w[rowmap(4 * q4 + i, g) * ld + cb * 32 + m];
                }
}

static inline unsigned short host_bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float host_bf16_to_f32(unsigned short b) { const unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

void pack_head_x3(const float *w, int k_in, float *dst) {
    // W is [k_in rows (32 or 64)][64 out], row stride 64.  dst[t][cb][kc][lane][d] (dwords), kc < k_in / 16: t = piece (bf16 of w, of
    // the remainder, of the rest), lane = (g << 5) | m, cout = 32 cb + m, k slot e = 2 d (+1 in the high half) <-> input row
    // 32 (kc / 2) + rowmap(8 (kc & 1) + e, g): the registers of the 32-row accumulator tile(s) the MFMA wave holds
    unsigned *o = reinterpret_cast<unsigned *>(dst);
    const int nkc = k_in / 16;
    for (int t = 0; t < 3; ++t)
        for (int cb = 0; cb < 2; ++cb)
            for (int kc = 0; kc < nkc; ++kc)
                for (int lane = 0; lane < 64; ++lane)
                    for (int d = 0; d < 4; ++d) {
                        const int g = lane >> 5, m = lane & 31;
                        unsigned short piece[2];
                        for (int hh = 0; hh < 2; ++hh) {
                            const int e = 2 * d + hh, ch = 32 * (kc / 2) + rowmap(8 * (kc & 1) + e, g);
                            float x = w[ch * 64 + 32 * cb + m];
                            unsigned short b = 0;
                            for (int k = 0; k <= t; ++k) { b = host_bf16_rne(x); x -= host_bf16_to_f32(b); }
                            piece[hh] = b;
                        }
                        o[((((t * 2 + cb) * nkc + kc) * 64 + lane) * 4) + d] = (unsigned)piece[0] | ((unsigned)piece[1] << 16);
                    }
}

void pack_head_lg(const float *w, int n_class, float *dst) {
    // W is [64][n_class].  dst[g][c][cb*16 + r] = W[cb*32 + rowmap(r, g)][c]
    for (int g = 0; g < 2; ++g)
        for (int c = 0; c < n_class; ++c)
            for (int cb = 0; cb < 2; ++cb)
                for (int r = 0; r < 16; ++r)
                    dst[(g * n_class + c) * 32 + cb * 16 + r] = w[(cb * 32 + rowmap(r, g)) * n_class + c];
}

// ---------------------------------------------------------------------------
// U-Net logits: 1x1 conv C -> n_class + bias, softmax / argmax
// (reference common/network_ao.py:63,159-160).  C = 16: 48 MAC per pixel,
// bandwidth-bound; one thread per pixel on the vector ALU.
// ---------------------------------------------------------------------------
template <int C, int NCLS, bool IBF = false>
__global__ __launch_bounds__(256) void logits_kernel(const LogitsArgs a) {
    __shared__ float wl[C * NCLS + NCLS];
    for (int i = threadIdx.x; i < C * NCLS; i += 256) wl[i] = a.w[i];
    for (int i = threadIdx.x; i < NCLS; i += 256) wl[C * NCLS + i] = a.bias[i];
    __syncthreads();
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < a.npix; q += (int64_t)gridDim.x * 256) {
        float xin[C];
        if constexpr (IBF) {                                  // bf16 -> f32 is a 16-bit shift
            const unsigned short *ib = reinterpret_cast<const unsigned short *>(a.in) + q * C;
#pragma unroll
            for (int j = 0; j < C / 8; ++j) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ib + 8 * j);
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xin[8 * j + 2 * k] = __builtin_bit_cast(float, u[k] << 16);
                    xin[8 * j + 2 * k + 1] = __builtin_bit_cast(float, u[k] & 0xffff0000u);
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(a.in + q * C + 4 * j);
            xin[4 * j] = v.x; xin[4 * j + 1] = v.y; xin[4 * j + 2] = v.z; xin[4 * j + 3] = v.w;
        }
        }
        float lg[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < C; ++k) s = fmaf(xin[k], wl[k * NCLS + c], s);
            lg[c] = s + wl[C * NCLS + c];
        }
        if (a.logits) {
#pragma unroll
            for (int c = 0; c < NCLS; ++c) a.logits[q * NCLS + c] = lg[c];
        }
        float pr[NCLS];
        const int best = softmax_argmax_opt<NCLS>(lg, a.prob != nullptr, pr);
        if (a.pred) a.pred[q] = best;
        if (a.prob) {
#pragma unroll
            for (int c = 0; c < NCLS; ++c) a.prob[q * NCLS + c] = pr[c];
        }
    }
}

hipError_t launch_logits(const LogitsArgs &a, hipStream_t s) {
    unsigned grid = (unsigned)((a.npix + 255) / 256);
    if (grid > 256u * 16u) grid = 256u * 16u;
    if (a.in_bf16) {
        if (a.C == 16 && a.n_class == 3) hipLaunchKernelGGL((logits_kernel<16, 3, true>), dim3(grid), dim3(256), 0, s, a);
        else if (a.C == 16 && a.n_class == 2) hipLaunchKernelGGL((logits_kernel<16, 2, true>), dim3(grid), dim3(256), 0, s, a);
        else if (a.C == 16 && a.n_class == 4) hipLaunchKernelGGL((logits_kernel<16, 4, true>), dim3(grid), dim3(256), 0, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (a.C == 16 && a.n_class == 3) hipLaunchKernelGGL((logits_kernel<16, 3>), dim3(grid), dim3(256), 0, s, a);
    else if (a.C == 16 && a.n_class == 2) hipLaunchKernelGGL((logits_kernel<16, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (a.C == 16 && a.n_class == 4) hipLaunchKernelGGL((logits_kernel<16, 4>), dim3(grid), dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ukbb
