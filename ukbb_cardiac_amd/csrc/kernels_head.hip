// Fused FCN head for gfx950.  One kernel replaces, for the full-resolution
// part of build_FCN (reference common/network.py:201-229) and the prob / pred
// definition (common/train_network.py:198-199):
//
//   same_dim0 (1x1, 16->32, BN, ReLU)
//   transpose_upsample2d x2/x4/x8/x16 of the squeezed maps of levels 1..4
//   concat -> 160 channels                         (never materialised)
//   out0 1x1 160->64 BN ReLU, out1 1x1 64->64 BN ReLU, logits 1x1 64->n_class + bias
//   softmax, argmax
//
// Mapping: one wave owns 32 consecutive pixels (linear index over N*H*W); the
// pixel sits on the MFMA N dimension (lane & 31) and channels on M, so every
// 1x1 layer is D[cout][pixel] = W[cout][k] * X[k][pixel] with
// v_mfma_f32_32x32x2_f32.  The 32x32 result tile has its column (pixel) on the
// lane and its rows (channels) in the 16 accumulator registers, so after
// bias+ReLU the registers ARE the next layer's B operand (k-step r supplies
// rows rowmap(r,0) on lanes 0-31 and rowmap(r,1) on lanes 32-63); the weights
// are packed on the host in that k order.  Nothing goes through LDS.
//
// The bilinear "transposed conv" upsampling (network.py:138-167) is evaluated
// as its <= 2x2 non-zero taps per output pixel, gathered straight from the
// low-resolution maps (L2 resident), with TF's SAME crop offset and
// un-normalised borders: out[o] = sum_i x[i] * h[o + pb - i*f], pb = (f-1)/2
// (oracle/fcn_oracle.py transpose_upsample2d_separable).
#include "kernels.h"

namespace ukbb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ __forceinline__ constexpr int rowmap(int r, int g) {
    // row of a 32x32 f32 MFMA result held in register r by lane half g
    return (r & 3) + 8 * (r >> 2) + 4 * g;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void tap1d(int o, int l, int n_in, int &i0, float &w0, int &i1, float &w1) {
    const int f = 1 << l;
    const int t = o + ((f - 1) >> 1);
    i1 = t >> l;
    const int j1 = t & (f - 1);
    const float inv = 1.0f / (float)f;
    w1 = (float)(j1 + 1) * inv;
    w0 = (float)(f - 1 - j1) * inv;
    i0 = i1 - 1;
    if (i1 >= n_in) { i1 = n_in - 1; w1 = 0.f; }
    if (i0 < 0) { i0 = 0; w0 = 0.f; }
}

template <int NCLS>
__global__ __launch_bounds__(256) void fcn_head_kernel(const HeadArgs a) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 31, g = lane >> 5;
    const long long nblk = ((long long)a.N * a.H * a.W) >> 5;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwave = (long long)gridDim.x * 4;

    for (long long blk = wave0; blk < nblk; blk += nwave) {
        const long long q = blk * 32 + p;
        const int hw = a.H * a.W;
        const int n = (int)(q / hw);
        const int rem = (int)(q - (long long)n * hw);
        const int y = rem / a.W, x = rem - y * a.W;

        // ---- same_dim0: S[32][px] = Ws0[32][16] * conv0[16][px] -----------------
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
        {
            const float4 *xp = reinterpret_cast<const float4 *>(a.conv0 + q * 16 + 8 * g);
            const float4 x0 = xp[0], x1 = xp[1];
            const float xin[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float4 *wp = reinterpret_cast<const float4 *>(a.w_s0 + lane * 8);
            const float4 w0 = wp[0], w1 = wp[1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) S = MFMA32(wv[s], xin[s], S);
        }
        // bias + ReLU -> B operand of out0's level-0 slice
        f32x16 P0, P1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { P0[r] = 0.f; P1[r] = 0.f; }
        {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b = *reinterpret_cast<const float4 *>(a.b_s0 + 8 * j + 4 * g);
                S[4 * j + 0] = fmaxf(S[4 * j + 0] + b.x, 0.f);
                S[4 * j + 1] = fmaxf(S[4 * j + 1] + b.y, 0.f);
                S[4 * j + 2] = fmaxf(S[4 * j + 2] + b.z, 0.f);
                S[4 * j + 3] = fmaxf(S[4 * j + 3] + b.w, 0.f);
            }
            // w_o0 layout: [level][cb][quad q4][lane][4]  (k-steps 4*q4 .. 4*q4+3)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 wa = *reinterpret_cast<const float4 *>(a.w_o0 + (((0 * 2 + 0) * 4 + q4) * 64 + lane) * 4);
                const float4 wb = *reinterpret_cast<const float4 *>(a.w_o0 + (((0 * 2 + 1) * 4 + q4) * 64 + lane) * 4);
                P0 = MFMA32(wa.x, S[4 * q4 + 0], P0); P1 = MFMA32(wb.x, S[4 * q4 + 0], P1);
                P0 = MFMA32(wa.y, S[4 * q4 + 1], P0); P1 = MFMA32(wb.y, S[4 * q4 + 1], P1);
                P0 = MFMA32(wa.z, S[4 * q4 + 2], P0); P1 = MFMA32(wb.z, S[4 * q4 + 2], P1);
                P0 = MFMA32(wa.w, S[4 * q4 + 3], P0); P1 = MFMA32(wb.w, S[4 * q4 + 3], P1);
            }
        }
        // ---- levels 1..4: gather-upsample 32 channels, feed out0 ---------------
#pragma unroll
        for (int l = 1; l <= 4; ++l) {
            const int hl = a.H >> l, wl = a.W >> l;
            int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
            tap1d(y, l, hl, y0, wy0, y1, wy1);
            tap1d(x, l, wl, x0, wx0, x1, wx1);
            const float *base = a.sq[l - 1] + (size_t)n * hl * wl * 32 + 16 * g;
            const float4 *t00 = reinterpret_cast<const float4 *>(base + ((size_t)y0 * wl + x0) * 32);
            const float4 *t01 = reinterpret_cast<const float4 *>(base + ((size_t)y0 * wl + x1) * 32);
            const float4 *t10 = reinterpret_cast<const float4 *>(base + ((size_t)y1 * wl + x0) * 32);
            const float4 *t11 = reinterpret_cast<const float4 *>(base + ((size_t)y1 * wl + x1) * 32);
            const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
            float f[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v00 = t00[j], v01 = t01[j], v10 = t10[j], v11 = t11[j];
                f[4 * j + 0] = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
                f[4 * j + 1] = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
                f[4 * j + 2] = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
                f[4 * j + 3] = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 wa = *reinterpret_cast<const float4 *>(a.w_o0 + (((l * 2 + 0) * 4 + q4) * 64 + lane) * 4);
                const float4 wb = *reinterpret_cast<const float4 *>(a.w_o0 + (((l * 2 + 1) * 4 + q4) * 64 + lane) * 4);
                P0 = MFMA32(wa.x, f[4 * q4 + 0], P0); P1 = MFMA32(wb.x, f[4 * q4 + 0], P1);
                P0 = MFMA32(wa.y, f[4 * q4 + 1], P0); P1 = MFMA32(wb.y, f[4 * q4 + 1], P1);
                P0 = MFMA32(wa.z, f[4 * q4 + 2], P0); P1 = MFMA32(wb.z, f[4 * q4 + 2], P1);
                P0 = MFMA32(wa.w, f[4 * q4 + 3], P0); P1 = MFMA32(wb.w, f[4 * q4 + 3], P1);
            }
        }
        // ---- out0 bias + ReLU ; out1: Q[64][px] = W1[64][64] * X[64][px] ----------
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b0 = *reinterpret_cast<const float4 *>(a.b_o0 + 8 * j + 4 * g);
            const float4 b1 = *reinterpret_cast<const float4 *>(a.b_o0 + 32 + 8 * j + 4 * g);
            P0[4 * j + 0] = fmaxf(P0[4 * j + 0] + b0.x, 0.f); P1[4 * j + 0] = fmaxf(P1[4 * j + 0] + b1.x, 0.f);
            P0[4 * j + 1] = fmaxf(P0[4 * j + 1] + b0.y, 0.f); P1[4 * j + 1] = fmaxf(P1[4 * j + 1] + b1.y, 0.f);
            P0[4 * j + 2] = fmaxf(P0[4 * j + 2] + b0.z, 0.f); P1[4 * j + 2] = fmaxf(P1[4 * j + 2] + b1.z, 0.f);
            P0[4 * j + 3] = fmaxf(P0[4 * j + 3] + b0.w, 0.f); P1[4 * j + 3] = fmaxf(P1[4 * j + 3] + b1.w, 0.f);
        }
        f32x16 Q0, Q1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Q0[r] = 0.f; Q1[r] = 0.f; }
        // w_o1 layout: [kb][cb][q4][lane][4]
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 wa = *reinterpret_cast<const float4 *>(a.w_o1 + (((0 * 2 + 0) * 4 + q4) * 64 + lane) * 4);
            const float4 wb = *reinterpret_cast<const float4 *>(a.w_o1 + (((0 * 2 + 1) * 4 + q4) * 64 + lane) * 4);
            Q0 = MFMA32(wa.x, P0[4 * q4 + 0], Q0); Q1 = MFMA32(wb.x, P0[4 * q4 + 0], Q1);
            Q0 = MFMA32(wa.y, P0[4 * q4 + 1], Q0); Q1 = MFMA32(wb.y, P0[4 * q4 + 1], Q1);
            Q0 = MFMA32(wa.z, P0[4 * q4 + 2], Q0); Q1 = MFMA32(wb.z, P0[4 * q4 + 2], Q1);
            Q0 = MFMA32(wa.w, P0[4 * q4 + 3], Q0); Q1 = MFMA32(wb.w, P0[4 * q4 + 3], Q1);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 wa = *reinterpret_cast<const float4 *>(a.w_o1 + (((1 * 2 + 0) * 4 + q4) * 64 + lane) * 4);
            const float4 wb = *reinterpret_cast<const float4 *>(a.w_o1 + (((1 * 2 + 1) * 4 + q4) * 64 + lane) * 4);
            Q0 = MFMA32(wa.x, P1[4 * q4 + 0], Q0); Q1 = MFMA32(wb.x, P1[4 * q4 + 0], Q1);
            Q0 = MFMA32(wa.y, P1[4 * q4 + 1], Q0); Q1 = MFMA32(wb.y, P1[4 * q4 + 1], Q1);
            Q0 = MFMA32(wa.z, P1[4 * q4 + 2], Q0); Q1 = MFMA32(wb.z, P1[4 * q4 + 2], Q1);
            Q0 = MFMA32(wa.w, P1[4 * q4 + 3], Q0); Q1 = MFMA32(wb.w, P1[4 * q4 + 3], Q1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b0 = *reinterpret_cast<const float4 *>(a.b_o1 + 8 * j + 4 * g);
            const float4 b1 = *reinterpret_cast<const float4 *>(a.b_o1 + 32 + 8 * j + 4 * g);
            Q0[4 * j + 0] = fmaxf(Q0[4 * j + 0] + b0.x, 0.f); Q1[4 * j + 0] = fmaxf(Q1[4 * j + 0] + b1.x, 0.f);
            Q0[4 * j + 1] = fmaxf(Q0[4 * j + 1] + b0.y, 0.f); Q1[4 * j + 1] = fmaxf(Q1[4 * j + 1] + b1.y, 0.f);
            Q0[4 * j + 2] = fmaxf(Q0[4 * j + 2] + b0.z, 0.f); Q1[4 * j + 2] = fmaxf(Q1[4 * j + 2] + b1.z, 0.f);
            Q0[4 * j + 3] = fmaxf(Q0[4 * j + 3] + b0.w, 0.f); Q1[4 * j + 3] = fmaxf(Q1[4 * j + 3] + b1.w, 0.f);
        }
        // ---- logits on the vector ALU: each lane holds 32 of the 64 channels ------
        // w_lg layout: [g][c][32] with index cb*16 + r  <->  channel cb*32 + rowmap(r, g)
        float lg[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const float4 *wp = reinterpret_cast<const float4 *>(a.w_lg + (g * NCLS + c) * 32);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 w = wp[j];
                s = fmaf(w.x, Q0[4 * j + 0], s); s = fmaf(w.y, Q0[4 * j + 1], s);
                s = fmaf(w.z, Q0[4 * j + 2], s); s = fmaf(w.w, Q0[4 * j + 3], s);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 w = wp[4 + j];
                s = fmaf(w.x, Q1[4 * j + 0], s); s = fmaf(w.y, Q1[4 * j + 1], s);
                s = fmaf(w.z, Q1[4 * j + 2], s); s = fmaf(w.w, Q1[4 * j + 3], s);
            }
            // both halves compute (lower + upper) in the SAME order -> identical bits
            const float other = __shfl_xor(s, 32);
            lg[c] = (g == 0 ? s + other : other + s) + a.b_lg[c];
        }
        if (g == 0) {
            if (a.logits) {
#pragma unroll
                for (int c = 0; c < NCLS; ++c) a.logits[q * NCLS + c] = lg[c];
            }
            int best = 0; float m = lg[0];
#pragma unroll
            for (int c = 1; c < NCLS; ++c) if (lg[c] > m) { m = lg[c]; best = c; }
            if (a.pred) a.pred[q] = best;
            if (a.prob) {
                float e[NCLS]; float sum = 0.f;
#pragma unroll
                for (int c = 0; c < NCLS; ++c) { e[c] = expf(lg[c] - m); sum += e[c]; }
                const float inv = 1.0f / sum;
#pragma unroll
                for (int c = 0; c < NCLS; ++c) a.prob[q * NCLS + c] = e[c] * inv;
            }
        }
    }
}

hipError_t launch_head(const HeadArgs &a, hipStream_t s) {
    const long long npix = (long long)a.N * a.H * a.W;
    if (npix % 32) return hipErrorInvalidValue;
    const long long nblk = npix / 32;
    long long wg = (nblk + 3) / 4;
    if (wg > 256 * 8) wg = 256 * 8;
    dim3 grid((unsigned)wg), block(256);
    switch (a.n_class) {
        case 2: hipLaunchKernelGGL(fcn_head_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(fcn_head_kernel<3>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(fcn_head_kernel<4>, grid, block, 0, s, a); break;
        case 5: hipLaunchKernelGGL(fcn_head_kernel<5>, grid, block, 0, s, a); break;
        case 6: hipLaunchKernelGGL(fcn_head_kernel<6>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- host-side weight packers (k order documented at the top) ---------------
void pack_head_s0(const float *w, float *dst) {
    // dst[lane][s] = W[ci = 8*g + s][co = m],  lane = (g<<5)|m, W is [16][32]
    for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 8; ++s) dst[lane * 8 + s] = w[(8 * (lane >> 5) + s) * 32 + (lane & 31)];
}

void pack_head_o0(const float *w, float *dst) {
    // W is [160][64].  dst[level][cb][q4][lane][i], k-step s = 4*q4 + i:
    //   level 0 : ci = rowmap(s, g)              (B operand = same_dim0 accumulator)
    //   level l : ci = 32*l + 16*g + s            (B operand = gathered f[s])
    for (int l = 0; l < 5; ++l)
        for (int cb = 0; cb < 2; ++cb)
            for (int q4 = 0; q4 < 4; ++q4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int s = 4 * q4 + i, g = lane >> 5, m = lane & 31;
                        const int ci = (l == 0) ? rowmap(s, g) : 32 * l + 16 * g + s;
                        dst[((((l * 2 + cb) * 4 + q4) * 64 + lane) * 4) + i] = w[ci * 64 + cb * 32 + m];
                    }
}

void pack_head_o1(const float *w, float *dst) {
    // W is [64][64].  dst[kb][cb][q4][lane][i]: ci = 32*kb + rowmap(4*q4+i, g), co = 32*cb + m
    for (int kb = 0; kb < 2; ++kb)
        for (int cb = 0; cb < 2; ++cb)
            for (int q4 = 0; q4 < 4; ++q4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 4; ++i) {
                        const int s = 4 * q4 + i, g = lane >> 5, m = lane & 31;
                        dst[((((kb * 2 + cb) * 4 + q4) * 64 + lane) * 4) + i] = w[(32 * kb + rowmap(s, g)) * 64 + cb * 32 + m];
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
template <int C, int NCLS>
__global__ __launch_bounds__(256) void logits_kernel(const LogitsArgs a) {
    __shared__ float wl[C * NCLS + NCLS];
    for (int i = threadIdx.x; i < C * NCLS; i += 256) wl[i] = a.w[i];
    for (int i = threadIdx.x; i < NCLS; i += 256) wl[C * NCLS + i] = a.bias[i];
    __syncthreads();
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < a.npix; q += (int64_t)gridDim.x * 256) {
        float xin[C];
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(a.in + q * C + 4 * j);
            xin[4 * j] = v.x; xin[4 * j + 1] = v.y; xin[4 * j + 2] = v.z; xin[4 * j + 3] = v.w;
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
        int best = 0; float m = lg[0];
#pragma unroll
        for (int c = 1; c < NCLS; ++c) if (lg[c] > m) { m = lg[c]; best = c; }
        if (a.pred) a.pred[q] = best;
        if (a.prob) {
            float e[NCLS]; float sum = 0.f;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) { e[c] = expf(lg[c] - m); sum += e[c]; }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) a.prob[q * NCLS + c] = e[c] * inv;
        }
    }
}

hipError_t launch_logits(const LogitsArgs &a, hipStream_t s) {
    unsigned grid = (unsigned)((a.npix + 255) / 256);
    if (grid > 256u * 16u) grid = 256u * 16u;
    if (a.C == 16 && a.n_class == 3) hipLaunchKernelGGL((logits_kernel<16, 3>), dim3(grid), dim3(256), 0, s, a);
    else if (a.C == 16 && a.n_class == 2) hipLaunchKernelGGL((logits_kernel<16, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (a.C == 16 && a.n_class == 4) hipLaunchKernelGGL((logits_kernel<16, 4>), dim3(grid), dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ukbb
