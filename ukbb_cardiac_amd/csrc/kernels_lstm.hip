// BiConvLSTM head of the aortic UNet-LSTM model (reference common/network_ao.py:255-319 with
// tf.contrib.rnn.Conv2DLSTMCell semantics, SURVEY.md App. B.6 [TF-recall]).
//
// r05: the gate convolution AND the cell update run in one kernel (kernels_wino24.hip, ConvArgs::ls_mode): the x half of the gate
// conv (W_x * x_t + b, the same for every window a frame is part of) is evaluated once per frame and direction, each time step
// then convolves the 16 hidden channels only and forms  c' = sigmoid(f + 1) * c + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o)
// in the epilogue -- the 64-channel gate map never exists in HBM.  What is left for this file:
//   lstm_out_kernel   prob / pred / logits of one time step from the two directions' hidden maps: the 1x1 output conv over
//                     concat([h_fw, h_bw]) + bias, softmax, argmax (network_ao.py:305-312, 396-397);
//   lstm_tile_kernel  the same per (window, step) and, fused behind it, the weighted circular tiling of
//                     deploy_network_ao.py:176-183 in the reference's accumulation order and arithmetic (float32 accumulator
//                     updated through float64) -- the per-window probabilities are never stored.
#include "kernels.h"

namespace ukbb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NH = 16;                       // hidden channels (train_network_ao.py num_hidden = 16)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 16 hidden channels of one pixel, fp32 or bf16 in memory
template <bool BF>
__device__ __forceinline__ void load_h16(const float *base, long long elem, float (&v)[NH]) {
    if constexpr (BF) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned short *>(base) + elem);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 d = p[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned w = d[i];
                v[8 * q + 2 * i] = __builtin_bit_cast(float, w << 16);
                v[8 * q + 2 * i + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < NH / 4; ++q) {
            const f32x4 d = *reinterpret_cast<const f32x4 *>(base + elem + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * q + i] = d[i];
        }
    }
}

// logits of one pixel: W[0:16] . h_fw + W[16:32] . h_bw + b (each direction's sum formed on its own, ascending hidden channel)
template <int NCLS>
__device__ __forceinline__ void out_conv(const float (&vf)[NH], const float (&vb)[NH], const float (&w)[2 * NH][NCLS], const float (&b)[NCLS], float (&lg)[NCLS]) {
    float pf[NCLS], pb[NCLS];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) { pf[k] = 0.f; pb[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) { pf[k] = fmaf(vf[i], w[i][k], pf[k]); pb[k] = fmaf(vb[i], w[NH + i][k], pb[k]); }
#pragma unroll
    for (int k = 0; k < NCLS; ++k) lg[k] = (pf[k] + pb[k]) + b[k];
}

template <int NCLS>
__device__ __forceinline__ void load_out_weights(const float *w_out, const float *b_out, float (&w)[2 * NH][NCLS], float (&b)[NCLS]) {
#pragma unroll
    for (int i = 0; i < 2 * NH; ++i)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) w[i][k] = w_out[i * NCLS + k];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) b[k] = b_out[k];
}

// one thread = one pixel of one window
template <int NCLS, bool BF>
__global__ __launch_bounds__(256) void lstm_out_kernel(const LstmOutArgs a) {
    float w[2 * NH][NCLS], b[NCLS];
    load_out_weights<NCLS>(a.w_out, a.b_out, w, b);
    const long long total = (long long)a.M * a.HW;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long m = id / a.HW, pix = id - m * a.HW;
        const long long mf = a.mapf ? a.mapf[m] : m, mb = a.mapb ? a.mapb[m] : m;
        float lg[NCLS], pr[NCLS], vf[NH], vb[NH];
        load_h16<BF>(a.hf, (mf * a.HW + pix) * NH, vf);
        load_h16<BF>(a.hb, (mb * a.HW + pix) * NH, vb);
        out_conv<NCLS>(vf, vb, w, b, lg);
        const int best = softmax_argmax<NCLS>(lg, pr);
        float *po = a.prob + m * a.m_stride + pix * NCLS;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) po[k] = pr[k];
        if (a.logits) {
            float *lo = a.logits + m * a.m_stride + pix * NCLS;
#pragma unroll
            for (int k = 0; k < NCLS; ++k) lo[k] = lg[k];
        }
        if (a.pred) a.pred[m * (a.m_stride / NCLS) + pix] = best;
    }
}

// prob[f] = (sum over the <= K windows containing f, in the reference's order, of prob(window w, step k) * w_k) / wsum[f]
//   prob(w, k) = softmax(out_conv(h_fw[k][w], h_bw[k][w])) exactly as lstm_out_kernel forms it (forward_seq of that window);
//   a frame no window reaches (time_step > window) has wsum = 0: 0/0 = NaN and argmax 0, as numpy gives the reference
//   reference arithmetic: prob is float32, `prob[..., idx] += prob_idx * w` runs in float64 and is cast back
//   per addition; `prob /= weight` likewise (deploy_network_ao.py:176-183).
template <int NCLS, bool BF>
__global__ __launch_bounds__(256) void lstm_tile_kernel(const LstmTileArgs a) {
    float w[2 * NH][NCLS], b[NCLS];
    load_out_weights<NCLS>(a.w_out, a.b_out, w, b);
    const long long total = (long long)a.F * a.HW;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const int f = (int)(id / a.HW);
        const long long pix = id - (long long)f * a.HW;
        float acc[NCLS];
#pragma unroll
        for (int k = 0; k < NCLS; ++k) acc[k] = 0.f;
        for (int j = 0; j < a.K; ++j) {
            const int wk_ = a.order[f * a.K + j];
            if (wk_ < 0) break;                                             // fewer than K windows reach this frame (time_step > 1, F < K)
            const int wi = wk_ / a.K, k = wk_ - wi * a.K;
            // the first step of a direction comes from the per-FRAME maps of the x pass, later steps from that direction's per-window maps
            const long long mf = k == 0 ? a.map_first[wi] : wi, mb = k == a.K - 1 ? a.map_last[wi] : wi;
            float lg[NCLS], p[NCLS], vf[NH], vb[NH];
            load_h16<BF>(k == 0 ? a.h1f : a.hf, (k == 0 ? 0 : (long long)k * a.k_stride) + (mf * a.HW + pix) * NH, vf);
            load_h16<BF>(k == a.K - 1 ? a.h1b : a.hb, (k == a.K - 1 ? 0 : (long long)k * a.k_stride) + (mb * a.HW + pix) * NH, vb);
            out_conv<NCLS>(vf, vb, w, b, lg);
            (void)softmax_argmax<NCLS>(lg, p);
            const double wt = a.wk[k];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) acc[c] = (float)((double)acc[c] + (double)p[c] * wt);
        }
        const double ws = a.wsum[f];
        int best = 0;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) acc[c] = (float)((double)acc[c] / ws);
#pragma unroll
        for (int c = 1; c < NCLS; ++c) if (acc[c] > acc[best]) best = c;
        float *o = a.prob + id * NCLS;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) o[c] = acc[c];
        if (a.pred) a.pred[id] = best;
    }
}

}  // namespace

hipError_t launch_lstm_out(const LstmOutArgs &a, hipStream_t s) {
    const long long total = (long long)a.M * a.HW;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 g((unsigned)blocks), t(256);
    switch (a.n_class * 2 + (a.h_bf16 ? 1 : 0)) {
        case 4: hipLaunchKernelGGL((lstm_out_kernel<2, false>), g, t, 0, s, a); break;
        case 5: hipLaunchKernelGGL((lstm_out_kernel<2, true>), g, t, 0, s, a); break;
        case 6: hipLaunchKernelGGL((lstm_out_kernel<3, false>), g, t, 0, s, a); break;
        case 7: hipLaunchKernelGGL((lstm_out_kernel<3, true>), g, t, 0, s, a); break;
        case 8: hipLaunchKernelGGL((lstm_out_kernel<4, false>), g, t, 0, s, a); break;
        case 9: hipLaunchKernelGGL((lstm_out_kernel<4, true>), g, t, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lstm_tile(const LstmTileArgs &a, hipStream_t s) {
    const long long total = (long long)a.F * a.HW;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 g((unsigned)blocks), t(256);
    switch (a.C * 2 + (a.h_bf16 ? 1 : 0)) {
        case 4: hipLaunchKernelGGL((lstm_tile_kernel<2, false>), g, t, 0, s, a); break;
        case 5: hipLaunchKernelGGL((lstm_tile_kernel<2, true>), g, t, 0, s, a); break;
        case 6: hipLaunchKernelGGL((lstm_tile_kernel<3, false>), g, t, 0, s, a); break;
        case 7: hipLaunchKernelGGL((lstm_tile_kernel<3, true>), g, t, 0, s, a); break;
        case 8: hipLaunchKernelGGL((lstm_tile_kernel<4, false>), g, t, 0, s, a); break;
        case 9: hipLaunchKernelGGL((lstm_tile_kernel<4, true>), g, t, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ukbb
