// BiConvLSTM head of the aortic UNet-LSTM model (reference common/network_ao.py:255-319 with
// tf.contrib.rnn.Conv2DLSTMCell semantics, SURVEY.md App. B.6 [TF-recall]).  The gate convolution
// (3x3 over concat([x_t, h]) -> 4*16 channels) runs on the Winograd MFMA kernel; what is left per step is
// HBM-bound element-wise work, fused here into one pass per step:
//
//   i, j, f, o = split(gates);  c' = sigmoid(f + 1) * c + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o)
//   + this direction's half of the 1x1 output conv (16 -> n_class) accumulated per pixel, and in the
//   backward direction's pass the bias, softmax and argmax (network_ao.py:305-312, 396-397),
//
// so the per-step hidden maps of the two directions are never stored for a later concat.
// lstm_tile_kernel then performs the weighted circular tiling of deploy_network_ao.py:176-183 in the
// reference's accumulation order and arithmetic (float32 accumulator updated through float64).
#include "kernels.h"

namespace ukbb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NH = 16;                       // hidden channels (train_network_ao.py num_hidden = 16)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); exact enough in fp32 (|err| ~ 1e-7) and saturates cleanly
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// One thread = 4 hidden channels of one pixel; 4 consecutive lanes = one pixel.
template <int NCLS>
__global__ __launch_bounds__(256) void lstm_cell_kernel(const LstmCellArgs a) {
    const long long total = (long long)a.M * a.HW * (NH / 4);
    const int q = threadIdx.x & 3;
    float w[4][NCLS];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < NCLS; ++c) w[i][c] = a.w_out[(4 * q + i) * NCLS + c];
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const long long px = id >> 2;                                   // global pixel index m*HW + pix
        const float *g = a.gates + px * (4 * NH) + 4 * q;
        const f32x4 gi = *reinterpret_cast<const f32x4 *>(g);
        const f32x4 gj = *reinterpret_cast<const f32x4 *>(g + NH);
        const f32x4 gf = *reinterpret_cast<const f32x4 *>(g + 2 * NH);
        const f32x4 go = *reinterpret_cast<const f32x4 *>(g + 3 * NH);
        f32x4 c = *reinterpret_cast<const f32x4 *>(a.c + px * NH + 4 * q);
        f32x4 h;
        float part[NCLS];
#pragma unroll
        for (int k = 0; k < NCLS; ++k) part[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = sigmoidf_(gf[i] + a.forget_bias) * c[i] + sigmoidf_(gi[i]) * tanhf_(gj[i]);
            h[i] = tanhf_(c[i]) * sigmoidf_(go[i]);
#pragma unroll
            for (int k = 0; k < NCLS; ++k) part[k] = fmaf(h[i], w[i][k], part[k]);
        }
        *reinterpret_cast<f32x4 *>(a.c + px * NH + 4 * q) = c;
        *reinterpret_cast<f32x4 *>(a.h + px * NH + 4 * q) = h;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) {                                // reduce the 4 channel quads of the pixel
            part[k] += __shfl_xor(part[k], 1);
            part[k] += __shfl_xor(part[k], 2);
        }
        if (q == 0) {
            const long long m = px / a.HW, pix = px - m * a.HW;
            float *acc = a.acc + m * a.m_stride + pix * NCLS;
            if (!a.finish) {
#pragma unroll
                for (int k = 0; k < NCLS; ++k) acc[k] = part[k];
            } else {
                float lg[NCLS];
#pragma unroll
                for (int k = 0; k < NCLS; ++k) lg[k] = acc[k] + part[k] + a.b_out[k];
                if (a.logits) {
                    float *lo = a.logits + m * a.m_stride + pix * NCLS;
#pragma unroll
                    for (int k = 0; k < NCLS; ++k) lo[k] = lg[k];
                }
                float pr[NCLS];
                const int best = softmax_argmax<NCLS>(lg, pr);
#pragma unroll
                for (int k = 0; k < NCLS; ++k) acc[k] = pr[k];
                if (a.pred) a.pred[m * (a.m_stride / NCLS) + pix] = best;
            }
        }
    }
}

// prob[f] = (sum over the <= K windows containing f, in the reference's order, of probw * w_k) / wsum[f]
//   a frame no window reaches (time_step > window) has wsum = 0: 0/0 = NaN and argmax 0, as numpy gives the reference
//   reference arithmetic: prob is float32, `prob[..., idx] += prob_idx * w` runs in float64 and is cast back
//   per addition; `prob /= weight` likewise (deploy_network_ao.py:176-183).
template <int NCLS>
__global__ __launch_bounds__(256) void lstm_tile_kernel(const LstmTileArgs a) {
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
            const int w = wk_ / a.K, k = wk_ - w * a.K;
            const float *p = a.probw + (((long long)k * a.Wn + w) * a.HW + pix) * NCLS;
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

hipError_t launch_lstm_cell(const LstmCellArgs &a, hipStream_t s) {
    const long long total = (long long)a.M * a.HW * (NH / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (a.n_class) {
        case 2: hipLaunchKernelGGL(lstm_cell_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(lstm_cell_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(lstm_cell_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lstm_tile(const LstmTileArgs &a, hipStream_t s) {
    const long long total = (long long)a.F * a.HW;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (a.C) {
        case 2: hipLaunchKernelGGL(lstm_tile_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(lstm_tile_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(lstm_tile_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ukbb
