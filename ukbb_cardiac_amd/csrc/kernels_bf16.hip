// Persistent form of the LAST layer of the aortic U-Net in UKBB_PREC_BF16: up0_1 (3x3, 16 -> 16, BN, ReLU) with the 1x1 logits
// conv + softmax / argmax in its epilogue (reference common/network_ao.py:52-63,159-160 through network.py:19-25); bf16
// activations in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Why (r03, profiles/r03_notes.md section 1): the layer has ONE 16-channel chunk and ONE 32-row Cout block, so in the
// tile-per-workgroup kernel (conv_mfma_kernel<..., BFIO, FUSE = 2>) a workgroup's life is a latency chain -- fetch the halo
// tile (2-3 us from HBM), stage the same 9 KB of packed weights again, 18 MFMAs per wave, epilogue -- and 12 800 such
// workgroups per launch overlap only through occupancy: 111-125 us.  Here a workgroup is persistent over tiles: the packed
// weights and the logits weights are staged ONCE per workgroup, the next tile's global loads are in flight while the current
// tile is computed (registers) and are parked in the other LDS buffer afterwards, ONE barrier per tile: 103-110 us.
// (The same structure was built for the fused FIRST layer and measured no better than the tile-per-workgroup form, 157-168
// vs 150-156 us: that layer is bound by vector-ALU + fp32-MFMA issue of the conv0_0 evaluation, not by latency; dropped.)
// Fragment layouts, packed weights (pack_conv_weights_bf16, one Cout block of 32 rows of which 16 are real), rounding and
// the epilogue arithmetic are those of conv_mfma_kernel<..., BFIO, 2>: results are bit-identical to that kernel's.
#include "kernels.h"

#include <type_traits>

namespace ukbb {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(const f32x4 &a, const f32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;       // RNE; v_cvt_pk_bf16_f32
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_steps(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_steps<N, I + 1>(f); }
}

constexpr int XS = 12;                                  // dwords per staged halo pixel: 16 bf16 channels + 4 pad
constexpr int WSLAB = 9 * 64 * 4;                       // packed weights of the one Cout block: 9 taps x 64 lanes x 4 dwords

__host__ __device__ constexpr int pk16_lds_bytes(int th, int tw) {
    return 4 * (WSLAB + 2 * (th + 2) * (tw + 2) * XS);
}

template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv16_logits_pk_kernel(const ConvArgs a) {
    constexpr int NPIX = TH * TW, PBW = NPIX / 128;     // 32-pixel blocks per wave (4 waves along the pixels)
    constexpr int IH = TH + 2, IW = TW + 2, HP = IH * IW;
    constexpr int XBUF = HP * XS;
    constexpr int NIT = (HP * 2 + 255) / 256;           // 16-byte pieces of the bf16 halo tile per thread
    constexpr int NWT = (WSLAB / 4 + 255) / 256;
    static_assert(NPIX % 128 == 0, "tile = 4 waves x PBW x 32 pixels");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ws = lds;                                     // [9][64][4]
    float *xs0 = lds + WSLAB;                            // two halo tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 5, pl = lane & 31;
    const int tiles = a.tiles_x * a.tiles_y, ntiles = a.N * tiles;
    const int my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my <= 0) return;

    int lbase[PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; ++pb) {
        const int q = (wave + pb * 4) * 32 + pl;
        lbase[pb] = ((q / TW) * IW + q % TW) * XS + 4 * g;
    }
    const int wbase = lane * 4;
    auto locate = [&](int k, int &n, int &oy0, int &ox0) {
        int t = blockIdx.x + k * gridDim.x;
        n = t / tiles; t -= n * tiles;
        const int ty = t / a.tiles_x;
        oy0 = ty * TH; ox0 = (t - ty * a.tiles_x) * TW;
    };

    // ---- once per workgroup: packed weights -> LDS; bias and logits weights -> registers ----
    {
        f32x4 wr[NWT];
#pragma unroll
        for (int it = 0; it < NWT; ++it) wr[it] = *reinterpret_cast<const f32x4 *>((it * 256 + tid < WSLAB / 4) ? a.wpk + 4 * (it * 256 + tid) : a.wpk);
#pragma unroll
        for (int it = 0; it < NWT; ++it) if (it * 256 + tid < WSLAB / 4) *reinterpret_cast<f32x4 *>(ws + 4 * (it * 256 + tid)) = wr[it];
    }
    float4 bi[2];                                        // this lane's 2 x 4 real output channels: 4g + 8j ..
#pragma unroll
    for (int j = 0; j < 2; ++j) bi[j] = *reinterpret_cast<const float4 *>(a.bias + 4 * g + 8 * j);
    float w8[8][4], bl[4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) w8[k][c] = c < a.lg_ncls ? a.lg_w[(4 * g + (k & 3) + 8 * (k >> 2)) * a.lg_ncls + c] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) bl[c] = c < a.lg_ncls ? a.lg_b[c] : 0.f;

    f32x4 xr[NIT];
    int piy[NIT], pix_[NIT];                             // halo coordinates of this thread's pieces (tile independent)
    const int c2 = tid & 1, pix0 = tid >> 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = pix0 + it * 128;
        piy[it] = pix < HP ? pix / IW : -1000000;        // pieces beyond the tile never pass the bounds test
        pix_[it] = pix - (pix / IW) * IW;
    }

    // ---- tile input: request (global -> registers) and park (registers -> LDS) ----
    auto request = [&](int k) {
        int n, oy0, ox0;
        locate(k, n, oy0, ox0);
        const unsigned short *src = reinterpret_cast<const unsigned short *>(a.in0) + (size_t)n * a.H * a.W * 16 + 8 * c2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {               // unconditional loads from clamped addresses, all in flight at once
            const int gy = oy0 - 1 + piy[it], gx = ox0 - 1 + pix_[it];
            const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (ok ? ((size_t)gy * a.W + gx) * 16 : 0));
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            xr[it] = ok ? v : zero4;                     // zero padding decided here: the registers are parked as they are
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = pix0 + it * 128;
            if (pix < HP) *reinterpret_cast<f32x4 *>(xs0 + buf * XBUF + pix * XS + 4 * c2) = xr[it];
        }
    };

    // ---- prologue ----
    request(0);
    park(0);
    if (my > 1) request(1);
    __syncthreads();                                     // weights, first halo tile

    for (int k = 0; k < my; ++k) {
        const float *xs = xs0 + (k & 1) * XBUF;
        if (k + 1 < my) park((k + 1) & 1);               // the buffer the tile before last was computed from (barrier at the end of k-1)
        if (k + 2 < my) request(k + 2);
        // ---- 9 taps: A = packed weights, B = halo pixels, both from LDS, reads one tap ahead ----
        f32x16 acc[PBW];
#pragma unroll
        for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
        {
            f32x4 av[2], bv[2][PBW];
            auto fetch = [&](auto tc, int set) {
                constexpr int t = decltype(tc)::value, kh = t / 3, kw = t % 3;
                av[set] = *reinterpret_cast<const f32x4 *>(ws + wbase + t * 256);
#pragma unroll
                for (int pb = 0; pb < PBW; ++pb) bv[set][pb] = *reinterpret_cast<const f32x4 *>(xs + lbase[pb] + (kh * IW + kw) * XS);
            };
            fetch(std::integral_constant<int, 0>{}, 0);
            unroll_steps<9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t + 1 < 9) fetch(std::integral_constant<int, t + 1>{}, (t + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < PBW; ++pb) acc[pb] = mfma_bf16(av[t & 1], bv[t & 1][pb], acc[pb]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- epilogue: bias, ReLU, bf16 rounding (as a store would), 16 -> n_class, softmax / argmax (kernels.h) ----
        int n, oy0, ox0;
        locate(k, n, oy0, ox0);
#pragma unroll
        for (int pb = 0; pb < PBW; ++pb) {
            const int q = (wave + pb * 4) * 32 + pl;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int j = kk >> 2, i = kk & 3;
                float v = acc[pb][4 * j + i] + (&bi[j].x)[i];
                if (a.relu) v = fmaxf(v, 0.f);
                const float r = __builtin_bit_cast(float, pack_bf16x2(v, 0.f) << 16);      // the value a bf16 store would have held
#pragma unroll
                for (int c = 0; c < 4; ++c) part[c] = fmaf(r, w8[kk][c], part[c]);
            }
            float lgv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) lgv[c] = (part[c] + __shfl_xor(part[c], 32)) + bl[c];
            if (g == 0 && oy < a.Ho && ox < a.Wo) {
                const size_t px = (size_t)(n * a.Ho + oy) * a.Wo + ox;
                auto finish = [&](auto nc) {
                    constexpr int NC = decltype(nc)::value;
                    float l[NC], p[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) l[c] = lgv[c];
                    const int best = softmax_argmax_opt<NC>(l, a.lg_prob != nullptr, p);
                    if (a.lg_pred) a.lg_pred[px] = best;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (a.lg_logits) a.lg_logits[px * NC + c] = l[c];
                        if (a.lg_prob) a.lg_prob[px * NC + c] = p[c];
                    }
                };
                if (a.lg_ncls == 2) finish(std::integral_constant<int, 2>{});
                else if (a.lg_ncls == 3) finish(std::integral_constant<int, 3>{});
                else finish(std::integral_constant<int, 4>{});
            }
        }
        __syncthreads();                                 // buffer (k+1)&1 complete; everyone is done reading buffer k&1
    }
}

}  // namespace

// P(id, TH, TW); ConvConfig::pc == 5 (bf16 storage), fuse == 2, one 32-row Cout block, kc = 16
#define UKBB_PK16_CONFIGS(P)                   \
    P(324, 8, 32)                              \
    P(325, 16, 16)

#define UKBB_PK16_ENTRY(ID, TH, TW)                                                                     \
    {ID, 3, 1, 32, TH, TW, 16, 1, 4, 1, pk16_lds_bytes(TH, TW), 5, "convBF16pk_3x3s1+logits_t" #TH "x" #TW, 2},
static const ConvConfig g_pk16_cfgs[] = {UKBB_PK16_CONFIGS(UKBB_PK16_ENTRY)};

int num_pk16_configs() { return (int)(sizeof(g_pk16_cfgs) / sizeof(g_pk16_cfgs[0])); }
const ConvConfig &pk16_config(int i) { return g_pk16_cfgs[i]; }

hipError_t launch_conv16_pk(int cfg_id, const ConvArgs &a, hipStream_t s) {
    const ConvConfig *c = nullptr;
    for (const ConvConfig &k : g_pk16_cfgs) if (k.id == cfg_id) c = &k;
    if (!c) return hipErrorInvalidValue;
    // one source of 16 channels, 16 real output channels in a 32-row block, logits in the epilogue
    if (a.in0_map || a.in1 || a.C1 || a.C0 != 16 || a.Cout != 32 || a.cout_store != 16 || a.up2) return hipErrorInvalidValue;
    if (!a.lg_w || !a.lg_b || a.lg_ncls < 2 || a.lg_ncls > 4) return hipErrorInvalidValue;
    const long long ntiles = (long long)a.N * a.tiles_y * a.tiles_x;
    const int per_cu = 160 * 1024 / c->lds_bytes < 4 ? 160 * 1024 / c->lds_bytes : 4;
    const long long cap = (long long)device_cu_count() * (per_cu < 1 ? 1 : per_cu);
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
    switch (cfg_id) {
#define UKBB_PK16_CASE(ID, TH, TW)                                                              \
    case ID: {                                                                                  \
        auto k = conv16_logits_pk_kernel<TH, TW>;                                               \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), c->lds_bytes, s, a);                             \
        break;                                                                                  \
    }
        UKBB_PK16_CONFIGS(UKBB_PK16_CASE)
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ukbb
