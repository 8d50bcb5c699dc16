// conv3x3 / conv1x1 (+ folded BN bias, ReLU) as implicit GEMM on the gfx950
// f32 MFMA pipes.  Replaces the TF ops behind conv2d_bn_relu
// (reference common/network.py:19-25) in inference mode.
//
// GEMM view (per image):  D[cout][pixel] = sum_{tap,ci} W'[tap][ci][cout] * X[pixel+tap][ci]
//   MFMA A operand = weights  (M = output channels)
//   MFMA B operand = pixels   (N = output pixels)   -> the accumulator holds,
//   per lane, 4 consecutive output channels of ONE pixel: NHWC float4 stores.
//
// Data movement per workgroup (256 threads = 4 waves):
//   * the input halo tile of KC channels is staged once into LDS as
//     xs[halo pixel][KC + 4 pad] (straight float4 copies of the NHWC rows); every one
//     of the 9 taps then re-reads it with ONE ds_read_b128/b64 per pixel block that
//     returns the lane's B operands for KC/NG consecutive k-steps (the k order is
//     permuted so that lane group g owns channels g*KC/NG ...; the weights are packed
//     to match).  Address = lane_base + compile-time immediate (tap shift).
//   * weights are pre-packed on the host in A-fragment order, so a lane reads
//     its KC/KK k-steps of one tap as one 16/32-byte global load (L1/L2 hits:
//     every workgroup of a layer streams the same few KB).
// The f32 MFMA rate equals the vector rate (64 FLOP/clk/SIMD); with one LDS read per
// 4-8 MFMAs the matrix pipe is the only bound (r01 PMC: the earlier planar/ds_read_b32
// layout spent 44 % of its LDS cycles in bank conflicts, profiles/r01_pmc_summary_head2.csv).
#include "kernels.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <vector>
#include <type_traits>

namespace ukbb {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ReLU as one v_max_i32 on the bit pattern (fmaxf on an MFMA result compiles to two v_max_f32).
__device__ __forceinline__ float relu_bits(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

template <int MB> struct Mfma;
template <> struct Mfma<32> {
    using Acc = f32x16;
    static constexpr int KK = 2;    // k per instruction
    static constexpr int NACC = 16;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<16> {
    using Acc = f32x4;
    static constexpr int KK = 4;
    static constexpr int NACC = 4;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 inputs, fp32 accumulate: one v_mfma_f32_32x32x16_bf16 covers a whole KC = 16 chunk of a tap.
// Lane (m = lane & 31, h = lane >> 5) supplies A[m][k = 8h..8h+7] and B[k = 8h..8h+7][m] as 8 bf16
// (4 dwords); the accumulator layout is the same as the f32 32x32 form.
__device__ __forceinline__ f32x16 mfma_bf16(const f32x4 &a, const f32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;       // RNE; hipcc emits v_cvt_pk_bf16_f32
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}

__host__ __device__ constexpr int xs_stride(int kc) {
    // floats per staged halo pixel: KC channels + 4 pad, so the 16-byte reads of the 16 lanes
    // of a ds_read_b128 group (consecutive pixels) fall on distinct bank quads (5p mod 16).
    return kc + 4;
}

__host__ __device__ constexpr int conv_lds_bytes(int ks, int s, int mb, int th, int tw, int kc, int wm, int cb) {
    const int xs = xs_stride(kc) * ((th - 1) * s + ks) * ((tw - 1) * s + ks);
    return (xs + wm * cb * ks * ks * 64 * (kc / (mb == 32 ? 2 : 4))) * 4;
}

__host__ __device__ constexpr int conv_min_waves(int ks, int s, int mb, int th, int tw, int kc, int wm, int wn, int cb) {
    // Waves per SIMD to ask of the register allocator: 2 (two resident workgroups per CU)
    // unless the accumulators + staging registers clearly do not fit 256 VGPRs.
    const int pb = mb, npb = (th * tw + pb - 1) / pb, pbw = (npb + wn - 1) / wn;
    const int hp = ((th - 1) * s + ks) * ((tw - 1) * s + ks), c4 = kc / 4;
    const int nit = (hp * c4 + 255) / 256;
    const int nwt = (wm * cb * ks * ks * 64 * (kc / (mb == 32 ? 2 : 4)) / 4 + 255) / 256;
    const int est = cb * pbw * (mb == 32 ? 16 : 4) + 4 * (nit + nwt) + pbw + nit + 2 * pbw + 48;
    return est <= 170 ? 2 : 1;
}

template <int N> struct VecLoad;
template <> struct VecLoad<2> {
    static __device__ __forceinline__ void ld(const float *p, float *d) {
        float2 v = *reinterpret_cast<const float2 *>(p); d[0] = v.x; d[1] = v.y; }
};
template <> struct VecLoad<4> {
    static __device__ __forceinline__ void ld(const float *p, float *d) {
        float4 v = *reinterpret_cast<const float4 *>(p); d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
};
template <> struct VecLoad<8> {
    static __device__ __forceinline__ void ld(const float *p, float *d) {
        VecLoad<4>::ld(p, d); VecLoad<4>::ld(p + 4, d + 4); }
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_taps(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        unroll_taps<N, I + 1>(f);
    }
}

// BF = true: bf16 operands (activations converted while staging, weights pre-converted on the host),
// fp32 accumulation; requires MB = 32, KC = 16.  LDS then holds [halo px][8 dwords + 4 pad] and
// one packed 4-dword fragment per lane per tap.
// BFIO = true (with BF): activations are bf16 NHWC in HBM on both sides (UKBB_PREC_BF16 of the aortic U-Net, r03): staging
// is a straight 16-byte copy of 8 channels (no conversion), the epilogue rounds the fp32 accumulator + bias (+ ReLU) to
// bf16 once and stores 8 bytes per lane and 4-channel group.  ConvArgs::Cout is then the PADDED channel count the weights
// were packed for (multiple of the workgroup's group) and ConvArgs::cout_store the real one: a 16-channel layer runs on the
// 32-row MFMA with 16 zero rows whose results are not stored.
template <int KS, int STRIDE, int MB, int TH, int TW, int KC, int WM, int WN, int CB, bool BF = false, bool BFIO = false, int FUSE = 0>
__global__ __launch_bounds__(256, conv_min_waves(KS, STRIDE, MB, TH, TW, KC, WM, WN, CB)) void conv_mfma_kernel(const ConvArgs a) {
    static_assert(FUSE == 0 || (BFIO && KS == 3 && STRIDE == 1 && CB == 1 && WM == 1), "fused first layer / logits: bf16 storage, 3x3 s1, one Cout block");
    static_assert(!BF || (MB == 32 && KC == 16), "bf16 path: 32x32x16 MFMA, one chunk = one K step");
    static_assert(!BFIO || BF, "bf16 storage needs the bf16 operand path");
    using M = Mfma<MB>;
    using Acc = typename M::Acc;
    constexpr int KK = M::KK, KSTEPS = BF ? 4 : KC / KK, PB = MB;   // BF: "k-steps" = the 4 dwords of one bf16 fragment
    constexpr int NPIX = TH * TW, NPB = (NPIX + PB - 1) / PB, PBW = (NPB + WN - 1) / WN;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int HP = IH * IW, XS = BF ? KC / 2 + 4 : xs_stride(KC), C4 = BFIO ? KC / 8 : KC / 4, KS2 = KS * KS;   // C4: 16-byte pieces per halo pixel
    constexpr int ES = BFIO ? 2 : 4;                     // bytes per element of the activations in HBM
    constexpr int NCBL = WM * CB;                       // Cout blocks per workgroup
    constexpr int SLAB = KS2 * 64 * KSTEPS;             // packed weights of one Cout block, one chunk
    constexpr int NIT = (HP * C4 + 255) / 256;          // activation float4 per thread per chunk
    constexpr int WF4 = NCBL * SLAB / 4;                // weight float4 per workgroup per chunk
    constexpr int NWT = (WF4 + 255) / 256;
    constexpr int PSTEP = 256 / C4;                     // halo pixels advanced per staging iteration
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(KC % 4 == 0 && KC % KK == 0 && 256 % C4 == 0, "KC");
    static_assert(KSTEPS == 2 || KSTEPS == 4 || KSTEPS == 8, "KSTEPS");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs = lds;                                     // [HP][XS]  halo pixels x (KC channels + pad)
    float *ws = lds + HP * XS;                           // [NCBL][KS2][64][KSTEPS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane / PB, pl = lane % PB;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int cbg = blockIdx.y * NCBL + wm * CB;         // first Cout block of this wave

    int lbase[PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; ++pb) {
        int q = (wn + pb * WN) * PB + pl;
        if (q >= NPIX) q = 0;
        const int oy = q / TW, ox = q % TW;
        lbase[pb] = ((oy * STRIDE) * IW + ox * STRIDE) * XS + KSTEPS * g;   // lane group g owns channels g*KSTEPS..
    }

    Acc acc[CB][PBW];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[cb][pb][r] = 0.f;

    // Sub-pixel form of the 3x3 stride-2 transposed conv (up2 > 0, KS = 2): output phase (py, px) of a Cout block uses
    // tap (a, b) of the 2x2 window only if (py == 0 || a == 1) && (px == 0 || b == 1) -- 9 of the 16 (tap, phase) pairs;
    // the other 7 hold zero weights (kernels.h, tconv_as_conv2x2) and their MFMAs are skipped (wave-uniform test).
    unsigned tapmask[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        tapmask[cb] = 0xffffffffu;
        if (KS == 2 && a.up2 > 0) {
            tapmask[cb] = 0;
            const int ph0 = ((cbg + cb) * MB) / a.up2, ph1 = ((cbg + cb) * MB + MB - 1) / a.up2;   // a block covers > 1 phase when up2 < MB
            for (int ph = ph0; ph <= ph1 && ph < 4; ++ph) {
                const int py = ph >> 1, px = ph & 1;
                for (int t = 0; t < 4; ++t)
                    if ((py == 0 || (t >> 1) == 1) && (px == 0 || (t & 1) == 1)) tapmask[cb] |= 1u << t;
            }
        }
    }

    const int nchunk = (a.C0 + a.C1) / KC;
    const int iy0 = oy0 * STRIDE - a.pad_y, ix0 = ox0 * STRIDE - a.pad_x;
    const int c4 = tid % C4, pix0 = tid / C4;           // this thread's channel quad / first halo pixel
    const float *wsrc = a.wpk + (size_t)blockIdx.y * nchunk * (NCBL * SLAB);

    // Per-iteration global offsets of this thread's halo pixels (chunk independent), -1 = zero padding.
    int goff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pix = pix0 + it * PSTEP;
        const int iy = pix / IW, ix = pix % IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool ok = pix < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        // bf16 storage is channel-blocked ([N][C/16][H][W][16], kernels.h): the offset is relative to the image's plane
        goff[it] = ok ? (BFIO ? gy * a.W + gx : (n * a.H + gy) * a.W + gx) : -1;      // -1: zero padding / outside the tile
    }

    f32x4 xr[NIT], wr[NWT];
#define UKBB_PREFETCH_X(CH)                                                                        \
    {                                                                                              \
        const int ch_ = (CH);                                                                      \
        const char *src_; int cs_;                                                                 \
        if (ch_ * KC < a.C0) { src_ = reinterpret_cast<const char *>(a.in0) + (size_t)(ch_ * KC) * ES; cs_ = a.C0; } \
        else                 { src_ = reinterpret_cast<const char *>(a.in1) + (size_t)(ch_ * KC - a.C0) * ES; cs_ = a.C1; } \
        if constexpr (BFIO) {   /* plane of this 16-channel chunk of image n: [N][C/16][H][W][16] bf16 */             \
            const int chl_ = ch_ * KC < a.C0 ? ch_ : ch_ - a.C0 / KC;                              \
            src_ = (ch_ * KC < a.C0 ? reinterpret_cast<const char *>(a.in0) : reinterpret_cast<const char *>(a.in1)) + \
                   ((size_t)n * (cs_ / KC) + chl_) * a.H * a.W * 32;                               \
            cs_ = KC;                                                                              \
        }                                                                                          \
        src_ += 16 * c4;                                                                           \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                       \
            /* unconditional load from a clamped address: a branch around the load makes hipcc  */ \
            /* wait for it at the join, which would serialise the prefetch with the MFMA phase; */ \
            /* padding pixels are zeroed when the registers are written to LDS instead.          */ \
            xr[it] = *reinterpret_cast<const f32x4 *>(src_ + (size_t)(goff[it] < 0 ? 0 : goff[it]) * cs_ * ES); \
        }                                                                                          \
    }
#define UKBB_PREFETCH_W(CH)                                                                        \
    {                                                                                              \
        const float *wp_ = wsrc + (size_t)(CH) * (NCBL * SLAB);                                    \
        _Pragma("unroll") for (int it = 0; it < NWT; ++it)                                         \
            wr[it] = *reinterpret_cast<const f32x4 *>((it * 256 + tid < WF4) ? wp_ + 4 * (it * 256 + tid) : a.wpk); \
    }
#define UKBB_PREFETCH(CH) { UKBB_PREFETCH_X(CH) UKBB_PREFETCH_W(CH) }

    // Fused first layer (bf16 storage, 3x3 stride 1 only): in0 is then the network's 1-channel fp32 image and this conv's 16
    // input channels are relu(BN(conv3x3(image))) (network_ao.py:31-35 with l = 0, i = 0) evaluated here for every halo pixel:
    // the raw (IH+2) x (IW+2) tile goes to LDS, then per 16 halo pixels three v_mfma_f32_16x16x4_f32 (fp32-exact; A = the
    // [16 x 12] folded filter with taps 9..11 zero, B = the lane's pixel at tap k, bias as the C operand) give a lane 4
    // consecutive channels of its pixel, which are rounded to bf16 once -- what the stand-alone first-layer kernel would have
    // stored up to the fp32 summation order (MFMA with the bias as C against an fmaf chain plus bias) -- and written into the halo tile.  conv0_0's output (420 MB per 100 slices, written + re-read) never exists.
    constexpr bool fusedf = FUSE == 1;
    // first stage requested before anything else: in the fused form the packed weights travel while the raw tile is fetched and
    // the first layer is evaluated
    if constexpr (fusedf) UKBB_PREFETCH_W(0) else UKBB_PREFETCH(0)
    if constexpr (fusedf) {
        {
            constexpr int RH = IH + 2, RW = IW + 2, RP = RH * RW, NBLK = (HP + 15) / 16;
            float *raw = ws + NCBL * SLAB;
            const float *img = a.in0 + (size_t)n * a.H * a.W;
            constexpr int NRAW = (RP + 255) / 256;
            float rv[NRAW];
#pragma unroll
            for (int k = 0; k < NRAW; ++k) {                   // all loads in flight at once (clamped address, select afterwards)
                const int i = tid + 256 * k;
                const int ry = i / RW, rx = i - ry * RW, gy = iy0 - 1 + ry, gx = ix0 - 1 + rx;
                const bool ok = i < RP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const float v = img[ok ? (size_t)gy * a.W + gx : 0];
                rv[k] = ok ? v : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NRAW; ++k)
                if (tid + 256 * k < RP) raw[tid + 256 * k] = rv[k];
            const int pj = lane & 15, pg = lane >> 4;
            float wA[3]; int toff[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int t = 4 * k + pg;
                wA[k] = t < 9 ? a.first_w[t * 16 + pj] : 0.f;
                const int tt = t < 9 ? t : 8;                  // zero weight: any finite value of the tile will do
                toff[k] = (tt / 3) * RW + (tt % 3);
            }
            const f32x4 biasq = *reinterpret_cast<const f32x4 *>(a.first_b + 4 * pg);
            __syncthreads();
            for (int blk = wave; blk < NBLK; blk += 4) {       // wave-uniform trip count
                const int pix = blk * 16 + pj;
                const int hy = pix / IW, hx = pix - hy * IW;
                const int ro = pix < HP ? hy * RW + hx : 0;
                f32x4 acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[0], raw[ro + toff[0]], biasq, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[1], raw[ro + toff[1]], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[2], raw[ro + toff[2]], acc0, 0, 0, 0);
                // halo pixels outside the image are THIS conv's zero padding, not conv0_0 of padded input
                const bool in = (unsigned)(iy0 + hy) < (unsigned)a.H && (unsigned)(ix0 + hx) < (unsigned)a.W;
                uint2 pk;
                pk.x = in ? pack_bf16x2(fmaxf(acc0[0], 0.f), fmaxf(acc0[1], 0.f)) : 0u;
                pk.y = in ? pack_bf16x2(fmaxf(acc0[2], 0.f), fmaxf(acc0[3], 0.f)) : 0u;
                if (pix < HP) *reinterpret_cast<uint2 *>(xs + pix * XS + 2 * pg) = pk;
            }
        }
    }

    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch > 0) __syncthreads();                    // all waves done reading the previous chunk
        // ---- registers -> LDS (straight 16-byte copies) ----
        if constexpr (!fusedf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = pix0 + it * PSTEP;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BFIO) {
                if (pix < HP) *reinterpret_cast<f32x4 *>(xs + pix * XS + 4 * c4) = goff[it] < 0 ? zero4 : xr[it];   // 8 bf16 as they come
            } else if constexpr (BF) {
                const f32x4 v = goff[it] < 0 ? zero4 : xr[it];
                uint2 pk; pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                if (pix < HP) *reinterpret_cast<uint2 *>(xs + pix * XS + 2 * c4) = pk;
            } else {
                if (pix < HP) *reinterpret_cast<f32x4 *>(xs + pix * XS + 4 * c4) = goff[it] < 0 ? zero4 : xr[it];
            }
        }
        }
#pragma unroll
        for (int it = 0; it < NWT; ++it)
            if (it * 256 + tid < WF4) *reinterpret_cast<f32x4 *>(ws + it * 1024 + 4 * tid) = wr[it];
        __syncthreads();
        if (ch + 1 < nchunk) UKBB_PREFETCH(ch + 1)      // in flight during the MFMA phase below
        // ---- MFMA over the taps: A and B both from LDS, software-pipelined by one tap:
        //      the LDS reads of tap t+1 are issued before the MFMAs of tap t (two register sets,
        //      statically indexed because the tap loop is fully unrolled).
        {
            const float *wbase = ws + (wm * CB) * KS2 * 64 * KSTEPS + lane * KSTEPS;
            float av[2][CB][KSTEPS], bv[2][PBW][KSTEPS];
#define UKBB_LOAD_TAP(T, SET)                                                                       \
            {                                                                                      \
                constexpr int kh_ = (T) / KS, kw_ = (T) % KS;                                      \
                _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                                  \
                    VecLoad<KSTEPS>::ld(wbase + (cb * KS2 + (T)) * 64 * KSTEPS, av[SET][cb]);      \
                _Pragma("unroll") for (int pb = 0; pb < PBW; ++pb)                                 \
                    VecLoad<KSTEPS>::ld(xs + lbase[pb] + (kh_ * IW + kw_) * XS, bv[SET][pb]);      \
            }
            UKBB_LOAD_TAP(0, 0)
            unroll_taps<KS2>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t + 1 < KS2) UKBB_LOAD_TAP(t + 1, (t + 1) & 1)
                // Pin the software pipeline: without the barriers hipcc sinks the reads of tap
                // t+1 down to their first use (exposing the LDS latency before every MFMA group).
                __builtin_amdgcn_sched_barrier(0);
                // k-step outer / pixel block inner: consecutive MFMAs write different accumulators
                // (a 16x16x4 f32 MFMA has 40 cycles dependent latency but issues every 32).
                if constexpr (BF) {
#pragma unroll
                    for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) {
                            if (KS == 2 && !((tapmask[cb] >> t) & 1)) continue;
                            const f32x4 af = {av[t & 1][cb][0], av[t & 1][cb][1], av[t & 1][cb][2], av[t & 1][cb][3]};
                            const f32x4 bf = {bv[t & 1][pb][0], bv[t & 1][pb][1], bv[t & 1][pb][2], bv[t & 1][pb][3]};
                            acc[cb][pb] = mfma_bf16(af, bf, acc[cb][pb]);
                        }
                } else if constexpr (KS == 2) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        if ((tapmask[cb] >> t) & 1) {
#pragma unroll
                            for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
                                for (int pb = 0; pb < PBW; ++pb)
                                    acc[cb][pb] = M::run(av[t & 1][cb][s], bv[t & 1][pb][s], acc[cb][pb]);
                        }
                } else {
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
                    for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[cb][pb] = M::run(av[t & 1][cb][s], bv[t & 1][pb][s], acc[cb][pb]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#undef UKBB_LOAD_TAP
        }
    }

    // ---- epilogue: + bias, ReLU, NHWC float4 stores -----------------------------
    constexpr int NJ = M::NACC / 4;              // float4 groups per accumulator (4 or 1)
    if constexpr (BFIO) {
        // bf16 NHWC: per lane and 4-channel group one 8-byte store; channels >= cout_store (zero-padded rows) are dropped
        const int cst = a.cout_store > 0 ? a.cout_store : a.Cout;
        unsigned short *ob = reinterpret_cast<unsigned short *>(a.out);
        if constexpr (FUSE == 2) {
            // Last layer with the logits fused (16 real channels: lane half g holds channels 4g..4g+3 and 8+4g..8+4g+3 of its
            // pixel): the activation is rounded to bf16 as it would have been stored, each half forms its 8-channel part of the
            // 16 -> n_class product, the halves are added across lanes l / l^32, then bias, softmax / argmax (kernels.h).
            float w8[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) w8[k][c] = c < a.lg_ncls ? a.lg_w[(4 * g + (k & 3) + 8 * (k >> 2)) * a.lg_ncls + c] : 0.f;
            float bl[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bl[c] = c < a.lg_ncls ? a.lg_b[c] : 0.f;
            float4 bi2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bi2[j] = *reinterpret_cast<const float4 *>(a.bias + 4 * g + 8 * j);
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb) {
                const int q = (wn + pb * WN) * PB + pl;
                const int oy = oy0 + q / TW, ox = ox0 + q % TW;
                float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int j = k >> 2, i = k & 3;
                    float v = acc[0][pb][4 * j + i] + (&bi2[j].x)[i];
                    if (a.relu) v = fmaxf(v, 0.f);
                    const float r = __builtin_bit_cast(float, pack_bf16x2(v, 0.f) << 16);      // the value the bf16 store would have held
#pragma unroll
                    for (int c = 0; c < 4; ++c) part[c] = fmaf(r, w8[k][c], part[c]);
                }
                float lgv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) lgv[c] = (part[c] + __shfl_xor(part[c], 32)) + bl[c];
                if (g == 0 && q < NPIX && oy < a.Ho && ox < a.Wo) {
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
            return;
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int co0 = (cbg + cb) * MB + 4 * g;
            float4 bi[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bi[j] = *reinterpret_cast<const float4 *>(a.bias + co0 + 8 * j);
#pragma unroll
            for (int pb = 0; pb < PBW; ++pb) {
                const int q = (wn + pb * WN) * PB + pl;
                const int oy = oy0 + q / TW, ox = ox0 + q % TW;
                if (q < NPIX && oy < a.Ho && ox < a.Wo) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int chn = co0 + 8 * j;
                        unsigned short *o;
                        // channel-blocked bf16 storage: [N][C/16][H][W][16]
                        if (a.up2 == 0) {
                            if (chn >= cst) continue;
                            o = ob + ((((size_t)n * (cst >> 4) + (chn >> 4)) * a.Ho + oy) * a.Wo + ox) * 16 + (chn & 15);
                        } else {                      // sub-pixel scatter of a transposed conv: phase / channel of this 4-channel group
                            const int ph = chn / a.up2, co = chn % a.up2;
                            o = ob + ((((size_t)n * (a.up2 >> 4) + (co >> 4)) * (2 * a.Ho) + 2 * oy + (ph >> 1)) * (2 * a.Wo) + 2 * ox + (ph & 1)) * 16 + (co & 15);
                        }
                        float v0 = acc[cb][pb][4 * j + 0] + bi[j].x, v1 = acc[cb][pb][4 * j + 1] + bi[j].y;
                        float v2 = acc[cb][pb][4 * j + 2] + bi[j].z, v3 = acc[cb][pb][4 * j + 3] + bi[j].w;
                        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        uint2 pk; pk.x = pack_bf16x2(v0, v1); pk.y = pack_bf16x2(v2, v3);
                        *reinterpret_cast<uint2 *>(o) = pk;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int co0 = (cbg + cb) * MB + 4 * g;  // MB=32: + 8j ; MB=16: NJ = 1
        float4 bi[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bi[j] = *reinterpret_cast<const float4 *>(a.bias + co0 + 8 * j);
#pragma unroll
        for (int pb = 0; pb < PBW; ++pb) {
            const int q = (wn + pb * WN) * PB + pl;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            if (q < NPIX && oy < a.Ho && ox < a.Wo) {
                float *o;
                if (a.up2 == 0) {
                    o = a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co0;
                } else {                          // sub-pixel scatter of a transposed conv
                    const int ph = co0 / a.up2, co = co0 % a.up2;   // a 32-channel block never straddles phases
                    o = a.out + ((size_t)(n * 2 * a.Ho + 2 * oy + (ph >> 1)) * (2 * a.Wo) + 2 * ox + (ph & 1)) * a.up2 + co;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float4 v;
                    v.x = acc[cb][pb][4 * j + 0] + bi[j].x;
                    v.y = acc[cb][pb][4 * j + 1] + bi[j].y;
                    v.z = acc[cb][pb][4 * j + 2] + bi[j].z;
                    v.w = acc[cb][pb][4 * j + 3] + bi[j].w;
                    if (a.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<float4 *>(o + 8 * j) = v;
                }
            }
        }
    }
}

#ifdef UKBB_DIAG
#define UKBB_PBAR() { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __syncthreads(); st_pbar += __builtin_amdgcn_s_memtime() - t_; }
#define UKBB_PSTORE(X) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); X; st_pstore += __builtin_amdgcn_s_memtime() - t_; }
#else
#define UKBB_PBAR() __syncthreads()
#define UKBB_PSTORE(X) X
#endif

// ---------------------------------------------------------------------------
// Producer/consumer variant (512 threads, persistent over work items):
//   waves 0-3  "consumers": only ds_read + MFMA (+ the epilogue stores of finished items);
//   waves 4-7  "producers": all global loads and LDS writes, one stage (= one KC-channel chunk
//              of one item) ahead, into a double-buffered LDS.
// One barrier per stage is the only synchronisation: at barrier #s buffer s&1 holds stage s and
// buffer (s+1)&1 has been fully read (stage s-1), so producers may overwrite it.  The MFMA waves
// never wait on global memory, never issue an LDS write and never compute a staging address
// (r01 stamps: those phases cost the single-role kernel ~15 % of every workgroup's lifetime and
// co-resident workgroups did not hide them).  A work item is (Cout group, image, tile); items
// are walked group-major so concurrently running workgroups stream the same weight slab from L2.
// ---------------------------------------------------------------------------
// FIRST = true fuses the network's first layer (conv3x3, C_in = 1, BN, ReLU; reference
// common/network.py:186 with l = 0) into this conv: the producers evaluate it for every halo
// pixel directly from the 1-channel image (9 taps x KC channels on the vector ALU, same fma
// order as conv_first_kernel) and write the KC-channel tile to LDS, so that layer's output
// never exists in HBM.  Halo pixels outside the image are this conv's zero padding.
template <int KS, int STRIDE, int MB, int TH, int TW, int KC, int WM, int WN, int CB, bool FIRST = false, bool DI = false>
__global__ __launch_bounds__(512, 2) void conv_pc_kernel(const ConvArgs a) {
    using M = Mfma<MB>;
    using Acc = typename M::Acc;
    constexpr int KK = M::KK, KSTEPS = KC / KK, PB = MB;
    constexpr int NPIX = TH * TW, NPB = (NPIX + PB - 1) / PB, PBW = (NPB + WN - 1) / WN;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int HP = IH * IW, XS = xs_stride(KC), C4 = KC / 4, KS2 = KS * KS;
    constexpr int NCBL = WM * CB, SLAB = KS2 * 64 * KSTEPS;
    constexpr int NIT = (HP * C4 + 255) / 256, WF4 = NCBL * SLAB / 4, NWT = (WF4 + 255) / 256;
    constexpr int PSTEP = 256 / C4;
    constexpr int BUF = HP * XS + NCBL * SLAB;
    constexpr int RH = IH + 2, RW = IW + 2, RP = RH * RW, NRAW = (RP + 255) / 256;   // FIRST: raw image tile
    static_assert(WM * WN == 4, "4 consumer waves");
    // r06: stride-2 halo rows are staged DE-INTERLEAVED -- even halo columns first (TW + 1 of them), then the odd ones -- so that the
    // consumers' B fragment of tap (kh, kw), whose 32 lanes are consecutive OUTPUT pixels, reads consecutive LDS slots (80 bytes
    // apart: the eight lanes of a ds_read_b128 group cover all 32 banks) instead of every second one (160 bytes apart: four bank
    // quads, a 2-way conflict on every fragment read; profiles/r02_notes.md section 2).  UKBB_CONV_S2_INTERLEAVED=1 (environment, read per launch) keeps the old layout (A/B).
    // Measured at N = 64 (r06, tools/ab_libs.sh, three alternating rounds on one box): conv2_0 66.7 -> 65.0 us, conv3_0 62.8 -> 61.1, conv4_0
    // 59.3 -> 58.0, but conv1_0 (ONE 16-channel chunk per item: the producers' LDS writes, which now alternate between the two halves
    // of a row, are its critical path, not the consumers' reads) 69.3 -> 71.0: the launcher picks DI for layers of more than one chunk.
    constexpr bool DEINT = DI && STRIDE == 2 && KS == 3 && !FIRST;
    constexpr int NEVEN = (IW + 1) / 2;                  // even halo columns 0, 2, .., IW - 1 (IW = 2 TW + 1)
    auto slot_x = [](int px) { return DEINT ? ((px & 1) ? NEVEN + (px >> 1) : (px >> 1)) : px; };   // column of the staged row a halo column goes to

    extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x BUF

    const int nchunk = FIRST ? 1 : (a.C0 + a.C1) / KC;
    const int tiles = a.tiles_x * a.tiles_y;
    const int per_group = a.N * tiles;
    const int nitems = per_group * (a.Cout / (MB * NCBL));
    const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstages = my_items * nchunk;
    // Which half of the workgroup loads.  The SIMD's arbiter favours the older wave when two are ready (r02 A/B): the fused
    // first layer, whose loaders run conv0_0 on the vector ALU (~200 instructions per tile that must find issue slots next to
    // the MFMA stream), is 5 us faster with the loaders in waves 0-3; the other layers' loaders issue no VALU in steady state
    // and are 1.5-2 us faster the usual way round.
    constexpr bool PRODUCERS_FIRST = FIRST;
    const bool producer = PRODUCERS_FIRST ? __builtin_amdgcn_readfirstlane(threadIdx.x) < 256
                                          : __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    const int role_tid = threadIdx.x & 255;

    if (producer) {
        // ===================== producers: global -> registers -> LDS =====================
        const int tid = role_tid;
        const int c4 = tid % C4, pix0 = tid / C4;
        int item = blockIdx.x, ch = 0;
        if constexpr (FIRST) {
            // ---- fused first layer: raw 1-channel tile (halo of the halo) -> LDS -> conv0_0 -> xs ----
            // Stage k == item k of this workgroup (one chunk).  In the iteration after barrier #s the
            // producers (a) turn raw tile s+1 (LDS) into the KC-channel halo tile of stage s+1,
            // (b) park raw tile s+2 (registers, loaded one iteration ago) in LDS, (c) request raw tile s+3.
            // VALU-lean like the other producers (every VALU instruction here is time taken from the MFMA
            // waves): ReLU as one v_max_i32, per-thread LDS / global offsets computed once, raw pixels outside
            // the image as out-of-range buffer loads, and the halo-validity select only in tiles that touch the
            // image border.
            float *raw = lds + 2 * BUF;                        // [2][RP]
            // conv0_0 is a [16 x 9] x [9 x pixels] product: three v_mfma_f32_16x16x4_f32 per 16 halo pixels (taps 9..11 carry
            // zero weights), bias as the C operand; the D layout (lane = pixel, 4 consecutive channels per lane) is exactly
            // the float4 the halo tile stores.  30 instead of ~200 vector-ALU instructions per thread and tile.
            static_assert(KC == 16, "fused first layer: 16 channels");
            constexpr int NBLK = (HP + 15) / 16, NR = (NBLK + 3) / 4;      // 16-pixel blocks of the halo tile; rounds per producer wave
            const int lane_ = tid & 63, pw_ = tid >> 6, pj = lane_ & 15, pg = lane_ >> 4;
            float wA[3];                                       // A operand: W[cout = lane % 16][tap = 4 ks + lane / 16]
            int toff[3];                                       // raw-tile offset of that tap (B operand: lane % 16 = pixel, lane / 16 = k)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int t = 4 * ks + pg;
                wA[ks] = t < 9 ? a.first_w[t * KC + pj] : 0.f;
                const int tt = t < 9 ? t : 8;                  // zero weight: any finite value of the tile will do
                toff[ks] = (tt / 3) * RW + (tt % 3);
            }
            const f32x4 biasq = *reinterpret_cast<const f32x4 *>(a.first_b + 4 * pg);
            int hy[NR], hx[NR], roff[NR], xoff[NR];            // halo coordinates of this lane's pixel per round, raw / halo-tile offsets
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int pix = (pw_ + 4 * r) * 16 + pj;
                hy[r] = pix / IW; hx[r] = pix - hy[r] * IW;
                roff[r] = pix < HP ? hy[r] * RW + hx[r] : 0;
                xoff[r] = pix < HP ? pix * XS + 4 * pg : -1;
            }
            int ry_[NRAW], rx_[NRAW];
            unsigned rvo[NRAW];
#pragma unroll
            for (int k = 0; k < NRAW; ++k) {
                const int idx = tid + 256 * k;
                ry_[k] = idx / RW; rx_[k] = idx - ry_[k] * RW;
                rvo[k] = idx < RP ? (unsigned)(ry_[k] * a.W + rx_[k]) * 4u : 0x80000000u;
            }
            unsigned rr[NRAW];
            int n_ = 0, hy0_ = 0, hx0_ = 0;                    // image and halo-tile origin of `item`
            auto locate = [&](int it_item) {
                const int rest = it_item % per_group;
                n_ = rest / tiles;
                const int t = rest - n_ * tiles;
                const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
                hy0_ = ty * TH - a.pad_y; hx0_ = tx * TW - a.pad_x;
            };
            auto raw_load = [&](int it_item) {                 // global -> registers
                locate(it_item);
                const int ry0 = hy0_ - 1, rx0 = hx0_ - 1;
                const float *src = a.in0 + ((long long)(n_ * a.H + ry0) * a.W + rx0);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
                const int ylo = ry0 < 0 ? -ry0 : 0, yhi = a.H - ry0 < RH ? a.H - ry0 : RH;
                const int xlo = rx0 < 0 ? -rx0 : 0, xhi = a.W - rx0 < RW ? a.W - rx0 : RW;
                if (ylo == 0 && xlo == 0 && yhi == RH && xhi == RW) {
#pragma unroll
                    for (int k = 0; k < NRAW; ++k) rr[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, rvo[k], 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < NRAW; ++k) {
                        const bool ok = (unsigned)(ry_[k] - ylo) < (unsigned)(yhi - ylo) && (unsigned)(rx_[k] - xlo) < (unsigned)(xhi - xlo);
                        rr[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? rvo[k] : 0x80000000u, 0, 0);
                    }
                }
            };
            auto raw_store = [&](int b) {
#pragma unroll
                for (int k = 0; k < NRAW; ++k)
                    if (tid + 256 * k < RP) reinterpret_cast<unsigned *>(raw)[b * RP + tid + 256 * k] = rr[k];
            };
            auto first_layer = [&](int braw, int bx, int it_item) {   // raw tile (LDS) -> relu(conv0_0 + b) -> xs[bx]
                locate(it_item);
                // halo pixels outside the image are conv0_1's zero padding, not conv0_0 of padded input
                const int ylo = hy0_ < 0 ? -hy0_ : 0, yhi = a.H - hy0_ < IH ? a.H - hy0_ : IH;
                const int xlo = hx0_ < 0 ? -hx0_ : 0, xhi = a.W - hx0_ < IW ? a.W - hx0_ : IW;
                const bool interior = ylo == 0 && xlo == 0 && yhi == IH && xhi == IW;
                const float *rt = raw + braw * RP;
                float bvv[NR][3];
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) bvv[r][ks] = rt[roff[r] + toff[ks]];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if ((pw_ + 4 * r) * 16 >= HP) break;       // wave-uniform: this wave has no block in the last round
                    f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[0], bvv[r][0], biasq, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[1], bvv[r][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[2], bvv[r][2], acc, 0, 0, 0);
                    f32x4 v = {relu_bits(acc[0]), relu_bits(acc[1]), relu_bits(acc[2]), relu_bits(acc[3])};
                    if (!interior) {
                        const bool ok = (unsigned)(hy[r] - ylo) < (unsigned)(yhi - ylo) && (unsigned)(hx[r] - xlo) < (unsigned)(xhi - xlo);
                        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (xoff[r] >= 0) *reinterpret_cast<f32x4 *>(lds + bx * BUF + xoff[r]) = v;
                }
            };
            const int step = gridDim.x;
            if (nstages > 0) {
                raw_load(item);
                raw_store(0);                                  // R(0)
                if (nstages > 1) raw_load(item + step);        // R(1) in registers
            }
            // single Cout group (enforced by the launcher): this conv's weights are the same for every
            // item, so they are staged ONCE into both buffers instead of once per stage
            {
                const float *wp = a.wpk + (size_t)(item / per_group) * nchunk * (NCBL * SLAB);
#pragma unroll
                for (int it = 0; it < NWT; ++it) {
                    const int i4 = it * 256 + tid;
                    if (i4 < WF4) {
                        const f32x4 w = *reinterpret_cast<const f32x4 *>(wp + 4 * i4);
                        *reinterpret_cast<f32x4 *>(lds + HP * XS + 4 * i4) = w;
                        *reinterpret_cast<f32x4 *>(lds + BUF + HP * XS + 4 * i4) = w;
                    }
                }
            }
            __syncthreads();                                   // extra barrier: R(0) visible to all producers
            if (nstages > 0) {
                first_layer(0, 0, item);                       // stage 0 complete
                if (nstages > 1) raw_store(1);
                if (nstages > 2) raw_load(item + 2 * step);
            }
#ifdef UKBB_DIAG
            unsigned long long fp_bar = 0, fp_fl = 0, fp_rs = 0;
            const unsigned long long fp_t0 = __builtin_amdgcn_s_memtime();
#endif
            for (int s = 0; s < nstages; ++s) {
#ifdef UKBB_DIAG
                const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads();                               // barrier #s
#ifdef UKBB_DIAG
                const unsigned long long q1 = __builtin_amdgcn_s_memtime();
                fp_bar += q1 - q0;
#endif
                if (s + 1 < nstages) {
                    item += step;
                    first_layer((s + 1) & 1, (s + 1) & 1, item);
#ifdef UKBB_DIAG
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long q2 = __builtin_amdgcn_s_memtime();
                    fp_fl += q2 - q1;
#endif
                    if (s + 2 < nstages) raw_store(s & 1);     // R(s+2) replaces R(s)
                    if (s + 3 < nstages) raw_load(item + 2 * step);
#ifdef UKBB_DIAG
                    fp_rs += __builtin_amdgcn_s_memtime() - q2;
#endif
                }
            }
#ifdef UKBB_DIAG
            if ((a.diag & 16) && role_tid == 0) {
                unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(a.in1)) + (size_t)blockIdx.x * 16;
                o[8] = __builtin_amdgcn_s_memtime() - fp_t0; o[9] = fp_bar; o[10] = fp_rs; o[11] = fp_fl;
            }
#endif
        } else {
            // Lean producer (r01: fp32 MFMA and VALU instructions serialise on a SIMD, tools/mfma_coissue.hip,
            // so every VALU instruction here is time taken from the consumers).  Halo-tile coordinates and
            // byte offsets of this thread's float4s are computed once; per stage the scalar unit builds a
            // buffer descriptor at the tile origin whose range ends with the image, so halo rows below the image
            // fall out of range by themselves (the hardware returns 0 = this conv's zero padding, no select at
            // the LDS write); halo columns right of the image (last tile column only) are masked with lane masks
            // computed once (one v_cndmask per load); only tiles that touch the top / left padding (3x3 stride 1)
            // take the per-lane range test.
            // r02: loads run TWO stages ahead (two register sets): with one stage of lead the stride-2 layers
            // lost 19-20 us of 77-82 to load latency (ablation UKBB_CONV_DIAG=4, profiles/r02_notes.md).
#ifdef UKBB_DIAG
            unsigned long long st_pbar = 0, st_pstore = 0;
            const unsigned long long st_p0 = __builtin_amdgcn_s_memtime();
#endif
            int hy[NIT], hx[NIT];
            unsigned pre[NIT];
            bool rbad[NIT];                             // halo column beyond the image when the tile is in the last tile column
            u32x4 xq[2][NIT], wq[2][NWT];
            const int xhi_last = a.W - ((a.tiles_x - 1) * TW * STRIDE - a.pad_x);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pix = pix0 + it * PSTEP;
                hy[it] = pix / IW; hx[it] = pix - hy[it] * IW;
                pre[it] = 0x80000000u;
                rbad[it] = hx[it] >= xhi_last;
            }
            unsigned wvo[NWT];
#pragma unroll
            for (int it = 0; it < NWT; ++it) wvo[it] = it * 256 + tid < WF4 ? 16u * (it * 256 + tid) : 0x80000000u;
            int cur_cs = 0;
            int n_ = 0, iy0_ = 0, ix0_ = 0, grp_ = 0;
            auto locate = [&]() {
                grp_ = item / per_group;
                const int rest = item - grp_ * per_group;
                n_ = rest / tiles;
                const int t = rest - n_ * tiles;
                const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
                iy0_ = ty * TH * STRIDE - a.pad_y; ix0_ = tx * TW * STRIDE - a.pad_x;
            };
            int wkey0 = -1, wkey1 = -1;                 // (group, chunk) whose weights each LDS buffer holds
            bool wfresh[2] = {false, false};            // weights of the stage held in register set 0 / 1 need storing
            auto load = [&](auto setc, int b) {         // request the cursor stage into register set SET (destined for LDS buffer b)
                constexpr int SET = decltype(setc)::value;
                const float *src; int cs;
                if (ch * KC < a.C0) { src = a.in0 + ch * KC; cs = a.C0; }
                else                { src = a.in1 + (ch * KC - a.C0); cs = a.C1; }
                if (cs != cur_cs) {                     // uniform; once per source switch
                    cur_cs = cs;
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
                        pre[it] = pix0 + it * PSTEP < HP ? (unsigned)((hy[it] * a.W + hx[it]) * cs + 4 * c4) * 4u : 0x80000000u;
                }
                src += ((long long)(n_ * a.H + iy0_) * a.W + ix0_) * cs;      // may precede the tensor; such lanes are masked
#ifdef UKBB_DIAG
                if (a.diag & 4) return;                 // ablation: no global loads at all (LDS holds garbage)
#endif
                // range = from the tile origin to the end of THIS image: rows below the image read as zeros
                const long long to_end = ((long long)(a.H - iy0_) * a.W - ix0_) * cs * 4;
                const bool by_range = to_end > 0 && to_end < 0x7fffffffll;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, by_range ? (int)to_end : 0x7fffffff, 0x00020000);
                const bool top = iy0_ < 0, left = ix0_ < 0, right = ix0_ + IW > a.W, bottom = iy0_ + IH > a.H;
                if (!top && !left && !right && (by_range || !bottom)) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) xq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, pre[it], 0, 0);
                } else if (!top && !left && (by_range || !bottom)) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) xq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbad[it] ? 0x80000000u : pre[it], 0, 0);
                } else {
                    const int ylo = iy0_ < 0 ? -iy0_ : 0, yhi = a.H - iy0_ < IH ? a.H - iy0_ : IH;
                    const int xlo = ix0_ < 0 ? -ix0_ : 0, xhi = a.W - ix0_ < IW ? a.W - ix0_ : IW;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const bool ok = (unsigned)(hy[it] - ylo) < (unsigned)(yhi - ylo) && (unsigned)(hx[it] - xlo) < (unsigned)(xhi - xlo);
                        xq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? pre[it] : 0x80000000u, 0, 0);
                    }
                }
                // weights: skipped when this buffer already holds the slab of (group, chunk)
                const int key = grp_ * nchunk + ch;
                const int held = b ? wkey1 : wkey0;
                wfresh[SET] = key != held;
                if (wfresh[SET]) {
                    if (b) wkey1 = key; else wkey0 = key;
                    const float *wp = a.wpk + (size_t)key * (NCBL * SLAB);
                    const __amdgpu_buffer_rsrc_t ws_ = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, NCBL * SLAB * 4, 0x00020000);
#pragma unroll
                    for (int it = 0; it < NWT; ++it) wq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(ws_, wvo[it], 0, 0);
                }
            };
            float *const xs_w = lds + pix0 * XS + 4 * c4;
            float *const ws_w = lds + HP * XS + 4 * tid;
            int xs_de[DEINT ? NIT : 1];                 // de-interleaved rows: the staged slot of each of this thread's halo pixels
            if constexpr (DEINT) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) xs_de[it] = (hy[it] * IW + slot_x(hx[it])) * XS + 4 * c4;
            }
            auto store = [&](auto setc, int b) {        // register set SET -> LDS buffer b
                constexpr int SET = decltype(setc)::value;
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (pix0 + it * PSTEP < HP) {
                        if constexpr (DEINT) *reinterpret_cast<u32x4 *>(lds + b * BUF + xs_de[it]) = xq[SET][it];
                        else *reinterpret_cast<u32x4 *>(xs_w + b * BUF + it * PSTEP * XS) = xq[SET][it];
                    }
                if (wfresh[SET]) {
#pragma unroll
                    for (int it = 0; it < NWT; ++it)
                        if (it * 256 + tid < WF4) *reinterpret_cast<u32x4 *>(ws_w + b * BUF + it * 1024) = wq[SET][it];
                }
            };
            auto advance = [&]() {
                if (++ch == nchunk) { ch = 0; item += gridDim.x; if (item < nitems) locate(); }
            };
            constexpr std::integral_constant<int, 0> S0{};
            constexpr std::integral_constant<int, 1> S1{};
            // ---- straight-line pipeline for the stride-2 encoder layers (network.py:184-186) ----------------------
            // Loads run two stages ahead of the LDS buffer they fill.  That only pays if the `s_waitcnt vmcnt(n)`
            // hipcc places in front of the LDS writes counts exactly the loads issued since (n = the other register
            // set's loads); with ANY branch around a load it falls back to vmcnt(0) and the lead is lost (r02: the
            // first two-set version measured no gain).  So the steady state below issues the same loads in the same
            // order on every path: one source, no top/left padding (TF SAME with stride 2 on an even size pads
            // only after), rows below the image through the descriptor's range, the columns
            // right of it through an and-or with a per-tile scalar mask, weights either every stage or never.
            const bool straight = STRIDE == 2 && a.C1 == 0 && a.pad_y == 0 && a.pad_x == 0 && nstages >= 5 &&
                                  xhi_last >= IW - 1 &&     // at most the LAST halo column is beyond the image (tiles divide the map)
                                  (long long)a.H * a.W * a.C0 * 4 < 0x7fffffffll;
            if (straight) {
                // r02 stamps: with even a handful of vector-ALU instructions per stage the producer waves fell behind --
                // next to a dense fp32 MFMA stream a VALU instruction of another wave only issues in that stream's
                // bubbles -- and the MFMA waves waited 1.8 k cycles per stage at the barrier.  So the steady state
                // has none.  The only per-tile decision left is "is the last halo column beyond the image" (true
                // exactly in the last tile column): the float4s of that column are dealt to wave-instructions of
                // their own (the work list is: all other columns, padding to a multiple of 64 float4s, the last
                // column), and those instructions read through a descriptor whose range is 4 bytes when the column is
                // outside -- a scalar select.  Global and LDS byte offsets per thread are computed once.
                constexpr int MAIN_F4 = IH * (IW - 1) * C4, MAIN_PAD = (MAIN_F4 + 63) / 64 * 64, LAST_F4 = IH * C4;
                constexpr int NIT2 = (MAIN_PAD + LAST_F4 + 255) / 256;
                static_assert(NIT2 <= NIT + 1, "register budget of the straight-line producer");
                unsigned gofs[NIT2];
                char *lptr[NIT2];
                bool lastseg[NIT2];
                const int wave_f0 = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int f = it * 256 + tid;
                    int py = 0, px = 0, cq = 0;
                    bool ok = false;
                    if (f < MAIN_F4) { const int pp = f / C4; cq = f - pp * C4; py = pp / (IW - 1); px = pp - py * (IW - 1); ok = true; }
                    else if (f >= MAIN_PAD && f - MAIN_PAD < LAST_F4) { const int q = f - MAIN_PAD; py = q / C4; cq = q - py * C4; px = IW - 1; ok = true; }
                    gofs[it] = ok ? (unsigned)((py * a.W + px) * a.C0 + 4 * cq) * 4u : 0x80000000u;
                    lptr[it] = reinterpret_cast<char *>(lds) + (ok ? ((py * IW + slot_x(px)) * XS + 4 * cq) * 4 : KC * 4);   // idle lanes: the pad of halo pixel 0
                    lastseg[it] = it * 256 + wave_f0 >= MAIN_PAD;                                         // wave-uniform
                }
                u32x4 yq[2][NIT2];
                auto sl_load = [&](auto setc, auto wc) {
                    constexpr int SET = decltype(setc)::value;
                    constexpr bool WLOAD = decltype(wc)::value;
                    const float *src = a.in0 + ch * KC + ((long long)(n_ * a.H + iy0_) * a.W + ix0_) * a.C0;
                    const int to_end = ((a.H - iy0_) * a.W - ix0_) * a.C0 * 4;
                    const int to_end_last = ix0_ + IW > a.W ? 4 : to_end;      // last halo column outside the image: a range below every offset of that
                                                                                // column ((IW-1)*C0*4 bytes and up) makes its loads return zeros (0 would mean 'no range')
#pragma unroll
                    for (int it = 0; it < NIT2; ++it) {
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, lastseg[it] ? to_end_last : to_end, 0x00020000);
                        yq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(rs, gofs[it], 0, 0);
                    }
                    if constexpr (WLOAD) {
                        const float *wp = a.wpk + (size_t)(grp_ * nchunk + ch) * (NCBL * SLAB);
                        const __amdgpu_buffer_rsrc_t ws_ = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, NCBL * SLAB * 4, 0x00020000);
#pragma unroll
                        for (int it = 0; it < NWT; ++it) wq[SET][it] = __builtin_amdgcn_raw_buffer_load_b128(ws_, wvo[it], 0, 0);
                    }
                };
                auto sl_store = [&](auto setc, auto wc) {   // register set SET -> LDS buffer SET
                    constexpr int SET = decltype(setc)::value;
                    constexpr bool WLOAD = decltype(wc)::value;
#pragma unroll
                    for (int it = 0; it < NIT2; ++it)
                        *reinterpret_cast<u32x4 *>(lptr[it] + SET * BUF * 4) = yq[SET][it];
                    if constexpr (WLOAD) {
#pragma unroll
                        for (int it = 0; it < NWT; ++it)
                            if (it * 256 + tid < WF4) *reinterpret_cast<u32x4 *>(ws_w + SET * BUF + it * 1024) = wq[SET][it];
                    }
                };
                auto pipeline = [&](auto wc) {
                    locate();
                    sl_load(S0, wc);                           // stage 0
                    advance(); sl_load(S1, wc);                // stage 1
                    sl_store(S0, wc);
                    advance(); sl_load(S0, wc);                // stage 2
                    int s = 0;
                    do {                                       // in flight on entry: set 1 = stage s+1, set 0 = stage s+2
                        UKBB_PBAR();                       // barrier #s
                        UKBB_PSTORE(sl_store(S1, wc));
                        advance(); sl_load(S1, wc);            // stage s+3
                        UKBB_PBAR();                       // barrier #s+1
                        UKBB_PSTORE(sl_store(S0, wc));
                        advance(); sl_load(S0, wc);            // stage s+4
                        s += 2;
                    } while (s + 4 < nstages);
                    const bool more = s + 3 < nstages;         // 3 or 4 stages left: s+1, s+2 (in flight) and maybe s+3
                    UKBB_PBAR();                           // barrier #s
                    sl_store(S1, wc);
                    if (more) { advance(); sl_load(S1, wc); }
                    UKBB_PBAR();                           // barrier #s+1
                    sl_store(S0, wc);
                    UKBB_PBAR();                           // barrier #s+2
                    if (more) { sl_store(S1, wc); UKBB_PBAR(); }   // barrier #s+3
                };
                if (nchunk == 1 && a.Cout == MB * NCBL) {
                    // one weight slab for the whole layer: staged once into both buffers, never reloaded
#pragma unroll
                    for (int it = 0; it < NWT; ++it) {
                        const int i4 = it * 256 + tid;
                        if (i4 < WF4) {
                            const f32x4 w = *reinterpret_cast<const f32x4 *>(a.wpk + 4 * i4);
                            *reinterpret_cast<f32x4 *>(lds + HP * XS + 4 * i4) = w;
                            *reinterpret_cast<f32x4 *>(lds + BUF + HP * XS + 4 * i4) = w;
                        }
                    }
                    pipeline(std::false_type{});
                } else {
                    pipeline(std::true_type{});
                }
            } else {
                // generic pipeline (any stride / padding / two sources, short runs): same two register sets, but the
                // branches around its loads make hipcc wait for everything before each LDS write (one stage of lead)
                if (nstages > 0) {
                    locate();
                    load(S0, 0);                               // stage 0
                    if (nstages > 1) { advance(); load(S1, 1); }   // stage 1
                    store(S0, 0);
                    if (nstages > 2) { advance(); load(S0, 0); }   // stage 2 (set 0 is free again)
                }
                auto iter = [&](auto setc, int s) {     // after barrier #s: stage s+1 sits in register set SET = (s+1)&1
                    constexpr int SET = decltype(setc)::value;
                    if (s + 1 < nstages) {
                        UKBB_PSTORE(store(setc, SET));
                        if (s + 3 < nstages) { advance(); load(setc, SET); }
                    }
                };
#pragma unroll 1
                for (int s = 0; s < nstages; s += 2) {
                    UKBB_PBAR();                       // barrier #s
                    iter(S1, s);
                    if (s + 1 < nstages) {
                        UKBB_PBAR();                   // barrier #s+1
                        iter(S0, s + 1);
                    }
                }
            }
#ifdef UKBB_DIAG
            if ((a.diag & 16) && role_tid == 0) {
                unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(a.first_w)) + (size_t)blockIdx.x * 16;
                o[8] = __builtin_amdgcn_s_memtime() - st_p0; o[9] = st_pbar; o[10] = st_pstore; o[11] = straight ? 1 : 0;
            }
#endif
        }
    } else {
        // ===================== consumers: LDS -> MFMA -> global =====================
        // wave index through readfirstlane: hipcc must KNOW it is wave-uniform, or the output descriptor built from wm below
        // counts as divergent and every buffer store of the epilogue becomes a waterfall loop (4 v_readfirstlane + compares +
        // exec masking per store -- r02: 2.4 k cycles of epilogue per tile in the fused first layer, in the shadow of the other
        // workgroup's MFMA stream)
        const int tid = role_tid, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int g = lane / PB, pl = lane % PB;
        int lbase[PBW];
        unsigned ooff[PBW];                             // byte offset of this lane's first output float from the tile's first one
#pragma unroll
        for (int pb = 0; pb < PBW; ++pb) {
            int q = (wn + pb * WN) * PB + pl;
            ooff[pb] = q < NPIX ? (unsigned)(((q / TW) * a.Wo + (q % TW)) * a.Cout + 4 * g) * 4u : 0x80000000u;   // out of range: dropped
            if (q >= NPIX) q = 0;
            lbase[pb] = (((q / TW) * STRIDE) * IW + (q % TW) * (DEINT ? 1 : STRIDE)) * XS + KSTEPS * g;
        }
        int s = 0;
        if constexpr (FIRST) __syncthreads();          // matches the producers' raw-tile barrier
        // The folded-BN bias enters as the C operand of the first MFMA of every accumulator (no zeroing
        // moves, no bias adds in the epilogue); the tile is reloaded only when the Cout group changes.
        constexpr int NJ = M::NACC / 4;
        Acc biasT[CB];
        int cur_grp = -1;
#ifdef UKBB_DIAG
        unsigned long long st_wait = 0, st_mfma = 0, st_epi = 0;
        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            const int grp = item / per_group;
            if (grp != cur_grp) {
                cur_grp = grp;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const float *bp = a.bias + (grp * NCBL + wm * CB + cb) * MB + 4 * g;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const f32x4 bv4 = *reinterpret_cast<const f32x4 *>(bp + 8 * j);
#pragma unroll
                        for (int i = 0; i < 4; ++i) biasT[cb][4 * j + i] = bv4[i];
                    }
                }
                // Wait for the bias HERE (rare: once per Cout group).  Left to hipcc the wait lands in front of the first
                // MFMA of EVERY item as s_waitcnt vmcnt(0), which also waits for the previous item's output stores to be
                // acknowledged by memory -- once per stage for layers whose items are a single stage.
                __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), other counters untouched
            }
            Acc acc[CB][PBW];
            auto chunk = [&](auto firstc) {
                constexpr bool FIRSTCH = decltype(firstc)::value;
#ifdef UKBB_DIAG
                const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads();               // barrier #s: buffer s&1 holds this stage
#ifdef UKBB_DIAG
                const unsigned long long st_b = __builtin_amdgcn_s_memtime();
                st_wait += st_b - st_a;
#endif
                const float *xs = lds + (s & 1) * BUF;
                const float *wbase = xs + HP * XS + (wm * CB) * KS2 * 64 * KSTEPS + lane * KSTEPS;
                float av[2][CB][KSTEPS], bv[2][PBW][KSTEPS];
#define UKBB_LOAD_TAP(T, SET)                                                                       \
                {                                                                                  \
                    constexpr int kh_ = (T) / KS, kw_ = (T) % KS;                                  \
                    _Pragma("unroll") for (int cb = 0; cb < CB; ++cb)                              \
                        VecLoad<KSTEPS>::ld(wbase + (cb * KS2 + (T)) * 64 * KSTEPS, av[SET][cb]);  \
                    _Pragma("unroll") for (int pb = 0; pb < PBW; ++pb)                             \
                        VecLoad<KSTEPS>::ld(xs + lbase[pb] + (kh_ * IW + (DEINT ? ((kw_ & 1) ? NEVEN : kw_ / 2) : kw_)) * XS, bv[SET][pb]);  \
                }
#ifdef UKBB_DIAG
                if (a.diag & 2) {                       // ablation: no LDS reads / MFMAs, barriers only
                    if (FIRSTCH) {
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                            for (int pb = 0; pb < PBW; ++pb) acc[cb][pb] = biasT[cb];
                    }
                    ++s;
                    return;
                }
#endif
                UKBB_LOAD_TAP(0, 0)
                unroll_taps<KS2>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (t + 1 < KS2) UKBB_LOAD_TAP(t + 1, (t + 1) & 1)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks_ = 0; ks_ < KSTEPS; ++ks_)
#pragma unroll
                        for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) {
                                if (FIRSTCH && t == 0 && ks_ == 0)
                                    acc[cb][pb] = M::run(av[t & 1][cb][ks_], bv[t & 1][pb][ks_], biasT[cb]);
                                else
                                    acc[cb][pb] = M::run(av[t & 1][cb][ks_], bv[t & 1][pb][ks_], acc[cb][pb]);
                            }
                    __builtin_amdgcn_sched_barrier(0);
                });
#undef UKBB_LOAD_TAP
#ifdef UKBB_DIAG
                st_mfma += __builtin_amdgcn_s_memtime() - st_b;
#endif
                ++s;
            };
            chunk(std::true_type{});
#pragma unroll 1
            for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{});
            // ---- epilogue of this item (bias already inside the accumulators) ----
            const int rest = item - grp * per_group;
            const int n = rest / tiles, t = rest - n * tiles;
            const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
            const int oy0 = ty * TH, ox0 = tx * TW;
            const int cbg = grp * NCBL + wm * CB;
#ifdef UKBB_DIAG
            if (a.diag & 8) continue;                  // ablation: no epilogue
            const unsigned long long st_e = __builtin_amdgcn_s_memtime();
            struct StEpi { unsigned long long &acc, t0; __device__ ~StEpi() { acc += __builtin_amdgcn_s_memtime() - t0; } } st_epi_guard{st_epi, st_e};
#endif
            if (a.relu) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                        for (int r = 0; r < M::NACC; ++r) acc[cb][pb][r] = relu_bits(acc[cb][pb][r]);
            }
            if (a.up2 == 0 && oy0 + TH <= a.Ho && ox0 + TW <= a.Wo) {
                // whole tile inside the map (every tile of the FCN / U-Net shapes): the scalar unit builds a descriptor at
                // the tile's first output element, the per-lane byte offsets were computed once (ooff), the Cout block
                // and float4 index ride in the scalar offset -- no vector ALU work per store
                float *base = a.out + ((size_t)(n * a.Ho + oy0) * a.Wo + ox0) * a.Cout + cbg * MB;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            f32x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[cb][pb][4 * j + i];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, ooff[pb], (cb * MB + 8 * j) * 4, 0);
                        }
                continue;
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int co0 = (cbg + cb) * MB + 4 * g;
#pragma unroll
                for (int pb = 0; pb < PBW; ++pb) {
                    const int q = (wn + pb * WN) * PB + pl;
                    const int oy = oy0 + q / TW, ox = ox0 + q % TW;
                    if (q < NPIX && oy < a.Ho && ox < a.Wo) {
                        float *o;
                        if (a.up2 == 0) {
                            o = a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co0;
                        } else {
                            const int ph = co0 / a.up2, co = co0 % a.up2;
                            o = a.out + ((size_t)(n * 2 * a.Ho + 2 * oy + (ph >> 1)) * (2 * a.Wo) + 2 * ox + (ph & 1)) * a.up2 + co;
                        }
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            f32x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = acc[cb][pb][4 * j + i];
                            *reinterpret_cast<f32x4 *>(o + 8 * j) = v;
                        }
                    }
                }
            }
        }
#ifdef UKBB_DIAG
        if ((a.diag & 16) && role_tid == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(FIRST ? a.in1 : a.first_w)) + (size_t)blockIdx.x * 16;
            o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
            o[2] = st_wait; o[3] = st_mfma; o[4] = st_epi; o[5] = (unsigned long long)nstages;
            o[6] = st_r0; o[7] = __builtin_amdgcn_s_memrealtime();
        }
        if ((a.diag & 16) && lane == 0) {              // every consumer wave's own barrier wait and MFMA phase: who arrives last?
            unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(FIRST ? a.in1 : a.first_w)) + (size_t)blockIdx.x * 16;
            o[12 + wave] = (st_wait << 32) | (st_mfma & 0xffffffffull);
        }
#endif
    }
}

// ---------------------------------------------------------------------------
// Compiled tilings.  X(id, KS, STRIDE, MB, TH, TW, KC, WM, WN, CB)
// ---------------------------------------------------------------------------
#define UKBB_CONV_CONFIGS(X)                         \
    /* 3x3 stride 1 */                               \
    X(0, 3, 1, 16, 16, 52, 16, 1, 4, 1)              \
    X(10, 3, 1, 16, 8, 52, 16, 1, 4, 1)              \
    X(1, 3, 1, 16, 8, 26, 16, 2, 2, 1)               \
    X(2, 3, 1, 32, 24, 26, 16, 1, 4, 1)              \
    X(3, 3, 1, 16, 12, 26, 16, 2, 2, 1)              \
    X(4, 3, 1, 32, 12, 26, 16, 2, 2, 1)              \
    X(5, 3, 1, 16, 12, 13, 16, 2, 2, 1)              \
    X(6, 3, 1, 16, 12, 13, 16, 4, 1, 1)              \
    X(7, 3, 1, 16, 16, 16, 16, 1, 4, 1)              \
    X(8, 3, 1, 16, 16, 16, 16, 2, 2, 1)              \
    X(9, 3, 1, 32, 16, 16, 16, 2, 2, 1)              \
    X(11, 3, 1, 16, 8, 16, 16, 1, 4, 1)              \
    X(12, 3, 1, 16, 8, 26, 16, 1, 4, 1)              \
    X(13, 3, 1, 16, 12, 13, 16, 1, 4, 1)             \
    X(14, 3, 1, 32, 12, 13, 16, 2, 2, 1)             \
    X(15, 3, 1, 16, 6, 26, 16, 2, 2, 1)              \
    X(16, 3, 1, 16, 8, 13, 16, 2, 2, 1)              \
    X(17, 3, 1, 16, 16, 13, 16, 2, 2, 1)             \
    X(18, 3, 1, 16, 12, 13, 16, 2, 2, 2)             \
    X(19, 3, 1, 32, 12, 13, 16, 4, 1, 1)             \
    /* 3x3 stride 2 */                               \
    X(20, 3, 2, 16, 8, 26, 16, 2, 2, 1)              \
    X(21, 3, 2, 16, 12, 26, 8, 2, 2, 1)              \
    X(22, 3, 2, 32, 12, 26, 8, 2, 2, 1)              \
    X(23, 3, 2, 16, 12, 13, 16, 2, 2, 1)             \
    X(24, 3, 2, 16, 12, 13, 16, 4, 1, 1)             \
    X(25, 3, 2, 16, 8, 16, 16, 2, 2, 1)              \
    X(26, 3, 2, 16, 6, 13, 16, 2, 2, 1)              \
    X(27, 3, 2, 16, 12, 13, 8, 2, 2, 1)              \
    X(28, 3, 2, 32, 12, 13, 8, 2, 2, 1)              \
    X(29, 3, 2, 16, 8, 8, 16, 2, 2, 1)               \
    X(30, 3, 2, 16, 8, 13, 16, 2, 2, 1)              \
    X(31, 3, 2, 16, 12, 13, 16, 2, 2, 2)             \
    /* 1x1 */                                        \
    X(40, 1, 1, 16, 8, 52, 16, 2, 2, 1)              \
    X(41, 1, 1, 16, 12, 26, 16, 2, 2, 1)             \
    X(42, 1, 1, 16, 12, 13, 16, 2, 2, 1)             \
    X(43, 1, 1, 16, 16, 16, 16, 2, 2, 1)             \
    X(44, 1, 1, 16, 6, 13, 16, 2, 2, 1)              \
    X(45, 1, 1, 16, 4, 13, 16, 2, 2, 1)              \
    X(46, 1, 1, 16, 8, 16, 16, 2, 2, 1)              \
    /* 2x2 (transposed conv as sub-pixel conv) */    \
    X(60, 2, 1, 16, 12, 13, 16, 2, 2, 2)             \
    X(61, 2, 1, 16, 12, 13, 16, 2, 2, 1)             \
    X(62, 2, 1, 16, 8, 16, 16, 2, 2, 2)              \
    X(63, 2, 1, 16, 8, 16, 16, 2, 2, 1)

#define UKBB_CFG_ENTRY(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                   \
    {ID, KS, S, MB, TH, TW, KC, WM, WN, CB,                                                     \
     conv_lds_bytes(KS, S, MB, TH, TW, KC, WM, CB), 0,                                          \
     "conv" #KS "x" #KS "s" #S "_mb" #MB "_t" #TH "x" #TW "_kc" #KC "_w" #WM "x" #WN "_cb" #CB},
#define UKBB_PC_ENTRY(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                    \
    {ID, KS, S, MB, TH, TW, KC, WM, WN, CB,                                                     \
     2 * conv_lds_bytes(KS, S, MB, TH, TW, KC, WM, CB), 1,                                      \
     "convPC" #KS "x" #KS "s" #S "_mb" #MB "_t" #TH "x" #TW "_kc" #KC "_w" #WM "x" #WN "_cb" #CB},

// Producer/consumer tilings.  Y(id, KS, STRIDE, MB, TH, TW, KC, WM, WN, CB)
#define UKBB_PC_CONFIGS(Y)                           \
    Y(100, 3, 1, 32, 12, 26, 16, 2, 2, 1)            \
    Y(101, 3, 1, 16, 12, 13, 16, 4, 1, 1)            \
    Y(102, 3, 1, 32, 12, 13, 8, 4, 1, 1)             \
    Y(103, 3, 1, 16, 12, 26, 16, 2, 2, 1)            \
    Y(104, 3, 1, 16, 16, 16, 16, 1, 4, 1)            \
    Y(105, 3, 1, 16, 16, 26, 16, 1, 4, 1)            \
    Y(106, 3, 1, 32, 12, 26, 16, 1, 4, 1)            \
    Y(107, 3, 1, 16, 12, 13, 16, 2, 2, 2)            \
    Y(120, 3, 2, 16, 12, 13, 16, 2, 2, 1)            \
    Y(121, 3, 2, 16, 12, 13, 16, 4, 1, 1)            \
    Y(122, 3, 2, 32, 12, 26, 8, 2, 2, 1)             \
    Y(123, 3, 2, 16, 8, 16, 16, 2, 2, 1)             \
    Y(124, 3, 2, 16, 12, 13, 8, 2, 2, 2)             \
    Y(125, 3, 2, 16, 8, 26, 16, 2, 2, 1)             \
    Y(126, 3, 2, 16, 12, 13, 8, 4, 1, 2)             \
    Y(127, 3, 2, 16, 16, 13, 8, 2, 2, 2)             \
    Y(128, 3, 2, 16, 12, 13, 8, 4, 1, 1)             \
    Y(129, 3, 2, 16, 6, 13, 8, 4, 1, 2)              \
    Y(140, 3, 2, 32, 12, 13, 8, 4, 1, 1)             \
    Y(141, 3, 2, 16, 12, 13, 8, 2, 2, 1)             \
    Y(142, 3, 2, 16, 11, 13, 8, 2, 2, 2)             \
    Y(143, 3, 2, 16, 8, 13, 16, 2, 2, 1)             \
    Y(144, 3, 2, 16, 11, 13, 16, 2, 2, 1)            \
    Y(145, 3, 2, 16, 13, 16, 8, 2, 2, 2)

// bf16-operand tilings (single-role kernel, MB = 32, KC = 16).  B(id, KS, STRIDE, TH, TW, WM, WN, CB)
#define UKBB_BF_CONFIGS(B)                 \
    B(200, 3, 1, 12, 13, 2, 2, 1)          \
    B(201, 3, 1, 12, 26, 2, 2, 1)          \
    B(202, 3, 1, 16, 16, 1, 4, 1)          \
    B(203, 3, 1, 16, 16, 2, 2, 1)          \
    B(210, 3, 2, 12, 13, 2, 2, 1)          \
    B(211, 3, 2, 8, 16, 1, 4, 1)           \
    B(212, 3, 2, 8, 16, 2, 2, 1)           \
    B(220, 2, 1, 12, 13, 2, 2, 1)          \
    B(221, 2, 1, 16, 16, 2, 2, 1)

// The same tilings with bf16 activations in HBM on both sides (ConvConfig::pc == 5).
#define UKBB_BFIO_CONFIGS(B)               \
    B(230, 3, 1, 12, 13, 2, 2, 1)          \
    B(231, 3, 1, 12, 26, 2, 2, 1)          \
    B(232, 3, 1, 16, 16, 1, 4, 1)          \
    B(233, 3, 1, 16, 16, 2, 2, 1)          \
    B(234, 3, 1, 8, 16, 1, 4, 1)           \
    B(235, 3, 1, 8, 16, 2, 2, 1)           \
    B(236, 3, 1, 16, 32, 1, 4, 1)          \
    B(237, 3, 1, 8, 16, 2, 2, 2)           \
    B(238, 3, 1, 16, 16, 2, 2, 2)          \
    B(239, 3, 1, 16, 16, 1, 4, 2)          \
    B(290, 3, 1, 8, 16, 4, 1, 1)           \
    B(291, 3, 1, 8, 16, 4, 1, 2)           \
    B(292, 3, 1, 8, 32, 2, 2, 1)           \
    B(293, 3, 1, 16, 32, 2, 2, 1)          \
    B(240, 3, 2, 12, 13, 2, 2, 1)          \
    B(241, 3, 2, 8, 16, 1, 4, 1)           \
    B(242, 3, 2, 8, 16, 2, 2, 1)           \
    B(243, 3, 2, 16, 16, 1, 4, 1)          \
    B(244, 3, 2, 8, 16, 2, 2, 2)           \
    B(245, 3, 2, 8, 8, 2, 2, 1)            \
    B(246, 3, 2, 8, 16, 4, 1, 1)           \
    B(247, 3, 2, 16, 16, 2, 2, 1)          \
    B(248, 3, 2, 4, 16, 2, 2, 1)           \
    B(250, 2, 1, 12, 13, 2, 2, 1)          \
    B(251, 2, 1, 16, 16, 2, 2, 1)          \
    B(252, 2, 1, 16, 16, 1, 4, 1)          \
    B(253, 2, 1, 8, 16, 2, 2, 1)           \
    B(254, 2, 1, 8, 16, 2, 2, 2)           \
    B(255, 2, 1, 8, 16, 4, 1, 1)           \
    B(256, 2, 1, 8, 8, 2, 2, 1)            \
    B(257, 2, 1, 4, 16, 2, 2, 1)           \
    B(258, 2, 1, 8, 32, 2, 2, 1)

// Producer/consumer tilings with the fused first layer (C_in = 1 -> KC, then this conv).
#define UKBB_PCF_CONFIGS(Z)                          \
    Z(130, 3, 1, 16, 16, 16, 16, 1, 4, 1)            \
    Z(131, 3, 1, 16, 8, 16, 16, 1, 4, 1)             \
    Z(132, 3, 1, 16, 16, 26, 16, 1, 4, 1)            \
    Z(133, 3, 1, 16, 12, 16, 16, 1, 4, 1)

#define UKBB_PCF_ENTRY(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                   \
    {ID, KS, S, MB, TH, TW, KC, WM, WN, CB,                                                     \
     2 * conv_lds_bytes(KS, S, MB, TH, TW, KC, WM, CB) + 2 * 4 * (((TH - 1) * S + KS + 2) * ((TW - 1) * S + KS + 2)), 2, \
     "convPCfirst" #KS "x" #KS "s" #S "_mb" #MB "_t" #TH "x" #TW "_kc" #KC "_w" #WM "x" #WN "_cb" #CB},

__host__ __device__ constexpr int conv_bf_lds_bytes(int ks, int s, int th, int tw, int wm, int cb) {
    return ((16 / 2 + 4) * ((th - 1) * s + ks) * ((tw - 1) * s + ks) + wm * cb * ks * ks * 64 * 4) * 4;
}
#define UKBB_BF_ENTRY(ID, KS, S, TH, TW, WM, WN, CB)                                            \
    {ID, KS, S, 32, TH, TW, 16, WM, WN, CB, conv_bf_lds_bytes(KS, S, TH, TW, WM, CB), 3,       \
     "convBF16_" #KS "x" #KS "s" #S "_t" #TH "x" #TW "_w" #WM "x" #WN "_cb" #CB},

#define UKBB_BFIO_ENTRY(ID, KS, S, TH, TW, WM, WN, CB)                                          \
    {ID, KS, S, 32, TH, TW, 16, WM, WN, CB, conv_bf_lds_bytes(KS, S, TH, TW, WM, CB), 5,       \
     "convBF16io_" #KS "x" #KS "s" #S "_t" #TH "x" #TW "_w" #WM "x" #WN "_cb" #CB, 0},
// F(id, TH, TW, WN-by-4 tiling, FUSE): 3x3 stride 1, one 32-row Cout block (the 16-channel layers of level 0)
#define UKBB_BFIOF_CONFIGS(F)              \
    F(294, 16, 16, 1)                      \
    F(295, 16, 32, 1)                      \
    F(296, 8, 32, 1)                       \
    F(297, 16, 16, 2)                      \
    F(298, 16, 32, 2)                      \
    F(299, 8, 32, 2)
#define UKBB_BFIOF_ENTRY(ID, TH, TW, FUSE)                                                      \
    {ID, 3, 1, 32, TH, TW, 16, 1, 4, 1,                                                         \
     conv_bf_lds_bytes(3, 1, TH, TW, 1, 1) + ((FUSE) == 1 ? 4 * ((TH) + 4) * ((TW) + 4) : 0), 5,   /* + raw tile of the fused first layer */ \
     (FUSE) == 1 ? "convBF16io_first+3x3s1_t" #TH "x" #TW "_w1x4_cb1" : "convBF16io_3x3s1+logits_t" #TH "x" #TW "_w1x4_cb1", FUSE},

static const ConvConfig g_cfgs[] = {UKBB_CONV_CONFIGS(UKBB_CFG_ENTRY) UKBB_PC_CONFIGS(UKBB_PC_ENTRY)
                                    UKBB_PCF_CONFIGS(UKBB_PCF_ENTRY) UKBB_BF_CONFIGS(UKBB_BF_ENTRY) UKBB_BFIO_CONFIGS(UKBB_BFIO_ENTRY) UKBB_BFIOF_CONFIGS(UKBB_BFIOF_ENTRY)
                                    // Winograd F(2x2,3x3): region 4x8 tiles (8x16 px), 64 Cout per item, KC 16
                                    {300, 3, 1, 16, 8, 16, 16, 4, 1, 1, 125184, 4, "winogradF2x2_3x3_t8x16_kc16_cout64"},
                                    {301, 3, 1, 16, 8, 16, 16, 2, 1, 1, 125184, 4, "winogradF2x2_3x3_t8x16_kc16_cout32"},
                                    {302, 3, 1, 16, 16, 8, 16, 4, 1, 1, 125184, 4, "winogradF2x2_3x3_t16x8_kc16_cout64"},
                                    {303, 3, 1, 16, 16, 8, 16, 2, 1, 1, 125184, 4, "winogradF2x2_3x3_t16x8_kc16_cout32"},
                                    {304, 3, 1, 16, 8, 32, 16, 4, 1, 1, 152960, 4, "winogradF2x4_3x3_t8x32_kc16_cout64"},
                                    {305, 3, 1, 16, 8, 16, 16, 4, 1, 1, 87808, 4, "winogradF2x4_3x3_t8x16_kc16_cout64"},
                                    {306, 3, 1, 16, 8, 16, 16, 4, 1, 1, 95488, 4, "winogradF2x4_3x3_t8x16pair_kc16_cout64"},
                                    {307, 3, 1, 16, 8, 32, 16, 2, 1, 1, 152960, 4, "winogradF2x4_3x3_t8x32_kc16_cout32"}};

static constexpr int N_BASE_CFGS = (int)(sizeof(g_cfgs) / sizeof(g_cfgs[0]));
int num_conv_configs() { return N_BASE_CFGS + num_pk16_configs() + num_ws_configs(); }
const ConvConfig &conv_config(int i) {
    return i < N_BASE_CFGS ? g_cfgs[i] : i < N_BASE_CFGS + num_pk16_configs() ? pk16_config(i - N_BASE_CFGS) : ws_config(i - N_BASE_CFGS - num_pk16_configs());
}

hipError_t launch_conv(int cfg_id, const ConvArgs &a_in, hipStream_t s) {
    for (int i = 0; i < num_pk16_configs(); ++i)
        if (pk16_config(i).id == cfg_id) return launch_conv16_pk(cfg_id, a_in, s);
    for (int i = 0; i < num_ws_configs(); ++i)
        if (ws_config(i).id == cfg_id) return launch_conv_ws(cfg_id, a_in, s);
    ConvArgs a = a_in;
    const bool s2_interleaved = getenv("UKBB_CONV_S2_INTERLEAVED") != nullptr;     // A/B knob of the stride-2 halo layout (conv_pc_kernel DI)
#ifdef UKBB_DIAG
    { const char *e = getenv("UKBB_CONV_DIAG"); a.diag = e ? atoi(e) : 0; }
    static unsigned long long *d_stamps = nullptr;
    const char *scfg = getenv("UKBB_CONV_STAMP_CFG");
    const bool stamp = (a.diag & 16) && scfg && atoi(scfg) == cfg_id && (!a.first_w || !a.in1);   // fused-first kernel: in1 is free
    if (a.diag & 16) {
        if (!stamp) a.diag &= ~16;
        else {
            if (!d_stamps && hipMalloc(reinterpret_cast<void **>(&d_stamps), 1024 * 16 * 8) != hipSuccess) return hipErrorOutOfMemory;
            (void)hipMemsetAsync(d_stamps, 0, 1024 * 16 * 8, s);
            if (a.first_w) a.in1 = reinterpret_cast<const float *>(d_stamps);
            else a.first_w = reinterpret_cast<const float *>(d_stamps);
        }
    }
    struct StampDump {
        bool on; hipStream_t s; unsigned long long *d; int cfg;
        ~StampDump() {
            if (!on) return;
            static int shots = 0;
            if (++shots < 6) return;                      // let the clocks settle: report the 6th launch only
            if (shots > 6) return;
            std::vector<unsigned long long> h(1024 * 16);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double m[16] = {0}; int n = 0;
            for (int w = 0; w < 1024; ++w) if (h[w * 16]) { ++n; for (int k = 0; k < 16; ++k) m[k] += (double)h[w * 16 + k]; }
            if (!n) return;
            for (int k = 0; k < 16; ++k) m[k] /= n;
            unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
            for (int w = 0; w < 1024; ++w) if (h[w * 16]) {
                s0 = std::min(s0, h[w * 16 + 6]); s1 = std::max(s1, h[w * 16 + 6]);
                e0 = std::min(e0, h[w * 16 + 7]); e1 = std::max(e1, h[w * 16 + 7]);
            }
            fprintf(stderr, "[stamps cfg %d] workgroup start spread %.2f us, end spread %.2f us, first start -> last end %.2f us, mean lifetime %.2f us\n",
                    cfg, (s1 - s0) * 0.01, (e1 - e0) * 0.01, (e1 - s0) * 0.01, m[1] * 0.01);
            fprintf(stderr, "[stamps cfg %d] %d WGs: consumer total %.0f cyc (%.2f GHz), stages %.1f: per stage barrier-wait %.0f, mfma %.0f; epilogue total %.0f | "
                            "producer total %.0f, barrier-wait %.0f per stage, store(+vmcnt wait) %.0f per stage, straight %.0f\n",
                    cfg, n, m[0], m[0] / (m[1] * 10.0) , m[5], m[2] / m[5], m[3] / m[5], m[4], m[8], m[9] / m[5], m[10] / m[5], m[11]);
            double ww[4] = {0}, wm_[4] = {0};
            for (int w = 0; w < 1024; ++w) if (h[w * 16]) for (int k = 0; k < 4; ++k) { ww[k] += (double)(h[w * 16 + 12 + k] >> 32); wm_[k] += (double)(h[w * 16 + 12 + k] & 0xffffffffull); }
            fprintf(stderr, "[stamps cfg %d] per consumer wave (0..3), per stage: barrier-wait %.0f %.0f %.0f %.0f | mfma %.0f %.0f %.0f %.0f\n", cfg,
                    ww[0] / n / m[5], ww[1] / n / m[5], ww[2] / n / m[5], ww[3] / n / m[5], wm_[0] / n / m[5], wm_[1] / n / m[5], wm_[2] / n / m[5], wm_[3] / n / m[5]);
        }
    } stamp_dump{stamp, s, d_stamps, cfg_id};
#endif
    const ConvConfig *c = nullptr;
    for (const auto &e : g_cfgs) if (e.id == cfg_id) c = &e;
    if (!c) return hipErrorInvalidValue;
    const int group = c->mb * c->cb * c->wm;
    if (c->pc == 4) return is_wino24(*c) ? launch_wino24(a, c->tw, c->id == 306, c->wm, s) : launch_wino(a, c->wm, c->th / 2, s);
    if (a.in0_map) return hipErrorInvalidValue;       // image remapping exists in the Winograd kernel only
    if (c->pc == 2) {
        if (!a.first_w || !a.first_b || a.Cout != group) return hipErrorInvalidValue;
    } else if (a.Cout % group || (a.C0 + a.C1) % c->kc || a.C0 % c->kc) {
        return hipErrorInvalidValue;
    }
    dim3 grid((unsigned)(a.N * a.tiles_y * a.tiles_x), (unsigned)(a.Cout / group), 1);
    const long long nitems = (long long)a.N * a.tiles_y * a.tiles_x * (a.Cout / group);
    const int n_cu = device_cu_count();
    switch (cfg_id) {
#define UKBB_CFG_CASE(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                    \
    case ID: {                                                                                  \
        auto k = conv_mfma_kernel<KS, S, MB, TH, TW, KC, WM, WN, CB>;                           \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), c->lds_bytes, s, a);                             \
        break;                                                                                  \
    }
        UKBB_CONV_CONFIGS(UKBB_CFG_CASE)
#define UKBB_PC_CASE(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                     \
    case ID: {                                                                                  \
        auto k = conv_pc_kernel<KS, S, MB, TH, TW, KC, WM, WN, CB>;                             \
        if (S == 2 && KS == 3 && a.C0 + a.C1 > KC && !s2_interleaved) k = conv_pc_kernel<KS, S, MB, TH, TW, KC, WM, WN, CB, false, true>;   \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        const int per_cu = c->lds_bytes * 2 <= 160 * 1024 ? 2 : 1;                              \
        const long long cap = (long long)n_cu * per_cu;                                         \
        dim3 pgrid((unsigned)(nitems < cap ? nitems : cap), 1, 1);                              \
        hipLaunchKernelGGL(k, pgrid, dim3(512), c->lds_bytes, s, a);                            \
        break;                                                                                  \
    }
        UKBB_PC_CONFIGS(UKBB_PC_CASE)
#define UKBB_PCF_CASE(ID, KS, S, MB, TH, TW, KC, WM, WN, CB)                                    \
    case ID: {                                                                                  \
        auto k = conv_pc_kernel<KS, S, MB, TH, TW, KC, WM, WN, CB, true>;                       \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        const int per_cu = c->lds_bytes * 2 <= 160 * 1024 ? 2 : 1;                              \
        const long long cap = (long long)n_cu * per_cu;                                         \
        dim3 pgrid((unsigned)(nitems < cap ? nitems : cap), 1, 1);                              \
        hipLaunchKernelGGL(k, pgrid, dim3(512), c->lds_bytes, s, a);                            \
        break;                                                                                  \
    }
        UKBB_PCF_CONFIGS(UKBB_PCF_CASE)
#define UKBB_BF_CASE(ID, KS, S, TH, TW, WM, WN, CB)                                             \
    case ID: {                                                                                  \
        auto k = conv_mfma_kernel<KS, S, 32, TH, TW, 16, WM, WN, CB, true>;                     \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), c->lds_bytes, s, a);                             \
        break;                                                                                  \
    }
        UKBB_BF_CONFIGS(UKBB_BF_CASE)
#define UKBB_BFIO_CASE(ID, KS, S, TH, TW, WM, WN, CB)                                           \
    case ID: {                                                                                  \
        auto k = conv_mfma_kernel<KS, S, 32, TH, TW, 16, WM, WN, CB, true, true>;               \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), c->lds_bytes, s, a);                             \
        break;                                                                                  \
    }
        UKBB_BFIO_CONFIGS(UKBB_BFIO_CASE)
#define UKBB_BFIOF_CASE(ID, TH, TW, FUSE)                                                       \
    case ID: {                                                                                  \
        if ((FUSE) == 1 ? (!a.first_w || !a.first_b) : (!a.lg_w || !a.lg_b || a.lg_ncls < 2 || a.lg_ncls > 4)) return hipErrorInvalidValue; \
        auto k = conv_mfma_kernel<3, 1, 32, TH, TW, 16, 1, 4, 1, true, true, FUSE>;             \
        static OncePerDevice lds_ok;                                                            \
        {                                                                                       \
            hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), c->lds_bytes); \
            if (e != hipSuccess) return e;                                                      \
        }                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), c->lds_bytes, s, a);                             \
        break;                                                                                  \
    }
        UKBB_BFIOF_CONFIGS(UKBB_BFIOF_CASE)
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

size_t pack_conv_weights(const float *w, int ks, int cin, int cout, int mb, int kc, int ncbl, float *dst) {
    // dst[group][chunk][cbl][tap][lane][s] = W[tap][ci][co],  cb = group*ncbl + cbl
    //   k-step s of lane group g uses channel ci = chunk*kc + g*ksteps + s  (so the B operands
    //   of a lane for all k-steps of a tap are ksteps CONSECUTIVE channels = one LDS vector read)
    //   mb = 32: lane = (g<<5)|m (g < 2), co = cb*32 + m ;  mb = 16: lane = (g<<4)|m (g < 4), co = cb*16 + m
    // One workgroup (= one group of ncbl Cout blocks) reads, per chunk, one contiguous slab.
    const int kk = (mb == 32) ? 2 : 4, ksteps = kc / kk, ks2 = ks * ks, nchunk = cin / kc;
    size_t o = 0;
    for (int grp = 0; grp < cout / (mb * ncbl); ++grp)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cbl = 0; cbl < ncbl; ++cbl)
                for (int tap = 0; tap < ks2; ++tap)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane % mb, g = lane / mb;
                        for (int s = 0; s < ksteps; ++s) {
                            const int ci = ch * kc + g * ksteps + s, co = (grp * ncbl + cbl) * mb + m;
                            dst[o++] = w[((size_t)tap * cin + ci) * cout + co];
                        }
                    }
    return o;
}

static inline unsigned short f32_to_bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

size_t pack_conv_weights_bf16(const float *w, int ks, int cin, int cout, int ncbl, float *dst) {
    // dst[group][chunk][cbl][tap][lane][4 dwords]; dword d of lane (g<<5)|m holds bf16 pair
    //   W[tap][ci = chunk*16 + 8g + 2d (+1)][co = cb*32 + m]   (low half = even channel)
    const int ks2 = ks * ks, nchunk = cin / 16;
    unsigned *out = reinterpret_cast<unsigned *>(dst);
    size_t o = 0;
    for (int grp = 0; grp < cout / (32 * ncbl); ++grp)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cbl = 0; cbl < ncbl; ++cbl)
                for (int tap = 0; tap < ks2; ++tap)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 31, g = lane >> 5, co = (grp * ncbl + cbl) * 32 + m;
                        for (int d = 0; d < 4; ++d) {
                            const int ci = ch * 16 + 8 * g + 2 * d;
                            const unsigned lo = f32_to_bf16_rne(w[((size_t)tap * cin + ci) * cout + co]);
                            const unsigned hi = f32_to_bf16_rne(w[((size_t)tap * cin + ci + 1) * cout + co]);
                            out[o++] = lo | (hi << 16);
                        }
                    }
    return o;
}

// ---------------------------------------------------------------------------
// First layer (C_in = 1): direct 3x3 stencil on the vector ALU.  K = 9 is too
// thin for the matrix pipe and the layer is 0.4 % of the MACs.
// One thread = one pixel x all Cout; weights broadcast from LDS.
// ---------------------------------------------------------------------------
template <int COUT, bool OBF = false>
__global__ __launch_bounds__(256) void conv_first_kernel(const FirstArgs a) {
    __shared__ float wl[9 * COUT + COUT];
    for (int i = threadIdx.x; i < 9 * COUT; i += 256) wl[i] = a.w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) wl[9 * COUT + i] = a.bias[i];
    __syncthreads();
    const size_t total = (size_t)a.N * a.H * a.W;
    if constexpr (OBF) {
        // bf16 output: one thread = 8 channels of one pixel = one 16-byte store, consecutive lanes -> consecutive 16 bytes
        // (a thread per pixel would store 2 x 16 bytes at a 32-byte lane stride: half-used write bursts, 2.6 TB/s measured)
        for (size_t q2 = (size_t)blockIdx.x * 256 + threadIdx.x; q2 < 2 * total; q2 += (size_t)gridDim.x * 256) {
            const size_t q = q2 >> 1;
            const int c = (int)(q2 & 1) * 8;
            const int x = (int)(q % a.W);
            const int y = (int)((q / a.W) % a.H);
            const float *img = a.in + (q - (size_t)y * a.W - x);
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                v[t] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? img[(size_t)yy * a.W + xx] : 0.f;
            }
            float rp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) s = fmaf(v[t], wl[t * COUT + c + j], s);
                rp[j] = fmaxf(s + wl[9 * COUT + c + j], 0.f);
            }
            uint4 pk;
            pk.x = pack_bf16x2(rp[0], rp[1]); pk.y = pack_bf16x2(rp[2], rp[3]);
            pk.z = pack_bf16x2(rp[4], rp[5]); pk.w = pack_bf16x2(rp[6], rp[7]);
            *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(a.out) + q * COUT + c) = pk;
        }
        return;
    }
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int x = (int)(q % a.W);
        const int y = (int)((q / a.W) % a.H);
        const float *img = a.in + (q - (size_t)y * a.W - x);
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            v[t] = ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) ? img[(size_t)yy * a.W + xx] : 0.f;
        }
        float *o = a.out + q * COUT;
#pragma unroll
        for (int c = 0; c < COUT; c += 4) {
            float4 r;
            float *rp = &r.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) s = fmaf(v[t], wl[t * COUT + c + j], s);
                rp[j] = fmaxf(s + wl[9 * COUT + c + j], 0.f);
            }
            *reinterpret_cast<float4 *>(o + c) = r;
        }
    }
}

hipError_t launch_first(const FirstArgs &a, hipStream_t s) {
    const size_t total = (size_t)a.N * a.H * a.W * (a.out_bf16 ? 2 : 1);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 256u * 16u) grid = 256u * 16u;
    if (a.Cout == 16 && a.out_bf16) hipLaunchKernelGGL((conv_first_kernel<16, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.Cout == 16) hipLaunchKernelGGL((conv_first_kernel<16, false>), dim3(grid), dim3(256), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ukbb
