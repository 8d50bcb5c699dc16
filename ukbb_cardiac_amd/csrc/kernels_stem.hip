// Fused stem of the aortic U-Net in UKBB_PREC_BF16: conv0_0 (3x3, 1 -> 16, BN, ReLU) and conv0_1 (3x3, 16 -> 16, BN, ReLU)
// (reference common/network_ao.py:31-35 with l = 0 through network.py:19-25) in one launch: reads the fp32 image, writes the
// level-0 skip map (bf16, [N][H][W][16]); conv0_0's 256 x 256 x 16 output never exists in HBM.
//
// Same machine as kernels_tail.hip (r04): independent waves with a private LDS area, no barrier, v_mfma_f32_16x16x32_bf16 with M = the
// 16 output channels, weights in registers; a tile is R x 30 output pixels.
//   stage 0   conv0_0 on the (R + 2) x 32 pixels stage 1 needs.  K = 9 taps fit ONE MFMA per 16 pixels: k = 4 kh + kw (kw = 3 and
//             k >= 12 carry zero weights), so the B fragment of lane (pixel c, k-quarter kq) is FOUR CONSECUTIVE image pixels of row
//             r + kh -- quarter 0: rows kh = 0, 1, quarter 1: row kh = 2, quarters 2, 3: zeros.  The image halo lives in LDS as bf16 in four
//             copies shifted by 0..3 pixels, so that the window of ANY column starts 8-byte aligned in copy (c & 3): one ds_read_b64.
//             (The tile-per-workgroup kernel evaluated conv0_0 with three fp32 16x16x4 MFMAs per 16 pixels in chains: 64 us of its
//             141 us, r03 ablation.)  The image is rounded to bf16 (r03: Dice unchanged, 0.9928 / 0.9916).
//   mid       bias (C operand), ReLU, zero outside the image, bf16, into the wave's mid tile -- as kernels_tail.hip.
//   stage 1   conv0_1 on R x 32 pixels from the mid tile, two taps per MFMA (K = 32 = 2 x 16 channels), five MFMAs per 16 pixels.
//   store     bias, ReLU, bf16; lane (pixel, quarter kq) holds channels 4 kq .. 4 kq + 3: 8-byte stores, 512 contiguous bytes per
//             instruction.
#include "kernels.h"

#include <cstring>
#include <type_traits>

namespace ukbb {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(const u32x4 &a, const u32x4 &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_steps(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_steps<N, I + 1>(f); }
}

constexpr int ST_TW = 30, ST_MW = 32, ST_IW = 34;       // stored columns, stage-0 columns, image halo columns
constexpr int ST_RP = 40;                                // bf16 elements per row of an image copy (34 + shift + window overhang, 8-byte rows)

__host__ __device__ constexpr int st_copy_bytes(int r) { return (r + 4) * ST_RP * 2; }
__host__ __device__ constexpr int st_mplane(int r) { return ((r + 2) * ST_MW * 16 + 255) / 256 * 256; }
__host__ __device__ constexpr int st_wave_bytes(int r) { return (4 * st_copy_bytes(r) + 255) / 256 * 256 + 2 * st_mplane(r) + 256; }

template <int R, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void unet_stem_kernel(const StemArgs a) {
    constexpr int R1 = R + 2, HR0 = R + 4, HP0 = HR0 * ST_IW, NLD = (HP0 + 63) / 64, COPY = st_copy_bytes(R), MPLANE = st_mplane(R);
    constexpr int IMG = (4 * COPY + 255) / 256 * 256;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    unsigned char *const wl = lds + wave * st_wave_bytes(R);      // 4 image copies | mid [2 halves][R1 x 32] | pad
    unsigned char *const mid = wl + IMG;

    const int tiles_x = (a.W + ST_TW - 1) / ST_TW, tiles_y = (a.H + R - 1) / R, tiles = tiles_x * tiles_y, ntiles = a.N * tiles;
    const int nwalk = (int)gridDim.x, walker = (int)blockIdx.x;
#ifdef UKBB_TILE_ORDER_WAVE_MAJOR
    const int worker = wave * nwalk + walker, nworkers = nwalk * NW;
#else
    // Tiles that share halo columns / rows run at the same time on the same XCD: the NW waves of a workgroup take NW consecutive
    // tiles of a tile row, and the workgroups of one XCD (blockIdx % 8 on this chip's round-robin dispatch) take consecutive runs
    // of such groups, so a halo pixel is fetched from HBM once and re-read from that XCD's L2 (r04: the narrow tiles re-read 1.9x).
    const int per_xcd = nwalk / 8, xcd = walker & 7, slot = walker >> 3;
    const int chunk = (nwalk % 8 == 0) ? xcd * per_xcd + slot : walker;
    const int worker = chunk * NW + wave, nworkers = nwalk * NW;
#endif
    const int my = worker < ntiles ? (ntiles - worker + nworkers - 1) / nworkers : 0;
    if (my == 0) return;

    // ---- A fragments and biases into registers, once ----
    const u32x4 A0 = reinterpret_cast<const u32x4 *>(a.wA0)[lane];
    u32x4 A1[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) A1[p] = reinterpret_cast<const u32x4 *>(a.wA1)[p * 64 + lane];
    const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(a.b0 + 4 * kq), bias1 = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * kq);

    // ---- image staging: lane loads halo pixel p = lane + 64 i (one fp32), rounds it to bf16 and writes it into the four shifted copies:
    //      copy s holds halo pixel (y, x) at element y * ST_RP + x - s + 4 (the + 4 keeps x - s >= 0 at a positive index) ----
    unsigned geo[NLD], pbase[NLD], cofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = lane + 64 * i, px = p < HP0 ? p : HP0 - 1;
        const int hy = px / ST_IW, hx = px - hy * ST_IW;
        geo[i] = (unsigned)hy | ((unsigned)hx << 8);
        pbase[i] = (unsigned)((hy * a.W + hx) * 4);
        cofs[i] = (unsigned)((hy * ST_RP + hx + 4) * 2);
    }
    const unsigned char *const img = reinterpret_cast<const unsigned char *>(a.image);
    const int img_bytes = a.H * a.W * 4;

    struct Cur { int n, ty, tx; };
    const int st_n = nworkers / tiles, st_y = (nworkers % tiles) / tiles_x, st_x = (nworkers % tiles) % tiles_x;
    auto advance = [&](Cur &c) {
        c.tx += st_x; const int cx = c.tx >= tiles_x ? 1 : 0; c.tx -= cx * tiles_x;
        c.ty += st_y + cx; const int cy = c.ty >= tiles_y ? 1 : 0; c.ty -= cy * tiles_y;
        c.n += st_n + cy;
    };
    Cur cl, cc;
    cl.n = worker / tiles; cl.ty = (worker % tiles) / tiles_x; cl.tx = (worker % tiles) % tiles_x;
    cc = cl;
    float xq[NLD];
    auto request = [&](bool valid) {
        const int oy0 = cl.ty * R, ox0 = cl.tx * ST_TW;
        const int n = valid ? cl.n : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(img + (size_t)n * img_bytes), 0, valid ? img_bytes : 0, 0x00020000);
        const int toff = ((oy0 - 2) * a.W + (ox0 - 2)) * 4;
        unsigned vo[NLD];
        if (oy0 >= 2 && oy0 + R + 2 <= a.H && ox0 >= 2 && ox0 + ST_IW - 2 <= a.W) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) vo[i] = pbase[i] + (unsigned)toff;
        } else {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int hy = (int)(geo[i] & 0xffu), hx = (int)(geo[i] >> 8);
                const bool ok = (unsigned)(oy0 - 2 + hy) < (unsigned)a.H && (unsigned)(ox0 - 2 + hx) < (unsigned)a.W;
                vo[i] = ok ? pbase[i] + (unsigned)toff : OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) xq[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[i], 0, 0));
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x2 t2; t2.x = xq[i]; t2.y = 0.f;
            const unsigned short b = (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t2, bf16x2)) & 0xffffu);
#pragma unroll
            for (int sft = 0; sft < 4; ++sft) *reinterpret_cast<unsigned short *>(wl + sft * COPY + cofs[i] - 2 * sft) = b;
        }
    };

    // ---- per-lane LDS read bases ----
    // stage 0: lane (pixel column c = 16 blk + n16, quarter kq): copy c & 3 = n16 & 3, window start element (row) * ST_RP + c - (c & 3) + 4;
    //          quarter 0 reads rows kh = 0 (k 0..3) and 1 (k 4..7), quarter 1 row 2 (k 8..11) and nothing (k 12..15 = 0), quarters 2, 3 nothing
    const unsigned char *const s0_lane = wl + (n16 & 3) * COPY + (n16 - (n16 & 3) + 4) * 2 + (kq == 1 ? 2 * ST_RP * 2 : 0);
    const unsigned keep_lo = kq < 2 ? 0xffffffffu : 0u, keep_hi = kq == 0 ? 0xffffffffu : 0u;
    const unsigned char *s2_lane[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int t = 2 * p + (kq >> 1);
        const int tt = t < 9 ? t : 8;
        s2_lane[p] = mid + (kq & 1) * MPLANE + ((tt / 3) * ST_MW + (tt % 3) + n16) * 16;
    }
    unsigned char *const mid_w = mid + (kq >> 1) * MPLANE + n16 * 16 + (kq & 1) * 8;
    unsigned char *const outb = reinterpret_cast<unsigned char *>(a.out);
    const int out_img_bytes = a.H * a.W * 32;

    auto compute = [&]() {
        const int n = cc.n, oy0 = cc.ty * R, ox0 = cc.tx * ST_TW;
        // ---- stage 0: conv0_0, one MFMA per (row, block) ----
        f32x4 acc1[R1][2];
        {
            constexpr int S0 = R1 * 2, PD0 = 6, NB0 = 7;
            u32x2 Wl[NB0], Wh[NB0];                     // window rows kh (quarter 0: 0 | quarter 1: 2) and kh + 1 (quarter 0: 1)
            auto readW = [&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / 2, blk = s % 2;
                Wl[s % NB0] = *reinterpret_cast<const u32x2 *>(s0_lane + (r * ST_RP + 16 * blk) * 2);
                Wh[s % NB0] = *reinterpret_cast<const u32x2 *>(s0_lane + ((r + 1) * ST_RP + 16 * blk) * 2);
            };
            unroll_steps<PD0>([&](auto sc) { readW(sc); });
            unroll_steps<S0>([&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / 2, blk = s % 2;
                if constexpr (s + PD0 < S0) readW(std::integral_constant<int, s + PD0>{});
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 B = {Wl[s % NB0].x & keep_lo, Wl[s % NB0].y & keep_lo, Wh[s % NB0].x & keep_hi, Wh[s % NB0].y & keep_hi};
                acc1[r][blk] = mfma16(A0, B, bias0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- mid tile: ReLU, zero outside the image (only border tiles have such pixels), bf16 ----
        auto mid_store = [&](auto maskc) {
            constexpr bool MASK = decltype(maskc)::value;
            unroll_steps<R1>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const bool rowok = (unsigned)(oy0 - 1 + r) < (unsigned)a.H;
                unroll_steps<2>([&](auto bc) {
                    constexpr int blk = decltype(bc)::value;
                    const float e0 = acc1[r][blk][0], e1 = acc1[r][blk][1], e2 = acc1[r][blk][2], e3 = acc1[r][blk][3];
                    f32x2 lo2, hi2;
                    lo2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), 0)); lo2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), 0));
                    hi2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e2), 0)); hi2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e3), 0));
                    u32x2 pk;
                    pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2));
                    pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2));
                    if constexpr (MASK) {
                        const bool ok = rowok && (unsigned)(ox0 - 1 + 16 * blk + n16) < (unsigned)a.W;
                        const unsigned m = ok ? 0xffffffffu : 0u;
                        pk.x &= m; pk.y &= m;
                    }
                    *reinterpret_cast<u32x2 *>(mid_w + (r * ST_MW + 16 * blk) * 16) = pk;
                });
            });
        };
        if (oy0 >= 1 && oy0 + R + 1 <= a.H && ox0 >= 1 && ox0 + ST_MW - 1 <= a.W) mid_store(std::false_type{});
        else mid_store(std::true_type{});
        // ---- stage 1: conv0_1 on R rows x 2 blocks, two taps per MFMA ----
        f32x4 acc2[R][2];
        {
            constexpr int S2 = R * 2 * 5, PD2 = 8, NB2 = 9;
            u32x4 Bq[NB2];
            auto readB = [&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / 10, blk = (s / 5) % 2, p = s % 5;
                Bq[s % NB2] = *reinterpret_cast<const u32x4 *>(s2_lane[p] + (r * ST_MW + 16 * blk) * 16);
            };
            unroll_steps<PD2>([&](auto sc) { readB(sc); });
            unroll_steps<S2>([&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / 10, blk = (s / 5) % 2, p = s % 5;
                if constexpr (s + PD2 < S2) readB(std::integral_constant<int, s + PD2>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p == 0) acc2[r][blk] = mfma16(A1[p], Bq[s % NB2], bias1);
                else acc2[r][blk] = mfma16(A1[p], Bq[s % NB2], acc2[r][blk]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- store: ReLU, bf16, 8 bytes per lane (channels 4 kq ..) of pixel (row r, column 16 blk + n16) ----
        unsigned char *const obase = outb + (size_t)n * out_img_bytes;
        unroll_steps<R>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int oy = oy0 + r;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, oy < a.H ? out_img_bytes : 0, 0x00020000);
            unroll_steps<2>([&](auto bc) {
                constexpr int blk = decltype(bc)::value;
                const int c0 = 16 * blk + n16;
                const bool own = c0 < ST_TW && ox0 + c0 < a.W;
                const float e0 = acc2[r][blk][0], e1 = acc2[r][blk][1], e2 = acc2[r][blk][2], e3 = acc2[r][blk][3];
                f32x2 lo2, hi2;
                lo2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), 0)); lo2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), 0));
                hi2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e2), 0)); hi2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e3), 0));
                u32x2 pk;
                pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2));
                pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2));
                __builtin_amdgcn_raw_buffer_store_b64(pk, ro, own ? (unsigned)((oy * a.W + ox0 + c0) * 32 + kq * 8) : OOB, 0, 0);
            });
        });
    };

    // zero the copies once (window overhangs and the + 4 margin read them) and the pad
    for (int i = lane; i < (IMG + 2 * MPLANE + 256) / 16; i += 64) *reinterpret_cast<u32x4 *>(wl + i * 16) = u32x4{0u, 0u, 0u, 0u};
    request(true);
#pragma unroll 1
    for (int k = 0; k < my; ++k) {
        park();
        advance(cl);
        request(k + 1 < my);
        compute();
        advance(cc);
    }
}

}  // namespace

// A fragments (lane l: row l & 15, k = 8 (l >> 4) + j):
//   wA0 [64 lanes][4 dwords]: conv0_0 [3][3][1][16] folded: k = 4 kh + kw for kw < 3, kh < 3; zero elsewhere
static inline unsigned short st_bf16(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
void pack_stem_weights(const float *w0 /*[3][3][1][16] folded*/, float *dst0 /*64*4*/) {
    unsigned *o0 = reinterpret_cast<unsigned *>(dst0);
    for (int l = 0; l < 64; ++l)
        for (int d = 0; d < 4; ++d) {
            unsigned short h[2];
            for (int e = 0; e < 2; ++e) {
                const int m = l & 15, k = 8 * (l >> 4) + 2 * d + e, kh = k >> 2, kw = k & 3;
                h[e] = (kh < 3 && kw < 3) ? st_bf16(w0[(kh * 3 + kw) * 16 + m]) : (unsigned short)0;
            }
            o0[l * 4 + d] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
}

hipError_t launch_unet_stem(const StemArgs &a, hipStream_t s) {
    if (!a.image || !a.wA0 || !a.wA1 || !a.b0 || !a.b1 || !a.out || a.N < 1) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * 32 >= 0x7fffffffll) return hipErrorInvalidValue;
#if defined(UKBB_STEM_R)
    constexpr int R = UKBB_STEM_R, NW = UKBB_STEM_NW;   // A/B builds
#else
    constexpr int R = 8, NW = 4;
#endif
    const long long ntiles = (long long)a.N * ((a.H + R - 1) / R) * ((a.W + ST_TW - 1) / ST_TW);
    const int cus = device_cu_count();
    const long long want = (ntiles + NW - 1) / NW;
    // two workgroups per CU (240 registers, 57 KB of LDS each): two waves per SIMD cover each other's non-MFMA phases (101 -> 78 us at N = 100; r04)
    static const int per_cu = getenv("UKBB_STEM_WGS_PER_CU") ? atoi(getenv("UKBB_STEM_WGS_PER_CU")) : 2;   // A/B knob
    const long long cap = (long long)cus * (per_cu > 0 ? per_cu : 1);
    const int grid = (int)(want < cap ? want : cap);
    constexpr int bytes = NW * st_wave_bytes(R);
    static_assert(bytes <= 160 * 1024, "LDS");
    auto k = unet_stem_kernel<R, NW>;
    static OncePerDevice lds_ok;
    hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), bytes, s, a);
    return hipGetLastError();
}

}  // namespace ukbb
