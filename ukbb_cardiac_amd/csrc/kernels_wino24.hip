// Winograd F(2x4, 3x3) convolution (+ folded BN bias, ReLU): the 64-channel-group stride-1 3x3 layers of levels 2-3 of build_FCN
// (reference common/network.py:19-25, 186-188) on maps of at least 8 x 32 pixels.
//
//   Y = A_y^T [ sum_ci (G_y g G_x^T) .* (B_y^T d B_x) ] A_x       per 2 x 4 output tile, 4 x 6 input patch d
//
// rows through F(2,3) (entries 0, +-1, 1/2: exactly the transform of kernels_wino.hip), columns through F(4,3) (points 0, +-1, +-2, inf).
// 24 multiplies per 8 outputs instead of the 32 of F(2x2,3x3) and the 72 of the direct sum: the levels this runs on are bound by
// the MFMA stream of the consumer waves (r04 stamps: 4.1 k cycles of MFMA against 3.5-3.8 k of producer work per stage), so the
// quarter fewer MFMAs is time.  Why not F(4x4): its rounding error is 6-13x that of F(2x2) (numpy model, r04_notes.md); with the
// large factors of F(4,3) along ONE axis only the error stays at the level of a direct fp32 sum (rms 9e-8 of the activation scale
// against 3.6e-8 for F(2x2) and 3.1e-8 direct; max 6e-7 against 1.6e-7 direct at C = 64, 7e-7 / 4e-7 at C = 256).
//
// Structure as kernels_wino.hip (persistent producer / consumer workgroup of 512 threads, one barrier per stage):
//   item  = (group of 64 output channels, image, region of 4 x 8 tiles = 8 x 32 pixels)
//   stage = one chunk of 16 input channels of one item
//   producers (waves 4-7): global -> registers -> raw 10 x 34 halo tile XS (LDS) -> input transform -> VS[24][32 tiles][16] (LDS);
//   consumers (waves 0-3, one 16-channel block each): 24 GEMM positions x 2 tile blocks x 4 k-steps = 192 v_mfma_f32_16x16x4_f32 per
//       stage, A fragments straight from L2 through a rolling register queue, then the output transform, ReLU and NHWC stores.
// LDS: two raw tiles (pixel stride 20 floats: the 4-pixel tile step makes the patch reads conflict-free) and two transformed
// stages WITHOUT padding -- the 16-byte quad q of tile t sits at slot q ^ (3 * ((t >> 3) & 1)), which keeps the consumers' ds_read_b128
// conflict-free at a pixel stride of 16 floats (slot = q ^ 3 for tiles 8-15 of each 16-tile block): 153 KB of the 160.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace ukbb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TRY = 4;                                // tile rows per region (8 pixel rows)
constexpr int NK = 24;                                // GEMM positions: k = 6 i + j, i = row position 0..3, j = column position 0..5
constexpr int WKC = 16;                               // input channels per stage
constexpr int XS_S = 20;                              // raw tile: floats per pixel
// TBW = 16-tile MFMA column blocks per region: 2 -> 4 x 8 tiles = 8 x 32 pixels, 1 -> 4 x 4 tiles = 8 x 16 pixels (twice the items:
// maps whose 8 x 32 regions do not fill the CUs evenly, e.g. 24 x 26 at N = 64: 384 items on 256 CUs)
// PAIR (TBW = 1 only, maps with Ho = 8 m + 4, e.g. the 12 x 13 maps of level 4): the four left-over rows of an image would fill half a region.
// Images are processed in pairs instead: m regions of image A, one SEAM region whose tile rows 0-1 are A's last four rows and whose tile
// rows 2-3 are B's first four, then m regions of image B starting at its row 4 -- 2 m + 1 regions per pair with every tile slot in use
// along y.  The seam's raw tile has 12 rows: A's rows 8 m - 1 .. 8 m + 4 (the last one is padding, forced to zero although B's row 0 lies
// there in memory) and B's rows -1 .. 4 (the first one forced to zero although A's last row lies there).
template <int TBW, bool PAIR = false> struct W24 {
    static_assert(!PAIR || TBW == 1, "paired regions come in the 8 x 16 form only");
    static constexpr int NT = 16 * TBW, TRX = 4 * TBW;
    static constexpr int WIH = 2 * TRY + (PAIR ? 4 : 2), WIW = 4 * TRX + 2;   // raw halo tile: 10 x 34 / 10 x 18 / 12 x 18 pixels
    static constexpr int WHP = WIH * WIW;
    static constexpr int NITX = (WHP * (WKC / 4) + 255) / 256;        // staging passes of the 256 producer threads (6 / 3 / 4)
    // row pitch of the raw tile in pixels: the patch reads of a 16-lane ds_read_b128 group touch tiles of two tile rows when a tile row has
    // only 4 tiles; with a pitch of 24 pixels (two rows = a multiple of 256 bytes) their 64-byte pieces stay on different banks
    static constexpr int RS = TBW == 2 ? WIW : 24;
    static constexpr int XSZ = WIH * RS * XS_S + 32;                  // + a slot for the idle lanes of the last pass
    static constexpr int VSZ = NK * NT * WKC;                         // one transformed stage
    static constexpr int L_XS = 0;                                    // [2][XSZ]
    static constexpr int L_VS = L_XS + 2 * XSZ;                       // [2][24][NT][16]
    static constexpr int LDS_FLOATS = L_VS + 2 * VSZ;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");
};

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
#ifndef UKBB_LSTM_NT
#define UKBB_LSTM_NT 1
#endif
// streamed-once data of the ConvLSTM epilogue (gx, c): nontemporal forms (r05 A/B on one box, two alternating rounds: fp32 cine 18.79-18.86 -> 18.67-18.72 ms, bf16
// 13.07-13.09 -> 12.97; -DUKBB_LSTM_NT=0 gives the plain forms)
__device__ __forceinline__ f32x4 ld4_s(const float *p) {
#if UKBB_LSTM_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
#else
    return *reinterpret_cast<const f32x4 *>(p);
#endif
}
__device__ __forceinline__ float ld1_s(const float *p) {
#if UKBB_LSTM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st1_s(float *p, float v) {
#if UKBB_LSTM_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st4(float *p, const f32x4 &v) { *reinterpret_cast<f32x4 *>(p) = v; }

#ifdef UKBB_NO_PACKED_F32
// A/B form (r06): the same arithmetic as pairs of scalar VALU instructions -- the guide prices a packed f32 op
// beside MFMAs above two scalar ones; tools/ab_packed.sh measures it next to the fp32 MFMA streams of these kernels.
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return f32x2{s_add(a[0], b[0]), s_add(a[1], b[1])}; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return f32x2{s_sub(a[0], b[0]), s_sub(a[1], b[1])}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 s, f32x2 b) { return f32x2{s_fma(a[0], s[0], b[0]), s_fma(a[1], s[1], b[1])}; }
#else
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a * s + b with a scalar factor in both halves (v_pk_fma_f32; the factor is splat into a register pair by the caller)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 s, f32x2 b) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(s), "v"(b)); return r; }
#endif

struct V4 { f32x2 lo, hi; };                           // four channels as two packed pairs
__device__ __forceinline__ V4 operator+(const V4 &a, const V4 &b) { return V4{pk_add(a.lo, b.lo), pk_add(a.hi, b.hi)}; }
__device__ __forceinline__ V4 operator-(const V4 &a, const V4 &b) { return V4{pk_sub(a.lo, b.lo), pk_sub(a.hi, b.hi)}; }
__device__ __forceinline__ V4 fma4(const V4 &a, const f32x2 &s, const V4 &b) { return V4{pk_fma(a.lo, s, b.lo), pk_fma(a.hi, s, b.hi)}; }
__device__ __forceinline__ V4 to4(const f32x4 &v) { return V4{f32x2{v[0], v[1]}, f32x2{v[2], v[3]}}; }
__device__ __forceinline__ f32x4 from4(const V4 &v) { return f32x4{v.lo[0], v.lo[1], v.hi[0], v.hi[1]}; }

__device__ __forceinline__ float relu1(float x) { const int b = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, b > 0 ? b : 0); }

template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_k(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_k<N, I + 1>(f); }
}

// F(4,3) input transform of six values along a row (B_x^T d): 12 operations
//   r0 = 4 d0 - 5 d2 + d4      r1 = (d4 - 4 d2) + (d3 - 4 d1)     r2 = (d4 - 4 d2) - (d3 - 4 d1)
//   r5 = 4 d1 - 5 d3 + d5      r3 = (d4 - d2) + 2 (d3 - d1)       r4 = (d4 - d2) - 2 (d3 - d1)
__device__ __forceinline__ void bx_transform(const V4 (&d)[6], V4 (&r)[6]) {
    const f32x2 c4 = {4.f, 4.f}, cm4 = {-4.f, -4.f}, cm5 = {-5.f, -5.f}, c2 = {2.f, 2.f}, cm2 = {-2.f, -2.f};
    r[0] = fma4(d[2], cm5, fma4(d[0], c4, d[4]));
    r[5] = fma4(d[3], cm5, fma4(d[1], c4, d[5]));
    const V4 a = fma4(d[2], cm4, d[4]), b = fma4(d[1], cm4, d[3]);
    r[1] = a + b; r[2] = a - b;
    const V4 c = d[4] - d[2], e = d[3] - d[1];
    r[3] = fma4(e, c2, c); r[4] = fma4(e, cm2, c);
}

}  // namespace

// ---- ConvLSTM cell pieces (tf.contrib.rnn.Conv2DLSTMCell): v_exp_f32 / v_rcp_f32 (1 ulp each) ----
namespace {
__device__ __forceinline__ float ls_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }
__device__ __forceinline__ float ls_tanh(float x) {      // 1 - 2 / (e^2x + 1): saturates cleanly at both ends (rcp(inf) = 0)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.88539008f) + 1.0f), 1.0f);
}
// gates g = (i, j, f, o) of one hidden channel of one pixel, old cell state c -> new cell state (returned through c) and hidden state
__device__ __forceinline__ float ls_cell(const f32x4 &g, float &c, float forget_bias) {
    const float sf = ls_sigmoid(g[2] + forget_bias), si = ls_sigmoid(g[0]), tj = ls_tanh(g[1]);
    c = __builtin_fmaf(sf, c, si * tj);
    return ls_tanh(c) * ls_sigmoid(g[3]);
}
}  // namespace

// Diagnostic only (-DUKBB_WINO_STAMPS + UKBB_STAMPS=1): s_memtime stamps around the phases of a stage (perturbs the MFMA stream)
#ifdef UKBB_WINO_STAMPS
#define STAMP(v) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); }
#define STAMP_DO(...) __VA_ARGS__
__device__ unsigned long long g_w24stamps[16];
#else
#define STAMP(v)
#define STAMP_DO(...)
#endif
// NCB = 16-channel blocks per item: 4 (64 output channels, consumer wave w = block w, all tile blocks) or 2 (32 output channels, TBW = 2 only:
// wave w = block w & 1 of tile block w >> 1 -- layers with 32 output channels, which are bound by the producers in the F(2x2) kernel)
// LS: 0 = plain conv; 1 | 2 = ConvLSTM cell in the epilogue (ConvArgs::ls_mode; the producers and the MFMA phase are the plain ones)
// BF (LS != 0 only; UKBB_PREC_BF16 on a UNet-LSTM handle): source 0, gx and the hidden maps are bf16 in HBM (the arithmetic stays fp32
// Winograd on the widened values, the cell state stays fp32): the time steps are bound by their bytes, not by the matrix pipe
template <int TBW, bool PAIR, int NCB, int LS = 0, bool BF = false>
__global__ __launch_bounds__(512) void wino24_pc_kernel(const ConvArgs a) {
    static_assert(!BF || LS != 0, "bf16 storage exists in the ConvLSTM forms only");
    using G = W24<TBW, PAIR>;
    static_assert(NCB == 4 || (NCB == 2 && TBW == 2 && !PAIR), "32-channel items come with 8 x 32-pixel regions");
    static_assert(LS == 0 || (NCB == 4 && !PAIR), "the ConvLSTM epilogue needs the four gate blocks of a hidden channel quad in one wave");
    constexpr int CTB = NCB == 2 ? 1 : TBW;             // tile blocks per consumer wave
    constexpr int NT = G::NT, TRX = G::TRX, WIH = G::WIH, WIW = G::WIW, WHP = G::WHP, NITX = G::NITX, RS = G::RS, XSZ = G::XSZ, VSZ = G::VSZ, L_XS = G::L_XS, L_VS = G::L_VS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cin = a.C0 + a.C1;
    const int nchunk = LS != 0 ? 1 : cin / WKC;        // ConvLSTM forms: 16 channels in, one stage per item (a compile-time fact for the loops below)
    const int regs_x = (a.Wo + 4 * TRX - 1) / (4 * TRX);
    // PAIR: `regions` counts the regions of an image PAIR (2 m + 1 rows of them), `a.N` images are (N + 1) / 2 pairs
    const int pm = a.Ho / 8;                            // PAIR: m
    const int regs_y = PAIR ? 2 * pm + 1 : (a.Ho + 2 * TRY - 1) / (2 * TRY);
    const int regions = regs_x * regs_y;
    const int per_group = (PAIR ? (a.N + 1) / 2 : a.N) * regions;
    const int nitems = per_group * (a.Cout / (16 * NCB));
    const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstages = my_items * nchunk;
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;

    if (producer) {
        const int tid = threadIdx.x - 256;
        constexpr int C4 = WKC / 4, PSTEP = 256 / C4;                 // 64 halo pixels per pass, 6 passes
        const int c4 = tid % C4, pix0 = tid / C4;
        // In-image test of a halo pixel: its row bit and column bit (10 + 34 bits) against the region's 64-bit mask
        unsigned tb_lo[NITX], tb_hi[NITX], pixoff[NITX], pre[NITX], xoff[NITX];
        unsigned pixoff_s[PAIR ? NITX : 1];
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int pix = pix0 + it * PSTEP;
            const int iy = pix / WIW, ix = pix - iy * WIW;
            const unsigned long long m = pix < WHP ? (1ull << iy) | (1ull << (WIH + ix)) : (1ull << 63);      // bit 63 is never set in a region mask
            tb_lo[it] = (unsigned)m; tb_hi[it] = (unsigned)(m >> 32);
            pixoff[it] = (unsigned)(iy * a.W + ix);
            if constexpr (PAIR) pixoff_s[it] = (unsigned)((iy - (iy >= 6 ? 2 : 0)) * a.W + ix);    // seam: B's row -1 is the flat row behind A's row 8 m + 3
            pre[it] = 0;
            xoff[it] = pix < WHP ? (unsigned)((iy * RS + ix) * XS_S + 4 * c4) : (unsigned)(XSZ - 32 + 4 * c4);
        }
        int cur_cs = 0;
        u32x4 xr[NITX];
        int l_ch = 0, l_item = blockIdx.x, l_n = 0, l_n0 = 0, l_iy0 = 0, l_ix0 = 0;
        bool l_seam = false;                            // PAIR: the cursor's region is a seam region
        auto locate = [&]() {
            const int rest = l_item % per_group;
            l_n = rest / regions;
            const int r = rest - l_n * regions;
            const int ry = r / regs_x, rx = r - ry * regs_x;
            l_iy0 = ry * 2 * TRY - 1; l_ix0 = rx * 4 * TRX - 1;
            if constexpr (PAIR) {                       // l_n = pair index so far
                l_seam = ry == pm;
                if (ry <= pm) l_n = 2 * l_n;                                     // image A (the seam's raw tile starts in A)
                else { l_n = 2 * l_n + 1; l_iy0 = 4 + (ry - pm - 1) * 8 - 1; }   // image B, regions from its row 4
            }
            // ConvLSTM windows index shared feature frames (source 0 only).  readfirstlane: the loaded index arrives in a VGPR, and without it the
            // buffer descriptors built from it count as divergent -- every halo load became a waterfall loop (kernels_wino.hip, r02)
            l_n0 = (!PAIR && a.in0_map) ? __builtin_amdgcn_readfirstlane(a.in0_map[l_n]) : l_n;
        };
        locate();
        // The in-image test and the byte offsets of a thread's pieces depend on the region and on the source's channel count only: they are
        // recomputed when the cursor enters a new item or switches source, not per stage -- every vector instruction of a producer waits
        // for a gap in the co-resident consumer's MFMA stream (r02: ~28-60 cycles each), and the transform below already needs ~75.
        unsigned vo[NITX];
        auto loadx = [&]() {                            // raw halo of the cursor stage -> registers; advances the cursor
            const float *src; int cs;
            const bool from0 = l_ch * WKC < a.C0;
            if (from0) { src = a.in0 + l_ch * WKC; cs = a.C0; }
            else       { src = a.in1 + (l_ch * WKC - a.C0); cs = a.C1; }
            if (l_ch == 0 || cs != cur_cs) {            // uniform
                if (!PAIR && cs != cur_cs) {
                    cur_cs = cs;
#pragma unroll
                    for (int it = 0; it < NITX; ++it) pre[it] = BF ? pixoff[it] * (unsigned)(cs * 2) + 8u * c4 : pixoff[it] * (unsigned)(cs * 4) + 16u * c4;
                }
                if constexpr (PAIR) {                   // the seam's lower six raw rows lie two flat rows earlier: offsets per item
                    cur_cs = cs;
#pragma unroll
                    for (int it = 0; it < NITX; ++it) pre[it] = (l_seam ? pixoff_s[it] : pixoff[it]) * (unsigned)(cs * 4) + 16u * c4;
                }
                constexpr int NROW = PAIR ? 10 : WIH;   // rows of an ordinary region's raw tile (the seam uses all 12)
                const int ylo = l_iy0 < 0 ? -l_iy0 : 0, yhi = a.H - l_iy0 < NROW ? a.H - l_iy0 : NROW;
                const int xlo = l_ix0 < 0 ? -l_ix0 : 0, xhi = a.W - l_ix0 < WIW ? a.W - l_ix0 : WIW;
                unsigned long long rowm = ((1ull << yhi) - 1ull) & ~((1ull << ylo) - 1ull);
                if constexpr (PAIR) {
                    if (l_seam) {                       // A: rows 8 m - 1 .. 8 m + 3 (raw 0..4; raw 5 = padding), B: rows 0 .. 4 (raw 7..11; raw 6 = padding) if B exists
                        rowm = (l_iy0 < 0 ? 0x1eull : 0x1full) | (l_n + 1 < a.N ? (a.H > 4 ? 0xf80ull : 0x780ull) : 0ull);   // B's row 4 exists unless H = 4
                    } else if (l_n >= a.N) rowm = 0;    // the B half of a last, single image: nothing to read (and nothing stored)
                }
                const unsigned long long cm = rowm | ((((1ull << xhi) - 1ull) & ~((1ull << xlo) - 1ull)) << WIH);
                const unsigned cm_lo = (unsigned)cm, cm_hi = (unsigned)(cm >> 32);
#pragma unroll
                for (int it = 0; it < NITX; ++it) {
                    const bool ok = (cm_lo & tb_lo[it]) == tb_lo[it] && (cm_hi & tb_hi[it]) == tb_hi[it];
                    vo[it] = ok ? pre[it] : 0x80000000u;    // out of range -> zeros
                }
            }
            if constexpr (BF) {                         // 16 bf16 channels per pixel: 8 bytes per thread, widened in storex
                const unsigned short *sb = reinterpret_cast<const unsigned short *>(a.in0) + l_ch * WKC + ((long long)(l_n0 * a.H + l_iy0) * a.W + l_ix0) * cs;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sb, 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int it = 0; it < NITX; ++it) {
                    const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, vo[it], 0, 0);
                    xr[it][0] = d[0]; xr[it][1] = d[1];
                }
            } else {
            src += ((long long)((from0 ? l_n0 : l_n) * a.H + l_iy0) * a.W + l_ix0) * cs;   // may point before the tensor; masked lanes never use it
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int it = 0; it < NITX; ++it) xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[it], 0, 0);
            }
            if (++l_ch == nchunk) { l_ch = 0; l_item += gridDim.x; if (l_item < nitems) locate(); }
        };
        auto storex = [&](auto par) {
            constexpr int B = decltype(par)::value;
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                u32x4 v = xr[it];
                if constexpr (BF) { const unsigned d0 = xr[it][0], d1 = xr[it][1]; v = u32x4{d0 << 16, d0 & 0xffff0000u, d1 << 16, d1 & 0xffff0000u}; }
                *reinterpret_cast<u32x4 *>(lds + L_XS + B * XSZ + xoff[it]) = v;
            }
        };
        // V = B_y^T d B_x per (tile, channel quad).  The producer waves split the row positions i = 0..3 of B_y^T d
        //   (d0 - d2, d1 + d2, d2 - d1, d1 - d3):
        // TBW = 2: wave pairs -- waves 4-5 produce i = 0, 1 (from patch rows 0-2), waves 6-7 i = 2, 3 (rows 1-3): 18 reads, two F(4,3)
        //          column transforms and 12 writes per thread;
        // TBW = 1: one position per wave (rows {0,2}, {1,2}, {1,2}, {1,3}): 12 reads, one column transform, 6 writes.
        constexpr int PARTS = 4 / TBW, PP = TBW;        // parts of the producer half of the workgroup, row positions per part
        const int part = __builtin_amdgcn_readfirstlane(tid) / (256 / PARTS);
        const int xt = tid % (256 / PARTS), x_tile = xt / C4, x_q = xt - x_tile * C4;
        // first patch row this thread reads and the distance to its second one (TBW = 1)
        const int row0 = TBW == 2 ? part : (part == 0 ? 0 : 1), rstep = TBW == 2 ? 1 : (part == 0 ? 2 : part == 3 ? 2 : 1);
        const float *const xs_r = lds + L_XS + ((2 * (x_tile / TRX) + row0) * RS + 4 * (x_tile % TRX)) * XS_S + 4 * x_q;
        // transformed stage: [k][tile][16] floats, quad q of tile t in slot q ^ (3 * ((t >> 3) & 1)): a consumer ds_read_b128's 16-lane
        // group holds tiles {0-3, 12-15} of one k-quarter g and {4-11} of g + 1 -- with the flip all sixteen 16-byte slots differ
        float *const vs_w = lds + L_VS + part * (6 * PP) * NT * WKC + x_tile * WKC + 4 * (x_q ^ (3 * ((x_tile >> 3) & 1)));
        constexpr int NR = TBW == 2 ? 3 : 2;            // patch rows a thread reads
        V4 e[NR][6];
        int x_stage = 0;                                // PAIR: stage the next xform_read works on -> is its region a seam?
        auto xform_read = [&](auto par) {
            constexpr int B = decltype(par)::value;
            const float *xs = xs_r + B * XSZ;
            if constexpr (PAIR) {
                const int it_ = (int)blockIdx.x + (x_stage / nchunk) * (int)gridDim.x;
                const int ry_ = ((it_ % per_group) % regions) / regs_x;
                if (ry_ == pm && x_tile / TRX >= 2) xs += 2 * RS * XS_S;
                ++x_stage;
            }
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) e[i][j] = to4(ld4(xs + (i * rstep * RS + j) * XS_S));
        };
        auto xform_finish = [&](auto par) {
            constexpr int B = decltype(par)::value;
            float *vs = vs_w + B * VSZ;
            if constexpr (TBW == 2) {
                auto rows = [&](auto hc) {              // a real (uniform) branch per wave pair
                    constexpr int HALF = decltype(hc)::value;
                    V4 wa[6], wb[6], ra[6], rb[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        if constexpr (HALF == 0) { wa[j] = e[0][j] - e[2][j]; wb[j] = e[1][j] + e[2][j]; }   // i = 0: d0 - d2, i = 1: d1 + d2
                        else                     { wa[j] = e[1][j] - e[0][j]; wb[j] = e[0][j] - e[2][j]; }   // i = 2: d2 - d1, i = 3: d1 - d3 (rows 1..3 are e[0..2])
                    }
                    bx_transform(wa, ra);
                    bx_transform(wb, rb);
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        st4(vs + j * NT * WKC, from4(ra[j]));
                        st4(vs + (6 + j) * NT * WKC, from4(rb[j]));
                    }
                };
                if (part == 0) rows(std::integral_constant<int, 0>{});
                else           rows(std::integral_constant<int, 1>{});
            } else {
                // e[0], e[1] = rows {0,2}, {1,2}, {1,2}, {1,3} for i = 0..3
                V4 w[6], r[6];
                if (part == 1) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) w[j] = e[0][j] + e[1][j];        // d1 + d2
                } else if (part == 2) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) w[j] = e[1][j] - e[0][j];        // d2 - d1
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) w[j] = e[0][j] - e[1][j];        // d0 - d2 / d1 - d3
                }
                bx_transform(w, r);
#pragma unroll
                for (int j = 0; j < 6; ++j) st4(vs + j * NT * WKC, from4(r[j]));
            }
        };
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 1> P1{};
        // prologue: XS[0] <- stage 0, XS[1] <- stage 1, registers <- stage 2
        if (nstages > 0) { loadx(); storex(P0); }
        if (nstages > 1) { loadx(); storex(P1); }
        if (nstages > 2) loadx();
        __syncthreads();                                // barrier X: XS[0], XS[1] visible to every producer
        if (nstages > 0) { xform_read(P0); xform_finish(P0); }
        STAMP_DO(unsigned long long pw = 0, px_ = 0, pl = 0, t0, t1, t2, t3;)
        auto stage = [&](auto par, auto npar, int s) {
            STAMP(t0)
            __syncthreads();                            // barrier #s: VS[s&1] ready / VS[(s+1)&1], XS[s&1] free
            STAMP(t1)
            const bool xf = s + 1 < nstages;
            if (xf) xform_read(npar);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < nstages) storex(par);           // stage s+2 (requested an iteration ago) replaces stage s
            if (s + 3 < nstages) loadx();
            __builtin_amdgcn_sched_barrier(0);
            STAMP(t2)
            if (xf) xform_finish(npar);
            STAMP(t3)
            STAMP_DO(pw += t1 - t0; px_ += t2 - t1; pl += t3 - t2;)
        };
#pragma unroll 1
        for (int s = 0; s < nstages; s += 2) {
            stage(P0, P1, s);
            if (s + 1 < nstages) stage(P1, P0, s + 1);
        }
        STAMP_DO(if (threadIdx.x == 256) { atomicAdd(g_w24stamps + 0, pw); atomicAdd(g_w24stamps + 1, px_); atomicAdd(g_w24stamps + 2, pl); atomicAdd(g_w24stamps + 3, (unsigned long long)nstages); })
    } else {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int wave = NCB == 4 ? wid : (wid & 1);                     // Cout block within the item
        const int tbk = NCB == 4 ? 0 : (wid >> 1);                       // NCB = 2: this wave's tile block
        const int t16 = lane & 15, g = lane >> 4;
        __syncthreads();                                // barrier X
        int s = 0;
        constexpr int AD = CTB == 2 ? 4 : 8;            // A-fragment queue depth (k positions; 256 / 128 MFMA cycles each)
        f32x4 aq[AD];
        STAMP_DO(unsigned long long cw = 0, cc = 0, ce = 0, sc0, sc1, sc2, sc3, ls_a = 0, ls_b = 0, ls_c = 0;)
        // this lane's B operand of tile block tb: tile t = 16 tb + t16, quad g -> slot g ^ (3 * ((t >> 3) & 1))
        const int vofs0 = (16 * tbk + t16) * WKC + 4 * (g ^ (3 * ((t16 >> 3) & 1)));
        const int vofs1 = (CTB == 2 ? 16 + t16 : t16) * WKC + 4 * (g ^ (3 * ((t16 >> 3) & 1)));
        for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
            const int grp = item / per_group, rest = item - grp * per_group;
            const int n = rest / regions;               // PAIR: pair index
            const int r = rest - n * regions;
            const int ry = r / regs_x, rx = r - ry * regs_x;
            const int co = (grp * NCB + wave) * 16 + 4 * g;
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LS != 2) bias = ld4(a.bias + co);   // LS 2: the gate bias is part of gx
            // the folded-BN bias rides through the output transform as M[1][1] (k = 7): A_y^T column 1 and A_x^T column 1 are all ones
            f32x4 acc[NK][CTB];
            const float *wbase = a.wpk + ((size_t)grp * nchunk * NCB + wave) * (NK * 64 * 4) + lane * 4;
            const int item_n = item + (int)gridDim.x < nitems ? item + (int)gridDim.x : item;
            const float *wnext = a.wpk + ((size_t)(item_n / per_group) * nchunk * NCB + wave) * (NK * 64 * 4) + lane * 4;
            if (item == (int)blockIdx.x) {
#pragma unroll
                for (int i = 0; i < AD; ++i) aq[i] = ld4(wbase + i * 64 * 4);
            }
            // LS 2: the frame indices behind this window, requested NOW -- read at the head of the epilogue they were two serial full-latency
            // round trips (global_load + vmcnt(0) each) in front of every gx / c load (r05 ISA reading); here they return under the matrix phase
            [[maybe_unused]] int fm_raw = 0, ci_raw = n;
            if constexpr (LS == 2) {
                fm_raw = a.ls_gx_map[n];
                if (a.in0_map) ci_raw = a.in0_map[n];
            }
            // LS 2 with ONE tile block per wave (8 x 16-pixel regions: 96 accumulators): room for the whole epilogue's gx / c (40 registers) to be requested inside the
            // matrix phase, right behind the item's last own weight-fragment request (vmcnt retires in order; the fragments requested behind them are the next item's)
            constexpr bool PRE = LS == 2 && CTB == 1;
            [[maybe_unused]] f32x4 gxp[PRE ? 8 : 1];
            [[maybe_unused]] float cpf[PRE ? 8 : 1];
            [[maybe_unused]] auto ls_prefetch = [&]() {
                if constexpr (PRE) {
                    constexpr int GBp = BF ? 8 : 16;
                    const int fm = __builtin_amdgcn_readfirstlane(fm_raw), ci = __builtin_amdgcn_readfirstlane(ci_raw);
                    const unsigned char *gx_r = reinterpret_cast<const unsigned char *>(a.ls_gx) + ((((size_t)fm * regions + r) * 4 + wave) * 8 * 64 + lane) * GBp;
                    const float *c_r = a.ls_c_in + (((size_t)ci * regions + r) * 4 + wave) * 8 * 64 + lane;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (BF) {
                            const u32x2 d = *reinterpret_cast<const u32x2 *>(gx_r + (size_t)e * 64 * GBp);
                            const unsigned d0 = d[0], d1 = d[1];
                            gxp[e] = f32x4{__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d0 & 0xffff0000u),
                                           __builtin_bit_cast(float, d1 << 16), __builtin_bit_cast(float, d1 & 0xffff0000u)};
                        } else gxp[e] = ld4_s(reinterpret_cast<const float *>(gx_r + (size_t)e * 64 * GBp));
                        cpf[e] = ld1_s(c_r + (size_t)e * 64);
                    }
                }
            };
            auto chunk = [&](auto firstc, int ch) {
                constexpr bool FIRST = decltype(firstc)::value;
                STAMP(sc0)
                __syncthreads();                        // barrier #s
                STAMP(sc1)
                const float *vs = lds + L_VS + (s & 1) * VSZ;
                const float *wp = wbase + (size_t)ch * NCB * (NK * 64 * 4);
                const float *wn = ch + 1 < nchunk ? wp + NCB * (NK * 64 * 4) : wnext;
                if constexpr (CTB == 1) {
                    // one tile block per wave: the four MFMAs of a position form a dependent chain, so two positions run interleaved
                    // (stamps of the first build: 4.7 k cycles per stage for 96 MFMAs = 3.1 k)
                    constexpr int BD = 2, BRG = BD + 1;             // B pairs in flight ahead of the MFMAs, ring size
                    f32x4 bp[BRG][2];
#pragma unroll
                    for (int q = 0; q < BD; ++q) { bp[q][0] = ld4(vs + (2 * q) * NT * WKC + vofs0); bp[q][1] = ld4(vs + (2 * q + 1) * NT * WKC + vofs0); }
                    unroll_k<NK / 2>([&](auto pc) {
                        constexpr int p = decltype(pc)::value, k0 = 2 * p, k1 = 2 * p + 1;
                        const f32x4 av0 = aq[k0 % AD], av1 = aq[k1 % AD];
                        if constexpr (k0 + AD < NK) aq[k0 % AD] = ld4(wp + (k0 + AD) * 64 * 4);
                        else                        aq[k0 % AD] = ld4(wn + (k0 + AD - NK) * 64 * 4);
                        if constexpr (k1 + AD < NK) aq[k1 % AD] = ld4(wp + (k1 + AD) * 64 * 4);
                        else                        aq[k1 % AD] = ld4(wn + (k1 + AD - NK) * 64 * 4);
                        if constexpr (PRE && k1 + AD == NK - 1) ls_prefetch();
                        if constexpr (p + BD < NK / 2) {
                            bp[(p + BD) % BRG][0] = ld4(vs + (k0 + 2 * BD) * NT * WKC + vofs0);
                            bp[(p + BD) % BRG][1] = ld4(vs + (k1 + 2 * BD) * NT * WKC + vofs0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (FIRST) {
                            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                            acc[k0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[0], bp[p % BRG][0][0], k0 == 7 ? bias : z, 0, 0, 0);
                            acc[k1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[0], bp[p % BRG][1][0], k1 == 7 ? bias : z, 0, 0, 0);
                        }
#pragma unroll
                        for (int i = FIRST ? 1 : 0; i < 4; ++i) {
                            acc[k0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[i], bp[p % BRG][0][i], acc[k0][0], 0, 0, 0);
                            acc[k1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[i], bp[p % BRG][1][i], acc[k1][0], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                f32x4 b0[2], b1[2];
                if constexpr (CTB == 2) { b0[0] = ld4(vs + vofs0); b1[0] = ld4(vs + vofs1); }
                if constexpr (CTB == 2) unroll_k<NK>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const f32x4 av = aq[k % AD];
                    if constexpr (k + AD < NK) aq[k % AD] = ld4(wp + (k + AD) * 64 * 4);
                    else                       aq[k % AD] = ld4(wn + (k + AD - NK) * 64 * 4);
                    if constexpr (k + 1 < NK) {         // B operands of k+1 in flight during the MFMAs of k
                        b0[(k + 1) & 1] = ld4(vs + (k + 1) * NT * WKC + vofs0);
                        if constexpr (CTB == 2) b1[(k + 1) & 1] = ld4(vs + (k + 1) * NT * WKC + vofs1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (FIRST) {
                        const f32x4 c0v = k == 7 ? bias : f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b0[k & 1][0], c0v, 0, 0, 0);
                        if constexpr (CTB == 2) acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b1[k & 1][0], c0v, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = FIRST ? 1 : 0; i < 4; ++i) {
                        acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b0[k & 1][i], acc[k][0], 0, 0, 0);
                        if constexpr (CTB == 2) acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b1[k & 1][i], acc[k][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                STAMP(sc2)
                STAMP_DO(cw += sc1 - sc0; cc += sc2 - sc1;)
                ++s;
            };
            chunk(std::true_type{}, 0);
#pragma unroll 1
            for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, ch);
            STAMP(sc2)
            // ---- output transform Y = A_y^T M A_x, ReLU, NHWC stores ----
            asm volatile("s_nop 15" ::: "memory");        // MFMA -> VALU wait states before the inline-asm packed adds (kernels_wino.hip)
            const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
            if constexpr (LS != 0) {
                // ---- ConvLSTM: gates = A_y^T M A_x (+ gx), cell update, c lane-native, h as NHWC ----
                // this lane: hidden channel 4 wave + g of the 2 x 4 pixels of tile q; its f32x4 after the output transform is (i, j, f, o)
                const size_t rows_per_wave = (size_t)CTB * 8;
                const size_t row_w = (((size_t)n * regions + r) * 4 + wave) * rows_per_wave;          // this item's lane-slot rows (c_out, mode 1 gx)
                float *const c_out = a.ls_c_out + (LS == 1 ? (size_t)grp * a.ls_c_dir : 0) + row_w * 64 + lane;
                constexpr int HB = BF ? 2 : 4, GB = BF ? 8 : 16;        // bytes per hidden value / per lane-slot of gx (four gates)
                unsigned char *const h_out = reinterpret_cast<unsigned char *>(a.out) + (LS == 1 ? (size_t)grp * a.ls_h_dir * HB : 0);
                f32x4 gxv[CTB][8];
                float cv[CTB][8];
                V4 T0[CTB][6], T1[CTB][6];
                [[maybe_unused]] const unsigned char *gx_r = nullptr;
                [[maybe_unused]] const float *c_r = nullptr;
                [[maybe_unused]] unsigned char *gx_w = nullptr;
#ifdef UKBB_DIAG
                // ablation bits (UKBB_LSTM_DIAG, results are garbage): 1 = no cell arithmetic, 2 = no gx / c loads, 4 = no c / h stores, 8 = no epilogue at all
                const int dg = a.diag;
#else
                constexpr int dg = 0;
#endif
                if (dg & 8) continue;
                if constexpr (LS == 2) {
                    const int fm = __builtin_amdgcn_readfirstlane(fm_raw), ci = __builtin_amdgcn_readfirstlane(ci_raw);
                    gx_r = reinterpret_cast<const unsigned char *>(a.ls_gx) + ((((size_t)fm * regions + r) * 4 + wave) * rows_per_wave * 64 + lane) * GB;
                    c_r = a.ls_c_in + (((size_t)ci * regions + r) * 4 + wave) * rows_per_wave * 64 + lane;
                } else {
                    gx_w = reinterpret_cast<unsigned char *>(a.ls_gx) + ((size_t)grp * a.ls_gx_dir / 4 + row_w * 64 + lane) * GB;
                }
#pragma unroll
                for (int tb = 0; tb < CTB; ++tb) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {       // rows: t = A_y^T M
                        const V4 m0 = to4(acc[j][tb]), m1 = to4(acc[6 + j][tb]), m2 = to4(acc[12 + j][tb]), m3 = to4(acc[18 + j][tb]);
                        T0[tb][j] = (m0 + m1) + m2;
                        T1[tb][j] = (m1 - m2) - m3;
                    }
                    if constexpr (PRE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { gxv[tb][e] = (dg & 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : gxp[e]; cv[tb][e] = (dg & 2) ? 0.f : cpf[e]; }
                    } else if constexpr (LS == 2) {     // this tile block's gx and c on their way while the next block's rows are formed
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (dg & 2) { gxv[tb][e] = f32x4{0.f, 0.f, 0.f, 0.f}; cv[tb][e] = 0.f; continue; }
                            if constexpr (BF) {
                                const u32x2 d = *reinterpret_cast<const u32x2 *>(gx_r + (size_t)(tb * 8 + e) * 64 * GB);
                                const unsigned d0 = d[0], d1 = d[1];
                                gxv[tb][e] = f32x4{__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d0 & 0xffff0000u),
                                                   __builtin_bit_cast(float, d1 << 16), __builtin_bit_cast(float, d1 & 0xffff0000u)};
                            } else {
                                gxv[tb][e] = ld4_s(reinterpret_cast<const float *>(gx_r + (size_t)(tb * 8 + e) * 64 * GB));
                            }
                            cv[tb][e] = ld1_s(c_r + (size_t)(tb * 8 + e) * 64);
                        }
                    }
                }
                STAMP_DO(unsigned long long e0_, e1_ = 0, e2_ = 0; STAMP(e0_) ls_a += e0_ - sc2;)
#pragma unroll
                for (int tb = 0; tb < CTB; ++tb) {
                    const int q = (tb + tbk) * 16 + t16;
                    const int oy = (ry * TRY + q / TRX) * 2, ox = (rx * TRX + q % TRX) * 4;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        STAMP_DO(if (tb == 0 && i == 1) { STAMP(e1_) ls_b += e1_ - e0_; } if (tb == 1 && i == 0) { STAMP(e2_) ls_c += e2_ - e1_; })
                        const V4 (&t)[6] = i == 0 ? T0[tb] : T1[tb];
                        const V4 s1 = t[1] + t[2], d1 = t[1] - t[2], s2 = t[3] + t[4], d2 = t[3] - t[4];
                        V4 y[4];
                        y[0] = (t[0] + s1) + s2;
                        y[1] = fma4(d2, c2, d1);
                        y[2] = fma4(s2, c4, s1);
                        y[3] = fma4(d2, c8, d1) + t[5];
                        unsigned hb[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 gt = from4(y[j]);
                            float c = 0.f;
                            if constexpr (LS == 2) { gt += gxv[tb][i * 4 + j]; c = cv[tb][i * 4 + j]; }
                            else if constexpr (BF) {    // the steps add the ROUNDED gx: the x pass's own first step uses the same values
                                const bf16x2 lo = __builtin_convertvector(f32x2{gt[0], gt[1]}, bf16x2), hi = __builtin_convertvector(f32x2{gt[2], gt[3]}, bf16x2);
                                const unsigned d0 = __builtin_bit_cast(unsigned, lo), d1 = __builtin_bit_cast(unsigned, hi);
                                *reinterpret_cast<u32x2 *>(gx_w + (size_t)(tb * 8 + i * 4 + j) * 64 * GB) = u32x2{d0, d1};
                                gt = f32x4{__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d0 & 0xffff0000u),
                                           __builtin_bit_cast(float, d1 << 16), __builtin_bit_cast(float, d1 & 0xffff0000u)};
                            } else st4(reinterpret_cast<float *>(gx_w + (size_t)(tb * 8 + i * 4 + j) * 64 * GB), gt);
                            const float hv = (dg & 1) ? gt[0] + c : ls_cell(gt, c, a.ls_forget_bias);
                            if (!(dg & 4)) st1_s(c_out + (size_t)(tb * 8 + i * 4 + j) * 64, c);
                            hb[j] = __builtin_bit_cast(unsigned, hv);
                        }
                        // 4 x 4 transpose over (pixel column j, lane group g): afterwards register e of lane group g is hidden channel 4 wave + e
                        // of pixel column g -- one 16-byte (bf16: 8-byte) piece of the NHWC map per lane
                        const auto s02 = __builtin_amdgcn_permlane32_swap(hb[0], hb[2], false, false);
                        const auto s13 = __builtin_amdgcn_permlane32_swap(hb[1], hb[3], false, false);
                        const auto p01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                        const auto p23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                        // (elements copied to scalars first: __builtin_bit_cast on a vector ELEMENT expression reads element 0, hipcc 7.2)
                        const unsigned u0 = p01[0], u1 = p01[1], u2 = p23[0], u3 = p23[1];
                        const f32x4 hv4 = {__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), __builtin_bit_cast(float, u2), __builtin_bit_cast(float, u3)};
                        if (oy + i < a.Ho && ox + g < a.Wo && !(dg & 4)) {
                            unsigned char *const hp = h_out + (((size_t)(n * a.Ho + oy + i) * a.Wo + ox + g) * 16 + 4 * wave) * HB;
                            if constexpr (BF) {
                                const bf16x2 lo = __builtin_convertvector(f32x2{hv4[0], hv4[1]}, bf16x2), hi = __builtin_convertvector(f32x2{hv4[2], hv4[3]}, bf16x2);
                                *reinterpret_cast<u32x2 *>(hp) = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                            } else st4(reinterpret_cast<float *>(hp), hv4);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int tb = 0; tb < CTB; ++tb) {
                const int q = (tb + tbk) * 16 + t16;
                int oy = (ry * TRY + q / TRX) * 2, on = n;
                const int ox = (rx * TRX + q % TRX) * 4;
                if constexpr (PAIR) {
                    const int tr = q / TRX;
                    if (ry < pm) on = 2 * n;                                                  // image A
                    else if (ry == pm) { on = 2 * n + (tr >= 2 ? 1 : 0); oy = tr >= 2 ? 2 * (tr - 2) : 8 * pm + 2 * tr; }   // seam
                    else { on = 2 * n + 1; oy = 4 + (ry - pm - 1) * 8 + 2 * tr; }             // image B
                    if (on >= a.N) oy = a.Ho;                                                 // no such image: nothing is stored
                }
                V4 t0[6], t1[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {           // rows: t = A_y^T M  (y0 = m0 + m1 + m2, y1 = m1 - m2 - m3)
                    const V4 m0 = to4(acc[j][tb]), m1 = to4(acc[6 + j][tb]), m2 = to4(acc[12 + j][tb]), m3 = to4(acc[18 + j][tb]);
                    t0[j] = (m0 + m1) + m2;
                    t1[j] = (m1 - m2) - m3;
                }
                float *const o00 = a.out + ((size_t)(on * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
                const size_t dx = (size_t)a.Cout, dy = (size_t)a.Wo * a.Cout;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const V4 (&t)[6] = i == 0 ? t0 : t1;
                    // columns: y = t A_x   (A_x^T rows: [1 1 1 1 1 0], [0 1 -1 2 -2 0], [0 1 1 4 4 0], [0 1 -1 8 -8 1])
                    const V4 s1 = t[1] + t[2], d1 = t[1] - t[2], s2 = t[3] + t[4], d2 = t[3] - t[4];
                    V4 y[4];
                    y[0] = (t[0] + s1) + s2;
                    y[1] = fma4(d2, c2, d1);
                    y[2] = fma4(s2, c4, s1);
                    y[3] = fma4(d2, c8, d1) + t[5];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = from4(y[j]);
                        if (a.relu) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = relu1(v[c]);
                        }
                        if (oy + i < a.Ho && ox + j < a.Wo) st4(o00 + i * dy + j * dx, v);
                    }
                }
            }
            }
            STAMP(sc3)
            STAMP_DO(ce += sc3 - sc2;)
        }
        STAMP_DO(if (threadIdx.x == 0) { atomicAdd(g_w24stamps + 4, cw); atomicAdd(g_w24stamps + 5, cc); atomicAdd(g_w24stamps + 6, ce);
                                         atomicAdd(g_w24stamps + 8, ls_a); atomicAdd(g_w24stamps + 9, ls_b); atomicAdd(g_w24stamps + 10, ls_c); })
    }
}

int wino24_lds_bytes(int tbw, int pair) { return (tbw == 2 ? W24<2>::LDS_FLOATS : pair ? W24<1, true>::LDS_FLOATS : W24<1>::LDS_FLOATS) * 4; }

template <int TBW, bool PAIR, int NCB>
static hipError_t launch_wino24_t(const ConvArgs &a, hipStream_t s) {
    using G = W24<TBW, PAIR>;
    const int n_cu = device_cu_count();
    static OncePerDevice lds_ok;
    hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(wino24_pc_kernel<TBW, PAIR, NCB>), G::LDS_FLOATS * 4);
    if (e != hipSuccess) return e;
    const int regs_x = (a.Wo + 4 * G::TRX - 1) / (4 * G::TRX);
    const long long per_image_or_pair = PAIR ? (long long)(2 * (a.Ho / 8) + 1) * ((a.N + 1) / 2) : (long long)((a.Ho + 2 * TRY - 1) / (2 * TRY)) * a.N;
    const long long nitems = per_image_or_pair * regs_x * (a.Cout / (16 * NCB));
    dim3 grid((unsigned)(nitems < n_cu ? nitems : n_cu));
#ifdef UKBB_WINO_STAMPS
    const bool on = getenv("UKBB_STAMPS") != nullptr;
    unsigned long long z[8] = {0};
    if (on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w24stamps), z, 64);
#endif
    hipLaunchKernelGGL((wino24_pc_kernel<TBW, PAIR, NCB>), grid, dim3(512), G::LDS_FLOATS * 4, s, a);
#ifdef UKBB_WINO_STAMPS
    if (on) {
        unsigned long long h[8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w24stamps), 64);
        const double st = (double)h[3];
        fprintf(stderr, "WINO24STAMPS TBW %d%s Cin %d Cout %d Ho %d: per stage: producer wait %.0f read+store+load %.0f transform %.0f | consumer wait %.0f "
                        "mfma %.0f epilogue(avg/stage) %.0f (stages/WG %.0f)\n", TBW, PAIR ? " paired" : "", a.C0 + a.C1, a.Cout, a.Ho, h[0] / st, h[1] / st, h[2] / st,
                h[4] / st, h[5] / st, h[6] / st, st / grid.x);
    }
#endif
    return hipGetLastError();
}

// tile_cols: 32 | 16 (regions of 8 x 32 / 8 x 16 pixels); pair: images in pairs with seam regions (16 only, Ho % 8 == 4); ncb: 4 | 2 (2: tile_cols 32)
hipError_t launch_wino24(const ConvArgs &a, int tile_cols, int pair, int ncb, hipStream_t s) {
    if (a.Cout % (16 * ncb) || (a.C0 + a.C1) % WKC || a.C0 % WKC || a.up2 || (a.in0_map && pair)) return hipErrorInvalidValue;
    if (ncb == 2) return (tile_cols == 32 && !pair) ? launch_wino24_t<2, false, 2>(a, s) : hipErrorInvalidValue;
    if (ncb != 4) return hipErrorInvalidValue;
    if (pair) return (tile_cols == 16 && a.Ho % 8 == 4 && a.Ho == a.H) ? launch_wino24_t<1, true, 4>(a, s) : hipErrorInvalidValue;
    return tile_cols == 32 ? launch_wino24_t<2, false, 4>(a, s) : tile_cols == 16 ? launch_wino24_t<1, false, 4>(a, s) : hipErrorInvalidValue;
}

template <int TBW, int LS, bool BF>
static hipError_t launch_wino24_lstm_t(const ConvArgs &a, hipStream_t s) {
    using G = W24<TBW, false>;
    const int n_cu = device_cu_count();
    static OncePerDevice lds_ok;
    hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(wino24_pc_kernel<TBW, false, 4, LS, BF>), G::LDS_FLOATS * 4);
    if (e != hipSuccess) return e;
    const int regs_x = (a.Wo + 4 * G::TRX - 1) / (4 * G::TRX);
    const long long nitems = (long long)((a.Ho + 2 * TRY - 1) / (2 * TRY)) * a.N * regs_x * (a.Cout / 64);
    dim3 grid((unsigned)(nitems < n_cu ? nitems : n_cu));
#ifdef UKBB_WINO_STAMPS
    const bool on = getenv("UKBB_STAMPS") != nullptr;
    unsigned long long z[16] = {0};
    if (on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w24stamps), z, 128);
#endif
    hipLaunchKernelGGL((wino24_pc_kernel<TBW, false, 4, LS, BF>), grid, dim3(512), G::LDS_FLOATS * 4, s, a);
#ifdef UKBB_WINO_STAMPS
    if (on) {
        unsigned long long h[16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w24stamps), 128);
        const double st = (double)h[3];
        fprintf(stderr, "LSTMSTAMPS LS %d TBW %d bf %d: per item: producer wait %.0f read+store+load %.0f transform %.0f | consumer wait %.0f mfma %.0f epilogue %.0f "
                        "(row transforms + load issue %.0f, block 0 row 0 %.0f, block 0 row 1 %.0f, rest %.0f) (items/WG %.0f)\n", LS, TBW, (int)BF, h[0] / st, h[1] / st, h[2] / st,
                h[4] / st, h[5] / st, h[6] / st, h[8] / st, h[9] / st, h[10] / st, (h[6] - h[8] - h[9] - h[10]) / st, st / grid.x);
    }
#endif
    return hipGetLastError();
}

hipError_t launch_wino24_lstm(const ConvArgs &a, int tile_cols, hipStream_t s) {
    if ((a.ls_mode != 1 && a.ls_mode != 2) || a.C0 != WKC || a.C1 || a.up2 || a.H != a.Ho || a.W != a.Wo || (tile_cols != 32 && tile_cols != 16)) return hipErrorInvalidValue;
    if (a.ls_mode == 1 ? (a.Cout != 64 && a.Cout != 128) || a.in0_map || !a.bias || !a.ls_gx || !a.ls_c_out : a.Cout != 64 || !a.ls_gx || !a.ls_gx_map || !a.ls_c_in || !a.ls_c_out)
        return hipErrorInvalidValue;
    if (a.ls_bf16) {
        if (a.ls_mode == 1) return tile_cols == 32 ? launch_wino24_lstm_t<2, 1, true>(a, s) : launch_wino24_lstm_t<1, 1, true>(a, s);
        return tile_cols == 32 ? launch_wino24_lstm_t<2, 2, true>(a, s) : launch_wino24_lstm_t<1, 2, true>(a, s);
    }
    if (a.ls_mode == 1) return tile_cols == 32 ? launch_wino24_lstm_t<2, 1, false>(a, s) : launch_wino24_lstm_t<1, 1, false>(a, s);
    return tile_cols == 32 ? launch_wino24_lstm_t<2, 2, false>(a, s) : launch_wino24_lstm_t<1, 2, false>(a, s);
}

static size_t lstm_regions(int Ho, int Wo, int tile_cols) { return (size_t)((Ho + 7) / 8) * ((Wo + tile_cols - 1) / tile_cols); }
// per region: 4 consumer waves x (tile_cols / 16 tile blocks x 8 pixels of a tile) x 64 lanes, one float (c) or four (gx) each
size_t wino24_lstm_c_floats(int Ho, int Wo, int tile_cols) { return lstm_regions(Ho, Wo, tile_cols) * 4 * (size_t)(tile_cols / 16) * 8 * 64; }
size_t wino24_lstm_gx_floats(int Ho, int Wo, int tile_cols) { return 4 * wino24_lstm_c_floats(Ho, Wo, tile_cols); }

size_t pack_wino24_weights(const float *w, int cin, int cout, int ncb, float *dst);
size_t pack_lstm_gate_weights(const float *w, int cin_total, int c_first, const float *bias, float *dst, float *bias_perm) {
    // packed channel P = 16 w + m (MFMA row m = 4 g + e of block w)  <-  original channel 16 e + (4 w + g): gate e of hidden channel 4 w + g
    float tmp[9 * WKC * 64];
    for (int P = 0; P < 64; ++P) {
        const int wv = P / 16, m = P % 16, orig = (m % 4) * 16 + 4 * wv + m / 4;
        if (bias_perm) bias_perm[P] = bias ? bias[orig] : 0.f;
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < WKC; ++ci) tmp[(t * WKC + ci) * 64 + P] = w[((size_t)t * cin_total + c_first + ci) * 64 + orig];
    }
    return pack_wino24_weights(tmp, WKC, 64, 4, dst);
}

size_t pack_wino24_weights(const float *w, int cin, int cout, int ncb, float *dst) {
    // w: folded [3][3][cin][cout].  U = G_y g G_x^T per (ci, co); G_y = F(2,3) rows, G_x = F(4,3) columns.
    // dst[group of 16 ncb][chunk][cbl 0..ncb-1][k = 6 i + j][lane][s]:  lane = (g << 4) | m, ci = chunk*16 + 4*g + s, co = (group*ncb + cbl)*16 + m
    static const double GY[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    static const double GX[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nchunk = cin / WKC;
    size_t o = 0;
    for (int grp = 0; grp < cout / (16 * ncb); ++grp)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cbl = 0; cbl < ncb; ++cbl)
                for (int k = 0; k < NK; ++k)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s) {
                            const int m = lane & 15, g = lane >> 4;
                            const int ci = ch * WKC + 4 * g + s, co = (grp * ncb + cbl) * 16 + m;
                            const int i = k / 6, j = k % 6;
                            double u = 0.0;
                            for (int p = 0; p < 3; ++p)
                                for (int q = 0; q < 3; ++q)
                                    u += GY[i][p] * (double)w[((size_t)(p * 3 + q) * cin + ci) * cout + co] * GX[j][q];
                            dst[o++] = (float)u;
                        }
    return o;
}

}  // namespace ukbb
