// Winograd F(2x2, 3x3) convolution (+ folded BN bias, ReLU) for the stride-1 3x3 layers with
// C_out >= 64 (reference common/network.py:19-25; levels 2-4 of build_FCN and of the U-Net).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d
//
// 16 multiplies per 2x2 outputs instead of 36: the MFMA work of a layer drops 2.25x.  The 16
// element-wise products are 16 independent GEMMs over (C_in) between transformed weights
// U[k][cout][ci] (computed once on the host) and transformed patches V[k][ci][tile], so they run on
// v_mfma_f32_16x16x4_f32 with A = U (cout on M), B = V (tile on N) exactly like the direct kernel.
// Arithmetic stays fp32 end to end; the transforms only add/subtract (entries 0, +-1) on the
// input/output side and the +-1/2 factors sit in the pre-computed weights, so the result differs
// from the direct sum by ordinary fp32 rounding (measured ~1e-6 of the activation scale).
//
// Structure: persistent producer/consumer workgroup (512 threads, one per CU).
//   item  = (group of 64 output channels, image, region of 4 x 8 Winograd tiles = 8 x 16 pixels)
//   stage = one chunk of KC = 16 input channels of one item
//   producers (waves 4-7): global -> registers -> raw halo tile XS (LDS) -> input transform ->
//       VS[16][32 tiles][KC+4] (LDS), three stages deep (load s+3, park s+2, transform s+1);
//   consumers (waves 0-3, one 16-channel block each): per k: one A fragment straight from
//       global/L2 (packed on the host), two B vectors from VS, 8 MFMAs; after the last chunk the
//       output transform, bias, ReLU and the NHWC stores.
// One barrier per stage.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace ukbb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 32;                                // Winograd tiles per region (two MFMA N blocks): 4 x 8 or 8 x 4 tiles
constexpr int WKC = 16, WXS = WKC + 4;                // channels per stage, LDS pixel stride (floats)
constexpr int WHP = 180;                              // raw halo tile: 10 x 18 or 18 x 10 pixels
constexpr int NITX = (WHP * (WKC / 4) + 255) / 256;   // staging passes of the 256 producer threads (3)
// Raw halo tile XS: pixel stride WXSX = 24 floats and row pitch RS pixels (18 for the 18-wide halo of 4 x 8-tile regions, 12 for the
// 10-wide halo of 8 x 4-tile regions).  The input transform reads the 4 x 4 patch of every tile, i.e. pixels TWO apart from lane to
// lane; at the consumer-friendly stride of 20 floats the 16-lane groups of those ds_read_b128 collided 2-way (8 x 16 regions) and
// 3-way (16 x 8): this was the SQ_LDS_BANK_CONFLICT = 0.50-0.55 x SQ_LDS_IDX_ACTIVE of r03's counter passes.  With (24, 18) / (24, 12)
// every group is conflict-free (searched exhaustively, r04_notes.md).  The transformed tiles VS keep the stride of 20.
constexpr int WXSX = 24;
constexpr int XSZ = 5376 + 32;                        // floats per XS buffer: 18 rows x 12 x 24 (the larger of the two shapes) + a slot for idle lanes
constexpr int L_XS = 0;                               // [2][XSZ]
constexpr int L_VS = L_XS + 2 * XSZ;            // [2][16][NT][WXS]
constexpr int WINO_LDS_FLOATS = L_VS + 2 * 16 * NT * WXS;

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void st4(float *p, const f32x4 &v) { *reinterpret_cast<f32x4 *>(p) = v; }

// Packed fp32 add / subtract on register pairs.  hipcc scalarises the subtractions of the input
// transform into v_sub_f32 (+ moves) when left to itself; every VALU instruction of a producer wave
// costs an MFMA slot, so the packed forms are spelled out.
#ifdef UKBB_NO_PACKED_F32
// A/B form (r06): the same arithmetic as pairs of scalar VALU instructions -- the guide prices a packed f32 op
// beside MFMAs above two scalar ones; tools/ab_packed.sh measures it next to the fp32 MFMA streams of these kernels.
__device__ __forceinline__ float s_add(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return f32x2{s_add(a[0], b[0]), s_add(a[1], b[1])}; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return f32x2{s_sub(a[0], b[0]), s_sub(a[1], b[1])}; }
#else
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
#endif

__device__ __forceinline__ float relu1(float x) {   // one v_max_i32 (fmaxf adds a canonicalising second instruction)
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 add4(const f32x4 &a, const f32x4 &b) {
    const f32x2 lo = pk_add(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_add(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 sub4(const f32x4 &a, const f32x4 &b) {
    const f32x2 lo = pk_sub(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_sub(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_k(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_k<N, I + 1>(f); }
}

}  // namespace

// Diagnostic only (-DUKBB_WINO_STAMPS + UKBB_STAMPS=1): s_memtime stamps around the phases of a stage.
#ifdef UKBB_WINO_STAMPS
#define STAMP(v) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); }
#define STAMP_DO(...) __VA_ARGS__
__device__ unsigned long long g_wstamps[8];
#else
#define STAMP(v)
#define STAMP_DO(...)
#endif
// NCB = blocks of 16 output channels per item.  NCB = 4: consumer wave w owns channel block w and both
// 16-tile halves of the region (128 MFMAs per stage); NCB = 2 (layers with 32 output channels):
// wave w owns channel block w & 1 and tile half w >> 1 (64 MFMAs per stage).
// TRY x (32 / TRY) tiles per region: 4 x 8 (8 x 16 pixels) or 8 x 4 (16 x 8 pixels) -- whichever wastes less of
// the map (48 x 52: 24 regions of 8 x 16 at 81 % fill, or 21 of 16 x 8 at 93 %).
template <int NCB, int TRY>
__global__ __launch_bounds__(512, 2) void wino_pc_kernel(const ConvArgs a) {
    constexpr int WNCBL = NCB, TBW = NCB == 4 ? 2 : 1;
    constexpr int TRX = NT / TRY, WIH = 2 * TRY + 2, WIW = 2 * TRX + 2;
    static_assert(WIH * WIW == WHP && WIH + WIW < 31, "halo tile / mask layout");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cin = a.C0 + a.C1;
    const int nchunk = cin / WKC;
    const int regs_x = (a.Wo + 2 * TRX - 1) / (2 * TRX), regs_y = (a.Ho + 2 * TRY - 1) / (2 * TRY);
    const int regions = regs_x * regs_y;
    const int per_group = a.N * regions;
    const int nitems = per_group * (a.Cout / (16 * WNCBL));
    const int my_items = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstages = my_items * nchunk;
    // Regions of one image can cost differently (12-row maps: the lower region runs half the MFMAs, see `half` below), and with
    // a grid that is a multiple of `regions` a workgroup would meet the same region of some image in every round: rotate the
    // region index by the round so that every workgroup gets its share of cheap and expensive ones (a bijection on the items
    // of a round as long as a round holds whole images' region sets).
    const bool rotate = (int)gridDim.x % regions == 0;
    const bool producer = __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;

    if (producer) {
        // On a SIMD the producers' VALU instructions share the issue port with the consumer's MFMA
        // stream and only get a slot between two MFMAs (measured: ~60 cycles per VALU op at equal
        // priority, ~28 at raised priority, each stealing a few MFMA cycles).  So this role is written
        // for a minimal VALU count (41 per stage against 128 MFMAs): buffer loads whose out-of-image lanes carry an out-of-range offset
        // (the hardware returns 0; no select at the LDS write), all per-thread offsets precomputed,
        // region / chunk bookkeeping on the scalar unit, the stage loop unrolled by two so LDS
        // addresses are immediates, packed fp32 adds in the transform.  (s_setprio on the producers was
        // measured: it shortens their phase but lengthens the MFMA phase by the same amount.)
        const int tid = threadIdx.x - 256;
        constexpr int C4 = WKC / 4, PSTEP = 256 / C4;                 // 64 halo pixels per pass, 3 passes
        const int c4 = tid % C4, pix0 = tid / C4;
        unsigned tb[NITX], pixoff[NITX], pre[NITX];
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int pix = pix0 + it * PSTEP;
            const int iy = pix / WIW, ix = pix - iy * WIW;
            tb[it] = pix < WHP ? (1u << iy) | (1u << (WIH + ix)) : 0x80000000u;   // bit 31 is never set in the stage mask
            pixoff[it] = (unsigned)(iy * a.W + ix);
            pre[it] = 0;
        }
        int cur_cs = 0;
        u32x4 xr[NITX];
        // scalar cursor of the next stage to request
        int l_ch = 0, l_item = blockIdx.x, l_n = 0, l_n0 = 0, l_iy0 = 0, l_ix0 = 0, l_round = 0;
        auto locate = [&]() {
            const int rest = l_item % per_group;
            l_n = rest / regions;
            // ConvLSTM windows index shared feature frames.  readfirstlane: the loaded index arrives in a VGPR, and without it
            // the buffer descriptors built from it count as divergent -- every halo load became a waterfall loop (r02)
            l_n0 = a.in0_map ? __builtin_amdgcn_readfirstlane(a.in0_map[l_n]) : l_n;
            int r = rest - l_n * regions;
            if (rotate) r = (r + l_round) % regions;
            const int ry = r / regs_x, rx = r - ry * regs_x;
            l_iy0 = ry * 2 * TRY - 1; l_ix0 = rx * 2 * TRX - 1;
        };
        locate();
        auto loadx = [&]() {                            // raw halo of the cursor stage -> registers; advances the cursor
            const float *src; int cs;
            const bool from0 = l_ch * WKC < a.C0;
            if (from0) { src = a.in0 + l_ch * WKC; cs = a.C0; }
            else       { src = a.in1 + (l_ch * WKC - a.C0); cs = a.C1; }
            const int nimg = from0 ? l_n0 : l_n;
            if (cs != cur_cs) {                         // uniform; once per source switch
                cur_cs = cs;
#pragma unroll
                for (int it = 0; it < NITX; ++it) pre[it] = pixoff[it] * (unsigned)(cs * 4) + 16u * c4;
            }
            src += ((long long)(nimg * a.H + l_iy0) * a.W + l_ix0) * cs;   // may point before the tensor; masked lanes never use it
            const int ylo = l_iy0 < 0 ? -l_iy0 : 0, yhi = a.H - l_iy0 < WIH ? a.H - l_iy0 : WIH;
            const int xlo = l_ix0 < 0 ? -l_ix0 : 0, xhi = a.W - l_ix0 < WIW ? a.W - l_ix0 : WIW;
            const unsigned cm = (((1u << yhi) - 1u) & ~((1u << ylo) - 1u)) | ((((1u << xhi) - 1u) & ~((1u << xlo) - 1u)) << WIH);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int it = 0; it < NITX; ++it) {
                const unsigned vo = (cm & tb[it]) == tb[it] ? pre[it] : 0x80000000u;   // out of range -> zeros
                xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            }
            if (++l_ch == nchunk) { l_ch = 0; l_item += gridDim.x; ++l_round; if (l_item < nitems) locate(); }
        };
        constexpr int RS = TRY == 4 ? 18 : 12;          // row pitch of the raw tile in pixels
        static_assert(WIH * RS * WXSX <= XSZ - 32, "raw tile fits its buffer");
        unsigned xoff[NITX];                            // this thread's LDS float offsets inside an XS buffer (idle lanes of the last pass: the spare slot)
#pragma unroll
        for (int it = 0; it < NITX; ++it) {
            const int pix = pix0 + it * PSTEP;
            const int iy = pix / WIW, ix = pix - iy * WIW;
            xoff[it] = pix < WHP ? (unsigned)((iy * RS + ix) * WXSX + 4 * c4) : (unsigned)(XSZ - 32 + 4 * c4);
        }
        auto storex = [&](auto par) {
            constexpr int B = decltype(par)::value;
#pragma unroll
            for (int it = 0; it < NITX; ++it)
                *reinterpret_cast<u32x4 *>(lds + L_XS + B * XSZ + xoff[it]) = xr[it];
        };
        // V = B^T d B per (tile, channel quad).  The two producer wave pairs split the row pass: waves
        // 4-5 produce rows 0-1 of B^T d (from patch rows 0-2), waves 6-7 rows 2-3 (from patch rows 1-3),
        // so each thread issues 12 reads, 32 packed adds and 8 writes.
        const int half = __builtin_amdgcn_readfirstlane(tid) >> 7;
        const int xt = tid & 127, x_tile = xt / C4, x_q = xt - x_tile * C4;
        const float *const xs_r = lds + L_XS + ((2 * (x_tile / TRX) + half) * RS + 2 * (x_tile % TRX)) * WXSX + 4 * x_q;
        float *const vs_w = lds + L_VS + half * 8 * NT * WXS + x_tile * WXS + 4 * x_q;
        f32x2 e[3][4][2];
        auto xform_read = [&](auto par) {               // issue the 12 patch reads of this thread
            constexpr int B = decltype(par)::value;
            const float *xs = xs_r + B * XSZ;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = ld4(xs + (i * RS + j) * WXSX);
                    e[i][j][0] = f32x2{v[0], v[1]}; e[i][j][1] = f32x2{v[2], v[3]};
                }
        };
        auto xform_finish = [&](auto par) {             // row pass, column pass, 8 writes
            constexpr int B = decltype(par)::value;
            float *vs = vs_w + B * 16 * NT * WXS;
            auto put = [&](int k, const f32x2 &lo, const f32x2 &hi) { st4(vs + k * NT * WXS, f32x4{lo[0], lo[1], hi[0], hi[1]}); };
            auto rows = [&](auto hc) {                  // a real (uniform) branch per wave pair: stores inside keep it from being if-converted
                constexpr int HALF = decltype(hc)::value;
                f32x2 wa[4][2], wb[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr (HALF == 0) { wa[j][h] = pk_sub(e[0][j][h], e[2][j][h]); wb[j][h] = pk_add(e[1][j][h], e[2][j][h]); }   // rows 0, 1
                        else                     { wa[j][h] = pk_sub(e[1][j][h], e[0][j][h]); wb[j][h] = pk_sub(e[0][j][h], e[2][j][h]); }   // rows 2, 3
                    }
                put(0, pk_sub(wa[0][0], wa[2][0]), pk_sub(wa[0][1], wa[2][1]));
                put(1, pk_add(wa[1][0], wa[2][0]), pk_add(wa[1][1], wa[2][1]));
                put(2, pk_sub(wa[2][0], wa[1][0]), pk_sub(wa[2][1], wa[1][1]));
                put(3, pk_sub(wa[1][0], wa[3][0]), pk_sub(wa[1][1], wa[3][1]));
                put(4, pk_sub(wb[0][0], wb[2][0]), pk_sub(wb[0][1], wb[2][1]));
                put(5, pk_add(wb[1][0], wb[2][0]), pk_add(wb[1][1], wb[2][1]));
                put(6, pk_sub(wb[2][0], wb[1][0]), pk_sub(wb[2][1], wb[1][1]));
                put(7, pk_sub(wb[1][0], wb[3][0]), pk_sub(wb[1][1], wb[3][1]));
            };
            if (half == 0) rows(std::integral_constant<int, 0>{});
            else           rows(std::integral_constant<int, 1>{});
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
        // One stage of the producer role.  The patch reads of the transform are issued first and their
        // latency is covered by parking stage s+2 in XS and requesting stage s+3 from global memory.
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
        STAMP_DO(if (threadIdx.x == 256) { atomicAdd(g_wstamps + 0, pw); atomicAdd(g_wstamps + 1, px_); atomicAdd(g_wstamps + 2, pl); atomicAdd(g_wstamps + 3, (unsigned long long)nstages); })
    } else {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        const int wave = NCB == 4 ? wid : (wid & 1);        // Cout block within the item
        const int tb0 = NCB == 4 ? 0 : (wid >> 1);         // first 16-tile block of this wave
        const int t16 = lane & 15, g = lane >> 4;
        __syncthreads();                                // barrier X
        int s = 0;
        constexpr int AD = TBW == 2 ? 4 : 8;             // A-fragment queue depth (k positions)
        constexpr int BA = TBW == 2 ? 1 : 2, BR = BA + 1; // B lookahead (k positions) and ring size
        f32x4 aq[AD];
        STAMP_DO(unsigned long long cw = 0, cc = 0, ce = 0, c0, c1, c2, c3;)
        int round = 0;
        for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++round) {
            const int grp = item / per_group, rest = item - grp * per_group;
            const int n = rest / regions;
            int r = rest - n * regions;
            if (rotate) r = (r + round) % regions;
            const int ry = r / regs_x, rx = r - ry * regs_x;
            const int co = (grp * WNCBL + wave) * 16 + 4 * g;
            const f32x4 bias = ld4(a.bias + co);        // needed after the last chunk; in flight during the stages
            // Accumulators are never zeroed with moves: the first MFMA of the first chunk takes its C operand
            // as the constant 0 -- or, for Winograd position (1,1), the bias: A^T M A adds M[1][1] to all four
            // outputs of the tile, so the folded-BN bias rides through the output transform for free.
            f32x4 acc[16][TBW];
            // packed weights: [group][chunk][cbl][k][lane][4].  A fragments come straight from global/L2,
            // through a rolling register queue over the flattened (item, chunk, k) sequence: the load for
            // k+AD is issued before the MFMAs of k (AD x 256 or 128 MFMA cycles cover the L2 latency); the
            // queue runs across stage barriers and across items (the last chunk of an item requests the
            // first fragments of the next item's channel group).
            const float *wbase = a.wpk + ((size_t)grp * nchunk * WNCBL + wave) * (16 * 64 * 4) + lane * 4;
            const int item_n = item + (int)gridDim.x < nitems ? item + (int)gridDim.x : item;
            const float *wnext = a.wpk + ((size_t)(item_n / per_group) * nchunk * WNCBL + wave) * (16 * 64 * 4) + lane * 4;
            if (item == (int)blockIdx.x) {
#pragma unroll
                for (int i = 0; i < AD; ++i) aq[i] = ld4(wbase + i * 64 * 4);
            }
            auto chunk = [&](auto firstc, auto halfc, int ch) {
                constexpr bool FIRST = decltype(firstc)::value;
                constexpr bool TWO = TBW == 2 && !decltype(halfc)::value;   // second 16-tile block holds real tiles
                STAMP(c0)
                __syncthreads();                        // barrier #s
                STAMP(c1)
                const float *vs = lds + L_VS + (s & 1) * 16 * NT * WXS + (tb0 * 16 + t16) * WXS + 4 * g;
                const float *wp = wbase + (size_t)ch * WNCBL * (16 * 64 * 4);
                const float *wn = ch + 1 < nchunk ? wp + WNCBL * (16 * 64 * 4) : wnext;
                f32x4 b0[BR], b1[BR];
#pragma unroll
                for (int i = 0; i < BA; ++i) {
                    b0[i] = ld4(vs + i * NT * WXS);
                    if constexpr (TWO) b1[i] = ld4(vs + i * NT * WXS + 16 * WXS);
                }
                unroll_k<16>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const f32x4 av = aq[k % AD];
                    if constexpr (k + AD < 16) aq[k % AD] = ld4(wp + (k + AD) * 64 * 4);
                    else                       aq[k % AD] = ld4(wn + (k + AD - 16) * 64 * 4);
                    if constexpr (k + BA < 16) {        // B operands of k+BA in flight during the MFMAs of k
                        b0[(k + BA) % BR] = ld4(vs + (k + BA) * NT * WXS);
                        if constexpr (TWO) b1[(k + BA) % BR] = ld4(vs + (k + BA) * NT * WXS + 16 * WXS);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (FIRST) {
                        const f32x4 c0v = k == 5 ? bias : f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b0[k % BR][0], c0v, 0, 0, 0);
                        if constexpr (TWO)
                            acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], b1[k % BR][0], c0v, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = FIRST ? 1 : 0; i < 4; ++i) {
                        acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b0[k % BR][i], acc[k][0], 0, 0, 0);
                        if constexpr (TWO)
                            acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b1[k % BR][i], acc[k][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                STAMP(c2)
                STAMP_DO(cw += c1 - c0; cc += c2 - c1;)
                ++s;
            };
            // A region whose lower two tile rows lie entirely below the image (12-row maps: rows 8-11 of the second
            // region) has no real tile in its second 16-tile block: skip that block's MFMAs and output transform.
            const bool half = TBW == 2 && ry * 2 * TRY + TRY >= a.Ho;
            if (half) {
                chunk(std::true_type{}, std::true_type{}, 0);
#pragma unroll 1
                for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, std::true_type{}, ch);
            } else {
                chunk(std::true_type{}, std::false_type{}, 0);
#pragma unroll 1
                for (int ch = 1; ch < nchunk; ++ch) chunk(std::false_type{}, std::false_type{}, ch);
            }
            STAMP(c2)
            // ---- output transform Y = A^T M A, bias, ReLU, NHWC stores -----------------------------
            // The packed adds below are inline asm, which the compiler's hazard recogniser does not treat as a
            // VALU read of MFMA results: cover the MFMA -> VALU wait states (11 for this 8-pass MFMA) by hand.
            asm volatile("s_nop 15" ::: "memory");
#pragma unroll
            for (int tb = 0; tb < TBW; ++tb) {
                if (tb == 1 && half) break;
                const int q = (tb0 + tb) * 16 + t16;
                const int oy = (ry * TRY + q / TRX) * 2, ox = (rx * TRX + q % TRX) * 2;
                f32x4 t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {           // rows: t = A^T M
                    t0[j] = add4(add4(acc[0 + j][tb], acc[4 + j][tb]), acc[8 + j][tb]);
                    t1[j] = sub4(sub4(acc[4 + j][tb], acc[8 + j][tb]), acc[12 + j][tb]);
                }
                f32x4 y[2][2];
                y[0][0] = add4(add4(t0[0], t0[1]), t0[2]);
                y[0][1] = sub4(sub4(t0[1], t0[2]), t0[3]);
                y[1][0] = add4(add4(t1[0], t1[1]), t1[2]);
                y[1][1] = sub4(sub4(t1[1], t1[2]), t1[3]);
                float *const o00 = a.out + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co;
                const size_t dx = (size_t)a.Cout, dy = (size_t)a.Wo * a.Cout;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4 v = y[i][j];
                        if (a.relu) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = relu1(v[c]);
                        }
#ifdef UKBB_WINO_STAMPS
                        if (a.diag & 32) continue;              // ablation: no output stores
#endif
                        if (oy + i < a.Ho && ox + j < a.Wo) st4(o00 + i * dy + j * dx, v);
                    }
            }
            STAMP(c3)
            STAMP_DO(ce += c3 - c2;)
        }
        STAMP_DO(if (threadIdx.x == 0) { atomicAdd(g_wstamps + 4, cw); atomicAdd(g_wstamps + 5, cc); atomicAdd(g_wstamps + 6, ce); })
    }
}

int wino_lds_bytes() { return WINO_LDS_FLOATS * 4; }

template <int NCB, int TRY>
static hipError_t launch_wino_t(const ConvArgs &a, hipStream_t s) {
    constexpr int TRX = NT / TRY;
    if (a.Cout % (16 * NCB) || (a.C0 + a.C1) % WKC || a.C0 % WKC || a.up2) return hipErrorInvalidValue;
    const int n_cu = device_cu_count();
    static OncePerDevice lds_ok;
    {
        hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(wino_pc_kernel<NCB, TRY>), WINO_LDS_FLOATS * 4);
        if (e != hipSuccess) return e;
    }
    const int regs_x = (a.Wo + 2 * TRX - 1) / (2 * TRX), regs_y = (a.Ho + 2 * TRY - 1) / (2 * TRY);
    const long long nitems = (long long)a.N * regs_x * regs_y * (a.Cout / (16 * NCB));
    dim3 grid((unsigned)(nitems < n_cu ? nitems : n_cu));
#ifdef UKBB_WINO_STAMPS
    const bool on = getenv("UKBB_STAMPS") != nullptr;
    unsigned long long z[8] = {0};
    if (on) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), z, 64);
#endif
    hipLaunchKernelGGL((wino_pc_kernel<NCB, TRY>), grid, dim3(512), WINO_LDS_FLOATS * 4, s, a);
#ifdef UKBB_WINO_STAMPS
    if (on) {
        unsigned long long h[8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wstamps), 64);
        const double st = (double)h[3];
        fprintf(stderr, "WINOSTAMPS Cin %d Cout %d Ho %d: per stage: producer wait %.0f read+store+load %.0f transform %.0f | consumer wait %.0f "
                        "mfma %.0f epilogue(avg/stage) %.0f (stages/WG %.0f)\n", a.C0 + a.C1, a.Cout, a.Ho, h[0] / st, h[1] / st, h[2] / st,
                h[4] / st, h[5] / st, h[6] / st, st / grid.x);
    }
#endif
    return hipGetLastError();
}

hipError_t launch_wino(const ConvArgs &a_in, int ncb, int tile_rows, hipStream_t s) {
    ConvArgs a = a_in;
#ifdef UKBB_WINO_STAMPS
    { const char *e = getenv("UKBB_CONV_DIAG"); a.diag = e ? atoi(e) : 0; }
#endif
    if (ncb == 4) return tile_rows == 8 ? launch_wino_t<4, 8>(a, s) : launch_wino_t<4, 4>(a, s);
    if (ncb == 2) return tile_rows == 8 ? launch_wino_t<2, 8>(a, s) : launch_wino_t<2, 4>(a, s);
    return hipErrorInvalidValue;
}

size_t pack_wino_weights(const float *w, int cin, int cout, int ncb, float *dst) {
    const int WNCBL = ncb;
    // w: folded [3][3][cin][cout].  U = G g G^T per (ci, co), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
    // dst[group][chunk][cbl][k = 4*xi + nu][lane][s]:  lane = (g << 4) | m,
    //   ci = chunk*16 + 4*g + s, co = (group*4 + cbl)*16 + m          (A fragment of v_mfma_f32_16x16x4_f32)
    static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    const int nchunk = cin / WKC;
    size_t o = 0;
    for (int grp = 0; grp < cout / (16 * WNCBL); ++grp)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cbl = 0; cbl < WNCBL; ++cbl)
                for (int k = 0; k < 16; ++k)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s) {
                            const int m = lane & 15, g = lane >> 4;
                            const int ci = ch * WKC + 4 * g + s, co = (grp * WNCBL + cbl) * 16 + m;
                            const int xi = k >> 2, nu = k & 3;
                            double u = 0.0;                 // exact products of small dyadic factors; one rounding
                            for (int i = 0; i < 3; ++i)
                                for (int j = 0; j < 3; ++j)
                                    u += (double)G[xi][i] * (double)w[((size_t)(i * 3 + j) * cin + ci) * cout + co] * (double)G[nu][j];
                            dst[o++] = (float)u;
                        }
    return o;
}

}  // namespace ukbb
