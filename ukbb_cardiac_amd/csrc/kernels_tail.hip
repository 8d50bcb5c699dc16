// Fused tail of the aortic U-Net in UKBB_PREC_BF16: up0_0 (3x3, concat[skip, up] 16 + 16 -> 16, BN, ReLU), up0_1 (3x3, 16 -> 16, BN,
// ReLU), the 1x1 logits conv + bias and softmax / argmax (reference common/network_ao.py:51-63,159-160 through network.py:19-25)
// in ONE launch: neither conv's 256 x 256 x 16 output ever exists in HBM (420 MB written + read per 100 slices, and one full pass of
// per-pixel work), the kernel reads the two 16-channel inputs and writes the label map.
//
// Why its own kernel (r04): at 16 channels a conv has ~9 bf16 MFMAs of work per 32 pixels; the level-0 layers were bound by the
// per-pixel instruction and memory cost of each PASS, not by their arithmetic (up0_0 126 us + up0_1/logits 93 us at N = 100 against
// 105 + 39 us of HBM time).  Structure, as kernels_ws.hip: every wave is an independent worker with a private LDS area, no barrier.
//   tile      R rows x 30 columns of the label map.
//   stage 1   up0_0 on the (R + 2) x 32 pixels stage 2 needs, from the (R + 4) x 34 halo of both inputs (staged global -> registers
//             -> LDS one tile ahead, channel-blocked storage: contiguous 32-byte pixels).  v_mfma_f32_16x16x32_bf16: M = the 16 output
//             channels (no zero-padded rows as on the 32-row MFMA), N = 16 pixels, K = 32 = skip 16 + up 16 channels of one tap -- lanes
//             of k-quarters 0, 1 read the skip stage, 2, 3 the up stage.  The nine A fragments live in registers for the whole launch.
//             A B fragment (halo row r', column shift kw, 16-pixel block) serves the three output rows r' - kh.
//             Result: bias (C operand), ReLU, zero outside the image (it is up0_1's zero padding), rounded to bf16 exactly as the
//             unfused layer stores it, written to the wave's "mid" tile in LDS in the [k half][pixel] layout stage 2 reads.
//   stage 2   up0_1 on R x 32 pixels (30 stored): K = 32 = TWO taps of 16 channels per MFMA (lanes of k-quarters 2, 3 read the
//             next tap's pixel), five MFMAs per 16 pixels, A fragments in registers.
//   logits    16 -> n_class on the matrix pipe as in kernels_ws.hip (bf16 hi + lo pieces of the fp32 weights, bias as C, the
//             lane's own four channels as its k-slots: no cross-lane movement), softmax / argmax (kernels.h), buffer stores.
// Arithmetic: the same bf16 products and fp32 accumulation as the unfused kernels in a different summation order; the intermediate
// rounding points (bf16 after up0_0 and after up0_1) are kept, so results agree with the unfused plan to a bf16 ulp of a few
// intermediate values (tools/check_ws.py, tests/test_gpu_parity.py).
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

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

// NB = 16-pixel MFMA blocks per tile row: label columns per tile 16 NB - 2, stage-1 columns 16 NB, input halo columns 16 NB + 2.
// NB = 2 (30-column tiles) needs 37 KB of LDS and 415 registers per wave: ONE wave per SIMD, and its non-MFMA phases (park, request,
// mid, epilogue: 5-6 k of 10.4 k cycles per tile, r04 stamps) leave the matrix pipe idle.  NB = 1 (14-column tiles) fits 19.7 KB and
// 256 registers: TWO waves per SIMD cover each other's phases, for 7 % more MFMAs and 14 % more halo reads per label pixel.
__host__ __device__ constexpr int tl_tw(int nb) { return 16 * nb - 2; }
__host__ __device__ constexpr int tl_mw(int nb) { return 16 * nb; }
__host__ __device__ constexpr int tl_iw(int nb) { return 16 * nb + 2; }

__host__ __device__ constexpr int tl_hp(int r, int nb) { return (r + 4) * tl_iw(nb); }            // input halo pixels
__host__ __device__ constexpr int tl_nld(int r, int nb) { return (2 * tl_hp(r, nb) + 63) / 64; }  // 16-byte pieces per lane and source
// input stage of one source: [halo pixel][k half] x 16 B = the pixel's 32 bytes as they lie in HBM, LINEAR in the order the load
// instructions deliver them (conflict-free ds_write_b128 at immediate offsets, no per-lane offsets); rounded up to whole load
// instructions so that the idle lanes of the last one write padding of the same stage.  Fragment reads (lane: pixel n16, k-quarter kq)
// at 32-byte pixel stride stay conflict-free because a ds_read_b128's 16-lane group holds pixels {0-3, 12-15} of one quarter and {4-11}
// of the next: even and odd 16-byte slots, all different.
__host__ __device__ constexpr int tl_stage(int r, int nb) { return tl_nld(r, nb) * 1024; }
// mid tile: [k half][pixel] planes a multiple of 256 bytes apart (16-byte pixel stride; the same group argument with whole planes)
__host__ __device__ constexpr int tl_mplane(int r, int nb) { return ((r + 2) * tl_mw(nb) * 16 + 255) / 256 * 256; }
__host__ __device__ constexpr int tl_wave_bytes(int r, int nb) { return 2 * tl_stage(r, nb) + 2 * tl_mplane(r, nb) + 256; }   // + pad the last row's column overhang reads into
__host__ __device__ constexpr int tl_lds_bytes(int r, int nw, int nb) { return nw * tl_wave_bytes(r, nb); }

// NC: classes (compile time: the epilogue is straight-line code for exactly this count); FULL: also store logits / probabilities
// (the hot path of the deploy loop asks for the label map only)
// STRIP (r06): the tiles of a worker run DOWN a column strip (image, tile column) in segments of a.seg_tiles tiles, and the HR0 - R = 4 halo
// rows two vertically adjacent tiles share stay in LDS: after the first tile of a segment only the R new halo rows are requested from memory
// (8 of 12: a third of the tail's input bytes, the largest single stream of the bf16 forward), the shared rows are moved from the bottom of the
// wave's stage to its top by an LDS -> LDS copy before the new rows are parked.  Every fragment address stays an immediate.
template <int R, int NW, int NB, int NC, bool FULL, bool STRIP = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void unet_tail_kernel(const TailArgs a) {
    constexpr int TL_TW = tl_tw(NB), TL_MW = tl_mw(NB), TL_IW = tl_iw(NB);
    constexpr int R1 = R + 2, HR0 = R + 4, HP0 = tl_hp(R, NB), NLD = tl_nld(R, NB), STAGE = tl_stage(R, NB), MPLANE = tl_mplane(R, NB);
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int n16 = lane & 15, kq = lane >> 4;          // MFMA column (pixel) and k-quarter / row group of this lane
    unsigned char *const wl = lds + wave * tl_wave_bytes(R, NB);      // [2 sources][2 halves][HP0] | mid [2 halves][R1 x 32] | pad
    unsigned char *const mid = wl + 2 * STAGE;

    const int tiles_x = (a.W + TL_TW - 1) / TL_TW, tiles_y = (a.H + R - 1) / R, tiles = tiles_x * tiles_y, ntiles = a.N * tiles;
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
    // STRIP: unit u = segment u % segs of strip u / segs; strip = (image, tile column); a segment = seg consecutive tile rows
    const int seg = STRIP ? (a.seg_tiles > 0 ? a.seg_tiles : tiles_y) : 1;
    const int segs = (tiles_y + seg - 1) / seg, nunits = a.N * tiles_x * segs;
    int my = 0;
    if constexpr (STRIP) {
        for (int u = worker; u < nunits; u += nworkers) { const int j = u % segs; my += (j + 1) * seg <= tiles_y ? seg : tiles_y - j * seg; }
    } else {
        my = worker < ntiles ? (ntiles - worker + nworkers - 1) / nworkers : 0;
    }
    if (my == 0) return;                                // no barrier anywhere: a wave may simply leave

    // ---- A fragments and biases into registers, once ----
    u32x4 A0[9], A1[5];
#pragma unroll
    for (int t = 0; t < 9; ++t) A0[t] = reinterpret_cast<const u32x4 *>(a.wA0)[t * 64 + lane];
#pragma unroll
    for (int p = 0; p < 5; ++p) A1[p] = reinterpret_cast<const u32x4 *>(a.wA1)[p * 64 + lane];
    const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(a.b0 + 4 * kq), bias1 = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * kq);
    // logits weights as A fragments, FOUR row-shifted copies: copy u puts class c in row 4 u + c, so that the products of four (row, block)
    // units accumulate into ONE 16 x 16 result whose lane quarter u holds unit u's classes of its pixel -- one argmax and one store
    // instruction then serve 64 pixels with every lane at work (as one result per unit three quarters of the lanes computed rows
    // that are no classes: 4972 of 13703 cycles per tile, r04 stamps).  k-slot 8 kq + i = channel 4 kq + i for i < 4 (the channels
    // stage 2 leaves in this lane quarter's accumulator), zero above; fp32 weight = bf16 hi + bf16 lo.
    u32x4 Alg[4][2];
    f32x4 lgC;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        unsigned hi[4], lo[4];
        const int c = n16 - 4 * u;                      // this lane's A row n16 is class c of copy u
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float w = (c >= 0 && c < NC) ? a.lg_w[(4 * kq + i) * NC + c] : 0.f;
            f32x2 t2; t2.x = w; t2.y = 0.f;
            const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(t2, bf16x2)) & 0xffffu;
            t2.x = w - __builtin_bit_cast(float, hb << 16);
            hi[i] = hb; lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t2, bf16x2)) & 0xffffu;
        }
        Alg[u][0] = u32x4{hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), 0u, 0u};
        Alg[u][1] = u32x4{lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) lgC[i] = i < NC ? a.lg_b[i] : 0.f;     // every quarter's rows 4 kq + c carry the bias of class c

    // ---- staging geometry (tile independent): piece p = lane + 64 i is half p & 1 of halo pixel p >> 1 (idle lanes of the last
    //      instruction re-fetch the last pixel into the stage's padding) ----
    unsigned geo[NLD], pbase[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = lane + 64 * i, gg = p & 1, px = (p >> 1) < HP0 ? (p >> 1) : HP0 - 1;
        const int hy = px / TL_IW, hx = px - hy * TL_IW;
        geo[i] = (unsigned)hy | ((unsigned)hx << 8);
        pbase[i] = (unsigned)((hy * a.W + hx) * 32 + 16 * gg);
    }
    const unsigned char *const in0 = reinterpret_cast<const unsigned char *>(a.in0);
    const unsigned char *const in1 = reinterpret_cast<const unsigned char *>(a.in1);
    const int img_bytes = a.H * a.W * 32;

    // ---- tile cursors: (image, tile row, tile column) advanced by the worker stride with carries -- scalar adds and compares instead
    //      of two integer divisions per tile ----
    struct Cur { int n, ty, tx, unit, ty_end; bool first; };      // first: first tile of its segment (STRIP): the whole halo comes from memory
    const int st_n = nworkers / tiles, st_y = (nworkers % tiles) / tiles_x, st_x = (nworkers % tiles) % tiles_x;
    auto open_unit = [&](Cur &c, int u) {               // STRIP: two integer divisions per SEGMENT
        c.unit = u;
        const int strip = u / segs, j = u - strip * segs;
        c.n = strip / tiles_x; c.tx = strip - c.n * tiles_x;
        c.ty = j * seg; c.ty_end = c.ty + seg < tiles_y ? c.ty + seg : tiles_y;
        c.first = true;
    };
    auto advance = [&](Cur &c) {
        if constexpr (STRIP) {
            c.first = false;
            if (++c.ty == c.ty_end) open_unit(c, c.unit + nworkers);     // past the last unit: coordinates of no tile; request(false) ignores them
        } else {
            c.tx += st_x; const int cx = c.tx >= tiles_x ? 1 : 0; c.tx -= cx * tiles_x;
            c.ty += st_y + cx; const int cy = c.ty >= tiles_y ? 1 : 0; c.ty -= cy * tiles_y;
            c.n += st_n + cy;
        }
    };
    Cur cl, cc;                                         // load cursor (one tile ahead) and compute cursor
    if constexpr (STRIP) open_unit(cl, worker);
    else { cl.n = worker / tiles; cl.ty = (worker % tiles) / tiles_x; cl.tx = (worker % tiles) % tiles_x; cl.unit = 0; cl.ty_end = 0; cl.first = true; }
    cc = cl;
    // STRIP: pieces of the halo rows a tile shares with the one above it (rows 0 .. HR0 - R - 1 = the first SH_PIECES pieces of a stage)
    constexpr int SH_ROWS = HR0 - R, SH_PIECES = 2 * SH_ROWS * TL_IW, SH_NI = (SH_PIECES + 63) / 64, SH_FULL = SH_PIECES / 64;
    constexpr int SH_SRC = R * TL_IW * 32;              // byte offset of halo row R (the first shared row as the tile above saw it)
    static_assert(!STRIP || SH_SRC >= SH_PIECES * 16, "strip mode: the shared rows' old and new places must not overlap");
    u32x4 xq[2][NLD];
    auto request = [&](bool valid) {                    // both inputs' halos of the load cursor's tile -> registers (past the last tile: zeros)
        const int oy0 = cl.ty * R, ox0 = cl.tx * TL_TW;
        const int n = valid ? cl.n : 0;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)(in0 + (size_t)n * img_bytes), 0, valid ? img_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(in1 + (size_t)n * img_bytes), 0, valid ? img_bytes : 0, 0x00020000);
        const int toff = ((oy0 - 2) * a.W + (ox0 - 2)) * 32;
        unsigned vo[NLD];
        // tiles whose whole halo lies inside the image (all but the border tiles) need no per-piece test
        if (oy0 >= 2 && oy0 + R + 2 <= a.H && ox0 >= 2 && ox0 + TL_IW - 2 <= a.W) {
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
        // STRIP, not the first tile of its segment: the shared rows are in LDS already -- whole instructions of them are not issued, the
        // one instruction that straddles the boundary fetches its shared pieces out of range (no memory traffic; their LDS slots are
        // overwritten by the copy)
        const bool fresh = !STRIP || cl.first;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (STRIP && i < SH_FULL && !fresh) continue;
            if (STRIP && i == SH_FULL && SH_FULL < SH_NI && !fresh && lane + 64 * i < SH_PIECES) vo[i] = OOB;
            xq[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rs0, vo[i], 0, 0);
            xq[1][i] = __builtin_amdgcn_raw_buffer_load_b128(rs1, vo[i], 0, 0);
        }
    };
    auto park = [&](bool fresh) {                       // fresh: the whole halo was requested (always, unless STRIP)
        [[maybe_unused]] u32x4 sh[2][STRIP ? SH_NI : 1];
        if constexpr (STRIP) {
            if (!fresh) {                                // the rows shared with the tile above: read at their old place before anything is stored
#pragma unroll
                for (int i = 0; i < SH_NI; ++i) {
                    sh[0][i] = *reinterpret_cast<const u32x4 *>(wl + SH_SRC + lane * 16 + i * 1024);
                    sh[1][i] = *reinterpret_cast<const u32x4 *>(wl + STAGE + SH_SRC + lane * 16 + i * 1024);
                }
            }
            // The stores below hit, through OTHER lanes, the bytes just read: per thread the two address sets (same lane term, different
            // constants) never alias, so hipcc may sink the reads behind the stores -- the first build did, and the shared rows came out as
            // the NEW tile's rows.  The LDS executes a wave's instructions in order; the compiler has to keep them in order.
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (STRIP && i < SH_FULL && !fresh) continue;     // never requested (request())
            *reinterpret_cast<u32x4 *>(wl + lane * 16 + i * 1024) = xq[0][i];
            *reinterpret_cast<u32x4 *>(wl + STAGE + lane * 16 + i * 1024) = xq[1][i];
        }
        if constexpr (STRIP) {
            if (!fresh) {
#pragma unroll
                for (int i = 0; i < SH_NI; ++i) {
                    if (i < SH_FULL || lane + 64 * i < SH_PIECES) {
                        *reinterpret_cast<u32x4 *>(wl + lane * 16 + i * 1024) = sh[0][i];
                        *reinterpret_cast<u32x4 *>(wl + STAGE + lane * 16 + i * 1024) = sh[1][i];
                    }
                }
            }
        }
    };

    // per-lane LDS read bases
    //   stage 1: k-quarters 0, 1 -> skip stage halves 0, 1; 2, 3 -> up stage halves 0, 1; pixel n16
    const unsigned char *const s1_lane = wl + (kq >> 1) * STAGE + n16 * 32 + (kq & 1) * 16;
    //   stage 2: k-quarters 0, 1 -> tap t (halves 0, 1); 2, 3 -> tap t + 1 (halves 0, 1): the tap difference goes into the per-pair lane base
    const unsigned char *s2_lane[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int t = 2 * p + (kq >> 1);
        const int tt = t < 9 ? t : 8;                   // pair 4 has no second tap (zero weights): read tap 8 again, any finite value will do
        s2_lane[p] = mid + (kq & 1) * MPLANE + ((tt / 3) * TL_MW + (tt % 3) + n16) * 16;
    }
    unsigned char *const mid_w = mid + (kq >> 1) * MPLANE + n16 * 16 + (kq & 1) * 8;      // this lane's 4 channels 4 kq .. of pixel n16: half kq >> 1, bytes 8 (kq & 1) ..
    const int npx = a.H * a.W;

#ifdef UKBB_DIAG
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define UKBB_TL_STAMP(I) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[I] += t_ - tprev; tprev = t_; }
#else
#define UKBB_TL_STAMP(I)
#endif
    auto compute = [&]() {                              // the compute cursor's tile (always a real one)
        const int n = cc.n, oy0 = cc.ty * R, ox0 = cc.tx * TL_TW;
        constexpr bool valid = true;
        // ---- stage 1: up0_0 on R1 rows x 2 blocks ----
        // LDS fragment reads software-pipelined by hand (kernels_ws.hip: left to hipcc every read is waited for with lgkmcnt(0) right
        // behind it, and with one wave per SIMD nothing covers the ~120 cycles): step s = (block, kw, halo row) reads its B fragment
        // PD1 steps ahead into a rotating buffer; sched_barrier pins the order.
        f32x4 acc1[R1][NB];
        {
            constexpr int S1 = NB * 3 * HR0, PD1 = 4, NB1 = 5;
            u32x4 Bq[NB1];
            auto readB = [&](auto sc) {
                constexpr int s = decltype(sc)::value, blk = s / (3 * HR0), kw = (s / HR0) % 3, rp = s % HR0;
                Bq[s % NB1] = *reinterpret_cast<const u32x4 *>(s1_lane + (rp * TL_IW + 16 * blk + kw) * 32);
            };
            unroll_steps<PD1>([&](auto sc) { readB(sc); });
            unroll_steps<S1>([&](auto sc) {
                constexpr int s = decltype(sc)::value, blk = s / (3 * HR0), kw = (s / HR0) % 3, rp = s % HR0;
                if constexpr (s + PD1 < S1) readB(std::integral_constant<int, s + PD1>{});
                __builtin_amdgcn_sched_barrier(0);
                unroll_steps<3>([&](auto khc) {
                    constexpr int kh = decltype(khc)::value, r = rp - kh;
                    if constexpr (r >= 0 && r < R1) {
                        if constexpr (kw == 0 && kh == 0) acc1[r][blk] = mfma16(A0[kh * 3 + kw], Bq[s % NB1], bias0);
                        else acc1[r][blk] = mfma16(A0[kh * 3 + kw], Bq[s % NB1], acc1[r][blk]);
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        UKBB_TL_STAMP(2)
        // ---- mid tile: ReLU, zero outside the image (only border tiles have such pixels), bf16 ----
        auto mid_store = [&](auto maskc) {
            constexpr bool MASK = decltype(maskc)::value;
            unroll_steps<R1>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const bool rowok = (unsigned)(oy0 - 1 + r) < (unsigned)a.H;
                unroll_steps<NB>([&](auto bc) {
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
                    *reinterpret_cast<u32x2 *>(mid_w + (r * TL_MW + 16 * blk) * 16) = pk;
                });
            });
        };
        if (oy0 >= 1 && oy0 + R + 1 <= a.H && ox0 >= 1 && ox0 + TL_MW - 1 <= a.W) mid_store(std::false_type{});
        else mid_store(std::true_type{});
        UKBB_TL_STAMP(3)
        // ---- stage 2: up0_1 on R rows x 2 blocks, two taps per MFMA; one read per 16-cycle MFMA: reads PD2 steps ahead ----
        f32x4 acc2[R][NB];
        {
            constexpr int S2 = R * NB * 5, PD2 = 8, NB2 = 9;
            u32x4 Bq[NB2];
            auto readB = [&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / (5 * NB), blk = (s / 5) % NB, p = s % 5;
                Bq[s % NB2] = *reinterpret_cast<const u32x4 *>(s2_lane[p] + (r * TL_MW + 16 * blk) * 16);
            };
            unroll_steps<PD2>([&](auto sc) { readB(sc); });
            unroll_steps<S2>([&](auto sc) {
                constexpr int s = decltype(sc)::value, r = s / (5 * NB), blk = (s / 5) % NB, p = s % 5;
                if constexpr (s + PD2 < S2) readB(std::integral_constant<int, s + PD2>{});
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p == 0) acc2[r][blk] = mfma16(A1[p], Bq[s % NB2], bias1);
                else acc2[r][blk] = mfma16(A1[p], Bq[s % NB2], acc2[r][blk]);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        UKBB_TL_STAMP(4)
        // ---- logits on the matrix pipe, softmax / argmax, stores ----
        const __amdgpu_buffer_rsrc_t ro_pred = __builtin_amdgcn_make_buffer_rsrc((void *)(a.pred + (size_t)n * npx), 0, a.pred ? npx * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro_lg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.logits + (size_t)n * npx * NC), 0, (FULL && a.logits) ? npx * NC * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro_pr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.prob + (size_t)n * npx * NC), 0, (FULL && a.prob) ? npx * NC * 4 : 0, 0x00020000);
        // units (row r, block blk) in groups of four: unit u lands in lane quarter u.  NB = 2: unit u = (row 2 q + (u >> 1), block u & 1);
        // NB = 1: unit u = row 4 q + u
        static_assert(R % (4 / NB) == 0, "rows pair up in the logits groups");
        constexpr int GR = 4 / NB;                       // rows per group
        const int c0 = NB == 2 ? 16 * (kq & 1) + n16 : n16;     // this lane's column in the tile and ...
        const bool colok = c0 < TL_TW && ox0 + c0 < a.W;
        unroll_steps<R / GR>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            f32x4 lg = lgC;
            unroll_steps<4>([&](auto uc) {
                constexpr int u = decltype(uc)::value, r = NB == 2 ? 2 * q + (u >> 1) : 4 * q + u, blk = NB == 2 ? (u & 1) : 0;
                const float e0 = acc2[r][blk][0], e1 = acc2[r][blk][1], e2 = acc2[r][blk][2], e3 = acc2[r][blk][3];
                f32x2 lo2, hi2;
                lo2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), 0)); lo2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), 0));
                hi2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e2), 0)); hi2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e3), 0));
                const u32x4 bq = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2)), 0u, 0u};
                lg = mfma16(Alg[u][1], bq, lg);         // smaller term first
                lg = mfma16(Alg[u][0], bq, lg);
            });
            const int oy = oy0 + GR * q + (NB == 2 ? (kq >> 1) : kq);     // ... its row in this group
            const bool own = valid && colok && oy < a.H;
            const unsigned px = (unsigned)(oy * a.W + ox0 + c0);
            float l[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) l[c] = lg[c];
            if constexpr (FULL) {
                float p[NC];
                const int best = softmax_argmax<NC>(l, p);
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)best, ro_pred, own ? px * 4u : OOB, 0, 0);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p[c]), ro_pr, own ? (px * NC + c) * 4u : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l[c]), ro_lg, own ? (px * NC + c) * 4u : OOB, 0, 0);
                }
            } else {
                const int best = softmax_argmax<NC>(l, nullptr);
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)best, ro_pred, own ? px * 4u : OOB, 0, 0);
            }
        });
        UKBB_TL_STAMP(5)
    };

    // zero the pad once (the column overhang of the last mid row reads it)
    if (lane < 16) *reinterpret_cast<u32x4 *>(wl + 2 * STAGE + 2 * MPLANE + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    request(true);
#pragma unroll 1
    for (int k = 0; k < my; ++k) {
        park(cl.first);                                 // tile k: registers -> LDS (waits for its loads)
        UKBB_TL_STAMP(0)
        advance(cl);
        request(k + 1 < my);                            // tile k + 1 in flight during everything below
        UKBB_TL_STAMP(1)
        compute();
        advance(cc);
    }
#ifdef UKBB_DIAG
    if (a.stamps && lane == 0) {
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = ph[i];
        o[7] = (unsigned long long)my;
    }
#endif
}

}  // namespace

// Host side: A fragments of v_mfma_f32_16x16x32_bf16 (lane l: row l & 15, k = 8 (l >> 4) + j, j = 0..7; two bf16 per dword).
//   wA0 [9 taps][64 lanes][4 dwords]: up0_0, k = input channel of concat[skip 0..15, up 16..31], row = output channel
//   wA1 [5 pairs][64][4]: up0_1, k < 16: tap 2 p, channel k; k >= 16: tap 2 p + 1, channel k - 16 (zero for the missing tap 9)
static inline unsigned short tl_bf16(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
void pack_tail_weights(const float *w0 /*[3][3][32][16] folded*/, const float *w1 /*[3][3][16][16] folded*/, float *dst0 /*9*64*4*/, float *dst1 /*5*64*4*/) {
    unsigned *o0 = reinterpret_cast<unsigned *>(dst0), *o1 = reinterpret_cast<unsigned *>(dst1);
    for (int t = 0; t < 9; ++t)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                const int m = l & 15, k = 8 * (l >> 4) + 2 * d;
                o0[(t * 64 + l) * 4 + d] = (unsigned)tl_bf16(w0[((size_t)t * 32 + k) * 16 + m]) | ((unsigned)tl_bf16(w0[((size_t)t * 32 + k + 1) * 16 + m]) << 16);
            }
    for (int p = 0; p < 5; ++p)
        for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
                const int m = l & 15, k = 8 * (l >> 4) + 2 * d, t = 2 * p + (k >> 4), c = k & 15;
                const float v0 = t < 9 ? w1[((size_t)t * 16 + c) * 16 + m] : 0.f, v1 = t < 9 ? w1[((size_t)t * 16 + c + 1) * 16 + m] : 0.f;
                o1[(p * 64 + l) * 4 + d] = (unsigned)tl_bf16(v0) | ((unsigned)tl_bf16(v1) << 16);
            }
}

hipError_t launch_unet_tail(const TailArgs &a_in, hipStream_t s) {
    TailArgs a = a_in;
    a.stamps = nullptr;
    if (!a.in0 || !a.in1 || !a.wA0 || !a.wA1 || !a.b0 || !a.b1 || !a.lg_w || !a.lg_b || a.ncls < 2 || a.ncls > 4 || a.N < 1) return hipErrorInvalidValue;
    if ((long long)a.H * a.W * 32 >= 0x7fffffffll) return hipErrorInvalidValue;
#if defined(UKBB_TAIL_R)
    constexpr int R = UKBB_TAIL_R, NW = UKBB_TAIL_NW, NB = UKBB_TAIL_NB;   // A/B builds
#else
    constexpr int R = 8, NW = 8, NB = 1;
#endif
    constexpr int TL_TW = tl_tw(NB);
    const long long ntiles = (long long)a.N * ((a.H + R - 1) / R) * ((a.W + TL_TW - 1) / TL_TW);
    const int cus = device_cu_count();
    const long long want = (ntiles + NW - 1) / NW;
    const int grid = (int)(want < cus ? want : cus);
    constexpr int bytes = tl_lds_bytes(R, NW, NB);
    static_assert(bytes <= 160 * 1024, "LDS");
    const bool full = a.logits || a.prob;
    // strip mode (r06; UKBB_TAIL_STRIPS=0 = the row-major walk of r04 / r05): segments per strip chosen so that the busiest worker has the
    // fewest tiles -- ceil(units / workers) * seg -- and among equals the longest segments (each segment re-reads 4 halo rows).
    // Measured at N = 100 x 256x256 (profiles/r06_ab_tail.txt): identical bits; HBM traffic of the launch 674 -> 513 MB (counters: 2 x FETCH_SIZE +
    // WRITE_SIZE; L2 hit rate 0.30 -> 0.40), the forward's 4128 -> 3966 MB; time unchanged (158.5 vs 162.8 us under rocprofv3, minima 151.2 / 150.2;
    // forward +0.75 %): the tail is not bound by its bytes, the strips are the default for the traffic they save.
    const char *es = getenv("UKBB_TAIL_STRIPS");
    const bool strips = !es || atoi(es) != 0;
    a.seg_tiles = 0;
    if (strips) {
        const int tiles_y = (a.H + R - 1) / R, tiles_x = (a.W + TL_TW - 1) / TL_TW;
        const long long nstrips = (long long)a.N * tiles_x, workers = (long long)grid * NW;
        long long best = -1; int best_seg = tiles_y;
        for (int segs = 1; segs <= tiles_y; ++segs) {
            const int seg = (tiles_y + segs - 1) / segs;
            if ((tiles_y + seg - 1) / seg != segs) continue;                   // not a distinct split
            const long long units = nstrips * segs, cost = (units + workers - 1) / workers * seg;
            if (best < 0 || cost < best) { best = cost; best_seg = seg; }
        }
        if (const char *e = getenv("UKBB_TAIL_SEG")) { const int v = atoi(e); if (v >= 1) best_seg = v; }
        a.seg_tiles = best_seg;
    }
#ifdef UKBB_DIAG
    static unsigned long long *d_st = nullptr;
    const bool stamp = getenv("UKBB_TAIL_STAMPS") != nullptr;
    if (stamp) {
        if (!d_st && hipMalloc(reinterpret_cast<void **>(&d_st), 4096 * 8 * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, 4096 * 8 * 8, s);
        a.stamps = d_st;
    }
    struct Dump {
        bool on; hipStream_t s; unsigned long long *d; int nwv;
        ~Dump() {
            if (!on) return;
            static int shots = 0;
            if (++shots != 6) return;
            std::vector<unsigned long long> h((size_t)nwv * 8);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            double m[8] = {0}; int n = 0;
            for (int w = 0; w < nwv; ++w) if (h[(size_t)w * 8 + 7]) { ++n; for (int i = 0; i < 7; ++i) m[i] += (double)h[(size_t)w * 8 + i] / (double)h[(size_t)w * 8 + 7]; }
            if (!n) return;
            fprintf(stderr, "[tail stamps] %d waves, cycles per tile: park(+load wait) %.0f, request %.0f, stage1 %.0f, mid %.0f, stage2 %.0f, logits+argmax+stores %.0f; total %.0f\n",
                    n, m[0] / n, m[1] / n, m[2] / n, m[3] / n, m[4] / n, m[5] / n, (m[0] + m[1] + m[2] + m[3] + m[4] + m[5]) / n);
        }
    } dump{stamp, s, d_st, grid * NW > 4096 ? 4096 : grid * NW};
#endif
    auto go = [&](auto k, OncePerDevice &lds_ok) -> hipError_t {
        hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), bytes, s, a);
        return hipGetLastError();
    };
#define UKBB_TAIL_CASE(NC)                                                                         \
    case NC: {                                                                                     \
        static OncePerDevice ok_full, ok_pred;                                                     \
        static OncePerDevice ok_full_s, ok_pred_s;                                                 \
        if (strips) return full ? go(unet_tail_kernel<R, NW, NB, NC, true, true>, ok_full_s) : go(unet_tail_kernel<R, NW, NB, NC, false, true>, ok_pred_s); \
        return full ? go(unet_tail_kernel<R, NW, NB, NC, true>, ok_full) : go(unet_tail_kernel<R, NW, NB, NC, false>, ok_pred); \
    }
    switch (a.ncls) {
        UKBB_TAIL_CASE(2)
        UKBB_TAIL_CASE(3)
        UKBB_TAIL_CASE(4)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ukbb
