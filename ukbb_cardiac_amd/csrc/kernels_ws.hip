// Weight-stationary persistent 3x3 stride-1 convolution (+ folded BN bias, ReLU) for UKBB_PREC_BF16 of the aortic U-Net:
// bf16 NHWC activations in HBM on both sides, v_mfma_f32_32x32x16_bf16, fp32 accumulation.  Replaces conv2d_bn_relu
// (reference common/network.py:19-25) as used by the encoder / decoder blocks of common/network_ao.py:31-55, including the
// skip concat of :51 as a two-source K loop.
//
// Why (r04): the tile-per-workgroup kernel (conv_mfma_kernel<..., BFIO>) spends a workgroup's life in
// a chain of (load -> LDS -> barrier -> 18-36 MFMAs -> barrier) stages: levels 1-3 took 44-90 us per layer for 12-24 us of
// matrix time, every workgroup re-streamed its Cout block's packed weights from L2 (conv3_1: 236 MB per launch, nine times the
// activation map), and the matrix pipe was busy 42 % of a workgroup's lifetime at four workgroups per CU.  Here
//   * a workgroup OWNS one group of 32 x CB output channels for the whole launch: its packed weights for ALL input channels
//     and taps are copied to LDS once (<= 72 KB) and never move again;
//   * every WAVE is an independent worker that walks its own row tiles (R rows x 32 pixels) of the batch: it stages the
//     16-channel chunks of its halo tile into a PRIVATE two-stage LDS ring (global -> registers -> LDS, two register sets, the
//     loads running three chunks ahead of the MFMAs that consume them) and multiplies from there.  After the weight copy there
//     is NO barrier and no inter-wave dependence at all; the latency of one wave's loads is covered by its own earlier
//     chunks and by the other waves of the SIMD;
//   * LDS operand traffic per MFMA is 0.5-0.75 ds_read_b128: a pixel block is one image row x 32 pixels, so the B fragment
//     of halo row r' at column shift kw serves the three output rows r' - kh; the stage image is [k half][halo pixel] x 16 B,
//     linear in the order the loads arrive (conflict-free ds_write_b128) and conflict-free for the fragment reads
//     (consecutive lanes = consecutive pixels = consecutive 16-byte slots);
//   * image borders cost nothing in the steady state: halo pieces outside the image are requested with an out-of-range
//     buffer offset (the hardware returns 0 = the conv's zero padding), ghost tiles past a worker's last tile likewise, and
//     stores of pixels / channels that do not exist are dropped the same way -- the loop body has no branch around a
//     vector-memory instruction, so hipcc's counted s_waitcnt vmcnt(n) keeps the younger register set in flight.
// Fragment layouts and packed weights are those of conv_mfma_kernel<..., BFIO> (pack_conv_weights_bf16 with ncbl = CB);
// the accumulation order over (chunk, tap) differs (kw outermost inside a chunk), so results agree with that kernel to fp32
// rounding of the accumulator, not bit for bit.
//
// KIND = 1 is the same machine for conv2d_transpose 3x3 stride 2 'SAME' + BN + ReLU (reference common/network.py:28-34 as used
// by network_ao.py:48-49) in its 4-phase sub-pixel form (kernels.h, tconv_as_conv2x2): a tile is R INPUT rows x 32 input pixels
// (2R x 64 outputs), the halo is one row above and one column left, the four 2x2 taps (a, b) read input (m + a - 1, x + b - 1),
// and output phase (py, px) uses tap (a, b) only if (py == 0 || a == 1) && (px == 0 || b == 1): 9 of the 16 (tap, phase) pairs.
// A workgroup owns TWO 32-row blocks of the 4 x Cout phase-major "virtual" channels, paired so that every workgroup has about the
// same work -- {phase 00, phase 11} (4 + 1 taps) or {phase 01, phase 10} (2 + 2) of one 32-channel slice; for Cout = 16 the
// blocks are {00 + 01} and {10 + 11} -- and the zero (tap, phase) pairs are skipped at COMPILE time (one specialisation per pairing).
#include "kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace ukbb {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 &a, const u32x4 &b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 16-byte buffer store whose soffset is an SGPR, followed by pinned wait states.
// gfx950, found r06 (profiles/r06_notes.md section 10): the vector-memory unit reads the data registers of a store wider than 8 bytes over
// several cycles AFTER issue, and a VALU write to one of them in the next issue slots changes what is stored.  hipcc pads this hazard only
// when the store's soffset is a constant (LLVM GCNHazardRecognizer::createsVALUHazard assumes an SGPR soffset makes it impossible);
// with an SGPR soffset it emitted "buffer_store_dwordx4 v[128:131], ... s55 offen" directly followed by "v_cvt_pk_bf16_f32 v128, ..."
// (the packed registers of the next store).  With the memory pipeline to itself the store drains before the overwrite lands; under
// back-pressure from another queue's kernels it does not, and whole 16-byte pieces of a map carried the NEXT piece's values (logits
// of 1e33 a few layers later).  s_nop 7 x 2 behind every such store measured 0 failures in 160 two-stream forwards against 85 %
// without, at no cost in time; sched_barrier keeps both schedulers from moving anything across.  tests/test_store_hazard.py checks
// the ISA of every code object for the pattern.
__device__ __forceinline__ void store_b128_sofs(const u32x4 &v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, soff, 0);
    __builtin_amdgcn_sched_barrier(0);
#if !defined(UKBB_STORE_PAD) || UKBB_STORE_PAD == 4
    asm volatile("s_waitcnt expcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#elif UKBB_STORE_PAD == 1                                  // experiment builds (tools/ab_store_pad.sh): how little is enough
    asm volatile("s_nop 0" ::: "memory");
#elif UKBB_STORE_PAD == 2
    asm volatile("s_waitcnt expcnt(0)" ::: "memory");
#elif UKBB_STORE_PAD == 3
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#elif UKBB_STORE_PAD == 5
    asm volatile("s_nop 3" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;       // RNE; v_cvt_pk_bf16_f32
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void unroll_steps(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); unroll_steps<N, I + 1>(f); }
}

constexpr int WS_TW = 32;                               // tile width = one 32-pixel MFMA block per image row
// KIND 0: 3x3 stride 1 (halo R + 2 rows x 34 pixels, 9 taps); KIND 1: transposed conv as 2x2 sub-pixel conv (R + 1 rows x 33 pixels, 4 taps)
__host__ __device__ constexpr int ws_hr(int kind, int r) { return kind == 0 ? r + 2 : r + 1; }
__host__ __device__ constexpr int ws_iw(int kind) { return kind == 0 ? WS_TW + 2 : WS_TW + 1; }
__host__ __device__ constexpr int ws_taps(int kind) { return kind == 0 ? 9 : 4; }
__host__ __device__ constexpr int ws_hp(int kind, int r) { return ws_hr(kind, r) * ws_iw(kind); }
__host__ __device__ constexpr int ws_nld(int kind, int r) { return (2 * ws_hp(kind, r) + 63) / 64; }     // 16-byte pieces per lane and stage
// one k-half plane of a stage: halo pixels x 16 bytes, padded so that the planes lie an odd multiple of 64 bytes apart modulo 128: the
// 8 lanes of a ds_write_b128 group (4 pixels x 2 halves) then fall on 8 different 16-byte slots
__host__ __device__ constexpr int ws_plane_bytes(int kind, int r) {
    return (ws_hp(kind, r) * 16 / 128) * 128 + 64 + (ws_hp(kind, r) * 16 % 128 > 64 ? 128 : 0);
}
__host__ __device__ constexpr int ws_stage_bytes(int kind, int r) { return (2 * ws_plane_bytes(kind, r) + 32 + 127) / 128 * 128; }
__host__ __device__ constexpr int ws_lds_bytes(int kind, int r, int cb, int nw, int nch) {
    return nch * cb * ws_taps(kind) * 1024 + cb * 128 + nw * 2 * ws_stage_bytes(kind, r);
}

// Transposed conv (KIND 1): does block `blk` of pairing TM use tap (a, b)?  TM 0: blocks {phase 00, phase 11}; TM 1: {01, 10};
// TM 2 (Cout = 16, a block holds two phases): {00 + 01, 10 + 11}.  Phase (py, px) uses (a, b) iff (py == 0 || a == 1) && (px == 0 || b == 1).
__host__ __device__ constexpr bool wst_needed(int tm, int blk, int a, int b) {
    return tm == 0 ? (blk == 0 ? true : (a == 1 && b == 1))
         : tm == 1 ? (blk == 0 ? b == 1 : a == 1)
                   : (blk == 0 ? true : a == 1);
}
// first tap of a block in step order (b outer, a inner): its MFMA takes the bias as C
__host__ __device__ constexpr bool wst_first(int tm, int blk, int a, int b) {
    for (int bb = 0; bb < 2; ++bb)
        for (int aa = 0; aa < 2; ++aa)
            if (wst_needed(tm, blk, aa, bb)) return aa == a && bb == b;
    return false;
}

// KIND: 0 = 3x3 stride 1, 1 = transposed conv (sub-pixel 2x2); TM: KIND 1 only, the block pairing (wst_needed)
// LG (KIND 0, one block of 16 real channels): the 1x1 logits conv + softmax / argmax of network_ao.py:63,159-160 in the epilogue; the
//     conv's own output is rounded to bf16 as a store would have done and never written (as conv_mfma_kernel<..., FUSE = 2>)
// LS (KIND 0, CB = 2, one 16-channel chunk): the ConvLSTM cell of network_ao.py:255-319 in the epilogue (ConvArgs::ls_mode 1 | 2, kernels.h) -- the bf16 form of
//     the fused gate-conv / cell kernel (UKBB_PREC_BF16 on a UNet-LSTM handle, r05).  The 64 gate channels are packed (pack_lstm_gate_weights_bf16) so that the
//     accumulators of lane half g ARE the four gates of hidden channels 8 g .. 8 g + 7 of the lane's pixel: block 0 registers 0-7 = i, 8-15 = j, block 1
//     registers 0-7 = f, 8-15 = o.  No data moves between lanes; gx (bf16) and the cell state (fp32) live in the lanes' own order
//     ([image][tile][row][16-byte piece][lane]: every load / store instruction is one contiguous KB), h is written as bf16 NHWC (lane half g = channels 8 g ..:
//     32 pixels x 32 bytes = one contiguous KB per row).  Eight independent waves per workgroup and two per SIMD hide the epilogue's loads and its
//     transcendental chains behind each other's MFMAs -- what the fp32 Winograd form (kernels_wino24.hip) cannot do with its 192 accumulators per wave.
//     LS 3 (r06): a time step with the x half of the gate convolution NOT hoisted -- two chunks, source 0 = the feature frame of window n at this step
//     (in0 through ls_gx_map), source 1 = the previous hidden map (in1 through in0_map), bias as the C operand, no gx anywhere.  The bf16 step is bound by
//     its bytes, not its MFMAs: reading x (32 bytes per pixel) and multiplying again costs 9 more MFMAs per 32 x 32 block and saves the 128 bytes per pixel
//     of gx (320 -> 224 bytes per pixel and step).
template <int KIND, int R, int CB, int NW, int NCH, bool TWO, int TM, bool LG = false, int LS = 0>
__device__ __forceinline__ void ws_main(const ConvArgs &a, const int grp, const int walker, const int nwalk) {
    static_assert(!LG || (KIND == 0 && CB == 1), "fused logits: 3x3 conv with one Cout block");
    static_assert(LS == 0 || (KIND == 0 && CB == 2 && !LG && ((LS != 3 && NCH == 1 && !TWO) || (LS == 3 && NCH == 2 && TWO))),
                  "ConvLSTM epilogue: 16 channels in (LS 3: x 16 + h 16), the 64 gate channels of one direction per workgroup");
    constexpr int WS_IW = ws_iw(KIND), HR = ws_hr(KIND, R), TAPS = ws_taps(KIND);
    constexpr int HP = HR * WS_IW, NLD = ws_nld(KIND, R), STAGE = ws_stage_bytes(KIND, R), PLANE = ws_plane_bytes(KIND, R);
    constexpr int WSLAB = NCH * CB * TAPS * 1024;       // bytes: [chunk][cb][tap][lane][16]
    static_assert(KIND == 0 || CB == 2, "transposed conv: two paired blocks per workgroup");
    constexpr int TPB = NCH % 2 ? 2 : 1;                // tiles per unrolled loop body (the body covers an even number of chunks)
    constexpr int U = TPB * NCH;
    static_assert(!TWO || NCH % 2 == 0, "two sources: equal halves");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const ws = lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
#ifdef UKBB_DIAG
    // diagnostic build only: per-wave cycle stamps (entry, weights resident, prologue done, end of every tile) into a buffer of their own
    unsigned long long *const stamps = a.diag ? reinterpret_cast<unsigned long long *>(const_cast<float *>(a.first_w)) + ((size_t)blockIdx.x * NW + wave) * 16 : nullptr;
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime(), st_rt0 = __builtin_amdgcn_s_memrealtime();
#define UKBB_WS_STAMP(SLOT) { if (stamps && lane == 0 && (SLOT) < 14) stamps[SLOT] = __builtin_amdgcn_s_memtime(); }
#else
#define UKBB_WS_STAMP(SLOT)
#endif
    const int g = lane >> 5, pl = lane & 31;
    unsigned char *const ring = lds + WSLAB + CB * 128 + wave * (2 * STAGE);

    const int tiles = a.tiles_x * a.tiles_y, ntiles = a.N * tiles;
    // wave-major numbering: when the tiles do not divide evenly the workers with one tile more are spread one wave per workgroup
    // (one SIMD of a CU carries 7 tiles, the others 6) instead of filling whole workgroups (8 against 6)
    // Tile t of a round goes to worker t.  Default: wave-major numbering, consecutive tiles on consecutive walkers (= different XCDs).
    // xcd_local (large maps, set by the launcher): the walkers of one XCD (walker & 7 after ws_place) take a contiguous run of tiles and
    // the waves of a workgroup consecutive ones, so vertically adjacent row bands -- which share two halo rows -- are read through ONE
    // XCD's L2 at about the same time.  Measured at N = 100 (r04): 128x128 maps conv1_1 50.5 -> 46.8 us, up1_0 75.4 -> 66.4, up1_1
    // 51.1 -> 49.0; 64x64 and 32x32 maps 1-4 us SLOWER per layer (few tiles per image: the eight waves of a workgroup then sit on
    // one image's rows and queue on the same channels), hence the switch.
    const int per_xcd_ = nwalk / 8, chunk_ = (nwalk % 8 == 0) ? (walker & 7) * per_xcd_ + (walker >> 3) : walker;
    const int worker = a.xcd_local ? chunk_ * NW + wave : wave * nwalk + walker, nworkers = nwalk * NW;
    const int my = worker < ntiles ? (ntiles - worker + nworkers - 1) / nworkers : 0;

    // ---- once per workgroup: this group's packed weights and bias -> LDS.  All of a thread's pieces are requested before the first is
    //      stored (as a load -> store loop the copy was a chain of 9-18 dependent L2 round trips in front of every launch), and the
    //      first three activation chunks are requested behind them BEFORE the weights are waited for (one round trip, not two).
    constexpr int NWP = WSLAB / 16 / (NW * 64);         // 16-byte pieces per thread: NCH * CB * 9 / NW ...
    constexpr int NWR = WSLAB / 16 - NWP * (NW * 64);   // ... and a remainder of fewer than NW * 64 pieces
    u32x4 wq[NWP + 1];
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.wpk) + (size_t)grp * (WSLAB / 16);
#pragma unroll
        for (int i = 0; i < NWP; ++i) wq[i] = src[tid + i * (NW * 64)];
        if constexpr (NWR > 0) wq[NWP] = src[tid < NWR ? tid + NWP * (NW * 64) : 0];
    }

    // ---- per-lane geometry of the staging pieces (tile independent).  Activations are channel-blocked in HBM ([N][C/16][H][W][16],
    //      kernels.h): the 16-channel chunk of a halo row is one contiguous run of 32-byte pixels, and lanes 2i / 2i+1 of a load
    //      instruction fetch the two 16-byte halves of one pixel -- 1 KB of consecutive bytes per instruction.  In LDS the halves
    //      go to their planes: piece p -> [k half p & 1][halo pixel p >> 1].
    unsigned geo[NLD], pbase[NLD];
    unsigned lofs[NLD];                                 // LDS byte offset of the piece inside a stage
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = lane + 64 * i;
        const int gg = p & 1, px = p >> 1;
        const int hy = px / WS_IW, hx = px - hy * WS_IW;
        geo[i] = (unsigned)hy | ((unsigned)hx << 8) | (px < HP ? 0u : OOB);
        pbase[i] = (unsigned)((hy * a.W + hx) * 32 + 16 * gg);
        lofs[i] = (unsigned)((px < HP ? gg * PLANE + px * 16 : 2 * PLANE + (lane & 1) * 16));    // idle lanes: a 32-byte scratch slot behind the planes
    }
    const unsigned char *const in0 = reinterpret_cast<const unsigned char *>(a.in0);
    const unsigned char *const in1 = reinterpret_cast<const unsigned char *>(a.in1);
    const int plane_bytes = a.H * a.W * 32;             // one 16-channel plane of one image
    const int nb0 = a.C0 / 16, nb1 = a.C1 / 16;

    auto tile_coords = [&](int k, int &n, int &oy0, int &ox0) -> bool {
        const bool valid = k < my;
        const int t = valid ? worker + k * nworkers : 0;
        n = t / tiles;
        const int r = t - n * tiles, ty = r / a.tiles_x;
        oy0 = ty * R; ox0 = (r - ty * a.tiles_x) * WS_TW;
        return valid;
    };

    // ---- load cursor: the tile whose chunks are being requested ----
    unsigned voff[NLD];
    __amdgpu_buffer_rsrc_t rs0, rs1;
    // window -> frame tables (LS 2) through the SCALAR cache (constant address space + a wave-uniform index = s_load_dword, waited for on lgkmcnt): a vector
    // load would queue behind three tiles of halo prefetch (vmcnt retires in order)
    typedef const int __attribute__((address_space(4))) *cint_p;
    [[maybe_unused]] const cint_p map0 = (cint_p)(a.in0_map), mapg = (cint_p)(a.ls_gx_map);
    auto load_setup = [&](int k) {
        int n, oy0, ox0;
        const bool valid = tile_coords(k, n, oy0, ox0);
        int n1 = n;                                          // image of source 1
        if constexpr (LS == 2) { if (a.in0_map) n = map0[n]; }   // image n reads frame in0_map[n]
        if constexpr (LS == 3) { if (a.in0_map) n1 = map0[n]; n = mapg[n]; }   // source 0 = the feature frame of window n at this step, source 1 = its previous hidden map
        else n1 = n;
        // range = the image's planes of that source: a chunk's plane is selected by the scalar offset of the load
        rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)(in0 + (size_t)n * nb0 * plane_bytes), 0, nb0 * plane_bytes, 0x00020000);
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)((TWO ? in1 : in0) + (size_t)n1 * (TWO ? nb1 : nb0) * plane_bytes), 0, (TWO ? nb1 : nb0) * plane_bytes, 0x00020000);
        const int toff = ((oy0 - 1) * a.W + (ox0 - 1)) * 32;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int hy = (int)(geo[i] & 0xffu), hx = (int)((geo[i] >> 8) & 0xffu);
            const bool ok = valid && !(geo[i] & OOB) && (unsigned)(oy0 - 1 + hy) < (unsigned)a.H && (unsigned)(ox0 - 1 + hx) < (unsigned)a.W;
            voff[i] = ok ? pbase[i] + (unsigned)toff : OOB;
        }
    };
    u32x4 xq[2][NLD];
    auto issue = [&](auto setc, auto chc) {             // request chunk CH of the load cursor's tile into register set SET
        constexpr int SET = decltype(setc)::value, CH = decltype(chc)::value;
        constexpr bool second = TWO && CH >= NCH / 2;
        const int soff = (second ? CH - NCH / 2 : CH) * plane_bytes;
#pragma unroll
        for (int i = 0; i < NLD; ++i) xq[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(second ? rs1 : rs0, voff[i], soff, 0);
    };
    auto park = [&](auto setc) {                        // register set SET -> ring stage SET: the halves to their planes
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4 *>(ring + SET * STAGE + lofs[i]) = xq[SET][i];
    };

    // ---- compute cursor ----
    // The LDS fragment reads are software-pipelined BY HAND across steps, chunks and tiles (left to itself hipcc reads a B fragment
    // into one buffer and waits lgkmcnt(0) right behind it: r04 first build, 18 exposed LDS round trips per chunk, 3100 cycles for
    // 1152 cycles of MFMA).  A step = one B fragment (halo row rp at column shift kw) and the <= 3 CB MFMAs it feeds; S steps per
    // chunk.  At step T the read of step T + PD goes out into buffer (T + PD) % NB; the A fragments of the next (chunk, kw) group go
    // out at the first step of the current group into the other A set.  sched_barrier pins the order (hipcc would sink the reads to
    // their first use); its counted lgkmcnt(n) waits then leave the younger reads in flight.
    // KIND 0: step = (kw, halo row rp): S = 3 HR, an A group = the three kh taps of column kw.
    // KIND 1: step = (b, halo row rp): S = 2 HR, an A group = the two taps (a, b), a = 0, 1.
    constexpr int NGRP = KIND == 0 ? 3 : 2, NAF = KIND == 0 ? 3 : 2;
    constexpr int S = NGRP * HR, PD = 3, NB = 4;
    static_assert((U * S) % NB == 0 && (U * NGRP) % 2 == 0, "fragment buffers rotate consistently across loop iterations");
    f32x16 acc[CB][R], biasv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (LS != 2) b4 = *reinterpret_cast<const f32x4 *>(a.bias + (grp * CB + cb) * 32 + 8 * j + 4 * g);   // LS 2: the gate bias is part of gx
#pragma unroll
            for (int i = 0; i < 4; ++i) biasv[cb][4 * j + i] = b4[i];
        }
    const unsigned char *const xs_lane = ring + g * PLANE + pl * 16;
    const unsigned char *const ws_lane = ws + lane * 16;
    u32x4 Bq[NB], Aq[2][CB][NAF];
    auto readB = [&](auto tc) {                         // step TT of the body (>= U * S: the first steps of the next loop iteration)
        constexpr int TT = decltype(tc)::value, T = TT % (U * S), Qn = T / S, s = T % S, kw = s / HR, rp = s % HR;
        Bq[TT % NB] = *reinterpret_cast<const u32x4 *>(xs_lane + (Qn & 1) * STAGE + (rp * WS_IW + kw) * 16);
    };
    auto readA = [&](auto gc) {                         // (chunk, kw | b) group GG of the body
        constexpr int GG = decltype(gc)::value, G = GG % (U * NGRP), CH = (G / NGRP) % NCH, kw = G % NGRP;
        unroll_steps<CB>([&](auto cbc) {
            constexpr int cb = decltype(cbc)::value;
            unroll_steps<NAF>([&](auto kc) {
                constexpr int kh = decltype(kc)::value;     // KIND 1: kh = a, kw = b
                constexpr int tap = KIND == 0 ? kh * 3 + kw : kh * 2 + kw;
                if constexpr (KIND == 0 || wst_needed(TM, cb, kh, kw))
                    Aq[GG & 1][cb][kh] = *reinterpret_cast<const u32x4 *>(ws_lane + ((CH * CB + cb) * TAPS + tap) * 1024);
            });
        });
    };
    // ---- LS 2: the tile's gx (bf16, four 16-byte pieces per row = one gate of the lane's eight hidden channels each) and cell state (fp32, two pieces per row)
    //      requested at the START of the tile's matrix phase: they return under its MFMAs and the other waves' work
    [[maybe_unused]] u32x4 gxq[LS == 2 ? R : 1][4];
    [[maybe_unused]] f32x4 cq[LS >= 2 ? R : 1][2];
    [[maybe_unused]] auto ls_prefetch = [&](int k) {
        if constexpr (LS >= 2) {
            int n, oy0, ox0;
            const bool valid = tile_coords(k, n, oy0, ox0);
            const int rtile = (oy0 / R) * a.tiles_x + ox0 / WS_TW;
            const int ci = a.in0_map ? map0[n] : n;
            const unsigned char *cp = reinterpret_cast<const unsigned char *>(a.ls_c_in) + ((size_t)ci * tiles + rtile) * (R * 2048) + lane * 16;
            [[maybe_unused]] const unsigned char *gp = nullptr;
            if constexpr (LS == 2) gp = reinterpret_cast<const unsigned char *>(a.ls_gx) + ((size_t)mapg[n] * tiles + rtile) * (R * 4096) + lane * 16;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr (LS == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) gxq[r][q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(gp + (r * 4 + q) * 1024));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) cq[r][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(cp + (r * 2 + q) * 1024));
            }
        }
    };
    auto compute = [&](auto qc) {                       // position Q of the body: chunk Q % NCH from ring stage Q & 1
        constexpr int Q = decltype(qc)::value, CH = Q % NCH;
        unroll_steps<S>([&](auto sc) {
            constexpr int s = decltype(sc)::value, T = Q * S + s, kw = s / HR, rp = s % HR, G = Q * NGRP + kw;
            readB(std::integral_constant<int, T + PD>{});
            if constexpr (rp == 0) readA(std::integral_constant<int, G + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            unroll_steps<NAF>([&](auto khc) {
                constexpr int kh = decltype(khc)::value, r = rp - kh;
                if constexpr (r >= 0 && r < R) {
                    unroll_steps<CB>([&](auto cbc) {
                        constexpr int cb = decltype(cbc)::value;
                        // the folded-BN bias is the C operand of an accumulator's first MFMA of the tile (no zeroing, no bias adds)
                        if constexpr (KIND == 0) {
                            if constexpr (CH == 0 && kw == 0 && kh == 0) acc[cb][r] = mfma_bf16(Aq[G & 1][cb][kh], Bq[T % NB], biasv[cb]);
                            else acc[cb][r] = mfma_bf16(Aq[G & 1][cb][kh], Bq[T % NB], acc[cb][r]);
                        } else if constexpr (wst_needed(TM, cb, kh, kw)) {
                            if constexpr (CH == 0 && wst_first(TM, cb, kh, kw)) acc[cb][r] = mfma_bf16(Aq[G & 1][cb][kh], Bq[T % NB], biasv[cb]);
                            else acc[cb][r] = mfma_bf16(Aq[G & 1][cb][kh], Bq[T % NB], acc[cb][r]);
                        }
                    });
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // ---- epilogue: ReLU (one v_max_i32 on the bit pattern), bf16 rounding (v_cvt_pk_bf16_f32), 8-byte NHWC stores.  Per-lane store offsets
    //      are tile independent (pixel column and channel of the lane); the tile's row offsets are scalar (soffset); rows below the map
    //      and ghost tiles store through a descriptor of range 0, columns right of the map / channels beyond the real count through an
    //      out-of-range lane offset: dropped by the hardware, no branch.
    unsigned char *const outb = reinterpret_cast<unsigned char *>(a.out);
    // KIND 0: cst real channels at the input's resolution.  KIND 1: a.up2 real channels at twice the resolution; the accumulator rows
    // are (phase, channel) pairs: the phase picks the output pixel (2 m + py, 2 x + px), the channel the plane and the offset in it.
    const int cst = KIND == 1 ? a.up2 : a.cout_store > 0 ? a.cout_store : a.Cout;
    const int OW = KIND == 1 ? 2 * a.Wo : a.Wo, OH = KIND == 1 ? 2 * a.Ho : a.Ho;
    const int out_plane_bytes = OH * OW * 32, out_img_bytes = (cst / 16) * out_plane_bytes;     // channel-blocked like the inputs
    // Stores are 16 bytes per lane: the accumulator gives lane half g channels 8 j + 4 g .. + 3 of its pixel (8 bytes as bf16); for a pair
    // of groups (jb, jb + 1) the halves exchange their packed values with v_permlane32_swap (lower lanes keep group jb and receive the
    // partner's 8 bytes of it, upper lanes group jb + 1), so every lane then holds 8 consecutive channels = one 16-byte half of a pixel
    // of the channel-blocked map.  Half as many store instructions, each 1 KB of whole 16-byte pieces.
    unsigned svoff[CB][2];
    unroll_steps<CB>([&](auto cbc) {
        constexpr int cb = decltype(cbc)::value;
        unroll_steps<2>([&](auto pc) {
            constexpr int jb = 2 * decltype(pc)::value;
            if constexpr (KIND == 0) {
                const int chn = (grp * CB + cb) * 32 + 8 * (jb + g);
                svoff[cb][jb / 2] = chn < cst ? (unsigned)((chn >> 4) * out_plane_bytes + pl * 32 + (chn & 15) * 2) : OOB;
            } else {
                constexpr int ph = TM == 0 ? (cb == 0 ? 0 : 3) : TM == 1 ? (cb == 0 ? 1 : 2) : 2 * cb + (jb >> 1);
                const int co = TM == 2 ? 8 * g : (grp >> 1) * 32 + 8 * (jb + g);
                svoff[cb][jb / 2] = (unsigned)((co >> 4) * out_plane_bytes + (2 * pl + (ph & 1)) * 32 + (co & 15) * 2);
            }
        });
    });
    const int relu_lo = a.relu ? 0 : (int)0x80000000;   // max_i32(bits, 0) = ReLU; max_i32(bits, INT_MIN) = identity
    // fused logits ON THE MATRIX PIPE (as vector FMAs the 16 -> n_class product, with its cross-half shuffles, cost ~120 VALU
    // instructions per 32-pixel row against the row's 9 MFMAs and made this layer VALU-bound): one more K = 16 product per row with
    // A = the logits weights (rows = classes), B = the lane's own 8 activations rounded to bf16 -- the K order is permuted so that
    // k-slot 8 g + i IS the channel the accumulator layout gives lane half g (4 g + i for i < 4, 8 + 4 g + i - 4 above): no data
    // moves between lanes.  The fp32 weights enter as bf16 hi + lo pieces (two MFMAs), so the product is exact to ~2^-17 of a weight;
    // the bias is the C operand.  Lane half 0 then holds the row's n_class logits of its pixel in registers 0..3.
    u32x4 lgA[2];
    f32x16 lgC;
    if constexpr (LG) {
        const int m = lane & 31;
        unsigned hi[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = i < 4 ? 4 * g + i : 8 + 4 * g + (i - 4);
            const float w = m < a.lg_ncls ? a.lg_w[ch * a.lg_ncls + m] : 0.f;
            f32x2 t; t.x = w; t.y = 0.f;
            const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)) & 0xffffu;
            t.x = w - __builtin_bit_cast(float, hb << 16);
            hi[i] = hb; lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)) & 0xffffu;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) { lgA[0][d] = hi[2 * d] | (hi[2 * d + 1] << 16); lgA[1][d] = lo[2 * d] | (lo[2 * d + 1] << 16); }
#pragma unroll
        // (rows >= n_class carry distinct values -row: lane half 1 runs softmax_argmax on rows 4..7, and equal values there sent the whole
        //  wave through its tie path -- expf and a division -- on every call)
        for (int q = 0; q < 16; ++q) { const int row = 8 * (q >> 2) + 4 * g + (q & 3); lgC[q] = row < a.lg_ncls ? a.lg_b[row] : -(float)row; }
    }
    auto epilogue = [&](int k) {
        int n, oy0, ox0;
        const bool valid = tile_coords(k, n, oy0, ox0);
        if constexpr (LS != 0) {
            const int rtile = (oy0 / R) * a.tiles_x + ox0 / WS_TW;
            // mode 1: groups = directions, per-frame outputs side by side (ls_*_dir count elements); mode 2: per window
            const size_t img = (size_t)n * tiles + rtile;
            // (wave-uniform bases: a buffer descriptor built from a per-lane pointer makes hipcc loop over the lanes; the lane's 16-byte slot is the vector offset)
            unsigned char *const cw = reinterpret_cast<unsigned char *>(a.ls_c_out) + (LS == 1 ? (size_t)grp * a.ls_c_dir * 4 : 0) + img * (R * 2048);
            [[maybe_unused]] unsigned char *const gw = reinterpret_cast<unsigned char *>(a.ls_gx) + (LS == 1 ? (size_t)grp * a.ls_gx_dir * 2 : 0) + img * (R * 4096);
            const unsigned lvo = (unsigned)lane * 16u;
            unsigned char *const hbase = outb + (LS == 1 ? (size_t)grp * a.ls_h_dir * 2 : 0) + (size_t)n * a.Ho * a.Wo * 32;
            const int himg_bytes = a.Ho * a.Wo * 32;
            const unsigned hvo = ox0 + pl < a.Wo ? (unsigned)(pl * 32 + g * 16) : OOB;
            const float fb = a.ls_forget_bias;
            unroll_steps<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int oy = oy0 + r;
                float hv[8], cn[8];
                [[maybe_unused]] unsigned gh[4][8];         // mode 1: the gates rounded to bf16 (halfwords), as stored: [gate][hidden channel of this lane half]
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    // (elements copied to scalars first: __builtin_bit_cast applied to an ext_vector element expression reads element 0, hipcc 7.2)
                    float gi = acc[0][r][m], gj = acc[0][r][8 + m], gf = acc[1][r][m], go = acc[1][r][8 + m];
                    float c = 0.f;
                    if constexpr (LS == 2) {
                        const unsigned wi = gxq[r][0][m >> 1], wj = gxq[r][1][m >> 1], wf = gxq[r][2][m >> 1], wo = gxq[r][3][m >> 1];
                        auto wide = [&](unsigned w) { return __builtin_bit_cast(float, (m & 1) ? (w & 0xffff0000u) : (w << 16)); };
                        gi += wide(wi); gj += wide(wj); gf += wide(wf); go += wide(wo);
                        const float cold = cq[r][m >> 2][m & 3];
                        c = cold;
                    } else if constexpr (LS == 3) {         // gates = W_x x + W_h h + b straight from the accumulator (fp32, nothing rounded)
                        const float cold = cq[r][m >> 2][m & 3];
                        c = cold;
                    } else if (a.ls_gx) {                   // the time steps add the ROUNDED gx: the x pass's own first step uses the same values
                        auto rnd = [&](float v) { f32x2 t; t.x = v; t.y = 0.f; return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)) & 0xffffu; };
                        gh[0][m] = rnd(gi); gh[1][m] = rnd(gj); gh[2][m] = rnd(gf); gh[3][m] = rnd(go);
                        gi = __builtin_bit_cast(float, gh[0][m] << 16); gj = __builtin_bit_cast(float, gh[1][m] << 16);
                        gf = __builtin_bit_cast(float, gh[2][m] << 16); go = __builtin_bit_cast(float, gh[3][m] << 16);
                    } else {                                // x pass of the un-hoisted form (LS 3 steps follow): no gx is kept, the first step uses the fp32 gates
                        gh[0][m] = gh[1][m] = gh[2][m] = gh[3][m] = 0u;
                    }
                    const float sgf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((gf + fb) * -1.44269504f));
                    const float sgi = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gi * -1.44269504f));
                    const float tj = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(gj * 2.88539008f) + 1.0f), 1.0f);
                    c = __builtin_fmaf(sgf, c, sgi * tj);
                    const float tc = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(c * 2.88539008f) + 1.0f), 1.0f);
                    cn[m] = c;
                    hv[m] = tc * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(go * -1.44269504f));
                }
                // stores through descriptors of range 0 for ghost tiles / rows below the map (no branch around a vector-memory instruction)
                const bool rowok = valid && oy < a.Ho;
                const __amdgpu_buffer_rsrc_t rc_ = __builtin_amdgcn_make_buffer_rsrc((void *)cw, 0, valid ? R * 2048 : 0, 0x00020000);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x4 v = {__builtin_bit_cast(unsigned, cn[4 * q]), __builtin_bit_cast(unsigned, cn[4 * q + 1]),
                                     __builtin_bit_cast(unsigned, cn[4 * q + 2]), __builtin_bit_cast(unsigned, cn[4 * q + 3])};
                    store_b128_sofs(v, rc_, lvo, (r * 2 + q) * 1024);   // an offset above 4095 is an SGPR, not an immediate
                }
                if constexpr (LS == 1) {
                    const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc((void *)gw, 0, (valid && a.ls_gx) ? R * 4096 : 0, 0x00020000);   // no gx asked for: range 0, stores dropped
#pragma unroll
                    for (int q = 0; q < 4; ++q) {           // piece q = gate q (i, j, f, o): halfwords m = 0..7
                        const u32x4 v = {gh[q][0] | (gh[q][1] << 16), gh[q][2] | (gh[q][3] << 16), gh[q][4] | (gh[q][5] << 16), gh[q][6] | (gh[q][7] << 16)};
                        store_b128_sofs(v, rg_, lvo, (r * 4 + q) * 1024);
                    }
                }
                u32x4 hvw;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    f32x2 t; t.x = hv[2 * d]; t.y = hv[2 * d + 1];
                    hvw[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
                }
                const __amdgpu_buffer_rsrc_t rh_ = __builtin_amdgcn_make_buffer_rsrc((void *)hbase, 0, rowok ? himg_bytes : 0, 0x00020000);
                store_b128_sofs(hvw, rh_, hvo, (oy * a.Wo + ox0) * 32);
            });
            return;
        }
        if constexpr (LG) {
            const int ox = ox0 + pl;
            const int npx = a.Ho * a.Wo;
            const __amdgpu_buffer_rsrc_t ro_pred = __builtin_amdgcn_make_buffer_rsrc((void *)(a.lg_pred + (size_t)n * npx), 0, a.lg_pred ? npx * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t ro_lg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.lg_logits + (size_t)n * npx * a.lg_ncls), 0, a.lg_logits ? npx * a.lg_ncls * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t ro_pr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.lg_prob + (size_t)n * npx * a.lg_ncls), 0, a.lg_prob ? npx * a.lg_ncls * 4 : 0, 0x00020000);
            unroll_steps<R>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int oy = oy0 + r;
                u32x4 bq;
#pragma unroll
                for (int d = 0; d < 4; ++d) {                      // k-slots 2 d, 2 d + 1 = accumulator registers 4 (d >> 1) + 2 (d & 1) + {0, 1}
                    const float e0 = acc[0][r][4 * (d >> 1) + 2 * (d & 1)], e1 = acc[0][r][4 * (d >> 1) + 2 * (d & 1) + 1];
                    f32x2 v2;
                    v2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), relu_lo));
                    v2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), relu_lo));
                    bq[d] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16x2));      // what a bf16 store would have held
                }
                f32x16 lg = mfma_bf16(lgA[1], bq, lgC);            // smaller term first
                lg = mfma_bf16(lgA[0], bq, lg);
                // Outputs through buffer stores whose lane offset is out of range for lanes that own no pixel and whose range is 0 for
                // an output that was not asked for: NO branch around a vector-memory instruction (with the stores inside an if, hipcc
                // made the next park wait for vmcnt(0) -- the three tiles of loads in flight drained at every tile, 6000 cycles per tile
                // for 1150 cycles of MFMA).
                const bool own = g == 0 && valid && oy < a.Ho && ox < a.Wo;
                const unsigned px = (unsigned)((oy * a.Wo + ox));
                auto finish = [&](auto nc) {
                    constexpr int NC = decltype(nc)::value;
                    float l[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) l[c] = lg[c];
                    // the probabilities only on the (uniform) path that stores them: handed to softmax_argmax through a pointer that
                    // may be null, the array lived in scratch memory and its reload sat in the vmcnt queue behind the prefetched tiles
                    if (a.lg_prob) {
                        float p[NC];
                        const int best = softmax_argmax<NC>(l, p);
                        __builtin_amdgcn_raw_buffer_store_b32((unsigned)best, ro_pred, own ? px * 4u : OOB, 0, 0);
#pragma unroll
                        for (int c = 0; c < NC; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p[c]), ro_pr, own ? (px * NC + c) * 4u : OOB, 0, 0);
                    } else {
                        const int best = softmax_argmax<NC>(l, nullptr);
                        __builtin_amdgcn_raw_buffer_store_b32((unsigned)best, ro_pred, own ? px * 4u : OOB, 0, 0);
                    }
                    if (a.lg_logits) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l[c]), ro_lg, own ? (px * NC + c) * 4u : OOB, 0, 0);
                    }
                };
                if (a.lg_ncls == 2) finish(std::integral_constant<int, 2>{});
                else if (a.lg_ncls == 3) finish(std::integral_constant<int, 3>{});
                else finish(std::integral_constant<int, 4>{});
            });
            return;
        }
        unsigned char *const obase = outb + (size_t)n * out_img_bytes;
        const bool colok = ox0 + pl < a.Wo;
        unsigned vo[CB][2];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 2; ++q) vo[cb][q] = colok ? svoff[cb][q] : OOB;
        unroll_steps<R>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int oy = oy0 + r;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (valid && oy < a.Ho) ? out_img_bytes : 0, 0x00020000);
            unroll_steps<CB>([&](auto cbc) {
                constexpr int cb = decltype(cbc)::value;
                unroll_steps<2>([&](auto pc) {
                    constexpr int jb = 2 * decltype(pc)::value;
                    int srow;
                    if constexpr (KIND == 0) srow = (oy * a.Wo + ox0) * 32;
                    else {
                        constexpr int ph = TM == 0 ? (cb == 0 ? 0 : 3) : TM == 1 ? (cb == 0 ? 1 : 2) : 2 * cb + (jb >> 1);
                        srow = ((2 * oy + (ph >> 1)) * OW + 2 * ox0) * 32;
                    }
                    u32x2 pk[2];
                    unroll_steps<2>([&](auto dc) {
                        constexpr int j = jb + decltype(dc)::value;
                        // (elements copied to scalars first: __builtin_bit_cast applied to an ext_vector element expression reads element 0, hipcc 7.2)
                        const float e0 = acc[cb][r][4 * j + 0], e1 = acc[cb][r][4 * j + 1], e2 = acc[cb][r][4 * j + 2], e3 = acc[cb][r][4 * j + 3];
                        f32x2 lo2, hi2;
                        lo2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), relu_lo));
                        lo2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), relu_lo));
                        hi2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e2), relu_lo));
                        hi2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e3), relu_lo));
                        pk[j - jb].x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2));
                        pk[j - jb].y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2));
                    });
                    // v_permlane32_swap a, b: a's upper 32 lanes <-> b's lower 32 lanes.  Afterwards (a, b) of a lower lane = (own group jb,
                    // partner's part of group jb), of an upper lane = (partner's part of group jb + 1, own): ascending channels either way.
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
                    store_b128_sofs(v, ro, vo[cb][jb / 2], srow);
                });
            });
        });
    };

    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    // ---- prologue: chunks 0, 1, 2 of the flattened (tile, chunk) sequence requested; chunk 0 parked ----
    int lk = 0;                                         // tile index (among this worker's) of the load cursor
    load_setup(0);
    // position q of the flattened sequence = tile q / NCH, chunk q % NCH.  Requests run 3 positions ahead of the compute cursor.
    auto request = [&](auto setc, auto posc) {          // posc: position modulo U (compile time); advances the cursor when a tile ends
        constexpr int POS = decltype(posc)::value;
        issue(setc, std::integral_constant<int, POS % NCH>{});
        if constexpr (POS % NCH == NCH - 1) { ++lk; load_setup(lk); }
    };
    request(S0, std::integral_constant<int, 0>{});
    request(S1, std::integral_constant<int, 1 % U>{});
    {
#pragma unroll
        for (int i = 0; i < NWP; ++i) reinterpret_cast<u32x4 *>(ws)[tid + i * (NW * 64)] = wq[i];
        if constexpr (NWR > 0) { if (tid < NWR) reinterpret_cast<u32x4 *>(ws)[tid + NWP * (NW * 64)] = wq[NWP]; }
    }
    __syncthreads();                                    // the only barrier of the kernel: weights resident
    if (my == 0) return;
#ifdef UKBB_DIAG
    if (stamps && lane == 0) { stamps[0] = st_entry; stamps[14] = st_rt0; }
#endif
    UKBB_WS_STAMP(1)
    park(S0);
    request(S0, std::integral_constant<int, 2 % U>{});
    readA(std::integral_constant<int, 0>{});
    unroll_steps<PD>([&](auto tc) { readB(tc); });
    UKBB_WS_STAMP(2)

    const int bodies = (my + TPB - 1) / TPB;
    auto body = [&](int bi) {
        unroll_steps<U>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;      // position modulo U of the compute cursor
            constexpr int SETN = (Q + 1) & 1;           // register set / stage of position Q + 1
            if constexpr (SETN == 0) { park(S0); request(S0, std::integral_constant<int, (Q + 3) % U>{}); }
            else                     { park(S1); request(S1, std::integral_constant<int, (Q + 3) % U>{}); }
            if constexpr (LS >= 2 && Q % NCH == 0) ls_prefetch(bi * TPB + Q / NCH);
            compute(qc);
            if constexpr (Q % NCH == NCH - 1) { epilogue(bi * TPB + Q / NCH); UKBB_WS_STAMP(3 + bi * TPB + Q / NCH) }
        });
    };
    // first iteration peeled: hipcc merges the wait-count state of the loop entry with that of the back edge, and with the prologue
    // in front of the loop the merged state made the first LDS park of every iteration wait for ALL but one outstanding
    // vector-memory operation (both register sets in flight and the tile's stores): s_waitcnt vmcnt(8) instead of vmcnt(30)
    body(0);
#pragma unroll 1
    for (int bi = 1; bi < bodies; ++bi) body(bi);
#ifdef UKBB_DIAG
    if (stamps && lane == 0) { stamps[13] = (unsigned long long)my; stamps[15] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weights THROUGH A RING (tilings 420-): the same machine for layers whose Cout group's filter does not fit LDS (K = 9 x 256: 147 KB
// for 32 output channels -- up3_0, conv4_1 of network_ao.py).  The activations keep their private per-wave rings and tiles; the packed
// weights of ONE 16-channel chunk (9 KB per Cout block) stream through a two-stage ring shared by the workgroup: every thread fetches
// its share of chunk c + 2 while chunk c is multiplied, and ONE barrier per chunk hands stage (c + 1) & 1 over (written after the
// barrier that follows the last read of its previous content, read after the next one).  The waves of a workgroup therefore run their
// tiles in lockstep per chunk (a wave without a tile left runs a ghost tile: out-of-range loads, dropped stores), and the chunk loop is a
// run-time loop unrolled by two (register-set / stage parity) -- the weights of any number of chunks stream through the same code.
template <int R, int CB, int NW>
__device__ __forceinline__ void wr_main(const ConvArgs &a, const int grp, const int walker, const int nwalk) {
    constexpr int KIND = 0, WS_IW = ws_iw(KIND), HR = ws_hr(KIND, R);
    constexpr int HP = HR * WS_IW, NLD = ws_nld(KIND, R), STAGE = ws_stage_bytes(KIND, R), PLANE = ws_plane_bytes(KIND, R);
    constexpr int WCH = CB * 9 * 1024;                  // bytes of one chunk of this group's packed weights: [cb][tap][lane][16]
    constexpr int NWL = (WCH / 16 + NW * 64 - 1) / (NW * 64);      // 16-byte pieces per thread and chunk
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const wring = lds;                   // [2][WCH]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const int g = lane >> 5, pl = lane & 31;
    unsigned char *const ring = lds + 2 * WCH + wave * (2 * STAGE);
    const int nch = (a.C0 + a.C1) / 16, nch0 = a.C0 / 16;

    const int tiles = a.tiles_x * a.tiles_y, ntiles = a.N * tiles;
    // Tile t of a round goes to worker t.  Default: wave-major numbering, consecutive tiles on consecutive walkers (= different XCDs).
    // xcd_local (large maps, set by the launcher): the walkers of one XCD (walker & 7 after ws_place) take a contiguous run of tiles and
    // the waves of a workgroup consecutive ones, so vertically adjacent row bands -- which share two halo rows -- are read through ONE
    // XCD's L2 at about the same time.  Measured at N = 100 (r04): 128x128 maps conv1_1 50.5 -> 46.8 us, up1_0 75.4 -> 66.4, up1_1
    // 51.1 -> 49.0; 64x64 and 32x32 maps 1-4 us SLOWER per layer (few tiles per image: the eight waves of a workgroup then sit on
    // one image's rows and queue on the same channels), hence the switch.
    const int per_xcd_ = nwalk / 8, chunk_ = (nwalk % 8 == 0) ? (walker & 7) * per_xcd_ + (walker >> 3) : walker;
    const int worker = a.xcd_local ? chunk_ * NW + wave : wave * nwalk + walker, nworkers = nwalk * NW;
    const int my = worker < ntiles ? (ntiles - worker + nworkers - 1) / nworkers : 0;
    // rounds of the whole workgroup = the largest tile count among its waves = that of wave 0, whose worker id is the smallest of the
    // workgroup in EITHER numbering (wave-major: walker; xcd_local: chunk_ * NW -- not walker: ADVICE r04, tiles were dropped when
    // ntiles % nworkers fell between the two)
    const int w0 = a.xcd_local ? chunk_ * NW : walker;
    const int rounds = w0 < ntiles ? (ntiles - w0 + nworkers - 1) / nworkers : 0;
    if (rounds == 0) return;                            // uniform over the workgroup

    // ---- staging geometry of the activations (as ws_main) ----
    unsigned geo[NLD], pbase[NLD], lofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = lane + 64 * i, gg = p & 1, px = p >> 1;
        const int hy = px / WS_IW, hx = px - hy * WS_IW;
        geo[i] = (unsigned)hy | ((unsigned)hx << 8) | (px < HP ? 0u : OOB);
        pbase[i] = (unsigned)((hy * a.W + hx) * 32 + 16 * gg);
        lofs[i] = (unsigned)((px < HP ? gg * PLANE + px * 16 : 2 * PLANE + (lane & 1) * 16));
    }
    const unsigned char *const in0 = reinterpret_cast<const unsigned char *>(a.in0);
    const unsigned char *const in1 = reinterpret_cast<const unsigned char *>(a.in1);
    const int plane_bytes = a.H * a.W * 32;
    const int nb0 = a.C0 / 16, nb1 = a.C1 / 16;
    auto tile_coords = [&](int k, int &n, int &oy0, int &ox0) -> bool {
        const bool valid = k < my;
        const int t = valid ? worker + k * nworkers : 0;
        n = t / tiles;
        const int r = t - n * tiles, ty = r / a.tiles_x;
        oy0 = ty * R; ox0 = (r - ty * a.tiles_x) * WS_TW;
        return valid;
    };
    unsigned voff[NLD];
    __amdgpu_buffer_rsrc_t rs0, rs1;
    auto load_setup = [&](int k) {
        int n, oy0, ox0;
        const bool valid = tile_coords(k, n, oy0, ox0);
        rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)(in0 + (size_t)n * nb0 * plane_bytes), 0, nb0 * plane_bytes, 0x00020000);
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)((nb1 ? in1 : in0) + (size_t)n * (nb1 ? nb1 : nb0) * plane_bytes), 0, (nb1 ? nb1 : nb0) * plane_bytes, 0x00020000);
        const int toff = ((oy0 - 1) * a.W + (ox0 - 1)) * 32;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int hy = (int)(geo[i] & 0xffu), hx = (int)((geo[i] >> 8) & 0xffu);
            const bool ok = valid && !(geo[i] & OOB) && (unsigned)(oy0 - 1 + hy) < (unsigned)a.H && (unsigned)(ox0 - 1 + hx) < (unsigned)a.W;
            voff[i] = ok ? pbase[i] + (unsigned)toff : OOB;
        }
    };
    // weights of this group: [chunk][cb][tap][lane][16] contiguous per chunk
    const unsigned char *const wsrc = reinterpret_cast<const unsigned char *>(a.wpk) + (size_t)grp * nch * WCH;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wsrc, 0, nch * WCH, 0x00020000);
    unsigned wvo[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) wvo[i] = (tid + i * (NW * 64)) * 16 < WCH ? (unsigned)((tid + i * (NW * 64)) * 16) : OOB;

    u32x4 xq[2][NLD], wq[2][NWL];
    int lk = 0, lch = 0;                                // load cursor: tile (among this wave's), chunk
    auto request = [&](auto setc) {                     // activations of (tile lk, chunk lch) and this thread's share of the weights of chunk lch
        constexpr int SET = decltype(setc)::value;
        const bool second = lch >= nch0;
        const int soff = (second ? lch - nch0 : lch) * plane_bytes;
#pragma unroll
        for (int i = 0; i < NLD; ++i) xq[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(second ? rs1 : rs0, voff[i], soff, 0);
#pragma unroll
        for (int i = 0; i < NWL; ++i) wq[SET][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[i], lch * WCH, 0);
        if (++lch == nch) { lch = 0; ++lk; load_setup(lk); }
    };
    auto park = [&](auto setc) {                        // register set SET -> activation stage SET and weight stage SET
        constexpr int SET = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4 *>(ring + SET * STAGE + lofs[i]) = xq[SET][i];
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if ((tid + i * (NW * 64)) * 16 < WCH) *reinterpret_cast<u32x4 *>(wring + SET * WCH + (tid + i * (NW * 64)) * 16) = wq[SET][i];
    };

    // ---- compute (fragment reads pipelined by hand as in ws_main; a chunk = S steps, two chunks per unrolled body) ----
    constexpr int NGRP = 3, NAF = 3, S = NGRP * HR, PD = 3, NB = 4, U = 2;
    static_assert((U * S) % NB == 0, "fragment buffers rotate consistently across loop iterations");
    f32x16 acc[CB][R], biasv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + (grp * CB + cb) * 32 + 8 * j + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) biasv[cb][4 * j + i] = b4[i];
        }
    const unsigned char *const xs_lane = ring + g * PLANE + pl * 16;
    const unsigned char *const ws_lane = wring + lane * 16;
    u32x4 Bq[NB], Aq[2][CB][NAF];
    auto readB = [&](auto tc) {
        constexpr int TT = decltype(tc)::value, T = TT % (U * S), Qn = T / S, s = T % S, kw = s / HR, rp = s % HR;
        Bq[TT % NB] = *reinterpret_cast<const u32x4 *>(xs_lane + (Qn & 1) * STAGE + (rp * WS_IW + kw) * 16);
    };
    auto readA = [&](auto gc) {                         // (chunk parity, kw) group
        constexpr int GG = decltype(gc)::value, G = GG % (U * NGRP), Qn = G / NGRP, kw = G % NGRP;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                Aq[GG & 1][cb][kh] = *reinterpret_cast<const u32x4 *>(ws_lane + (Qn & 1) * WCH + (cb * 9 + kh * 3 + kw) * 1024);
    };
    auto compute = [&](auto qc) {                       // chunk of parity Q: activation stage Q, weight stage Q
        constexpr int Q = decltype(qc)::value;
        unroll_steps<S>([&](auto sc) {
            constexpr int s = decltype(sc)::value, T = Q * S + s, kw = s / HR, rp = s % HR, G = Q * NGRP + kw;
            // the reads of the NEXT chunk's first steps / first A group may only go out once its stages are visible: they are issued by
            // the caller behind the barrier (T + PD >= S of the last chunk before a barrier is handled by splitting the prefetch there)
            if constexpr (s + PD < S) readB(std::integral_constant<int, T + PD>{});
            if constexpr (rp == 0 && kw + 1 < NGRP) readA(std::integral_constant<int, G + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            unroll_steps<NAF>([&](auto khc) {
                constexpr int kh = decltype(khc)::value, r = rp - kh;
                if constexpr (r >= 0 && r < R) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) acc[cb][r] = mfma_bf16(Aq[G & 1][cb][kh], Bq[T % NB], acc[cb][r]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto prime = [&](auto qc) {                         // first reads of chunk parity Q (behind the barrier that made its stages visible)
        constexpr int Q = decltype(qc)::value;
        readA(std::integral_constant<int, Q * NGRP>{});
        unroll_steps<PD>([&](auto tc) { readB(std::integral_constant<int, Q * S + decltype(tc)::value>{}); });
    };

    // ---- epilogue (as ws_main KIND 0: 16-byte stores through v_permlane32_swap) ----
    unsigned char *const outb = reinterpret_cast<unsigned char *>(a.out);
    const int cst = a.cout_store > 0 ? a.cout_store : a.Cout;
    const int out_plane_bytes = a.Ho * a.Wo * 32, out_img_bytes = (cst / 16) * out_plane_bytes;
    unsigned svoff[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int chn = (grp * CB + cb) * 32 + 8 * (2 * q + g);
            svoff[cb][q] = chn < cst ? (unsigned)((chn >> 4) * out_plane_bytes + pl * 32 + (chn & 15) * 2) : OOB;
        }
    const int relu_lo = a.relu ? 0 : (int)0x80000000;
    auto epilogue = [&](int k) {
        int n, oy0, ox0;
        const bool valid = tile_coords(k, n, oy0, ox0);
        unsigned char *const obase = outb + (size_t)n * out_img_bytes;
        const bool colok = ox0 + pl < a.Wo;
        unroll_steps<R>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const int oy = oy0 + r;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, (valid && oy < a.Ho) ? out_img_bytes : 0, 0x00020000);
            const int srow = (oy * a.Wo + ox0) * 32;
            unroll_steps<CB>([&](auto cbc) {
                constexpr int cb = decltype(cbc)::value;
                unroll_steps<2>([&](auto pc) {
                    constexpr int jb = 2 * decltype(pc)::value;
                    u32x2 pk[2];
                    unroll_steps<2>([&](auto dc) {
                        constexpr int j = jb + decltype(dc)::value;
                        const float e0 = acc[cb][r][4 * j + 0], e1 = acc[cb][r][4 * j + 1], e2 = acc[cb][r][4 * j + 2], e3 = acc[cb][r][4 * j + 3];
                        f32x2 lo2, hi2;
                        lo2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e0), relu_lo));
                        lo2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e1), relu_lo));
                        hi2.x = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e2), relu_lo));
                        hi2.y = __builtin_bit_cast(float, max(__builtin_bit_cast(int, e3), relu_lo));
                        pk[j - jb].x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, bf16x2));
                        pk[j - jb].y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, bf16x2));
                    });
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
                    store_b128_sofs(v, ro, colok ? svoff[cb][jb / 2] : OOB, srow);
                });
            });
        });
    };

    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    // ---- flattened (round, chunk) sequence, position q: stage / register set q & 1.  Iteration q: barrier; park q + 1; request q + 3;
    //      compute q.  nch is even for every layer this kernel serves (checked by the launcher), so a tile starts at an even position.
    load_setup(0);
    request(S0);                                        // q = 0
    request(S1);                                        // q = 1
    park(S0);
    request(S0);                                        // q = 2
    const int total = rounds * nch;
    int ck = 0, cch = 0;                                // compute cursor
    auto step = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;          // parity of the position being computed
        __syncthreads();                                // stage Q (parked an iteration ago) visible; stage Q ^ 1 free (its chunk was multiplied an iteration ago)
        if constexpr (Q == 0) { park(S1); request(S1); } else { park(S0); request(S0); }
        prime(qc);
        if (cch == 0) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[cb][r] = biasv[cb];
        }
        compute(qc);
        if (++cch == nch) { cch = 0; epilogue(ck); ++ck; }
    };
#pragma unroll 1
    for (int q = 0; q < total; q += 2) { step(S0); step(S1); }
}

// ---- workgroup -> (Cout group, walker).  Workgroups b and b + 8 share an XCD (observed round-robin placement, speed only):
//      the nG workgroups that walk the same tiles with different Cout groups are given equal b % 8 so that a tile is fetched
//      into ONE per-XCD L2.
__device__ __forceinline__ void ws_place(int nG, int &grp, int &walker, int &nwalk) {
    nwalk = (int)gridDim.x / nG;                        // the launcher makes the grid a multiple of nG
    if ((int)gridDim.x % (8 * nG) == 0) {
        const int j = (int)blockIdx.x >> 3;
        grp = j % nG; walker = ((int)blockIdx.x & 7) + 8 * (j / nG);
    } else {
        grp = (int)blockIdx.x % nG; walker = (int)blockIdx.x / nG;
    }
}

// R: output rows per tile; CB: 32-channel Cout blocks per workgroup (and per wave); NW: waves (= independent workers) per workgroup;
// NCH: 16-channel chunks of the input (both sources together); TWO: chunks NCH/2.. come from the second source (C0 == C1)
template <int R, int CB, int NW, int NCH, bool TWO, bool LG = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_ws_kernel(const ConvArgs a) {
    int grp, walker, nwalk;
    ws_place(a.Cout / (32 * CB), grp, walker, nwalk);
    ws_main<0, R, CB, NW, NCH, TWO, 0, LG>(a, grp, walker, nwalk);
}

template <int R, int CB, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_wr_kernel(const ConvArgs a) {
    int grp, walker, nwalk;
    ws_place(a.Cout / (32 * CB), grp, walker, nwalk);
    wr_main<R, CB, NW>(a, grp, walker, nwalk);
}

// transposed conv: R input rows per tile; C16: Cout = 16 (a block holds two phases), else the block pairing alternates with the group
template <int R, int NW, int NCH, bool C16>
__global__ __launch_bounds__(NW * 64, NW / 4) void tconv_ws_kernel(const ConvArgs a) {
    int grp, walker, nwalk;
    ws_place(a.Cout / 64, grp, walker, nwalk);
    if constexpr (C16) ws_main<1, R, 2, NW, NCH, false, 2>(a, grp, walker, nwalk);
    else if (grp & 1) ws_main<1, R, 2, NW, NCH, false, 1>(a, grp, walker, nwalk);       // workgroup-uniform branch
    else ws_main<1, R, 2, NW, NCH, false, 0>(a, grp, walker, nwalk);
}

// ConvLSTM forms (ConvArgs::ls_mode): 16 channels in, 64 gate channels per group (mode 1: two groups = the two directions), R = 2 rows x 32 pixels per tile
template <int R, int NW, int LS>
__global__ __launch_bounds__(NW * 64, NW / 4) void lstm_ws_kernel(const ConvArgs a) {
    int grp, walker, nwalk;
    ws_place(a.Cout / 64, grp, walker, nwalk);
    if constexpr (LS == 3) ws_main<0, R, 2, NW, 2, true, 0, false, 3>(a, grp, walker, nwalk);
    else ws_main<0, R, 2, NW, 1, false, 0, false, LS>(a, grp, walker, nwalk);
}

}  // namespace

// W(id, R, CB, NW): ConvConfig::pc == 6, 3x3 stride 1, mb 32, th = R, tw = 32, kc 16, wm 1, wn = NW, cb = CB
#define UKBB_WS_CONFIGS(W)          \
    W(400, 2, 1, 8)                 \
    W(401, 4, 1, 4)                 \
    W(402, 2, 2, 8)                 \
    W(403, 4, 2, 4)
// L(id, R, NW): 16 -> 16 channels with the logits conv + softmax / argmax in the epilogue (ConvConfig::fuse == 2), cb 1
#define UKBB_WSL_CONFIGS(L)         \
    L(404, 4, 4)                    \
    L(405, 2, 8)                    \
    L(406, 4, 8)
// S(id, R, CB, NW): weights through a ring (any even number of chunks): ConvConfig::kc = 32 marks them (two chunks per loop body)
#define UKBB_WR_CONFIGS(S)          \
    S(420, 4, 2, 4)                 \
    S(421, 4, 1, 4)                 \
    S(422, 2, 2, 8)                 \
    S(423, 2, 1, 8)
// T(id, R, NW): the transposed conv as 2x2 sub-pixel conv (ks 2), th = R input rows, two paired 32-row blocks per workgroup (cb 2)
#define UKBB_WST_CONFIGS(T)         \
    T(410, 4, 4)                    \
    T(411, 2, 8)                    \
    T(412, 2, 4)

#define UKBB_WS_ENTRY(ID, R, CB, NW) \
    {ID, 3, 1, 32, R, WS_TW, 16, 1, NW, CB, 0, 6, "convBF16ws_3x3s1_r" #R "x32_cb" #CB "_w" #NW, 0},
#define UKBB_WST_ENTRY(ID, R, NW) \
    {ID, 2, 1, 32, R, WS_TW, 16, 1, NW, 2, 0, 6, "tconvBF16ws_2x2_r" #R "x32_cb2_w" #NW, 0},
#define UKBB_WSL_ENTRY(ID, R, NW) \
    {ID, 3, 1, 32, R, WS_TW, 16, 1, NW, 1, 0, 6, "convBF16ws_3x3s1+logits_r" #R "x32_cb1_w" #NW, 2},
#define UKBB_WR_ENTRY(ID, R, CB, NW) \
    {ID, 3, 1, 32, R, WS_TW, 32, 1, NW, CB, 0, 6, "convBF16wr_3x3s1_r" #R "x32_cb" #CB "_w" #NW, 0},
static const ConvConfig g_ws_cfgs[] = {UKBB_WS_CONFIGS(UKBB_WS_ENTRY) UKBB_WSL_CONFIGS(UKBB_WSL_ENTRY) UKBB_WST_CONFIGS(UKBB_WST_ENTRY) UKBB_WR_CONFIGS(UKBB_WR_ENTRY)};

// Order of the 4 x cout phase-major virtual channels (phase = 2 py + px, then channel) in the packed filter / bias of the transposed
// conv tilings: dst column v takes source column wst_pack_order(cout, v).  cout = 16: unchanged (blocks {00 + 01}, {10 + 11}).
// cout % 32 == 0: workgroup (group) 2 i holds {phase 00, phase 11} of channels 32 i .., group 2 i + 1 holds {01, 10} (tconv_ws_kernel).
int wst_pack_order(int cout, int v) {
    if (cout == 16) return v;
    const int nb = v / 32, r = v % 32, grp = nb / 2, k = nb % 2, i = grp / 2;
    const int ph = (grp & 1) ? (k == 0 ? 1 : 2) : (k == 0 ? 0 : 3);
    return ph * cout + i * 32 + r;
}

int num_ws_configs() { return (int)(sizeof(g_ws_cfgs) / sizeof(g_ws_cfgs[0])); }
const ConvConfig &ws_config(int i) { return g_ws_cfgs[i]; }
int ws_lds_bytes_for(const ConvConfig &c, int cin) {
    if (c.kc == 32) return 2 * c.cb * 9 * 1024 + c.wn * 2 * ws_stage_bytes(0, c.th);      // ring tilings: two weight stages, whatever the K
    return ws_lds_bytes(c.ks == 2 ? 1 : 0, c.th, c.cb, c.wn, cin / 16);
}

namespace {
template <class K>
hipError_t launch_ws_kernel(K k, int bytes, int threads, const ConvArgs &a, int grid, hipStream_t s, OncePerDevice &lds_ok) {
    hipError_t e = allow_dynamic_lds(lds_ok, reinterpret_cast<const void *>(k), bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), bytes, s, a);
    return hipGetLastError();
}
template <int R, int CB, int NW, int NCH, bool TWO>
hipError_t launch_ws_one(const ConvArgs &a, int grid, hipStream_t s) {
    constexpr int bytes = ws_lds_bytes(0, R, CB, NW, NCH);
    if constexpr (bytes > 160 * 1024) { return hipErrorInvalidValue; }
    else {
        static OncePerDevice lds_ok;
        return launch_ws_kernel(conv_ws_kernel<R, CB, NW, NCH, TWO>, bytes, NW * 64, a, grid, s, lds_ok);
    }
}
template <int R, int CB, int NW>
hipError_t launch_ws_cfg(const ConvArgs &a, int grid, hipStream_t s) {
    const int nch = (a.C0 + a.C1) / 16;
    const bool two = a.C1 > 0;
    switch (nch) {
        case 1: return two ? hipErrorInvalidValue : launch_ws_one<R, CB, NW, 1, false>(a, grid, s);
        case 2: return two ? launch_ws_one<R, CB, NW, 2, true>(a, grid, s) : launch_ws_one<R, CB, NW, 2, false>(a, grid, s);
        case 4: return two ? launch_ws_one<R, CB, NW, 4, true>(a, grid, s) : launch_ws_one<R, CB, NW, 4, false>(a, grid, s);
        case 8: return two ? launch_ws_one<R, CB, NW, 8, true>(a, grid, s) : launch_ws_one<R, CB, NW, 8, false>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}
template <int R, int CB, int NW>
hipError_t launch_wr_cfg(const ConvArgs &a, int grid, hipStream_t s) {
    const int nch = (a.C0 + a.C1) / 16;
    if (nch < 2 || (nch & 1) || (a.C1 && (a.C0 / 16) % 2)) return hipErrorInvalidValue;   // two chunks per loop body; a source switch at an even chunk
    constexpr int bytes = 2 * CB * 9 * 1024 + NW * 2 * ws_stage_bytes(0, R);
    static_assert(bytes <= 160 * 1024, "LDS");
    static OncePerDevice lds_ok;
    return launch_ws_kernel(conv_wr_kernel<R, CB, NW>, bytes, NW * 64, a, grid, s, lds_ok);
}
template <int R, int NW>
hipError_t launch_wsl_cfg(const ConvArgs &a, int grid, hipStream_t s) {
    if (a.C0 != 16 || a.C1 || a.Cout != 32 || a.cout_store != 16 || !a.lg_w || !a.lg_b || a.lg_ncls < 2 || a.lg_ncls > 4) return hipErrorInvalidValue;
    constexpr int bytes = ws_lds_bytes(0, R, 1, NW, 1);
    static OncePerDevice lds_ok;
    return launch_ws_kernel(conv_ws_kernel<R, 1, NW, 1, false, true>, bytes, NW * 64, a, grid, s, lds_ok);
}
template <int R, int NW, int NCH, bool C16>
hipError_t launch_wst_one(const ConvArgs &a, int grid, hipStream_t s) {
    constexpr int bytes = ws_lds_bytes(1, R, 2, NW, NCH);
    if constexpr (bytes > 160 * 1024) { return hipErrorInvalidValue; }
    else {
        static OncePerDevice lds_ok;
        return launch_ws_kernel(tconv_ws_kernel<R, NW, NCH, C16>, bytes, NW * 64, a, grid, s, lds_ok);
    }
}
template <int R, int NW>
hipError_t launch_wst_cfg(const ConvArgs &a, int grid, hipStream_t s) {
    const int nch = a.C0 / 16;
    if (a.up2 == 16) return nch == 2 ? launch_wst_one<R, NW, 2, true>(a, grid, s) : hipErrorInvalidValue;
    switch (nch) {
        case 4: return launch_wst_one<R, NW, 4, false>(a, grid, s);
        case 8: return launch_wst_one<R, NW, 8, false>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}
}  // namespace

// ---- bf16 ConvLSTM (UKBB_PREC_BF16 on a UNet-LSTM handle) ----
namespace { constexpr int LSW_R = 2, LSW_NW = 4; }
size_t lstm_ws_tiles(int H, int W) { return (size_t)((H + LSW_R - 1) / LSW_R) * ((W + WS_TW - 1) / WS_TW); }
size_t lstm_ws_gx_elems(int H, int W) { return lstm_ws_tiles(H, W) * LSW_R * 2048; }      // bf16 values per image: 64 lanes x 32 per tile row
size_t lstm_ws_c_floats(int H, int W) { return lstm_ws_tiles(H, W) * LSW_R * 512; }       // fp32 values per image: 64 lanes x 8 per tile row

hipError_t launch_lstm_ws(const ConvArgs &a_in, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.ls_mode < 1 || a.ls_mode > 3 || !a.ls_bf16 || a.C0 != 16 || a.up2 || a.H != a.Ho || a.W != a.Wo || a.pad_y != 1 || a.pad_x != 1) return hipErrorInvalidValue;
    if (a.ls_mode == 3 ? (a.C1 != 16 || !a.in1) : (a.C1 || a.in1)) return hipErrorInvalidValue;
    // mode 1: gx optional (nullptr = the un-hoisted form follows, nothing is kept); mode 2: gx + its frame map; mode 3: the frame map (of in0) and the bias
    if (a.ls_mode == 1 ? (a.Cout != 64 && a.Cout != 128) || a.in0_map || !a.bias || !a.ls_c_out
        : a.ls_mode == 2 ? a.Cout != 64 || !a.ls_gx || !a.ls_gx_map || !a.ls_c_in || !a.ls_c_out
                         : a.Cout != 64 || !a.bias || !a.ls_gx_map || !a.ls_c_in || !a.ls_c_out)
        return hipErrorInvalidValue;
    if ((long long)a.H * a.W * 32 >= 0x7fffffffll) return hipErrorInvalidValue;
    a.tiles_y = (a.Ho + LSW_R - 1) / LSW_R; a.tiles_x = (a.Wo + WS_TW - 1) / WS_TW;
    {   // tile order as launch_conv_ws
        static const char *const force = getenv("UKBB_WS_XCD_LOCAL");
        a.xcd_local = force ? atoi(force) : ((long long)a.H * a.W >= 128ll * 128);
    }
    const int nG = a.Cout / 64;
    const long long ntiles = (long long)a.N * a.tiles_y * a.tiles_x;
    const int cus = device_cu_count();
    long long want = ((ntiles + LSW_NW - 1) / LSW_NW) * nG;
    int grid = cus >= 8 * nG ? cus / (8 * nG) * (8 * nG) : cus / nG * nG;
    if (grid < nG) grid = nG;
    if (want < grid) grid = (int)((want + nG - 1) / nG * nG);
    constexpr int bytes = ws_lds_bytes(0, LSW_R, 2, LSW_NW, 1), bytes3 = ws_lds_bytes(0, LSW_R, 2, LSW_NW, 2);
    static_assert(bytes <= 160 * 1024 && 2 * bytes3 <= 160 * 1024, "LDS (two workgroups per CU)");
    static OncePerDevice ok1, ok2, ok3;
    if (a.ls_mode == 1) return launch_ws_kernel(lstm_ws_kernel<LSW_R, LSW_NW, 1>, bytes, LSW_NW * 64, a, grid, s, ok1);
    if (a.ls_mode == 3) return launch_ws_kernel(lstm_ws_kernel<LSW_R, LSW_NW, 3>, bytes3, LSW_NW * 64, a, grid, s, ok3);
    return launch_ws_kernel(lstm_ws_kernel<LSW_R, LSW_NW, 2>, bytes, LSW_NW * 64, a, grid, s, ok2);
}

size_t pack_lstm_gate_weights_bf16_xh(const float *w, const float *bias, float *dst, float *bias_perm) {
    // both 16-channel halves of the gate kernel [3][3][16 + 16][64] as the two chunks of ONE filter (chunk 0 = x rows, chunk 1 = h rows),
    // output channels permuted as in pack_lstm_gate_weights_bf16: the resident filter of ws_main LS 3
    std::vector<float> tmp((size_t)9 * 32 * 64);
    for (int P = 0; P < 64; ++P) {
        const int cb = P / 32, row = P % 32, j = row / 8, g = (row / 4) & 1, i = row % 4;
        const int orig = (2 * cb + (j >> 1)) * 16 + 8 * g + 4 * (j & 1) + i;
        if (bias_perm) bias_perm[P] = bias ? bias[orig] : 0.f;
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < 32; ++ci) tmp[((size_t)t * 32 + ci) * 64 + P] = w[((size_t)t * 32 + ci) * 64 + orig];
    }
    return pack_conv_weights_bf16(tmp.data(), 3, 32, 64, 2, dst);
}

size_t pack_lstm_gate_weights_bf16(const float *w, int cin_total, int c_first, const float *bias, float *dst, float *bias_perm) {
    // packed channel P = 32 cb + 8 j + 4 g + i (MFMA row 8 j + 4 g + i of block cb)  <-  gate e = 2 cb + (j >> 1) of hidden channel 8 g + 4 (j & 1) + i:
    // the accumulator registers of lane half g are then (i | j) in block 0 and (f | o) in block 1, eight hidden channels each (ws_main, LS)
    std::vector<float> tmp((size_t)9 * 16 * 64);
    for (int P = 0; P < 64; ++P) {
        const int cb = P / 32, row = P % 32, j = row / 8, g = (row / 4) & 1, i = row % 4;
        const int orig = (2 * cb + (j >> 1)) * 16 + 8 * g + 4 * (j & 1) + i;
        if (bias_perm) bias_perm[P] = bias ? bias[orig] : 0.f;
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < 16; ++ci) tmp[((size_t)t * 16 + ci) * 64 + P] = w[((size_t)t * cin_total + c_first + ci) * 64 + orig];
    }
    return pack_conv_weights_bf16(tmp.data(), 3, 16, 64, 2, dst);
}

hipError_t launch_conv_ws(int cfg_id, const ConvArgs &a_in, hipStream_t s) {
    const ConvConfig *c = nullptr;
    for (const ConvConfig &k : g_ws_cfgs) if (k.id == cfg_id) c = &k;
    if (!c) return hipErrorInvalidValue;
    ConvArgs a = a_in;
    const bool tconv = c->ks == 2;
    if (a.in0_map || a.first_w || (a.lg_w != nullptr) != (c->fuse == 2)) return hipErrorInvalidValue;
    const bool ring = c->kc == 32;
    if (tconv) {
        // virtual channels = 4 phases x up2 real ones, packed in the paired block order of wst_pack_order(); one source
        if (a.up2 < 16 || a.up2 % 16 || (a.up2 > 16 && a.up2 % 32) || a.Cout != 4 * a.up2 || a.C1 || a.in1) return hipErrorInvalidValue;
    } else {
        if (a.up2) return hipErrorInvalidValue;
        if (a.C1 && a.C1 != a.C0 && !ring) return hipErrorInvalidValue;     // the K loop switches source at the half
    }
    if (a.C0 % 16 || a.Cout % (32 * c->cb) || a.pad_y != 1 || a.pad_x != 1 || a.Ho != a.H || a.Wo != a.W) return hipErrorInvalidValue;
    const long long out_ch = tconv ? 4ll * a.up2 : a.Cout;         // bytes of the largest map of one image must fit a buffer range
    if ((long long)a.H * a.W * (a.C0 > out_ch ? a.C0 : out_ch) * 2 >= 0x7fffffffll) return hipErrorInvalidValue;
    a.tiles_y = (a.Ho + c->th - 1) / c->th; a.tiles_x = (a.Wo + WS_TW - 1) / WS_TW;
    {   // tile order, see ws_main; UKBB_WS_XCD_LOCAL=0 / 1 forces it (A/B)
        static const char *const force = getenv("UKBB_WS_XCD_LOCAL");
        a.xcd_local = force ? atoi(force) : ((long long)a.H * a.W >= 128ll * 128);
    }
    const int nG = a.Cout / (32 * c->cb);
    const long long ntiles = (long long)a.N * a.tiles_y * a.tiles_x;
    // one workgroup per CU; a multiple of 8 nG where the chip allows it (XCD-aware walker mapping), never more walkers than tiles need
    int cus = device_cu_count();
    { static const int per_cu = getenv("UKBB_WS_WGS_PER_CU") ? atoi(getenv("UKBB_WS_WGS_PER_CU")) : 1; if (per_cu > 1) cus *= per_cu; }   // A/B knob
    long long want = ((ntiles + c->wn - 1) / c->wn) * nG;           // workgroups that would give every wave one tile
    int grid = cus >= 8 * nG ? cus / (8 * nG) * (8 * nG) : cus / nG * nG;
    if (grid < nG) grid = nG;
    if (want < grid) grid = (int)((want + nG - 1) / nG * nG);
#ifdef UKBB_DIAG
    // UKBB_WS_STAMPS=<cfg id>: the 6th launch of that tiling reports where its waves spent their cycles
    static unsigned long long *d_st = nullptr;
    const char *sc = getenv("UKBB_WS_STAMPS");
    const bool stamp = sc && atoi(sc) == cfg_id;
    const int nwv = grid * c->wn;
    if (stamp) {
        if (!d_st && hipMalloc(reinterpret_cast<void **>(&d_st), 4096 * 16 * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_st, 0, 4096 * 16 * 8, s);
        a.first_w = reinterpret_cast<const float *>(d_st); a.diag = 1;
    }
    struct Dump {
        bool on; hipStream_t s; unsigned long long *d; int cfg, nwv;
        ~Dump() {
            if (!on) return;
            static int shots = 0;
            if (++shots != 6) return;
            std::vector<unsigned long long> h((size_t)nwv * 16);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0;
            double wgt = 0, pro = 0, life = 0, tile[10] = {0}; int ntile[10] = {0}, n = 0; double clk = 0;
            for (int w = 0; w < nwv; ++w) {
                const unsigned long long *q = &h[(size_t)w * 16];
                if (!q[0]) continue;
                ++n;
                const int my = (int)q[13];
                unsigned long long end = q[2];
                for (int k = 0; k < my && k < 10; ++k) { tile[k] += (double)(q[3 + k] - (k ? q[2 + k] : q[2])); ++ntile[k]; end = q[3 + k]; }
                t0 = std::min(t0, q[0]); t1 = std::max(t1, end);
                wgt += (double)(q[1] - q[0]); pro += (double)(q[2] - q[1]); life += (double)(end - q[0]);
                if (q[15] > q[14]) clk += (double)(end - q[0]) / (double)(q[15] - q[14]) * 0.1;
            }
            if (!n) return;
            fprintf(stderr, "[ws stamps cfg %d] %d working waves: weights resident after %.0f cyc, prologue %.0f, mean lifetime %.0f cyc, first entry -> last end %.0f cyc, clock %.2f GHz\n",
                    cfg, n, wgt / n, pro / n, life / n, (double)(t1 - t0), clk / n);
            for (int k = 0; k < 10; ++k) if (ntile[k]) fprintf(stderr, "[ws stamps cfg %d]   tile %d: %d waves, mean %.0f cyc\n", cfg, k, ntile[k], tile[k] / ntile[k]);
        }
    } dump{stamp, s, d_st, cfg_id, nwv > 4096 ? 4096 : nwv};
#endif
    switch (cfg_id) {
#define UKBB_WS_CASE(ID, R, CB, NW) case ID: return launch_ws_cfg<R, CB, NW>(a, grid, s);
        UKBB_WS_CONFIGS(UKBB_WS_CASE)
#define UKBB_WR_CASE(ID, R, CB, NW) case ID: return launch_wr_cfg<R, CB, NW>(a, grid, s);
        UKBB_WR_CONFIGS(UKBB_WR_CASE)
#define UKBB_WSL_CASE(ID, R, NW) case ID: return launch_wsl_cfg<R, NW>(a, grid, s);
        UKBB_WSL_CONFIGS(UKBB_WSL_CASE)
#define UKBB_WST_CASE(ID, R, NW) case ID: return launch_wst_cfg<R, NW>(a, grid, s);
        UKBB_WST_CONFIGS(UKBB_WST_CASE)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ukbb
