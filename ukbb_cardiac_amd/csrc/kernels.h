// Internal interface between the engine (engine.cpp) and the gfx950 kernels.
// Not part of the public ABI (that is include/ukbb_fcn.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_state.h"

namespace ukbb {

// ---- per-device launch state (device_state.h) on top of the HIP runtime ------------------------------------------------
inline int current_device() { int d = 0; return hipGetDevice(&d) == hipSuccess ? d : -1; }
// compute units of the CURRENT device (sizes the persistent grids); 256 if the query fails
inline int device_cu_count() {
    static PerDeviceInt cu;
    const int d = current_device();
    return cu.get(d, [d] { hipDeviceProp_t p; return d >= 0 && hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 0; }, 256);
}
// grant `kernel` `bytes` of dynamic LDS on the CURRENT device, once per device (`once`: one static per kernel instantiation)
inline hipError_t allow_dynamic_lds(OncePerDevice &once, const void *kernel, int bytes) {
    return (hipError_t)once.run(current_device(), [&] {
        return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
}

// prob / pred of train_network.py:198-199 (network_ao.py:159-160): prob = softmax(logits), pred = argmax(prob) -- the argmax
// is taken over the float32 PROBABILITIES, lowest index on ties.  It equals argmax(logits) unless another class's logit lies
// within a few ulps of 1.0 (in exp's argument) of the maximum: exp(l - m) then rounds to 1, or the products with 1/sum
// round to the same float, and the lower index wins although its logit is smaller.  Only in that case (rare: |logits| of the
// top two below ~1 and almost equal) the probabilities are formed to decide; the common path costs a subtract and a compare
// per class.  ``p`` (optional) receives the probabilities, formed the same way, so argmax(p) == return value always.
// (r04: callers pass either a real array or the literal nullptr.  Handed `cond ? array : nullptr`, hipcc kept the array in SCRATCH
// memory -- 15-36 scratch instructions in the epilogues of the head / fused-logits kernels, whose reloads sit in the same vmcnt queue
// as the prefetched tiles.  softmax_argmax_opt() below is the call shape for an optional output.)
template <int NCLS>
__device__ __forceinline__ int softmax_argmax(const float (&lg)[NCLS], float *p) {
    int best = 0; float m = lg[0];
#pragma unroll
    for (int c = 1; c < NCLS; ++c) if (lg[c] > m) { m = lg[c]; best = c; }
    bool near = false;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) near |= (c != best) & (m - lg[c] < 4e-7f);
    if (p || near) {
        float e[NCLS]; float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) { e[c] = expf(lg[c] - m); sum += e[c]; }
        const float inv = 1.0f / sum;
        float pm = -1.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const float pc = e[c] * inv;
            if (p) p[c] = pc;
            if (pc > pm) { pm = pc; best = c; }
        }
    }
    return best;
}

// pred (and, only if `want_prob`, the probabilities into p) without an escaping conditional pointer: two straight-line instantiations
template <int NCLS>
__device__ __forceinline__ int softmax_argmax_opt(const float (&lg)[NCLS], bool want_prob, float (&p)[NCLS]) {
    if (want_prob) return softmax_argmax<NCLS>(lg, p);
#pragma unroll
    for (int c = 0; c < NCLS; ++c) p[c] = 0.f;
    return softmax_argmax<NCLS>(lg, nullptr);
}

// ---------------------------------------------------------------------------
// Generic implicit-GEMM convolution (3x3 or 1x1, stride 1 or 2) + bias + ReLU
// on the f32 MFMA pipes.  NHWC activations, weights pre-packed in MFMA
// A-fragment order by pack_conv_weights().
// ---------------------------------------------------------------------------
struct ConvArgs {
    const float *in0;   // first source  [N,H,W,C0]
    const float *in1;   // optional second source [N,H,W,C1] (U-Net skip concat, network_ao.py:51)
    int C0, C1;
    const float *wpk;   // packed A fragments
    const float *bias;  // [Cout] folded BN shift (or conv bias)
    float *out;         // [N,Ho,Wo,Cout]
    int N, H, W;        // input spatial size
    int Ho, Wo, Cout;
    int pad_y, pad_x;   // TF 'SAME' pad_before (oracle/fcn_oracle.py same_pads)
    int tiles_y, tiles_x;
    int relu;
    const float *first_w;   // fused first layer (conv_pc_kernel<..., FIRST>): folded conv0_0 weights [9][KC]
    const float *first_b;   //   and bias [KC]; in0 is then the 1-channel network input [N,H,W]
    int up2;            // 0: plain conv.  C > 0: the 4*C output channels are the 4 sub-pixel phases of a
                        // stride-2 transposed conv with C real channels; scatter to out[N,2Ho,2Wo,C]
    const int *in0_map; // optional (Winograd kernel only): image n of the batch reads in0 image in0_map[n]
                        // (ConvLSTM windows share cached U-Net feature frames); in1 / out are not remapped
    int diag;           // diagnostic builds only (-DUKBB_DIAG, env UKBB_CONV_DIAG): ablation bits of conv_pc_kernel
    // bf16-storage tilings only (pc == 5): the 1x1 logits conv + softmax / argmax of network_ao.py:63,159-160 evaluated in the
    // epilogue of the LAST 16-channel conv (out is then not written at all): lg_w [16][lg_ncls], lg_b [lg_ncls]
    const float *lg_w, *lg_b;
    float *lg_logits, *lg_prob;   // optional [N,Ho,Wo,lg_ncls]
    int32_t *lg_pred;             // optional [N,Ho,Wo]
    int lg_ncls;
    int xcd_local;      // kernels_ws.hip only: tile order (set by launch_conv_ws)
    int cout_store;     // bf16-storage tilings (ConvConfig::pc == 5) only: real channel count of `out` when Cout is the zero-padded
                        // count the weights were packed for (a 16-channel layer on the 32-row MFMA); 0 = Cout
    // ---- ConvLSTM cell in the epilogue of the F(2x4) Winograd kernel (kernels_wino24.hip, launch_wino24_lstm; network_ao.py:255-319) ----
    // ls_mode 1 ("x pass", once per call): in0 = the U-Net feature frames [N][H][W][16], Cout = 64 per direction (groups = directions),
    //   filter = the x rows of the gate kernel, bias = the gate bias.  Per frame and direction the epilogue stores gx = W_x * x + b
    //   (ls_gx), and the cell's first step from the zero state: c1 (ls_c_out), h1 (out, [N][H][W][16]).
    // ls_mode 2 (one time step of one direction): in0 = the previous hidden state [N][H][W][16] (image n reads in0_map[n] if given),
    //   Cout = 64, filter = the h rows of the gate kernel, bias = nullptr; gates = W_h * h + gx[ls_gx_map[n]], cell state read from
    //   ls_c_in (entry in0_map[n] if given, else n) and written to ls_c_out[n], h written to out.
    // gx / c live in the consumer lanes' own order ("lane-native": [image][region][wave][tile block][pixel of the 2x4 tile][lane]), the gate
    // channels are packed so that one lane holds i, j, f, o of ONE hidden channel (pack_lstm_gate_weights).
    int ls_mode;
    float *ls_gx;               // mode 1: written; mode 2: read (const in effect)
    const int *ls_gx_map;       // mode 2: frame whose gx window n adds at this step
    const float *ls_c_in;       // mode 2
    float *ls_c_out;
    long long ls_gx_dir, ls_c_dir, ls_h_dir;   // mode 1: floats between the two directions' halves of ls_gx / ls_c_out / out
    float ls_forget_bias;
    int ls_bf16;                // 1: in0, gx and the hidden maps (out) are bf16 in HBM (UKBB_PREC_BF16 on a UNet-LSTM handle); c stays fp32.
                                //    Strides (ls_*_dir) still count ELEMENTS; ls_gx_dir counts gx values (4 per lane-slot)
};

// One compiled tiling of the conv kernel.
struct ConvConfig {
    int id;
    int ks, stride;     // kernel size (1|3), stride (1|2)
    int mb;             // MFMA M block: 16 -> v_mfma_f32_16x16x4_f32, 32 -> v_mfma_f32_32x32x2_f32
    int th, tw;         // output tile per workgroup
    int kc;             // input channels staged in LDS per pass
    int wm, wn, cb;     // waves along Cout / along pixels; Cout blocks per wave
    int lds_bytes;
    int pc;             // 0: single-role kernel; 1: producer/consumer persistent kernel (512 threads);
                        // 2: producer/consumer with the C_in = 1 first layer fused into the producers;
                        // 3: single-role kernel with bf16 operands / fp32 accumulation (mb = 32, kc = 16);
                        // 4: Winograd F(2x2,3x3) producer/consumer kernel (kernels_wino.hip)
                        // 5: as 3 with bf16 activations in HBM on both sides (in0 / in1 / out point at bf16 data), CHANNEL-BLOCKED:
                        //    [N][C/16][H][W][16] -- a 16-channel K chunk of a row of pixels is one contiguous run of 32-byte pixels
                        //    (r04: in plain NHWC a wave's 16-byte pieces of one chunk lay 2 C bytes apart, every piece pulled a whole
                        //    128-byte line out of L2 for 16-32 useful bytes); for C = 16 the two layouts coincide
                        // 6: bf16 storage as 5, weight-stationary persistent kernel of independent waves (kernels_ws.hip)
    const char *name;
    int fuse;           // pc == 5 only: 1 = the C_in = 1 first layer evaluated in this conv's staging (ConvArgs::first_w / first_b,
                        // in0 = the fp32 image), 2 = the logits conv + softmax / argmax in its epilogue (ConvArgs::lg_*); else 0
};

int num_conv_configs();
const ConvConfig &conv_config(int id);
// the persistent last-layer tilings of the bf16 U-Net (ids 324-325) live in kernels_bf16.hip; the three functions above cover them
int num_pk16_configs();
const ConvConfig &pk16_config(int i);
hipError_t launch_conv16_pk(int cfg_id, const ConvArgs &a, hipStream_t s);
// weight-stationary persistent 3x3 stride-1 tilings of the bf16 U-Net (ids 400-, ConvConfig::pc == 6: bf16 storage as pc 5; th = rows per
// tile, tw = 32, wn = independent waves per workgroup, cb = 32-channel Cout blocks per workgroup) live in kernels_ws.hip; the three
// functions above cover them.  LDS need depends on the layer's input channels (the whole packed filter of a Cout group is resident).
int num_ws_configs();
const ConvConfig &ws_config(int i);
int ws_lds_bytes_for(const ConvConfig &c, int cin);
int wst_pack_order(int cout, int v);    // transposed-conv tilings (ids 410-, ks 2): source column of packed virtual channel v
hipError_t launch_conv_ws(int cfg_id, const ConvArgs &a, hipStream_t s);
// cout handled by one workgroup = mb*cb*wm
hipError_t launch_conv(int cfg_id, const ConvArgs &a, hipStream_t s);

// Host-side: pack folded weights W[ks][ks][Cin][Cout] into A-fragment order
// for tiling (mb, kc) and workgroups of ncbl = wm*cb Cout blocks.  Returns floats written.
size_t pack_conv_weights(const float *w, int ks, int cin, int cout, int mb, int kc, int ncbl, float *dst);
// bf16 operand variant (ConvConfig::pc == 3): 8 bf16 per lane per tap, returns dwords written.
size_t pack_conv_weights_bf16(const float *w, int ks, int cin, int cout, int ncbl, float *dst);

// Winograd F(2x2,3x3) producer/consumer kernel (kernels_wino.hip): 3x3, stride 1, C_out % 64 == 0,
// C_in % 16 == 0.  ConvConfig::pc == 4, id 300.  Same ConvArgs; wpk from pack_wino_weights().
// thread-local error string behind ukbb_fcn_last_error() (engine.cpp)
void set_error(const char *fmt, ...);

hipError_t launch_wino(const ConvArgs &a, int ncb /*16-channel blocks per item: 4 or 2*/, int tile_rows /*4: 8x16-pixel regions, 8: 16x8*/,
                       hipStream_t s);
size_t pack_wino_weights(const float *w /*[3][3][cin][cout] folded*/, int cin, int cout, int ncb, float *dst /*16*cin*cout*/);
int wino_lds_bytes();
// Winograd F(2x4,3x3) (kernels_wino24.hip): 64-channel output groups, 8 x 32- or 8 x 16-pixel regions, ConvConfig::pc == 4, ids 304 / 305 / 306 / 307 (306: 8 x 16 over image pairs; 307: 32-channel items)
inline bool is_wino24(const ConvConfig &c) { return c.pc == 4 && c.id >= 304; }
hipError_t launch_wino24(const ConvArgs &a, int tile_cols /*32 | 16*/, int pair /*id 306: image pairs with seam regions, Ho % 8 == 4*/, int ncb /*4 | 2 (id 307)*/,
                         hipStream_t s);
size_t pack_wino24_weights(const float *w /*[3][3][cin][cout] folded*/, int cin, int cout, int ncb, float *dst /*24*cin*cout*/);
int wino24_lds_bytes(int tbw, int pair);
// ConvLSTM forms of the same kernel (ConvArgs::ls_mode 1 | 2): tile_cols 32 | 16.  Both region shapes give identical bits.
hipError_t launch_wino24_lstm(const ConvArgs &a, int tile_cols, hipStream_t s);
// floats of the lane-native gx / c buffers per image (frame or window) for maps of Ho x Wo
size_t wino24_lstm_gx_floats(int Ho, int Wo, int tile_cols);
size_t wino24_lstm_c_floats(int Ho, int Wo, int tile_cols);
// gate filter rows [3][3][cin_total][64] (HWIO, channels i | j | f | o x 16 hidden) -> packed F(2x4) A fragments of the input-channel slice
// [c_first, c_first + 16) with the output channels permuted so that MFMA row 4 g + e of 16-channel block w is gate e of hidden channel
// 4 w + g; `bias_perm` (optional, 64) receives the bias in the same order.  Returns floats written to dst (24 * 16 * 64).
size_t pack_lstm_gate_weights(const float *w, int cin_total, int c_first, const float *bias, float *dst, float *bias_perm);
// bf16 form of the same pair of kernels (kernels_ws.hip, ws_main LS): direct 3x3 conv on v_mfma_f32_32x32x16_bf16 (bf16 hidden maps / features / gx, fp32
// accumulation and cell state), weight-stationary independent waves.  ConvArgs as above with ls_bf16 = 1; its own lane-native layouts and packing:
hipError_t launch_lstm_ws(const ConvArgs &a, hipStream_t s);
size_t lstm_ws_gx_elems(int H, int W);      // bf16 values of gx per image (frame)
size_t lstm_ws_c_floats(int H, int W);      // fp32 values of the cell state per image
size_t pack_lstm_gate_weights_bf16(const float *w, int cin_total, int c_first, const float *bias, float *dst /*9 * 16 * 64 / 2 dwords*/, float *bias_perm);
// r06, ls_mode 3 (a time step with the x half NOT hoisted: in0 = feature frames through ls_gx_map, in1 = previous hidden maps through in0_map, bias = the
// permuted gate bias, no gx): the whole [3][3][32][64] gate kernel as one two-chunk filter
size_t pack_lstm_gate_weights_bf16_xh(const float *w /*[3][3][32][64]*/, const float *bias, float *dst /*9 * 32 * 64 / 2 dwords*/, float *bias_perm);

// ---------------------------------------------------------------------------
// First layer: conv3x3, C_in = 1 (network.py:186 with l = 0), direct stencil.
// ---------------------------------------------------------------------------
struct FirstArgs {
    const float *in;    // [N,H,W,1]
    const float *w;     // [9][Cout] folded
    const float *bias;  // [Cout]
    float *out;         // [N,H,W,Cout]
    int N, H, W, Cout;
    int out_bf16;       // 1: out holds bf16 (UKBB_PREC_BF16 of the U-Net: every activation between layers is bf16)
};
hipError_t launch_first(const FirstArgs &a, hipStream_t s);

// ---------------------------------------------------------------------------
// FCN head (network.py:201-229 + train_network.py:198-199), see kernels_head.hip.
// sqg: G_l[px][64] = W0_l * relu(BN(Ws_l * conv_l[px]))   at the resolution of level l = 1..4
// head: same_dim0 -> out0 (level-0 slice) + sum_l bilinear-gather(G_l) -> out1 -> logits
//       -> softmax / argmax, at full resolution.
// ---------------------------------------------------------------------------
struct SqgArgs {
    const float *x;          // [npix][cin] level-l features
    const float *w_s;        // pack_sq(same_dim_l)            [cin/8][64][4]
    const float *b_s;        // [32]
    const float *w_g;        // pack_rowmap_32x64(out0 rows 32l..32l+31)
    float *out;              // [npix][64]
    long long npix;
    int cin;
};
hipError_t launch_sqg(const SqgArgs &a, hipStream_t s);
struct SqgMultiArgs { SqgArgs lv[4]; int nb[4]; };   // levels 1..4 (C_in 32, 64, 128, 256) and their virtual grid sizes
hipError_t launch_sqg_multi(const SqgArgs lv[4], hipStream_t s);

struct HeadArgs {
    const float *conv0;      // [N,H,W,16] level-0 features
    const float *G[4];       // projected maps of levels 1..4: [N,H>>l,W>>l,64]
    const float *w_s0;       // pack_sq(same_dim0, 16)
    const float *b_s0;       // [32]
    const float *w_o0;       // pack_rowmap_32x64(out0 rows 0..31)
    const float *b_o0;       // [64]
    const float *w_o1;       // pack_rowmap_32x64(out1 rows 0..31) ++ pack_rowmap_32x64(out1 rows 32..63)
    const float *b_o1;       // [64]
    const float *w_o0x3;     // optional (experiment UKBB_HEAD_X3): pack_head_x3(out0 rows 0..31, 32) -- three bf16 pieces per weight; may be null
    const float *w_o1x3;     // optional: pack_head_x3(out1, 64)
    const float *w_lg;       // pack_head_lg: [2][n_class][32]
    const float *b_lg;       // [n_class]
    float *logits;           // optional [N,H,W,n_class]
    float *prob;             // optional
    int32_t *pred;           // optional [N,H,W]
    int N, H, W, n_class;
    int x3;                  // UKBB_PREC_F32X3: out0's level-0 slice and out1 from bf16 pieces (needs w_o0x3 / w_o1x3)
    int diag;                // diagnostic builds only (-DUKBB_DIAG, env UKBB_HEAD_DIAG): ablation bits of fcn_head_pc_kernel
};
hipError_t launch_head(const HeadArgs &a, hipStream_t s);
void pack_sq(const float *w /*[cin][32] folded*/, int cin, float *dst /*cin*32*/);
void pack_rowmap_32x64(const float *w /*32 rows x 64 cols*/, int ld, float *dst /*2*4*64*4*/);
void pack_head_lg(const float *w /*[64][n_class]*/, int n_class, float *dst /*2*n_class*32*/);
void pack_head_x3(const float *w /*[k_in][64 out]*/, int k_in /*32 | 64*/, float *dst /*3*2*(k_in/16)*64*4 dwords*/);

// ---------------------------------------------------------------------------
// U-Net pieces (network_ao.py:48-63)
// ---------------------------------------------------------------------------
// conv2d_transpose 3x3 stride 2 'SAME' (network.py:28-34 via network_ao.py:49) is run by
// conv_mfma_kernel as a 2x2-tap stride-1 conv over the INPUT grid with 4*Cout output channels
// (phase-major: (py,px,co)) and ConvArgs::up2 = Cout.  This builds that dense filter
// [2][2][Cin][4*Cout] from the folded transposed filter w[3][3][Cin][Cout] (already re-laid
// out as HWIO by the engine):  out[2m+py] = sum_a x[m+a-1] * w[kh(py,a)],
//   py = 0: a=0 -> kh=2, a=1 -> kh=0 ;  py = 1: a=1 -> kh=1 (a=0: no tap).
void tconv_as_conv2x2(const float *w, int cin, int cout, float *dst);

// Fused tail of the bf16 U-Net (kernels_tail.hip): up0_0 -> up0_1 -> logits -> softmax / argmax in one launch (network_ao.py:51-63,159-160)
struct TailArgs {
    const float *in0, *in1;     // bf16 [N][H][W][16]: the level-0 skip map (conv0) and the transposed conv's output (up0_t)
    const float *wA0, *wA1;     // pack_tail_weights(): MFMA A fragments of up0_0 (9 taps) and up0_1 (5 tap pairs)
    const float *b0, *b1;       // folded BN shifts [16]
    const float *lg_w, *lg_b;   // logits conv [16][n_class], [n_class]
    float *logits, *prob;       // optional [N,H,W,n_class]
    int32_t *pred;              // optional [N,H,W]
    int N, H, W, ncls;
    unsigned long long *stamps;  // diagnostic builds only (-DUKBB_DIAG, env UKBB_TAIL_STAMPS): per-wave phase cycle sums
    int seg_tiles;               // strip mode (set by the launcher): tile rows per segment of a column strip; 0 = whole strips
};
hipError_t launch_unet_tail(const TailArgs &a, hipStream_t s);
void pack_tail_weights(const float *w0 /*[3][3][32][16] folded*/, const float *w1 /*[3][3][16][16] folded*/, float *dst0 /*9*64*4*/, float *dst1 /*5*64*4*/);

// Fused stem of the bf16 U-Net (kernels_stem.hip): conv0_0 -> conv0_1 in one launch (network_ao.py:31-35 with l = 0)
struct StemArgs {
    const float *image;         // fp32 [N][H][W]
    const float *wA0, *wA1;     // pack_stem_weights() (conv0_0) and the second output of pack_tail_weights() (conv0_1: 5 tap pairs)
    const float *b0, *b1;       // folded BN shifts [16]
    float *out;                 // bf16 [N][H][W][16]
    int N, H, W;
};
hipError_t launch_unet_stem(const StemArgs &a, hipStream_t s);
void pack_stem_weights(const float *w0 /*[3][3][1][16] folded*/, float *dst0 /*64*4*/);

struct LogitsArgs {         // 1x1 conv C -> n_class + bias, softmax / argmax (network_ao.py:63,159-160)
    const float *in;        // [N,H,W,C]
    const float *w;         // [C][n_class]
    const float *bias;
    float *logits; float *prob; int32_t *pred;
    int64_t npix; int C, n_class;
    int in_bf16;            // 1: `in` holds bf16
};
hipError_t launch_logits(const LogitsArgs &a, hipStream_t s);

// ---- BiConvLSTM head of the aortic UNet-LSTM (network_ao.py:255-319), kernels_lstm.hip ----
struct LstmOutArgs {        // one time step's outputs from the two directions' hidden maps (1x1 output conv + bias, softmax, argmax)
    const float *hf, *hb;   // [.][HW][NH] hidden maps of the forward / backward cell at this step
    const int *mapf, *mapb; // optional: window m reads entry mapf[m] / mapb[m] (the first step of a direction lives per FRAME)
    const float *w_out;     // [2*NH][n_class] (forward rows first, network_ao.py:305-312)
    const float *b_out;     // [n_class]
    float *prob;            // per (window m, pixel): n_class floats at prob + m*m_stride + pix*n_class
    float *logits;          // optional, same addressing
    int32_t *pred;          // optional: argmax at pred + m*(m_stride/n_class) + pix
    long long m_stride;     // floats between consecutive windows in prob / logits
    int M, HW, n_class;
    int h_bf16;             // 1: hf / hb hold bf16
};
hipError_t launch_lstm_out(const LstmOutArgs &a, hipStream_t s);

struct LstmTileArgs {       // per-(window, step) outputs + the weighted tiling of deploy_network_ao.py:176-183 in one pass
    const float *hf, *hb;   // [K][Wn][HW][NH] hidden maps per step k and window (entry k = 0 of hf / k = K-1 of hb unused, see h1f / h1b)
    const float *h1f, *h1b; // [frames][HW][NH]: the first step of each direction, per frame (the x pass of the fused kernel)
    const int *map_first, *map_last;   // [Wn]: frame of window w at step 0 / at step K-1
    long long k_stride;     // floats between steps in hf / hb
    const float *w_out, *b_out;
    const int *order;       // [F][K] for frame f: the <= K contributing (window w, position k) pairs packed w*K + k (-1 ends the list),
                            //        sorted the way the reference's loop over t adds them
    const double *wk;       // [K] window weights
    const double *wsum;     // [F] accumulated weight per frame
    float *prob;            // [F][HW][C]
    int32_t *pred;          // [F][HW]
    int F, K, Wn, HW, C;
    int h_bf16;             // 1: hf / hb / h1f / h1b hold bf16 (k_stride still counts elements)
};
hipError_t launch_lstm_tile(const LstmTileArgs &a, hipStream_t s);

}  // namespace ukbb
