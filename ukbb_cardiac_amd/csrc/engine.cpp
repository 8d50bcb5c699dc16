// Engine behind include/ukbb_fcn.h: owns device weights (BN folded, packed in
// MFMA fragment order), the activation workspace in HBM and the launch plan.
//
// Reference counterpart: the TensorFlow session + restored graph of
// common/deploy_network.py:44-49 and the sess.run call at :110-111.
#include "../../include/ukbb_fcn.h"
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace ukbb;

extern "C" int ukbb_fcn_debug_poison_lds(uint32_t pattern, void *stream);     // kernels_prep.hip (debugging aid, not in the public header)
extern "C" int ukbb_fcn_debug_fence_kernel(void *stream);                       // kernels_prep.hip (debugging aid)

namespace {

thread_local std::string g_err;

void set_err(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

}  // namespace

void ukbb::set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

namespace {

#define HIP_TRY(expr, code)                                                            \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return code;                                                               \
        }                                                                              \
    } while (0)

constexpr float BN_EPS = 1e-3f;   // tf.layers.batch_normalization default

struct HostLayer {            // one conv(+BN) unit with BN folded (fp32, same op order as
    std::string name;         // ukbb_cardiac_amd/weights.py fold_bn)
    int ks = 0, cin = 0, cout = 0;
    bool transposed = false, relu = true;
    std::vector<float> w;     // [ks][ks][cin][cout], scale folded in
    std::vector<float> b;     // [cout]
};

struct DevBuf {
    float *p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(float));
        if (e == hipSuccess) n = want;
        return e;
    }
    hipError_t upload(const std::vector<float> &v) {
        hipError_t e = ensure(v.size());
        if (e != hipSuccess) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
    }
};

enum OpKind { OP_FIRST, OP_CONV, OP_HEAD, OP_TCONV, OP_LOGITS, OP_SQG, OP_SQG_MULTI, OP_TAIL, OP_STEM };

struct Op {                    // one kernel launch of the plan
    OpKind kind;
    std::string name;
    int layer = -1;            // index into host layers (OP_FIRST/OP_CONV/OP_TCONV/OP_LOGITS)
    int cfg = -1;              // conv config id
    int in0 = -1, in1 = -1;    // activation buffer ids (-1: network input / none)
    int out = -1;
    int sq[4] = {-1, -1, -1, -1};   // OP_HEAD: squeezed maps of levels 1..4
    int mlayer[4] = {-1, -1, -1, -1}, min_[4] = {-1, -1, -1, -1}, mout[4] = {-1, -1, -1, -1}, mh[4] = {0, 0, 0, 0}, mw[4] = {0, 0, 0, 0};   // OP_SQG_MULTI: levels 1..4
    bool fused_first = false;       // OP_CONV: conv0_0 (C_in = 1) evaluated by this kernel's producers
    bool fused_logits = false;      // OP_CONV (bf16 storage): the 1x1 logits conv + softmax / argmax evaluated in this kernel's epilogue
    bool on_side = false;           // launched on the handle's side stream (fork/join by events)
    int H = 0, W = 0, Ho = 0, Wo = 0, stride = 1, pad_y = 0, pad_x = 0;
    double macs_per_image = 0; // algorithmic
    double mfma_macs_per_image = -1; // issued to the matrix pipe; -1 = same as algorithmic
    double padded_macs_per_image = -1;   // ... including the slots of partly filled tiles / Winograd regions; -1 = same as mfma_macs_per_image
    const float *wpk = nullptr, *bias = nullptr;
};

}  // namespace

struct ukbb_fcn_handle {
    ukbb_fcn_arch arch{};
    int device = 0;
    std::vector<HostLayer> layers;
    std::map<std::string, int> layer_index;

    // device-side parameters
    std::map<std::string, std::unique_ptr<DevBuf>> dev;   // keyed by "<layer>/<what>"

    // activation workspace
    std::vector<std::unique_ptr<DevBuf>> act;
    std::vector<size_t> act_per_image;        // floats per image at the planned H,W
    std::vector<int> act_ch;                  // channels of the map (0: not a channel map); bf16 plans store maps with C > 16 channel-blocked
    std::vector<std::string> act_name;
    DevBuf io_image, io_logits, io_prob, io_pred;   // staging for forward_host

    // plan
    int precision = 0;                        // 0: fp32; 1: bf16 operands for the MFMA convs (fp32 accumulate); 2: fp32 from bf16 pieces (head)
    int plan_h = 0, plan_w = 0, cap_n = 0;
    bool plan_small = false;                  // plan built with the small-batch tilings
    int plan_n = 0;                           // ... for this largest batch
    int max_n = 0;                            // largest batch this handle was asked for (reserve / forward): the small-batch plan is
                                              // used only while that stays <= SMALL_BATCH, so a large-batch caller's tail batches do
                                              // not flip the plan (a rebuild re-allocates the workspace) back and forth
    bool plan_bfio = false;                   // plan stores every activation between layers as bf16 (UKBB_PREC_BF16, U-Net)
    std::vector<Op> ops;
    int last_n = 0;

    // UNet-LSTM (kind 2)
    int feat_buf = -1;                        // activation index of net['conv0_up']
    bool lstm_bf_hoist = true;                // bf16 time steps read the hoisted gx (r05; default) -- false (UKBB_LSTM_BF16_UNHOIST at plan build): they re-multiply x (r06 experiment)
    bool lstm_bf_wino = false;                // UKBB_LSTM_BF16_WINOGRAD at plan build: fp32 Winograd arithmetic on bf16 storage (A/B form)
    bool lstm_bw_zero = false;                // the backward cell's kernel and bias are all zero (the single-direction head Conv_LSTM of network_ao.py:214-252 embedded by
                                              // weights.embed_unidirectional_lstm): its hidden maps are exactly zero, run_bilstm clears them instead of running its time steps
    int lstm_tile_cols = 0;                   // region shape of the fused gate-conv / cell kernel (kernels_wino24.hip): 32 | 16
    // lstm_gx / lstm_c1 / lstm_h1: per direction and FRAME (the x pass); lstm_c: per window; lstm_hall: per direction, step and window
    DevBuf lstm_gx, lstm_c1, lstm_h1, lstm_c, lstm_hall, lstm_probw, lstm_aux;   // lstm_aux: int maps / orders / double weights (raw bytes)
    long long lstm_aux_key = -1;              // which tables lstm_aux holds (shape-keyed, uploaded once per shape)

    // image-slice streams (experiment UKBB_SPLIT, run_plan): consecutive conv ops run as S independent image ranges on S streams
    // side stream for kernels that only feed the head (sqg_l): fork after level l, join before the head
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t ev_split_fork = nullptr, ev_split_join = nullptr;
    int debug_first_op = 0, debug_last_op = 1 << 30;   // UKBB_DEBUG_OPS="first,last" at plan build: run_plan launches only these ops
    int split_first = -1, split_last = -2;      // op range run as two half-batch chains (UKBB_SPLIT_FROM at plan build)
    std::vector<hipEvent_t> ev_fork, ev_join;
    bool use_side = false;

    // timing
    bool timing = false;
    int timing_only = -1;                     // -1: every kernel; else only this op index
    std::vector<hipEvent_t> ev;               // 2 per op
    std::vector<double> t_sum;
    std::vector<int64_t> t_cnt;
    bool ev_pending = false;

    ~ukbb_fcn_handle() {
        for (auto e : ev) (void)hipEventDestroy(e);
        if (ev_split_fork) (void)hipEventDestroy(ev_split_fork);
        if (ev_split_join) (void)hipEventDestroy(ev_split_join);
        if (side2) (void)hipStreamDestroy(side2);
        for (auto e : ev_fork) (void)hipEventDestroy(e);
        for (auto e : ev_join) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

const float *dev_ptr(ukbb_fcn_handle *h, const std::string &key) {
    auto it = h->dev.find(key);
    return it == h->dev.end() ? nullptr : it->second->p;
}

int upload(ukbb_fcn_handle *h, const std::string &key, const std::vector<float> &v) {
    auto &slot = h->dev[key];
    if (!slot) slot.reset(new DevBuf);
    HIP_TRY(slot->upload(v), UKBB_EDEVICE);
    return UKBB_OK;
}

// ---- architecture walk ---------------------------------------------------------
struct Spec { std::string name; int ks, cin, cout; bool bn, bias, transposed; };

bool arch_specs(const ukbb_fcn_arch &a, std::vector<Spec> &out) {
    out.clear();
    if (a.n_level < 1 || a.n_level > UKBB_FCN_MAX_LEVEL || a.n_class < 1) return false;
    int cin = 1;
    char nm[64];
    for (int l = 0; l < a.n_level; ++l) {
        if (a.n_block[l] < 1 || a.n_filter[l] < 1) return false;
        for (int i = 0; i < a.n_block[l]; ++i) {
            snprintf(nm, sizeof nm, "conv%d_%d", l, i);
            out.push_back({nm, 3, cin, a.n_filter[l], true, false, false});
            cin = a.n_filter[l];
        }
    }
    if (a.kind == UKBB_KIND_FCN) {
        for (int l = 0; l < a.n_level; ++l) {
            snprintf(nm, sizeof nm, "same_dim%d", l);
            out.push_back({nm, 1, a.n_filter[l], a.same_dim, true, false, false});
        }
        out.push_back({"out0", 1, a.same_dim * a.n_level, a.fc, true, false, false});
        out.push_back({"out1", 1, a.fc, a.fc, true, false, false});
        out.push_back({"logits", 1, a.fc, a.n_class, false, true, false});
    } else if (a.kind == UKBB_KIND_UNET || a.kind == UKBB_KIND_UNET_LSTM) {
        for (int l = a.n_level - 2; l >= 0; --l) {
            snprintf(nm, sizeof nm, "up%d_t", l);
            out.push_back({nm, 3, a.n_filter[l + 1], a.n_filter[l], true, false, true});
            int c = 2 * a.n_filter[l];
            for (int i = 0; i < a.n_block[l]; ++i) {
                snprintf(nm, sizeof nm, "up%d_%d", l, i);
                out.push_back({nm, 3, c, a.n_filter[l], true, false, false});
                c = a.n_filter[l];
            }
        }
        if (a.kind == UKBB_KIND_UNET) {
            out.push_back({"logits", 1, a.n_filter[0], a.n_class, false, true, false});
        } else {                                  // BiConv_LSTM, network_ao.py:255-319 (same_dim = hidden channels)
            if (a.same_dim < 1 || a.fc < 1) return false;
            out.push_back({"lstm_fw", 3, a.n_filter[0] + a.same_dim, 4 * a.same_dim, false, true, false});
            out.push_back({"lstm_bw", 3, a.n_filter[0] + a.same_dim, 4 * a.same_dim, false, true, false});
            out.push_back({"lstm_out", 1, 2 * a.same_dim, a.n_class, false, true, false});
        }
    } else {
        return false;
    }
    return true;
}

size_t spec_floats(const Spec &s) {
    size_t n = (size_t)s.ks * s.ks * s.cin * s.cout;
    if (s.bn) n += 4 * (size_t)s.cout;
    if (s.bias) n += s.cout;
    return n;
}

bool supported(const ukbb_fcn_arch &a, std::string &why) {
    if (a.n_level != 5) { why = "n_level must be 5"; return false; }
    if (a.n_filter[0] != 16) { why = "n_filter[0] must be 16"; return false; }
    for (int l = 1; l < a.n_level; ++l)
        if (a.n_filter[l] % 32) { why = "n_filter[l>0] must be multiples of 32"; return false; }
    if (a.kind == UKBB_KIND_FCN) {
        if (a.same_dim != 32 || a.fc != 64) { why = "FCN head kernel is built for same_dim=32, fc=64"; return false; }
        if (a.n_class < 2 || a.n_class > 6) { why = "n_class must be in 2..6"; return false; }
    } else {
        if (a.n_class < 2 || a.n_class > 4) { why = "UNet n_class must be in 2..4"; return false; }
        if (a.kind == UKBB_KIND_UNET_LSTM) {
            if (a.same_dim != 16) { why = "ConvLSTM kernels are built for 16 hidden channels"; return false; }
            if (a.fc < 1 || a.fc > 31 || !(a.fc & 1)) { why = "the time window must be odd and < 32 steps"; return false; }
        }
    }
    return true;
}

// ---- conv tiling choice --------------------------------------------------------
int override_cfg(const std::string &layer) {
    const char *env = getenv("UKBB_CONV_CFG");     // e.g. "conv0_1:7,conv4_1:6"
    if (!env) return -1;
    std::string s(env);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t e = s.find(',', pos);
        if (e == std::string::npos) e = s.size();
        std::string item = s.substr(pos, e - pos);
        size_t c = item.find(':');
        if (c != std::string::npos && item.substr(0, c) == layer) return atoi(item.c_str() + c + 1);
        pos = e + 1;
    }
    return -1;
}

// Tilings measured best on MI355X (tools/tune_convs.py, profiles/r01_tune_convs*.txt): {ks, stride, cin, cout, cfg}
// per layer type, for large batches (tuned at N = 64, 192x208) and for small ones (tuned at N = 10, the
// reference's own per-frame call, deploy_network.py:103-111: there the persistent kernels have fewer work
// items than CUs, and tilings with smaller channel groups / tiles win).  Other image sizes of the same layer
// type reuse the entry (e.g. the long-axis models at 176x208).
struct Tuned { int ks, stride, cin, cout, cfg, alt, alt2, alt3 = -1; };   // alt.. (or -1): the first of the four whose tiles divide the map wins
const Tuned g_tuned_large[] = {
    {3, 1, 16, 16, 11, -1, -1}, {3, 2, 16, 32, 120, 123, -1},  {3, 1, 32, 32, 307, 301, -1},
    {3, 2, 32, 64, 124, 123, 142, 145},  {3, 1, 64, 64, 304, 300, -1},  {3, 2, 64, 128, 124, 123, 142, 145},
    {3, 1, 128, 128, 304, 300, -1},  {3, 2, 128, 256, 124, 123, 142, 145}, {3, 1, 256, 256, 304, 305, 300},
};      // r03: 13x16 tiles (145) divide the 208x256 pyramid (52x64, 26x32, 13x16: conv2_0 112 -> 86 us, conv3_0 / conv4_0 -6 / -7 at N = 64);
        // r02: the stride-2 layers moved to the producer/consumer kernel once its loads ran two stages ahead (profiles/r02_notes.md);
        // its straight-line producer needs tiles that divide the map: 12x13 tiles for the 192x208 pyramid, 8x16 (123) for the
        // power-of-two maps of the aortic U-Net (256x256: 148 / 143 / 133 / 136 us instead of 184 / 208 / 159 / 156 at N = 100),
        // 11x13 (142) for the long-axis models' 176x208 pyramid (88x104, 44x52, 22x26, 11x13)
const Tuned g_tuned_small[] = {
    {3, 1, 16, 16, 11, -1, -1}, {3, 2, 16, 32, 29, -1, -1},  {3, 1, 32, 32, 301, -1, -1},
    {3, 2, 32, 64, 20, -1, -1},  {3, 1, 64, 64, 300, -1, -1},  {3, 2, 64, 128, 123, -1, -1},
    {3, 1, 128, 128, 301, -1, -1},  {3, 2, 128, 256, 26, -1, -1}, {3, 1, 256, 256, 301, -1, -1},
};
constexpr int SMALL_BATCH = 16;
// The small-batch table is opt-in (UKBB_SMALL_BATCH_TILINGS=1; +30 % at N = 10): with it the tiling, and so
// the fp32 summation order, would depend on the batch size, and by default the engine guarantees bit-identical
// results for a slice whatever batch it is part of (tests: batch independence).
bool small_batch_tilings() { static const bool on = getenv("UKBB_SMALL_BATCH_TILINGS") != nullptr; return on; }

// A tuned tiling is reused for another image size only if its tiles still fit that size well.
bool tile_fit_ok(const ConvConfig &c, int Ho, int Wo) {
    const int th = c.th, tw = c.tw;
    const double covered = (double)((Ho + th - 1) / th * th) * ((Wo + tw - 1) / tw * tw);
    // Winograd kept its lead over the direct tilings down to 61 % region fill (12x13 maps, r01 sweep)
    // F(2x4) regions: at 61 % fill (12 x 13 maps) the F(2x2) kernel with its half regions (81 %) is as fast, at 74 % (44 x 52, 22 x 26) F(2x4) still wins by 2 %
    return (double)Ho * Wo >= (is_wino24(c) ? 0.7 : c.pc == 4 ? 0.5 : 0.8) * covered;
}
// Fallback preference (small tiles / high occupancy won everywhere in the sweep).
const int g_pref[] = {4, 5, 18, 3, 11, 7, 31, 23, 22, 29, 27, 26};

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// bf16: 0 = fp32 tilings, 1 = bf16 operands / fp32 storage (pc 3), 2 = bf16 operands and storage (pc 5)
bool cfg_valid(const ConvConfig &c, int ks, int stride, int c0, int c1, int cout, bool fused_first = false,
               int bf16 = 0, int fuse = 0) {
    if (c.ks != ks || c.stride != stride) return false;
    if (c.fuse != fuse) return false;
    if ((c.pc == 2) != fused_first) return false;
    if ((c.pc == 3) != (bf16 == 1) || (c.pc == 5 || c.pc == 6) != (bf16 == 2)) return false;
    if (c.pc == 5 || c.pc == 6) cout = round_up(cout, 32);         // 16-channel layers run zero-padded on the 32-row MFMA
    if (c.pc == 6) {                                  // weight-stationary: the Cout group's whole packed filter + the waves' rings in LDS
        const int nch = (c0 + c1) / 16;
        if (stride != 1 || (c0 + c1) % 16) return false;
        if (ks == 2) {                                // transposed conv as 2x2 sub-pixel conv: cout = 4 x real channels (16, or multiples of 32)
            const int real = cout / 4;
            if (c1 || cout % 64 || (real != 16 && real % 32) || (real == 16 ? nch != 2 : (nch != 4 && nch != 8))) return false;
        } else if (c.kc == 32) {                      // weights through a ring: any even number of chunks, source switch at an even chunk
            if (ks != 3 || nch < 2 || (nch & 1) || (c1 && (c0 / 16) % 2)) return false;
        } else if (ks != 3 || (c1 && c1 != c0) || (nch != 1 && nch != 2 && nch != 4 && nch != 8) || (c1 && nch < 2)) return false;
        if (cout % (32 * c.cb)) return false;
        return ws_lds_bytes_for(c, c0 + c1) <= 160 * 1024;
    }
    if (c.pc == 4) {                                  // Winograd: 3x3 s1, 64-channel output groups, single source ok
        static const bool off = getenv("UKBB_NO_WINOGRAD") != nullptr;
        if (c.id == 306) {                            // image pairs with seam regions (maps with Ho % 8 == 4): only where UKBB_CONV_CFG names it -- at N = 64 its 384
            const char *e = getenv("UKBB_CONV_CFG");   // items leave half the CUs idle in the second round, and the plan must not depend on the batch (r04_notes.md)
            if (!e || !strstr(e, ":306")) return false;
        }
        if (is_wino24(c)) {                           // F(2x4,3x3), kernels_wino24.hip: 64-channel groups, K >= 64 (the MFMA-bound layers; no frame map)
            static const bool off24 = getenv("UKBB_NO_WINOGRAD24") != nullptr;
            if (off24) return false;
            if (c.wm == 2) { if (cout != 32) return false; }   // 32-channel items (307): the layers with exactly 32 output channels
            else if (cout % 64 || c0 + c1 < 32) return false;   // K = 32: the ConvLSTM gate convs (16 + 16 -> 64)
        }
        return !off && !fused_first && ks == 3 && stride == 1 && cout % (16 * c.wm) == 0 && c0 % 16 == 0 && c1 % 16 == 0;
    }
    if (c.pc == 2 && cout != c.mb * c.cb * c.wm) return false;   // fused kernel stages its weights once: one Cout group
    if (c.lds_bytes > 160 * 1024) return false;      // LDS per CU on gfx950
    const int group = c.mb * c.cb * c.wm;
    return !(cout % group || c0 % c.kc || c1 % c.kc);
}

// Winograd regions are 8x16 (ids 300/301) or 16x8 pixels (302/303): take the orientation with fewer regions.
int wino_orient(int id, int Ho, int Wo) {
    if (id < 300 || id > 303) return id;
    const int base = 300 + (id & 1);
    const long long r_8x16 = (long long)((Ho + 7) / 8) * ((Wo + 15) / 16), r_16x8 = (long long)((Ho + 15) / 16) * ((Wo + 7) / 8);
    return r_16x8 < r_8x16 ? base + 2 : base;
}

int choose_cfg_raw(const std::string &layer, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int N,
                   bool fused_first, int want_bf16);
int find_cfg(int id, ConvConfig &out);

// Small batches (N <= SMALL_BATCH, e.g. the reference's own sess.run of one frame's 10 slices, deploy_network.py:103-111): the
// deep levels have fewer work items than the chip has CUs, and a CU streaming an item's weights alone pulls only ~25-50 GB/s
// from L2, so those layers are bound by the number of CUs at work.  Swap the tiling for a FINER SIBLING THAT COMPUTES EVERY
// OUTPUT WITH THE SAME ARITHMETIC -- same algorithm, MFMA shape, channels per stage and tile, only fewer output channels per
// work item -- so results stay bit-identical whatever batch a slice is part of (tests: batch independence, slices of the
// bench batch against single-slice runs).  The r01 small-batch table (other tiles / KC) stays opt-in for that reason.
int finer_sibling(int id, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int N) {
    static const bool off = getenv("UKBB_NO_SMALL_BATCH_SIBLINGS") != nullptr;    // A/B knob
    if (off || N > SMALL_BATCH) return id;
    static const int sib[][2] = {{300, 301}, {302, 303},     // Winograd: 64 -> 32 output channels per item
                                 {124, 141}};                // stride-2 producer/consumer, mb16 12x13 kc8: Cout blocks per wave 2 -> 1
    ConvConfig c, f;
    if (find_cfg(id, c)) return id;
    for (const auto &p : sib) {
        if (p[0] != id || find_cfg(p[1], f) || !cfg_valid(f, ks, stride, c0, c1, cout)) continue;
        const long long tiles = (long long)((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw) * N;
        const long long items = tiles * (cout / (c.pc == 4 ? 16 * c.wm : c.mb * c.cb * c.wm));
        if (items <= device_cu_count() / 2) return p[1];      // at most half the CUs (of the current device) at work: halve the item (r03 on 256 CUs: 160-210 items were faster left alone)
    }
    return id;
}

// Winograd F(2x4,3x3) comes with 8 x 32-pixel regions (304) and 8 x 16 (305); both compute every tile with the same arithmetic (same tile
// grid, same transforms, same K order), so the choice is a matter of filling the CUs: the region shape whose item count wastes less of
// the last round wins, 304 on a tie (fewer, longer items: FCN level 2 60 us against 65); small batches take the finer one.
int pick_wino24(int id, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int N) {
    if (id != 304 && id != 305) return id;
    const int cus = device_cu_count();
    int best = id; double best_eff = -1.0;
    for (int cand : {304, 305}) {
        ConvConfig c;
        if (find_cfg(cand, c) || !cfg_valid(c, ks, stride, c0, c1, cout) || !tile_fit_ok(c, Ho, Wo)) continue;
        const long long items = (long long)N * ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw) * (cout / 64);
        const long long rounds = (items + cus - 1) / cus;
        // makespan in units of an 8 x 16 region's work (an 8 x 32 item is two): the shorter wins -- that counts the padding columns of the
        // wider regions as well as the idle CUs of the last round
        double eff = 1.0 / (double)(rounds * (c.tw / 16));
        if (N <= SMALL_BATCH) eff = cand == 305 ? 2.0 : 1.0;          // fewer items than CUs either way: more of them
        if (eff > best_eff + 1e-12) { best_eff = eff; best = cand; }
    }
    return best;
}

int choose_cfg(const std::string &layer, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int N,
               bool fused_first = false, int want_bf16 = 0) {
    const int id = choose_cfg_raw(layer, ks, stride, c0, c1, cout, Ho, Wo, N, fused_first, want_bf16);
    if (override_cfg(layer) >= 0) return id;
    return finer_sibling(wino_orient(pick_wino24(id, ks, stride, c0, c1, cout, Ho, Wo, N), Ho, Wo), ks, stride, c0, c1, cout, Ho, Wo, N);
}

// bf16-storage tilings measured best per layer type of the aortic U-Net at N = 100 x 256 x 256 (tools/sweep_convs.py with
// PREC=bf16, profiles/r03_sweep_bf16.txt): {ks, stride, cin (both sources), cout (4 x cout for the 2x2 form of a transposed conv), cfg}.
// Levels 2-4 sit on a 35-50 us floor per launch whatever the tiling (launch + first-load latency + tail at 100-400 tiles);
// the table mostly avoids the bad cases (conv3_0 108 -> 47 us, up2_0 104 -> 80, conv2_0 61 -> 45).
// r04: the weight-stationary persistent tilings (400-403, kernels_ws.hip) where a Cout group's whole filter fits LDS (K <= 1152); up3_0
// (K = 2304) on the ring-streamed form 422 (72 vs 78 us; one barrier per chunk keeps it from the ws rate, r04_notes.md)
const Tuned g_tuned_bfio[] = {
    {3, 1, 16, 16, 236, 232, -1},   {3, 1, 32, 32, 401, 232, -1},   {3, 1, 64, 64, 402, 235, 232},    {3, 1, 128, 128, 400, 235, 232},
    {3, 1, 256, 256, 239, 232, -1}, {3, 1, 256, 128, 422, 239, 232}, {3, 1, 128, 64, 400, 239, 232},  {3, 1, 64, 32, 401, 232, -1},
    {3, 1, 32, 16, 401, 236, 232},
    {3, 2, 16, 32, 241, -1, -1},    {3, 2, 32, 64, 242, 241, -1},   {3, 2, 64, 128, 244, 241, -1},  {3, 2, 128, 256, 244, 241, -1},
    {2, 1, 256, 512, 253, 251, -1}, {2, 1, 128, 256, 411, 253, 251}, {2, 1, 64, 128, 411, 253, 251},  {2, 1, 32, 64, 410, 258, 253},
};

int choose_cfg_raw(const std::string &layer, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int N,
                   bool fused_first, int want_bf16) {
    if (want_bf16 == 2 && !fused_first && override_cfg(layer) < 0) {
        for (const Tuned &t : g_tuned_bfio) {
            if (t.ks != ks || t.stride != stride || t.cin != c0 + c1 || t.cout != cout) continue;
            for (int cand : {t.cfg, t.alt, t.alt2}) {
                ConvConfig cc;
                if (cand >= 0 && find_cfg(cand, cc) == 0 && cfg_valid(cc, ks, stride, c0, c1, cout, false, 2) && tile_fit_ok(cc, Ho, Wo)) return cand;
            }
        }
    }
    if (want_bf16 && !fused_first) {          // bf16 tilings first; fall back to fp32 where none fits (e.g. Cout = 16 with fp32 storage)
        const int forced_bf = override_cfg(layer);
        double best = 1e300; int best_id = -1;
        for (int i = 0; i < num_conv_configs(); ++i) {
            const ConvConfig &c = conv_config(i);
            if (!cfg_valid(c, ks, stride, c0, c1, cout, false, want_bf16)) continue;
            if (c.id == forced_bf) return c.id;
            const int group = c.mb * c.cb * c.wm;
            const int coutp = want_bf16 == 2 ? round_up(cout, 32) : cout;
            const int tiles = ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
            const int npb = (c.th * c.tw + c.mb - 1) / c.mb, pbw = (npb + c.wn - 1) / c.wn;
            const double cost = (double)tiles * (coutp / group) * pbw * c.cb;
            if (cost < best) { best = cost; best_id = c.id; }
        }
        if (best_id >= 0) return best_id;
        if (want_bf16 == 2) return -1;        // bf16 storage has no fp32 fallback
    }
    const int forced = override_cfg(layer);
    ConvConfig fc;
    if (forced >= 0) {
        for (int i = 0; i < num_conv_configs(); ++i)
            if (conv_config(i).id == forced && cfg_valid(conv_config(i), ks, stride, c0, c1, cout, fused_first)) return forced;
    }
    (void)fc;
    if (!fused_first && c1 == 0) {
        const bool small = small_batch_tilings() && N <= SMALL_BATCH;
        const Tuned *tab = small ? g_tuned_small : g_tuned_large;
        const size_t ntab = small ? sizeof(g_tuned_small) / sizeof(Tuned) : sizeof(g_tuned_large) / sizeof(Tuned);
        for (size_t j = 0; j < ntab; ++j) {
            const Tuned &t = tab[j];
            if (t.ks == ks && t.stride == stride && t.cin == c0 && t.cout == cout) {
                const int cand[4] = {t.cfg, t.alt, t.alt2, t.alt3};
                int first_ok = -1;
                for (int k = 0; k < 4; ++k) {
                    ConvConfig cc;
                    if (cand[k] < 0 || find_cfg(cand[k], cc) || !cfg_valid(cc, ks, stride, c0, c1, cout) || !tile_fit_ok(cc, Ho, Wo)) continue;
                    // F(2x4) on 32-channel layers pays only where its 8 x 32 regions fill the map (U-Net 128 x 128: 177 -> 150 us; FCN 96 x 104: 81 %
                    // fill against 100 % of the 16 x 8 F(2x2) regions, no gain)
                    if (cand[k] == 307 && (Ho % 8 || Wo % 32)) continue;
                    if (Ho % cc.th == 0 && Wo % cc.tw == 0) return cand[k];      // tiles divide the map: straight-line producer applies
                    if (first_ok < 0) first_ok = cand[k];
                }
                if (first_ok >= 0) return first_ok;
            }
        }
    }
    if (!fused_first && c1 > 0 && ks == 3 && stride == 1 && cout == 32) {
        // skip-concat conv of the U-Net's level 1 (network_ao.py:51-53, 32 + 32 -> 32): the two-source Winograd kernel in its
        // 32-channel form (r02 sweep at 256x256, N = 100: 345 us against 544 for the best direct tiling)
        ConvConfig cw;
        if (Ho % 8 == 0 && Wo % 32 == 0 && find_cfg(307, cw) == 0 && cfg_valid(cw, ks, stride, c0, c1, cout)) return 307;   // F(2x4): 321 -> 252 us (r04)
        if (find_cfg(301, cw) == 0 && cfg_valid(cw, ks, stride, c0, c1, cout) && tile_fit_ok(cw, Ho, Wo)) return 301;
    }
    double best = 1e300;
    int best_id = -1;
    for (int i = 0; i < num_conv_configs(); ++i) {
        const ConvConfig &c = conv_config(i);
        if (c.id == 306 || !cfg_valid(c, ks, stride, c0, c1, cout, fused_first)) continue;
        const int group = c.mb * c.cb * c.wm;
        const int tiles = ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
        const int npb = (c.th * c.tw + c.mb - 1) / c.mb;
        const int pbw = (npb + c.wn - 1) / c.wn;
        // matrix-pipe cycles per wave x workgroups = padded work (tile overhang + block rounding)
        const double cyc = (double)pbw * c.cb * (ks * ks * (c0 + c1) / (c.mb == 32 ? 2 : 4)) * (c.mb == 32 ? 64 : 32);
        double cost = (double)tiles * (cout / group) * cyc;
        // Winograd: one stage (16 input channels of one 8x16 region, 64 output channels) costs ~5.2k cycles
        // per CU measured; 5800 puts it on the scale of the direct estimate above (which ignores the direct
        // kernels' ~70 % matrix-pipe efficiency), calibrated on the three tuned shapes.
        if (c.pc == 4) cost = (double)tiles * (cout / (16 * c.wm)) * ((c0 + c1) / 16) * (is_wino24(c) ? (c.tw == 32 ? 9300.0 : 4700.0) : c.wm == 4 ? 5800.0 : 3500.0);   // F(2x4): 256 / 128 pixels per stage
        int rank = 12;
        for (int r = 0; r < (int)(sizeof(g_pref) / sizeof(g_pref[0])); ++r)
            if (g_pref[r] == c.id) { rank = r % 6; break; }
        cost *= 1.0 + 0.04 * rank;
        if (cost < best) { best = cost; best_id = c.id; }
    }
    return best_id;
}

int find_cfg(int id, ConvConfig &out) {
    for (int i = 0; i < num_conv_configs(); ++i)
        if (conv_config(i).id == id) { out = conv_config(i); return 0; }
    return -1;
}

// ---- plan ------------------------------------------------------------------------
int new_act(ukbb_fcn_handle *h, const std::string &name, size_t per_image, int channels = 0) {
    h->act.emplace_back(new DevBuf);
    h->act_per_image.push_back(per_image);
    h->act_name.push_back(name);
    h->act_ch.push_back(channels);
    return (int)h->act.size() - 1;
}

int ensure_packed(ukbb_fcn_handle *h, int layer, const ConvConfig &c, const float **wpk) {
    const HostLayer &L = h->layers[layer];
    char key[128];
    const bool bfpk = c.pc == 3 || c.pc == 5 || c.pc == 6;
    const int coutp = (c.pc == 5 || c.pc == 6) ? round_up(L.cout, 32) : L.cout;
    const bool w24 = is_wino24(c);
    snprintf(key, sizeof key, "%s/pk%s_mb%d_kc%d_g%d", L.name.c_str(), bfpk ? "bf16" : w24 ? "wino24" : c.pc == 4 ? "wino" : "", c.mb, c.kc, c.wm * c.cb);
    if (!dev_ptr(h, key)) {
        std::vector<float> pk(w24 ? (size_t)24 * L.cin * L.cout : c.pc == 4 ? (size_t)16 * L.cin * L.cout : (size_t)L.ks * L.ks * L.cin * coutp);
        if (w24) pack_wino24_weights(L.w.data(), L.cin, L.cout, c.wm, pk.data());
        else if (c.pc == 4) pack_wino_weights(L.w.data(), L.cin, L.cout, c.wm, pk.data());
        else if (bfpk && coutp != L.cout) {           // zero rows up to the MFMA's 32
            std::vector<float> wp((size_t)L.ks * L.ks * L.cin * coutp, 0.f);
            for (size_t r = 0; r < (size_t)L.ks * L.ks * L.cin; ++r)
                std::copy(L.w.begin() + r * L.cout, L.w.begin() + (r + 1) * L.cout, wp.begin() + r * coutp);
            pack_conv_weights_bf16(wp.data(), L.ks, L.cin, coutp, c.wm * c.cb, pk.data());
        }
        else if (bfpk) pack_conv_weights_bf16(L.w.data(), L.ks, L.cin, L.cout, c.wm * c.cb, pk.data());
        else pack_conv_weights(L.w.data(), L.ks, L.cin, L.cout, c.mb, c.kc, c.wm * c.cb, pk.data());
        int rc = upload(h, key, pk);
        if (rc) return rc;
    }
    if (coutp != L.cout && !dev_ptr(h, L.name + "/bias_pad")) {
        std::vector<float> bp((size_t)coutp, 0.f);
        std::copy(L.b.begin(), L.b.end(), bp.begin());
        int rc = upload(h, L.name + "/bias_pad", bp);
        if (rc) return rc;
    }
    *wpk = dev_ptr(h, key);
    return UKBB_OK;
}

// UKBB_PREC_BF16 on the aortic U-Net: bf16 operands AND bf16 activations in HBM between all layers (r03);
// on the other graphs: bf16 operands, fp32 storage (r01).
// r05: the U-Net of a UNet-LSTM handle takes the same bf16-storage plan (its last map, net['conv0_up'], feeds the ConvLSTM as bf16; the
// LSTM then keeps gx and the hidden maps in bf16 as well, cell state and arithmetic fp32: run_bilstm)
int bf16_mode(const ukbb_fcn_handle *h) { return h->precision != 1 ? 0 : h->arch.kind != UKBB_KIND_FCN ? 2 : 1; }

// bf16 storage: the fused variants of the level-0 tilings (ConvConfig::fuse: 1 = first layer in the staging, 2 = logits in the
// epilogue), first fit in measured order; -1 if none fits (the plan then keeps that layer as a launch of its own).
int pick_fused_bf_cfg(const std::string &lname, int ks, int stride, int c0, int c1, int cout, int Ho, int Wo, int fuse_bf) {
    const int forced = override_cfg(lname);
    // fused logits: the persistent kernel first (kernels_bf16.hip: 104-110 vs 124 us), then the tile-per-workgroup tilings in
    // measured order; fused first layer: tile-per-workgroup only (its persistent form was no faster, r03_notes.md)
    for (int cand : {forced, fuse_bf == 1 ? 296 : 404, fuse_bf == 1 ? 294 : 325, fuse_bf == 1 ? 295 : 324, fuse_bf == 1 ? -1 : 298, fuse_bf == 1 ? -1 : 297, fuse_bf == 1 ? -1 : 299}) {
        ConvConfig cc;
        if (cand >= 0 && find_cfg(cand, cc) == 0 && cfg_valid(cc, ks, stride, c0, c1, cout, false, 2, fuse_bf) &&
            (cand == forced || tile_fit_ok(cc, Ho, Wo))) return cand;
    }
    return -1;
}

int add_conv(ukbb_fcn_handle *h, const std::string &lname, int in0, int in1, int c1, int H, int W, int stride,
             int n_hint, int *out_buf, bool fused_first = false, bool fused_logits = false) {
    const int li = h->layer_index.at(lname);
    const HostLayer &L = h->layers[li];
    Op op;
    op.kind = OP_CONV; op.name = lname; op.layer = li; op.in0 = in0; op.in1 = in1;
    op.H = H; op.W = W; op.stride = stride;
    op.Ho = (H + stride - 1) / stride; op.Wo = (W + stride - 1) / stride;
    // TF 'SAME' pad_before
    op.pad_y = std::max((op.Ho - 1) * stride + L.ks - H, 0) / 2;
    op.pad_x = std::max((op.Wo - 1) * stride + L.ks - W, 0) / 2;
    const int c0 = L.cin - c1;
    op.fused_first = fused_first;
    const int fuse_bf = bf16_mode(h) != 2 ? 0 : fused_first ? 1 : fused_logits ? 2 : 0;
    if (fuse_bf) op.cfg = pick_fused_bf_cfg(lname, L.ks, stride, c0, c1, L.cout, op.Ho, op.Wo, fuse_bf);
    else
    op.cfg = choose_cfg(lname, L.ks, stride, c0, c1, L.cout, op.Ho, op.Wo, n_hint, fused_first, bf16_mode(h));
    op.fused_logits = fuse_bf == 2;
    if (op.cfg < 0) { set_err("no conv tiling for layer %s (ks %d stride %d cin %d+%d cout %d)", lname.c_str(), L.ks, stride, c0, c1, L.cout); return UKBB_EARCH; }
    ConvConfig c;
    find_cfg(op.cfg, c);
    int rc = ensure_packed(h, li, c, &op.wpk);
    if (rc) return rc;
    op.bias = ((c.pc == 5 || c.pc == 6) && L.cout % 32) ? dev_ptr(h, lname + "/bias_pad") : dev_ptr(h, lname + "/bias");
    op.out = new_act(h, lname, (size_t)op.Ho * op.Wo * L.cout, L.cout);
    op.macs_per_image = (double)op.Ho * op.Wo * L.ks * L.ks * L.cin * L.cout;
    if (is_wino24(c)) {
        op.mfma_macs_per_image = op.macs_per_image * (24.0 / 72.0);   // F(2x4,3x3): 24 products per 8 outputs
        const double regs = (double)((op.Ho + 7) / 8) * ((op.Wo + c.tw - 1) / c.tw);      // every region issues all its tile slots (c.tw / 4 x 4)
        op.padded_macs_per_image = regs * c.tw * 24.0 * L.cin * L.cout;
    } else if (c.pc == 4) {
        op.mfma_macs_per_image = op.macs_per_image * (16.0 / 36.0);   // F(2x2,3x3): 16 products per 4 outputs
        // what the kernel ISSUES: every region runs two MFMA column blocks of 16 tile slots (one for a region whose lower half lies below
        // the map in the 64-channel form, kernels_wino.hip `half`), whatever part of its 4 x 8 (8 x 4) tiles the map fills
        const int trY = c.th / 2, trX = 32 / trY;        // tiles per region along y / x
        const int regs_y = (op.Ho + 2 * trY - 1) / (2 * trY), regs_x = (op.Wo + 2 * trX - 1) / (2 * trX);
        double slots = 0;
        for (int ry = 0; ry < regs_y; ++ry) slots += (double)regs_x * ((c.wm == 4 && ry * 2 * trY + trY >= op.Ho) ? 16 : 32);
        op.padded_macs_per_image = slots * 16.0 * L.cin * L.cout;
    } else if (c.pc <= 2) {                            // direct tilings: tiles x pixel blocks of the MFMA's N width
        const int npb = (c.th * c.tw + c.mb - 1) / c.mb;
        const double tiles = (double)((op.Ho + c.th - 1) / c.th) * ((op.Wo + c.tw - 1) / c.tw);
        op.padded_macs_per_image = tiles * npb * c.mb * L.ks * L.ks * (double)L.cin * L.cout;
    }
    else if (fused_first) op.mfma_macs_per_image = op.macs_per_image;              // conv0_0 itself runs on the vector ALU
    h->ops.push_back(op);
    *out_buf = op.out;
    return UKBB_OK;
}

// conv2d_transpose 3x3 s2 + BN + ReLU as a 2x2 sub-pixel conv (kernels.h, tconv_as_conv2x2)
int add_tconv(ukbb_fcn_handle *h, const std::string &lname, int in0, int H, int W, int n_hint, int *out_buf) {
    const int li = h->layer_index.at(lname);
    const HostLayer &L = h->layers[li];
    Op op;
    op.kind = OP_TCONV; op.name = lname; op.layer = li; op.in0 = in0;
    op.H = H; op.W = W; op.Ho = H; op.Wo = W; op.stride = 1; op.pad_y = 1; op.pad_x = 1;
    op.cfg = choose_cfg(lname, 2, 1, L.cin, 0, 4 * L.cout, H, W, n_hint, false, bf16_mode(h));
    if (op.cfg < 0) { set_err("no tiling for transposed conv %s", lname.c_str()); return UKBB_EARCH; }
    ConvConfig c;
    find_cfg(op.cfg, c);
    char key[128];
    const bool bfpk = c.pc == 3 || c.pc == 5 || c.pc == 6;
    const bool paired = c.pc == 6;                    // weight-stationary tilings: virtual channels in the paired block order (wst_pack_order)
    snprintf(key, sizeof key, "%s/pk2x2%s%s_mb%d_kc%d_g%d", L.name.c_str(), bfpk ? "bf16" : "", paired ? "ws" : "", c.mb, c.kc, c.wm * c.cb);
    const std::string bkey = lname + (paired ? "/bias4ws" : "/bias4");
    if (!dev_ptr(h, key)) {
        const int vc = 4 * L.cout;
        std::vector<float> w2((size_t)4 * L.cin * vc), pk(w2.size());
        tconv_as_conv2x2(L.w.data(), L.cin, L.cout, w2.data());
        std::vector<float> b4((size_t)vc);
        for (int ph = 0; ph < 4; ++ph) std::copy(L.b.begin(), L.b.end(), b4.begin() + (size_t)ph * L.cout);
        if (paired) {
            std::vector<float> w2p(w2.size()), b4p(b4.size());
            for (int v = 0; v < vc; ++v) {
                const int src = wst_pack_order(L.cout, v);
                b4p[v] = b4[src];
                for (size_t r = 0; r < (size_t)4 * L.cin; ++r) w2p[r * vc + v] = w2[r * vc + src];
            }
            w2.swap(w2p); b4.swap(b4p);
        }
        if (bfpk) pack_conv_weights_bf16(w2.data(), 2, L.cin, vc, c.wm * c.cb, pk.data());
        else pack_conv_weights(w2.data(), 2, L.cin, vc, c.mb, c.kc, c.wm * c.cb, pk.data());
        int rc = upload(h, key, pk);
        if (rc) return rc;
        rc = upload(h, bkey, b4);
        if (rc) return rc;
    }
    op.wpk = dev_ptr(h, key);
    op.bias = dev_ptr(h, bkey);
    op.out = new_act(h, lname, (size_t)4 * H * W * L.cout, L.cout);
    op.macs_per_image = (double)H * W * 9 * L.cin * L.cout;
    h->ops.push_back(op);
    *out_buf = op.out;
    return UKBB_OK;
}

int build_plan(ukbb_fcn_handle *h, int H, int W, int n_hint) {
    h->ops.clear();
    h->act.clear(); h->act_per_image.clear(); h->act_name.clear(); h->act_ch.clear();
    h->cap_n = 0;
    const ukbb_fcn_arch &a = h->arch;
    char nm[64];
    // encoder (network.py:179-189 / network_ao.py:31-41)
    int cur = -1, ch = 1, cw = 1;
    std::vector<int> level_out(a.n_level), lh(a.n_level), lw(a.n_level), sqg_out(a.n_level, -1);
    Op multi;
    multi.kind = OP_FIRST;                         // becomes OP_SQG_MULTI when the first merged level arrives
    int hh = H, ww = W;
    for (int l = 0; l < a.n_level; ++l) {
        for (int i = 0; i < a.n_block[l]; ++i) {
            snprintf(nm, sizeof nm, "conv%d_%d", l, i);
            const int stride = (l > 0 && i == 0) ? 2 : 1;
            static const bool no_fuse = getenv("UKBB_NO_FUSE_FIRST") != nullptr;    // A/B knob
            // bf16 storage: only if a fused tiling fits conv0_1 at this size (otherwise conv0_0 runs as its own launch, bf16 out)
            const bool can_fuse = !no_fuse && a.n_block[0] >= 2 && a.n_filter[0] == 16 &&
                                  (bf16_mode(h) != 2 || pick_fused_bf_cfg("conv0_1", 3, 1, 16, 0, 16, H, W, 1) >= 0);
            // bf16 storage with the standard 1 -> 16 -> 16 stem: conv0_0 and conv0_1 as ONE launch of kernels_stem.hip
            // (UKBB_NO_FUSE_STEM=1: the r03 form, conv0_0 evaluated in conv0_1's staging)
            const bool stem = a.kind != UKBB_KIND_FCN && bf16_mode(h) == 2 && getenv("UKBB_NO_FUSE_STEM") == nullptr && a.n_block[0] >= 2 &&
                              a.n_filter[0] == 16 && override_cfg("conv0_1") < 0;
            if (l == 0 && i == 0 && stem) continue;
            if (l == 0 && i == 1 && stem) {
                const int l0 = h->layer_index.at("conv0_0"), l1 = h->layer_index.at("conv0_1");
                const HostLayer &L0 = h->layers[l0], &L1 = h->layers[l1];
                if (!dev_ptr(h, "stem/wA0")) {
                    std::vector<float> p0((size_t)64 * 4), dummy((size_t)9 * 64 * 4), p1((size_t)5 * 64 * 4), w0z((size_t)9 * 32 * 16, 0.f);
                    pack_stem_weights(L0.w.data(), p0.data());
                    pack_tail_weights(w0z.data(), L1.w.data(), dummy.data(), p1.data());
                    int rc = upload(h, "stem/wA0", p0);
                    if (rc) return rc;
                    rc = upload(h, "stem/wA1", p1);
                    if (rc) return rc;
                }
                Op op; op.kind = OP_STEM; op.name = "conv0_0+conv0_1"; op.layer = l1;
                op.H = op.Ho = H; op.W = op.Wo = W;
                op.out = new_act(h, "conv0_1", (size_t)H * W * L1.cout, L1.cout);
                op.macs_per_image = (double)H * W * 9 * (L0.cin * L0.cout + L1.cin * L1.cout);
                h->ops.push_back(op);
                cur = op.out;
                continue;
            }
            if (l == 0 && i == 0) {
                if (can_fuse) continue;              // evaluated inside conv0_1's producers
                Op op; op.kind = OP_FIRST; op.name = nm; op.layer = h->layer_index.at(nm);
                op.H = op.Ho = H; op.W = op.Wo = W;
                op.out = new_act(h, nm, (size_t)H * W * a.n_filter[0], a.n_filter[0]);
                op.macs_per_image = (double)H * W * 9 * a.n_filter[0];
                op.mfma_macs_per_image = 0;          // vector ALU kernel
                h->ops.push_back(op);
                cur = op.out;
            } else {
                int out;
                const bool fused = (l == 0 && i == 1 && can_fuse);
                int rc = add_conv(h, nm, cur, -1, 0, hh, ww, stride, n_hint, &out, fused);
                if (rc) return rc;
                if (fused) {
                    h->ops.back().name = "conv0_0+conv0_1";
                    h->ops.back().macs_per_image += (double)H * W * 9 * a.n_filter[0];
                }
                cur = out;
                if (stride == 2) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; }
            }
        }
        level_out[l] = cur; lh[l] = hh; lw[l] = ww;
        h->act_name[cur] = std::string("conv") + std::to_string(l);
        if (a.kind == UKBB_KIND_FCN && l >= 1) {
            // same_dim_l + out0's level-l slice at low resolution (same_dim0 lives inside the head kernel).
            // Emitted right after its level and run on the side stream: it only feeds the head, is
            // memory-bound, and overlaps with the MFMA-bound convs of the deeper levels.
            snprintf(nm, sizeof nm, "same_dim%d", l);
            Op op; op.kind = OP_SQG; op.name = std::string("sqg") + std::to_string(l);
            op.layer = h->layer_index.at(nm); op.in0 = level_out[l];
            op.H = op.Ho = lh[l]; op.W = op.Wo = lw[l]; op.stride = l;
            op.on_side = h->use_side;
            op.out = new_act(h, std::string("g") + std::to_string(l), (size_t)lh[l] * lw[l] * a.fc);
            // algorithmic MACs: the squeeze; the 32->64 projection is out0's work moved to low
            // resolution and is accounted to the head (so the per-layer sums equal Appendix A)
            op.macs_per_image = (double)lh[l] * lw[l] * a.n_filter[l] * a.same_dim;
            op.mfma_macs_per_image = (double)lh[l] * lw[l] * (a.n_filter[l] * a.same_dim + a.same_dim * a.fc);
            sqg_out[l] = op.out;
            // levels 1-4 of the standard filter pyramid go out as ONE launch after level 4 (sqg_multi_kernel)
            static const bool split1 = getenv("UKBB_SQG1_SEPARATE") != nullptr;    // A/B knob: level 1 as a launch of its own
            const bool merge = !h->use_side && a.n_level == 5 && a.n_filter[1] == 32 && a.n_filter[2] == 64 && a.n_filter[3] == 128 && a.n_filter[4] == 256;
            if (merge && l >= 1 && !(split1 && l == 1)) {
                if (multi.kind != OP_SQG_MULTI) { multi = Op(); multi.kind = OP_SQG_MULTI; multi.name = split1 ? "sqg2-4" : "sqg1-4"; multi.macs_per_image = 0; multi.mfma_macs_per_image = 0; }
                multi.mlayer[l - 1] = op.layer; multi.min_[l - 1] = op.in0; multi.mout[l - 1] = op.out;
                multi.mh[l - 1] = lh[l]; multi.mw[l - 1] = lw[l];
                multi.macs_per_image += op.macs_per_image; multi.mfma_macs_per_image += op.mfma_macs_per_image;
                if (l == 4) h->ops.push_back(multi);
            } else {
                h->ops.push_back(op);
            }
        }
    }
    (void)ch; (void)cw;
    if (a.kind == UKBB_KIND_FCN) {
        std::vector<int> &sq = sqg_out;
        Op op; op.kind = OP_HEAD; op.name = "head"; op.in0 = level_out[0];
        op.H = op.Ho = H; op.W = op.Wo = W;
        op.macs_per_image = (double)H * W * (a.n_filter[0] * a.same_dim + a.same_dim * a.n_level * a.fc +
                                             a.fc * a.fc + a.fc * a.n_class);
        // matrix pipe: same_dim0, the level-0 slice of out0, out1 (the logits run on the vector ALU)
        op.mfma_macs_per_image = (double)H * W * (a.n_filter[0] * a.same_dim + a.same_dim * a.fc + a.fc * a.fc);
        for (int l = 1; l < 5; ++l) op.sq[l - 1] = sq[l];
        h->ops.push_back(op);
    } else {
        // decoder (network_ao.py:44-55): transposed conv, concat [skip, up] (two-source conv), convs
        int up = level_out[a.n_level - 1];
        for (int l = a.n_level - 2; l >= 0; --l) {
            snprintf(nm, sizeof nm, "up%d_t", l);
            int t;
            int rc = add_tconv(h, nm, up, lh[l + 1], lw[l + 1], n_hint, &t);
            if (rc) return rc;
            int x = -1;
            // bf16 storage, level 0 with the standard two 16-channel convs: up0_0, up0_1, logits and softmax / argmax as ONE launch
            // (kernels_tail.hip); UKBB_NO_FUSE_TAIL=1 keeps the separate launches (A/B knob)
            const bool no_tail = getenv("UKBB_NO_FUSE_TAIL") != nullptr;      // read at every plan build (tools/check_tail.py switches it inside one process)
            if (l == 0 && a.kind == UKBB_KIND_UNET && bf16_mode(h) == 2 && !no_tail && a.n_block[0] == 2 && a.n_filter[0] == 16 &&
                a.n_class >= 2 && a.n_class <= 4 && override_cfg("up0_0") < 0 && override_cfg("up0_1") < 0) {
                const int l0 = h->layer_index.at("up0_0"), l1 = h->layer_index.at("up0_1");
                const HostLayer &L0 = h->layers[l0], &L1 = h->layers[l1];
                if (!dev_ptr(h, "tail/wA0")) {
                    std::vector<float> p0((size_t)9 * 64 * 4), p1((size_t)5 * 64 * 4);
                    pack_tail_weights(L0.w.data(), L1.w.data(), p0.data(), p1.data());
                    int rc = upload(h, "tail/wA0", p0);
                    if (rc) return rc;
                    rc = upload(h, "tail/wA1", p1);
                    if (rc) return rc;
                }
                Op op; op.kind = OP_TAIL; op.name = "up0_0+up0_1+logits"; op.layer = l0; op.in0 = level_out[0]; op.in1 = t;
                op.H = op.Ho = lh[0]; op.W = op.Wo = lw[0];
                op.macs_per_image = (double)lh[0] * lw[0] * (9.0 * L0.cin * L0.cout + 9.0 * L1.cin * L1.cout + (double)a.n_filter[0] * a.n_class);
                h->ops.push_back(op);
                h->feat_buf = -1;
                up = -1;
                continue;
            }
            for (int i = 0; i < a.n_block[l]; ++i) {
                snprintf(nm, sizeof nm, "up%d_%d", l, i);
                static const bool no_fuse_lg = getenv("UKBB_NO_FUSE_LOGITS") != nullptr;    // A/B knob
                // bf16 storage: logits + softmax / argmax ride in the epilogue of the very last conv (its output is never stored)
                const bool flg = a.kind == UKBB_KIND_UNET && bf16_mode(h) == 2 && !no_fuse_lg && l == 0 && i == a.n_block[0] - 1 &&
                                 i > 0 && a.n_filter[0] == 16 && pick_fused_bf_cfg(nm, 3, 1, 16, 0, 16, lh[0], lw[0], 2) >= 0;
                rc = (i == 0) ? add_conv(h, nm, level_out[l], t, a.n_filter[l], lh[l], lw[l], 1, n_hint, &x)
                              : add_conv(h, nm, x, -1, 0, lh[l], lw[l], 1, n_hint, &x, false, flg);
                if (rc) return rc;
            }
            up = x;
            h->act_name[up] = std::string("up") + std::to_string(l);
        }
        h->feat_buf = up;                              // net['conv0_up']: what UNet_LSTM_Model feeds the LSTM (:343-347)
        if (a.kind == UKBB_KIND_UNET && h->ops.back().kind == OP_TAIL) {
            // logits, softmax / argmax are part of the fused tail launch
        } else if (a.kind == UKBB_KIND_UNET && h->ops.back().kind == OP_CONV && h->ops.back().fused_logits) {
            Op &last = h->ops.back();
            last.name += "+logits";
            last.macs_per_image += (double)H * W * a.n_filter[0] * a.n_class;
            h->act_name[up] = "";                      // net['conv0_up'] does not exist in HBM in this plan
        } else if (a.kind == UKBB_KIND_UNET) {
            Op op; op.kind = OP_LOGITS; op.name = "logits"; op.layer = h->layer_index.at("logits"); op.in0 = up;
            op.H = op.Ho = H; op.W = op.Wo = W;
            op.macs_per_image = (double)H * W * a.n_filter[0] * a.n_class;
            h->ops.push_back(op);
        } else {
            // ConvLSTM: region shape of the fused gate-conv / cell kernel, chosen once per plan (both shapes give identical bits);
            // packed filters: the x rows of both directions as ONE 128-channel conv (groups = directions), the h rows per direction
            // (read per plan build, like UKBB_NO_FUSE_TAIL: the A/B knob can be toggled inside one process by re-planning)
            h->lstm_bf_wino = getenv("UKBB_LSTM_BF16_WINOGRAD") != nullptr;
            // r06 experiment, measured and NOT the default: UKBB_LSTM_BF16_UNHOIST=1 makes the bf16 time steps re-multiply x instead of reading the hoisted gx
            // (320 -> 224 bytes per pixel and step).  100-frame 256x256 cine, three alternating rounds + rocprofv3 (profiles/r06_ab_lstm_unhoist.txt): step
            // 354.8 -> 372.1 us, x pass 754 -> 636 us, cine 8.17 -> 8.31 ms: the step is not bound by its bytes alone -- the second chunk's staging and MFMAs
            // cost more issue time than the gx loads they replace.  The hoisted form (r05) stays.
            h->lstm_bf_hoist = getenv("UKBB_LSTM_BF16_UNHOIST") == nullptr;
            const int cfg = choose_cfg("lstm_fw", 3, 1, a.n_filter[0], a.same_dim, 4 * a.same_dim, H, W, n_hint, false, false);
            ConvConfig c;
            const bool have24 = cfg >= 0 && !find_cfg(cfg, c) && is_wino24(c) && c.wm == 4 && (c.tw == 32 || c.tw == 16);
            // the bf16 plan's time steps run on launch_lstm_ws (kernels_ws.hip) and never touch the F(2x4) kernel: only the fp32 plan and the
            // bf16-storage Winograd A/B form need that tiling
            if (!have24 && (bf16_mode(h) != 2 || h->lstm_bf_wino)) {
                set_err("the ConvLSTM needs the Winograd F(2x4) kernel (unset UKBB_NO_WINOGRAD / UKBB_NO_WINOGRAD24 / UKBB_CONV_CFG overrides)");
                return UKBB_EARCH;
            }
            h->lstm_tile_cols = have24 ? c.tw : 32;
            if (const char *e = getenv("UKBB_LSTM_TILE_COLS")) { const int v = atoi(e); if (v == 16 || v == 32) h->lstm_tile_cols = v; }   // A/B knob (identical bits)
            {
                const HostLayer &B = h->layers[h->layer_index.at("lstm_bw")];
                bool z = getenv("UKBB_LSTM_RUN_ZERO_CELL") == nullptr;      // knob: run the zero cell anyway (tests compare both ways)
                for (size_t i = 0; z && i < B.w.size(); ++i) z = B.w[i] == 0.f;
                for (size_t i = 0; z && i < B.b.size(); ++i) z = B.b[i] == 0.f;
                h->lstm_bw_zero = z;
            }
            if (!dev_ptr(h, "lstm/wx")) {
                const size_t per = (size_t)24 * 16 * 64;
                std::vector<float> wx(2 * per), bx(2 * 64), wh(per);
                int d = 0;
                for (const char *nm2 : {"lstm_fw", "lstm_bw"}) {
                    const HostLayer &L = h->layers[h->layer_index.at(nm2)];
                    if (L.cin != 32 || L.cout != 64 || L.ks != 3) { set_err("ConvLSTM gate kernel must be 3x3x(16+16)x64"); return UKBB_EARCH; }
                    pack_lstm_gate_weights(L.w.data(), L.cin, 0, L.b.data(), wx.data() + d * per, bx.data() + d * 64);
                    pack_lstm_gate_weights(L.w.data(), L.cin, a.n_filter[0], nullptr, wh.data(), nullptr);
                    int rc = upload(h, std::string(nm2) + "/wh", wh);
                    if (rc) return rc;
                    ++d;
                }
                int rc = upload(h, "lstm/wx", wx);
                if (rc) return rc;
                rc = upload(h, "lstm/bx", bx);
                if (rc) return rc;
                // bf16 form (kernels_ws.hip, ws_main LS): direct 3x3 conv on the bf16 matrix instruction, its own channel order
                const size_t perb = (size_t)9 * 16 * 64 / 2;          // dwords
                std::vector<float> wxb(2 * perb), bxb(2 * 64), whb(perb);
                d = 0;
                for (const char *nm2 : {"lstm_fw", "lstm_bw"}) {
                    const HostLayer &L = h->layers[h->layer_index.at(nm2)];
                    pack_lstm_gate_weights_bf16(L.w.data(), L.cin, 0, L.b.data(), wxb.data() + d * perb, bxb.data() + d * 64);
                    pack_lstm_gate_weights_bf16(L.w.data(), L.cin, a.n_filter[0], nullptr, whb.data(), nullptr);
                    rc = upload(h, std::string(nm2) + "/wh_bf16", whb);
                    if (rc) return rc;
                    std::vector<float> wxhb(2 * perb);               // r06: both halves as one two-chunk filter (un-hoisted time steps, ls_mode 3)
                    pack_lstm_gate_weights_bf16_xh(L.w.data(), nullptr, wxhb.data(), nullptr);
                    rc = upload(h, std::string(nm2) + "/wxh_bf16", wxhb);
                    if (rc) return rc;
                    ++d;
                }
                rc = upload(h, "lstm/wx_bf16", wxb);
                if (rc) return rc;
                rc = upload(h, "lstm/bx_bf16", bxb);
                if (rc) return rc;
            }
        }
    }
    h->plan_h = H; h->plan_w = W; h->plan_small = n_hint <= SMALL_BATCH; h->plan_n = n_hint;
    h->plan_bfio = bf16_mode(h) == 2;
    // r06: the levels >= k of a plan (U-Net: conv{k}_0 .. up{k}_1) run as two half-batch chains on two streams (run_plan), so that one half's
    // fill / drain / serial chains hide under the other half's body.  Measured at N = 100 x 256x256 (profiles/r06_split_levels.txt and
    // r06_split_after_fix.txt), labels bit-identical to the unsplit plan in every run: fp32 U-Net 4.02 -> 3.86 ms per forward with k = 1
    // (+4 %; k = 2: 3.88, k = 3: 3.92, k = 4: no change), bf16-storage U-Net 1.042 -> 1.042 (nothing to hide once the walkers fill the chip),
    // FCN 0.5-1 % slower.  So: ON from level 1 for a UKBB_KIND_UNET plan in fp32, off everywhere else; UKBB_SPLIT_FROM=k overrides (0 = off).
    // (While this was first tried the 300-case bf16 sweep met sporadic wrong tiles with it; that was the wide-store hazard of kernels_ws.hip,
    // store_b128_sofs there and profiles/r06_notes.md section 10 -- fixed, and the sweep is clean with the split forced on.)
    h->split_first = -1; h->split_last = -2;
    {
        const char *e = getenv("UKBB_SPLIT_FROM");
        const int k = e ? atoi(e) : (a.kind == UKBB_KIND_UNET && bf16_mode(h) == 0) ? 1 : 0;
        if (k >= 1 && k < a.n_level) {
            // U-Net: conv{k}_0 .. up{k}_1; FCN (no decoder): conv{k}_0 .. the last encoder conv (the squeeze launches and the head follow unsplit)
            const std::string c0 = "conv" + std::to_string(k) + "_0";
            const std::string u0 = a.kind == UKBB_KIND_FCN ? "conv" + std::to_string(a.n_level - 1) + "_" : "up" + std::to_string(k) + "_";
            for (size_t i = 0; i < h->ops.size(); ++i) {
                if (h->split_first < 0 && h->ops[i].name.compare(0, c0.size(), c0) == 0) h->split_first = (int)i;
                if (h->ops[i].name.compare(0, u0.size(), u0) == 0 && h->ops[i].kind != OP_TAIL) h->split_last = (int)i;
            }
            if (h->split_first < 0 || h->split_last < h->split_first) { h->split_first = -1; h->split_last = -2; }
        }
        h->debug_first_op = 0; h->debug_last_op = 1 << 30;
        if (const char *o = getenv("UKBB_DEBUG_OPS")) { int f = 0, l = 1 << 30; if (sscanf(o, "%d,%d", &f, &l) >= 1) { h->debug_first_op = f; h->debug_last_op = l; } }
        if (const char *o = getenv("UKBB_SPLIT_OP")) {     // debugging: ONLY op i runs as two half-batch launches on two streams
            const int i = atoi(o);
            if (i >= 0 && i < (int)h->ops.size()) { h->split_first = i; h->split_last = i; }
        }
    }
    // events
    for (auto e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->t_sum.assign(h->ops.size(), 0.0);
    h->t_cnt.assign(h->ops.size(), 0);
    h->ev_pending = false;
    return UKBB_OK;
}

int ensure_capacity(ukbb_fcn_handle *h, int n) {
    if (n <= h->cap_n) return UKBB_OK;
    for (size_t i = 0; i < h->act.size(); ++i)
        HIP_TRY(h->act[i]->ensure(h->act_per_image[i] * (size_t)n), UKBB_ENOMEM);
    h->cap_n = n;
    return UKBB_OK;
}

int check_shape(int n, int H, int W) {
    if (n < 1 || H < 16 || W < 16 || (H % 16) || (W % 16)) {
        set_err("invalid batch shape n=%d h=%d w=%d: h and w must be positive multiples of 16 "
                "(the reference pads to that, common/deploy_network.py:97)", n, H, W);
        return UKBB_EINVAL;
    }
    if ((long long)n * H * W > (1ll << 31) - 1) { set_err("batch too large for 32-bit pixel indexing"); return UKBB_EINVAL; }
    return UKBB_OK;
}

int prepare(ukbb_fcn_handle *h, int n, int H, int W) {
    int rc = check_shape(n, H, W);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device), UKBB_EDEVICE);
    if (n > h->max_n) h->max_n = n;
    // the finer siblings of the small-batch plan are chosen from the work items at the largest batch seen (finer_sibling): while the
    // handle stays in the small regime a larger batch re-plans (a handle first used at N = 1 must not keep halved items at N = 16)
    if (H != h->plan_h || W != h->plan_w || (h->max_n <= SMALL_BATCH) != h->plan_small || (h->plan_small && h->max_n > h->plan_n)) {
        HIP_TRY(hipDeviceSynchronize(), UKBB_EDEVICE);
        rc = build_plan(h, H, W, h->max_n);
        if (rc) { h->plan_h = h->plan_w = 0; return rc; }
    }
    if (n > h->cap_n) {
        HIP_TRY(hipDeviceSynchronize(), UKBB_EDEVICE);
        rc = ensure_capacity(h, n);
        if (rc) return rc;
    }
    return UKBB_OK;
}

int collect_events(ukbb_fcn_handle *h) {
    if (!h->ev_pending) return UKBB_OK;
    HIP_TRY(hipEventSynchronize(h->ev[h->timing_only >= 0 ? 2 * h->timing_only + 1 : h->ev.size() - 1]), UKBB_EDEVICE);
    for (size_t i = 0; i < h->ops.size(); ++i) {
        if (h->timing_only >= 0 && (int)i != h->timing_only) continue;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]), UKBB_EDEVICE);
        h->t_sum[i] += ms;
        h->t_cnt[i] += 1;
    }
    h->ev_pending = false;
    return UKBB_OK;
}

int run_plan(ukbb_fcn_handle *h, const float *image, int n, float *logits, float *prob, int32_t *pred,
             hipStream_t s) {
    const ukbb_fcn_arch &a = h->arch;
    if (h->timing) {
        int rc = collect_events(h);
        if (rc) return rc;
        if (h->ev.size() != 2 * h->ops.size()) {
            for (auto e : h->ev) (void)hipEventDestroy(e);
            h->ev.assign(2 * h->ops.size(), nullptr);
            for (auto &e : h->ev) HIP_TRY(hipEventCreate(&e), UKBB_EDEVICE);
        }
    }
    hipStream_t s_main = s;
    bool forked = false;
    // One launch of op i over images [n0, n0 + n) of the batch on stream s (the whole batch everywhere except inside a split range, below).
    const size_t act_esz = h->plan_bfio ? 2 : 4;             // bytes per stored activation element
    auto launch_one = [&](size_t i, int n0, int n, hipStream_t s, hipError_t &e) -> int {
        const Op &op = h->ops[i];
        auto actp = [&](int id) -> float * {
            return reinterpret_cast<float *>(reinterpret_cast<char *>(h->act[id]->p) + (size_t)n0 * h->act_per_image[id] * act_esz);
        };
        e = hipSuccess;
        switch (op.kind) {
            case OP_FIRST: {
                const HostLayer &L = h->layers[op.layer];
                FirstArgs fa{image, dev_ptr(h, L.name + "/w"), dev_ptr(h, L.name + "/bias"), actp(op.out),
                             n, op.H, op.W, L.cout, h->plan_bfio ? 1 : 0};
                e = launch_first(fa, s);
                break;
            }
            case OP_CONV: {
                const HostLayer &L = h->layers[op.layer];
                ConvConfig c;
                find_cfg(op.cfg, c);
                ConvArgs ca{};
                ca.in0 = op.fused_first ? image : actp(op.in0);
                if (op.fused_first) { ca.first_w = dev_ptr(h, "conv0_0/w"); ca.first_b = dev_ptr(h, "conv0_0/bias"); }
                ca.in1 = op.in1 >= 0 ? actp(op.in1) : nullptr;
                ca.C1 = op.in1 >= 0 ? (int)(h->act_per_image[op.in1] / ((size_t)op.H * op.W)) : 0;
                ca.C0 = L.cin - ca.C1;
                ca.wpk = op.wpk; ca.bias = op.bias; ca.out = actp(op.out);
                ca.N = n; ca.H = op.H; ca.W = op.W; ca.Ho = op.Ho; ca.Wo = op.Wo; ca.Cout = L.cout;
                if (c.pc == 5 || c.pc == 6) { ca.Cout = round_up(L.cout, 32); ca.cout_store = L.cout; }
                if (op.fused_logits) {
                    ca.lg_w = dev_ptr(h, "logits/w"); ca.lg_b = dev_ptr(h, "logits/bias");
                    ca.lg_logits = logits; ca.lg_prob = prob; ca.lg_pred = pred; ca.lg_ncls = a.n_class;
                }
                ca.pad_y = op.pad_y; ca.pad_x = op.pad_x;
                ca.tiles_y = (op.Ho + c.th - 1) / c.th; ca.tiles_x = (op.Wo + c.tw - 1) / c.tw;
                ca.relu = L.relu ? 1 : 0;
                e = launch_conv(op.cfg, ca, s);
                break;
            }
            case OP_SQG: {
                const HostLayer &L = h->layers[op.layer];
                const std::string ls = std::to_string(op.stride);
                SqgArgs sa{};
                sa.x = actp(op.in0);
                sa.w_s = dev_ptr(h, "sqg" + ls + "/w_s"); sa.b_s = dev_ptr(h, L.name + "/bias");
                sa.w_g = dev_ptr(h, "sqg" + ls + "/w_g");
                sa.out = actp(op.out);
                sa.npix = (long long)n * op.H * op.W; sa.cin = L.cin;
                e = launch_sqg(sa, s);
                break;
            }
            case OP_SQG_MULTI: {
                SqgArgs sa[4];
                for (int j = 0; j < 4; ++j) {
                    sa[j] = SqgArgs{};
                    sa[j].cin = 32 << j;
                    if (op.mlayer[j] < 0) continue;              // level handled by a launch of its own: npix = 0 -> no blocks
                    const HostLayer &L = h->layers[op.mlayer[j]];
                    const std::string ls = std::to_string(j + 1);
                    sa[j].x = actp(op.min_[j]);
                    sa[j].w_s = dev_ptr(h, "sqg" + ls + "/w_s"); sa[j].b_s = dev_ptr(h, L.name + "/bias");
                    sa[j].w_g = dev_ptr(h, "sqg" + ls + "/w_g");
                    sa[j].out = actp(op.mout[j]);
                    sa[j].npix = (long long)n * op.mh[j] * op.mw[j]; sa[j].cin = L.cin;
                }
                e = launch_sqg_multi(sa, s);
                break;
            }
            case OP_HEAD: {
                HeadArgs ha{};
                ha.conv0 = actp(op.in0);
                for (int l = 0; l < 4; ++l) ha.G[l] = actp(op.sq[l]);
                ha.w_s0 = dev_ptr(h, "head/w_s0"); ha.b_s0 = dev_ptr(h, "same_dim0/bias");
                ha.w_o0 = dev_ptr(h, "head/w_o0"); ha.b_o0 = dev_ptr(h, "out0/bias");
                ha.w_o1 = dev_ptr(h, "head/w_o1"); ha.b_o1 = dev_ptr(h, "out1/bias");
                ha.w_o1x3 = dev_ptr(h, "head/w_o1x3"); ha.w_o0x3 = dev_ptr(h, "head/w_o0x3");
                ha.w_lg = dev_ptr(h, "head/w_lg"); ha.b_lg = dev_ptr(h, "logits/bias");
                ha.logits = logits; ha.prob = prob; ha.pred = pred;
                ha.N = n; ha.H = op.H; ha.W = op.W; ha.n_class = a.n_class;
                ha.x3 = h->precision == UKBB_PREC_F32X3;
                e = launch_head(ha, s);
                break;
            }
            case OP_TCONV: {
                const HostLayer &L = h->layers[op.layer];
                ConvConfig c;
                find_cfg(op.cfg, c);
                ConvArgs ca{};
                ca.in0 = actp(op.in0); ca.in1 = nullptr; ca.C0 = L.cin; ca.C1 = 0;
                ca.wpk = op.wpk; ca.bias = op.bias; ca.out = actp(op.out);
                ca.N = n; ca.H = op.H; ca.W = op.W; ca.Ho = op.Ho; ca.Wo = op.Wo; ca.Cout = 4 * L.cout;
                ca.pad_y = 1; ca.pad_x = 1;
                ca.tiles_y = (op.Ho + c.th - 1) / c.th; ca.tiles_x = (op.Wo + c.tw - 1) / c.tw;
                ca.relu = 1; ca.up2 = L.cout;
                e = launch_conv(op.cfg, ca, s);
                break;
            }
            case OP_STEM: {
                StemArgs sa{};
                sa.image = image; sa.wA0 = dev_ptr(h, "stem/wA0"); sa.wA1 = dev_ptr(h, "stem/wA1");
                sa.b0 = dev_ptr(h, "conv0_0/bias"); sa.b1 = dev_ptr(h, "conv0_1/bias");
                sa.out = actp(op.out); sa.N = n; sa.H = op.H; sa.W = op.W;
                e = launch_unet_stem(sa, s);
                break;
            }
            case OP_TAIL: {
                TailArgs ta{};
                ta.in0 = actp(op.in0); ta.in1 = actp(op.in1);
                ta.wA0 = dev_ptr(h, "tail/wA0"); ta.wA1 = dev_ptr(h, "tail/wA1");
                ta.b0 = dev_ptr(h, "up0_0/bias"); ta.b1 = dev_ptr(h, "up0_1/bias");
                ta.lg_w = dev_ptr(h, "logits/w"); ta.lg_b = dev_ptr(h, "logits/bias");
                ta.logits = logits; ta.prob = prob; ta.pred = pred;
                ta.N = n; ta.H = op.H; ta.W = op.W; ta.ncls = a.n_class;
                e = launch_unet_tail(ta, s);
                break;
            }
            case OP_LOGITS: {
                const HostLayer &L = h->layers[op.layer];
                LogitsArgs la{};
                la.in = actp(op.in0); la.w = dev_ptr(h, "logits/w"); la.bias = dev_ptr(h, "logits/bias");
                la.logits = logits; la.prob = prob; la.pred = pred;
                la.npix = (int64_t)n * op.H * op.W; la.C = L.cin; la.n_class = a.n_class;
                la.in_bf16 = h->plan_bfio ? 1 : 0;
                e = launch_logits(la, s);
                break;
            }
            default:
                set_err("op kind %d not implemented", (int)op.kind);
                return UKBB_EARCH;
        }
        return UKBB_OK;
    };
    // UKBB_SPLIT_FROM (plan build): the ops of levels >= k -- conv{k}_0 .. up{k}_1, the launches whose fill / drain and serial chains
    // are the largest part of their time -- run as TWO half-batch chains on two streams, enqueued interleaved, joined before the next op
    const int sp0 = h->split_first, sp1 = h->split_last;
    // debugging aids (r06, tools/two_stream_bisect.py): run only the ops [first, last] of the plan (whatever they read was left by an earlier full forward)
    const int dbg_first = h->debug_first_op, dbg_last = h->debug_last_op;
    const char *poison_env = getenv("UKBB_DEBUG_POISON_LDS");            // hex pattern written to every CU's LDS in front of every launch
    const bool poison_on = poison_env != nullptr;
    const bool fence_on = getenv("UKBB_DEBUG_FENCE_KERNEL") != nullptr;
    const bool sync_on = getenv("UKBB_DEBUG_SYNC_EVERY_OP") != nullptr;
    const uint32_t poison_pat = poison_on ? (uint32_t)strtoul(poison_env, nullptr, 16) : 0u;
    for (size_t i = 0; i < h->ops.size(); ++i) {
        const Op &op = h->ops[i];
        s = s_main;
        if ((int)i < dbg_first || (int)i > dbg_last) continue;
        if ((int)i == sp0 && sp1 >= sp0 && n >= 8 && !h->timing) {      // per-kernel timing (set_timing) measures whole-batch launches
            if (!h->side2) {
                HIP_TRY(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking), UKBB_EDEVICE);
                HIP_TRY(hipEventCreateWithFlags(&h->ev_split_fork, hipEventDisableTiming), UKBB_EDEVICE);
                HIP_TRY(hipEventCreateWithFlags(&h->ev_split_join, hipEventDisableTiming), UKBB_EDEVICE);
            }
            HIP_TRY(hipEventRecord(h->ev_split_fork, s_main), UKBB_EDEVICE);
            HIP_TRY(hipStreamWaitEvent(h->side2, h->ev_split_fork, 0), UKBB_EDEVICE);
            const int nA = (n + 1) / 2;
            for (int j = sp0; j <= sp1; ++j) {
                hipError_t e;
                int rc = launch_one(j, 0, nA, s_main, e);
                if (rc) return rc;
                if (e == hipSuccess) rc = launch_one(j, nA, n - nA, h->side2, e);
                if (rc) return rc;
                if (e != hipSuccess) { set_err("launch of %s (split) failed: %s", h->ops[j].name.c_str(), hipGetErrorString(e)); return UKBB_EDEVICE; }
            }
            HIP_TRY(hipEventRecord(h->ev_split_join, h->side2), UKBB_EDEVICE);
            HIP_TRY(hipStreamWaitEvent(s_main, h->ev_split_join, 0), UKBB_EDEVICE);
            i = (size_t)sp1;
            continue;
        }
        if (op.on_side) {                              // fork: side stream waits for everything issued so far
            if (!h->side) {
                HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking), UKBB_EDEVICE);
                h->ev_fork.assign(UKBB_FCN_MAX_LEVEL, nullptr);
                h->ev_join.assign(UKBB_FCN_MAX_LEVEL, nullptr);
                for (auto &e : h->ev_fork) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), UKBB_EDEVICE);
                for (auto &e : h->ev_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), UKBB_EDEVICE);
            }
            HIP_TRY(hipEventRecord(h->ev_fork[op.stride], s_main), UKBB_EDEVICE);
            HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork[op.stride], 0), UKBB_EDEVICE);
            s = h->side;
            forked = true;
        }
        if (op.kind == OP_HEAD && forked) {            // join: the head needs every side-stream result
            for (const Op &o2 : h->ops)
                if (o2.on_side) HIP_TRY(hipStreamWaitEvent(s_main, h->ev_join[o2.stride], 0), UKBB_EDEVICE);
        }
        const bool timed = h->timing && (h->timing_only < 0 || h->timing_only == (int)i);
        if (timed) HIP_TRY(hipEventRecord(h->ev[2 * i], s), UKBB_EDEVICE);
        hipError_t e = hipSuccess;
        if (poison_on) (void)ukbb_fcn_debug_poison_lds(poison_pat, s);      // debugging aid: a kernel that reads LDS it never wrote now reads this pattern
        {
            const int rc = launch_one(i, 0, n, s, e);
            if (rc) return rc;
        }
        if (e != hipSuccess) { set_err("launch of %s failed: %s", op.name.c_str(), hipGetErrorString(e)); return UKBB_EDEVICE; }
        if (fence_on) (void)ukbb_fcn_debug_fence_kernel(s);                  // debugging aid: an explicit system-scope fence launch behind every op
        if (sync_on) (void)hipStreamSynchronize(s);                           // debugging aid: the host waits for every op before it enqueues the next
        if (timed) HIP_TRY(hipEventRecord(h->ev[2 * i + 1], s), UKBB_EDEVICE);
        if (op.on_side) HIP_TRY(hipEventRecord(h->ev_join[op.stride], h->side), UKBB_EDEVICE);
    }
    if (h->timing) h->ev_pending = true;
    h->last_n = n;
    return UKBB_OK;
}

}  // namespace

// =============================== C ABI ===========================================
extern "C" {

int ukbb_fcn_abi_version(void) { return UKBB_FCN_ABI_VERSION; }

// debugging aid, not in the public header (tools/two_stream_bisect.py): forwards launch only the ops [first, last] of the plan from now on
int ukbb_fcn_debug_set_ops(ukbb_fcn_handle *h, int first, int last) {
    if (!h) return UKBB_EINVAL;
    h->debug_first_op = first; h->debug_last_op = last;
    return UKBB_OK;
}

const char *ukbb_fcn_last_error(void) { return g_err.c_str(); }

size_t ukbb_fcn_weight_count(const ukbb_fcn_arch *arch) {
    if (!arch) return 0;
    std::vector<Spec> specs;
    if (!arch_specs(*arch, specs)) return 0;
    size_t n = 0;
    for (auto &s : specs) n += spec_floats(s);
    return n;
}

ukbb_fcn_handle *ukbb_fcn_create(const ukbb_fcn_arch *arch, const float *weights, size_t n_floats, int device) {
    if (!arch || !weights) { set_err("create: NULL argument"); return nullptr; }
    std::vector<Spec> specs;
    if (!arch_specs(*arch, specs)) { set_err("create: malformed architecture descriptor"); return nullptr; }
    std::string why;
    if (!supported(*arch, why)) { set_err("create: unsupported architecture: %s", why.c_str()); return nullptr; }
    size_t want = 0;
    for (auto &s : specs) want += spec_floats(s);
    if (want != n_floats) { set_err("create: expected %zu weight floats, got %zu", want, n_floats); return nullptr; }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_err("create: no HIP device visible (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_err("create: device %d out of range (0..%d)", device, ndev - 1); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_err("create: hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err("create: device %d is %s; kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { set_err("create: hipSetDevice failed"); return nullptr; }

    std::unique_ptr<ukbb_fcn_handle> h(new ukbb_fcn_handle);
    h->arch = *arch;
    h->device = device;
    // r01 measurement: running sqg_l concurrently with the deeper convs made the step 9 % SLOWER
    // (1.88 vs 1.72 ms: the sqg waves take SIMD slots and L2 bandwidth from the MFMA-bound persistent
    // conv kernels), so the fork/join path is off unless UKBB_SIDE_STREAM is set.
    h->use_side = getenv("UKBB_SIDE_STREAM") != nullptr;

    // ---- fold BN (fp32; same op order as weights.py fold_bn) --------------------------
    const float *p = weights;
    for (auto &s : specs) {
        HostLayer L;
        L.name = s.name; L.ks = s.ks; L.cin = s.cin; L.cout = s.cout; L.transposed = s.transposed;
        L.relu = s.bn;
        const size_t nk = (size_t)s.ks * s.ks * s.cin * s.cout;
        const float *k = p; p += nk;
        std::vector<float> scale(s.cout, 1.f);
        L.b.assign(s.cout, 0.f);
        if (s.bn) {
            const float *gamma = p, *beta = p + s.cout, *mean = p + 2 * s.cout, *var = p + 3 * s.cout;
            p += 4 * (size_t)s.cout;
            for (int c = 0; c < s.cout; ++c) {
                const volatile float sc = gamma[c] / sqrtf(var[c] + BN_EPS);
                const volatile float ms = mean[c] * sc;            // volatile: no fma contraction
                scale[c] = sc;
                L.b[c] = beta[c] - ms;
            }
        }
        if (s.bias) { for (int c = 0; c < s.cout; ++c) L.b[c] = p[c]; p += s.cout; }
        L.w.resize(nk);
        if (!s.transposed) {
            for (size_t i = 0; i < nk; ++i) L.w[i] = k[i] * scale[i % s.cout];
        } else {
            // TF transposed filter [kh][kw][Cout][Cin] -> [kh][kw][Cin][Cout]
            for (int t = 0; t < s.ks * s.ks; ++t)
                for (int co = 0; co < s.cout; ++co)
                    for (int ci = 0; ci < s.cin; ++ci)
                        L.w[((size_t)t * s.cin + ci) * s.cout + co] = k[((size_t)t * s.cout + co) * s.cin + ci] * scale[co];
        }
        h->layer_index[L.name] = (int)h->layers.size();
        h->layers.push_back(std::move(L));
    }

    // ---- upload biases and the non-MFMA weights ------------------------------------------
    for (auto &L : h->layers)
        if (upload(h.get(), L.name + "/bias", L.b)) return nullptr;
    {
        const HostLayer &L0 = h->layers[h->layer_index.at("conv0_0")];
        if (upload(h.get(), "conv0_0/w", L0.w)) return nullptr;      // [9][16]
    }
    if (arch->kind == UKBB_KIND_UNET_LSTM) {
        const HostLayer &lo = h->layers[h->layer_index.at("lstm_out")];   // [2*NH][n_class]
        if (upload(h.get(), "lstm_out/w", lo.w)) return nullptr;
    }
    if (arch->kind == UKBB_KIND_FCN) {
        const HostLayer &s0 = h->layers[h->layer_index.at("same_dim0")];
        const HostLayer &o0 = h->layers[h->layer_index.at("out0")];
        const HostLayer &o1 = h->layers[h->layer_index.at("out1")];
        const HostLayer &lg = h->layers[h->layer_index.at("logits")];
        std::vector<float> v;
        v.assign(16 * 32, 0.f);                 pack_sq(s0.w.data(), 16, v.data());
        if (upload(h.get(), "head/w_s0", v)) return nullptr;
        v.assign(2 * 4 * 64 * 4, 0.f);          pack_rowmap_32x64(o0.w.data(), 64, v.data());
        if (upload(h.get(), "head/w_o0", v)) return nullptr;
        v.assign(2 * 2 * 4 * 64 * 4, 0.f);
        pack_rowmap_32x64(o1.w.data(), 64, v.data());
        pack_rowmap_32x64(o1.w.data() + 32 * 64, 64, v.data() + 2 * 4 * 64 * 4);
        if (upload(h.get(), "head/w_o1", v)) return nullptr;
        v.assign(3 * 2 * 4 * 64 * 4, 0.f);      pack_head_x3(o1.w.data(), 64, v.data());
        if (upload(h.get(), "head/w_o1x3", v)) return nullptr;
        v.assign(3 * 2 * 2 * 64 * 4, 0.f);      pack_head_x3(o0.w.data(), 32, v.data());
        if (upload(h.get(), "head/w_o0x3", v)) return nullptr;
        v.assign(2 * arch->n_class * 32, 0.f);  pack_head_lg(lg.w.data(), arch->n_class, v.data());
        if (upload(h.get(), "head/w_lg", v)) return nullptr;
        for (int l = 1; l < arch->n_level; ++l) {
            const HostLayer &sl = h->layers[h->layer_index.at("same_dim" + std::to_string(l))];
            v.assign((size_t)sl.cin * 32, 0.f);  pack_sq(sl.w.data(), sl.cin, v.data());
            if (upload(h.get(), "sqg" + std::to_string(l) + "/w_s", v)) return nullptr;
            v.assign(2 * 4 * 64 * 4, 0.f);       pack_rowmap_32x64(o0.w.data() + (size_t)32 * l * 64, 64, v.data());
            if (upload(h.get(), "sqg" + std::to_string(l) + "/w_g", v)) return nullptr;
        }
    } else if (arch->kind == UKBB_KIND_UNET) {
        const HostLayer &lg = h->layers[h->layer_index.at("logits")];
        if (upload(h.get(), "logits/w", lg.w)) return nullptr;
    }
    return h.release();
}

void ukbb_fcn_destroy(ukbb_fcn_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    delete h;
}

int ukbb_fcn_reserve(ukbb_fcn_handle *h, int n, int height, int width) {
    if (!h) { set_err("reserve: NULL handle"); return UKBB_EINVAL; }
    return prepare(h, n, height, width);
}

int ukbb_fcn_forward(ukbb_fcn_handle *h, const float *image, int n, int height, int width,
                     float *logits, float *prob, int32_t *pred, void *stream) {
    if (!h || !image) { set_err("forward: NULL argument"); return UKBB_EINVAL; }
    if (h->arch.kind == UKBB_KIND_UNET_LSTM) { set_err("forward: UNet-LSTM models take sequences: use ukbb_fcn_forward_seq / ukbb_fcn_forward_cine"); return UKBB_EINVAL; }
    int rc = prepare(h, n, height, width);
    if (rc) return rc;
    return run_plan(h, image, n, logits, prob, pred, static_cast<hipStream_t>(stream));
}

int ukbb_fcn_forward_host(ukbb_fcn_handle *h, const float *image, int n, int height, int width,
                          float *logits, float *prob, int32_t *pred) {
    if (!h || !image) { set_err("forward_host: NULL argument"); return UKBB_EINVAL; }
    if (h->arch.kind == UKBB_KIND_UNET_LSTM) { set_err("forward_host: UNet-LSTM models take sequences: use ukbb_fcn_forward_seq / ukbb_fcn_forward_cine"); return UKBB_EINVAL; }
    int rc = prepare(h, n, height, width);
    if (rc) return rc;
    const size_t npix = (size_t)n * height * width, ncls = h->arch.n_class;
    HIP_TRY(h->io_image.ensure(npix), UKBB_ENOMEM);
    if (logits) HIP_TRY(h->io_logits.ensure(npix * ncls), UKBB_ENOMEM);
    if (prob) HIP_TRY(h->io_prob.ensure(npix * ncls), UKBB_ENOMEM);
    if (pred) HIP_TRY(h->io_pred.ensure(npix), UKBB_ENOMEM);
    HIP_TRY(hipMemcpyAsync(h->io_image.p, image, npix * sizeof(float), hipMemcpyHostToDevice, nullptr), UKBB_EDEVICE);
    rc = run_plan(h, h->io_image.p, n, logits ? h->io_logits.p : nullptr, prob ? h->io_prob.p : nullptr,
                  pred ? reinterpret_cast<int32_t *>(h->io_pred.p) : nullptr, nullptr);
    if (rc) return rc;
    if (logits) HIP_TRY(hipMemcpyAsync(logits, h->io_logits.p, npix * ncls * sizeof(float), hipMemcpyDeviceToHost, nullptr), UKBB_EDEVICE);
    if (prob) HIP_TRY(hipMemcpyAsync(prob, h->io_prob.p, npix * ncls * sizeof(float), hipMemcpyDeviceToHost, nullptr), UKBB_EDEVICE);
    if (pred) HIP_TRY(hipMemcpyAsync(pred, h->io_pred.p, npix * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr), UKBB_EDEVICE);
    HIP_TRY(hipStreamSynchronize(nullptr), UKBB_EDEVICE);
    return UKBB_OK;
}

// ---- UNet-LSTM --------------------------------------------------------------------------------------
namespace {

// BiConvLSTM over Wn windows of T steps on NF cached feature frames.  d_map[k*Wn + w] = feature frame of step k of window w.
//   x pass (one launch, both directions): per frame gx = W_x * x + b, and the cell's first step from the zero state (:278,:290) c1, h1;
//   then per direction T - 1 launches of the fused gate-conv (hidden channels only) + cell kernel, every step's hidden map kept
//   ([dir][k][Wn][HW][16]) for the output conv over concat([h_fw, h_bw]) (:305-312), which the caller runs (lstm_out / lstm_tile).
int run_bilstm(ukbb_fcn_handle *h, const float *feat, int NF, const int *d_map, int Wn, int H, int W, hipStream_t s) {
    const ukbb_fcn_arch &a = h->arch;
    const int T = a.fc, NHID = a.same_dim, tc = h->lstm_tile_cols;
    const size_t HW = (size_t)H * W;
    // bf16 plan: the direct-conv bf16 form (kernels_ws.hip) unless UKBB_LSTM_BF16_WINOGRAD=1 asks for the fp32 Winograd arithmetic on bf16 storage (A/B)
    const bool wsf = h->plan_bfio && !h->lstm_bf_wino;
    // r06 experiment (UKBB_LSTM_BF16_UNHOIST=1): the direct-conv bf16 steps read the feature frame (32 bytes per pixel) and multiply it again
    // instead of reading gx (128 bytes per pixel); no gx buffer exists in that form.  Slower by 5 % per step (see build_plan), so not the default.
    const bool unhoist = wsf && !h->lstm_bf_hoist;
    const size_t gxf = wsf ? lstm_ws_gx_elems(H, W) : wino24_lstm_gx_floats(H, W, tc), cf = wsf ? lstm_ws_c_floats(H, W) : wino24_lstm_c_floats(H, W, tc);
    const bool bf = h->plan_bfio;                        // bf16 plan: features, gx and hidden maps are bf16 in HBM
    const size_t esz = bf ? 2 : 4;
    // Scratch of one cine (include/ukbb_fcn.h, forward_cine): gx 2 NF gxf + h1 2 NF HW NHID + hall 2 T Wn HW NHID elements of esz bytes,
    // cell state (2 NF + Wn) cf floats.  DevBuf counts 4-byte units: the bf16 maps take half as many.
    auto units = [esz](size_t elems) { return (elems * esz + 3) / 4; };
    if (!unhoist) HIP_TRY(h->lstm_gx.ensure(units(2 * (size_t)NF * gxf)), UKBB_ENOMEM);
    HIP_TRY(h->lstm_c1.ensure(2 * (size_t)NF * cf), UKBB_ENOMEM);
    HIP_TRY(h->lstm_h1.ensure(units(2 * (size_t)NF * HW * NHID)), UKBB_ENOMEM);
    HIP_TRY(h->lstm_c.ensure((size_t)Wn * cf), UKBB_ENOMEM);
    HIP_TRY(h->lstm_hall.ensure(units(2 * (size_t)T * Wn * HW * NHID)), UKBB_ENOMEM);
    auto at = [esz](float *p, size_t elems) { return reinterpret_cast<float *>(reinterpret_cast<char *>(p) + elems * esz); };
    ConvArgs base{};
    base.ls_bf16 = bf ? 1 : 0;
    base.C0 = a.n_filter[0]; base.C1 = 0;
    base.H = H; base.W = W; base.Ho = H; base.Wo = W;
    base.pad_y = 1; base.pad_x = 1; base.relu = 0;
    base.tiles_y = (H + 7) / 8; base.tiles_x = (W + tc - 1) / tc;
    base.ls_forget_bias = 1.0f;
    { const char *dg = getenv("UKBB_LSTM_DIAG"); base.diag = dg ? atoi(dg) : 0; }      // honoured by diagnostic builds (-DUKBB_DIAG) only
    {   // x pass
        ConvArgs ca = base;
        ca.in0 = feat; ca.N = NF; ca.Cout = 2 * 4 * NHID;
        ca.wpk = dev_ptr(h, wsf ? "lstm/wx_bf16" : "lstm/wx"); ca.bias = dev_ptr(h, wsf ? "lstm/bx_bf16" : "lstm/bx");
        ca.ls_mode = 1; ca.ls_gx = unhoist ? nullptr : h->lstm_gx.p; ca.ls_c_out = h->lstm_c1.p; ca.out = h->lstm_h1.p;
        ca.ls_gx_dir = (long long)NF * gxf; ca.ls_c_dir = (long long)NF * cf; ca.ls_h_dir = (long long)NF * HW * NHID;
        hipError_t e = wsf ? launch_lstm_ws(ca, s) : launch_wino24_lstm(ca, tc, s);
        if (e != hipSuccess) { set_err("ConvLSTM x-pass launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    }
    const size_t kst = (size_t)Wn * HW * NHID;           // one step's hidden maps
    // A zero backward cell (the single-direction head served through the bidirectional layer set): from the zero state i = o = 1/2, j = tanh(0) = 0,
    // so c and h stay exactly 0 at every step -- the x pass above already produced zeros for its first step; the other T - 1 maps are cleared, not computed.
    const int ndir = h->lstm_bw_zero ? 1 : 2;
    if (h->lstm_bw_zero) HIP_TRY(hipMemsetAsync(at(h->lstm_hall.p, (size_t)T * kst), 0, (size_t)T * kst * esz, s), UKBB_EDEVICE);
    for (int dir = 0; dir < ndir; ++dir) {
        float *const hall = at(h->lstm_hall.p, (size_t)dir * T * kst);
        for (int step = 1; step < T; ++step) {
            const int k = dir ? T - 1 - step : step, kprev = dir ? k + 1 : k - 1;
            ConvArgs ca = base;
            ca.N = Wn; ca.Cout = 4 * NHID;
            ca.wpk = dev_ptr(h, wsf ? (dir ? "lstm_bw/wh_bf16" : "lstm_fw/wh_bf16") : (dir ? "lstm_bw/wh" : "lstm_fw/wh")); ca.bias = nullptr;
            ca.ls_mode = 2;
            ca.ls_gx = unhoist ? nullptr : at(h->lstm_gx.p, (size_t)dir * NF * gxf); ca.ls_gx_map = d_map + (size_t)k * Wn;
            const float *hprev; const int *hmap;
            if (step == 1) {                                // previous state = the x pass's per-frame first step
                hprev = at(h->lstm_h1.p, (size_t)dir * NF * HW * NHID); hmap = d_map + (size_t)kprev * Wn;
                ca.ls_c_in = h->lstm_c1.p + (size_t)dir * NF * cf;
            } else {
                hprev = at(hall, (size_t)kprev * kst); hmap = nullptr;
                ca.ls_c_in = h->lstm_c.p;
            }
            ca.in0 = hprev; ca.in0_map = hmap;
            if (unhoist) {                                  // ls_mode 3: source 0 = the step's feature frames (through ls_gx_map), source 1 = the previous hidden maps (through in0_map)
                ca.ls_mode = 3;
                ca.in0 = feat; ca.in1 = hprev; ca.C1 = NHID;
                ca.wpk = dev_ptr(h, dir ? "lstm_bw/wxh_bf16" : "lstm_fw/wxh_bf16");
                ca.bias = dev_ptr(h, "lstm/bx_bf16") + dir * 64;
            }
            ca.ls_c_out = h->lstm_c.p;
            ca.out = at(hall, (size_t)k * kst);
            hipError_t e = wsf ? launch_lstm_ws(ca, s) : launch_wino24_lstm(ca, tc, s);
            if (e != hipSuccess) { set_err("ConvLSTM step launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
        }
    }
    return UKBB_OK;
}

int lstm_common_checks(ukbb_fcn_handle *h, const float *image, const char *what) {
    if (!h || !image) { set_err("%s: NULL argument", what); return UKBB_EINVAL; }
    if (h->arch.kind != UKBB_KIND_UNET_LSTM) { set_err("%s: the model is not a UNet-LSTM", what); return UKBB_EINVAL; }
    return UKBB_OK;
}

}  // namespace

int ukbb_fcn_forward_seq(ukbb_fcn_handle *h, const float *image, int n_seq, int height, int width,
                         float *logits, float *prob, int32_t *pred, void *stream) {
    int rc = lstm_common_checks(h, image, "forward_seq");
    if (rc) return rc;
    const int T = h->arch.fc, C = h->arch.n_class;
    if (n_seq < 1) { set_err("forward_seq: n_seq must be positive"); return UKBB_EINVAL; }
    rc = prepare(h, n_seq * T, height, width);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = run_plan(h, image, n_seq * T, nullptr, nullptr, nullptr, s);       // U-Net features of every frame
    if (rc) return rc;
    const size_t HW = (size_t)height * width;
    // map[k][w] = w*T + k; outputs straight into [N][T] order
    std::vector<int> map((size_t)T * n_seq);
    for (int k = 0; k < T; ++k)
        for (int w = 0; w < n_seq; ++w) map[(size_t)k * n_seq + w] = w * T + k;
    const long long key = ((long long)n_seq << 8) | T;                       // tables depend on (n_seq, T) only
    if (h->lstm_aux_key != key) {                                            // first call for this shape: upload (blocking)
        HIP_TRY(hipStreamSynchronize(s), UKBB_EDEVICE);                      // nothing in flight may still read the old tables
        HIP_TRY(h->lstm_aux.ensure((map.size() * sizeof(int) + 3) / 4), UKBB_ENOMEM);
        HIP_TRY(hipMemcpy(h->lstm_aux.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice), UKBB_EDEVICE);
        h->lstm_aux_key = key;
    }
    float *out = prob;
    if (!out) {                                                              // the output kernel always forms the probabilities
        HIP_TRY(h->lstm_probw.ensure((size_t)n_seq * T * HW * C), UKBB_ENOMEM);
        out = h->lstm_probw.p;
    }
    const int NF = n_seq * T, NHID = h->arch.same_dim;
    const int *d_map = reinterpret_cast<const int *>(h->lstm_aux.p);
    rc = run_bilstm(h, h->act[h->feat_buf]->p, NF, d_map, n_seq, height, width, s);
    if (rc) return rc;
    const size_t kst = (size_t)n_seq * HW * NHID;
    for (int k = 0; k < T; ++k) {                                            // outputs straight into [N][T] order
        LstmOutArgs oa{};
        const size_t esz = h->plan_bfio ? 2 : 4;
        auto at = [esz](const float *p, size_t elems) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + elems * esz); };
        oa.h_bf16 = h->plan_bfio ? 1 : 0;
        oa.hf = k == 0 ? h->lstm_h1.p : at(h->lstm_hall.p, (size_t)k * kst);
        oa.mapf = k == 0 ? d_map : nullptr;
        oa.hb = k == T - 1 ? at(h->lstm_h1.p, (size_t)NF * HW * NHID) : at(h->lstm_hall.p, (size_t)T * kst + (size_t)k * kst);
        oa.mapb = k == T - 1 ? d_map + (size_t)(T - 1) * n_seq : nullptr;
        oa.w_out = dev_ptr(h, "lstm_out/w"); oa.b_out = dev_ptr(h, "lstm_out/bias");
        oa.prob = out + (size_t)k * HW * C;
        oa.logits = logits ? logits + (size_t)k * HW * C : nullptr;
        oa.pred = pred ? pred + (size_t)k * HW : nullptr;
        oa.m_stride = (long long)T * HW * C; oa.M = n_seq; oa.HW = (int)HW; oa.n_class = C;
        hipError_t e = launch_lstm_out(oa, s);
        if (e != hipSuccess) { set_err("ConvLSTM output kernel launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    }
    return UKBB_OK;
}

int ukbb_fcn_forward_cine(ukbb_fcn_handle *h, const float *image, int n_frames, int height, int width,
                          int weight_R, double weight_r, int time_step, float *prob, int32_t *pred, void *stream) {
    int rc = lstm_common_checks(h, image, "forward_cine");
    if (rc) return rc;
    const int T = h->arch.fc, C = h->arch.n_class, F = n_frames;
    if (!prob) { set_err("forward_cine: prob must not be NULL"); return UKBB_EINVAL; }
    if (2 * weight_R - 1 != T) { set_err("forward_cine: time window 2*weight_R-1 = %d, the model is unrolled for %d steps", 2 * weight_R - 1, T); return UKBB_EINVAL; }
    if (time_step < 1) { set_err("forward_cine: time_step must be >= 1 (got %d)", time_step); return UKBB_EINVAL; }
    const int rad = (T - 1) / 2;
    // the reference wraps a window index once only (i < 0: i + T; i >= T: i - T, deploy_network_ao.py:151-157):
    // with fewer than rad frames the wrapped index is still out of range and numpy raises IndexError
    if (F < rad || F < 1) { set_err("forward_cine: %d frames, the circular window of radius %d needs at least %d (the reference raises IndexError)", F, rad, rad > 1 ? rad : 1); return UKBB_EINVAL; }
    rc = prepare(h, F, height, width);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = run_plan(h, image, F, nullptr, nullptr, nullptr, s);               // each frame's U-Net features, once
    if (rc) return rc;
    const size_t HW = (size_t)height * width;
    const int Wn = (F + time_step - 1) / time_step;                         // window centres range(0, F, time_step) (:147)
    // host-side tables: window maps (deploy_network_ao.py:147-158), weights (:134-144), per-frame order + weight sums
    std::vector<int> map((size_t)T * Wn), order((size_t)F * T, -1);
    std::vector<double> wk(T), wsum(F, 0.0);
    for (int k = 0; k < T; ++k) {
        const int d = k > rad ? k - rad : rad - k;
        wk[k] = d <= weight_R ? pow(1.0 - (double)d / weight_R, weight_r) : 0.0;
        for (int w = 0; w < Wn; ++w) {
            int i = w * time_step - rad + k;
            if (i < 0) i += F; else if (i >= F) i -= F;
            map[(size_t)k * Wn + w] = i;
        }
    }
    // `prob[..., idx] += p * w` with fancy indexing (:179-180) is prob[idx] = prob[idx] + p*w: when a frame occurs
    // more than once in a window's idx (only if F < T) the LAST occurrence wins instead of accumulating
    // (SURVEY.md App. C.7); same for `weight[..., idx] += w`.  So per window a frame receives at most one term.
    std::vector<int> cnt(F, 0), last(F);
    for (int w = 0; w < Wn; ++w) {                                          // the reference's loop over window centres
        std::fill(last.begin(), last.end(), -1);
        for (int k = 0; k < T; ++k) last[map[(size_t)k * Wn + w]] = k;
        for (int k = 0; k < T; ++k) {                                       // frames in idx order; the order across frames is irrelevant
            const int f = map[(size_t)k * Wn + w];
            if (last[f] != k) continue;
            order[(size_t)f * T + cnt[f]++] = w * T + k;
            wsum[f] += wk[k];
        }
    }
    const size_t b_map = map.size() * sizeof(int), b_ord = order.size() * sizeof(int);
    const size_t off_ord = (b_map + 7) / 8 * 8, off_wk = (off_ord + b_ord + 7) / 8 * 8, off_ws = off_wk + T * sizeof(double);
    const size_t total = off_ws + F * sizeof(double);
    long long wr_bits;
    memcpy(&wr_bits, &weight_r, sizeof wr_bits);
    const long long key = (((((long long)F << 8) | T) * 1000003ll + time_step) * 1000003ll) ^ wr_bits ^ (1ll << 62);   // cine tables: (F, T, time_step, weight_r)
    if (h->lstm_aux_key != key) {                                            // first call for this shape: upload (blocking)
        HIP_TRY(hipStreamSynchronize(s), UKBB_EDEVICE);
        HIP_TRY(h->lstm_aux.ensure((total + 3) / 4), UKBB_ENOMEM);
        char *aux0 = reinterpret_cast<char *>(h->lstm_aux.p);
        HIP_TRY(hipMemcpy(aux0, map.data(), b_map, hipMemcpyHostToDevice), UKBB_EDEVICE);
        HIP_TRY(hipMemcpy(aux0 + off_ord, order.data(), b_ord, hipMemcpyHostToDevice), UKBB_EDEVICE);
        HIP_TRY(hipMemcpy(aux0 + off_wk, wk.data(), T * sizeof(double), hipMemcpyHostToDevice), UKBB_EDEVICE);
        HIP_TRY(hipMemcpy(aux0 + off_ws, wsum.data(), F * sizeof(double), hipMemcpyHostToDevice), UKBB_EDEVICE);
        h->lstm_aux_key = key;
    }
    char *aux = reinterpret_cast<char *>(h->lstm_aux.p);
    const int *d_map = reinterpret_cast<const int *>(aux);
    rc = run_bilstm(h, h->act[h->feat_buf]->p, F, d_map, Wn, height, width, s);
    if (rc) return rc;
    const int NHID = h->arch.same_dim;
    LstmTileArgs ta{};
    ta.k_stride = (long long)Wn * HW * NHID;
    {
        const size_t esz = h->plan_bfio ? 2 : 4;
        auto at = [esz](const float *p, size_t elems) { return reinterpret_cast<const float *>(reinterpret_cast<const char *>(p) + elems * esz); };
        ta.h_bf16 = h->plan_bfio ? 1 : 0;
        ta.hf = h->lstm_hall.p; ta.hb = at(h->lstm_hall.p, (size_t)T * ta.k_stride);
        ta.h1f = h->lstm_h1.p; ta.h1b = at(h->lstm_h1.p, (size_t)F * HW * NHID);
    }
    ta.map_first = d_map; ta.map_last = d_map + (size_t)(T - 1) * Wn;
    ta.w_out = dev_ptr(h, "lstm_out/w"); ta.b_out = dev_ptr(h, "lstm_out/bias");
    ta.order = reinterpret_cast<const int *>(aux + off_ord);
    ta.wk = reinterpret_cast<const double *>(aux + off_wk); ta.wsum = reinterpret_cast<const double *>(aux + off_ws);
    ta.prob = prob; ta.pred = pred; ta.F = F; ta.K = T; ta.Wn = Wn; ta.HW = (int)HW; ta.C = C;
    hipError_t e = launch_lstm_tile(ta, s);
    if (e != hipSuccess) { set_err("tiling kernel launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

int ukbb_fcn_num_kernels(const ukbb_fcn_handle *h) { return h ? (int)h->ops.size() : 0; }

const char *ukbb_fcn_kernel_name(const ukbb_fcn_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->ops.size()) return "";
    return h->ops[i].name.c_str();
}

double ukbb_fcn_kernel_macs(const ukbb_fcn_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->ops.size()) return 0.0;
    return h->ops[i].macs_per_image * h->last_n;
}

double ukbb_fcn_kernel_mfma_macs(const ukbb_fcn_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->ops.size()) return 0.0;
    const Op &op = h->ops[i];
    return (op.mfma_macs_per_image >= 0 ? op.mfma_macs_per_image : op.macs_per_image) * h->last_n;
}

double ukbb_fcn_kernel_mfma_macs_issued(const ukbb_fcn_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->ops.size()) return 0.0;
    const Op &op = h->ops[i];
    if (op.padded_macs_per_image >= 0) return op.padded_macs_per_image * h->last_n;
    return ukbb_fcn_kernel_mfma_macs(h, i);
}

int ukbb_fcn_set_precision(ukbb_fcn_handle *h, int precision) {
    if (!h || (precision != UKBB_PREC_FP32 && precision != UKBB_PREC_BF16 && precision != UKBB_PREC_F32X3)) { set_err("set_precision: bad argument"); return UKBB_EINVAL; }
    if (precision != h->precision) {
        if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_err("set_precision: device sync failed"); return UKBB_EDEVICE; }
        h->precision = precision;
        h->plan_h = h->plan_w = 0;               // re-plan (tilings and packed weights differ)
    }
    return UKBB_OK;
}

int ukbb_fcn_kernel_config(const ukbb_fcn_handle *h, int i) {
    if (!h || i < 0 || i >= (int)h->ops.size() || (h->ops[i].kind != OP_CONV && h->ops[i].kind != OP_TCONV)) return -1;
    return h->ops[i].cfg;
}

const char *ukbb_fcn_conv_config_name(int id) {
    for (int i = 0; i < num_conv_configs(); ++i)
        if (conv_config(i).id == id) return conv_config(i).name;
    return "";
}

int ukbb_fcn_set_timing(ukbb_fcn_handle *h, int enable) {
    if (!h) { set_err("set_timing: NULL handle"); return UKBB_EINVAL; }
    int rc = collect_events(h);
    if (rc) return rc;
    h->timing = enable != 0;
    h->timing_only = -1;
    return UKBB_OK;
}

int ukbb_fcn_set_timing_kernel(ukbb_fcn_handle *h, int kernel) {
    if (!h) { set_err("set_timing_kernel: NULL handle"); return UKBB_EINVAL; }
    int rc = collect_events(h);
    if (rc) return rc;
    if (kernel >= (int)h->ops.size()) { set_err("set_timing_kernel: index out of range"); return UKBB_EINVAL; }
    h->timing = true;
    h->timing_only = kernel < 0 ? -1 : kernel;
    return UKBB_OK;
}

int ukbb_fcn_kernel_times(ukbb_fcn_handle *h, double *sum_ms, int64_t *count, int n, int reset) {
    if (!h) { set_err("kernel_times: NULL handle"); return UKBB_EINVAL; }
    int rc = collect_events(h);
    if (rc) return rc;
    const int m = std::min<int>(n, (int)h->ops.size());
    for (int i = 0; i < m; ++i) {
        if (sum_ms) sum_ms[i] = h->t_sum[i];
        if (count) count[i] = h->t_cnt[i];
    }
    if (reset) { std::fill(h->t_sum.begin(), h->t_sum.end(), 0.0); std::fill(h->t_cnt.begin(), h->t_cnt.end(), 0); }
    return m;
}

int64_t ukbb_fcn_get_activation(ukbb_fcn_handle *h, const char *name, float *dst, int64_t cap) {
    if (!h || !name) { set_err("get_activation: NULL argument"); return UKBB_EINVAL; }
    for (size_t i = 0; i < h->act.size(); ++i) {
        if (h->act_name[i] != name) continue;
        const int64_t n = (int64_t)h->act_per_image[i] * h->last_n;
        if (!dst) return n;
        if (cap < n) { set_err("get_activation: buffer too small (%lld < %lld)", (long long)cap, (long long)n); return UKBB_EINVAL; }
        if (h->plan_bfio) {                            // stored as bf16: widen on the host
            std::vector<uint16_t> tmp((size_t)n);
            if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(tmp.data(), h->act[i]->p, (size_t)n * 2, hipMemcpyDeviceToHost) != hipSuccess) {
                set_err("get_activation: device copy failed");
                return UKBB_EDEVICE;
            }
            // channel-blocked on the device ([N][C/16][H][W][16], kernels.h); handed out as NHWC like the fp32 plans' maps
            const int64_t C = h->act_ch[i], per = (int64_t)h->act_per_image[i], hw = C > 0 ? per / C : 0;
            for (int64_t k = 0; k < n; ++k) {
                int64_t src = k;
                if (C > 16 && C % 16 == 0) {
                    const int64_t img = k / per, r = k - img * per, px = r / C, c = r - px * C;
                    src = img * per + ((c >> 4) * hw + px) * 16 + (c & 15);
                }
                const uint32_t u = (uint32_t)tmp[(size_t)src] << 16; memcpy(dst + k, &u, 4);
            }
            return n;
        }
        if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(dst, h->act[i]->p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            set_err("get_activation: device copy failed");
            return UKBB_EDEVICE;
        }
        return n;
    }
    // ConvLSTM working buffers, raw (tools/debug_lstm.py decodes the lane-native ones): whatever the last sequence call left in them
    for (const auto &kv : {std::pair<const char *, const DevBuf *>{"lstm:h1", &h->lstm_h1}, {"lstm:hall", &h->lstm_hall}, {"lstm:gx", &h->lstm_gx},
                           {"lstm:c1", &h->lstm_c1}, {"lstm:c", &h->lstm_c}}) {
        if (strcmp(kv.first, name)) continue;
        const int64_t n = (int64_t)kv.second->n;
        if (!dst) return n;
        if (cap < n) { set_err("get_activation: buffer too small (%lld < %lld)", (long long)cap, (long long)n); return UKBB_EINVAL; }
        if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(dst, kv.second->p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            set_err("get_activation: device copy failed");
            return UKBB_EDEVICE;
        }
        return n;
    }
    set_err("get_activation: no activation named '%s'", name);
    return UKBB_EINVAL;
}

}  // extern "C"
