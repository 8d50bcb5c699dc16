/*
 * ukbb_fcn.h -- C ABI of the MI355X (gfx950) FCN / U-Net segmentation engine.
 *
 * This boundary replaces the TensorFlow session of the reference deployment
 * scripts.  Each entry point cites the reference interface it stands for
 * (paths relative to the reference repository root):
 *
 *   ukbb_fcn_create        tf.Session() + tf.train.import_meta_graph(...) +
 *                          saver.restore(sess, model_path)
 *                          (common/deploy_network.py:44-49,
 *                           common/deploy_network_ao.py:53-58)
 *   ukbb_fcn_forward_host  sess.run(['prob:0','pred:0'],
 *                                   feed_dict={'image:0': x, 'training:0': False})
 *                          (common/deploy_network.py:110-111,195-196;
 *                           common/deploy_network_ao.py:121-122,252-253)
 *   ukbb_fcn_forward       same call with inputs/outputs already resident in
 *                          HBM (no reference counterpart: TF always copies)
 *   ukbb_fcn_destroy       leaving the `with tf.Session() as sess:` block
 *
 * The graph evaluated is the one common/train_network.py:142-199 builds with
 * common/network.py:170-230 (build_FCN) or common/network_ao.py:18-64 (UNet),
 * in inference mode ('training:0' = False): conv -> BN(moving stats) -> ReLU.
 *
 * Conventions: plain C types only; tensors are dense NHWC float32 / int32;
 * no exceptions cross the ABI: functions return 0 on success or a negative
 * UKBB_E* code and leave a message retrievable with ukbb_fcn_last_error()
 * (thread-local).  A handle belongs to one device and must not be used from
 * two threads at once (the reference is single-threaded too).
 * There is NO CPU fallback: create() fails when no gfx950 device is usable.
 */
#ifndef UKBB_FCN_H
#define UKBB_FCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UKBB_FCN_ABI_VERSION 7
#define UKBB_FCN_MAX_LEVEL 8

#define UKBB_OK 0
#define UKBB_EINVAL (-1)   /* bad argument (shape not a multiple of 16, NULL, ...) */
#define UKBB_EARCH (-2)    /* architecture not supported by the kernels */
#define UKBB_EDEVICE (-3)  /* HIP runtime error / no device */
#define UKBB_ENOMEM (-4)

#define UKBB_KIND_FCN 0    /* common/network.py:170 build_FCN */
#define UKBB_KIND_UNET 1   /* common/network_ao.py:18 UNet    */
#define UKBB_KIND_UNET_LSTM 2 /* common/network_ao.py:322 UNet_LSTM_Model, bidirectional (BiConv_LSTM :255);
                               same_dim = ConvLSTM hidden channels (16), fc = unrolled time steps (9);
                               weights: the UNet layers without its conv_out, then per direction the gate
                               kernel [3,3,16+16,64] + bias[64] (forward, backward), then the output conv
                               kernel [1,1,32,n_class] + bias.  Use forward_seq / forward_cine.
                               The single-direction head Conv_LSTM (:214-252, bidirectional=False :349-352) is this
                               kind with an all-zero backward gate kernel + bias and the output kernel [W; 0]: exact
                               (that cell's hidden maps are 0 at every step), detected at create, its time steps are
                               skipped (round 6; weights.embed_unidirectional_lstm does the embedding). */

/* Hyper-parameters of build_FCN / UNet as bound in common/train_network.py:174-195
 * and common/train_network_ao.py:268,275-284. */
typedef struct ukbb_fcn_arch {
    int32_t kind;
    int32_t n_class;
    int32_t n_level;
    int32_t n_filter[UKBB_FCN_MAX_LEVEL];
    int32_t n_block[UKBB_FCN_MAX_LEVEL];
    int32_t same_dim;   /* FCN only (32) */
    int32_t fc;         /* FCN only (64) */
} ukbb_fcn_arch;

typedef struct ukbb_fcn_handle ukbb_fcn_handle;

int ukbb_fcn_abi_version(void);
const char *ukbb_fcn_last_error(void);

/* Number of floats ukbb_fcn_create expects for this architecture, or 0 if the
 * architecture is malformed.  Layout: for every layer in graph order
 * (encoder convs; FCN: same_dim0..4, out0, out1, logits; UNet: per decoder
 * level the transposed conv then its convs, logits): kernel in TF layout
 * (HWIO; [kh,kw,Cout,Cin] for the transposed conv), then gamma, beta,
 * moving_mean, moving_variance -- or bias for the logits layer. */
size_t ukbb_fcn_weight_count(const ukbb_fcn_arch *arch);

/* Bind a model to `device` (HIP ordinal).  Folds BN into the kernels, packs
 * MFMA operand fragments and uploads them.  `weights` is host memory and is
 * not referenced after return.  NULL on failure. */
ukbb_fcn_handle *ukbb_fcn_create(const ukbb_fcn_arch *arch, const float *weights,
                                 size_t n_floats, int device);
void ukbb_fcn_destroy(ukbb_fcn_handle *h);

/* Arithmetic of the MFMA convolutions (BASELINE config 5).  UKBB_PREC_FP32 (default): f32 inputs,
 * exact f32 MFMA.  UKBB_PREC_BF16: bf16 (RNE) MFMA operands, fp32 accumulation.  On a UKBB_KIND_UNET
 * handle (round 3) every activation between layers is ALSO stored as bf16 in HBM (rounded once, after
 * bias + ReLU), every layer runs on the bf16 matrix instructions (16-channel layers zero-padded to the
 * 32-row shape), the first layer (fp32 arithmetic) is evaluated inside the second one's staging and
 * the logits conv + softmax / argmax inside the fused tail: 20 launches, 3.6-3.8x the fp32 rate
 * at N = 100 x 256x256 (round 4-5 builds).  On a UKBB_KIND_UNET_LSTM handle (round 5) the U-Net runs the same bf16-storage
 * plan and the ConvLSTM runs as direct 3x3 convs on the bf16 matrix instruction with its features, the
 * hoisted x half of the gate pre-activations and the hidden maps as bf16 in HBM (accumulation and cell
 * state fp32): 8-9 ms instead of 19 per 100-frame cine, per-class Dice >= 0.98 against the fp32 cine.  On FCN handles only the operands are bf16 (fp32 activations in
 * HBM; layers without such a tiling stay fp32).  Not bit-compatible with the reference; meant to be
 * judged by Dice against the fp32 result (common/image_utils.py:171-175): 0.993 / 0.992 measured.
 * Concurrency (round 6): a handle of any kind and precision may run beside kernels of other streams of the process (one handle per stream,
 * one thread per handle; tests/test_concurrency_gpu.py runs two engines on two streams with batches in flight on both).  Until round 6 a
 * UKBB_KIND_UNET handle in UKBB_PREC_BF16 could not: 16-byte buffer stores of its weight-stationary kernels lacked a wait state that hipcc
 * does not emit when the store's soffset is an SGPR, and lost their data under another queue's memory traffic (kernels_ws.hip
 * store_b128_sofs, profiles/r06_notes.md section 10; tests/test_store_hazard.py guards the ISA of the whole library). */
#define UKBB_PREC_FP32 0
#define UKBB_PREC_BF16 1
/* UKBB_PREC_F32X3 (round 2, FCN head only so far): fp32 results from bf16 matrix instructions.  Every fp32 operand x is
 * split exactly into three bf16 pieces (h = bf16(x), m = bf16(x - h), l = x - h - m); the six partial products whose
 * magnitude can reach 2^-24 of |x||w| (hh, hm, mh, mm, hl, lh) run on the dense matrix cores with fp32 accumulation,
 * the other three (< 2^-24 |x||w|) are dropped -- less than what fp32 accumulation itself rounds away.  Measured
 * against the fp64 oracle the logits error is the same as UKBB_PREC_FP32's (4e-6 relative) and the label maps are
 * identical; it is a different instruction sequence from an fp32 MFMA, so it is offered as its own mode and reported
 * under its own name, never as the default.  Layers not converted yet run exactly as in UKBB_PREC_FP32. */
#define UKBB_PREC_F32X3 2
int ukbb_fcn_set_precision(ukbb_fcn_handle *h, int precision);

/* Pre-size the activation workspace for batches up to n x h x w (optional;
 * forward() grows it on demand, which synchronises the device). */
int ukbb_fcn_reserve(ukbb_fcn_handle *h, int n, int height, int width);

/* One forward pass, everything in device memory, asynchronous on `stream`
 * (a hipStream_t; NULL = the null stream).  image: [n,h,w,1] float32 with
 * h % 16 == 0 and w % 16 == 0 (the reference pads to that,
 * common/deploy_network.py:97).  Any of logits / prob ([n,h,w,n_class]
 * float32) and pred ([n,h,w] int32 = argmax over the float32 probabilities,
 * lowest index on ties, as common/train_network.py:199 defines it -- equal to
 * the argmax of the logits except where two classes' probabilities round to
 * the same float) may be NULL to skip producing it. */
int ukbb_fcn_forward(ukbb_fcn_handle *h, const float *image, int n, int height, int width,
                     float *logits, float *prob, int32_t *pred, void *stream);

/* Same with host buffers: H2D copy, forward, D2H copy, synchronous --
 * the exact shape of the reference's sess.run call. */
int ukbb_fcn_forward_host(ukbb_fcn_handle *h, const float *image, int n, int height, int width,
                          float *logits, float *prob, int32_t *pred);

/* ---- UNet-LSTM (kind 2): the default aortic model of demo_pipeline.py:116-117 -------------------
 * Reference call (common/deploy_network_ao.py:171-172):
 *   prob_idx = sess.run('prob:0', {'image:0': image_idx [N,T,X,Y,1], 'training:0': False})  -> [N,T,X,Y,C]
 * forward_seq is that call: n_seq sequences of T = arch.fc frames each, frame (s, t) at image + (s*T + t)*H*W;
 * outputs in the same [N][T] order (any of logits / prob / pred may be NULL).  Device pointers; asynchronous on
 * `stream` except for the first call with a new n_seq (one blocking upload of the window index table). */
int ukbb_fcn_forward_seq(ukbb_fcn_handle *h, const float *image, int n_seq, int height, int width,
                         float *logits, float *prob, int32_t *pred, void *stream);

/* The whole 'UNet-LSTM' branch of the reference's per-subject loop (common/deploy_network_ao.py:129-183,189)
 * for one slice position: n_frames cine frames at image[f]; frames range(0, n_frames, time_step) are each the
 * centre of one circular window of T frames (--time_step, :26,147), window probabilities are tiled with the
 * weights (1 - |k - rad|/weight_R)^weight_r exactly as the reference accumulates them (float32 accumulator
 * updated through float64, same order), then prob /= weight and pred = argmax.  The U-Net features of a frame
 * are computed once instead of once per window (T times in the reference); results are identical because the
 * U-Net acts per frame.  Reference corner cases reproduced: a frame no window reaches (time_step > window)
 * gets prob = NaN (0/0) and pred 0; with n_frames < T a frame that occurs twice in one window receives only
 * its LAST occurrence (numpy's `a[idx] += b` does not accumulate duplicates).
 * Requires 2*weight_R - 1 == arch.fc, time_step >= 1 and n_frames >= (T-1)/2 (below that the reference itself
 * raises IndexError).  prob [n_frames][H][W][C], pred [n_frames][H][W].
 * Device scratch the handle keeps for this call, besides the U-Net's activations (freed by destroy): with F = n_frames, Wn = windows
 * (= ceil(F / time_step)), HW = H*W, T = arch.fc, e = 4 bytes (fp32) or 2 (UKBB_PREC_BF16):
 *   every step's hidden maps 2*T*Wn*HW*16*e  +  hoisted gate pre-activations 2*F*HW*64*e  +  first-step hidden maps 2*F*HW*16*e
 *   +  cell state (2*F + Wn)*HW*16*4.
 * F = Wn = 100 frames of 256x256: 7.5 + 3.4 + 0.8 + 1.3 = 13.0 GB in fp32, 7.2 GB in bf16 -- several handles per GPU, or cines of
 * hundreds of frames, reach UKBB_ENOMEM on that, not on the U-Net (0.9 GB). */
int ukbb_fcn_forward_cine(ukbb_fcn_handle *h, const float *image, int n_frames, int height, int width,
                          int weight_R, double weight_r, int time_step, float *prob, int32_t *pred, void *stream);

/* ---- device-side pre/post-processing of the deploy loop ----------------
 * Stateless; device pointers; asynchronous on `stream` unless stated.  They take the host numpy work
 * (a full sort of ~20 M voxels per subject, pad, transposes) off the critical path of the reference's
 * common/deploy_network.py:86-131. */

/* Exact order statistics: out_host[i] = the ranks[i]-th smallest (0-based) of the n float32 values at
 * d_data (16-byte aligned, any order; NaNs are not expected in MR magnitudes and sort as their bit
 * pattern).  4-pass radix select on device; synchronous (the few results are copied back).
 * Replaces the sort inside np.percentile(image, thres), common/image_utils.py:72 -- the caller does
 * numpy's linear interpolation between neighbouring ranks on the host.  1 <= nranks <= 8. */
int ukbb_fcn_select_kth(const float *d_data, size_t n, const uint64_t *ranks, int nranks, float *out_host, void *stream);

/* (X,Y,Z,T) float32 volume with element strides (sx,sy,sz,st) -> network input d_batch[T*Z][X2][Y2]:
 * the in-place clip to [lo, hi] as stored in float32, (v - lo) / (hi - lo) in float64 rounded to
 * float32, centred zero padding (x_pre, y_pre before; X2, Y2 multiples of 16), batch index b = t*Z + z.
 * Replaces common/image_utils.py:73-76 and common/deploy_network.py:97-107. */
int ukbb_fcn_rescale_pack(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st,
                          double lo, double hi, int X2, int Y2, int x_pre, int y_pre, float *d_batch, void *stream);

/* int32 labels d_pred[T*Z][X2][Y2] -> cropped uint8 volume d_vol in NIfTI order (x fastest, then y, z, t)
 * and d_counts[T][n_class] = voxels of each class per frame (what the ES pick of
 * common/deploy_network.py:125-130 sums).  Replaces :114-116.  n_class <= 16. */
int ukbb_fcn_unpack_labels(const int32_t *d_pred, int X, int Y, int Z, int T, int X2, int Y2, int x_pre, int y_pre, int n_class,
                           uint8_t *d_vol, uint64_t *d_counts, void *stream);

/* -- the aortic z-score, common/image_utils.py:60-67 (normalise_intensity), bit-identical to numpy ----------------
 * np.mean / np.std over image[roi] are float32 PAIRWISE sums over the compressed array; the three calls below
 * reproduce numpy's summation tree rather than just its value (ukbb_cardiac_amd/device_pipeline.py
 * device_zscore_stats shows the sequence, including the percentile threshold from ukbb_fcn_select_kth). */

/* d_out[i] = the i-th element >= thr of the (X,Y,Z,T) float32 volume with element strides (sx,sy,sz,st), in
 * row-major INDEX order (last index fastest: the order of numpy's image[image >= thr]).  d_out holds up to
 * X*Y*Z*T floats; *n_host receives the count.  Synchronous. */
int ukbb_fcn_roi_compact(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st, float thr,
                         float *d_out, uint64_t *n_host, void *stream);

/* *sum_host = np.add.reduce over the n contiguous float32 values at d_a (squared_dev = 0) or over
 * (d_a[i] - mean)^2 evaluated in float32 (squared_dev = 1, the inner sum of np.var), with numpy's pairwise
 * summation order (blocks of <= 128 elements on 8 interleaved accumulators, halves split at a multiple of 8, one
 * such tree per 8192-element buffer of numpy's reduction iterator [np.getbufsize() default], buffers added in
 * sequence): every leaf block is summed by one device thread, the host adds the leaves up.  Synchronous. */
int ukbb_fcn_pairwise_sum(const float *d_a, uint64_t n, int squared_dev, float mean, float *sum_host, void *stream);

/* ukbb_fcn_rescale_pack with the z-score arithmetic: d_batch[t*Z+z][X2][Y2] = (v - mu) / den in float32 (IEEE
 * division), zero padding around it.  Replaces image_utils.py:67 + deploy_network_ao.py:105-108,147-150. */
int ukbb_fcn_zscore_pack(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st,
                         float mu, float den, int X2, int Y2, int x_pre, int y_pre, float *d_batch, void *stream);

/* ---- label-volume files (host only: no device, no stream) ---------------------------------------
 * What the reference does with the result: nib.save of np.zeros(image.shape) filled with the labels
 * (common/deploy_network.py:92,116,136-138; deploy_network_ao.py:189-196) -- for a short-axis subject
 * 160 MB of float64 through zlib, 0.5 s of a core against 11 ms of network time.
 *
 * ukbb_fcn_gzip_labels writes ONE gzip member (RFC 1952 / 1951, one deflate block) whose inflated
 * content is  prefix || labels converted to the NIfTI voxel type `nifti_datatype`
 * (2 uint8, 4 int16, 8 int32, 16 float32, 64 float64; little-endian),  i.e. the .nii.gz nibabel writes
 * when prefix is the 352-byte NIfTI-1 header, without forming the converted volume: a run of equal
 * labels becomes E literals + matches of distance E, its CRC-32 is computed per run.  labels[n] uint8
 * in file order (x fastest).  Returns the number of bytes written to out, or UKBB_ENOMEM when out_cap
 * is too small (ukbb_fcn_gzip_labels_bound is always sufficient; a label map of a real segmentation
 * needs ~2 % of it), or UKBB_EINVAL. */
uint64_t ukbb_fcn_gzip_labels_bound(uint64_t n_voxels, int nifti_datatype, uint64_t prefix_len);
int64_t ukbb_fcn_gzip_labels(const uint8_t *labels, uint64_t n_voxels, int nifti_datatype, const uint8_t *prefix, uint64_t prefix_len,
                             uint8_t *out, uint64_t out_cap);
/* Same, with the Huffman code set chosen: UKBB_GZIP_DYNAMIC (what ukbb_fcn_gzip_labels uses: a counting pass over the
 * runs, then codes built from the exact token histogram -- files no larger than zlib level 1 writes for the same volume)
 * or UKBB_GZIP_FIXED (the fixed codes of RFC 1951 3.2.6: no counting pass, files 2-4x larger).  Same inflated bytes. */
#define UKBB_GZIP_FIXED 0
#define UKBB_GZIP_DYNAMIC 1
int64_t ukbb_fcn_gzip_labels_mode(const uint8_t *labels, uint64_t n_voxels, int nifti_datatype, const uint8_t *prefix, uint64_t prefix_len,
                                  uint8_t *out, uint64_t out_cap, int mode);

/* ---- image files in (host only) -------------------------------------------------------------------
 * What the reference does first with every subject: nib.load(image_name).get_data()
 * (common/deploy_network.py:80-83, deploy_network_ao.py:88-92) -- for a short-axis cine 40 MB of int16
 * through zlib, 0.26 s of a core against 11 ms of network time; it bounds a cohort run.
 *
 * ukbb_fcn_gunzip inflates a whole .gz file image (every member, zero padding between / after members
 * skipped) from src into dst and returns the number of bytes written; each member's ISIZE and -- unless
 * verify_crc is 0 -- CRC-32 are checked.  Whole-buffer decoder (64-bit bit buffer, two-level tables, wide
 * match copies, carry-less-multiply CRC), 1.75-2.3x zlib 1.2.11 on MR image data.  Strict by design: returns
 * UKBB_ENOMEM when the content does not fit dst_cap and UKBB_EINVAL for anything else it does not accept
 * (truncated or invalid stream, header CRC flag, trailing bytes that are not a member, CRC / length
 * mismatch); the caller falls back to zlib, which raises -- or accepts -- as before.
 * ukbb_fcn_gzip_crc: zlib's crc32(crc, data, n). */
int64_t ukbb_fcn_gunzip(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap, int verify_crc);
uint32_t ukbb_fcn_gzip_crc(uint32_t crc, const uint8_t *data, uint64_t n);

/* ---- measurement / introspection (bench.py, tests) ---------------------- */

/* Kernel launches of one forward, in launch order. */
int ukbb_fcn_num_kernels(const ukbb_fcn_handle *h);
const char *ukbb_fcn_kernel_name(const ukbb_fcn_handle *h, int i);
/* Algorithmic MACs kernel i performs for the LAST forward's shape. */
double ukbb_fcn_kernel_macs(const ukbb_fcn_handle *h, int i);
/* Multiplies kernel i actually issues to the matrix pipe for that shape (tile padding excluded).
 * Differs from the algorithmic count where the kernel runs a cheaper algorithm: Winograd F(2x2,3x3)
 * layers (16/36 of the direct count), the head (level-0 slice of the 160->64 conv only; the other
 * slices run at low resolution inside the sqg kernels), the first layer (vector ALU, reported as 0). */
double ukbb_fcn_kernel_mfma_macs(const ukbb_fcn_handle *h, int i);
/* The same INCLUDING what the kernel issues for partly filled tiles / Winograd regions (every Winograd region runs its 32 tile slots
 * whatever part of them the map fills): = SQ_INSTS_MFMA x MACs per instruction of a rocprofv3 counter pass. */
double ukbb_fcn_kernel_mfma_macs_issued(const ukbb_fcn_handle *h, int i);

/* Tiling id the plan chose for kernel i (conv kernels; -1 for the others) and
 * its descriptive name; used by tools/tune_convs.py.  The environment variable
 * UKBB_CONV_CFG="layer:id,layer:id" overrides the choice at plan time. */
int ukbb_fcn_kernel_config(const ukbb_fcn_handle *h, int i);
const char *ukbb_fcn_conv_config_name(int id);

/* When enabled, every launch of forward() is bracketed by hipEvents recorded
 * on the forward's own stream; ukbb_fcn_kernel_times() then synchronises and
 * returns, per kernel, the summed duration in ms and the number of timed
 * launches since the last reset. */
int ukbb_fcn_set_timing(ukbb_fcn_handle *h, int enable);
/* Time only kernel `kernel` (index in launch order; < 0 = all): two event records per
 * forward instead of two per launch, so the timed region is not perturbed. */
int ukbb_fcn_set_timing_kernel(ukbb_fcn_handle *h, int kernel);
int ukbb_fcn_kernel_times(ukbb_fcn_handle *h, double *sum_ms, int64_t *count, int n, int reset);

/* Copy an intermediate activation of the LAST forward to host (tests only).
 * Names: "conv0".."conv4" (level outputs), "g1".."g4" (FCN: out0's level-l slice applied to
 * the squeezed map at low resolution, 64 channels),
 * "up3".."up0" (UNet decoder outputs).  Returns the number of floats, or a
 * negative error; with dst == NULL only the size is returned. */
int64_t ukbb_fcn_get_activation(ukbb_fcn_handle *h, const char *name, float *dst, int64_t cap);

/* Shader clock the chip holds RIGHT NOW (bench.py, next to its roofline: the MFMA peak scales with it).  Launches one wave on
 * `stream` that spins for about spin_us microseconds and compares the shader-cycle counter (s_memtime) with the constant 100 MHz
 * counter (s_memrealtime); waits for that wave only.  Run it on a stream of its own while the measured work is queued on another:
 * a single extra wave does not disturb it.  *mhz = the shader clock in MHz.  No reference counterpart (measurement only). */
int ukbb_fcn_clock_probe(int device, void *stream, int spin_us, double *mhz);

/* Synthetic subject for BASELINE.json configs[3] ("1000 synthetic subjects x 500 slices, generated on
 * device from seed = subject id"): d_out[i] = a * b / 2048 with (a, b) = the two low 12-bit fields of splitmix64's finaliser of
 * seed * 0x9E3779B97F4A7C15 + i -- exact in float32, so ukbb_cardiac_amd/synthetic_cohort.py reproduces it bit for bit in numpy
 * for the oracle.  Asynchronous on `stream`.  No reference counterpart (the reference reads sa.nii.gz, common/deploy_network.py:80-83);
 * measurement and tests only. */
int ukbb_fcn_synth_volume(uint64_t seed, size_t n, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UKBB_FCN_H */
