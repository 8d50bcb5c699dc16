/*
 * fcn_oracle.c -- plain-C fp32 restatement of the reference's FCN / U-Net
 * inference graph.  TEST INFRASTRUCTURE ONLY (checker + bench.py's
 * cpu_baseline "port" leg); nothing under ukbb_cardiac_amd/ links or calls it.
 *
 * PARITY UNPINNED vs TensorFlow (TF 1.x absent; see oracle/__init__.py).  It is
 * pinned to oracle/fcn_oracle.py (numpy fp64) by tests/test_c_oracle.py.
 *
 * Follows, op by op and UNFUSED (separate conv, BN, ReLU, materialised
 * upsampled maps and 160-channel concat), the reference files:
 *   common/network.py:19-25    conv2d_bn_relu          -> conv2d_same + bn_relu
 *   common/network.py:28-34    conv2d_transpose_bn_relu-> conv2d_transpose_same + bn_relu
 *   common/network.py:138-167  transpose_upsample2d    -> upsample_bilinear (per channel;
 *                              the reference's dense diagonal filter multiplies 31/32 zeros,
 *                              skipping them is a favour to this CPU baseline)
 *   common/network.py:170-230  build_FCN               -> fcn_forward
 *   common/network_ao.py:18-64 UNet                    -> unet_forward
 *   common/train_network.py:198-199 prob / pred        -> softmax_argmax
 * TF op semantics (SAME padding, transposed-conv crop, BN eps): SURVEY.md App. B.
 *
 * Weights arrive in the same flat canonical order as ukbb_fcn_create takes
 * (include/ukbb_fcn.h), unfolded.  Tensors NHWC, kernels HWIO.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-3f
#define MAXC 512

/* Scratch buffers are recycled between calls: a forward touches ~100 MB per
 * slice of intermediates, and mmap/munmap + first-touch page faults on every
 * call would otherwise dominate the timing on many-core hosts (the
 * cpu_baseline leg calls forward repeatedly).  Calls are serial. */
#define POOL_N 64
static struct { void *p; size_t n; int used; } g_pool[POOL_N];

static void *oalloc(size_t bytes) {
    int free_slot = -1, best = -1;
    for (int i = 0; i < POOL_N; ++i) {
        if (g_pool[i].p && !g_pool[i].used && g_pool[i].n >= bytes &&
            (best < 0 || g_pool[i].n < g_pool[best].n)) best = i;
        if (!g_pool[i].p && free_slot < 0) free_slot = i;
    }
    if (best >= 0 && g_pool[best].n <= bytes + bytes / 4 + 4096) { g_pool[best].used = 1; return g_pool[best].p; }
    if (free_slot < 0) {                     /* pool full: drop an idle block */
        for (int i = 0; i < POOL_N; ++i)
            if (!g_pool[i].used) { free(g_pool[i].p); g_pool[i].p = 0; free_slot = i; break; }
        if (free_slot < 0) return 0;
    }
    void *p = malloc(bytes);
    if (!p) return 0;
    g_pool[free_slot].p = p; g_pool[free_slot].n = bytes; g_pool[free_slot].used = 1;
    return p;
}

static void ofree(const void *p) {
    if (!p) return;
    for (int i = 0; i < POOL_N; ++i)
        if (g_pool[i].p == p) { g_pool[i].used = 0; return; }
}

static void same_pads(int n_in, int k, int s, int *n_out, int *before) {
    int o = (n_in + s - 1) / s;
    int tot = (o - 1) * s + k - n_in;
    if (tot < 0) tot = 0;
    *n_out = o;
    *before = tot / 2;
}

/* tf.layers.conv2d(padding='same', use_bias=False) */
static void conv2d_same(const float *x, int N, int H, int W, int Cin, const float *w, int K, int stride,
                        int Cout, float *out) {
    int Ho, Wo, pt, pl;
    same_pads(H, K, stride, &Ho, &pt);
    same_pads(W, K, stride, &Wo, &pl);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            float acc[MAXC];
            for (int ox = 0; ox < Wo; ++ox) {
                for (int co = 0; co < Cout; ++co) acc[co] = 0.f;
                for (int kh = 0; kh < K; ++kh) {
                    const int iy = oy * stride + kh - pt;
                    if (iy < 0 || iy >= H) continue;
                    for (int kw = 0; kw < K; ++kw) {
                        const int ix = ox * stride + kw - pl;
                        if (ix < 0 || ix >= W) continue;
                        const float *xp = x + ((size_t)(n * H + iy) * W + ix) * Cin;
                        const float *wp = w + (size_t)(kh * K + kw) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float xv = xp[ci];
                            const float *wr = wp + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) acc[co] += xv * wr[co];
                        }
                    }
                }
                memcpy(out + ((size_t)(n * Ho + oy) * Wo + ox) * Cout, acc, sizeof(float) * Cout);
            }
        }
}

/* tf.layers.conv2d_transpose(k=3, strides=2, padding='same'), filter [kh][kw][Cout][Cin]:
 * out[o] = sum_i x[i] * w[o + pb - i*s], pb = forward conv's pad_before. */
static void conv2d_transpose_same(const float *x, int N, int h, int wd, int Cin, const float *w, int K, int s,
                                  int Cout, float *out) {
    const int H = h * s, W = wd * s;
    int tmp, pt, pl;
    same_pads(H, K, s, &tmp, &pt);
    same_pads(W, K, s, &tmp, &pl);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy) {
            float acc[MAXC];
            for (int ox = 0; ox < W; ++ox) {
                for (int co = 0; co < Cout; ++co) acc[co] = 0.f;
                for (int kh = 0; kh < K; ++kh) {
                    const int ty = oy + pt - kh;
                    if (ty < 0 || ty % s) continue;
                    const int iy = ty / s;
                    if (iy >= h) continue;
                    for (int kw = 0; kw < K; ++kw) {
                        const int tx = ox + pl - kw;
                        if (tx < 0 || tx % s) continue;
                        const int ix = tx / s;
                        if (ix >= wd) continue;
                        const float *xp = x + ((size_t)(n * h + iy) * wd + ix) * Cin;
                        const float *wp = w + (size_t)(kh * K + kw) * Cout * Cin;
                        for (int co = 0; co < Cout; ++co) {
                            const float *wr = wp + (size_t)co * Cin;
                            float sum = 0.f;
                            for (int ci = 0; ci < Cin; ++ci) sum += xp[ci] * wr[ci];
                            acc[co] += sum;
                        }
                    }
                }
                memcpy(out + ((size_t)(n * H + oy) * W + ox) * Cout, acc, sizeof(float) * Cout);
            }
        }
}

/* tf.layers.batch_normalization(training=False) then tf.nn.relu; p = gamma,beta,mean,var */
static void bn_relu(float *x, size_t npix, int C, const float *p) {
    float inv[MAXC];
    for (int c = 0; c < C; ++c) inv[c] = p[c] / sqrtf(p[3 * C + c] + BN_EPS);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i) {
        float *v = x + i * C;
        for (int c = 0; c < C; ++c) {
            const float y = (v[c] - p[2 * C + c]) * inv[c] + p[C + c];
            v[c] = y > 0.f ? y : 0.f;
        }
    }
}

/* transpose_upsample2d(x, f): triangle kernel of size 2f-1, SAME crop, no border renormalisation.
 * Writes into channel slice [coff, coff+C) of a [N,H*f,W*f,Ctot] buffer (the concat). */
static void upsample_bilinear(const float *x, int N, int h, int w, int C, int f, float *out, int Ctot, int coff) {
    const int H = h * f, W = w * f, pb = (f - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy) {
            const int ty = oy + pb, y1 = ty / f, jy = ty % f, y0 = y1 - 1;
            const float wy1 = y1 < h ? (float)(jy + 1) / f : 0.f, wy0 = y0 >= 0 ? (float)(f - 1 - jy) / f : 0.f;
            for (int ox = 0; ox < W; ++ox) {
                const int tx = ox + pb, x1 = tx / f, jx = tx % f, x0 = x1 - 1;
                const float wx1 = x1 < w ? (float)(jx + 1) / f : 0.f, wx0 = x0 >= 0 ? (float)(f - 1 - jx) / f : 0.f;
                const int cy0 = y0 < 0 ? 0 : y0, cy1 = y1 >= h ? h - 1 : y1;
                const int cx0 = x0 < 0 ? 0 : x0, cx1 = x1 >= w ? w - 1 : x1;
                const float *a = x + ((size_t)(n * h + cy0) * w + cx0) * C, *b = x + ((size_t)(n * h + cy0) * w + cx1) * C;
                const float *c_ = x + ((size_t)(n * h + cy1) * w + cx0) * C, *d = x + ((size_t)(n * h + cy1) * w + cx1) * C;
                float *o = out + ((size_t)(n * H + oy) * W + ox) * Ctot + coff;
                const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
                for (int c = 0; c < C; ++c) o[c] = w00 * a[c] + w01 * b[c] + w10 * c_[c] + w11 * d[c];
            }
        }
}

static void copy_channels(const float *x, size_t npix, int C, float *out, int Ctot, int coff) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i) memcpy(out + i * Ctot + coff, x + i * C, sizeof(float) * C);
}

/* softmax over the last axis and int32 argmax of prob (lowest index on ties) */
static void softmax_argmax(const float *logits, size_t npix, int C, float *prob, int32_t *pred) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i) {
        const float *l = logits + i * C;
        float m = l[0];
        for (int c = 1; c < C; ++c) m = l[c] > m ? l[c] : m;
        float e[16], sum = 0.f;
        for (int c = 0; c < C; ++c) { e[c] = expf(l[c] - m); sum += e[c]; }
        int best = 0; float pb = -1.f;
        for (int c = 0; c < C; ++c) {
            const float p = e[c] / sum;
            if (prob) prob[i * C + c] = p;
            if (p > pb) { pb = p; best = c; }
        }
        if (pred) pred[i] = best;
    }
}

typedef struct {
    int32_t kind, n_class, n_level;
    int32_t n_filter[8], n_block[8];
    int32_t same_dim, fc;
} oracle_arch;   /* same layout as ukbb_fcn_arch */

static const float *unit(const float *x, int N, int H, int W, int Cin, int K, int stride, int Cout,
                         const float **wp, float *out) {
    const float *k = *wp;
    conv2d_same(x, N, H, W, Cin, k, K, stride, Cout, out);
    const float *bn = k + (size_t)K * K * Cin * Cout;
    int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    bn_relu(out, (size_t)N * Ho * Wo, Cout, bn);
    *wp = bn + 4 * (size_t)Cout;
    return out;
}

/* build_FCN forward.  Returns 0, or -1 on bad arguments / out of memory. */
int oracle_fcn_forward(const oracle_arch *a, const float *weights, const float *image, int N, int H, int W,
                       float *logits_out, float *prob_out, int32_t *pred_out) {
    if (!a || a->kind != 0 || a->n_level > 8 || (H % 16) || (W % 16)) return -1;
    const int L = a->n_level;
    const float *wp = weights;
    float *feat[8] = {0};
    int fh[8], fw[8];
    const float *x = image;
    int cin = 1, h = H, w = W;
    for (int l = 0; l < L; ++l) {
        for (int i = 0; i < a->n_block[l]; ++i) {
            const int stride = (l > 0 && i == 0) ? 2 : 1;
            const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
            float *out = (float *)oalloc(sizeof(float) * (size_t)N * ho * wo * a->n_filter[l]);
            if (!out) return -1;
            unit(x, N, h, w, cin, 3, stride, a->n_filter[l], &wp, out);
            if (i > 0) ofree(x);          /* level outputs (i == 0 inputs) are kept */
            x = out; cin = a->n_filter[l]; h = ho; w = wo;
        }
        feat[l] = (float *)x; fh[l] = h; fw[l] = w;
    }
    const int SD = a->same_dim, CT = SD * L;
    float *concat = (float *)oalloc(sizeof(float) * (size_t)N * H * W * CT);
    if (!concat) return -1;
    for (int l = 0; l < L; ++l) {
        float *sq = (float *)oalloc(sizeof(float) * (size_t)N * fh[l] * fw[l] * SD);
        if (!sq) return -1;
        unit(feat[l], N, fh[l], fw[l], a->n_filter[l], 1, 1, SD, &wp, sq);
        if (l == 0) copy_channels(sq, (size_t)N * H * W, SD, concat, CT, 0);
        else upsample_bilinear(sq, N, fh[l], fw[l], SD, 1 << l, concat, CT, SD * l);
        ofree(sq);
        ofree(feat[l]);
    }
    float *o0 = (float *)oalloc(sizeof(float) * (size_t)N * H * W * a->fc);
    float *o1 = (float *)oalloc(sizeof(float) * (size_t)N * H * W * a->fc);
    float *lg = logits_out ? logits_out : (float *)oalloc(sizeof(float) * (size_t)N * H * W * a->n_class);
    if (!o0 || !o1 || !lg) return -1;
    unit(concat, N, H, W, CT, 1, 1, a->fc, &wp, o0);
    ofree(concat);
    unit(o0, N, H, W, a->fc, 1, 1, a->fc, &wp, o1);
    ofree(o0);
    conv2d_same(o1, N, H, W, a->fc, wp, 1, 1, a->n_class, lg);
    const float *bias = wp + (size_t)a->fc * a->n_class;
    const size_t npix = (size_t)N * H * W;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i)
        for (int c = 0; c < a->n_class; ++c) lg[i * a->n_class + c] += bias[c];
    ofree(o1);
    softmax_argmax(lg, npix, a->n_class, prob_out, pred_out);
    if (!logits_out) ofree(lg);
    return 0;
}

/* UNet forward (network_ao.py:18-64). */
int oracle_unet_forward(const oracle_arch *a, const float *weights, const float *image, int N, int H, int W,
                        float *logits_out, float *prob_out, int32_t *pred_out) {
    if (!a || a->kind != 1 || a->n_level > 8 || (H % 16) || (W % 16)) return -1;
    const int L = a->n_level;
    const float *wp = weights;
    float *feat[8] = {0};
    int fh[8], fw[8];
    const float *x = image;
    int cin = 1, h = H, w = W;
    for (int l = 0; l < L; ++l) {
        for (int i = 0; i < a->n_block[l]; ++i) {
            const int stride = (l > 0 && i == 0) ? 2 : 1;
            const int ho = (h + stride - 1) / stride, wo = (w + stride - 1) / stride;
            float *out = (float *)oalloc(sizeof(float) * (size_t)N * ho * wo * a->n_filter[l]);
            if (!out) return -1;
            unit(x, N, h, w, cin, 3, stride, a->n_filter[l], &wp, out);
            if (i > 0) ofree(x);
            x = out; cin = a->n_filter[l]; h = ho; w = wo;
        }
        feat[l] = (float *)x; fh[l] = h; fw[l] = w;
    }
    float *up = feat[L - 1];
    for (int l = L - 2; l >= 0; --l) {
        const int nf = a->n_filter[l], nfu = a->n_filter[l + 1];
        const size_t npix = (size_t)N * fh[l] * fw[l];
        float *t = (float *)oalloc(sizeof(float) * npix * nf);
        float *cat = (float *)oalloc(sizeof(float) * npix * 2 * nf);
        if (!t || !cat) return -1;
        conv2d_transpose_same(up, N, fh[l + 1], fw[l + 1], nfu, wp, 3, 2, nf, t);
        const float *bn = wp + (size_t)9 * nf * nfu;
        bn_relu(t, npix, nf, bn);
        wp = bn + 4 * (size_t)nf;
        copy_channels(feat[l], npix, nf, cat, 2 * nf, 0);      /* skip first (network_ao.py:51) */
        copy_channels(t, npix, nf, cat, 2 * nf, nf);
        ofree(t); ofree(up); ofree(feat[l]);
        const float *y = cat; int c = 2 * nf;
        for (int i = 0; i < a->n_block[l]; ++i) {
            float *out = (float *)oalloc(sizeof(float) * npix * nf);
            if (!out) return -1;
            unit(y, N, fh[l], fw[l], c, 3, 1, nf, &wp, out);
            ofree(y);
            y = out; c = nf;
        }
        up = (float *)y;
    }
    const size_t npix = (size_t)N * H * W;
    float *lg = logits_out ? logits_out : (float *)oalloc(sizeof(float) * npix * a->n_class);
    if (!lg) return -1;
    conv2d_same(up, N, H, W, a->n_filter[0], wp, 1, 1, a->n_class, lg);
    const float *bias = wp + (size_t)a->n_filter[0] * a->n_class;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i)
        for (int c = 0; c < a->n_class; ++c) lg[i * a->n_class + c] += bias[c];
    ofree(up);
    softmax_argmax(lg, npix, a->n_class, prob_out, pred_out);
    if (!logits_out) ofree(lg);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
