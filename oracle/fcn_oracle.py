"""numpy restatement of the reference's FCN / U-Net inference graph (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED vs
TensorFlow (TF 1.x is absent; semantics marked [TF-recall] follow SURVEY.md
Appendix B and are cross-checked against an independent torch-CPU formulation
in ``tests/test_oracle_vs_torch.py``).

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  All tensors are NHWC, kernels HWIO, exactly as in the
reference.  ``dtype`` selects float64 ("truth") or float32 evaluation.

Parameter container (builder's choice; the reference stores TF variables):
``params[name] -> {"kernel", "gamma", "beta", "mean", "var"}`` for a
conv+BN+ReLU unit, ``{"kernel", "bias"}`` for the logits layer.
"""
import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon [TF-recall]


# --------------------------------------------------------------------------
# TF op semantics
# --------------------------------------------------------------------------
def same_pads(n_in, k, s):
    """TF 'SAME' padding for one spatial dim [TF-recall, SURVEY App. B.1].

    out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0);
    pad_before = pad_total // 2 (the extra pixel goes AFTER).
    """
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    before = total // 2
    return n_out, before, total - before


def conv2d_same(x, w, stride=1):
    """tf.layers.conv2d(..., padding='same', use_bias=False)  (common/network.py:21-22).

    x: [N,H,W,Cin]; w: [kh,kw,Cin,Cout] (HWIO), cross-correlation, no flip.
    """
    n, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    assert cin == cin2
    ho, pt, pb = same_pads(h, kh, stride)
    wo, pl, pr = same_pads(wd, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride, :]
            out += np.tensordot(patch, w[i, j].astype(x.dtype), axes=([3], [0]))
    return out


def conv2d_transpose_same(x, w, stride):
    """tf.nn.conv2d_transpose / tf.layers.conv2d_transpose, padding 'SAME',
    output size = in*stride (common/network.py:30,163) [TF-recall, App. B.4].

    x: [N,h,w,Cin]; w: [kh,kw,Cout,Cin] (TF transposed-conv filter layout).
    Defined as the gradient of the forward SAME conv from size in*stride to
    in: full scatter of length (in-1)*s+k, then crop pad_before from the start,
    where pad_before is the *forward* conv's pad_before.
    """
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert cin == cin2
    s = stride
    H, W = h * s, wd * s
    _, pt, _ = same_pads(H, kh, s)
    _, pl, _ = same_pads(W, kw, s)
    fh, fw = (h - 1) * s + kh, (wd - 1) * s + kw
    # full may be shorter than crop window when k < s; size generously
    full = np.zeros((n, max(fh, pt + H), max(fw, pl + W), cout), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            full[:, i:i + (h - 1) * s + 1:s, j:j + (wd - 1) * s + 1:s, :] += \
                np.tensordot(x, w[i, j].astype(x.dtype), axes=([3], [1]))
    return full[:, pt:pt + H, pl:pl + W, :]


def batch_norm_infer(x, gamma, beta, mean, var, eps=BN_EPS):
    """tf.layers.batch_normalization(training=False) (common/network.py:23)
    [TF-recall, App. B.2]: y = gamma*(x-mean)/sqrt(var+eps)+beta, axis=-1."""
    dt = x.dtype
    inv = gamma.astype(dt) / np.sqrt(var.astype(dt) + dt.type(eps))
    return (x - mean.astype(dt)) * inv + beta.astype(dt)


def relu(x):
    return np.maximum(x, 0)


def conv2d_bn_relu(x, p, kernel_size=3, strides=1):
    """common/network.py:19-25."""
    assert p["kernel"].shape[0] == kernel_size
    y = conv2d_same(x, p["kernel"], strides)
    y = batch_norm_infer(y, p["gamma"], p["beta"], p["mean"], p["var"])
    return relu(y)


def conv2d_transpose_bn_relu(x, p, kernel_size=3, strides=1):
    """common/network.py:28-34."""
    assert p["kernel"].shape[0] == kernel_size
    y = conv2d_transpose_same(x, p["kernel"], strides)
    y = batch_norm_infer(y, p["gamma"], p["beta"], p["mean"], p["var"])
    return relu(y)


def linear_1d(sz):
    """common/network.py:117-124."""
    if sz % 2 == 0:
        raise NotImplementedError('`Linear kernel` requires odd filter size.')
    c = (sz + 1) // 2
    h = np.array(list(range(1, c + 1)) + list(range(c - 1, 0, -1)), dtype=np.float32)
    h /= float(c)
    return h


def linear_2d(sz):
    """common/network.py:127-135 (separable: W = h (x) h)."""
    h = linear_1d(sz)
    return (h[:, None] * h[None, :]).astype(np.float32)


def transpose_upsample2d(x, factor):
    """common/network.py:138-167, executed the way the reference does: a DENSE
    [sz,sz,n,n] filter that is diagonal in the channel dims, through the
    generic transposed conv."""
    sz = factor * 2 - 1
    W = linear_2d(sz)
    n = x.shape[3]
    filt = np.zeros((sz, sz, n, n), dtype=np.float32)
    for i in range(n):
        filt[:, :, i, i] = W
    return conv2d_transpose_same(x, filt, factor)


def transpose_upsample2d_separable(x, factor):
    """Same result as ``transpose_upsample2d`` up to rounding, computed per
    channel with the <=2x2 contributing taps (what the HIP head kernel does).
    out[o] = sum_i x[i]*h[o + pb - i*f], pb = (f-1)//2, h = triangle/f."""
    n, h, w, c = x.shape
    f = factor

    def taps(n_in):
        o = np.arange(n_in * f)
        pb = (f - 1) // 2
        i1 = (o + pb) // f
        j1 = (o + pb) - i1 * f
        w1 = (j1 + 1) / f
        w0 = (f - 1 - j1) / f
        i0 = i1 - 1
        w1 = np.where(i1 < n_in, w1, 0.0)
        w0 = np.where(i0 >= 0, w0, 0.0)
        return np.clip(i0, 0, n_in - 1), w0.astype(x.dtype), np.clip(i1, 0, n_in - 1), w1.astype(x.dtype)

    y0, wy0, y1, wy1 = taps(h)
    x0, wx0, x1, wx1 = taps(w)
    rows = x[:, y0] * wy0[None, :, None, None] + x[:, y1] * wy1[None, :, None, None]
    out = rows[:, :, x0] * wx0[None, None, :, None] + rows[:, :, x1] * wx1[None, None, :, None]
    return out


def softmax(logits):
    """tf.nn.softmax(logits) over the last axis (common/train_network.py:198)."""
    m = logits.max(axis=-1, keepdims=True)
    e = np.exp(logits - m)
    return e / e.sum(axis=-1, keepdims=True)


def argmax_pred(prob_or_logits):
    """tf.cast(tf.argmax(prob, -1), int32) (common/train_network.py:199);
    lowest index wins on exact ties [TF-recall, App. B.5] (numpy does the same)."""
    return np.argmax(prob_or_logits, axis=-1).astype(np.int32)


# --------------------------------------------------------------------------
# Graphs
# --------------------------------------------------------------------------
def build_FCN(image, params, n_class, n_level=5, n_filter=(16, 32, 64, 128, 256),
              n_block=(2, 2, 3, 3, 3), same_dim=32, fc=64, dtype=np.float64, return_net=False):
    """common/network.py:170-230 with the hyper-parameters bound in
    common/train_network.py:174-195.  image: [N,H,W,1]; returns logits
    [N,H,W,n_class]."""
    x = np.asarray(image, dtype=dtype)
    net = {}
    for l in range(n_level):                                     # network.py:179-189
        strides = 1 if l == 0 else 2
        x = conv2d_bn_relu(x, params['conv%d_0' % l], 3, strides)
        for i in range(1, n_block[l]):
            x = conv2d_bn_relu(x, params['conv%d_%d' % (l, i)], 3)
        net['conv%d' % l] = x
    for l in range(n_level):                                     # network.py:201-204
        net['conv%d_same_dim' % l] = conv2d_bn_relu(net['conv%d' % l], params['same_dim%d' % l], 1)
    net['conv0_up'] = net['conv0_same_dim']                      # network.py:207-211
    for l in range(1, n_level):
        net['conv%d_up' % l] = transpose_upsample2d(net['conv%d_same_dim' % l], 2 ** l)
    net['concat'] = np.concatenate([net['conv%d_up' % l] for l in range(n_level)], axis=-1)  # :214-218
    x = conv2d_bn_relu(net['concat'], params['out0'], 1)         # network.py:227
    x = conv2d_bn_relu(x, params['out1'], 1)                     # network.py:228
    net['out1'] = x
    p = params['logits']                                         # network.py:229 (use_bias default True)
    logits = conv2d_same(x, p['kernel'], 1) + p['bias'].astype(dtype)
    assert logits.shape[-1] == n_class
    if return_net:
        return logits, net
    return logits


def UNet(images, params, n_class=3, n_level=5, n_filter=(16, 32, 64, 128, 256),
         n_block=(2, 2, 2, 2, 2), dtype=np.float64, return_net=False):
    """common/network_ao.py:18-64 with common/train_network_ao.py:268,275-284."""
    x = np.asarray(images, dtype=dtype)
    net = {}
    for l in range(n_level):                                     # network_ao.py:31-41
        strides = 1 if l == 0 else 2
        x = conv2d_bn_relu(x, params['conv%d_0' % l], 3, strides)
        for i in range(1, n_block[l]):
            x = conv2d_bn_relu(x, params['conv%d_%d' % (l, i)], 3)
        net['conv%d' % l] = x
    l = n_level - 1                                              # network_ao.py:44-46
    net['conv%d_up' % l] = net['conv%d' % l]
    for l in range(n_level - 2, -1, -1):                         # network_ao.py:48-55
        x = conv2d_transpose_bn_relu(net['conv%d_up' % (l + 1)], params['up%d_t' % l], 3, 2)
        x = np.concatenate([net['conv%d' % l], x], axis=-1)      # skip first (:51)
        for i in range(n_block[l]):
            x = conv2d_bn_relu(x, params['up%d_%d' % (l, i)], 3)
        net['conv%d_up' % l] = x
    p = params['logits']                                         # network_ao.py:63
    logits = conv2d_same(net['conv0_up'], p['kernel'], 1) + p['bias'].astype(dtype)
    assert logits.shape[-1] == n_class
    if return_net:
        return logits, net
    return logits


def unet_features(images, params, n_level=5, n_filter=(16, 32, 64, 128, 256), n_block=(2, 2, 2, 2, 2), dtype=np.float64):
    """net['conv0_up'] of common/network_ao.py:18-55 -- what UNet_LSTM_Model feeds to the LSTM (:343-347)."""
    x = np.asarray(images, dtype=dtype)
    net = {}
    for l in range(n_level):
        x = conv2d_bn_relu(x, params['conv%d_0' % l], 3, 1 if l == 0 else 2)
        for i in range(1, n_block[l]):
            x = conv2d_bn_relu(x, params['conv%d_%d' % (l, i)], 3)
        net['conv%d' % l] = x
    up = net['conv%d' % (n_level - 1)]
    for l in range(n_level - 2, -1, -1):
        x = conv2d_transpose_bn_relu(up, params['up%d_t' % l], 3, 2)
        x = np.concatenate([net['conv%d' % l], x], axis=-1)
        for i in range(n_block[l]):
            x = conv2d_bn_relu(x, params['up%d_%d' % (l, i)], 3)
        up = x
    return up


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def conv_lstm_cell(x, h, c, p, forget_bias=1.0):
    """One step of tf.contrib.rnn.Conv2DLSTMCell(kernel_shape=[3,3]) as used at common/network_ao.py:225,276,288
    [TF-recall, SURVEY.md App. B.6]: one SAME conv over concat([x, h]) -> 4*hidden channels (+bias),
    split in the order (input gate i, new input j, forget gate f, output gate o);
    c' = sigmoid(f + forget_bias) * c + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o).  No peepholes."""
    dt = x.dtype
    z = conv2d_same(np.concatenate([x, h], axis=-1), p['kernel'], 1) + p['bias'].astype(dt)
    i, j, f, o = np.split(z, 4, axis=-1)
    c_new = sigmoid(f + dt.type(forget_bias)) * c + sigmoid(i) * np.tanh(j)
    h_new = np.tanh(c_new) * sigmoid(o)
    return h_new, c_new


def biconv_lstm(features, params, n_hidden):
    """BiConv_LSTM, common/network_ao.py:255-319.  features [N,T,H,W,C] -> logits [N,T,H,W,n_class].
    Zero initial state (:278,:290); forward over t = 0..T-1, backward over t = T-1..0; the output conv sees
    concat([h_fw[t], h_bw[t]]) (:305: cell_outputs_bw is stored in processing order, hence the index
    n_step-1-t)."""
    N, T, H, W, _ = features.shape
    dt = features.dtype
    zeros = np.zeros((N, H, W, n_hidden), dt)
    h, c, fw = zeros, zeros, []
    for t in range(T):
        h, c = conv_lstm_cell(features[:, t], h, c, params['lstm_fw'])
        fw.append(h)
    h, c, bw = zeros, zeros, [None] * T
    for t in range(T - 1, -1, -1):
        h, c = conv_lstm_cell(features[:, t], h, c, params['lstm_bw'])
        bw[t] = h
    po = params['lstm_out']
    outs = [conv2d_same(np.concatenate([fw[t], bw[t]], axis=-1), po['kernel'], 1) + po['bias'].astype(dt) for t in range(T)]
    return np.stack(outs, axis=1)


def conv_lstm(features, params, n_hidden):
    """Conv_LSTM, common/network_ao.py:214-252 (the single-direction head, UNet_LSTM_Model with bidirectional=False :349-352):
    one cell from the zero state (:228) over t = 0..T-1, the 1x1 logits conv on every step's cell output (:244).
    params['lstm'] = the cell's {'kernel' [3,3,C+h,4h], 'bias'}, params['lstm_conv'] = {'kernel' [1,1,h,n_class], 'bias'}."""
    N, T, H, W, _ = features.shape
    dt = features.dtype
    h = c = np.zeros((N, H, W, n_hidden), dt)
    po = params['lstm_conv']
    outs = []
    for t in range(T):
        h, c = conv_lstm_cell(features[:, t], h, c, params['lstm'])
        outs.append(conv2d_same(h, po['kernel'], 1) + po['bias'].astype(dt))
    return np.stack(outs, axis=1)


def unet_lstm(images, params, n_hidden=16, n_level=5, n_filter=(16, 32, 64, 128, 256), n_block=(2, 2, 2, 2, 2),
              dtype=np.float64, bidirectional=True):
    """UNet_LSTM_Model inference graph, common/network_ao.py:322-399: images [N,T,H,W,1] -> logits [N,T,H,W,n_class];
    prob = softmax, pred = argmax (:396-397).  bidirectional (:349-352) picks BiConv_LSTM (the released model) or Conv_LSTM."""
    x = np.asarray(images, dtype=dtype)
    N, T, H, W, C = x.shape
    feats = unet_features(x.reshape(N * T, H, W, C), params, n_level, n_filter, n_block, dtype)
    head = biconv_lstm if bidirectional else conv_lstm
    return head(feats.reshape(N, T, H, W, feats.shape[-1]), params, n_hidden)


def aortic_lstm_prob_sequence(image, forward_seq, weight_R=5, weight_r=0.1, time_step=1, n_class=3):
    """'UNet-LSTM' branch of common/deploy_network_ao.py:99-107,129-183 on a normalised (X,Y,Z,T) volume:
    fixed 256x256 pad, circular 9-frame windows, weighted tiling of the window probabilities.
    ``forward_seq(image_idx[N,T,256,256,1] f32) -> prob[N,T,256,256,C]`` stands for sess.run (:171-172)."""
    X, Y, Z, T = image.shape
    prob = np.zeros((X, Y, Z, T, n_class), dtype=np.float32)
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_to_fixed(X, Y)
    image = np.pad(image, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    time_window = weight_R * 2 - 1
    weight = np.zeros((1, 1, 1, T, 1))
    w = np.reshape(aortic_window_weights(weight_R, weight_r), (1, 1, 1, time_window, 1))
    for t in range(0, T, time_step):
        idx = aortic_window_indices(t, T, weight_R)
        image_idx = np.transpose(image[:, :, :, idx], axes=(2, 3, 0, 1)).astype(np.float32)
        image_idx = np.expand_dims(image_idx, axis=-1)
        prob_idx = np.transpose(forward_seq(image_idx), axes=(2, 3, 0, 1, 4))
        prob[:, :, :, idx] += prob_idx[x_pre:x_pre + X, y_pre:y_pre + Y] * w
        weight[:, :, :, idx] += w
    prob /= weight
    return prob


def prob_pred(logits):
    """common/train_network.py:198-199 / common/network_ao.py:159-160."""
    prob = softmax(logits)
    return prob, argmax_pred(prob)


def top2_margin(logits):
    """Gap between the two largest logits per pixel (used by the tests to
    classify argmax disagreements as near-ties)."""
    s = np.sort(logits, axis=-1)
    return s[..., -1] - s[..., -2]


# --------------------------------------------------------------------------
# Host-side pre/post-processing of the deploy scripts
# --------------------------------------------------------------------------
def rescale_intensity(image, thres=(1.0, 99.0)):
    """common/image_utils.py:70-77 including the in-place clip quirk
    (SURVEY App. C.1): the caller's array is clipped."""
    val_l, val_h = np.percentile(image, thres)
    image2 = image
    image2[image < val_l] = val_l
    image2[image > val_h] = val_h
    image2 = (image2.astype(np.float32) - val_l) / (val_h - val_l)
    return image2


def normalise_intensity(image, thres_roi=10.0):
    """common/image_utils.py:60-67."""
    val_l = np.percentile(image, thres_roi)
    roi = (image >= val_l)
    mu, sigma = np.mean(image[roi]), np.std(image[roi])
    eps = 1e-6
    return (image - mu) / (sigma + eps)


def pad_to_multiple(X, Y, m=16):
    """common/deploy_network.py:97-99: centred zero pad up to a multiple of 16."""
    import math
    X2, Y2 = int(math.ceil(X / float(m))) * m, int(math.ceil(Y / float(m))) * m
    x_pre, y_pre = int((X2 - X) / 2), int((Y2 - Y) / 2)
    x_post, y_post = (X2 - X) - x_pre, (Y2 - Y) - y_pre
    return X2, Y2, x_pre, x_post, y_pre, y_post


def pad_to_fixed(X, Y, X2=256, Y2=256):
    """common/deploy_network_ao.py:105-107 (negative pads if X>256: np.pad raises)."""
    x_pre, y_pre = int((X2 - X) / 2), int((Y2 - Y) / 2)
    x_post, y_post = (X2 - X) - x_pre, (Y2 - Y) - y_pre
    return X2, Y2, x_pre, x_post, y_pre, y_post


def pick_es_frame(pred, seq_name, seg4=False):
    """common/deploy_network.py:125-130."""
    if seq_name == 'sa' or (seq_name == 'la_4ch' and seg4):
        return int(np.argmin(np.sum(pred == 1, axis=(0, 1, 2))))
    return int(np.argmax(np.sum(pred == 1, axis=(0, 1, 2))))


def deploy_sequence(image, forward, seq_name='sa', seg4=False):
    """The per-subject hot loop of common/deploy_network.py:83-131 on an
    in-memory (X,Y,Z,T) volume.  ``forward(image_fr[N,H,W,1] f32) -> pred[N,H,W]``
    stands for the sess.run call (:110-111).  Returns (pred float64 [X,Y,Z,T],
    clipped image (aliases input), ED index, ES index)."""
    X, Y, Z, T = image.shape
    orig_image = image
    image = rescale_intensity(image, (1, 99))
    pred = np.zeros(image.shape)
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_to_multiple(X, Y)
    image = np.pad(image, ((x_pre, x_post), (y_pre, y_post), (0, 0), (0, 0)), 'constant')
    for t in range(T):
        image_fr = image[:, :, :, t]
        image_fr = np.transpose(image_fr, axes=(2, 0, 1)).astype(np.float32)
        image_fr = np.expand_dims(image_fr, axis=-1)
        pred_fr = forward(image_fr)
        pred_fr = np.transpose(pred_fr, axes=(1, 2, 0))
        pred_fr = pred_fr[x_pre:x_pre + X, y_pre:y_pre + Y]
        pred[:, :, :, t] = pred_fr
    return pred, orig_image, 0, pick_es_frame(pred, seq_name, seg4)


def deploy_frame(image, forward):
    """ED/ES mode of common/deploy_network.py:163-200 on one in-memory (X,Y[,Z]) frame: 2-D frames gain a Z axis
    (:172-173), rescale_intensity (:179), centred zero pad to multiples of 16 (:185-188), (X2,Y2,Z) -> (Z,X2,Y2,1)
    float32 (:191-192), ``forward(image[N,H,W,1]) -> pred[N,H,W] int32`` for the sess.run (:195-196), transpose +
    crop (:199-200).  The result keeps TF's int32."""
    X, Y = image.shape[:2]
    if image.ndim == 2:
        image = np.expand_dims(image, axis=2)
    image = rescale_intensity(image, (1, 99))
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_to_multiple(X, Y)
    image = np.pad(image, ((x_pre, x_post), (y_pre, y_post), (0, 0)), 'constant')
    image = np.transpose(image, axes=(2, 0, 1)).astype(np.float32)
    image = np.expand_dims(image, axis=-1)
    pred = forward(image)
    pred = np.transpose(pred, axes=(1, 2, 0))
    return pred[x_pre:x_pre + X, y_pre:y_pre + Y]


def aortic_deploy_frame(image, forward, z_score=True):
    """ED/ES mode of common/deploy_network_ao.py:222-258: z-score (or rescale) normalisation (:233-236), pad to
    multiples of 16 -- not to the fixed 256 of the sequence mode -- (:240-243), (X,Y,Z) -> (Z,X2,Y2,1) float32
    (:247-249), ``forward`` for sess.run(['prob:0','pred:0']) (:253-254), transpose + crop (:257-258)."""
    X, Y = image.shape[:2]
    image = normalise_intensity(image, 10.0) if z_score else rescale_intensity(image, (1.0, 99.0))
    X2, Y2, x_pre, x_post, y_pre, y_post = pad_to_multiple(X, Y)
    image = np.pad(image, ((x_pre, x_post), (y_pre, y_post), (0, 0)), 'constant')
    image = np.transpose(image, axes=(2, 0, 1)).astype(np.float32)
    image = np.expand_dims(image, axis=-1)
    pred = forward(image)
    pred = np.transpose(pred, axes=(1, 2, 0))
    return pred[x_pre:x_pre + X, y_pre:y_pre + Y]


def aortic_window_weights(weight_R=5, weight_r=0.1):
    """common/deploy_network_ao.py:130-144."""
    time_window = weight_R * 2 - 1
    rad = int((time_window - 1) / 2)
    w = []
    for t in range(time_window):
        d = abs(t - rad)
        w += [pow(1 - float(d) / weight_R, weight_r) if d <= weight_R else 0]
    return np.array(w)


def aortic_window_indices(t, T, weight_R=5):
    """common/deploy_network_ao.py:147-158 (circular window)."""
    rad = int(((weight_R * 2 - 1) - 1) / 2)
    idx = []
    for i in range(t - rad, t + rad + 1):
        if i < 0:
            idx += [i + T]
        elif i >= T:
            idx += [i - T]
        else:
            idx += [i]
    return idx


def np_categorical_dice(pred, truth, k):
    """common/image_utils.py:171-175."""
    A = (pred == k).astype(np.float32)
    B = (truth == k).astype(np.float32)
    return 2 * np.sum(A * B) / (np.sum(A) + np.sum(B))
