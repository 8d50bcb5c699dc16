"""Independent torch-CPU formulation of the same graphs (second opinion for the
numpy oracle).  Written against torch.nn.functional
primitives (NCHW, asymmetric F.pad + F.conv2d, F.conv_transpose2d + crop), i.e.
a different code path from oracle/fcn_oracle.py's pad+tensordot / scatter.
What it does NOT cross-check: the SAME-padding rule itself -- ``_same_pad`` below is the same closed form as
``fcn_oracle.same_pads`` ([TF-recall]).  The checks that are independent of that formula live in
tests/test_oracle_vs_torch.py: torch's own ``padding='same'`` for the stride-1 layers, and torch.autograd's gradient of
the forward conv for the transposed convs (so App. B.4 follows from B.1); the stride-2 forward rule ("extra pixel after")
has no second source in this image.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ and by
bench.py's ``cpu_baseline`` leg, where ``TorchFCN`` -- this graph with its
weights converted once, fp32, all host threads -- is the stand-in SURVEY.md
section 8(d) specifies for the reference's TF-CPU ``deploy_network.py`` (TensorFlow
itself is not installable here).  PARITY UNPINNED vs TensorFlow like the rest."""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv_same(x, w_hwio, stride, dtype):
    w = _t(w_hwio, dtype).permute(3, 2, 0, 1).contiguous()
    pt, pb = _same_pad(x.shape[2], w.shape[2], stride)
    pl, pr = _same_pad(x.shape[3], w.shape[3], stride)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, stride=stride)


def conv_transpose_same(x, w_hwoi, stride, dtype):
    # TF filter [kh,kw,Cout,Cin] -> torch conv_transpose2d weight [Cin,Cout,kh,kw]
    w = _t(w_hwoi, dtype).permute(3, 2, 0, 1).contiguous()
    full = F.conv_transpose2d(x, w, stride=stride)
    H, W = x.shape[2] * stride, x.shape[3] * stride
    pt, _ = _same_pad(H, w.shape[2], stride)
    pl, _ = _same_pad(W, w.shape[3], stride)
    return full[:, :, pt:pt + H, pl:pl + W]


def bn_relu(x, p, dtype):
    g, b, m, v = (_t(p[k], dtype) for k in ('gamma', 'beta', 'mean', 'var'))
    y = F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)
    return F.relu(y)


def unit(x, p, stride, dtype):
    return bn_relu(conv_same(x, p['kernel'], stride, dtype), p, dtype)


def upsample(x, f, dtype):
    sz = 2 * f - 1
    c = (sz + 1) // 2
    h = torch.tensor(list(range(1, c + 1)) + list(range(c - 1, 0, -1)), dtype=torch.float32) / float(c)
    W2 = (h[:, None] * h[None, :]).to(dtype)
    C = x.shape[1]
    w = W2[None, None].repeat(C, 1, 1, 1)          # depthwise [C,1,k,k]
    full = F.conv_transpose2d(x, w, stride=f, groups=C)
    H, Wd = x.shape[2] * f, x.shape[3] * f
    pt, _ = _same_pad(H, sz, f)
    pl, _ = _same_pad(Wd, sz, f)
    return full[:, :, pt:pt + H, pl:pl + Wd]


def fcn_forward(image_nhwc, params, arch, dtype=torch.float64):
    x = _t(image_nhwc, dtype).permute(0, 3, 1, 2)
    feats = []
    for l in range(arch.n_level):
        x = unit(x, params['conv%d_0' % l], 1 if l == 0 else 2, dtype)
        for i in range(1, arch.n_block[l]):
            x = unit(x, params['conv%d_%d' % (l, i)], 1, dtype)
        feats.append(x)
    ups = []
    for l in range(arch.n_level):
        s = unit(feats[l], params['same_dim%d' % l], 1, dtype)
        ups.append(s if l == 0 else upsample(s, 2 ** l, dtype))
    x = torch.cat(ups, dim=1)
    x = unit(x, params['out0'], 1, dtype)
    x = unit(x, params['out1'], 1, dtype)
    p = params['logits']
    y = conv_same(x, p['kernel'], 1, dtype) + _t(p['bias'], dtype)[None, :, None, None]
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def unet_forward(image_nhwc, params, arch, dtype=torch.float64):
    x = _t(image_nhwc, dtype).permute(0, 3, 1, 2)
    feats = []
    for l in range(arch.n_level):
        x = unit(x, params['conv%d_0' % l], 1 if l == 0 else 2, dtype)
        for i in range(1, arch.n_block[l]):
            x = unit(x, params['conv%d_%d' % (l, i)], 1, dtype)
        feats.append(x)
    up = feats[-1]
    for l in range(arch.n_level - 2, -1, -1):
        p = params['up%d_t' % l]
        x = bn_relu(conv_transpose_same(up, p['kernel'], 2, dtype), p, dtype)
        x = torch.cat([feats[l], x], dim=1)
        for i in range(arch.n_block[l]):
            x = unit(x, params['up%d_%d' % (l, i)], 1, dtype)
        up = x
    if 'logits' not in params:                       # UNet-LSTM: the feature map itself (NCHW tensor)
        return up
    p = params['logits']
    y = conv_same(up, p['kernel'], 1, dtype) + _t(p['bias'], dtype)[None, :, None, None]
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def unet_lstm_forward(image_nthwc, params, arch, dtype=torch.float64):
    """Independent formulation of UNet_LSTM_Model (common/network_ao.py:322-399, bidirectional): torch convs,
    gates via chunk(4) in the order i, j, f, o, forget bias 1.0 [TF-recall]."""
    N, T, H, W, C = image_nthwc.shape
    feats = unet_forward(np.asarray(image_nthwc).reshape(N * T, H, W, C), params, arch, dtype)   # (N*T, 16, H, W)
    feats = feats.reshape(N, T, feats.shape[1], H, W)
    nh = arch.n_hidden

    def run(direction, order):
        p = params[direction]
        b = _t(p['bias'], dtype)[None, :, None, None]
        h = torch.zeros((N, nh, H, W), dtype=dtype)
        c = torch.zeros_like(h)
        out = {}
        for t in order:
            z = conv_same(torch.cat([feats[:, t], h], dim=1), p['kernel'], 1, dtype) + b
            i, j, f, o = torch.chunk(z, 4, dim=1)
            c = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
            h = torch.tanh(c) * torch.sigmoid(o)
            out[t] = h
        return out
    fw = run('lstm_fw', range(T))
    bw = run('lstm_bw', range(T - 1, -1, -1))
    po = params['lstm_out']
    ys = [conv_same(torch.cat([fw[t], bw[t]], dim=1), po['kernel'], 1, dtype) + _t(po['bias'], dtype)[None, :, None, None]
          for t in range(T)]
    return torch.stack(ys, dim=1).permute(0, 1, 3, 4, 2).contiguous().numpy()


class TorchFCN:
    """build_FCN (common/network.py:170-230 with common/train_network.py:156-199) as an op-by-op torch-CPU graph --
    conv, batch_norm (moving statistics), relu as separate ops like the TF graph -- with the weights converted to
    torch layout ONCE (as a restored TF session holds them).  ``__call__(image[N,H,W,1] f32) -> pred[N,H,W] int32``
    is the ``sess.run('pred:0', ...)`` of common/deploy_network.py:110-111."""

    def __init__(self, params, arch, dtype=torch.float32):
        self.arch, self.dtype = arch, dtype
        self.p = {}
        for name, q in params.items():
            e = {'w': _t(q['kernel'], dtype).permute(3, 2, 0, 1).contiguous()}
            for k in ('gamma', 'beta', 'mean', 'var', 'bias'):
                if k in q:
                    e[k] = _t(q[k], dtype)
            self.p[name] = e
        self.up_w = {}
        for l in range(1, arch.n_level):
            f = 2 ** l
            sz = 2 * f - 1
            c = (sz + 1) // 2
            h = torch.tensor(list(range(1, c + 1)) + list(range(c - 1, 0, -1)), dtype=torch.float32) / float(c)
            self.up_w[l] = (h[:, None] * h[None, :]).to(dtype)[None, None].repeat(arch.same_dim, 1, 1, 1).contiguous()

    def _unit(self, x, name, stride):
        e = self.p[name]
        pt, pb = _same_pad(x.shape[2], e['w'].shape[2], stride)
        pl, pr = _same_pad(x.shape[3], e['w'].shape[3], stride)
        y = F.conv2d(F.pad(x, (pl, pr, pt, pb)) if (pt or pb or pl or pr) else x, e['w'], stride=stride)
        return F.relu(F.batch_norm(y, e['mean'], e['var'], e['gamma'], e['beta'], training=False, eps=BN_EPS))

    def logits(self, image_nhwc):
        a = self.arch
        with torch.no_grad():
            x = torch.from_numpy(np.ascontiguousarray(image_nhwc)).to(self.dtype).permute(0, 3, 1, 2)
            ups = []
            for l in range(a.n_level):
                x = self._unit(x, 'conv%d_0' % l, 1 if l == 0 else 2)
                for i in range(1, a.n_block[l]):
                    x = self._unit(x, 'conv%d_%d' % (l, i), 1)
                s = self._unit(x, 'same_dim%d' % l, 1)
                if l:
                    f = 2 ** l
                    full = F.conv_transpose2d(s, self.up_w[l], stride=f, groups=a.same_dim)
                    H, W = s.shape[2] * f, s.shape[3] * f
                    pt, _ = _same_pad(H, 2 * f - 1, f)
                    pl, _ = _same_pad(W, 2 * f - 1, f)
                    s = full[:, :, pt:pt + H, pl:pl + W]
                ups.append(s)
            x = torch.cat(ups, dim=1)
            x = self._unit(x, 'out0', 1)
            x = self._unit(x, 'out1', 1)
            e = self.p['logits']
            return F.conv2d(x, e['w']) + e['bias'][None, :, None, None]

    def __call__(self, image_nhwc):
        return torch.argmax(self.logits(image_nhwc), dim=1).to(torch.int32).numpy()
