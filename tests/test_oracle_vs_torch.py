"""The numpy oracle against an independently written torch-CPU formulation, and
against hand-worked known answers."""
import numpy as np
import pytest
import torch

from oracle import fcn_oracle as O
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.weights import synthetic_params
from oracle import torch_oracle as T
from ukbb_cardiac_amd.phantom import cine_phantom


def test_same_pads_known_answers():
    # 3x3 s1 -> 1/1 ; 3x3 s2 on even input -> 0 before / 1 after ; 1x1 -> none
    assert O.same_pads(208, 3, 1) == (208, 1, 1)
    assert O.same_pads(208, 3, 2) == (104, 0, 1)
    assert O.same_pads(13, 3, 2) == (7, 1, 1)
    assert O.same_pads(192, 1, 1) == (192, 0, 0)


def test_conv_stride2_window_starts_at_2i():
    x = np.arange(8, dtype=np.float64).reshape(1, 1, 8, 1)
    w = np.zeros((1, 3, 1, 1)); w[0, 0, 0, 0] = 1.0       # picks the first tap
    y = O.conv2d_same(x, w, 2)[0, 0, :, 0]
    assert y.tolist() == [0, 2, 4, 6]


def test_transposed_conv_1d_worked_example():
    # f=2, w=[.5,1,.5], x=[a,b] -> [.5a, a, .5a+.5b, b]
    a, b = 3.0, 5.0
    x = np.array([a, b]).reshape(1, 1, 2, 1)
    w = np.array([.5, 1, .5]).reshape(1, 3, 1, 1)
    y = O.conv2d_transpose_same(x, w, 2)[0, 0, :, 0]     # H dim: k=1,s=2 handled too
    # H has k=1 < s: rows 0 gets data, row 1 zero
    assert np.allclose(y, [.5 * a, a, .5 * a + .5 * b, b])


@pytest.mark.parametrize('f', [2, 4, 8, 16])
def test_upsample_dense_vs_separable_vs_torch(f):
    rng = np.random.default_rng(f)
    x = rng.normal(size=(2, 3, 5, 4))
    d = O.transpose_upsample2d(x, f)
    s = O.transpose_upsample2d_separable(x, f)
    t = T.upsample(torch.from_numpy(x).permute(0, 3, 1, 2), f, torch.float64).permute(0, 2, 3, 1).numpy()
    assert d.shape == (2, 3 * f, 5 * f, 4)
    assert np.allclose(d, s, atol=1e-12)
    assert np.allclose(d, t, atol=1e-12)
    # interior weights sum to one (constant image stays constant away from borders)
    c = O.transpose_upsample2d(np.ones((1, 4, 4, 1)), f)[0, f:-f, f:-f, 0]
    assert np.allclose(c, 1.0)


@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('hw', [(12, 13), (16, 16), (7, 9)])
def test_conv_vs_torch(stride, hw):
    rng = np.random.default_rng(5)
    x = rng.normal(size=(2, hw[0], hw[1], 3))
    w = rng.normal(size=(3, 3, 3, 4))
    a = O.conv2d_same(x, w, stride)
    b = T.conv_same(torch.from_numpy(x).permute(0, 3, 1, 2), w, stride, torch.float64).permute(0, 2, 3, 1).numpy()
    assert a.shape == b.shape and np.allclose(a, b, atol=1e-12)


@pytest.mark.parametrize('hw', [(6, 5), (8, 8)])
def test_conv_transpose_vs_torch(hw):
    rng = np.random.default_rng(6)
    x = rng.normal(size=(2, hw[0], hw[1], 3))
    w = rng.normal(size=(3, 3, 4, 3))                    # [kh,kw,Cout,Cin]
    a = O.conv2d_transpose_same(x, w, 2)
    b = T.conv_transpose_same(torch.from_numpy(x).permute(0, 3, 1, 2), w, 2, torch.float64).permute(0, 2, 3, 1).numpy()
    assert a.shape == (2, 2 * hw[0], 2 * hw[1], 4) and np.allclose(a, b, atol=1e-12)


@pytest.mark.parametrize('name,hw', [('FCN_sa', (32, 48)), ('FCN_la_4ch_seg4', (16, 32))])
def test_fcn_graph_vs_torch(name, hw):
    arch = MODELS[name]
    params = synthetic_params(arch, 1234)
    img = cine_phantom(2, hw[0], hw[1], seed=3)
    a = O.build_FCN(img, params, arch.n_class, dtype=np.float64)
    b = T.fcn_forward(img, params, arch, torch.float64)
    assert a.shape == (2, hw[0], hw[1], arch.n_class)
    assert np.allclose(a, b, rtol=1e-10, atol=1e-10)
    # fp32 evaluation stays within the north-star tolerance of the fp64 one
    c = O.build_FCN(img, params, arch.n_class, dtype=np.float32)
    assert np.max(np.abs(c - a)) <= 1e-3 * np.max(np.abs(a))
    # several classes are populated with the synthetic weights
    assert len(np.unique(O.argmax_pred(a))) >= min(3, arch.n_class)


def test_unet_graph_vs_torch():
    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    img = np.random.default_rng(2).normal(size=(1, 32, 32, 1)).astype(np.float32)
    a = O.UNet(img, params, arch.n_class, n_block=arch.n_block, dtype=np.float64)
    b = T.unet_forward(img, params, arch, torch.float64)
    assert a.shape == (1, 32, 32, 3)
    assert np.allclose(a, b, rtol=1e-10, atol=1e-10)


def test_unet_lstm_graph_vs_torch():
    """BiConvLSTM head: numpy restatement vs an independent torch formulation."""
    from oracle.torch_oracle import unet_lstm_forward
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 1234)
    x = np.random.default_rng(8).standard_normal((2, 9, 32, 16, 1)).astype(np.float32)
    a = O.unet_lstm(x, params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)
    b = unet_lstm_forward(x, params, arch)
    assert a.shape == (2, 9, 32, 16, 3)
    assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    # a window is order-sensitive: reversing time must change the result (fw/bw weights differ)
    c = O.unet_lstm(x[:, ::-1], params, arch.n_hidden, n_block=arch.n_block, dtype=np.float64)[:, ::-1]
    assert np.abs(a - c).max() > 1e-3


def test_aortic_window_known_answers():
    """deploy_network_ao.py:130-158: 9 weights (1-|t-4|/5)^0.1 and the circular window."""
    w = O.aortic_window_weights(5, 0.1)
    assert len(w) == 9 and w[4] == 1.0 and abs(w[0] - 0.2 ** 0.1) < 1e-15 and np.allclose(w, w[::-1])
    assert O.aortic_window_indices(0, 50) == [46, 47, 48, 49, 0, 1, 2, 3, 4]
    assert O.aortic_window_indices(48, 50) == [44, 45, 46, 47, 48, 49, 0, 1, 2]


def test_softmax_argmax_ties_lowest_index():
    logits = np.array([[1.0, 3.0, 3.0, 0.0]])
    prob, pred = O.prob_pred(logits)
    assert pred.dtype == np.int32 and pred[0] == 1
    assert np.isclose(prob.sum(), 1.0)


# ---- checks that do NOT share the oracle's SAME-pad formula ---------------------------------------
# oracle/torch_oracle.py computes its pads with the same closed form as oracle/fcn_oracle.py (`_same_pad` == `same_pads`),
# so the comparisons above cross-check op mechanics, not the padding rule.  Two things torch can say on its own:

@pytest.mark.parametrize('k,cin,cout,hw', [(3, 5, 7, (12, 13)), (1, 6, 4, (9, 16)), (3, 1, 16, (16, 32))])
def test_stride1_same_against_torch_own_same_rule(k, cin, cout, hw):
    """F.conv2d(padding='same') is torch's OWN 'same' rule (stride 1 only): the stride-1 3x3 and 1x1 layers of
    network.py:19-25 (17 of the 21 conv layers of build_FCN) agree with it."""
    import torch.nn.functional as F
    rng = np.random.default_rng(k * 100 + cin)
    x = rng.standard_normal((2,) + hw + (cin,))
    w = rng.standard_normal((k, k, cin, cout))
    want = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1), padding='same')
    got = O.conv2d_same(x, w, 1)
    assert np.abs(got - want.permute(0, 2, 3, 1).numpy()).max() < 1e-12


@pytest.mark.parametrize('k,s,hw', [(3, 2, (6, 5)), (3, 2, (8, 13)), (7, 4, (3, 4)), (31, 16, (2, 3))])
def test_transposed_conv_is_the_autograd_gradient_of_the_forward_same_conv(k, s, hw):
    """App. B.4 says conv2d_transpose(SAME) IS the gradient of the forward SAME conv with respect to its input.  Here the
    gradient is taken by torch.autograd of the forward conv (not by a transposed-conv primitive and not by the oracle's
    scatter-and-crop), so the crop offsets of O.conv2d_transpose_same follow from the forward padding rule alone:
    given App. B.1 for the forward conv, B.4 needs no separate recollection."""
    import torch.nn.functional as F
    rng = np.random.default_rng(k + s)
    cin, cout = 3, 2                                             # transposed conv: cin -> cout, filter [k,k,cout,cin]
    x = rng.standard_normal((2,) + hw + (cin,))
    w = rng.standard_normal((k, k, cout, cin))
    H, W = hw[0] * s, hw[1] * s
    big = torch.zeros((2, cout, H, W), dtype=torch.float64, requires_grad=True)
    _, pt, pb = O.same_pads(H, k, s)
    _, pl, pr = O.same_pads(W, k, s)
    # forward conv cout -> cin with HWIO filter w[k,k,cout,cin]
    y = F.conv2d(F.pad(big, (pl, pr, pt, pb)), torch.from_numpy(w).permute(3, 2, 0, 1), stride=s)
    assert y.shape[2:] == hw
    y.backward(torch.from_numpy(x).permute(0, 3, 1, 2))
    got = O.conv2d_transpose_same(x, w, s)
    assert np.abs(got - big.grad.permute(0, 2, 3, 1).numpy()).max() < 1e-12
