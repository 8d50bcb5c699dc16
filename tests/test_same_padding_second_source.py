"""A second, independent source for the one TF rule the verdicts single out as having none: 'SAME' padding of a strided forward convolution
(3x3 stride 2 on an even size pads 0 before / 1 after).  The transformers package in this image carries Hugging Face's port of Google's
TensorFlow MobileNet checkpoints; to reproduce the TF outputs of those checkpoints in PyTorch it restates TF's rule as `apply_tf_padding`
(transformers/models/mobilenet_v1/modeling_mobilenet_v1.py, citing tensorflow.org/api_docs/python/tf/nn#notes_on_padding_2).  That function is third-party code
written against real TensorFlow weights and outputs -- not by this repository and not from the same recall.  The oracle's `same_pads` / `conv2d_same`
(oracle/fcn_oracle.py, following reference common/network.py:19-25) must agree with it.  This does not pin conv2d_transpose's crop, the BN epsilon or the ConvLSTM
cell: parity stays 'unpinned' for those."""
import numpy as np
import pytest

from oracle import fcn_oracle as O

torch = pytest.importorskip('torch')
mnv1 = pytest.importorskip('transformers.models.mobilenet_v1.modeling_mobilenet_v1')


def _hf_pads(n, k, s):
    conv = torch.nn.Conv2d(1, 1, kernel_size=(k, 1), stride=(s, 1), bias=False)
    x = torch.zeros((1, 1, n, 5))
    x[0, 0, :, 2] = torch.arange(1, n + 1, dtype=torch.float32)           # mark the rows: where they land tells pad_before
    y = mnv1.apply_tf_padding(x, conv)
    col = y[0, 0, :, 2].numpy()
    before = int(np.argmax(col == 1.0))
    after = len(col) - before - n
    return before, after


def test_same_pads_agree_with_the_huggingface_port_of_tensorflow_padding():
    for k in (1, 2, 3, 5, 7):
        for s in (1, 2, 3, 4):
            for n in range(max(k, 1), 70):
                n_out, before, after = O.same_pads(n, k, s)
                hb, ha = _hf_pads(n, k, s)
                assert (before, after) == (hb, ha), (n, k, s, before, after, hb, ha)
                assert n_out == -(-n // s) and (n + before + after - k) // s + 1 == n_out
    # the case every stride-2 layer of build_FCN / UNet hits (even sizes): nothing before, one after -- not PyTorch's padding=1
    assert O.same_pads(192, 3, 2) == (96, 0, 1) and O.same_pads(208, 3, 2) == (104, 0, 1) and O.same_pads(13, 3, 2) == (7, 1, 1)


@pytest.mark.parametrize('shape,k,s', [((2, 12, 16, 3), 3, 2), ((1, 13, 11, 2), 3, 2), ((1, 10, 14, 4), 3, 1), ((2, 9, 8, 3), 1, 1), ((1, 16, 12, 2), 5, 2)])
def test_conv2d_same_equals_torch_conv_on_the_tf_padded_input(shape, k, s):
    rng = np.random.default_rng(k * 10 + s)
    x = rng.standard_normal(shape)
    w = rng.standard_normal((k, k, shape[-1], 5))
    got = O.conv2d_same(x, w, s)
    conv = torch.nn.Conv2d(shape[-1], 5, kernel_size=k, stride=s, bias=False).double()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(np.transpose(w, (3, 2, 0, 1))))
        xt = torch.from_numpy(np.transpose(x, (0, 3, 1, 2)))
        ref = conv(mnv1.apply_tf_padding(xt, conv)).numpy()
    np.testing.assert_allclose(got, np.transpose(ref, (0, 2, 3, 1)), rtol=1e-12, atol=1e-12)
