"""Architecture descriptors for the networks on the deployment path.

The reference never calls ``build_FCN`` at deploy time -- it restores a
serialized graph (``common/deploy_network.py:48-49``) whose architecture is the
one fixed in ``common/train_network.py:156-195`` (FCN) and
``common/train_network_ao.py:268,275-284`` (aortic U-Net).  Those
hyper-parameters are hard-wired here.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

KIND_FCN = 0    # common/network.py:170-230  build_FCN
KIND_UNET = 1   # common/network_ao.py:18-64 UNet
KIND_UNET_LSTM = 2   # common/network_ao.py:322-399 UNet_LSTM_Model with BiConv_LSTM (:255-319)


@dataclass(frozen=True)
class LayerSpec:
    name: str
    kernel_shape: Tuple[int, int, int, int]   # HWIO, or [kh,kw,Cout,Cin] for transposed
    has_bn: bool                              # conv+BN+ReLU unit (network.py:19-34)
    has_bias: bool                            # only the logits layer (network.py:229)
    transposed: bool = False

    def n_floats(self) -> int:
        kh, kw, a, b = self.kernel_shape
        n = kh * kw * a * b
        cout = a if self.transposed else b
        if self.has_bn:
            n += 4 * cout
        if self.has_bias:
            n += cout
        return n


@dataclass(frozen=True)
class ModelArch:
    name: str
    kind: int
    n_class: int
    n_level: int = 5
    n_filter: Tuple[int, ...] = (16, 32, 64, 128, 256)
    n_block: Tuple[int, ...] = (2, 2, 3, 3, 3)
    same_dim: int = 32                    # KIND_UNET_LSTM: number of ConvLSTM hidden channels (n_hidden)
    fc: int = 64                          # KIND_UNET_LSTM: number of unrolled time steps (n_step)

    @property
    def n_hidden(self) -> int:
        return self.same_dim

    @property
    def n_step(self) -> int:
        return self.fc

    def layer_specs(self) -> List[LayerSpec]:
        """Canonical layer order == order of tensors in the flat weight array
        handed to ``ukbb_fcn_create`` (include/ukbb_fcn.h)."""
        L = []
        nf = self.n_filter
        cin = 1
        for l in range(self.n_level):
            for i in range(self.n_block[l]):
                L.append(LayerSpec('conv%d_%d' % (l, i), (3, 3, cin, nf[l]), True, False))
                cin = nf[l]
        if self.kind == KIND_FCN:
            for l in range(self.n_level):
                L.append(LayerSpec('same_dim%d' % l, (1, 1, nf[l], self.same_dim), True, False))
            L.append(LayerSpec('out0', (1, 1, self.same_dim * self.n_level, self.fc), True, False))
            L.append(LayerSpec('out1', (1, 1, self.fc, self.fc), True, False))
            L.append(LayerSpec('logits', (1, 1, self.fc, self.n_class), False, True))
        else:                                  # UNet and UNet-LSTM share encoder + decoder
            for l in range(self.n_level - 2, -1, -1):
                L.append(LayerSpec('up%d_t' % l, (3, 3, nf[l], nf[l + 1]), True, False, transposed=True))
                c = 2 * nf[l]
                for i in range(self.n_block[l]):
                    L.append(LayerSpec('up%d_%d' % (l, i), (3, 3, c, nf[l]), True, False))
                    c = nf[l]
            if self.kind == KIND_UNET:
                L.append(LayerSpec('logits', (1, 1, nf[0], self.n_class), False, True))
            else:
                # BiConv_LSTM (network_ao.py:255-319): one 3x3 conv over concat([x, h]) -> 4*n_hidden gate
                # channels (+bias) per direction, then 1x1 over concat([h_fw, h_bw]) -> n_class (+bias).
                # The UNet's own conv_out layer exists in the graph but its output is unused (:343-347).
                nh = self.n_hidden
                L.append(LayerSpec('lstm_fw', (3, 3, nf[0] + nh, 4 * nh), False, True))
                L.append(LayerSpec('lstm_bw', (3, 3, nf[0] + nh, 4 * nh), False, True))
                L.append(LayerSpec('lstm_out', (1, 1, 2 * nh, self.n_class), False, True))
        return L

    def n_weight_floats(self) -> int:
        return sum(s.n_floats() for s in self.layer_specs())

    def macs_per_pixel_table(self):
        """(name, kernel area, stride-accumulated downscale, Cin, Cout) rows used
        by bench.py to compute the algorithmic FLOPs."""
        rows = []
        for s in self.layer_specs():
            kh, kw, a, b = s.kernel_shape
            rows.append((s.name, kh * kw, a, b, s.transposed))
        return rows


# n_class table: common/train_network.py:156-168 (sa 4, la_2ch 2, la_4ch 3);
# seg4 has labels 1..5 in common/cardiac_utils.py:147 => 6 classes (inferred);
# aortic: common/train_network_ao.py:268.
MODELS = {
    'FCN_sa': ModelArch('FCN_sa', KIND_FCN, 4),
    'FCN_la_2ch': ModelArch('FCN_la_2ch', KIND_FCN, 2),
    'FCN_la_4ch': ModelArch('FCN_la_4ch', KIND_FCN, 3),
    'FCN_la_4ch_seg4': ModelArch('FCN_la_4ch_seg4', KIND_FCN, 6),
    'UNet_ao': ModelArch('UNet_ao', KIND_UNET, 3, n_block=(2, 2, 2, 2, 2)),
    # train_network_ao.py:292-298 with the demo model's name (...tw9_h16_bidir...): 9 steps, 16 hidden channels
    'UNet-LSTM_ao': ModelArch('UNet-LSTM_ao', KIND_UNET_LSTM, 3, n_block=(2, 2, 2, 2, 2), same_dim=16, fc=9),
}


def fcn_macs_per_slice(arch: ModelArch, H: int, W: int):
    """Algorithmic MACs per HxW slice, split (conv3x3, conv1x1); bilinear
    upsample counted as 0."""
    m3 = m1 = 0
    h, w = H, W
    nf = arch.n_filter
    cin = 1
    for l in range(arch.n_level):
        if l > 0:
            h, w = (h + 1) // 2, (w + 1) // 2
        for i in range(arch.n_block[l]):
            m3 += h * w * 9 * cin * nf[l]
            cin = nf[l]
        if arch.kind == KIND_FCN:
            m1 += h * w * nf[l] * arch.same_dim
    if arch.kind == KIND_FCN:
        m1 += H * W * (arch.same_dim * arch.n_level * arch.fc + arch.fc * arch.fc + arch.fc * arch.n_class)
    else:
        hs = [(H, W)]
        for l in range(1, arch.n_level):
            hs.append(((hs[-1][0] + 1) // 2, (hs[-1][1] + 1) // 2))
        for l in range(arch.n_level - 2, -1, -1):
            h, w = hs[l]
            hi, wi = hs[l + 1]
            m3 += hi * wi * 9 * nf[l + 1] * nf[l]        # transposed conv: 9 taps per INPUT pixel
            c = 2 * nf[l]
            for i in range(arch.n_block[l]):
                m3 += h * w * 9 * c * nf[l]
                c = nf[l]
        m1 += H * W * nf[0] * arch.n_class
    return m3, m1
