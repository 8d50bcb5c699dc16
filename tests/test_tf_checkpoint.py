"""TF checkpoint-V2 reader: round trips through an independently written
bundle writer (tests/tf_bundle_writer.py), variable naming of build_FCN / UNet, arch inference,
corruption detection.  No TensorFlow here: the reader is not pinned against a real checkpoint."""
import os

import numpy as np
import pytest

from tests.tf_bundle_writer import write_checkpoint
from ukbb_cardiac_amd import tf_checkpoint as tfc
from ukbb_cardiac_amd.arch import MODELS
from ukbb_cardiac_amd.weights import pack_flat, synthetic_params


def _tf_tensors(arch, params, with_slots=True):
    t = {}
    for layer, names in tfc.variable_names(arch).items():
        for key, tfname in names.items():
            t[tfname] = params[layer][key]
            if with_slots and key in ('gamma', 'beta', 'bias'):                # Adam slots the reader must ignore
                t[tfname + '/Adam'] = np.zeros_like(params[layer][key])
                t[tfname + '/Adam_1'] = np.ones_like(params[layer][key])
    if with_slots:
        t['beta1_power'] = np.float32(0.9)
        t['beta2_power'] = np.float32(0.999)
        t['global_step'] = np.int64(50000)
    return t


def test_crc32c_known_answers():
    assert tfc.crc32c(b'123456789') == 0xE3069283                      # standard CRC-32C check value
    assert tfc.crc32c(b'') == 0
    assert tfc.crc32c(b'6789', tfc.crc32c(b'12345')) == 0xE3069283     # incremental form
    assert tfc.mask_crc(0) == 0xa282ead8


def test_snappy_literal_and_copies():
    # "abcdabcdabcdX": literal "abcd", copy(len 8, offset 4) overlapping its own output, literal "X"
    comp = bytes([13]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((8 - 4) << 2) | 1, 4]) + bytes([0]) + b'X'
    assert tfc.snappy_uncompress(comp) == b'abcdabcdabcdX'
    comp2 = bytes([6]) + bytes([(2 - 1) << 2]) + b'ab' + bytes([((4 - 1) << 2) | 2, 2, 0])   # 2-byte-offset copy
    assert tfc.snappy_uncompress(comp2) == b'ababab'
    with pytest.raises(tfc.CheckpointError):
        tfc.snappy_uncompress(bytes([4]) + bytes([((4 - 4) << 2) | 1, 9]))                    # offset beyond output


@pytest.mark.parametrize('model', ['FCN_sa', 'FCN_la_2ch', 'FCN_la_4ch', 'FCN_la_4ch_seg4', 'UNet_ao', 'UNet-LSTM_ao'])
def test_round_trip_all_models(tmp_path, model):
    arch = MODELS[model]
    params = synthetic_params(arch, 7)
    prefix = str(tmp_path / model)
    write_checkpoint(prefix, _tf_tensors(arch, params), block_size=300, restart_interval=4, tensor_crc=(model == 'FCN_la_2ch'))
    reader = tfc.CheckpointReader(prefix)
    assert 'global_step' in reader.names() and reader.get_tensor('global_step') == 50000
    arch2, params2 = tfc.checkpoint_to_params(prefix, verify_crc=(model == 'FCN_la_2ch'))
    assert arch2 == arch                                               # inferred from the kernel shapes alone
    np.testing.assert_array_equal(pack_flat(arch2, params2), pack_flat(arch, params))


def test_variable_names_follow_the_reference_graphs():
    n = tfc.variable_names(MODELS['FCN_sa'])
    assert n['conv0_0']['kernel'] == 'conv2d/kernel' and n['conv0_1']['kernel'] == 'conv2d_1/kernel'
    assert n['conv4_2']['kernel'] == 'conv2d_12/kernel'               # 13 encoder convs (network.py:179-190)
    assert n['same_dim0']['kernel'] == 'conv2d_13/kernel'             # then the five squeeze convs (:203-206)
    assert n['out0']['kernel'] == 'conv2d_18/kernel' and n['out1']['gamma'] == 'batch_normalization_19/gamma'
    assert n['logits'] == {'kernel': 'conv2d_20/kernel', 'bias': 'conv2d_20/bias'}     # network.py:229
    u = tfc.variable_names(MODELS['UNet_ao'])
    assert u['conv0_0']['kernel'] == 'UNet/conv0/conv2d/kernel' and u['conv0_1']['kernel'] == 'UNet/conv0/conv2d_1/kernel'
    assert u['conv3_0']['mean'] == 'UNet/conv3/batch_normalization/moving_mean'
    assert u['up3_t']['kernel'] == 'UNet/conv3_up/conv2d_transpose/kernel'              # network_ao.py:44-47
    assert u['up3_t']['var'] == 'UNet/conv3_up/batch_normalization/moving_variance'
    assert u['up3_0']['kernel'] == 'UNet/conv3_up/conv2d/kernel' and u['up3_0']['gamma'] == 'UNet/conv3_up/batch_normalization_1/gamma'
    assert u['up3_1']['kernel'] == 'UNet/conv3_up/conv2d_1/kernel'
    assert u['logits']['bias'] == 'UNet/conv_out/conv2d/bias'                            # network_ao.py:63


def test_load_model_reads_the_checkpoint_prefix(tmp_path):
    from ukbb_cardiac_amd.engine import load_model
    arch = MODELS['FCN_la_4ch']
    params = synthetic_params(arch, 3)
    prefix = str(tmp_path / 'FCN_la_4ch')
    write_checkpoint(prefix, _tf_tensors(arch, params, with_slots=False), tensor_crc=False)
    arch2, params2 = load_model(prefix)                                # --model_path as demo_pipeline.py:63 passes it
    assert arch2.name == 'FCN_la_4ch'
    np.testing.assert_array_equal(pack_flat(arch2, params2), pack_flat(arch, params))
    assert tfc.main([prefix, '-o', str(tmp_path / 'm.ukbbw')]) == 0    # converter CLI
    arch3, params3 = load_model(str(tmp_path / 'm.ukbbw'))
    np.testing.assert_array_equal(pack_flat(arch3, params3), pack_flat(arch, params))


def test_corruption_is_detected(tmp_path):
    arch = MODELS['FCN_la_2ch']
    params = synthetic_params(arch, 5)
    prefix = str(tmp_path / 'm')
    write_checkpoint(prefix, _tf_tensors(arch, params, with_slots=False))
    raw = bytearray(open(prefix + '.index', 'rb').read())
    bad = bytearray(raw); bad[10] ^= 0x40
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(tfc.CheckpointError):
        tfc.CheckpointReader(prefix)
    bad = bytearray(raw); bad[-1] ^= 0xFF                                # footer magic
    open(prefix + '.index', 'wb').write(bytes(bad))
    with pytest.raises(tfc.CheckpointError):
        tfc.CheckpointReader(prefix)
    open(prefix + '.index', 'wb').write(bytes(raw))
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    r = tfc.CheckpointReader(prefix)
    with pytest.raises(tfc.CheckpointError):
        for n in r.names():
            r.get_tensor(n, verify_crc=True)
    os.remove(prefix + '.data-00000-of-00001')
    with pytest.raises(FileNotFoundError):
        tfc.CheckpointReader(prefix).get_tensor('conv2d/kernel')


def test_missing_variable_and_wrong_shape(tmp_path):
    arch = MODELS['FCN_sa']
    params = synthetic_params(arch, 9)
    t = _tf_tensors(arch, params, with_slots=False)
    del t['batch_normalization_7/moving_variance']
    prefix = str(tmp_path / 'm')
    write_checkpoint(prefix, t, tensor_crc=False)
    with pytest.raises(KeyError):
        tfc.checkpoint_to_params(prefix, arch)
    with pytest.raises(tfc.CheckpointError):                              # a 4-class checkpoint asked to be a 2-class model
        write_checkpoint(prefix, _tf_tensors(arch, params, with_slots=False), tensor_crc=False)
        tfc.checkpoint_to_params(prefix, MODELS['FCN_la_2ch'])


@pytest.mark.parametrize('cell', ['conv_lstm_cell', 'conv_2d_lstm_cell', 'Conv2DLSTMCell'])
def test_lstm_cell_scope_is_matched_not_assumed(tmp_path, cell):
    """network_ao.py:277,290 instantiate tf.contrib.rnn.Conv2DLSTMCell; which scope name its variables get depends on
    the TF 1.x release (`conv_lstm_cell` vs `conv_2d_lstm_cell`).  The importer takes whatever single scope sits below
    LSTM/<direction>/ with a kernel + biases, and is not confused by optimizer slots of those variables."""
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 11)
    t = _tf_tensors(arch, params)
    for old in [n for n in t if '/conv_lstm_cell/' in n]:
        t[old.replace('/conv_lstm_cell/', '/%s/' % cell)] = t.pop(old)
    for d in ('forward', 'backward'):
        for v in ('kernel', 'biases'):
            t['LSTM/%s/%s/%s/Adam' % (d, cell, v)] = np.zeros_like(t['LSTM/%s/%s/%s' % (d, cell, v)])
    prefix = str(tmp_path / 'UNet-LSTM_ao')
    write_checkpoint(prefix, t, tensor_crc=False)
    arch2, params2 = tfc.checkpoint_to_params(prefix)
    assert arch2 == arch
    np.testing.assert_array_equal(pack_flat(arch2, params2), pack_flat(arch, params))
    # two candidate cells in one direction: refuse rather than guess
    t['LSTM/forward/other_cell/kernel'] = t['LSTM/forward/%s/kernel' % cell]
    t['LSTM/forward/other_cell/biases'] = t['LSTM/forward/%s/biases' % cell]
    write_checkpoint(prefix, t, tensor_crc=False)
    with pytest.raises(tfc.CheckpointError, match='one cell scope'):
        tfc.checkpoint_to_params(prefix)


def test_unidirectional_conv_lstm_checkpoint_loads_as_a_zero_backward_cell(tmp_path):
    """common/network_ao.py:214-252 Conv_LSTM (train_network_ao.py --bidirectional=False) keeps ONE cell directly under LSTM/
    and its output conv as LSTM/conv2d.  (r05 refused such a checkpoint by name; r06 serves it: the importer embeds it in the
    bidirectional layer set with an all-zero backward cell, weights.embed_unidirectional_lstm -- exact, tests/test_unidirectional_lstm.py.)"""
    arch = MODELS['UNet-LSTM_ao']
    params = synthetic_params(arch, 12)
    t = _tf_tensors(arch, params, with_slots=False)
    uni = {n: v for n, v in t.items() if not n.startswith('LSTM/')}
    uni['LSTM/conv_lstm_cell/kernel'] = t['LSTM/forward/conv_lstm_cell/kernel']
    uni['LSTM/conv_lstm_cell/biases'] = t['LSTM/forward/conv_lstm_cell/biases']
    uni['LSTM/conv2d/kernel'] = t['LSTM/output/conv2d/kernel'][:, :, :16]                # [1,1,n_hidden,n_class]
    uni['LSTM/conv2d/bias'] = t['LSTM/output/conv2d/bias']
    uni['LSTM/conv_lstm_cell/kernel/Adam'] = np.zeros_like(uni['LSTM/conv_lstm_cell/kernel'])   # optimizer slots do not confuse it
    prefix = str(tmp_path / 'UNet-LSTM_uni')
    write_checkpoint(prefix, uni, tensor_crc=False)
    arch2, params2 = tfc.checkpoint_to_params(prefix)
    assert arch2 == arch
    np.testing.assert_array_equal(params2['lstm_fw']['kernel'], params['lstm_fw']['kernel'])
    np.testing.assert_array_equal(params2['lstm_fw']['bias'], params['lstm_fw']['bias'])
    assert not params2['lstm_bw']['kernel'].any() and not params2['lstm_bw']['bias'].any()
    np.testing.assert_array_equal(params2['lstm_out']['kernel'][:, :, :16], params['lstm_out']['kernel'][:, :, :16])
    assert not params2['lstm_out']['kernel'][:, :, 16:].any()
    np.testing.assert_array_equal(params2['lstm_out']['bias'], params['lstm_out']['bias'])
    for name in params:
        if not name.startswith('lstm'):
            for k in params[name]:
                np.testing.assert_array_equal(params2[name][k], params[name][k])
    assert pack_flat(arch2, params2).size == arch.n_weight_floats()
    # a checkpoint with BOTH the single cell and the directional scopes is the bidirectional model (the extra variables are ignored)
    both = dict(t); both.update({k: v for k, v in uni.items() if k.startswith('LSTM/')})
    write_checkpoint(prefix, both, tensor_crc=False)
    _, params3 = tfc.checkpoint_to_params(prefix)
    np.testing.assert_array_equal(pack_flat(arch, params3), pack_flat(arch, params))
