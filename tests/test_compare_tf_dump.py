"""tools/compare_tf_dump.py: the grader that will pin parity against real TensorFlow outputs once someone with a
TF-1.x environment runs the dumper of INTEGRATION.md section 5.  CPU: the grading rules.  GPU: the whole tool on a
checkpoint written by tests/tf_bundle_writer.py and a dump computed by the CPU oracle standing in for TensorFlow."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import compare_tf_dump as ctd   # noqa: E402


def _fake(n=2, h=8, w=8, c=4, seed=0):
    rng = np.random.default_rng(seed)
    lg = (3 * rng.standard_normal((n, h, w, c))).astype(np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True))
    return lg, (e / e.sum(-1, keepdims=True)).astype(np.float32), np.argmax(lg, -1).astype(np.int32)


def test_grade_passes_on_identical_and_on_fp32_noise():
    lg, pr, pd = _fake()
    rep = ctd.grade({'logits': lg, 'prob': pr, 'pred': pd}, {'logits': lg + 1e-6, 'prob': pr, 'pred': pd})
    assert rep['pass'] and rep['checks']['labels']['mismatches'] == 0 and rep['checks']['logits']['rel_err'] < 1e-5


def test_grade_fails_on_a_semantic_error_and_lists_it():
    lg, pr, pd = _fake()
    ours = {'logits': np.roll(lg, 1, axis=2), 'prob': np.roll(pr, 1, axis=2), 'pred': np.roll(pd, 1, axis=2)}   # one-pixel shift
    rep = ctd.grade({'logits': lg, 'prob': pr, 'pred': pd}, ours)
    assert not rep['pass'] and not rep['checks']['logits']['pass']
    lab = rep['checks']['labels']
    assert lab['mismatches_away_from_ties'] > 0 and lab['listed'][0]['tf_top2_margin'] > 0


def test_grade_tolerates_a_flip_only_at_a_numerical_tie():
    lg, pr, pd = _fake()
    lg[0, 0, 0] = [1.0, 1.0 + 2e-6, -3, -3]                       # TF says class 1 by 2e-6
    e = np.exp(lg - lg.max(-1, keepdims=True)); pr = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    pd = np.argmax(lg, -1).astype(np.int32)
    mine = pd.copy(); mine[0, 0, 0] = 0
    rep = ctd.grade({'logits': lg, 'prob': pr, 'pred': pd}, {'logits': lg, 'prob': pr, 'pred': mine})
    assert rep['pass'] and rep['checks']['labels']['mismatches_at_numerical_ties'] == 1
    mine[1, 3, 3] = (pd[1, 3, 3] + 1) % 4                         # ... but not elsewhere
    assert not ctd.grade({'logits': lg, 'prob': pr, 'pred': pd}, {'logits': lg, 'prob': pr, 'pred': mine})['pass']


def test_grade_without_logits_uses_log_probabilities():
    lg, pr, pd = _fake(seed=3)
    rep = ctd.grade({'prob': pr, 'pred': pd}, {'logits': lg + 0.7, 'prob': pr, 'pred': pd})   # logits are defined up to a constant
    assert rep['pass'] and 'log_prob' in rep['checks']
    rep = ctd.grade({'prob': pr, 'pred': pd}, {'logits': 1.05 * lg, 'prob': pr, 'pred': pd})
    assert not rep['checks']['log_prob']['pass']
    assert not ctd.grade({'pred': pd}, {'logits': lg, 'prob': pr, 'pred': (pd + 1) % 4})['pass']


# ---- tools/tf1_dump.py: the TF-side dumper, driven by a stub of the two TensorFlow objects it touches ----------------------------
class _StubTensor:
    def __init__(self, op, name):
        self.op, self.name = op, name


class _StubOp:
    def __init__(self, typ, name, inputs=()):
        self.type, self.name = typ, name
        self.inputs = [o.outputs[0] for o in inputs]
        self.outputs = [_StubTensor(self, name + ':0')]


class _StubGraph:
    """conv2d_20/BiasAdd -> Softmax 'prob' -> ArgMax -> Cast 'pred' (network.py:229, train_network.py:198-199)."""
    def __init__(self, with_bias_add=True):
        conv = _StubOp('Conv2D', 'conv2d_20/Conv2D')
        self.logits_op = _StubOp('BiasAdd', 'conv2d_20/BiasAdd', [conv]) if with_bias_add else conv
        self.ops = {'prob': _StubOp('Softmax', 'prob', [self.logits_op])}

    def get_operation_by_name(self, name):
        return self.ops[name]


class _StubSession:
    """sess.run(fetches, feed_dict) of deploy_network.py:110-111 over a fixed (logits, prob, pred) triple."""
    def __init__(self, graph, lg, pr, pd):
        self.graph, self.vals, self.calls = graph, {'prob:0': pr, 'pred:0': pd}, []
        self.lg = lg

    def run(self, fetches, feed_dict):
        assert set(feed_dict) == {'image:0', 'training:0'} and feed_dict['training:0'] is False
        assert feed_dict['image:0'].dtype == np.float32 and feed_dict['image:0'].flags['C_CONTIGUOUS']
        self.calls.append(fetches)
        out = []
        for f in fetches:
            if isinstance(f, _StubTensor):
                assert f is self.graph.logits_op.outputs[0]
                out.append(self.lg.astype(np.float64))           # whatever dtype the session hands back, the dump stores float32
            else:
                out.append(self.vals[f].astype(np.int64) if f == 'pred:0' else self.vals[f])
        return out


def test_tf1_dump_writes_exactly_what_the_grader_reads(tmp_path):
    """tools/tf1_dump.py (TensorFlow imported only inside main()) produces the keys / dtypes / shapes
    compare_tf_dump.py consumes; a dump of 'TensorFlow' outputs graded against the same outputs passes end to end."""
    import tf1_dump
    assert 'tensorflow' not in sys.modules
    lg, pr, pd = _fake(n=3, h=16, w=16, seed=5)
    image = np.random.default_rng(1).random((3, 16, 16, 1))          # float64 on purpose: the dumper feeds float32
    g = _StubGraph()
    sess = _StubSession(g, lg, pr, pd)
    d = tf1_dump.dump(sess, g, image)
    assert set(d) == set(tf1_dump.DUMP_KEYS) == {'image', 'pred', 'prob', 'logits'}
    for k, v in d.items():
        assert v.dtype == tf1_dump.DUMP_KEYS[k], k
    assert d['image'].shape == (3, 16, 16, 1) and d['pred'].shape == (3, 16, 16) and d['prob'].shape == d['logits'].shape == (3, 16, 16, 4)
    assert len(sess.calls) == 1 and sess.calls[0][1:] == ['prob:0', 'pred:0']      # ONE sess.run, the reference's two fetches + the logits
    # round trip through the file into the grader
    path = str(tmp_path / 'dump.npz')
    np.savez(path, **d)
    back = dict(np.load(path))
    rep = ctd.grade(back, {'logits': lg, 'prob': pr, 'pred': pd})
    assert rep['pass'] and set(rep['checks']) >= {'logits', 'prob', 'labels'}
    # a graph whose softmax input is not a bias add: no logits in the dump, the grader falls back to log-probabilities
    g2 = _StubGraph(with_bias_add=False)
    d2 = tf1_dump.dump(_StubSession(g2, lg, pr, pd), g2, image)
    assert set(d2) == {'image', 'pred', 'prob'} and ctd.grade(d2, {'logits': lg, 'prob': pr, 'pred': pd})['pass']
    with pytest.raises(ValueError):
        tf1_dump.dump(sess, g, image[..., 0])                       # not [N,H,W,1]
    # the command-line entry refuses politely where TensorFlow is absent (this environment), before touching any file
    with pytest.raises(SystemExit) as e:
        tf1_dump.main(['no_such_model', str(tmp_path / 'x.npz'), '--shape', '1,16,16'])
    assert 'TensorFlow' in str(e.value) and not (tmp_path / 'x.npz').exists()


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['FCN_sa', 'UNet_ao'])
def test_tool_end_to_end_on_a_checkpoint_prefix(tmp_path, model):
    from oracle import c_oracle
    from tests.test_tf_checkpoint import _tf_tensors
    from tests.tf_bundle_writer import write_checkpoint
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.phantom import cine_phantom
    from ukbb_cardiac_amd.weights import pack_flat, synthetic_params
    arch = MODELS[model]
    params = synthetic_params(arch, 77)
    prefix = str(tmp_path / model)
    write_checkpoint(prefix, _tf_tensors(arch, params), tensor_crc=False)          # .index + .data-00000-of-00001
    img = cine_phantom(2, 64, 80, seed=9)
    lg, pr, pd = c_oracle.forward(arch, pack_flat(arch, params), img, want_prob=True)   # stand-in for the TF session
    np.savez(str(tmp_path / 'dump.npz'), image=img, logits=lg, prob=pr, pred=pd)
    out = str(tmp_path / 'report.json')
    assert ctd.main([prefix, str(tmp_path / 'dump.npz'), '--json', out]) == 0
    rep = json.load(open(out))
    assert rep['pass'] and rep['model'] == arch.name and rep['checks']['logits']['rel_err'] < 1e-4
    np.savez(str(tmp_path / 'bad.npz'), image=img, logits=lg[:, ::-1], prob=pr[:, ::-1], pred=pd[:, ::-1])
    assert ctd.main([prefix, str(tmp_path / 'bad.npz')]) == 1
