#!/usr/bin/env python3
"""Grade the HIP engine against outputs of the REAL TensorFlow graph -- the one comparison that can turn
"parity vs this repo's restatement" into "parity vs the reference" (last bullet).

    python tools/compare_tf_dump.py MODEL_PATH DUMP.npz [--device 0] [--json report.json]

MODEL_PATH  the reference's --model_path: a TF checkpoint-V2 prefix (``trained_model/FCN_sa`` with its
            .index / .data-00000-of-00001 files, read by ukbb_cardiac_amd/tf_checkpoint.py without TensorFlow) or a
            ``.ukbbw`` blob.
DUMP.npz    arrays fetched from the reference's own session on one input batch (INTEGRATION.md section 5 has the
            dumper to run next to common/deploy_network.py:110-111 in a TF-1.x environment):
              image   float32 [N,H,W,1]     (UNet-LSTM: [N,T,H,W,1])   what was fed as 'image:0'
              pred    int32   [N,H,W]       'pred:0'
              prob    float32 [N,H,W,C]     'prob:0'                    (optional but recommended)
              logits  float32 [N,H,W,C]     the pre-softmax tensor       (optional)

Bar (BASELINE.json north_star): logits within 1e-3 relative (of max|logits| over the batch), label maps identical.
Where labels differ the pixel is listed with TF's own top-2 margin, so a reader can tell a numerical tie (margin at
the fp32 noise level, ~1e-5 of the logit scale) from a semantic error (SAME padding, transposed-conv crop, BN epsilon,
gate order ... -- the [TF-recall] items of SURVEY.md App. B).  Without ``logits`` in the dump the same test is made on
log-probabilities, which equal the logits up to a per-pixel constant.  Exit status 0 = pass.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LOGIT_RTOL = 1e-3
PROB_ATOL = 1e-4
TIE_RTOL = 1e-4           # a label disagreement counts as a tie if TF's top-2 margin < TIE_RTOL * max|logits|


def _log_softmax(x):
    x = x.astype(np.float64)
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def grade(dump, ours, max_list=20):
    """dump: {'pred', optional 'prob', 'logits'} from TensorFlow; ours: {'logits', 'prob', 'pred'} from the engine.
    Returns a JSON-able report with 'pass'."""
    rep = {'shape': list(np.asarray(dump['pred']).shape), 'checks': {}}
    ok = True
    tf_scores = None
    if 'logits' in dump:
        ref = np.asarray(dump['logits'], np.float64)
        scale = float(np.abs(ref).max())
        err = float(np.abs(ours['logits'].astype(np.float64) - ref).max())
        rep['checks']['logits'] = {'max_abs_err': err, 'scale_max_abs_logits': scale, 'rel_err': err / scale if scale else None,
                                   'tolerance_rel': LOGIT_RTOL, 'pass': bool(err <= LOGIT_RTOL * scale)}
        ok &= rep['checks']['logits']['pass']
        tf_scores = ref
    if 'prob' in dump:
        refp = np.asarray(dump['prob'], np.float64)
        perr = float(np.abs(ours['prob'].astype(np.float64) - refp).max())
        rep['checks']['prob'] = {'max_abs_err': perr, 'tolerance_abs': PROB_ATOL, 'pass': bool(perr <= PROB_ATOL)}
        ok &= rep['checks']['prob']['pass']
        if tf_scores is None:
            # no logits dumped: compare log-probabilities where TF's prob is not denormal-small
            with np.errstate(divide='ignore'):
                lref = np.log(refp)
            lo = _log_softmax(ours['logits'])
            mask = refp > 1e-30
            scale = float(np.abs(ours['logits']).max())
            lerr = float(np.abs(lo - lref)[mask].max()) if mask.any() else 0.0
            # log(prob) of a float32 prob carries a relative rounding of 6e-8 in prob = 6e-8 absolute in log; fine
            rep['checks']['log_prob'] = {'max_abs_err': lerr, 'scale_max_abs_logits': scale, 'rel_err': lerr / scale if scale else None,
                                         'tolerance_rel': LOGIT_RTOL, 'pass': bool(lerr <= LOGIT_RTOL * scale)}
            ok &= rep['checks']['log_prob']['pass']
            tf_scores = lref
    refl = np.asarray(dump['pred'])
    bad = np.asarray(ours['pred']) != refl
    lab = {'pixels': int(bad.size), 'mismatches': int(bad.sum()), 'pass': True}
    if bad.any():
        if tf_scores is not None:
            srt = np.sort(np.where(np.isfinite(tf_scores), tf_scores, -1e30), axis=-1)
            margin = srt[..., -1] - srt[..., -2]
            scale = float(np.abs(ours['logits']).max())
            tie = margin < TIE_RTOL * scale
            lab['mismatches_at_numerical_ties'] = int((bad & tie).sum())
            lab['mismatches_away_from_ties'] = int((bad & ~tie).sum())
            lab['pass'] = lab['mismatches_away_from_ties'] == 0
            idx = np.argwhere(bad)
            lab['listed'] = [{'index': [int(v) for v in i], 'tf_label': int(refl[tuple(i)]), 'engine_label': int(np.asarray(ours['pred'])[tuple(i)]),
                              'tf_top2_margin': float(margin[tuple(i)])} for i in idx[:max_list]]
        else:
            lab['pass'] = False
            lab['note'] = 'the dump holds neither prob nor logits, so mismatches cannot be classified as ties'
    rep['checks']['labels'] = lab
    ok &= lab['pass']
    rep['pass'] = bool(ok)
    return rep


def run_engine(model_path, image, device=0):
    from ukbb_cardiac_amd.arch import KIND_UNET_LSTM
    from ukbb_cardiac_amd.engine import Engine, load_model
    arch, params = load_model(model_path)
    x = np.ascontiguousarray(image, np.float32)
    with Engine(arch, params, device) as eng:
        if arch.kind == KIND_UNET_LSTM:
            out = eng.run_seq(x, want_logits=True)
        else:
            out = eng.run(x, want_logits=True)
    return arch, out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('model_path')
    ap.add_argument('dump')
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--json', default=None, help='also write the report here')
    args = ap.parse_args(argv)
    d = np.load(args.dump)
    dump = {k: d[k] for k in d.files}
    for need in ('image', 'pred'):
        if need not in dump:
            sys.exit('%s lacks the array %r' % (args.dump, need))
    arch, ours = run_engine(args.model_path, dump['image'], args.device)
    rep = grade(dump, ours)
    rep['model'] = arch.name
    rep['model_path'] = args.model_path
    rep['dump'] = args.dump
    text = json.dumps(rep, indent=1)
    print(text)
    if args.json:
        with open(args.json, 'w') as f:
            f.write(text + '\n')
    return 0 if rep['pass'] else 1


if __name__ == '__main__':
    sys.exit(main())
