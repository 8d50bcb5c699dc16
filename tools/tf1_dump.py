#!/usr/bin/env python3
"""Dump one batch of the REFERENCE's own TensorFlow graph for tools/compare_tf_dump.py -- the TF-side half of pinning parity
(SURVEY.md 8(c)).  Runs where the reference runs (TensorFlow 1.x: ``tf.Session``,
``tf.train.import_meta_graph``); this repository's environment has no TensorFlow and never imports it: the module is
imported lazily inside main(), and tests/test_compare_tf_dump.py drives dump() with a stub session.

    python tools/tf1_dump.py trained_model/FCN_sa dump_FCN_sa.npz [--shape 4,192,208] [--image real.npy] [--seed 0]

What it does is what common/deploy_network.py:44-49,110-111 does: import the meta graph, restore the checkpoint, feed
'image:0' (float32 [N,H,W,1]; UNet-LSTM: [N,T,H,W,1]) with 'training:0' False and fetch 'prob:0' and 'pred:0' -- plus the
pre-softmax tensor, found by walking back from the 'prob' op to the bias add of the last 1x1 conv (network.py:229).
The .npz holds exactly the keys compare_tf_dump.py reads: image float32, pred int32, prob float32, logits float32 (optional).
"""
import argparse
import sys

import numpy as np

DUMP_KEYS = {'image': np.float32, 'pred': np.int32, 'prob': np.float32, 'logits': np.float32}
LOGITS_OP_TYPES = ('BiasAdd', 'Add', 'AddV2')


def find_logits_tensor(graph, prob_op_name='prob', max_hops=8):
    """The tensor feeding the softmax: walk inputs[0] back from the 'prob' op (train_network.py:198) to the bias add of conv2d_20
    (network.py:229).  None if the graph does not look like that (the dump then simply has no 'logits')."""
    try:
        op = graph.get_operation_by_name(prob_op_name)
    except Exception:
        return None
    for _ in range(max_hops):
        if op.type in LOGITS_OP_TYPES:
            return op.outputs[0]
        if not len(op.inputs):
            return None
        op = op.inputs[0].op
    return None


def dump(sess, graph, image, want_logits=True):
    """sess.run on one batch exactly as deploy_network.py:110-111 calls it; returns the dict to np.savez."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    if image.ndim not in (4, 5) or image.shape[-1] != 1:
        raise ValueError("image must be float32 [N,H,W,1] (UNet-LSTM: [N,T,H,W,1]), got %r" % (image.shape,))
    feed = {'image:0': image, 'training:0': False}
    lt = find_logits_tensor(graph) if want_logits else None
    if lt is not None:
        logits, prob, pred = sess.run([lt, 'prob:0', 'pred:0'], feed_dict=feed)
    else:
        logits = None
        prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict=feed)
    out = {'image': image, 'prob': np.asarray(prob, np.float32), 'pred': np.asarray(pred, np.int32)}
    if logits is not None:
        out['logits'] = np.asarray(logits, np.float32)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('model_path', help="the reference's --model_path: checkpoint prefix with .meta / .index / .data files (demo_pipeline.py:50-54)")
    ap.add_argument('out_npz')
    ap.add_argument('--shape', default='4,192,208', help='N,H,W of a random batch in [0,1) (H, W multiples of 16, deploy_network.py:97); UNet-LSTM: N,T,H,W')
    ap.add_argument('--image', default=None, help='.npy with a real padded batch (the image_fr of deploy_network.py:105-107) instead of random data')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-logits', action='store_true')
    args = ap.parse_args(argv)
    if args.image:
        image = np.load(args.image).astype(np.float32)
        if image.shape[-1] != 1:
            image = image[..., None]
    else:
        dims = [int(v) for v in args.shape.split(',')]
        image = np.random.RandomState(args.seed).rand(*dims, 1).astype(np.float32)
    try:
        import tensorflow as tf                                        # TF 1.x (the reference's README.md:31); under TF 2 use compat.v1
        if not hasattr(tf, 'Session'):
            tf = tf.compat.v1
            tf.disable_eager_execution()
    except ImportError:
        sys.exit('tools/tf1_dump.py needs the TensorFlow environment the reference runs in (not available where this repository was built)')
    with tf.Session() as sess:
        saver = tf.train.import_meta_graph('{0}.meta'.format(args.model_path))    # deploy_network.py:48
        saver.restore(sess, '{0}'.format(args.model_path))                        # deploy_network.py:49
        d = dump(sess, tf.get_default_graph(), image, want_logits=not args.no_logits)
    np.savez(args.out_npz, **d)
    print('wrote %s: %s' % (args.out_npz, ', '.join('%s %s %s' % (k, v.dtype, v.shape) for k, v in d.items())))
    print('next, on a MI355X box with this repository:  python tools/compare_tf_dump.py %s %s --json parity_vs_tf.json' % (args.model_path, args.out_npz))


if __name__ == '__main__':
    main()
