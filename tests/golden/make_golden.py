"""Generate the committed golden fixtures (run in the BUILD container only:
it reads /root/reference, which does not exist on the GPU box).

    python tests/golden/make_golden.py

1. ``ref_numpy_helpers.npz`` -- outputs of the reference's OWN pure-numpy
   functions.  The modules cannot be imported (``import tensorflow`` / ``cv2`` at
   their top fails here), so the individual function definitions are pulled out
   of the reference source with ``ast`` and executed unmodified with numpy only.
   No stand-in module is written and no reference source is stored: only
   inputs and outputs are.
2. ``fcn_*.npz`` / ``unet_*.npz`` -- outputs of THIS repo's numpy oracle
   (fp64 truth + fp32) for seeded synthetic weights.  They pin the HIP kernels to
   the oracle, not to TensorFlow (parity vs TF is unpinned, see oracle/__init__.py).
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'


def extract_functions(path, names):
    """Compile selected top-level function defs of a reference file in a
    namespace that only has numpy."""
    src = open(path).read()
    tree = ast.parse(src)
    ns = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, 'exec'), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing
    return ns


def make_reference_helper_vectors():
    net = extract_functions(os.path.join(REF, 'common/network.py'), ['linear_1d', 'linear_2d'])
    iu = extract_functions(os.path.join(REF, 'common/image_utils.py'),
                           ['rescale_intensity', 'normalise_intensity', 'np_categorical_dice'])
    out = {}
    for sz in (3, 7, 15, 31):
        out['linear_1d_%d' % sz] = net['linear_1d'](sz)
        out['linear_2d_%d' % sz] = net['linear_2d'](sz)
    rng = np.random.default_rng(0)
    vol = (1000.0 * rng.gamma(2.0, 1.0, size=(24, 20, 3, 4))).astype(np.float32)   # MR-magnitude-like
    out['rescale_in'] = vol.copy()
    v = vol.copy()
    out['rescale_out'] = iu['rescale_intensity'](v, (1, 99))
    out['rescale_in_after'] = v                     # in-place clip quirk
    v = vol.copy()
    out['rescale_out_2_98'] = iu['rescale_intensity'](v, (2.0, 98.0))
    vol2 = (1000.0 * rng.gamma(2.0, 1.0, size=(20, 18, 1, 5))).astype(np.float32)
    out['normalise_in'] = vol2.copy()
    out['normalise_out'] = iu['normalise_intensity'](vol2.copy(), 10.0)
    a = rng.integers(0, 4, size=(16, 16, 3))
    b = rng.integers(0, 4, size=(16, 16, 3))
    out['dice_a'], out['dice_b'] = a, b
    out['dice_k'] = np.array([iu['np_categorical_dice'](a, b, k) for k in range(4)])
    np.savez_compressed(os.path.join(HERE, 'ref_numpy_helpers.npz'), **out)
    print('ref_numpy_helpers.npz', {k: v.shape for k, v in out.items()})


def make_oracle_vectors():
    from oracle import fcn_oracle as O
    from ukbb_cardiac_amd.arch import MODELS
    from ukbb_cardiac_amd.weights import synthetic_params
    from ukbb_cardiac_amd.phantom import cine_phantom, uniform_slices

    def fcn_case(tag, model, img):
        arch = MODELS[model]
        params = synthetic_params(arch, 1234)
        l64 = O.build_FCN(img, params, arch.n_class, dtype=np.float64)
        l32 = O.build_FCN(img, params, arch.n_class, dtype=np.float32)
        np.savez_compressed(os.path.join(HERE, tag + '.npz'), model=model, seed=1234,
                            image=img.astype(np.float32),
                            logits64=l64.astype(np.float32),          # fp64 result, stored rounded
                            margin64=O.top2_margin(l64).astype(np.float32),
                            pred64=O.argmax_pred(l64), pred32=O.argmax_pred(l32))
        print(tag, l64.shape, 'classes', np.bincount(O.argmax_pred(l64).ravel(), minlength=arch.n_class),
              'fp32-vs-fp64 pred flips', int((O.argmax_pred(l64) != O.argmax_pred(l32)).sum()))

    fcn_case('fcn_sa_2x32x48', 'FCN_sa', cine_phantom(2, 32, 48, seed=11))
    fcn_case('fcn_sa_1x192x208', 'FCN_sa', cine_phantom(1, 192, 208, seed=12))
    fcn_case('fcn_sa_1x192x208_uniform', 'FCN_sa', uniform_slices(1, 192, 208, seed=1))
    fcn_case('fcn_la2ch_1x176x208', 'FCN_la_2ch', cine_phantom(1, 176, 208, seed=13))
    fcn_case('fcn_seg4_1x80x112', 'FCN_la_4ch_seg4', cine_phantom(1, 80, 112, seed=14))
    fcn_case('fcn_la4ch_2x48x16', 'FCN_la_4ch', cine_phantom(2, 48, 16, seed=15))

    arch = MODELS['UNet_ao']
    params = synthetic_params(arch, 1234)
    img = ((cine_phantom(2, 64, 96, seed=16) - 0.3) / 0.25).astype(np.float32)
    l64 = O.UNet(img, params, arch.n_class, n_block=arch.n_block, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'unet_ao_2x64x96.npz'), model='UNet_ao', seed=1234, image=img,
                        logits64=l64.astype(np.float32), margin64=O.top2_margin(l64).astype(np.float32),
                        pred64=O.argmax_pred(l64),
                        prob64=O.softmax(l64).astype(np.float32))
    print('unet_ao_2x64x96', l64.shape, np.bincount(O.argmax_pred(l64).ravel(), minlength=3))


if __name__ == '__main__':
    if os.path.isdir(REF):
        make_reference_helper_vectors()
    else:
        print('no /root/reference here: keeping the committed ref_numpy_helpers.npz')
    make_oracle_vectors()
