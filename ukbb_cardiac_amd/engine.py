"""Python face of the HIP engine: ``Engine`` (one model bound to one GPU) and
``Session``, which mirrors the slice of ``tf.Session`` the reference deploy
scripts use (``common/deploy_network.py:44-49,110-111``)."""
import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from .arch import ModelArch, MODELS
from .weights import Params, load_blob, pack_flat, synthetic_params

MODEL_EXT = '.ukbbw'


class Engine:
    """One network on one device.  Host-array entry point ``run`` is the
    ``sess.run`` equivalent; ``run_device`` takes raw device pointers
    (e.g. ``torch.Tensor.data_ptr()``) and a HIP stream handle."""

    def __init__(self, arch: ModelArch, params: Params, device: int = 0):
        self.arch = arch
        flat = np.ascontiguousarray(pack_flat(arch, params), dtype=np.float32)
        a = _lib.arch_struct(arch)
        want = _lib.lib.ukbb_fcn_weight_count(C.byref(a))
        if want != flat.size:
            raise _lib.UkbbFcnError('weight count mismatch: library expects %d floats, got %d' % (want, flat.size))
        self._h = _lib.lib.ukbb_fcn_create(C.byref(a), _lib.f32ptr(flat), flat.size, int(device))
        if not self._h:
            raise _lib.UkbbFcnError('ukbb_fcn_create failed: ' + _lib.last_error())
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            _lib.lib.ukbb_fcn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host arrays -----------------------------------------------------------
    def run(self, image: np.ndarray, want_logits=False, want_prob=True, want_pred=True):
        """image: float32 [N,H,W,1] (or [N,H,W]).  Returns dict with the
        requested 'logits' [N,H,W,C] f32, 'prob' [N,H,W,C] f32, 'pred' [N,H,W] int32."""
        x = np.ascontiguousarray(image, dtype=np.float32)
        if x.ndim == 4:
            if x.shape[3] != 1:
                raise ValueError('image must have a single channel, got shape %s' % (x.shape,))
            x = x[..., 0]
        if x.ndim != 3:
            raise ValueError('image must be [N,H,W,1], got shape %s' % (image.shape,))
        n, h, w = x.shape
        c = self.arch.n_class
        out: Dict[str, np.ndarray] = {}
        lg = np.empty((n, h, w, c), np.float32) if want_logits else None
        pr = np.empty((n, h, w, c), np.float32) if want_prob else None
        pd = np.empty((n, h, w), np.int32) if want_pred else None
        rc = _lib.lib.ukbb_fcn_forward_host(
            self._h, _lib.f32ptr(x), n, h, w,
            _lib.f32ptr(lg) if lg is not None else None,
            _lib.f32ptr(pr) if pr is not None else None,
            _lib.i32ptr(pd) if pd is not None else None)
        _lib.check(rc, 'ukbb_fcn_forward_host')
        if lg is not None:
            out['logits'] = lg
        if pr is not None:
            out['prob'] = pr
        if pd is not None:
            out['pred'] = pd
        return out

    # -- device pointers -------------------------------------------------------
    def reserve(self, n, h, w):
        _lib.check(_lib.lib.ukbb_fcn_reserve(self._h, n, h, w), 'ukbb_fcn_reserve')

    def run_device(self, image_ptr: int, n: int, h: int, w: int, logits_ptr: int = 0, prob_ptr: int = 0,
                   pred_ptr: int = 0, stream: int = 0):
        rc = _lib.lib.ukbb_fcn_forward(self._h, C.c_void_p(image_ptr), n, h, w,
                                       C.c_void_p(logits_ptr or None), C.c_void_p(prob_ptr or None),
                                       C.c_void_p(pred_ptr or None), C.c_void_p(stream or None))
        _lib.check(rc, 'ukbb_fcn_forward')

    # -- UNet-LSTM (aortic default model) ---------------------------------------------
    def run_seq(self, image: np.ndarray, want_logits=False, want_prob=True, want_pred=True):
        """The reference's ``sess.run('prob:0', {'image:0': image_idx})`` for the UNet-LSTM graph
        (common/deploy_network_ao.py:171-172): image float32 [N,T,H,W,1] (or [N,T,H,W]) with T = arch.n_step
        -> 'prob' [N,T,H,W,C], 'pred' [N,T,H,W], optionally 'logits'.  Host arrays; staged through torch tensors."""
        import torch
        x = np.ascontiguousarray(image, dtype=np.float32)
        if x.ndim == 5:
            x = x[..., 0]
        if x.ndim != 4 or x.shape[1] != self.arch.n_step:
            raise ValueError('image must be [N,%d,H,W,1], got shape %s' % (self.arch.n_step, image.shape))
        n, t, h, w = x.shape
        c = self.arch.n_class
        dev = torch.device('cuda', self.device)
        stream = torch.cuda.current_stream(dev).cuda_stream
        xd = torch.from_numpy(x).to(dev)
        lg = torch.empty((n, t, h, w, c), dtype=torch.float32, device=dev) if want_logits else None
        pr = torch.empty((n, t, h, w, c), dtype=torch.float32, device=dev) if (want_prob or True) else None
        pd = torch.empty((n, t, h, w), dtype=torch.int32, device=dev) if want_pred else None
        rc = _lib.lib.ukbb_fcn_forward_seq(self._h, C.c_void_p(xd.data_ptr()), n, h, w,
                                           C.c_void_p(lg.data_ptr() if lg is not None else None),
                                           C.c_void_p(pr.data_ptr()), C.c_void_p(pd.data_ptr() if pd is not None else None),
                                           C.c_void_p(stream or None))
        _lib.check(rc, 'ukbb_fcn_forward_seq')
        out: Dict[str, np.ndarray] = {}
        if want_logits:
            out['logits'] = lg.cpu().numpy()
        if want_prob:
            out['prob'] = pr.cpu().numpy()
        if want_pred:
            out['pred'] = pd.cpu().numpy()
        return out

    def run_cine(self, frames: np.ndarray, weight_R: int = 5, weight_r: float = 0.1, time_step: int = 1):
        """One slice position of the 'UNet-LSTM' branch of common/deploy_network_ao.py:129-183,189: frames float32
        [F,H,W] (normalised, padded) -> (prob [F,H,W,C] float32, pred [F,H,W] int32), circular windows centred
        on frames range(0, F, time_step) tiled on the device; the U-Net features of each frame are computed once."""
        import torch
        x = np.ascontiguousarray(frames, dtype=np.float32)
        if x.ndim != 3:
            raise ValueError('frames must be [F,H,W], got shape %s' % (frames.shape,))
        f, h, w = x.shape
        dev = torch.device('cuda', self.device)
        stream = torch.cuda.current_stream(dev).cuda_stream
        xd = torch.from_numpy(x).to(dev)
        pr = torch.empty((f, h, w, self.arch.n_class), dtype=torch.float32, device=dev)
        pd = torch.empty((f, h, w), dtype=torch.int32, device=dev)
        rc = _lib.lib.ukbb_fcn_forward_cine(self._h, C.c_void_p(xd.data_ptr()), f, h, w, int(weight_R), float(weight_r), int(time_step),
                                            C.c_void_p(pr.data_ptr()), C.c_void_p(pd.data_ptr()), C.c_void_p(stream or None))
        _lib.check(rc, 'ukbb_fcn_forward_cine')
        return pr.cpu().numpy(), pd.cpu().numpy()

    def run_cine_device(self, frames_ptr, f, h, w, prob_ptr, pred_ptr, weight_R=5, weight_r=0.1, time_step=1, stream=0):
        """``run_cine`` on device pointers (frames [F,H,W] float32, prob [F,H,W,C] float32, pred [F,H,W] int32), asynchronous on
        ``stream``: the form device_pipeline.aortic_lstm_sequence_device uses."""
        rc = _lib.lib.ukbb_fcn_forward_cine(self._h, C.c_void_p(frames_ptr), int(f), int(h), int(w), int(weight_R), float(weight_r),
                                            int(time_step), C.c_void_p(prob_ptr), C.c_void_p(pred_ptr or None), C.c_void_p(stream or None))
        _lib.check(rc, 'ukbb_fcn_forward_cine')

    def set_precision(self, precision: str):
        """'fp32' (default), 'bf16' (bf16 MFMA inputs, fp32 accumulate; BASELINE config 5) or 'f32x3' (fp32 results from three
        bf16 pieces per operand on the dense matrix cores; FCN head so far; include/ukbb_fcn.h UKBB_PREC_F32X3)."""
        code = {'fp32': 0, 'bf16': 1, 'f32x3': 2}[precision]
        _lib.check(_lib.lib.ukbb_fcn_set_precision(self._h, code), 'ukbb_fcn_set_precision')

    # -- measurement -----------------------------------------------------------
    def kernel_names(self):
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        return [_lib.lib.ukbb_fcn_kernel_name(self._h, i).decode() for i in range(n)]

    def kernel_macs(self):
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        return [_lib.lib.ukbb_fcn_kernel_macs(self._h, i) for i in range(n)]

    def kernel_mfma_macs(self):
        """Multiplies each kernel issues to the matrix pipe (Winograd / head algebra run fewer than the
        reference graph's algorithmic count returned by kernel_macs)."""
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        return [_lib.lib.ukbb_fcn_kernel_mfma_macs(self._h, i) for i in range(n)]

    def kernel_mfma_macs_issued(self):
        """MACs each launch issues to the matrix pipe INCLUDING tile / Winograd-region padding (= SQ_INSTS_MFMA x MACs per instruction)."""
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        return [_lib.lib.ukbb_fcn_kernel_mfma_macs_issued(self._h, i) for i in range(n)]

    def kernel_configs(self):
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        return [_lib.lib.ukbb_fcn_kernel_config(self._h, i) for i in range(n)]

    def set_timing(self, enable: bool):
        _lib.check(_lib.lib.ukbb_fcn_set_timing(self._h, int(enable)), 'ukbb_fcn_set_timing')

    def set_timing_kernel(self, index: int):
        _lib.check(_lib.lib.ukbb_fcn_set_timing_kernel(self._h, int(index)), 'ukbb_fcn_set_timing_kernel')

    def kernel_times(self, reset=True):
        n = _lib.lib.ukbb_fcn_num_kernels(self._h)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(_lib.lib.ukbb_fcn_kernel_times(self._h, ms, cnt, n, int(reset)), 'ukbb_fcn_kernel_times')
        return list(ms), list(cnt)

    def activation(self, name: str) -> np.ndarray:
        n = _lib.lib.ukbb_fcn_get_activation(self._h, name.encode(), None, 0)
        _lib.check(int(n), 'ukbb_fcn_get_activation')
        buf = np.empty(int(n), np.float32)
        _lib.check(int(_lib.lib.ukbb_fcn_get_activation(self._h, name.encode(), _lib.f32ptr(buf), n)),
                   'ukbb_fcn_get_activation')
        return buf


def load_model(model_path: str):
    """Resolve the reference's ``--model_path`` (a TF checkpoint prefix,
    ``demo_pipeline.py:63``): this repo's weight blob ``<model_path>.ukbbw`` if present, else the
    checkpoint-V2 files themselves (``tf_checkpoint.py``)."""
    for cand in (model_path, model_path + MODEL_EXT):
        if os.path.isfile(cand):
            try:
                return load_blob(cand)
            except ValueError:
                continue
    # the reference's own format: a TF checkpoint-V2 prefix (deploy_network.py:48-49)
    from . import tf_checkpoint
    if tf_checkpoint.is_checkpoint(model_path):
        return tf_checkpoint.checkpoint_to_params(model_path)
    raise FileNotFoundError(
        'no weight blob at %s%s and no TF checkpoint at %s.index (convert with '
        '`python -m ukbb_cardiac_amd.tf_checkpoint %s`)' % (model_path, MODEL_EXT, model_path, model_path))


class Session:
    """The slice of ``tf.Session`` the deploy scripts touch.

    Reference usage (``common/deploy_network.py:44-49,110-111``)::

        with tf.Session() as sess:
            saver = tf.train.import_meta_graph(model_path + '.meta'); saver.restore(sess, model_path)
            prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': x, 'training:0': False})

    Here ``Session(model_path)`` binds the model and ``run`` accepts exactly those
    tensor names (``common/train_network.py:142,151,198,199``); anything else
    raises, as TF would for an unknown tensor.
    """

    FETCHABLE = ('prob:0', 'pred:0', 'logits:0')

    def __init__(self, model_path: Optional[str] = None, arch: Optional[ModelArch] = None,
                 params: Optional[Params] = None, device: int = 0):
        if model_path is not None:
            arch, params = load_model(model_path)
        if arch is None or params is None:
            raise ValueError('Session needs model_path or (arch, params)')
        self.engine = Engine(arch, params, device)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        self.engine.close()

    def run(self, fetches, feed_dict):
        single = isinstance(fetches, str)
        names: Sequence[str] = [fetches] if single else list(fetches)
        for f in names:
            if f not in self.FETCHABLE:
                raise KeyError('unknown tensor %r (fetchable: %s)' % (f, ', '.join(self.FETCHABLE)))
        for k in feed_dict:
            if k not in ('image:0', 'training:0'):
                raise KeyError('unknown placeholder %r' % (k,))
        if 'image:0' not in feed_dict:
            raise KeyError("feed_dict lacks 'image:0'")
        if feed_dict.get('training:0', False):
            raise ValueError("'training:0' must be False: only the inference graph (BN moving statistics) exists")
        out = self.engine.run(feed_dict['image:0'], want_logits='logits:0' in names,
                              want_prob='prob:0' in names, want_pred='pred:0' in names)
        res = [out[f.split(':')[0]] for f in names]
        return res[0] if single else res
