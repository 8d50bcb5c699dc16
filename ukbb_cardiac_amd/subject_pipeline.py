"""Subject-level software pipeline of the sequence loop (common/deploy_network.py:80-131) on one GPU.

``device_pipeline.segment_sequence_device`` runs one subject start to finish: pageable H2D copy of the 80 MB volume,
percentiles, pack, forward, unpack, D2H, float64 expansion -- 54 ms per 500-slice subject in round 1 for 11 ms of
network time.  Here consecutive subjects overlap on three HIP streams with pinned staging buffers:

    copy-in stream    H2D of subject k+1 (pinned -> HBM) and its exact percentiles (radix select)
    compute stream    clip / rescale / pad / transpose, the FCN forward over all T*Z slices, label unpack + class counts of subject k
    copy-out stream   D2H of subject k-1's uint8 label volume (20 MB instead of 160 MB of float64) and counts

Slots (device buffers + pinned label buffer + events) rotate round-robin; a slot is reused only after its result has
been collected.  Pinned INPUT buffers form a pool of their own so that reader threads can decompress the next files
straight into them (``stage()`` + ``nifti.load(path, alloc=...)``) while earlier subjects are still on the GPU; a
buffer stays with its subject until the result has been taken (the ED / ES image frames the reference saves are cut
from it, deploy_network.py:145) and returns to the pool on ``Result.done()``.  The float64 volume the reference writes (deploy_network.py:92,136) is produced by whoever saves the file
(``labels_as_float64``), off this thread.  Results are identical to ``pipeline.segment_sequence`` (tests).

PyTorch is used for what the task allows it for: pinned / device memory, streams and events.
"""
import queue

import numpy as np

from . import _lib
from .device_pipeline import lerp_like_numpy, percentile_ranks
from .pipeline import pad_amounts


class _Slot:
    pass


class Staged:
    """A pinned input buffer on loan from the pipeline: ``array`` is its (X,Y,Z,T) Fortran-ordered float32 view."""

    def __init__(self, array, buf):
        self.array, self.buf = array, buf


class SubjectPipeline:
    HEADROOM = 4096                                         # bytes of writable pinned memory in front of every staged array

    def __init__(self, engine, max_shape, batch_slices=128, depth=3, thres=(1, 99), extra_inputs=2, pinned_inputs=True):
        """max_shape: largest (X, Y, Z, T) expected (buffers are sized for it; larger volumes re-allocate).
        extra_inputs: pinned input buffers beyond ``depth`` (one per reader thread that may hold one).
        pinned_inputs=False: no pinned input pool at all -- for cohorts whose volumes are produced on the device
        (``submit_generated``; ``stage`` / ``submit`` of host arrays then block forever and must not be used)."""
        import torch
        self.torch = torch
        self.engine = engine
        self.batch_slices = int(batch_slices)
        self.thres = tuple(thres)
        self.dev = torch.device('cuda', engine.device)
        self.s_in = torch.cuda.Stream(self.dev)
        self.s_cmp = torch.cuda.Stream(self.dev)
        self.s_out = torch.cuda.Stream(self.dev)
        self.depth = int(depth)
        self.slots = [self._make_slot(max_shape) for _ in range(self.depth)]
        self._next = 0
        self._inflight = []                                   # slots in submission order
        self._in_cap = int(np.prod(max_shape))
        self._in_free = queue.Queue()
        for _ in range(self.depth + int(extra_inputs) if pinned_inputs else 0):
            # HEADROOM bytes in front of every buffer: nifti.load's whole-file decoder writes the 352-byte NIfTI header there and the
            # voxels straight behind it, i.e. into the array stage() hands out (the view keeps the whole allocation alive)
            self._in_free.put(torch.empty(self._in_cap + self.HEADROOM // 4, dtype=torch.float32, pin_memory=True)[self.HEADROOM // 4:])
        import threading
        self._lock = threading.Lock()
        self._staged = {}                                     # id(array) -> Staged, for arrays handed out by stage()

    # ---- buffers ----------------------------------------------------------------------------------
    def _make_slot(self, shape):
        torch = self.torch
        X, Y, Z, T = shape
        X2, Y2 = pad_amounts(X, Y)[:2]
        s = _Slot()
        s.cap_vox, s.cap_pix = X * Y * Z * T, X2 * Y2 * Z * T
        s.pin_lab = torch.empty(s.cap_vox, dtype=torch.uint8, pin_memory=True)
        s.pin_cnt = torch.empty(T * 16, dtype=torch.int64, pin_memory=True)
        s.d_vol = torch.empty(s.cap_vox, dtype=torch.float32, device=self.dev)
        s.d_batch = torch.empty(s.cap_pix, dtype=torch.float32, device=self.dev)
        s.d_pred = torch.empty(s.cap_pix, dtype=torch.int32, device=self.dev)
        s.d_lab = torch.empty(s.cap_vox, dtype=torch.uint8, device=self.dev)
        s.d_cnt = torch.empty(T * 16, dtype=torch.int64, device=self.dev)
        s.ev_in, s.ev_cmp, s.ev_out = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        s.busy = False
        return s

    def stage(self, shape, dtype=np.float32, timeout=None):
        """Borrow a pinned input buffer (thread-safe; blocks while all are in use): ``Staged.array`` is a Fortran-ordered
        float32 (X,Y,Z,T) view to fill -- e.g. ``nifti.load(path, alloc=lambda sh, dt: pipe.stage(sh, dt).array)`` makes
        the file decompress straight into pinned memory.  Pass the Staged object (or its array) to ``submit``."""
        if np.dtype(dtype) != np.float32:
            raise TypeError('the device pipeline is exact for float32 volumes only')
        n = int(np.prod(shape))
        if len(shape) != 4 or n > self._in_cap:
            raise ValueError('volume %s does not fit the staging buffers sized for %d voxels' % (shape, self._in_cap))
        buf = self._in_free.get(timeout=timeout)            # returned by Result.done()
        st = Staged(buf.numpy()[:n].reshape(shape, order='F'), buf)
        with self._lock:
            self._staged[id(st.array)] = st
        return st

    def release(self, array):
        """Give back a buffer ``stage()`` handed out that will not be submitted after all (its file failed to load)."""
        with self._lock:
            st = self._staged.pop(id(array), None)
        if st is not None:
            self._in_free.put(st.buf)
            return True
        return False

    def _acquire(self, shape):
        slot = self.slots[self._next]
        if slot.busy:
            raise RuntimeError('pipeline full: collect() a result before staging another subject (depth %d)' % self.depth)
        X, Y, Z, T = shape
        if X * Y * Z * T > slot.cap_vox or pad_amounts(X, Y)[0] * pad_amounts(X, Y)[1] * Z * T > slot.cap_pix or T * 16 > slot.pin_cnt.numel():
            self.torch.cuda.synchronize(self.dev)
            self.slots[self._next] = slot = self._make_slot(shape)
        return slot

    # ---- submit / collect -----------------------------------------------------------------------------
    def submit(self, image):
        """Enqueue one (X,Y,Z,T) float32 volume: a ``Staged`` object / the array ``stage()`` handed out (used in place),
        or any other array (copied into a pinned buffer first: one host memcpy).  Returns once the exact percentiles
        of the volume are known (the copy-in stream is waited for, the compute stream is not)."""
        torch = self.torch
        if isinstance(image, Staged):
            st = image
            with self._lock:
                self._staged.pop(id(st.array), None)
        else:
            with self._lock:
                st = self._staged.pop(id(image), None)
            if st is None:
                if image.ndim != 4 or image.dtype != np.float32:
                    raise TypeError('expected a 4-D float32 (X,Y,Z,T) volume')
                st = self.stage(image.shape)
                with self._lock:
                    self._staged.pop(id(st.array), None)
                st.array[...] = image
        self._enqueue(st.array.shape, st, None)

    def submit_generated(self, shape, fill):
        """Enqueue one (X,Y,Z,T) float32 volume that is PRODUCED ON THE DEVICE: ``fill(d_ptr, n, stream)`` enqueues, on the copy-in
        stream it is given, whatever writes the n voxels (x fastest, like the file) to device address d_ptr -- it takes the place
        of the H2D copy; everything behind it (percentiles, pack, forward, unpack, labels to pinned host memory) is the same code.
        The Result of such a subject has no ``image``."""
        if len(shape) != 4:
            raise ValueError('expected an (X,Y,Z,T) shape, got %s' % (shape,))
        self._enqueue(tuple(int(v) for v in shape), None, fill)

    def _enqueue(self, shape, st, fill):
        torch = self.torch
        X, Y, Z, T = shape
        slot = self._acquire(shape)
        n = X * Y * Z * T
        slot.busy = True
        slot.shape = shape
        slot.staged = st
        self._next = (self._next + 1) % self.depth
        n_class = self.engine.arch.n_class
        X2, Y2, x_pre, _, y_pre, _ = pad_amounts(X, Y)
        nsl = T * Z
        with torch.cuda.stream(self.s_in):
            if st is not None:
                slot.d_vol[:n].copy_(st.buf[:n], non_blocking=True)
            else:
                # the slot's previous subject was collected (its pack kernel, the last reader of d_vol, has finished long ago)
                fill(slot.d_vol.data_ptr(), n, self.s_in.cuda_stream)
            # exact np.percentile(volume, (1, 99)): two neighbouring order statistics per percentile from the device
            # (4-pass radix select; synchronises the copy-in stream only), numpy's own interpolation on the host
            ranks, gammas = [], []
            for q in self.thres:
                k, k1, g = percentile_ranks(n, q)
                ranks += [k, k1]
                gammas.append(g)
            import ctypes as C
            r = (C.c_uint64 * len(ranks))(*ranks)
            out = np.empty(len(ranks), np.float32)
            _lib.check(_lib.lib.ukbb_fcn_select_kth(slot.d_vol.data_ptr(), n, r, len(ranks), _lib.f32ptr(out), self.s_in.cuda_stream),
                       'ukbb_fcn_select_kth')
            lo, hi = (lerp_like_numpy(out[2 * i], out[2 * i + 1], gammas[i]) for i in range(2))
            slot.ev_in.record(self.s_in)
        slot.clip = (lo, hi)
        with torch.cuda.stream(self.s_cmp):
            self.s_cmp.wait_event(slot.ev_in)
            cs = self.s_cmp.cuda_stream
            # element strides of the Fortran-ordered (X,Y,Z,T) volume
            _lib.check(_lib.lib.ukbb_fcn_rescale_pack(slot.d_vol.data_ptr(), X, Y, Z, T, 1, X, X * Y, X * Y * Z, float(lo), float(hi),
                                                      X2, Y2, x_pre, y_pre, slot.d_batch.data_ptr(), cs), 'ukbb_fcn_rescale_pack')
            self.engine.reserve(min(self.batch_slices, nsl), X2, Y2)
            px = X2 * Y2
            for i in range(0, nsl, self.batch_slices):
                m = min(self.batch_slices, nsl - i)
                self.engine.run_device(slot.d_batch.data_ptr() + 4 * i * px, m, X2, Y2, pred_ptr=slot.d_pred.data_ptr() + 4 * i * px, stream=cs)
            _lib.check(_lib.lib.ukbb_fcn_unpack_labels(slot.d_pred.data_ptr(), X, Y, Z, T, X2, Y2, x_pre, y_pre, n_class,
                                                       slot.d_lab.data_ptr(), slot.d_cnt.data_ptr(), cs), 'ukbb_fcn_unpack_labels')
            slot.ev_cmp.record(self.s_cmp)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(slot.ev_cmp)
            slot.pin_lab[:n].copy_(slot.d_lab[:n], non_blocking=True)
            slot.pin_cnt[:T * n_class].copy_(slot.d_cnt[:T * n_class], non_blocking=True)
            slot.ev_out.record(self.s_out)
        self._inflight.append(slot)

    def pending(self):
        return len(self._inflight)

    def collect(self, copy=True):
        """Oldest submitted subject -> Result (labels uint8 (X,Y,Z,T), counts int64 [T, n_class], clip (lo, hi), image =
        the staged input volume).  Call ``Result.done()`` when the image is no longer needed.
        copy=False: ``labels`` is a view of the slot's pinned buffer, valid only until ``depth`` more subjects have been submitted
        (for callers that consume or drop it at once)."""
        slot = self._inflight.pop(0)
        slot.ev_out.synchronize()
        X, Y, Z, T = slot.shape
        n = X * Y * Z * T
        n_class = self.engine.arch.n_class
        lab = slot.pin_lab.numpy()[:n].reshape(slot.shape, order='F')
        if copy:
            lab = lab.copy(order='F')
        cnt = slot.pin_cnt.numpy()[:T * n_class].reshape(T, n_class).copy()
        st, slot.staged = slot.staged, None
        slot.busy = False
        return Result(self, lab, cnt, slot.clip, st)

    def run(self, volumes):
        """Generator: segment an iterable of volumes with up to depth-1 subjects in flight; yields Results in order
        (each already ``done()``: the staged image of a result is only valid until the next one is requested)."""
        last = None
        for v in volumes:
            if self.pending() >= self.depth - 1:
                if last is not None:
                    last.done()
                last = self.collect()
                yield last
            self.submit(v)
        while self.pending():
            if last is not None:
                last.done()
            last = self.collect()
            yield last
        if last is not None:
            last.done()


class Result:
    def __init__(self, pipe, labels, counts, clip, staged):
        self._pipe, self.labels, self.counts, self.clip, self._staged = pipe, labels, counts, clip, staged

    @property
    def image(self):
        """The input volume as staged (NOT clipped; the reference's saved frames are, see device_pipeline.clip_like_reference);
        None for a subject generated on the device."""
        return None if self._staged is None else self._staged.array

    def done(self):
        if self._staged is not None:
            self._pipe._in_free.put(self._staged.buf)
            self._staged = None


def labels_as_float64(lab_u8):
    """The array the reference saves as seg_{seq}.nii.gz: np.zeros(image.shape) filled with int32 predictions
    (deploy_network.py:92,116) = float64 labels."""
    out = np.zeros(lab_u8.shape)
    out[...] = lab_u8
    return out
