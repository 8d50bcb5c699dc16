#!/usr/bin/env python3
"""Drop-in for the reference's ``common/deploy_network.py``: short-axis and
long-axis cine segmentation, NIfTI in -> label-map NIfTI out, with the network
evaluated by the MI355X HIP engine instead of a TensorFlow session.

Accepts the reference's command line verbatim (``demo_pipeline.py:63-64,89-96``)::

    python3 ukbb_cardiac_amd/deploy_network.py --seq_name sa --data_dir demo_image \
        --model_path trained_model/FCN_sa
    ... --seq_name la_4ch --seg4 ...

and writes the same files (``deploy_network.py:136-151,207-216``):
``seg_{seq}.nii.gz`` (float64, affine + pixdim of the input), ``{seq}_ED/ES.nii.gz``,
``seg_{seq}_ED/ES.nii.gz``; prefix ``seg4_`` with ``la_4ch --seg4``.
``--model_path`` names ``<model_path>.ukbbw``.

Extra flags (not in the reference): ``--device``, ``--batch_slices``,
``--num_shards`` / ``--shard_index`` (multi-GPU batch split).
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ukbb_cardiac_amd import measures, nifti, pipeline             # noqa: E402
from ukbb_cardiac_amd.flags import FlagError, FlagSet              # noqa: E402
from ukbb_cardiac_amd.shard import ClaimQueue, apply_cpu_set_from_env, default_device, shard_from_env   # noqa: E402


def define_flags():
    fs = FlagSet()                                      # reference: deploy_network.py:25-40
    fs.DEFINE_enum('seq_name', 'sa', ['sa', 'la_2ch', 'la_4ch'], 'Sequence name.')
    fs.DEFINE_string('data_dir', 'ukbb_cardiac_demo',
                     'Path to the data set directory, under which images are organised in '
                     'subdirectories for each subject.')
    fs.DEFINE_string('model_path', '', 'Path to the saved trained model.')
    fs.DEFINE_boolean('process_seq', True, 'Process a time sequence of images.')
    fs.DEFINE_boolean('save_seg', True, 'Save segmentation.')
    fs.DEFINE_boolean('seg4', False, 'Segment all the 4 chambers in long-axis 4 chamber view.')
    env_idx, env_cnt = shard_from_env()
    fs.DEFINE_integer('device', default_device(), 'HIP device ordinal (after HIP_VISIBLE_DEVICES); defaults to '
                      'LOCAL_RANK under torch.distributed.run.')
    fs.DEFINE_integer('batch_slices', 128, 'Slices per forward call.')
    fs.DEFINE_boolean('device_preproc', True, 'Percentile rescale, padding, transposes and label counting on the GPU '
                      '(float32 sequences; results identical to the host path).')
    fs.DEFINE_boolean('numpy1_casting', False, 'Rescale intensities with the float32 arithmetic numpy 1.x used when the reference '
                      'was written (1 ulp from numpy 2; host pre-processing only; INTEGRATION.md section 5).')
    fs.DEFINE_integer('io_threads', int(os.environ.get('UKBB_IO_THREADS', 8)), 'Reader threads (gzip NIfTI -> pinned staging) and writer threads (float64 label volume, gzip) '
                      'around the GPU in sequence mode; 0 = strictly sequential subjects as in the reference.')
    fs.DEFINE_enum('precision', 'fp32', ['fp32', 'f32x3'], 'Arithmetic of the matrix products: fp32 MFMA (default) or fp32 results from three '
                   'bf16 pieces per operand (UKBB_PREC_F32X3, include/ukbb_fcn.h; same labels, faster head).')
    fs.DEFINE_enum('label_gzip', 'small', list(nifti.LABEL_GZIP_MODES), 'Deflate of the label volumes: small = run-length tokens + dynamic Huffman '
                   '(typically below the size of zlib level 1, ~20x less CPU), fast = fixed Huffman (2-4x larger files), zlib = as nibabel. Same inflated bytes.')
    fs.DEFINE_string('output_csv', '', '--seq_name sa, sequence mode: also write the spreadsheet of short_axis/eval_ventricular_volume.py '
                     '(same columns, same arithmetic) from the per-frame class counts the GPU leaves behind -- no second pass over '
                     'seg_sa.nii.gz.  Subjects already segmented by an earlier run are measured from their files.')
    fs.DEFINE_integer('num_shards', env_cnt, 'Number of workers sharing data_dir.')
    fs.DEFINE_integer('shard_index', env_idx, 'This worker: subjects i with i % num_shards == shard_index.')
    fs.DEFINE_boolean('work_stealing', True, 'With num_shards > 1: after its own share a worker takes subjects of the other shards that '
                      'nobody has claimed yet (claim file <subject>/.claim.<seg>_<seq>, created with O_EXCL; claims of dead workers are swept). '
                      '--nowork_stealing = the strict i % num_shards split.')
    return fs


def seg_prefix(FLAGS):
    return 'seg4' if (FLAGS.seq_name == 'la_4ch' and FLAGS.seg4) else 'seg'


def save_sequence_outputs(data_dir, pre, seq, affine, pixdim, pred, frames):
    """The five files of deploy_network.py:136-151.  pred: (X,Y,Z,T) labels, stored as float64 (:92); frames: {'ED'|'ES':
    (image frame, label frame)}."""
    for fr, (img_fr, seg_fr) in frames.items():
        nifti.save(img_fr, '{0}/{1}_{2}.nii.gz'.format(data_dir, seq, fr), affine)
        nifti.save(seg_fr, '{0}/{1}_{2}_{3}.nii.gz'.format(data_dir, pre, seq, fr), affine)
    # seg_{seq}.nii.gz doubles as the 'already segmented, skip' marker (:66-67), so it is written LAST and -- like every
    # file nifti.save writes -- appears under its name only when complete (tmp file + os.replace): a worker killed
    # anywhere in here leaves a subject that the rerun segments again, never a half-written one that it skips.
    nifti.save(pred, '{0}/{1}_{2}.nii.gz'.format(data_dir, pre, seq), affine, pixdim, as_dtype=np.float64)


def run_pipelined(FLAGS, engine, data_list, log=print, csv_rows=None, queue=None):
    """Sequence mode with subjects overlapped: reader threads decompress the next files into pinned staging buffers,
    the GPU thread (this one) keeps up to two subjects in flight on three streams (subject_pipeline.SubjectPipeline),
    writer threads expand the uint8 labels to the reference's float64 volume, gzip and save.  Same files, byte for byte,
    as the sequential loop; log lines of a subject are emitted together when its result arrives."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from ukbb_cardiac_amd import device_pipeline
    from ukbb_cardiac_amd.subject_pipeline import SubjectPipeline
    start_time = time.time()
    seq, pre = FLAGS.seq_name, seg_prefix(FLAGS)
    def candidates(names, second=False):
        """(data, data_dir, image_name) of the subjects still to segment, in walk order, each claimed right before its read is
        scheduled (with --work_stealing a worker so holds at most window + depth claims; the rest of the list stays open)."""
        for data in names:
            data_dir = os.path.join(FLAGS.data_dir, data)
            if not os.path.isdir(data_dir) or os.path.exists('{0}/{1}_{2}.nii.gz'.format(data_dir, pre, seq)):
                if not second:
                    log(data)
                continue
            image_name = '{0}/{1}.nii.gz'.format(data_dir, seq)
            if not os.path.exists(image_name):
                if not second:
                    log(data)
                    log('  Directory {0} does not contain an image with file name {1}. Skip.'.format(data_dir, os.path.basename(image_name)))
                continue
            if queue is not None and not queue.take(data):
                continue                                    # another worker is on it
            yield (data, data_dir, image_name)

    def todo_items():
        yield from candidates(data_list)
        if queue is not None:                               # subjects a live worker held when we passed: finished by now, or orphaned
            yield from candidates(queue.second_chance(), second=True)

    def release(data):
        if queue is not None:
            queue.done(data)
    nthr = max(1, int(FLAGS.io_threads))
    window = nthr + 1                                   # reads allowed to run ahead of the GPU thread
    depth = 3
    state = {'pipe': None}
    mk = threading.Lock()

    def alloc(shape, dt):
        if len(shape) == 4 and dt == np.float32:
            with mk:
                if state['pipe'] is None:                  # sized by the first volume; bigger ones fall back below
                    state['pipe'] = SubjectPipeline(engine, shape, FLAGS.batch_slices, depth=depth, extra_inputs=window)
            try:
                return state['pipe'].stage(shape, dt).array, SubjectPipeline.HEADROOM
            except ValueError:
                pass
        return np.empty(shape, dt, order='F'), 0

    def read(item):
        t0 = time.time()
        handed = []

        def alloc_tracked(shape, dt):
            a, headroom = alloc(shape, dt)
            handed.append(a)
            return a, headroom
        try:
            nim = nifti.load(item[2], alloc=alloc_tracked)
        except BaseException:
            pipe = state['pipe']                            # a truncated / corrupt file: the pinned buffer goes back to the pool
            for a in handed:
                if pipe is not None:
                    pipe.release(a)
            raise
        return nim, time.time() - t0

    processed, table_time, writes = [], [], []
    readers, writers = ThreadPoolExecutor(nthr), ThreadPoolExecutor(nthr)
    futures = {}
    inflight = []                                       # (item, nim, t_submit)

    def finish(item, nim, t0):
        data, data_dir, image_name = item
        res = state['pipe'].collect()
        seg_time = time.time() - t0
        k_ed, k_es = device_pipeline.pick_ed_es_from_counts(res.counts, seq, FLAGS.seg4)
        log(data)
        log('  Reading {} ...'.format(image_name))
        log('  Segmenting full sequence ...')
        log('  Segmentation time = {:3f}s'.format(seg_time))
        log('  ED frame = {:d}, ES frame = {:d}'.format(k_ed, k_es))
        table_time.append(seg_time)
        processed.append(data)
        if csv_rows is not None:
            csv_rows[data] = measures.sa_row(res.counts, nim.header['pixdim'])
        if FLAGS.save_seg:
            log('  Saving segmentation ...')
            # the saved frames are the CLIPPED intensities (alias quirk)
            frames = {fr: (device_pipeline.clip_like_reference(res.image[:, :, :, k], res.clip), res.labels[:, :, :, k].astype(np.float64))
                      for fr, k in (('ED', k_ed), ('ES', k_es))}
            labels, affine, pixdim = res.labels, nim.affine, nim.header['pixdim']

            def write_then_release():
                try:
                    save_sequence_outputs(data_dir, pre, seq, affine, pixdim, labels, frames)
                finally:
                    release(data)                           # the claim outlives the subject's last file (seg_{seq}.nii.gz, written last)
            writes.append(writers.submit(write_then_release))
        else:
            release(data)
        res.done()

    try:
        from collections import deque
        todo = todo_items()
        ahead = deque()                                     # (item, future of its read), in walk order

        def top_up():
            while len(ahead) < window:
                item = next(todo, None)
                if item is None:
                    return
                ahead.append((item, readers.submit(read, item)))
        top_up()
        while ahead:
            item, fut = ahead.popleft()
            nim, _ = fut.result()
            image = nim.get_data()
            pipe = state['pipe']
            if image.ndim != 4 or image.dtype != np.float32 or pipe is None or image.size > pipe._in_cap:
                while inflight:                             # odd subject: drain, then take the sequential path
                    finish(*inflight.pop(0))
                log(item[0])
                try:
                    _sequence_subject(FLAGS, item, nim, None, engine, log, processed, table_time, csv_rows)
                finally:
                    release(item[0])
                top_up()
                continue
            if len(inflight) >= depth - 1:
                finish(*inflight.pop(0))
            pipe.submit(image)
            inflight.append((item, nim, time.time()))
            top_up()
        while inflight:
            finish(*inflight.pop(0))
        for w in writes:
            w.result()
    finally:
        readers.shutdown(wait=True)
        writers.shutdown(wait=True)
        if queue is not None:
            queue.release_all()                             # an exception above must not leave live-looking claims behind
    return processed, table_time, start_time


def _sequence_subject(FLAGS, item, nim, forward, engine, log, processed, table_time, csv_rows=None):
    """One subject of sequence mode, start to finish on this thread (deploy_network.py:80-151)."""
    data, data_dir, image_name = item
    seq, pre = FLAGS.seq_name, seg_prefix(FLAGS)
    image = nim.get_data()
    if image.ndim != 4:
        log('  {0}: expected a 4-D sequence, found shape {1}. Skip.'.format(image_name, image.shape))
        return
    log('  Segmenting full sequence ...')
    t0 = time.time()
    np1 = bool(getattr(FLAGS, 'numpy1_casting', False))
    on_device = engine is not None and getattr(FLAGS, 'device_preproc', False) and image.dtype == np.float32 and not np1
    if on_device:
        from ukbb_cardiac_amd import device_pipeline
        pred, aux = device_pipeline.segment_sequence_device(image, engine, FLAGS.batch_slices, return_aux=True)
    else:
        if forward is None:
            forward = lambda b: {'pred': engine.run(b, want_prob=False)['pred']}
        pred = pipeline.segment_sequence(image, forward, FLAGS.batch_slices, np1)   # clips `image` in place
    seg_time = time.time() - t0
    log('  Segmentation time = {:3f}s'.format(seg_time))
    table_time.append(seg_time)
    processed.append(data)
    if on_device:
        k_ed, k_es = device_pipeline.pick_ed_es_from_counts(aux['counts'], seq, FLAGS.seg4)
    else:
        k_ed, k_es = pipeline.pick_ed_es(pred, seq, FLAGS.seg4)
    log('  ED frame = {:d}, ES frame = {:d}'.format(k_ed, k_es))
    if csv_rows is not None:
        n_class = 4 if engine is None else engine.arch.n_class
        counts = aux['counts'] if on_device else measures.counts_from_labels(pred, n_class)
        csv_rows[data] = measures.sa_row(counts, nim.header['pixdim'])
    if FLAGS.save_seg:
        log('  Saving segmentation ...')
        frames = {}
        for fr, k in (('ED', k_ed), ('ES', k_es)):
            # the saved frames are the CLIPPED intensities (alias quirk)
            frame = device_pipeline.clip_like_reference(image[:, :, :, k], aux['clip']) if on_device else image[:, :, :, k]
            frames[fr] = (frame, pred[:, :, :, k])
        save_sequence_outputs(data_dir, pre, seq, nim.affine, nim.header['pixdim'], pred, frames)


def write_measures_csv(FLAGS, subjects, csv_rows, log=print):
    """The spreadsheet of short_axis/eval_ventricular_volume.py:28-79 for this worker's subjects: rows measured during this run
    come from the device counts; a subject segmented by an earlier run (skipped above) is measured from its files the way the
    evaluation script does.  Same inclusion rule (:35: image and segmentation both exist), same order (sorted directory names)."""
    rows = []
    # with --work_stealing this worker may also have segmented subjects of other shards (their rows are in csv_rows), and another
    # worker may have taken some of this one's: both then hold a row for it -- identical text -- and the merge keeps one
    for data in sorted(set(subjects) | set(csv_rows)):
        data_dir = os.path.join(FLAGS.data_dir, data)
        image_name, seg_name = '{0}/sa.nii.gz'.format(data_dir), '{0}/seg_sa.nii.gz'.format(data_dir)
        if data in csv_rows:
            rows.append((data, csv_rows[data]))
        elif os.path.exists(image_name) and os.path.exists(seg_name):
            seg = nifti.load(seg_name).get_data()
            rows.append((data, measures.sa_row(measures.counts_from_labels(seg, 4), nifti.load_header(image_name)['pixdim'])))
    path = measures.shard_csv_name(FLAGS.output_csv, FLAGS.shard_index, FLAGS.num_shards)
    measures.write_csv(path, measures.SA_COLUMNS, rows)
    log('Clinical measures of {0} subjects written to {1}'.format(len(rows), path))


def run(FLAGS, forward, log=print, engine=None):
    """The subject loop of deploy_network.py:52-225 with ``forward`` standing for sess.run.
    With ``engine`` (and --device_preproc) float32 sequences take the device pipeline."""
    start_time = time.time()
    seq, pre = FLAGS.seq_name, seg_prefix(FLAGS)
    # the static split (subject i -> shard i mod num_shards) and, on top of it, claim-file work stealing in sequence mode
    queue = ClaimQueue(FLAGS.data_dir, sorted(os.listdir(FLAGS.data_dir)), FLAGS.shard_index, FLAGS.num_shards, '%s_%s' % (pre, seq),
                       stealing=bool(getattr(FLAGS, 'work_stealing', False)) and FLAGS.process_seq)
    data_list = list(queue)
    processed, table_time = [], []
    csv_rows = None
    if getattr(FLAGS, 'output_csv', ''):
        if seq != 'sa' or not FLAGS.process_seq:
            raise ValueError('--output_csv writes the table of short_axis/eval_ventricular_volume.py: it needs --seq_name sa in sequence mode')
        csv_rows = {}
    shard_subjects = list(queue.static)                 # whose earlier-run results this worker measures for --output_csv
    if (FLAGS.process_seq and engine is not None and getattr(FLAGS, 'device_preproc', False) and getattr(FLAGS, 'io_threads', 0) > 0
            and not getattr(FLAGS, 'numpy1_casting', False)):
        processed, table_time, _ = run_pipelined(FLAGS, engine, data_list, log, csv_rows, queue)
        data_list = []
    def one_subject(data, second):
        """One entry of the walk; ``second``: a subject another worker held when this one first came by (--work_stealing)."""
        data_dir = os.path.join(FLAGS.data_dir, data)
        if not second:
            log(data)
        if not os.path.isdir(data_dir):
            return
        if os.path.exists('{0}/{1}_{2}.nii.gz'.format(data_dir, pre, seq)):
            return                                   # already segmented: idempotent / resumable (:62-67)
        if FLAGS.process_seq:
            image_name = '{0}/{1}.nii.gz'.format(data_dir, seq)
            if not os.path.exists(image_name):
                if not second:
                    log('  Directory {0} does not contain an image with file name {1}. Skip.'.format(
                        data_dir, os.path.basename(image_name)))
                return
            if not queue.take(data):
                return                               # claimed by another worker (--work_stealing)
            try:
                if second:
                    log(data)
                log('  Reading {} ...'.format(image_name))
                nim = nifti.load(image_name)
                _sequence_subject(FLAGS, (data, data_dir, image_name), nim, forward, engine, log, processed, table_time, csv_rows)
            finally:
                queue.done(data)
        else:
            names = {fr: '{0}/{1}_{2}.nii.gz'.format(data_dir, seq, fr) for fr in ('ED', 'ES')}
            if not all(os.path.exists(p) for p in names.values()):
                log('  Directory {0} does not contain an image with file name {1} or {2}. Skip.'.format(
                    data_dir, os.path.basename(names['ED']), os.path.basename(names['ES'])))
                return
            for fr in ('ED', 'ES'):
                log('  Reading {} ...'.format(names[fr]))
                nim = nifti.load(names[fr])
                image = nim.get_data()
                log('  Segmenting {} frame ...'.format(fr))
                t0 = time.time()
                pred = pipeline.segment_frame(image, forward, FLAGS.batch_slices, bool(getattr(FLAGS, 'numpy1_casting', False)))
                seg_time = time.time() - t0
                log('  Segmentation time = {:3f}s'.format(seg_time))
                table_time.append(seg_time)
                processed.append(data)
                if FLAGS.save_seg:
                    log('  Saving segmentation ...')
                    nifti.save(pred, '{0}/{1}_{2}_{3}.nii.gz'.format(data_dir, pre, seq, fr), nim.affine,
                               nim.header['pixdim'])
    for data in data_list:
        one_subject(data, False)
    for data in queue.second_chance():                  # finished by its owner meanwhile (skip-if-exists), or orphaned (stale claim)
        one_subject(data, True)
    if csv_rows is not None:
        write_measures_csv(FLAGS, shard_subjects, csv_rows, log)
    if table_time:
        log('Average segmentation time = {:.3f}s per {}'.format(float(np.mean(table_time)),
                                                               'sequence' if FLAGS.process_seq else 'frame'))
    process_time = time.time() - start_time
    if processed:
        log('Including image I/O and device resource allocation, it took {:.3f}s for processing {:d} subjects '
            '({:.3f}s per subjects).'.format(process_time, len(processed), process_time / len(processed)))
    return processed


def main(argv=None):
    fs = define_flags()
    try:
        FLAGS, rest = fs.parse(sys.argv[1:] if argv is None else argv)
    except FlagError as e:
        sys.exit('FATAL Flags parsing error: %s\n%s' % (e, fs.usage()))
    if 'CUDA_VISIBLE_DEVICES' in os.environ and 'HIP_VISIBLE_DEVICES' not in os.environ:
        os.environ['HIP_VISIBLE_DEVICES'] = os.environ['CUDA_VISIBLE_DEVICES']   # demo_pipeline.py:25,63
    apply_cpu_set_from_env()                             # shard.launch's per-worker CPU set, before the first GPU call starts threads
    from ukbb_cardiac_amd.engine import Session          # raises if the HIP library is missing
    with Session(FLAGS.model_path, device=FLAGS.device) as sess:
        if FLAGS.precision != 'fp32':
            sess.engine.set_precision(FLAGS.precision)
        nifti.set_label_gzip(FLAGS.label_gzip)
        print('Start deployment on the data set ...')

        def forward(batch):
            pred = sess.run('pred:0', feed_dict={'image:0': batch, 'training:0': False})
            return {'pred': pred}
        run(FLAGS, forward, engine=sess.engine)


if __name__ == '__main__':
    main()
