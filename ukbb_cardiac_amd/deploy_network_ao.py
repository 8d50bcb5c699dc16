#!/usr/bin/env python3
"""Drop-in for the reference's ``common/deploy_network_ao.py`` (aortic cine
segmentation) on the MI355X HIP engine.

Command line as in ``demo_pipeline.py:116-117``.  Implemented: ``--model UNet``
(frame-wise 2-D U-Net, ``deploy_network_ao.py:111-128``) in sequence and ED/ES
mode, and the reference's default ``--model UNet-LSTM`` (U-Net features +
bidirectional ConvLSTM over circular 9-frame windows with weighted tiling,
``:129-183``) in sequence mode, any ``--time_step``.  ``Temporal-UNet`` is not
built and is refused with a clear message rather than silently replaced.

Output: ``seg_ao.nii.gz`` int32 with the input's affine and pixdim (``:189-196``).
"""
import os
import sys
import time

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ukbb_cardiac_amd import measures, nifti, pipeline             # noqa: E402
from ukbb_cardiac_amd.flags import FlagError, FlagSet              # noqa: E402
from ukbb_cardiac_amd.shard import default_device, shard_from_env, subjects_for_shard   # noqa: E402


def define_flags():
    fs = FlagSet()                                      # reference: deploy_network_ao.py:25-49
    fs.DEFINE_integer('time_step', 1, 'Time step during deployment of LSTM.')
    fs.DEFINE_enum('seq_name', 'ao', ['ao'], 'Sequence name.')
    fs.DEFINE_enum('model', 'UNet-LSTM', ['UNet', 'UNet-LSTM', 'Temporal-UNet'], 'Model name.')
    fs.DEFINE_string('data_dir', 'Biobank_ao/validation',
                     'Path to the test set directory, under which images are organised in '
                     'subdirectories for each subject.')
    fs.DEFINE_string('model_path', '', 'Path to the saved trained model.')
    fs.DEFINE_boolean('process_seq', True, 'Process a time sequence of images.')
    fs.DEFINE_boolean('save_seg', True, 'Save segmentation.')
    fs.DEFINE_boolean('z_score', True, 'Normalise the image intensity to z-score. Otherwise, rescale the intensity.')
    fs.DEFINE_integer('weight_R', 5, 'Radius of the weighting window.')
    fs.DEFINE_float('weight_r', 0.1, 'Power of weight for the seq2seq loss. 0: uniform; 1: linear; 2: square.')
    env_idx, env_cnt = shard_from_env()
    fs.DEFINE_integer('device', default_device(), 'HIP device ordinal (after HIP_VISIBLE_DEVICES); defaults to '
                      'LOCAL_RANK under torch.distributed.run.')
    fs.DEFINE_integer('batch_slices', 64, 'Slices per forward call.')
    fs.DEFINE_integer('io_threads', min(2, int(os.environ.get('UKBB_IO_THREADS', 2))), 'Sequence mode: threads that read (inflate) the next cines ahead of the GPU and threads that '
                      'write finished segmentations behind it; 0 = strictly sequential subjects as in the reference.')
    fs.DEFINE_boolean('device_preproc', True, 'Sequences: z-score, padding, transposes and the argmax on the GPU '
                      '(bit-identical to the host path; --nodevice_preproc restores it).')
    fs.DEFINE_enum('precision', 'fp32', ['fp32', 'bf16'], 'Arithmetic of the U-Net convolutions: fp32 MFMA (default) or bf16 MFMA operands with fp32 '
                   'accumulation (UKBB_PREC_BF16, include/ukbb_fcn.h; --model UNet: bf16 activations in HBM too, 2.8x the fp32 rate, '
                   'Dice 0.99 against fp32; BASELINE config 5.  Default UNet-LSTM model: the same U-Net plan, ConvLSTM on the bf16 matrix '
                   'instruction with bf16 hidden maps and fp32 cell state, 2.3x the fp32 rate, Dice >= 0.98 against fp32).')
    fs.DEFINE_enum('label_gzip', 'small', list(nifti.LABEL_GZIP_MODES), 'Deflate of the label volumes: small = run-length tokens + dynamic Huffman '
                   '(typically below the size of zlib level 1; never above it on segmentation-like volumes), fast = fixed Huffman (larger files), zlib = as nibabel.  Same inflated bytes.')
    fs.DEFINE_string('output_csv', '', 'Sequence mode: also write the spreadsheet of aortic/eval_aortic_area.py (same columns and arithmetic) from '
                     'the per-frame class counts the GPU leaves behind.  Of the evaluation script\'s quality control '
                     '(cardiac_utils.aorta_pass_quality_control, eval_aortic_area.py:68-69) the criteria that need only the areas are applied '
                     '(1: zero area in a frame, 4: abrupt change between adjacent frames, 5: max / min >= 2) and a failing subject is dropped with '
                     'the script\'s message; criteria 2 (image noise) and 3 (connected components) are NOT applied.')
    fs.DEFINE_boolean('aortic_qc', True, 'With --output_csv: apply the count-only quality-control criteria above (false: every segmented subject gets a row).')
    fs.DEFINE_string('pressure_csv', '', 'With --output_csv: the blood-pressure spreadsheet of eval_aortic_area.py:41-46 for the distensibility columns '
                     '(left empty without it).')
    fs.DEFINE_integer('num_shards', env_cnt, 'Number of workers sharing data_dir.')
    fs.DEFINE_integer('shard_index', env_idx, 'This worker: subjects i with i % num_shards == shard_index.')
    return fs


def _pp(central_pp, data, log=print):
    """central_pp.loc[int(data)] of eval_aortic_area.py:80; None (no distensibility) when no spreadsheet was given."""
    if not central_pp:
        return None
    try:
        key = str(int(data))
    except ValueError:
        key = str(data)
    if key not in central_pp:
        # the reference's central_pp.loc[int(data)] raises KeyError here and the whole evaluation stops; this script keeps the areas
        log('  Warning: subject {0} is not in the pressure spreadsheet: distensibility left empty.'.format(data))
    return central_pp.get(key, float('nan'))


def run(FLAGS, forward, log=print, cine_forward=None, engine=None):
    """``forward`` stands for the frame-wise sess.run ('UNet'); ``cine_forward`` for the windowed one ('UNet-LSTM').
    With ``engine`` (and --device_preproc, --z_score) float32 UNet-LSTM sequences take device_pipeline.aortic_lstm_sequence_device."""
    if FLAGS.model == 'Temporal-UNet':
        raise NotImplementedError("--model Temporal-UNet (common/network_ao.py:67-114, 3-D convolutions) is not built")
    if FLAGS.model == 'UNet-LSTM':
        if cine_forward is None:
            raise ValueError('--model UNet-LSTM needs a UNet-LSTM model (cine_forward)')
        if FLAGS.time_step < 1:
            raise ValueError('--time_step %d: range(0, T, time_step) needs a positive step '
                             '(common/deploy_network_ao.py:147)' % FLAGS.time_step)
    start_time = time.time()
    data_list = subjects_for_shard(sorted(os.listdir(FLAGS.data_dir)), FLAGS.shard_index, FLAGS.num_shards)
    processed = []
    seq = FLAGS.seq_name
    csv_rows = None
    if getattr(FLAGS, 'output_csv', ''):
        if not FLAGS.process_seq:
            raise ValueError('--output_csv writes the table of aortic/eval_aortic_area.py: it needs sequence mode')
        csv_rows = {}
        central_pp = measures.read_central_pp(FLAGS.pressure_csv) if getattr(FLAGS, 'pressure_csv', '') else {}

        def _qc_row(counts, pixdim, pp):
            """The subject's table line, or None when the count-only quality control drops it (the script's own message is printed)."""
            if getattr(FLAGS, 'aortic_qc', True):
                ok, why = measures.aorta_qc_from_counts(counts)
                if not ok:
                    log(why)
                    return None
            return measures.ao_row(counts, pixdim, pp)
    # Sequence mode with --io_threads > 0: the next cines are read (inflated) by reader threads while the GPU works on this one,
    # and the segmentation files are written behind it; order of subjects, log lines and files are those of the sequential loop.
    nthr = int(getattr(FLAGS, 'io_threads', 0)) if FLAGS.process_seq else 0
    readers = writers = None
    reads, writes = {}, []
    if nthr > 0:
        from concurrent.futures import ThreadPoolExecutor
        readers, writers = ThreadPoolExecutor(nthr), ThreadPoolExecutor(nthr)
        names = ['{0}/{1}.nii.gz'.format(os.path.join(FLAGS.data_dir, d), seq) for d in data_list]
        names = [n if os.path.isdir(os.path.dirname(n)) and os.path.exists(n) else None for n in names]
        ahead = [0]

        def prefetch(upto):
            while ahead[0] < min(upto, len(names)):
                if names[ahead[0]] is not None:
                    reads[ahead[0]] = readers.submit(nifti.load, names[ahead[0]])
                ahead[0] += 1

    def save(*args):
        if writers is not None:
            writes.append(writers.submit(nifti.save, *args))
        else:
            nifti.save(*args)

    for idx, data in enumerate(data_list):
        log(data)
        data_dir = os.path.join(FLAGS.data_dir, data)
        if not os.path.isdir(data_dir):
            continue
        if FLAGS.process_seq:
            image_name = '{0}/{1}.nii.gz'.format(data_dir, seq)
            if not os.path.exists(image_name):
                log('  Directory {0} does not contain an image with file name {1}. Skip.'.format(
                    data_dir, os.path.basename(image_name)))
                continue
            log('  Reading {} ...'.format(image_name))
            if readers is not None:
                prefetch(idx + 1 + nthr)
                nim = reads.pop(idx).result() if idx in reads else nifti.load(image_name)
            else:
                nim = nifti.load(image_name)
            image = nim.get_data()
            log('  Segmenting full sequence ...')
            t0 = time.time()
            on_device = (engine is not None and getattr(FLAGS, 'device_preproc', False)
                         and FLAGS.z_score and image.ndim == 4 and image.dtype == np.float32)
            if on_device:                                         # the device z-score mirrors numpy internals: verify once
                from ukbb_cardiac_amd.device_pipeline import device_zscore_matches_numpy
                on_device = device_zscore_matches_numpy(engine, warn=log)
            counts = None
            if on_device and FLAGS.model == 'UNet-LSTM':
                from ukbb_cardiac_amd.device_pipeline import aortic_lstm_sequence_device
                pred, aux = aortic_lstm_sequence_device(image, engine, True, FLAGS.weight_R, FLAGS.weight_r, FLAGS.time_step,
                                                        return_aux='counts')
                counts = aux['counts']
            elif on_device:
                from ukbb_cardiac_amd.device_pipeline import aortic_unet_sequence_device
                pred, aux = aortic_unet_sequence_device(image, engine, FLAGS.batch_slices, return_aux=True)
                counts = aux['counts']
            else:
                if FLAGS.model == 'UNet-LSTM':
                    prob = pipeline.aortic_lstm_prob_sequence(image, cine_forward, FLAGS.z_score, FLAGS.weight_R, FLAGS.weight_r,
                                                              time_step=FLAGS.time_step)
                else:
                    prob = pipeline.aortic_prob_sequence(image, forward, FLAGS.z_score, FLAGS.batch_slices)
                pred = np.argmax(prob, axis=-1).astype(np.int32)      # host argmax, as :189
            if FLAGS.save_seg:
                log('  Saving segmentation ...')
                save(pred, '{0}/seg_{1}.nii.gz'.format(data_dir, seq), nim.affine, nim.header['pixdim'])
            log('  Segmentation time = {:3f}s'.format(time.time() - t0))
            processed.append(data)
            if csv_rows is not None:
                if counts is None:
                    counts = measures.counts_from_labels(pred, 3)
                csv_rows[data] = _qc_row(counts, nim.header['pixdim'], _pp(central_pp, data, log))
        else:
            if FLAGS.model == 'UNet-LSTM':                             # reference: deploy_network_ao.py:202-205
                log('UNet-LSTM does not support frame-wise segmentation. Please use the -process_seq flag.')
                return processed
            names = {fr: '{0}/{1}_{2}.nii.gz'.format(data_dir, seq, fr) for fr in ('ED', 'ES')}
            if not all(os.path.exists(p) for p in names.values()):
                log('  Directory {0} does not contain an image with file name {1} or {2}. Skip.'.format(
                    data_dir, os.path.basename(names['ED']), os.path.basename(names['ES'])))
                continue
            for fr in ('ED', 'ES'):
                log('  Reading {} ...'.format(names[fr]))
                nim = nifti.load(names[fr])
                t0 = time.time()
                pred = pipeline.aortic_segment_frame(nim.get_data(), forward, FLAGS.z_score, FLAGS.batch_slices)
                log('  Segmentation time = {:3f}s'.format(time.time() - t0))
                if FLAGS.save_seg:
                    log('  Saving segmentation ...')
                    nifti.save(pred, '{0}/seg_{1}_{2}.nii.gz'.format(data_dir, seq, fr), nim.affine,
                               nim.header['pixdim'])
            processed.append(data)
    if readers is not None:
        try:
            for w in writes:
                w.result()
        finally:
            readers.shutdown(wait=True)
            writers.shutdown(wait=True)
    if csv_rows is not None:
        rows = []
        for data in data_list:                                  # eval_aortic_area.py:50-58: image and segmentation both exist
            data_dir = os.path.join(FLAGS.data_dir, data)
            image_name, seg_name = os.path.join(data_dir, 'ao.nii.gz'), os.path.join(data_dir, 'seg_ao.nii.gz')
            if data in csv_rows:
                row = csv_rows[data]
            elif os.path.exists(image_name) and os.path.exists(seg_name):
                log(data)
                seg = nifti.load(seg_name).get_data()
                row = _qc_row(measures.counts_from_labels(seg, 3), nifti.load_header(image_name)['pixdim'], _pp(central_pp, data, log))
            else:
                continue
            if row is not None:                                 # None: dropped by the quality control, as eval_aortic_area.py:68-69
                rows.append((data, row))
        path = measures.shard_csv_name(FLAGS.output_csv, FLAGS.shard_index, FLAGS.num_shards)
        measures.write_csv(path, measures.AO_COLUMNS, rows)
        log('Aortic areas of {0} subjects written to {1}'.format(len(rows), path))
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
        os.environ['HIP_VISIBLE_DEVICES'] = os.environ['CUDA_VISIBLE_DEVICES']
    if FLAGS.model == 'Temporal-UNet':
        sys.exit("Error: --model Temporal-UNet is not available on the HIP engine (see DESIGN.md section 7).")
    from ukbb_cardiac_amd.shard import apply_cpu_set_from_env
    apply_cpu_set_from_env()                             # shard.launch's per-worker CPU set, before the first GPU call starts threads
    from ukbb_cardiac_amd.arch import KIND_UNET_LSTM
    from ukbb_cardiac_amd.engine import Session
    nifti.set_label_gzip(FLAGS.label_gzip)
    with Session(FLAGS.model_path, device=FLAGS.device) as sess:
        is_lstm = sess.engine.arch.kind == KIND_UNET_LSTM
        if is_lstm != (FLAGS.model == 'UNet-LSTM'):
            sys.exit('Error: --model %s but %s holds a %s model.' % (FLAGS.model, FLAGS.model_path, sess.engine.arch.name))
        if FLAGS.precision != 'fp32':
            sess.engine.set_precision(FLAGS.precision)
        print('Start evaluating on the test set ...')

        def forward(batch):
            prob, pred = sess.run(['prob:0', 'pred:0'], feed_dict={'image:0': batch, 'training:0': False})
            return {'prob': prob, 'pred': pred}

        def cine_forward(frames, weight_R, weight_r, time_step=1):
            return sess.engine.run_cine(frames, weight_R, weight_r, time_step)[0]
        run(FLAGS, forward, cine_forward=cine_forward, engine=sess.engine)


if __name__ == '__main__':
    main()
