"""Clinical measures straight from the per-frame class counts the device pipeline already produces
(``device_pipeline.segment_sequence_device(..., return_aux=True)['counts']`` = ``ukbb_fcn_unpack_labels``),
i.e. without re-reading the 4-D label file the way the evaluation scripts do.

Same arithmetic, same order as short_axis/eval_ventricular_volume.py:40-71 and
aortic/eval_aortic_area.py:60-78: ``np.sum(seg == k, axis=(0,1,2))`` IS the count column k."""
import numpy as np


def ventricular_volumes(counts, pixdim, n_frames=None):
    """counts [T, n_class] (classes: 1 LV cavity, 2 myocardium, 3 RV cavity); pixdim = NIfTI header pixdim
    (dx, dy, dz at [1:4], frame duration at [4]).  Returns the dict of eval_ventricular_volume.py:57-71."""
    counts = np.asarray(counts)
    pixdim = np.asarray(pixdim)
    T = counts.shape[0] if n_frames is None else n_frames
    pd = pixdim[1:4]
    volume_per_pix = pd[0] * pd[1] * pd[2] * 1e-3                       # :42
    density = 1.05                                                      # :43
    duration_per_cycle = T * pixdim[4]                                  # :46 (header dim[4] = T)
    heart_rate = 60.0 / duration_per_cycle
    vol_t = counts[:, 1] * volume_per_pix                               # :54
    frame = {'ED': 0, 'ES': int(np.argmin(vol_t))}                      # :53-55
    val = {}
    for fr_name, fr in frame.items():                                   # :58-62
        val['LV{0}V'.format(fr_name)] = counts[fr, 1] * volume_per_pix
        val['LV{0}M'.format(fr_name)] = counts[fr, 2] * volume_per_pix * density
        val['RV{0}V'.format(fr_name)] = counts[fr, 3] * volume_per_pix
    val['LVSV'] = val['LVEDV'] - val['LVESV']                           # :64-70
    val['LVCO'] = val['LVSV'] * heart_rate * 1e-3
    val['LVEF'] = val['LVSV'] / val['LVEDV'] * 100
    val['RVSV'] = val['RVEDV'] - val['RVESV']
    val['RVCO'] = val['RVSV'] * heart_rate * 1e-3
    val['RVEF'] = val['RVSV'] / val['RVEDV'] * 100
    val['ES_frame'] = frame['ES']
    return val


def aortic_areas(counts, pixdim, central_pp=None):
    """counts [T, 3] (1 ascending, 2 descending aorta) -> eval_aortic_area.py:70-78 per vessel; the
    distensibility needs the subject's central pulse pressure (a CSV column in the reference, :41-46)."""
    counts = np.asarray(counts)
    dx, dy = np.asarray(pixdim)[1:3]
    area_per_pixel = dx * dy                                            # :60
    val = {}
    for l_name, l in (('AAo', 1), ('DAo', 2)):                          # :71-77
        A = counts[:, l] * area_per_pixel
        val[l_name] = {'max area': A.max(), 'min area': A.min()}
        if central_pp is not None:
            val[l_name]['distensibility'] = (A.max() - A.min()) / (A.min() * central_pp) * 1e3
    return val


# ---- the spreadsheets of the evaluation scripts, written by the deploy scripts (--output_csv) ------------------------
SA_COLUMNS = ['LVEDV (mL)', 'LVESV (mL)', 'LVSV (mL)', 'LVEF (%)', 'LVCO (L/min)', 'LVM (g)',
              'RVEDV (mL)', 'RVESV (mL)', 'RVSV (mL)', 'RVEF (%)']                       # eval_ventricular_volume.py:76-78
AO_COLUMNS = ['AAo max area (mm2)', 'AAo min area (mm2)', 'AAo distensibility (10-3 mmHg-1)',
              'DAo max area (mm2)', 'DAo min area (mm2)', 'DAo distensibility (10-3 mmHg-1)']   # eval_aortic_area.py:93-95


def sa_row(counts, pixdim):
    """The table line of eval_ventricular_volume.py:72-73 from counts [T, >= 4]."""
    v = ventricular_volumes(counts, pixdim)
    return [v['LVEDV'], v['LVESV'], v['LVSV'], v['LVEF'], v['LVCO'], v['LVEDM'], v['RVEDV'], v['RVESV'], v['RVSV'], v['RVEF']]


def ao_row(counts, pixdim, central_pp=None):
    """The table line of eval_aortic_area.py:83-84 from counts [T, 3]; distensibility is NaN without a pulse pressure."""
    v = aortic_areas(counts, pixdim, central_pp)
    nan = float('nan')
    return [v['AAo']['max area'], v['AAo']['min area'], v['AAo'].get('distensibility', nan),
            v['DAo']['max area'], v['DAo']['min area'], v['DAo'].get('distensibility', nan)]


def aorta_qc_from_counts(counts):
    """The criteria of cardiac_utils.aorta_pass_quality_control (reference common/cardiac_utils.py:1739-1796, called by
    aortic/eval_aortic_area.py:68-69) that need nothing but the per-frame areas, in the script's order per label (AAo = 1, then
    DAo = 2): 1 -- the area is 0 in no frame (:1741-1749); 4 -- no adjacent-frame area ratio >= 2 or <= 0.5, frame 0 against the LAST
    frame as the script's A[t-1] does (:1782-1788); 5 -- max / min area < 2 (:1790-1795).  Criteria 2 (intensity ratio) and 3
    (connected components) need the image / the label map itself and are NOT applied here.
    counts: [T, 3].  Returns (passed, message) with the script's own message for the first failing criterion."""
    counts = np.asarray(counts)
    T = counts.shape[0]
    for l_name, l in (('AAo', 1), ('DAo', 2)):
        A = counts[:, l]
        for t in range(T):
            if A[t] == 0:
                return False, 'The area of {0} is 0 at time frame {1}.'.format(l_name, t)
        for t in range(T):
            ratio = A[t] / float(A[t - 1])
            if ratio >= 2 or ratio <= 0.5:
                return False, 'There is abrupt change of area at time frame {0}.'.format(t)
        if np.max(A) / np.min(A) >= 2:
            return False, 'There is large change of area between maximum and minimum areas.'
    return True, ''


def counts_from_labels(seg, n_class):
    """[T, n_class] voxel counts of a (X,Y,Z,T) label volume: what np.sum(seg == k, axis=(0, 1, 2)) gives the scripts."""
    seg = np.asarray(seg)
    return np.stack([np.sum(seg == k, axis=(0, 1, 2)) for k in range(n_class)], axis=1).astype(np.int64)


def _fmt(x):
    """pandas' to_csv cell for a float: repr of the Python float (shortest round-trip), empty for NaN."""
    x = float(x)
    return '' if x != x else repr(x)


def write_csv(path, columns, rows):
    """rows: [(subject, [values])] -> the file ``pd.DataFrame(table, index=subjects, columns=columns).to_csv(path)``
    writes (eval_ventricular_volume.py:75-79): header with an empty index label, one line per subject, '\\n' line ends.
    Written under a temporary name and renamed."""
    import csv
    import io
    import os
    buf = io.StringIO()
    wr = csv.writer(buf, lineterminator='\n')
    wr.writerow([''] + list(columns))
    for subject, vals in rows:
        wr.writerow([subject] + [_fmt(v) for v in vals])
    tmp = '%s.tmp.%d' % (path, os.getpid())
    with open(tmp, 'w', newline='') as f:
        f.write(buf.getvalue())
    os.replace(tmp, path)


def shard_csv_name(path, shard_index, num_shards):
    """Where worker shard_index of num_shards writes its part of ``path`` (merged by merge_shard_csv / shard.launch)."""
    return path if num_shards <= 1 else '%s.shard%d-of-%d' % (path, shard_index, num_shards)


def merge_shard_csv(path, num_shards, remove=True):
    """Concatenate the per-worker parts into ``path`` with the subjects in sorted order (the reference's loop order,
    eval_ventricular_volume.py:28).  Missing parts (a worker with no subjects writes none) are skipped."""
    import csv
    import os
    header, rows, parts = None, [], []
    for i in range(num_shards):
        part = shard_csv_name(path, i, num_shards)
        if not os.path.exists(part):
            continue
        parts.append(part)
        with open(part, newline='') as f:
            rd = list(csv.reader(f))
        if rd:
            header = header or rd[0]
            rows += rd[1:]
    if header is None:
        return False
    rows.sort(key=lambda r: r[0])
    # one row per subject: with work stealing the worker that segmented a subject and the static owner of its index may both
    # have measured it (same counts, same text); the first one stays
    rows = [r for i, r in enumerate(rows) if i == 0 or r[0] != rows[i - 1][0]]
    tmp = '%s.tmp.%d' % (path, os.getpid())
    with open(tmp, 'w', newline='') as f:
        wr = csv.writer(f, lineterminator='\n')
        wr.writerow(header)
        wr.writerows(rows)
    os.replace(tmp, path)
    if remove and num_shards > 1:
        for part in parts:
            os.remove(part)
    return True


def read_central_pp(pressure_csv):
    """{subject id (str): central pulse pressure} as eval_aortic_area.py:41-46 derives it: the mean of the two
    'Central pulse pressure during PWA' columns 12678-2.0 / 12678-2.1 (NaNs skipped), values < 10 mmHg discarded."""
    import csv
    with open(pressure_csv, newline='') as f:
        rd = list(csv.reader(f))
    h0, h1 = rd[0], rd[1]
    top = ''
    cols = []
    for i, (a, b) in enumerate(zip(h0, h1)):
        top = a if a and not a.startswith('Unnamed') else top
        if i > 0 and top == 'Central pulse pressure during PWA' and b in ('12678-2.0', '12678-2.1'):
            cols.append(i)
    out = {}
    for r in rd[2:]:
        if not r or not r[0]:
            continue
        vals = [float(r[i]) for i in cols if i < len(r) and r[i] not in ('', 'nan', 'NaN')]
        pp = float(np.mean(vals)) if vals else float('nan')
        out[str(r[0])] = float('nan') if pp < 10 else pp
    return out
