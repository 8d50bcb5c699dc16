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
