"""Host-side intensity pre-processing of the deploy scripts.

Mirrors ``rescale_intensity`` / ``normalise_intensity`` of the reference
(``common/image_utils.py:60-77``) including their dtype behaviour under the
numpy of this image (2.x), which is what the committed golden vectors
(``tests/golden/ref_numpy_helpers.npz``, produced by running the reference's own
function bodies) pin bit-for-bit.
"""
import numpy as np


def rescale_intensity(image, thres=(1.0, 99.0), numpy1_casting=False):
    """Clip to the [thres[0], thres[1]] percentiles of the WHOLE array and map
    to [0, 1].

    ``numpy1_casting``: the reference was written for numpy 1.x, whose value-based
    casting keeps ``float32_array - float64_scalar`` in float32 (the scalar is
    rounded to float32 first) and divides by the float64 difference of the two
    percentile scalars rounded to float32.  numpy >= 2 (NEP 50, this image, and
    what the committed goldens pin) does the same arithmetic in float64 and
    rounds once, which can differ by 1 ulp in the network input.  Set it to
    compare label maps against a run of the real TF / numpy-1.x deployment
    (INTEGRATION.md section 5); the result is then float32.

    Quirks kept on purpose: the clip is applied IN
    PLACE to the caller's array (the reference's ``image2 = image`` aliases it,
    so the ED/ES frames it later saves are the clipped intensities), and the
    percentiles are joint over all slices and frames.  The result is float64
    (float32 array minus a float64 percentile scalar); the deploy loop casts to
    float32 when it builds the network input.
    """
    val_l, val_h = np.percentile(image, thres)
    image[image < val_l] = val_l
    image[image > val_h] = val_h
    lo, hi = np.float64(val_l), np.float64(val_h)
    if numpy1_casting:
        return (image.astype(np.float32) - np.float32(lo)) / np.float32(hi - lo)
    return (image.astype(np.float32).astype(np.float64) - lo) / (hi - lo)


def normalise_intensity(image, thres_roi=10.0):
    """Z-score using mean / population std of the voxels at or above the
    ``thres_roi`` percentile (reference ``common/image_utils.py:60-67``)."""
    val_l = np.percentile(image, thres_roi)
    roi = image >= val_l
    mu, sigma = np.mean(image[roi]), np.std(image[roi])
    eps = 1e-6
    return (image - mu) / (sigma + eps)


def np_categorical_dice(pred, truth, k):
    """Dice overlap of label ``k`` (reference ``common/image_utils.py:171-175``);
    used for the bf16-vs-fp32 check of BASELINE config 5."""
    a = (pred == k).astype(np.float32)
    b = (truth == k).astype(np.float32)
    return 2 * np.sum(a * b) / (np.sum(a) + np.sum(b))
