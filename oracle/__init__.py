"""CPU oracle for the FCN / U-Net segmentation inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ukbb_cardiac_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and only as the checker.

PARITY UNPINNED against TensorFlow: the reference's arithmetic lives in
TensorFlow 1.x (un-vendored, version not pinned, ``README.md:31`` of the
reference) which is not installable here, and the reference ships no tests,
golden vectors, weights or images.  What pins this oracle instead:

* the pure-numpy reference helpers (``linear_1d``, ``linear_2d``,
  ``rescale_intensity``, ``normalise_intensity``, ``np_categorical_dice``) are
  executed *from the reference's own source* in the build container and their
  outputs are committed under ``tests/golden/`` (``tests/golden/make_golden.py``);
* the TF op semantics (SAME padding, conv2d_transpose crop, BN inference) are
  restated twice independently (numpy here, torch-CPU in ``tests/``) and
  cross-checked, plus hand-worked 1-D known answers from SURVEY.md Appendix B.
"""
