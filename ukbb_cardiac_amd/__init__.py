"""ukbb_cardiac_amd -- MI355X-native FCN / U-Net segmentation inference path.

Drop-in for the reference's ``common/deploy_network.py`` /
``common/deploy_network_ao.py`` hot path: NIfTI in -> label map out, with the
network evaluated by hand-written gfx950 HIP kernels behind the C ABI declared
in ``include/ukbb_fcn.h``.  There is no CPU fallback: importing the compute
entry points without the built HIP library raises.
"""
__version__ = '0.1.0'
