"""r04 kernels of the aortic U-Net in UKBB_PREC_BF16, each against the kernel(s) it replaces on the same inputs (through the C ABI):
the weight-stationary conv / transposed-conv tilings (kernels_ws.hip), the fused tail (kernels_tail.hip) and the fused stem
(kernels_stem.hip).  The comparisons are those of tools/check_ws.py, check_tail.py and check_stem.py, run as the tools themselves.
(Regression guards: each compares a kernel with the kernel(s) it replaced.  The parity statement of these kernels against an independent computation is
tests/test_bf16_layers_gpu.py -- every stored map of the plan against numpy float64 on the bf16-rounded operands, half a bf16 ulp.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _tool(name, *args, **extra_env):
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    for k in ('UKBB_CONV_CFG', 'UKBB_NO_FUSE_TAIL', 'UKBB_NO_FUSE_STEM'):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', name)] + [str(a) for a in args], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == 'OK', r.stdout[-3000:]
    return r.stdout


def test_weight_stationary_tilings_match_the_tile_per_workgroup_kernels():
    """Every ws tiling on every layer type it serves, at a size with ragged tiles (64 x 96: 48-, 24-, 12-pixel rows) and at 256 x 256:
    the layer's own output differs from the r03 kernel's by at most a bf16 ulp on a handful of elements (same products, other fp32
    summation order); two sources (skip concat), 16-channel layers on the zero-padded block, all three transposed-conv pairings."""
    out = _tool('check_ws.py', 2, 64, 96, 'conv1_1:400,401', 'conv2_1:400,401,402,403', 'conv3_1:400,401', 'up1_0:400,401', 'up2_0:400,401',
                'up0_0:400,401', 'up0_t:410,411,412', 'up1_t:410,411,412', 'up2_t:410,411,412', 'up0_1+logits:404,405,406')
    assert 'NOT TAKEN' not in out
    _tool('check_ws.py', 3, 256, 256, 'conv1_1:401', 'conv2_1:402', 'conv3_1:400', 'up1_0:401', 'up2_0:400', 'up0_t:410', 'up1_t:411', 'up2_t:411')
    # the ring-streamed form (weights of one chunk at a time through a shared two-stage ring, one barrier per chunk): K = 2304 layers and,
    # for the comparison with the resident-filter form, layers both can run (there the two must agree bit for bit: same summation order)
    out = _tool('check_ws.py', 2, 64, 96, 'up3_0:420,421,422,423', 'conv4_1:421,423', 'conv3_1:421', 'up2_0:420,422', 'conv2_1:420')
    assert 'NOT TAKEN' not in out
    _tool('check_ws.py', 3, 256, 256, 'up3_0:420,422', 'conv4_1:422')
    # under the XCD-contiguous tile order (maps >= 128 x 128, forced here on every map) the ring-streamed form took its
    # workgroup's round count from the wrong worker id and the four-wave tilings 420 / 421 dropped their last tiles whenever
    # ntiles % nworkers fell between the two ids.  Batches chosen so that the tile counts sweep the remainders.
    for n in (1, 2, 3, 5, 7):
        out = _tool('check_ws.py', n, 64, 96, 'up3_0:420,421,422,423', 'conv3_1:421', 'up2_0:420', 'conv2_1:420,400', UKBB_WS_XCD_LOCAL='1')
        assert 'NOT TAKEN' not in out
    _tool('check_ws.py', 3, 256, 256, 'conv1_1:421,401', 'up1_0:421', 'conv2_1:420', UKBB_WS_XCD_LOCAL='1')   # 420 / 422 own 64 output channels per workgroup: not for the 32-channel layers


def test_fused_tail_matches_the_unfused_plan():
    """up0_0 -> up0_1 -> logits -> argmax in one launch against the three-kernel plan: logits to ~1e-3 of their scale (a bf16 ulp of a few
    intermediate values), labels equal away from ties, prob / pred-only paths consistent; ragged tiles (W = 96, 80, 16) and borders."""
    _tool('check_tail.py', 2, 64, 96, 3, 256, 256, 1, 48, 80, 2, 16, 16, 1, 32, 272)


def test_fused_stem_matches_the_r03_form():
    """conv0_0 + conv0_1 in one launch (image rounded to bf16) against conv0_0 in fp32 inside conv0_1's staging: the level-0 map's error
    against the fp32 path stays at the bf16 level, Dice against fp32 unchanged."""
    _tool('check_stem.py', 2, 64, 96, 3, 256, 256, 1, 48, 80, 2, 16, 16)


def test_winograd_f2x4_matches_f2x2_and_its_two_region_shapes_agree_bit_for_bit():
    """r04: kernels_wino24.hip (tilings 304 / 305) on the fp32 FCN and U-Net layers it serves, layer by layer against the F(2x2,3x3) kernel
    (<= 2e-6 of the map's scale: the level of a direct fp32 sum) and 304 against 305 (identical bits: the small-batch plan takes 305
    where the large one takes 304, and test_small_batch_plan_is_arithmetic_neutral relies on that)."""
    out = _tool('check_wino24.py')
    assert 'NOT TAKEN' not in out and 'not available' not in out


def test_tail_on_column_strips_gives_the_bits_of_the_row_major_walk():
    """r06, the default walk of the fused tail (kernels_tail.hip STRIP; UKBB_TAIL_STRIPS=0 = the row-major walk of r04 / r05: tiles walked down column strips, the four halo rows two vertically adjacent
    tiles share moved inside LDS instead of re-fetched): every output identical bits to the default walk for whole strips, segment lengths 1, 2, 3, 5
    and ragged / tiny maps, pred-only and full-output kernels against the row-major walk (tools/check_tail_strips.py; the first build failed exactly this test: hipcc sank the
    copy's LDS reads behind stores that alias them through other lanes)."""
    _tool('check_tail_strips.py', 2, 64, 96, 3, 256, 256, 1, 48, 80, 2, 16, 16, 1, 32, 272, 5, 112, 48)
