import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU: skip rather than fail at import of the device.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


# ---- parity record -------------------------------------------------------------------------
# The -m gpu parity tests report what they measured (max logits error, label disagreements and how many of those
# sit on a numerical tie) through the ``parity_log`` fixture; at session end the records are written to
# gpurun_out/parity_report.json on the GPU box, and the round's copy is committed as profiles/rNN_parity.json.
_PARITY = {}


@pytest.fixture
def parity_log(request):
    def log(**kw):
        _PARITY[request.node.name] = {k: (float(v) if isinstance(v, float) else v) for k, v in kw.items()}
    return log


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_report.json'), 'w') as f:
            json.dump({'note': 'written by tests/conftest.py from the -m gpu parity tests; oracle = this repo\'s CPU '
                               'restatement (parity vs TensorFlow itself is unpinned, DESIGN.md section 2)',
                       'tests': _PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass
