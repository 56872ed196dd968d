import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The library's measurement knobs are process-global (include/basedet_hip.h: bd_conv_set_patch3x3, bd_conv_set_dense1x1,
# bd_wgrad_set_transpose_read, bd_focal_set_fast, bd_conv_fp8_set_patch).  A test that flips one and then fails would leave it flipped
# for every later test of the run: every GPU test ends with the defaults restored, whatever happened inside it.
_KNOB_DEFAULTS = (("bd_conv_set_patch3x3", (3,)), ("bd_conv_set_dense1x1", (1,)), ("bd_wgrad_set_transpose_read", (1,)), ("bd_focal_set_fast", (1,)),
                  ("bd_conv_fp8_set_patch", (1,)), ("bd_groupnorm_set_chunks", (0, 0)), ("bd_rpn_set_nms_per_level", (1,)))


@pytest.fixture(autouse=True)
def _restore_library_knobs(request):
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from basedet_amd import _lib
    lib = _lib.load()
    for name, value in _KNOB_DEFAULTS:
        assert getattr(lib, name)(*value) == 0, name
