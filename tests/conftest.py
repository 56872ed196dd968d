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


# Kernel routing is per call since round 6 (bd_conv_desc.route: the library keeps no state); what a test sets is the PYTHON shim's
# current route (basedet_amd.ops.set_route), stamped into every descriptor it passes on.  A test that sets one and then fails would leave
# it set for every later test of the run: every GPU test ends with the shim's words back at "library default".
@pytest.fixture(autouse=True)
def _restore_route(request):
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from basedet_amd import ops
    ops.reset_route()
