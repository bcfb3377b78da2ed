import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_switches_back_to_defaults(request):
    """The library's tuning switches (dc_set_option) are process-global.  Whatever a GPU test leaves behind is put back after it, from the
    library's own table of defaults (dc_reset_options), so that no test runs on another test's switches."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    lib_mod = sys.modules.get("mlperf_deepcam_amd.lib")
    if lib_mod is not None and getattr(lib_mod, "_lib", None) is not None:
        lib_mod.call("dc_reset_options")
