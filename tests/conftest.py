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


@pytest.fixture(autouse=True)
def _engines_of_finished_tests_are_freed(request):
    """An engine at the benchmark shape holds 20 - 30 GB of activations and sits in reference cycles (its launch closures point back at it), so
    it lives until the cycle collector runs.  Left to the collector's own schedule the model tests have piled up 285 of the card's 288 GB and the
    next engine failed to allocate; collect after every GPU test and hand the freed blocks back when more than a third of the card is cached."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    torch = sys.modules.get("torch")
    if torch is None or not torch.cuda.is_initialized():
        return
    import gc
    gc.collect()
    if torch.cuda.memory_reserved() > 96 * 2 ** 30:
        torch.cuda.empty_cache()
