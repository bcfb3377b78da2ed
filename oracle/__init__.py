"""CPU oracle for the DeepCAM train step -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the reference's
hot path (azrael417/mlperf-deepcam, src/deepCam).  It exists so that tests,
``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs (``cpu_baseline``; the opt-in
``torch_rocm_baseline``, the same step through PyTorch's operators on the GPU) have
something to check the HIP path against and to time beside it.  Nothing under ``mlperf-deepcam_amd/``
may import it: the product path must fail loudly when the HIP library is
missing, never fall back to this code.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
against golden vectors produced by importing the reference itself in the build
container (tests/golden/make_golden.py, which is the only file that touches
/root/reference).  Exceptions, marked "parity unpinned" where they are defined:
LAMB and the warm-up schedule (apex / pytorch-gradual-warmup-lr are not vendored
in the reference and not installable here).
"""
