"""Host logic of the phase-bracketed profiling driver (reference: profile_hdf5_ddp.py:77-94): which (phase, step) pairs open the
profiler bracket, and that the roctx calls pair up.  No GPU: the CUDA synchronisation inside the bracket is stubbed."""
import pytest
import torch

from mlperf_deepcam_amd import profile as prof


class _FakeRoctx:
    def __init__(self):
        self.calls = []

    def push(self, name): self.calls.append(("push", name))
    def pop(self): self.calls.append(("pop",))
    def pause(self): self.calls.append(("pause",))
    def resume(self): self.calls.append(("resume",))


def test_profile_bracket_follows_the_reference_rule(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    r = _FakeRoctx()
    warm = 2
    for step in range(4):
        for flag in ("Forward", "Backward", "Optimizer"):
            with prof.Profile(r, "Backward", flag, step, warm) as p:
                assert p.active == (flag == "Backward" and step >= warm)
    # two profiled steps, one bracket each: resume, push(name), ..., pop, pause
    assert r.calls == [("resume",), ("push", "Backward"), ("pop",), ("pause",)] * 2


def test_profile_parser_mirrors_the_reference_flags():
    a = prof.build_parser().parse_args([])
    assert (a.num_warmup_steps, a.num_profile_steps, a.profile) == (5, 1, "Forward")          # profile_hdf5_ddp.py:270-272
    assert a.optimizer == "Adam" and a.start_lr == 1e-3 and a.adam_eps == 1e-8 and a.weight_decay == 1e-6 and a.amp_opt_level == "O0"
    with pytest.raises(SystemExit):
        prof.build_parser().parse_args(["--profile", "Everything"])


def test_roctx_wrapper_is_a_noop_without_the_library(monkeypatch):
    import ctypes

    def no_lib(name, *a, **k):
        raise OSError("not found")

    monkeypatch.setattr(ctypes, "CDLL", no_lib)
    r = prof.Roctx()
    assert r.lib is None
    r.push("x"); r.pop(); r.pause(); r.resume()          # must not raise
