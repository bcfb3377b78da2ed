"""The launch-list executor's host side (no GPU): the generated thunk table is in step with the binding table, every recordable entry
point is exported, and the argument-word marshalling of lib.Program matches include/deepcam_hip.h's description."""
import ctypes as C
import importlib.util
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlperf-deepcam_amd", "csrc")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_thunks", os.path.join(CSRC, "gen_thunks.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_thunk_table_is_generated_from_the_binding_table():
    assert _gen().generate() == open(os.path.join(CSRC, "program_thunks.inc")).read(), \
        "program_thunks.inc is stale: python mlperf-deepcam_amd/csrc/gen_thunks.py > mlperf-deepcam_amd/csrc/program_thunks.inc"


def test_every_int_entry_point_has_a_thunk():
    from mlperf_deepcam_amd import lib as L
    text = open(os.path.join(CSRC, "program_thunks.inc")).read()
    for name, (res, args) in L._SIGS.items():
        recordable = res is C.c_int and not name.startswith("dc_program_")
        assert (f'{{"{name}", &t_{name}, {len(args)},' in text) == recordable, name


def test_argument_words():
    from mlperf_deepcam_amd import lib as L
    assert L._word(L.I, 7) == 7 and L._word(L.L, -3) == -3 and L._word(L.SZ, 1 << 40) == 1 << 40
    assert L._word(L.F, 0.1) == struct.unpack("q", struct.pack("d", 0.1))[0]
    assert L._word(L.P, None) == 0 and L._word(L.P, C.c_void_p(0x1234)) == 0x1234 and L._word(L.P, 99) == 99
    d = L.ConvDesc(1, 3, 1, 1, 1, 0, 16, 32)
    assert L._word(L.CD, C.byref(d)) == C.addressof(d) == L._word(L.P, d)
    arr = (C.c_void_p * 3)(1, 2, 3)
    assert L._word(L.P, arr) == C.addressof(arr)
    with pytest.raises(L.DeepcamHipError):
        L._word(L.P, "a string")


def test_record_and_replay_without_a_gpu_and_only_the_recording_thread():
    """dc_program_* need no device for entry points that launch nothing: record dc_reset_options / dc_set_option from two threads, only the
    recording thread's calls land in the list; the replay runs them again (the option is back after a reset in between)."""
    import threading
    from mlperf_deepcam_amd import lib as L
    lib = L.load()
    prog = L.Program()
    with prog.recording():
        L.call("dc_reset_options")
        t = threading.Thread(target=lambda: L.call("dc_reset_options"))
        t.start(); t.join()
    assert prog.names() == ["dc_reset_options"]
    prog.run()
    with pytest.raises(L.DeepcamHipError, match="cannot record an argument of type bytes"):
        prog.append("dc_set_option", b"pw384", 1)      # a char* argument: host strings are not recordable words
    assert len(prog) == 1
