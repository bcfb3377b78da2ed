"""Static checks of the gfx950 code the compiler produces for the kernels whose speed depends on it (no GPU needed: hipcc cross-compiles).

The persistent 256 x 256 implicit-GEMM kernel (igemm256p.hip) runs its K loops at the register limit.  A register spill inside a K loop is a
scratch load followed by `s_waitcnt vmcnt(0)`, i.e. a wait for every LDS-DMA in flight, once per K step; whether the compiler spills there
changes with small edits (the first builds of that kernel held 3 - 27 scratch instructions per step).  The results stay correct either way, so
only this test notices."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mlperf-deepcam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _device_asm(src, tmp_path):
    out = os.path.join(str(tmp_path), os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", out], check=True, capture_output=True, timeout=600)
    return open(out).read()


def _kernels(asm, name):
    """{mangled name: body lines} of every kernel whose mangled name contains `name`."""
    out = {}
    for m in re.finditer(r"^(_Z\w*%s\w*):" % name, asm, re.M):
        end = asm.index("s_endpgm", m.end())
        out[m.group(1)] = asm[m.end():end].split("\n")
    return out


def _loops(lines):
    """(begin, end) line ranges of backward branches."""
    labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for n, l in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), n) < n:
            yield labels[m.group(1)], n


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_persistent_gemm_k_loops_hold_no_spills(tmp_path):
    kernels = _kernels(_device_asm("igemm256p.hip", tmp_path), "igemm256p_kernel")
    assert len(kernels) == 2, sorted(kernels)
    for name, lines in kernels.items():
        mfma_loops = [(a, b) for a, b in _loops(lines) if any("v_mfma" in x for x in lines[a:b])]
        assert mfma_loops, name
        for a, b in mfma_loops:
            body = lines[a:b]
            assert not any("scratch_" in x for x in body), f"{name}: scratch access inside the K loop at asm lines {a}-{b}"
            assert not any("v_accvgpr" in x for x in body), f"{name}: accumulator shuffling inside the K loop at asm lines {a}-{b}"
        # outside the loops: the state handed from the first instantiation of the tile loop to the second, once per launch
        assert sum("scratch_" in x for x in lines) <= 32, name


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,name,count", [("igemm384.hip", "pw384_kernel", 3), ("igemm256.hip", "igemm256_kernel", 1),
                                            ("wgrad384.hip", "wgrad384_kernel", 3),
                                            ("igemm192.hip", "pw192_kernel", 1), ("igemm224.hip", "pw224_kernel", 2)])
def test_gemm_k_loops_hold_no_spills(tmp_path, src, name, count):
    """The other MFMA kernels that run at the register limit.  (This scan is what found the spills of the 128-byte-row mode of
    pw384_kernel: six scratch reloads, each behind a vmcnt(0), per K step -- the reason that mode first measured slower than 64-byte rows.)"""
    kernels = _kernels(_device_asm(src, tmp_path), name)
    assert len(kernels) == count, sorted(kernels)
    for kname, lines in kernels.items():
        mfma_loops = [(a, b) for a, b in _loops(lines) if any("v_mfma" in x for x in lines[a:b])]
        assert mfma_loops, kname
        for a, b in mfma_loops:
            assert not any("scratch_" in x for x in lines[a:b]), f"{kname}: scratch access inside the K loop at asm lines {a}-{b}"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_pipelined_depthwise_tile_loops_hold_only_counted_memory_operations(tmp_path):
    """dwp_kernel (dwpipe.hip) decides "tile i has landed" from hand-counted `s_waitcnt vmcnt(N)` immediates: (NS-1) x stores + (NS-2) x LDS-DMAs.
    That count is exact only while the tile loop holds no OTHER vector-memory operation: a register spilled inside the loop (scratch store /
    reload) or a VGPR-destination global load would shift the in-order count, the stencil would read LDS before its DMA has arrived, and dx,
    the BatchNorm sums and the weight-gradient rows would be silently wrong (the DIL = 2 statistics + weight-gradient variant sits
    at 256 registers with one spill -- outside the loop).  Checked for every instantiation: inside every loop that issues LDS-DMAs there is no
    scratch access and no load with a register destination."""
    kernels = _kernels(_device_asm("dwpipe.hip", tmp_path), "dwp_kernel")
    assert len(kernels) >= 8, sorted(kernels)
    for name, lines in kernels.items():
        # the tile loops proper issue LDS-DMAs AND meet at a barrier per tile (the prologue, which fills the ring and loads the per-block
        # constants behind a vmcnt(0) of its own, is laid out with backward branches too but holds no barrier); innermost ones only
        dma_loops = [(a, b) for a, b in _loops(lines) if any("global_load_lds" in x or ("buffer_load" in x and " lds" in x) for x in lines[a:b])
                     and any("s_barrier" in x for x in lines[a:b])]
        assert dma_loops, name
        dma_loops = [(a, b) for a, b in dma_loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in dma_loops)]
        for a, b in dma_loops:
            body = lines[a:b]
            bad = [x.strip() for x in body if "scratch_" in x or re.search(r"\b(global|buffer|flat)_load_(?!lds)", x) and " lds" not in x]
            assert not bad, f"{name}: uncounted vector-memory operations inside the tile loop (asm lines {a}-{b}): {bad[:4]}"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,name,count,max_vgprs", [("pwbwd.hip", "pw_bn_bwd_kernel", 2, 256), ("pwbwd.hip", "pw_bn_bwd256_kernel", 2, 256),
                                                      ("sepfwd.hip", "sepconv_fwd_kernel", 2, 256)])
def test_one_pass_kernels_of_the_entry_flow_keep_their_occupancy(tmp_path, src, name, count, max_vgprs):
    """The one-pass kernels of round 5 (pwbwd.hip, sepfwd.hip) are sized for TWO workgroups (or one 512-thread workgroup) per CU: 256 registers
    per lane.  They hold their accumulators, the resident weight fragments and a stage of prefetched operands at 232 - 252 registers; a spill
    would put scratch traffic into an HBM-bound loop and more registers would halve the workgroups per CU -- both silently."""
    asm = _device_asm(src, tmp_path)
    kernels = _kernels(asm, name)
    assert len(kernels) == count, sorted(kernels)
    for kname, lines in kernels.items():
        assert not any("scratch_" in x for x in lines), f"{kname}: scratch access"
        m = re.search(r"\.amdhsa_kernel %s\b.*?\.amdhsa_next_free_vgpr (\d+)" % re.escape(kname), asm, re.S)
        assert m, kname
        assert int(m.group(1)) <= max_vgprs, f"{kname}: {m.group(1)} registers"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_forward_depthwise_tile_kernel_keeps_three_workgroups_per_cu(tmp_path):
    """dwt_kernel's 8 x 8 pixel halo tile of 256 channels is 52 KiB of dynamic LDS: three workgroups on a CU's 160 KiB, with 4 KiB to spare.
    The in-kernel BatchNorm finalize needs 18 KiB of static LDS (2 KiB when this was found) and is therefore an instantiation of its own (FIN): in the
    plain forward kernel those 2 KiB took the third workgroup off every CU and the 728-channel launches went from 28.6 to 32.1 us -- silently."""
    asm = _device_asm("dwtile.hip", tmp_path)
    fwd = {k: v for k, v in _kernels(asm, "dwt_kernel").items() if "Li1ELb0ELi32ELb0E" in k and "DF16b" in k}
    assert len(fwd) == 3, sorted(fwd)                       # <bf16, DIL 1, forward, CG 32, no weight gradient> x FIN {0, 1: row slab, 2: sum row}
    for kname in fwd:
        m = re.search(r"\.amdhsa_kernel %s\b.*?\.amdhsa_group_segment_fixed_size (\d+)" % re.escape(kname), asm, re.S)
        assert m, kname
        static_lds = int(m.group(1))
        if kname.split("Li1ELb0ELi32ELb0E")[1].startswith("Li1"):
            assert static_lds == 2048 + 16384, f"{kname}: {static_lds}"          # coefficients + the four row sequences' partial sums
        else:
            # (FIN = 2, the finalize over a sum row, keeps its coefficients in the 2 KiB the halo's last LDS-DMA instruction leaves unused)
            assert static_lds == 0, f"{kname}: {static_lds} bytes of static LDS beside the halo tile"
            assert 3 * (52 * 1024 + static_lds) <= 160 * 1024


def _most_loads_outstanding(lines):
    """Most global_load_dwordx4 results (a channel quad of a slab row each) outstanding at once: a load adds one, `s_waitcnt vmcnt(N)` leaves at
    most N (vector-memory results return in order)."""
    best = out = 0
    for x in lines:
        if re.search(r"\bglobal_load_dwordx4\s", x):
            out += 1
            best = max(best, out)
        else:
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", x)
            if m:
                out = min(out, int(m.group(1)))
    return best


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,name,select", [("dwtile.hip", "dwt_kernel", "Li1ELb0ELi32ELb0ELi1E"), ("bn.hip", "bn_apply_rows_kernel", "Li32ELb1E"),
                                             ("bn.hip", "bn_bwd_apply_kernel", "Li32E")])
def test_in_kernel_batchnorm_finalize_keeps_its_slab_loads_in_flight(tmp_path, src, name, select):
    """bn_fin.h: slab_quad_sum2 asks for eight rows of both sums (sixteen 16-byte loads) before it adds the first.  Written as a plain loop the
    compiler issued ONE load, waited and added (in the FIN instantiation of dwt_kernel; in the shared instantiation before it, the same source
    gave 32 in flight): a 54-row slab then costs 108 trips beyond L2 per workgroup and local batch 2 loses 0.7 ms per step -- with correct
    results.  (Serialized, the most outstanding would be the main loops' own four to twelve.)"""
    asm = _device_asm(src, tmp_path)
    kernels = {k: v for k, v in _kernels(asm, name).items() if select in k and "DF16b" in k}
    assert len(kernels) == 1, sorted(kernels)
    for kname, lines in kernels.items():
        assert _most_loads_outstanding(lines) >= 12, f"{kname}: at most {_most_loads_outstanding(lines)} slab loads in flight"


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,name", [("igemm224.hip", "pw224_kernel"), ("igemm384.hip", "pw384_kernel")])
def test_no_vector_register_is_written_between_the_last_look_ahead_read_and_its_wait(tmp_path, src, name):
    """The hand-scheduled pointwise kernels request the NEXT step's fragments (inline-assembly ds_read) in the last block of every K step,
    also of the last one, where nothing consumes them: for the compiler those destination registers are free from there on, while the reads
    are still on their way.  Safe only as long as the `s_waitcnt vmcnt(0) lgkmcnt(0)` behind the loop comes before any instruction that
    writes a vector register: this checks the path from the K loop's exit to that wait in the compiled kernel."""
    for kname, lines in _kernels(_device_asm(src, tmp_path), name).items():
        loops = [(a, b) for a, b in _loops(lines) if any("v_mfma" in x for x in lines[a:b])]
        assert loops, kname
        end = max(b for _, b in loops)
        # the first full wait behind the outermost MFMA loop
        wait = next(n for n in range(end, len(lines)) if re.search(r"s_waitcnt\s+vmcnt\(0\)\s+lgkmcnt\(0\)", lines[n]))
        between = [l.strip() for l in lines[end + 1:wait] if l.strip() and not l.strip().startswith((";", ".", "//"))]
        bad = [l for l in between if not re.match(r"s_", l)]
        assert not bad, f"{kname}: vector / memory instructions between the K loop and its final wait: {bad[:4]}"
