"""Static guards on the compiled ISA of kernels whose correctness depends on how the compiler lays out their control flow (no GPU needed:
hipcc cross-compiles gfx950 here).  Round 5: the first build of k_lk_track_units hung on the GPU because hipcc threaded the loop's two
`lane == 0` branches (hand-down at the end of a pass, queue fetch at the start of the next) across the back edge and sent lanes 1-63 round an
inner loop of their own, where `v_readfirstlane` ran without lane 0 (profiles/r5_ab_klt_units_hang.txt).  The fiber emulator runs the source's
semantics and cannot see this; the loop nest in the ISA can."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


_ISA_TEXT = {}  # source -> ISA text: a file is compiled once per run, whatever the number of kernels looked at


def _kernel_isa(tmp_path, source, mangled_prefix):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    if source not in _ISA_TEXT:
        out = str(tmp_path / "k.s")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                               os.path.join(ROOT, "pvio_amd", "csrc", source)], stderr=subprocess.DEVNULL)
        _ISA_TEXT[source] = open(out).read().splitlines(True)
    body, on = [], False
    for line in _ISA_TEXT[source]:
        if not on and line.startswith(mangled_prefix) and line.rstrip().split(";")[0].rstrip().endswith(":"):
            on = True
        if on:
            if line.startswith(".Lfunc_end"):
                break
            body.append(line)
    assert body, "kernel not found in the ISA"
    return body


def test_lk_unit_kernel_has_one_uniform_unit_loop(tmp_path):
    body = _kernel_isa(tmp_path, "klt.hip", "_ZN5pvklt16k_lk_track_unitsE")
    text = "".join(body)
    depths = [int(m.group(1)) for m in re.finditer(r"Loop Header: Depth=(\d+)", text)]
    # the unit loop, and inside it the spin on the predecessor's flag and the LK iterations -- nothing deeper, nothing beside the unit loop
    assert depths.count(1) == 1 and depths.count(2) == 2 and max(depths) == 2, depths
    head = text[text.index("Loop Header: Depth=1"):]
    head = head[:head.index("Inner Loop Header")]
    assert "; wave barrier" in head  # the convergent no-op that keeps the two lane-0 branches apart
    # the queue fetch is one LDS atomic under a lane mask whose exec is restored before the broadcast, and the exit test is scalar
    fetch = head[head.index("ds_add_rtn_u32"):]
    i_restore, i_bcast = fetch.index("s_or_b64 exec, exec"), fetch.index("v_readfirstlane_b32")
    assert i_restore < fetch.rindex("v_readfirstlane_b32", 0, fetch.index("s_cmp_ge_i32")) or i_restore < i_bcast
    assert re.search(r"s_cmp_ge_i32 s\d+, s\d+\n(\s+s_mov_b64[^\n]*\n)?\s+s_cbranch_scc1", fetch)
    assert "scratch_" not in text  # and nothing spills


def test_lk_track_kernel_is_loop_free_outside_its_iterations(tmp_path):
    """k_lk_track (a wave per track): four unrolled levels, each with its iteration loop and nothing else -- the form whose timing the
    stamps of tests/micro/klt_stamps.py describe"""
    body = _kernel_isa(tmp_path, "klt.hip", "_ZN5pvklt10k_lk_trackE")
    depths = [int(m.group(1)) for m in re.finditer(r"Loop Header: Depth=(\d+)", "".join(body))]
    assert depths == [1, 1, 1, 1], depths


def test_kernel_isa_is_the_gpu_verified_one():
    """tools/isa_ledger.py: every kernel's compiled body hashes to what tests/golden/isa_verified.json records as verified on an MI355X.  A
    kernel edit (or a compiler that decides otherwise) fails here until `pytest -m gpu` and smoke() have passed on a GPU with a library built
    from the edited tree and `python tools/isa_ledger.py --update "<that run>"` has recorded it -- the emulated tests cannot stand in for that."""
    import sys
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_ledger
    diff, why = isa_ledger.compare(verbose=True)
    if diff is None:
        pytest.skip(why)
    assert diff == [], "kernels whose ISA is not the GPU-verified one: %s" % ", ".join(diff)
