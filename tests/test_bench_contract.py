"""bench.py's host-side helpers (no GPU): workload names, the algorithmic-bytes figure of SURVEY section 8(d), and the rule that
`roofline.traffic` is only printed from a PMC summary that was collected on exactly the kernel sources in the tree."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pvio_amd import synth  # noqa: E402


def test_workload_names():
    assert bench.parse_workload("vio") == (10, 1000, True) and bench.parse_workload("vision") == (10, 1000, False)
    assert bench.parse_workload("30x50000_vio") == (30, 50000, True)
    assert bench.parse_workload("10X50000_vision") == (10, 50000, False)
    assert bench.parse_workload("8x200") == (8, 200, True)


def test_algorithmic_bytes_are_the_survey_formula():
    # 2 (20 F + 32 M) + 8 M + 8 (d N)^2 bytes per iteration; 612 000 at 10 x 1000 VIO (F = 9000 factors)
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=False)
    F, M, N = pb.n_obs, pb.n_landmarks, pb.n_frames
    assert synth.algorithmic_bytes_per_iteration(pb) == 2 * (20 * F + 32 * M) + 8 * M + 8 * (6 * N) ** 2
    assert 2 * (20 * 9000 + 32 * 1000) + 8 * 1000 + 8 * 150 ** 2 == 612000


def test_traffic_comes_from_counter_passes_of_the_run_or_is_null(monkeypatch):
    """`roofline.traffic` is measured inside the run (rocprofv3 child passes of bench.py itself) or null: nothing is read from a file
    committed earlier."""
    import argparse
    import inspect
    src = inspect.getsource(bench.live_pmc) + inspect.getsource(bench.pmc_traffic)
    assert "profiles/r" not in src and "_pmc_hbm" not in src and "glob" not in src
    args = argparse.Namespace(workload="vio", no_pmc=True)
    bench._PMC.update(done=False, kernels=None, note=None)
    assert bench.pmc_traffic("k_linearize", args) is None and bench._PMC["note"] == "rocprofv3 not run"
    bench._PMC.update(done=False, kernels=None, note=None)
    monkeypatch.setenv("PVIO_BENCH_NO_PMC", "1")
    assert bench.pmc_traffic("k_linearize", argparse.Namespace(workload="vio", no_pmc=False)) is None
    # a multi-GPU line never carries counters of one rank
    bench._PMC.update(done=True, kernels={"k_linearize": {"hbm_bytes_per_launch": 1.0}}, note="x")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.pmc_traffic("k_linearize", args) is None
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.pmc_traffic("k_linearize", args) == 1.0 and bench.pmc_traffic("k_nope", args) is None
    bench._PMC.update(done=False, kernels=None, note=None)


def test_dense_roofline_flops():
    # P^3 / 3 + 2 P^2 at P = 150: 1.17 Mflop per factoring launch; the FP64 peak quoted is the MI355X figure
    assert abs((150 ** 3 / 3.0 + 2.0 * 150 ** 2) - 1.17e6) < 1e4 and bench.FP64_PEAK_TFLOPS == 78.6


def test_bare_multi_gpu_invocation_becomes_a_one_rank_per_gpu_launch(monkeypatch):
    """`python bench.py --gpus 8` without a launcher ran on one GPU and printed "n_gpus": 8.  A bare --gpus N > 1 now re-executes
    itself under torch.distributed.run with N ranks (rendezvous on 127.0.0.1); a launcher's WORLD_SIZE that disagrees with the flag is an error;
    the printed n_gpus is the world size, not the flag."""
    import inspect
    import pytest
    calls = []

    class Stop(Exception):
        pass

    def fake_execv(exe, argv):
        calls.append((exe, argv))
        raise Stop()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(Stop):
        bench.main()
    exe, argv = calls[0]
    assert exe == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and argv[-7] == os.path.join(ROOT, "bench.py")
    # a launcher's world that disagrees with the flag: refused before anything touches a GPU
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE" in str(e.value)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE" in str(e.value)
    src = inspect.getsource(bench.main)
    assert '"n_gpus": world' in src and '"n_gpus": args.gpus' not in src
