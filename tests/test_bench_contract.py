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


def test_traffic_is_null_unless_the_counters_match_the_sources(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import summarize_pmc
    committed = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm.json"))
    assert committed, "no PMC summary committed"
    d = json.load(open(os.path.join(ROOT, "profiles", committed[-1])))
    # the committed summary of this round carries the fingerprint of the sources in the tree
    assert d["source_sha256"] == summarize_pmc.source_sha256(), "profiles/*_pmc_hbm.json was collected on other kernel sources: re-run profiles/collect.sh"
    assert bench.pmc_traffic("k_linearize") == d["kernels"]["k_linearize"]["hbm_bytes_per_launch"]
    # a summary of other sources is ignored
    fake_root = tmp_path / "repo"
    (fake_root / "profiles").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "profiles", "summarize_pmc.py"), fake_root / "profiles" / "summarize_pmc.py")
    d2 = dict(d, source_sha256="0" * 64)
    (fake_root / "profiles" / "r9_pmc_hbm.json").write_text(json.dumps(d2))
    (fake_root / "pvio_amd" / "csrc").mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    sys.modules.pop("summarize_pmc", None)
    assert bench.pmc_traffic("k_linearize") is None
    sys.modules.pop("summarize_pmc", None)
