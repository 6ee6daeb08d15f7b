"""What the keyframe solves of a SEQUENCE cost on the device, with and without pvio_hip_opts::reuse_identical_candidates (DESIGN.md sections 2c, 4): the
reference's own pvio::PVIO (oracle/_ref/libpvio_dropin.so: reference control plane, the product's HipImage + BundleAdjustor + PnP below it) over the rendered
60-frame sequences, PVIO_HIP_TIMING=1 making the adapter print every solve.  usage (GPU box): python tests/prof_sequence_solves.py"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "oracle", "_ref", "libpvio_dropin.so")
pat = re.compile(r"\[pvio-hip\] solve: (\d+) frames, (\d+) landmarks, (\d+) factors: flatten ([\d.]+) us, upload\+solve\+download ([\d.]+) us \(device ([\d.]+) us, (\d+) iterations\)")
pat_host = re.compile(r"\[pvio-hip\] solve host: reset \+ (\d+) slots enqueued after ([\d.]+) us \(([a-z ]+)\), stream drained after ([\d.]+) us")
pat_abi = re.compile(r"\[pvio-hip\] ba_solve: staging \+ upload enqueue ([\d.]+) us, iterations \+ read-back ([\d.]+) us")
med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")
# (scene, frames, window, gap): the 60-frame sequences of round 4 and the 360-frame back-and-forth sequences of tests/test_dropin_sequence.py (31 keyframe solves)
RUNS = [("full", 60, 6, 3), ("full_relief", 60, 6, 3), ("full_relief_sweep", 360, 8, 3), ("full_sweep", 360, 8, 3)]
if len(sys.argv) > 1:
    RUNS = [r for r in RUNS if r[0] in sys.argv[1:]]
for scene, n_frames, window, gap in RUNS:
    for reuse in ("0", "1"):
        env = dict(os.environ, PVIO_SEQ_IMAGE="hip", PVIO_HIP_TIMING="1", PVIO_HIP_REUSE_CANDIDATES=reuse)
        r = subprocess.run([sys.executable, os.path.join(HERE, "chain_run.py"), LIB, "/tmp/prof_seq_%s_%s" % (scene, reuse), str(n_frames), str(window), str(gap), "25.0", scene],
                           capture_output=True, text=True, timeout=1500, env=env)
        rows = [tuple(float(x) for x in m.groups()) for m in pat.finditer(r.stderr)]
        if not rows:
            print(scene, "reuse", reuse, "no solves parsed; rc", r.returncode, r.stderr[-300:])
            continue
        dev = [x[5] for x in rows]
        call = [x[4] for x in rows]
        flat = [x[3] for x in rows]
        host = [(float(m.group(2)), m.group(3), float(m.group(4))) for m in pat_host.finditer(r.stderr)]
        abi = [(float(m.group(1)), float(m.group(2))) for m in pat_abi.finditer(r.stderr)]
        # (the first calls of a process pay the module load, the first allocations and the growth of the window: medians, and the steady state = from the third solve on)
        print("%-18s reuse_identical_candidates=%s: %d keyframe solves of %d frames; device us per solve: %s%s; mean %.0f us (C-ABI call median %.0f us); iterations %s" % (
            scene, reuse, len(rows), n_frames, " ".join("%.0f" % d for d in dev[:8]), " ..." if len(dev) > 8 else "", sum(dev) / len(dev), med(call), " ".join("%d" % x[6] for x in rows[:8])))
        st = slice(2, None)
        if len(rows) > 4 and len(host) == len(rows) and len(abi) == len(rows):
            over = [c - d for c, d in zip(call[st], dev[st])]
            print("    steady state (solves 3..%d): device median %.0f us, C-ABI call median %.0f us, call - device median %.0f us (max %.0f); of the call: staging + upload enqueue %.0f us, "
                  "reset + slots enqueued after %.0f us (%s), stream drained after %.0f us; adapter flatten (outside the C ABI) %.0f us" % (
                      len(rows), med(dev[st]), med(call[st]), med(over), max(over), med([a[0] for a in abi[st]]), med([h[0] for h in host[st]]),
                      "/".join(sorted(set(h[1] for h in host[st]))), med([h[2] for h in host[st]]), med(flat[st])))
