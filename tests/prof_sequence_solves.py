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
for scene in ("full", "full_relief"):
    for reuse in ("0", "1"):
        env = dict(os.environ, PVIO_SEQ_IMAGE="hip", PVIO_HIP_TIMING="1", PVIO_HIP_REUSE_CANDIDATES=reuse)
        r = subprocess.run([sys.executable, os.path.join(HERE, "chain_run.py"), LIB, "/tmp/prof_seq_%s_%s" % (scene, reuse), "60", "6", "3", "25.0", scene],
                           capture_output=True, text=True, timeout=900, env=env)
        rows = [tuple(float(x) for x in m.groups()) for m in pat.finditer(r.stderr)]
        if not rows:
            print(scene, "reuse", reuse, "no solves parsed; rc", r.returncode, r.stderr[-300:])
            continue
        dev = [x[5] for x in rows]
        call = [x[4] for x in rows]
        # (the first call of a process pays the module load and the first allocations -- 2.4 ms: the C-ABI figure is the median, not the mean)
        print("%-12s reuse_identical_candidates=%s: %d keyframe solves; device us per solve: %s; mean %.0f us (C-ABI call median %.0f us); iterations %s" % (
            scene, reuse, len(rows), " ".join("%.0f" % d for d in dev), sum(dev) / len(dev), sorted(call)[len(call) // 2], " ".join("%d" % x[6] for x in rows)))
