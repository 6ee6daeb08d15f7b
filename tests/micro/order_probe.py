"""Same-box probe of library VARIANTS (tests/micro/build_variant.py): every library given on the command line solves the same windows
and is held against the CPU oracle iteration by iteration; prints, per library and window, whether the trace matches, the first
iteration whose states differ by more than 1e-6 and the worst difference, then a short resident-solve timing of the ones that pass.
Each library runs in its own process with a timeout (a wrong kernel may hang).
usage: python tests/micro/order_probe.py lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "vision_10x200": dict(n_frames=10, n_landmarks=200),
    "vision_4x30": dict(n_frames=4, n_landmarks=30),
    "vision_6x40": dict(n_frames=6, n_landmarks=40, visibility=3),
    "vision_10x1000": dict(n_frames=10, n_landmarks=1000),
    "vision_16x300": dict(n_frames=16, n_landmarks=300, visibility=8),
    "vio_3x30": dict(n_frames=3, n_landmarks=30, use_inertial=True),
    "vio_4x30": dict(n_frames=4, n_landmarks=30, use_inertial=True),
    "vio_6x40": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
    "vio_10x1000": dict(n_frames=10, n_landmarks=1000, use_inertial=True),
    "vio_11x80": dict(n_frames=11, n_landmarks=80, use_inertial=True, visibility=6),
}


def child(path):
    import time

    import numpy as np

    import ba_compare
    from oracle import oracle_py as O
    from pvio_amd import BAState, BASummary, capi
    from pvio_amd.solver import HipContext
    O.build()
    ctx = HipContext(lib=capi.load(path), device=0)
    out = {}
    for name, kw in CASES.items():
        pb = ba_compare.make(O, **kw)
        st0, sm0 = BAState(pb), BASummary(pb)
        O.solve(pb, st0, sm0)
        st1, sm1 = ctx.solve(pb)
        n = min(sm0.trace_len, sm1.trace_len)
        diffs = [float(np.nanmax(np.abs(sm1.trace_states[k] - sm0.trace_states[k]))) if np.isfinite(sm1.trace_states[k]).all() else float("inf") for k in range(n)]
        first_bad = next((k for k, d in enumerate(diffs) if not d <= 1e-6), -1)
        out[name] = dict(iters=(sm0.num_iterations, sm1.num_iterations), term=(sm0.termination, sm1.termination), first_bad=first_bad,
                         worst=max(diffs) if diffs else None, diffs=["%.1e" % d for d in diffs])
    ok = all(v["first_bad"] < 0 and v["iters"][0] == v["iters"][1] for v in out.values())
    if ok:
        pb = ba_compare.make(O, **CASES["vio_10x1000"])
        ctx.upload(pb)
        for _ in range(20):
            ctx.solve_resident(BASummary(pb, trace=False))
        rates = []
        for _ in range(5):
            sm = BASummary(pb, trace=False)
            t0, its = time.perf_counter(), 0
            for _ in range(200):
                ctx.solve_resident(sm)
                its += sm.num_iterations
            rates.append(its / (time.perf_counter() - t0))
        out["rate_it_per_s"] = sorted(rates)[2]
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    for p in sys.argv[1:]:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", p], capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("%-32s CRASH rc=%d %s" % (os.path.basename(p), r.returncode, r.stderr[-300:]))
                continue
            res = json.loads(line[0][7:])
            rate = res.pop("rate_it_per_s", None)
            print("%-32s %s" % (os.path.basename(p), "PASS  %.0f it/s" % rate if rate else "FAIL"))
            for name, v in res.items():
                if v["first_bad"] >= 0 or v["iters"][0] != v["iters"][1]:
                    print("    %-14s iterations %s termination %s first bad iteration %d worst %.2e  per-iteration: %s" % (
                        name, v["iters"], v["term"], v["first_bad"], v["worst"], " ".join(v["diffs"])))
        except subprocess.TimeoutExpired:
            print("%-32s TIMEOUT" % os.path.basename(p))
