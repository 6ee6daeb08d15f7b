"""A/B timing of builds of libpvio_hip.so on the SAME box (the pool varies by +-4 % from box to box): resident solves of the
10 x 1000 VIO window (or `frames landmarks` given after the paths), the libraries taking turns.

EVERY LIBRARY RUNS IN ITS OWN PROCESS (round 3).  Two builds of the library in one process share the dynamic symbols of their kernel
handles (weak template instantiations such as pvba::k_dense<true>): a build that is wrong when loaded alone passed every solve when
the product library had been loaded before it (profiles/r3_kdense_order_probe_*.txt), i.e. the second library was not running its own
kernels.  In-process A/Bs of round 2 that reported "no change" may have timed the same kernel twice.
A library may carry environment settings for its process: path@NAME=VALUE[@NAME2=VALUE2] (e.g. the same build with a code path off).
usage: python tests/prof_ab.py libA.so libB.so[@ENV=V] [libC.so ...] [frames landmarks]"""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(path, nf, nl, rounds):
    sys.path.insert(0, ROOT)
    from pvio_amd import synth, BASummary, capi
    from pvio_amd.solver import HipContext, preintegrate
    reps = 300 if nl <= 2000 else 12
    pb = synth.make_window(n_frames=nf, n_landmarks=nl, use_inertial=True, preintegrate=preintegrate)
    ctx = HipContext(lib=capi.load(path), device=0)
    ctx.upload(pb)
    for _ in range(20 if nl <= 2000 else 3):
        ctx.solve_resident(BASummary(pb, trace=False))
    out = []
    for _ in range(rounds):
        sm = BASummary(pb, trace=False)
        t0, its = time.perf_counter(), 0
        for _ in range(reps):
            ctx.solve_resident(sm)
            its += sm.num_iterations
        out.append(its / (time.perf_counter() - t0))
    print("RATES " + json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    args = sys.argv[1:]
    nf, nl = 10, 1000
    if len(args) >= 2 and args[-1].isdigit() and args[-2].isdigit():
        nf, nl = int(args[-2]), int(args[-1])
        args = args[:-2]
    res = {p: [] for p in args}
    for turn in range(3):  # the libraries take turns: box drift shows up as a trend over the turns, not as a difference
        for p in (args if turn % 2 == 0 else args[::-1]):
            parts = p.split("@")
            env = dict(os.environ)
            for kv in parts[1:]:
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", parts[0], str(nf), str(nl), "2"], capture_output=True, text=True, timeout=600, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("RATES ")]
            if line:
                res[p] += json.loads(line[0][6:])
            else:
                print("%s: child failed rc=%d %s" % (p, r.returncode, r.stderr[-200:]))
    for p in args:
        r = sorted(res[p])
        if r:
            print("%-44s median %.0f it/s  (min %.0f max %.0f, %d samples)" % (p, r[len(r) // 2], r[0], r[-1], len(r)))
