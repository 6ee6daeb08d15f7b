"""Timeline of ONE panel (panel 10 of 19, 10 x 1000 VIO window) of k_dense's look-ahead factorization: shader-clock ticks since the launch's first
instruction at the hand-over points of the factor wave (wave 0) and of update wave 1, one run per stamp site (PVIO_HIP_STAMP_SEL) so that the stamps do not add up.
Needs the stamp sites of profiles/r4_split_rows_experiment.patch (sites 20-27 of k_dense are not in the shipped kernel).
usage (GPU box): python tests/micro/prof_panel.py [lib.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pvio_amd import synth, BASummary, capi  # noqa: E402
from pvio_amd.solver import HipContext, preintegrate  # noqa: E402

SITES = [(13, "wave 0: panel 10 starts (before the wait for the published columns)"), (22, "wave 0: published columns seen"), (14, "wave 0: pivot loop done"),
         (15, "wave 0: block / L rows stored"), (16, "wave 0: signalled"),
         (23, "wave 1: panel 10 starts"), (20, "wave 1: published columns seen (split form)"), (21, "wave 1: factor wave's signal seen"),
         (24, "wave 1: own L row stored (split form)"), (25, "wave 1: all rows stored (split form)"), (26, "wave 1: next column published + signalled"), (27, "wave 1: panel 10 ends"),
         (5, "factorization done"), (7, "kernel ends")]


def main():
    lib = capi.load(sys.argv[1]) if len(sys.argv) > 1 else None
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=True, preintegrate=preintegrate)
    rows = []
    for sel, what in SITES:
        os.environ["PVIO_HIP_STAMP_SEL"] = str(sel)
        os.environ["PVIO_HIP_PROFILE_GRAPH"] = "1"
        ctx = HipContext(device=0, lib=lib) if lib is not None else HipContext(device=0)
        ctx.upload(pb)
        for _ in range(3):
            ctx.solve_resident(BASummary(pb, trace=False))
        vals = []
        for _ in range(3):
            ctx.profile_resident(BASummary(pb, trace=False))
            vals.append(ctx.last_phase_ticks["k_dense"][sel])
        ctx.close()
        rows.append((sel, what, sorted(vals)[1]))
    base = rows[0][2]
    for sel, what, t in rows:
        print("site %2d  %8d ticks  (%+7d from the factor wave's panel start)  %s" % (sel, t, t - base, what))


if __name__ == "__main__":
    main()
