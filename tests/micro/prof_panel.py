"""Timeline of panels 10 and 11 (of 19; 10 x 1000 VIO window) of k_dense's look-ahead factorization from ONE launch: shader-clock ticks since the launch's
first instruction at the hand-over points of the factor wave (wave 0) and of update wave 1.  Needs the stamp sites of the `timeline` recipe of
tests/micro/build_variant.py (every site stores in the same run: the kernel is ~10 % slower than the product's, the ORDER of events is what this shows).
usage (GPU box): python tests/micro/prof_panel.py tests/micro/variants/timeline.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pvio_amd import synth, BASummary, capi  # noqa: E402
from pvio_amd.solver import HipContext, preintegrate  # noqa: E402

SITES = {13: "wave 0: panel 10 starts (requests of the late patch out, about to wait for the published columns)", 22: "wave 0: panel 10: published columns seen",
         14: "wave 0: panel 10: pivot loop + forward substitution of the rows done", 15: "wave 0: panel 10: L rows stored", 16: "wave 0: panel 10: signalled",
         20: "wave 0: panel 11 starts", 24: "wave 0: panel 11: published columns seen", 25: "wave 0: panel 11: pivot loop done",
         23: "wave 1: panel 10 starts (about to wait for the factor wave)", 21: "wave 1: panel 10: factor wave's signal seen",
         26: "wave 1: panel 10: next tile column updated, published, signalled", 27: "wave 1: panel 10 ends (all columns updated)",
         5: "factorization done", 7: "kernel ends"}


def main():
    lib = capi.load(sys.argv[1])
    pb = synth.make_window(n_frames=10, n_landmarks=1000, use_inertial=True, preintegrate=preintegrate)
    os.environ["PVIO_HIP_STAMP_SEL"] = "-1"
    os.environ["PVIO_HIP_PROFILE_GRAPH"] = "1"
    ctx = HipContext(device=0, lib=lib)
    ctx.upload(pb)
    for _ in range(3):
        ctx.solve_resident(BASummary(pb, trace=False))
    for rep in range(3):
        ctx.profile_resident(BASummary(pb, trace=False))
        t = ctx.last_phase_ticks["k_dense"]
        base = t[13]
        print("run %d" % rep)
        for sel in sorted(SITES, key=lambda k: t[k]):
            print("  %8d ticks  (%+7d)  site %2d  %s" % (t[sel], t[sel] - base, sel, SITES[sel]))


if __name__ == "__main__":
    main()
