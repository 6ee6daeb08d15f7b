"""the keyframe solves of VIO windows reject most of their trust-region steps (45 of 50 in the rendered sequence, 8 of 10 in the
small synthetic VIO windows).  Product, oracle and the reference's own sources (oracle/_ref, tests/test_dropin_map.py) agree on every one of those
rejections, so it is not a parity failure; this probe asks WHY.  Hypothesis: the reference's PreIntegrationErrorCost reads the LIVE
frame_i->motion.bg / ba (preintegration_error_cost.h:57-58) -- the user state, which Ceres updates only after an accepted step -- instead of its
parameter block, so the residual at a candidate point is evaluated with the OLD biases in the bias-correction term: the actual cost change does not
follow the model's and the step is rejected.  The oracle can switch that read off (ORACLE_NO_LIVE_BIAS=1: biases taken at the evaluation point).
usage: python tests/probe_rejections.py   (CPU only; prints one line per window and mode)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "vio_partial_6x40": dict(n_frames=6, n_landmarks=40, use_inertial=True, visibility=4),
    "vio_zero_bias_4x30": dict(n_frames=4, n_landmarks=30, use_inertial=True, bias_init="zero", perturb_scale=1.0),
    "metric_10x1000_vio": dict(n_frames=10, n_landmarks=1000, use_inertial=True),
    "vio_8x200": dict(n_frames=8, n_landmarks=200, use_inertial=True, visibility=5),
}


def child():
    from oracle import oracle_py as O
    from pvio_amd import BAState, BASummary, synth
    O.build()
    for name, kw in CASES.items():
        pb = synth.make_window(preintegrate=O.preintegrate, **kw)
        st, sm = BAState(pb), BASummary(pb)
        O.solve(pb, st, sm)
        tr = sm.trace()
        rej = sum(1 for t in tr[1:] if not t["step_is_successful"])
        print("%-22s %-28s iterations %2d rejected %2d  cost %.6f -> %.6f" % (name, "live bias read OFF" if os.environ.get("ORACLE_NO_LIVE_BIAS") else "as the reference (live read)",
                                                                              sm.num_iterations, rej, sm.initial_cost, sm.final_cost))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for env in ({}, {"ORACLE_NO_LIVE_BIAS": "1"}):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
