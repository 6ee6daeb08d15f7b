"""Kernel-logic check WITHOUT a GPU: the product sources (pvio_amd/csrc) compiled against the fiber emulator
(tests/hipemu, test infrastructure only) must reproduce the oracle iteration by iteration.  The real parity
tests are tests/test_gpu_ba.py (-m gpu, same checks through libpvio_hip.so)."""
import os
import subprocess

import numpy as np
import pytest

import ba_compare
from pvio_amd import BASummary, capi
from pvio_amd.solver import HipContext

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")


@pytest.fixture(scope="module")
def emu_ctx():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    ctx = HipContext(lib=lib, use_graph=True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", sorted(set(ba_compare.CASES) - ba_compare.GPU_ONLY))
def test_emulated_kernels_match_oracle(emu_ctx, oracle, name):
    pb = ba_compare.make(oracle, **ba_compare.CASES[name])
    ba_compare.check_against_oracle(emu_ctx, oracle, pb)


@pytest.mark.parametrize("name", sorted(ba_compare.BIG_CASES))
def test_emulated_kernels_match_oracle_at_the_metric_size(emu_ctx, oracle, name):
    """the windows the metric is quoted on (10 KF x 1000 landmarks) and their variants: the same checks the GPU tests make, without a GPU"""
    pb = ba_compare.make(oracle, **ba_compare.BIG_CASES[name])
    ba_compare.check_against_oracle(emu_ctx, oracle, pb)


@pytest.mark.parametrize("seed", [15, 20, 37, 6, 25])
def test_emulated_sweep_windows_within_oracle_spread(emu_ctx, oracle, seed):
    """The three windows of the random sweep that miss north_star's 1e-6 on the GPU (15: 31 x 469, 20: 17 x 816 with planes, 37: 16 x 1391; all
    vision-only, every landmark seen by two frames) are the three on which the ORACLE differs from ITSELF by more than that when its sums over
    landmarks run in another order (ba_compare.oracle_spread): conditioning, not the kernels' arithmetic.  Held here to 4 x that spread in
    the kernels' summation order (CPU doubles); 6 and 25 are well-conditioned neighbours held to the plain 1e-6.  GPU: tests/test_gpu_ba.py."""
    kw, pb = ba_compare.sweep_window(oracle, seed)
    sp = ba_compare.oracle_spread(oracle, pb)
    assert sp["same_decisions"]
    assert (sp["state"] > 2.5e-7) == (seed in (15, 20, 37)), sp
    r = ba_compare.check_against_oracle_within_spread(emu_ctx, oracle, pb)
    assert r["worst_state_diff"] <= r["tol"]


def test_oracle_summation_orders_agree_on_a_well_conditioned_window(oracle):
    pb = ba_compare.make(oracle, **ba_compare.BIG_CASES["metric_10x1000_vio"])
    sp = ba_compare.oracle_spread(oracle, pb)
    # (1.8e-9 in the states -- the size of the kernels' own distance from the oracle on this window, 1.7e-9: profiles/r4_dropin_gpu.txt)
    assert sp["same_decisions"] and sp["state"] < 1e-7 and sp["cost_rel"] < 1e-7, sp


def test_emulated_eager_launches_match_graph_replay(emu_ctx, oracle):
    pb = ba_compare.make(oracle, **ba_compare.CASES["vio_partial"])
    eager = HipContext(lib=emu_ctx.lib, use_graph=False)
    st_a, sm_a = emu_ctx.solve(pb)
    st_b, sm_b = eager.solve(pb)
    assert (st_a.frame_state == st_b.frame_state).all() and (st_a.lm_inv_depth == st_b.lm_inv_depth).all()
    eager.close()


def test_emulated_resident_solve_is_repeatable(emu_ctx, oracle):
    from pvio_amd import BAState, BASummary
    pb = ba_compare.make(oracle, **ba_compare.CASES["vio_small"])
    emu_ctx.upload(pb)
    outs = []
    for _ in range(2):
        sm = BASummary(pb)
        emu_ctx.solve_resident(sm)
        st = BAState(pb)
        emu_ctx.download(st)
        outs.append((st.frame_state.copy(), st.lm_inv_depth.copy(), sm.num_iterations))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all() and outs[0][2] == outs[1][2]


def test_rejects_bad_input(emu_ctx, oracle):
    from pvio_amd.solver import HipError
    pb = ba_compare.make(oracle, **ba_compare.CASES["vision_small"])
    pb.obs_frame[1] = pb.obs_frame[0]  # same target frame twice for one landmark
    with pytest.raises(HipError):
        emu_ctx.solve(pb)
    # CSR that does not start at 0 / is not monotone, plane observation outside the window, negative counts, a second
    # rotation prior on one frame: PVIO_ERR_INVALID_ARGUMENT / UNSUPPORTED, never an out-of-bounds read
    def broken(edit, **kw):
        q = ba_compare.make(oracle, **dict(ba_compare.CASES["plane"], **kw))
        edit(q)
        with pytest.raises(HipError):
            emu_ctx.solve(q)

    def shift_ptr(q):
        q.lm_obs_ptr = q.lm_obs_ptr.copy()
        q.lm_obs_ptr[0] = -1
    broken(shift_ptr)

    def dip(q):
        q.lm_obs_ptr = q.lm_obs_ptr.copy()
        q.lm_obs_ptr[2] = q.lm_obs_ptr[1] - 1
    broken(dip)

    def plane_oob(q):
        q.plane_obs_frame = q.plane_obs_frame.copy()
        q.plane_obs_frame[3] = q.n_frames
    broken(plane_oob)

    def neg_iter(q):
        q.max_iterations = -3
    broken(neg_iter)

    def two_rot(q):
        q.rot_prior_frame = np.array([1, 1], np.int32)
        q.rot_prior_q0 = np.tile([0, 0, 0, 1.0], (2, 1))
        q.rot_prior_sqrt_info = np.tile(np.eye(3).ravel(), (2, 1))
    broken(two_rot)


import marg_compare  # noqa: E402


@pytest.mark.parametrize("victim", [0, 2])
def test_emulated_marginalize_matches_oracle(emu_ctx, oracle, victim):
    marg_compare.check_marginalize(emu_ctx, oracle, victim, n_frames=5, n_landmarks=60, use_inertial=True, visibility=4)


@pytest.mark.parametrize("victim", [1, 2])
def test_emulated_marginalize_folds_the_victims_rotation_prior(emu_ctx, oracle, victim):
    marg_compare.check_marginalize(emu_ctx, oracle, victim, n_frames=5, n_landmarks=40, use_inertial=True, visibility=4, rot_prior_frames=(1, 3, 4))


def test_emulated_marginalize_last_frame_no_prior(emu_ctx, oracle):
    import numpy as np
    pb, st = marg_compare.solved_window(oracle, n_frames=4, n_landmarks=40, use_inertial=True)
    pb.prior_frames = np.zeros(0, np.int32)  # no previous marginalization factor
    S0, s0, IM0, iv0 = oracle.marginalize(pb, st, 3)
    S1, s1, IM1, iv1 = emu_ctx.marginalize(pb, st, 3)
    np.testing.assert_allclose(IM1, IM0, rtol=1e-7, atol=1e-9 * np.abs(IM0).max())
    np.testing.assert_allclose(S1.T @ S1, S0.T @ S0, rtol=1e-6, atol=1e-7 * np.abs(IM0).max())


# ---- rarely taken solver paths, forced by fault injection on both sides (pvio_hip_opts::debug_*, oracle_debug_fault_injection) ----
FAULTS = [
    dict(fail=1, invalid=0),   # one failed factorization: mu x10, the accepted point is re-linearized, the iteration goes on
    dict(fail=3, invalid=0),   # three in a row inside one iteration
    dict(fail=0, invalid=1),   # HandleInvalidStep once
    dict(fail=0, invalid=2),
    dict(fail=0, invalid=5),   # five consecutive invalid steps = FAILURE, x unchanged
    dict(fail=8, invalid=0),   # mu escalates to max_mu: linear solver failure until the five-strikes rule ends the solve
]


def _fault_case(make_ctx, oracle, fail, invalid, **kw):
    import ctypes as C

    from oracle import oracle_py
    from pvio_amd import synth

    pb = synth.make_window(**kw) if not kw.get("use_inertial") else synth.make_window(preintegrate=oracle_py.preintegrate, **kw)
    L = oracle_py.lib()
    L.oracle_debug_fault_injection(C.c_int32(fail), C.c_int32(invalid))
    ctx = make_ctx(fail, invalid)
    try:
        return ba_compare.check_against_oracle(ctx, oracle, pb)
    finally:
        L.oracle_debug_fault_injection(C.c_int32(0), C.c_int32(0))
        ctx.close()


@pytest.mark.parametrize("fault", FAULTS, ids=lambda f: "fail%d_invalid%d" % (f["fail"], f["invalid"]))
@pytest.mark.parametrize("inertial", [False, True])
def test_emulated_fault_paths_match_oracle(oracle, fault, inertial):
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    mk = lambda f, i: HipContext(lib=lib, debug_fail_factorizations=f, debug_invalid_steps=i)  # noqa: E731
    _fault_case(mk, oracle, fault["fail"], fault["invalid"], n_frames=4, n_landmarks=30, use_inertial=inertial)


@pytest.mark.parametrize("fail", [1, 3, 8])
def test_emulated_negative_pivot_is_detected_like_an_injected_failure(oracle, fail):
    """debug_fail_factorizations = 1000 + n makes the next n factorizations MEET a negative pivot (the fourth of panel 0) instead of discarding their
    result: the factor wave's own detection -- one NaN test of the panel's last reciprocal pivot -- takes the solver down the same path the oracle's
    injected failures do.  (A window with more than 256 landmarks: the form of k_dense that patches the diagonal late is the one that has the hook.)"""
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    mk = lambda f, i: HipContext(lib=lib, debug_fail_factorizations=1000 + f, debug_invalid_steps=i)  # noqa: E731
    _fault_case(mk, oracle, fail, 0, n_frames=4, n_landmarks=300, use_inertial=True)


# ---- large-window accumulation (linearize_mode = 2): Schur complement on 16x16 f64 MFMA tiles, per-frame direct rows,
# contiguous chunk ranges with anchor flushes; forced here on windows small enough for the emulator ----
MM_CASES = {
    "vio_small": ba_compare.CASES["vio_small"],
    "vision_partial": ba_compare.CASES["vision_partial"],
    "vio_plane": ba_compare.CASES["vio_plane"],
    "vio_10x200_anchor_changes": dict(n_frames=10, n_landmarks=200, use_inertial=True, visibility=5),  # one workgroup, 8 chunks, 6 anchors
    "vision_16x120_twelve_tiles_per_wave": dict(n_frames=16, n_landmarks=120, visibility=9),
    # N <= 10: the direct part is dealt to the four waves (partial row sets added in wave order at anchor flushes and at the end)
    "vio_4x150_four_wave_direct_part": dict(n_frames=4, n_landmarks=150, use_inertial=True, visibility=3),
    "metric_10x1000_vio_four_wave_direct_part": dict(n_frames=10, n_landmarks=1000, use_inertial=True),
    # round 6 (ba_lin_tp.h): duplicate residual blocks; 30 frames = two direct tasks per thread, twenty tiles per wave, 16 landmarks per chunk;
    # a short-visibility window whose chunks hold the full 64 landmarks
    "vio_duplicate_blocks": ba_compare.CASES["vio_duplicate_blocks"],
    "vision_30x90_two_direct_tasks_per_thread": dict(n_frames=30, n_landmarks=90, visibility=11),
    "vio_8x400_full_chunks": dict(n_frames=8, n_landmarks=400, use_inertial=True, visibility=3),
    # 32 frames: 91 tiles (23 per wave), the frame mask of a landmark uses all 32 bits -- the walk over its unseen frames once shifted by 32 (a hang on the GPU,
    # a crash in the emulator: found by an edge-case run, fixed before it shipped)
    "vision_32x70_partial_visibility": dict(n_frames=32, n_landmarks=70, visibility=10),
}


@pytest.fixture(scope="module")
def emu_ctx_mm(emu_ctx):
    ctx = HipContext(lib=emu_ctx.lib, use_graph=True, linearize_mode=2)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", sorted(MM_CASES))
def test_emulated_mfma_tile_linearization_matches_oracle(emu_ctx_mm, oracle, name):
    pb = ba_compare.make(oracle, **MM_CASES[name])
    ba_compare.check_against_oracle(emu_ctx_mm, oracle, pb)


def test_emulated_large_window_role_with_unsorted_anchors_and_a_fixed_frame(emu_ctx_mm, oracle):
    """ba_lin_tp.h cuts a chunk wherever the anchor frame changes and flushes the (target, anchor) blocks there: a window whose landmarks come in
    RANDOM anchor order (every chunk a few landmarks, an anchor flush in front of most of them, the same anchor coming back) and one with a fixed
    frame (constant blocks have no Jacobian) must give what the oracle gives."""
    import numpy as np
    from pvio_amd import synth
    pb = ba_compare.make(oracle, n_frames=7, n_landmarks=150, use_inertial=True, visibility=3)
    pb2 = synth.permute_landmarks(pb, np.random.default_rng(3).permutation(pb.n_landmarks))
    assert (np.diff(pb2.lm_anchor_frame) != 0).sum() > 60
    ba_compare.check_against_oracle(emu_ctx_mm, oracle, pb2)
    pb3 = ba_compare.make(oracle, n_frames=6, n_landmarks=120, visibility=4)
    pb3.frame_fixed[2] = 1
    ba_compare.check_against_oracle(emu_ctx_mm, oracle, pb3)


@pytest.mark.parametrize("n", [2, 3, 10, 11, 15, 16, 22, 23, 27, 28, 31, 32])
def test_emulated_large_window_role_at_the_boundaries_of_its_geometry_classes(emu_ctx_mm, oracle, n):
    """frame counts on both sides of every switch of the role's compile-time geometry (tiles per wave 3 | 6 | 12 | 17 | 20 | 23, one | two direct tasks per thread) and
    of its U row width: two iterations against the oracle"""
    pb = ba_compare.make(oracle, n_frames=n, n_landmarks=30 + 2 * n, use_inertial=(n % 2 == 0), visibility=max(2, min(n, 3 + n // 4)), seed=300 + n, max_iterations=2)
    ba_compare.check_against_oracle(emu_ctx_mm, oracle, pb)


@pytest.mark.parametrize("victim", [0, 2, 5])
def test_emulated_mfma_tile_marginalization_matches_oracle(emu_ctx_mm, oracle, victim):
    import marg_compare
    marg_compare.check_marginalize(emu_ctx_mm, oracle, victim, n_frames=6, n_landmarks=40, use_inertial=True, visibility=4)


def test_emulated_one_rank_sharded_path(oracle):
    """debug_force_sharded with world_size 1: the sharded code path (eager launches, all-reduces through the communicator,
    assembly from the reduced buffer) with an identity all-reduce; tests/test_gpu_ba.py runs the same through RCCL."""
    import ctypes as C
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))

    @C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_long, C.c_int)
    def identity(buf, n, op_max):
        return 0
    lib.hipemu_set_allreduce(identity)
    ctx = HipContext(lib=lib, force_sharded=True)
    try:
        uid = (C.c_uint8 * 128)()
        assert lib.pvio_hip_comm_unique_id(uid) == 0 and lib.pvio_hip_comm_init(ctx.ctx, uid, 0, 1) == 0
        for name in ("vio_partial", "vio_plane"):
            ba_compare.check_against_oracle(ctx, oracle, ba_compare.make(oracle, **ba_compare.CASES[name]))
    finally:
        ctx.close()
        lib.hipemu_set_allreduce(C.cast(None, C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_long, C.c_int)))


# ---- pvio_hip_opts::reuse_identical_candidates: a candidate that is bit-identical to the one just rejected is not evaluated again ---------
REUSE_CASES = {  # (window, candidate evaluations the short-circuit must save: the runs of consecutive rejections with |gn| <= radius in its trace)
    "vio_partial": 2, "vio_zero_bias_quirk": 4, "vio_plane": 1, "vision_partial": 0, "vio_rot_prior": None, "vio_duplicate_blocks": None,
}


@pytest.fixture(scope="module")
def emu_ctx_reuse(emu_ctx):
    ctx = HipContext(lib=emu_ctx.lib, use_graph=True, reuse_identical_candidates=True)
    yield ctx
    ctx.close()


REUSE_CASES.update({n: None for n in ba_compare.CASES if n not in REUSE_CASES})  # the adapter's default is ON -- the whole matrix runs with it


@pytest.mark.parametrize("name", sorted(REUSE_CASES))
def test_emulated_identical_candidates_are_not_evaluated_twice(emu_ctx_reuse, oracle, name):
    """same iterations, same records, same states after every iteration as the oracle (which, like Ceres, evaluates every candidate), with the
    repeated evaluations skipped"""
    pb = ba_compare.make(oracle, **ba_compare.CASES[name])
    ba_compare.check_against_oracle(emu_ctx_reuse, oracle, pb)
    if REUSE_CASES[name] is not None:
        assert emu_ctx_reuse.last_candidate_repeats() == REUSE_CASES[name]


def test_emulated_identical_candidates_metric_window(emu_ctx_reuse, oracle):
    """the window the metric is quoted on ends in four rejections of ONE candidate (|gn| = 0.053 << radius): three evaluations saved, nothing else changes;
    and a resident re-solve starts from a clean record"""
    pb = ba_compare.make(oracle, **ba_compare.BIG_CASES["metric_10x1000_vio"])
    ba_compare.check_against_oracle(emu_ctx_reuse, oracle, pb)
    assert emu_ctx_reuse.last_candidate_repeats() == 3
    emu_ctx_reuse.upload(pb)
    for _ in range(2):
        sm = BASummary(pb, trace=False)
        emu_ctx_reuse.solve_resident(sm)
        assert sm.num_iterations == 10 and emu_ctx_reuse.last_candidate_repeats() == 3


def test_emulated_counter_form_of_the_lookahead_loop(oracle):
    """k_dense's look-ahead loop has two forms of its hand-overs: two hardware barriers per panel (round 5, shipped) and two polled LDS counters
    (-DPVIO_DENSE_LA_COUNTERS, rounds 3-4, kept for A/Bs: tests/micro/build_variant.py la_counters).  The counter form in its own emulated build (the two
    builds define the same kernel symbols: one process each): the metric window and a small VIO window against the oracle, as for the shipped form."""
    import sys
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu_counters.so"])
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import ba_compare\nfrom oracle import oracle_py as O\nfrom pvio_amd import capi\nfrom pvio_amd.solver import HipContext\nO.build()\n"
            "ctx = HipContext(lib=capi.load(%r), use_graph=True)\n"
            "for kw in (ba_compare.BIG_CASES['metric_10x1000_vio'], ba_compare.CASES['vio_small'], ba_compare.CASES['vio_11_frames_lds_limit']):\n"
            "    print(ba_compare.check_against_oracle(ctx, O, ba_compare.make(O, **kw)))\n") % (
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), os.path.join(EMU_DIR, "libpvio_hipemu_counters.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("worst_state_diff") == 3
