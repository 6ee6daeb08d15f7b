"""Edge cases of the bundle-adjustment path (the reference has no tests; these are the inputs its `solve` can meet):
ragged landmark lists with anchor-only landmarks, a window without any landmark (IMU + prior only), frames fixed or
unreferenced (Ceres drops constant / unused parameter blocks), a two-frame window.  Emulated kernels here, the product
library under `-m gpu`; both against the CPU oracle, iteration by iteration."""
import os
import subprocess

import numpy as np
import pytest

import ba_compare
from pvio_amd import capi
from pvio_amd.solver import HipContext

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")


def _drop_observations(pb, keep):
    """Keep observation o iff keep[o]; landmarks may end up with no factor at all (anchor only)."""
    ptr = pb.lm_obs_ptr
    new_ptr = np.zeros_like(ptr)
    for l in range(pb.n_landmarks):
        new_ptr[l + 1] = new_ptr[l] + int(keep[ptr[l]:ptr[l + 1]].sum())
    pb.obs_frame = np.ascontiguousarray(pb.obs_frame[keep])
    pb.obs_z = np.ascontiguousarray(pb.obs_z[keep])
    pb.lm_obs_ptr = new_ptr
    return pb


def _remove_landmarks(pb):
    pb.lm_anchor_frame = np.zeros(0, np.int32)
    pb.lm_anchor_z = np.zeros((0, 2))
    pb.lm_obs_ptr = np.zeros(1, np.int32)
    pb.obs_frame = np.zeros(0, np.int32)
    pb.obs_z = np.zeros((0, 2))
    pb.lm_inv_depth = np.zeros(0)
    if pb.truth_inv_depth is not None:
        pb.truth_inv_depth = np.zeros(0)
    return pb


def _cases(oracle):
    rng = np.random.default_rng(7)
    out = {}
    pb = ba_compare.make(oracle, n_frames=6, n_landmarks=60, use_inertial=True, visibility=4)
    keep = rng.random(pb.n_obs) < 0.6
    for l in range(0, pb.n_landmarks, 5):  # every fifth landmark keeps only its anchor observation
        keep[pb.lm_obs_ptr[l]:pb.lm_obs_ptr[l + 1]] = False
    out["ragged_vio"] = _drop_observations(pb, keep)
    pb = ba_compare.make(oracle, n_frames=5, n_landmarks=40, visibility=3)
    keep = np.ones(pb.n_obs, bool)
    keep[pb.obs_frame == 2] = False  # frame 2 is never a target ...
    pb = _drop_observations(pb, keep)
    sel = pb.lm_anchor_frame != 2      # ... nor an anchor: an unreferenced parameter block
    if not sel.all():
        ptr = pb.lm_obs_ptr
        obs_keep = np.concatenate([np.full(ptr[l + 1] - ptr[l], sel[l]) for l in range(pb.n_landmarks)]) if pb.n_obs else np.zeros(0, bool)
        counts = (ptr[1:] - ptr[:-1])[sel]
        pb.obs_frame, pb.obs_z = pb.obs_frame[obs_keep], pb.obs_z[obs_keep]
        pb.lm_anchor_frame, pb.lm_anchor_z, pb.lm_inv_depth = pb.lm_anchor_frame[sel], pb.lm_anchor_z[sel], pb.lm_inv_depth[sel]
        pb.lm_obs_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        if pb.truth_inv_depth is not None:
            pb.truth_inv_depth = pb.truth_inv_depth[sel]
    out["unreferenced_frame_vision"] = pb
    pb = ba_compare.make(oracle, n_frames=5, n_landmarks=50, use_inertial=True, visibility=4)
    pb.frame_fixed[:] = 1
    pb.frame_fixed[3] = 0               # one free pose among fixed ones
    out["mostly_fixed_vio"] = pb
    out["no_landmarks_vio"] = _remove_landmarks(ba_compare.make(oracle, n_frames=4, n_landmarks=8, use_inertial=True))
    out["two_frames_vision"] = ba_compare.make(oracle, n_frames=2, n_landmarks=30)
    return out


NAMES = ["ragged_vio", "unreferenced_frame_vision", "mostly_fixed_vio", "no_landmarks_vio", "two_frames_vision"]


@pytest.fixture(scope="module")
def emu_ctx():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    ctx = HipContext(lib=capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so")))
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", NAMES)
def test_emulated_edge_case_matches_oracle(emu_ctx, oracle, name):
    print(name, ba_compare.check_against_oracle(emu_ctx, oracle, _cases(oracle)[name]))


@pytest.fixture(scope="module")
def emu_ctx_tp():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    ctx = HipContext(lib=capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so")), linearize_mode=2)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", NAMES)
def test_emulated_edge_case_through_the_large_window_role(emu_ctx_tp, oracle, name):
    """linearize_mode 2 forced on the degenerate shapes (round 6: a window WITHOUT landmarks has no chunks -- upload() divided by the chunk count; such a
    window takes the register-tile role's empty walk now)."""
    print(name, ba_compare.check_against_oracle(emu_ctx_tp, oracle, _cases(oracle)[name]))


def _sparse_cases(oracle):
    out = {"one_landmark": ba_compare.make(oracle, n_frames=4, n_landmarks=1), "one_landmark_vio": ba_compare.make(oracle, n_frames=3, n_landmarks=1, use_inertial=True)}
    pb = ba_compare.make(oracle, n_frames=6, n_landmarks=40, visibility=4)  # landmarks without observations at the ends and in the middle of a chunk
    counts = np.diff(pb.lm_obs_ptr).copy()
    kill = np.zeros(len(counts), bool)
    kill[[0, 7, 8, 20, 39]] = True
    keep = np.repeat(~kill, counts)
    pb.obs_frame, pb.obs_z = pb.obs_frame[keep], pb.obs_z[keep]
    counts[kill] = 0
    pb.lm_obs_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    out["empty_landmarks"] = pb
    pb = ba_compare.make(oracle, n_frames=4, n_landmarks=12, use_inertial=True)  # chunks, but not a single factor
    pb.obs_frame, pb.obs_z = pb.obs_frame[:0], pb.obs_z[:0]
    pb.lm_obs_ptr = np.zeros(len(pb.lm_obs_ptr), np.int32)
    out["all_landmarks_empty_vio"] = pb
    return out


SPARSE = ["one_landmark", "one_landmark_vio", "empty_landmarks", "all_landmarks_empty_vio"]


@pytest.mark.parametrize("name", SPARSE)
def test_emulated_large_window_role_on_nearly_empty_windows(emu_ctx_tp, oracle, name):
    print(name, ba_compare.check_against_oracle(emu_ctx_tp, oracle, _sparse_cases(oracle)[name]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SPARSE)
def test_gpu_large_window_role_on_nearly_empty_windows(oracle, name):
    ctx = HipContext(device=0, linearize_mode=2)
    try:
        print(name, ba_compare.check_against_oracle(ctx, oracle, _sparse_cases(oracle)[name]))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_edge_case_through_the_large_window_role(oracle, name):
    ctx = HipContext(device=0, linearize_mode=2)
    try:
        print(name, ba_compare.check_against_oracle(ctx, oracle, _cases(oracle)[name]))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_edge_case_matches_oracle(oracle, name):
    ctx = HipContext(device=0)
    try:
        print(name, ba_compare.check_against_oracle(ctx, oracle, _cases(oracle)[name]))
    finally:
        ctx.close()
