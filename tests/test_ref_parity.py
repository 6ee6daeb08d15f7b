"""The HIP path held directly against the REFERENCE'S OWN code: oracle/_ref/libpvio_ref.so is bundle_adjustor.cpp, the
estimation/ceres cost functions, preintegrator.cpp, lie_algebra.cpp and the map layer of /root/reference compiled unedited
(oracle/ref/Makefile; Eigen and Ceres are the stand-ins of oracle/ref/).  The library is built in the container that has the
reference tree and travels to the GPU box with the snapshot -- nothing here reads /root/reference at run time.

CPU suite: the kernel sources through the fiber emulator.  -m gpu: libpvio_hip.so on the MI355X, incl. the window the metric is
quoted on (10 KF x 1000 landmarks, full VIO factor set)."""
import os
import subprocess

import numpy as np
import pytest

import ba_compare
from pvio_amd import capi
from pvio_amd.solver import HipContext
from test_ref_pin import ref  # noqa: F401  (fixture)

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")
# RotationPriorFactor has no reference counterpart; everything else the reference can express
SMALL = ["vision_partial", "vio_partial", "plane", "vio_plane", "vio_zero_bias_quirk", "config1_10x200", "vio_13_frames_global_matrix", "vio_duplicate_blocks"]


@pytest.fixture(scope="module")
def emu_ctx():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    ctx = HipContext(lib=lib, use_graph=True)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("name", SMALL)
def test_emulated_kernels_match_reference_sources(emu_ctx, ref, oracle, name):
    pb = ba_compare.make(oracle, **ba_compare.CASES[name])
    print(name, ba_compare.check_against_reference(emu_ctx, ref, pb))


@pytest.mark.parametrize("name", ["metric_10x1000_vio", "vio_plane_10x600"])
def test_emulated_kernels_match_reference_sources_metric_window(emu_ctx, ref, oracle, name):
    """the kernels (in the emulator) against the reference's own sources on the window the metric is quoted on"""
    pb = ba_compare.make(oracle, **ba_compare.BIG_CASES[name])
    print(name, ba_compare.check_against_reference(emu_ctx, ref, pb))


def test_emulated_marginalize_matches_reference_sources(emu_ctx, ref, oracle):
    """pvio_hip_ba_marginalize (kernels in the emulator, host tail as shipped) against the reference's marginalize_frame (bundle_adjustor.cpp:348-599)"""
    import marg_compare
    for victim in (0, 3):
        pb, st = marg_compare.solved_window(oracle, regular_prior=(victim != 0), n_frames=8, n_landmarks=200, use_inertial=True, visibility=5)
        S1, s1 = emu_ctx.marginalize(pb, st, victim)[:2]
        trk, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
        S0, s0, IM0, iv0 = ref.marginalize(pb, st.frame_state, trk, victim)
        scale = np.abs(IM0).max()
        np.testing.assert_allclose(S1.T @ S1, IM0, rtol=1e-6, atol=1e-7 * scale)
        np.testing.assert_allclose(S1.T @ s1, iv0, rtol=1e-6, atol=1e-6 * np.abs(iv0).max())


@pytest.fixture(scope="module")
def gpu_ctx():
    ctx = HipContext(device=0, use_graph=True)  # raises if the library or the GPU is missing: no fallback
    yield ctx
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL)
def test_gpu_matches_reference_sources(gpu_ctx, ref, oracle, name):
    pb = ba_compare.make(oracle, **ba_compare.CASES[name])
    print(name, ba_compare.check_against_reference(gpu_ctx, ref, pb))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["metric_10x1000_vio", "vio_plane_10x600"])
def test_gpu_matches_reference_sources_metric_window(gpu_ctx, ref, oracle, name):
    """the reference's solve of the 10 x 1000 window goes through mini-Ceres' dense Cholesky of all 1150 unknowns (seconds)"""
    pb = ba_compare.make(oracle, **ba_compare.BIG_CASES[name])
    print(name, ba_compare.check_against_reference(gpu_ctx, ref, pb))


@pytest.mark.gpu
def test_gpu_marginalize_matches_reference_sources(gpu_ctx, ref, oracle):
    """pvio_hip_ba_marginalize against the reference's marginalize_frame (bundle_adjustor.cpp:348-599) after a GPU solve"""
    import marg_compare
    for victim in (0, 3):
        pb, st = marg_compare.solved_window(oracle, regular_prior=(victim != 0), n_frames=8, n_landmarks=200, use_inertial=True, visibility=5)
        S1, s1 = gpu_ctx.marginalize(pb, st, victim)[:2]
        trk, _ = ref.tracks_of_problem(pb, inv_depth=st.lm_inv_depth)
        S0, s0, IM0, iv0 = ref.marginalize(pb, st.frame_state, trk, victim)
        scale = np.abs(IM0).max()
        np.testing.assert_allclose(S1.T @ S1, IM0, rtol=1e-6, atol=1e-7 * scale)
        np.testing.assert_allclose(S1.T @ s1, iv0, rtol=1e-6, atol=1e-6 * np.abs(iv0).max())
