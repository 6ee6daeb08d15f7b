"""pvio::BundleAdjustor adapter (pvio_amd/host/) exercised through the emulated kernel build (no GPU) and, under -m gpu,
through the product library."""
import pytest

import host_compare

CASES = {
    "vision": dict(n_frames=5, n_landmarks=40, visibility=4),
    "vio": dict(n_frames=5, n_landmarks=40, use_inertial=True, visibility=4),
    "vio_plane": dict(n_frames=5, n_landmarks=120, use_inertial=True, plane_fraction=0.5),
    # tracks of planes with fewer than 20 members: their reprojection blocks are listed twice / three times (bundle_adjustor.cpp:165-179);
    # the harness builds the small planes in the Map, the adapter has to find the multiplicities
    "vio_small_planes": dict(n_frames=5, n_landmarks=60, use_inertial=True, visibility=4, duplicate_fraction=0.4),
}


@pytest.fixture(scope="module")
def emu_host():
    import subprocess, os
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"), "libpvio_hipemu.so"])
    return host_compare.load("libpvio_host_emu.so")


@pytest.mark.parametrize("name", sorted(CASES))
def test_adapter_solve_emulated(emu_host, oracle, name):
    host_compare.check_adapter(emu_host, oracle, **CASES[name])


def test_adapter_marginalize_emulated(emu_host, oracle):
    host_compare.check_adapter_marginalize(emu_host, oracle, 0, n_frames=5, n_landmarks=50, use_inertial=True, visibility=4)
    host_compare.check_adapter_marginalize(emu_host, oracle, 2, n_frames=5, n_landmarks=50, use_inertial=True, visibility=4)


POST = dict(n_frames=12, n_landmarks=200, use_inertial=True, plane_fraction=0.4, plane_outliers=5)


def test_adapter_post_passes_emulated(emu_host, oracle):
    """plane-track re-validation (DLT triangulation, 0.1 m gate, plane erase, re-promotion to VALID) + depth gate / quality over
    ALL valid-or-plane tracks: bundle_adjustor.cpp:251-296"""
    moved, orphaned = host_compare.check_adapter_post_passes(emu_host, oracle, **POST)
    # the 5 off-plane tracks of each of the two planes (80 plane tracks, 40 per plane) are thrown out of their plane and
    # re-promoted; the remaining tracks of the first plane stay.  (The second plane loses all of its tracks: the sign
    # quirk of the plane factor's regularization row, SURVEY App. D item 3, drags the solved poses ~10 cm off there --
    # reference behaviour, reproduced by both sides.)
    assert moved[:5].all() and moved[40:45].all() and not moved[5:40].any()
    assert (moved == orphaned).all()


@pytest.mark.gpu
def test_adapter_post_passes_gpu(oracle):
    lib = host_compare.load("libpvio_host.so")
    moved, _ = host_compare.check_adapter_post_passes(lib, oracle, **POST)
    assert moved[:5].all() and moved[40:45].all() and not moved[5:40].any()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_adapter_solve_gpu(oracle, name):
    lib = host_compare.load("libpvio_host.so")
    kw = dict(CASES[name])
    kw["n_landmarks"] *= 5
    kw["n_frames"] = 8
    print(name, host_compare.check_adapter(lib, oracle, **kw))


@pytest.mark.gpu
def test_adapter_marginalize_gpu(oracle):
    lib = host_compare.load("libpvio_host.so")
    host_compare.check_adapter_marginalize(lib, oracle, 0, n_frames=8, n_landmarks=200, use_inertial=True, visibility=5)
