"""pvio::BundleAdjustor adapter (pvio_amd/host/) exercised through the emulated kernel build (no GPU) and, under -m gpu,
through the product library."""
import pytest

import host_compare

CASES = {
    "vision": dict(n_frames=5, n_landmarks=40, visibility=4),
    "vio": dict(n_frames=5, n_landmarks=40, use_inertial=True, visibility=4),
    "vio_plane": dict(n_frames=5, n_landmarks=120, use_inertial=True, plane_fraction=0.5),
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
