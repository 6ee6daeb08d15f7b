"""KLT kernel logic through the fiber emulator (no GPU); the real parity tests are tests/test_gpu_klt.py."""
import os
import subprocess

import pytest

import klt_compare
from pvio_amd import capi
from pvio_amd.solver import HipContext

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu")


@pytest.fixture(scope="module")
def emu_ctx():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    ctx = HipContext(lib=capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so")))
    yield ctx
    ctx.close()


def test_emulated_klt_matches_oracle(emu_ctx, oracle):
    klt_compare.check_klt(emu_ctx, oracle, 160, 120, 64)


def test_emulated_klt_odd_size_no_clahe(emu_ctx, oracle):
    klt_compare.check_klt(emu_ctx, oracle, 175, 131, 40, clahe=False)


def test_emulated_clahe_odd_size(emu_ctx, oracle):
    klt_compare.check_klt(emu_ctx, oracle, 150, 117, 30, clahe=True)



@pytest.mark.parametrize("blocks", [2, 3, 8])
def test_emulated_lk_unit_queue_is_bit_identical_to_a_wave_per_track(oracle, blocks, monkeypatch):
    """k_lk_track_units ((track, level) units from a queue in LDS, eight waves per block, a block's tracks b, b + G, ...) against
    k_lk_track (a wave per track) and the oracle: same status bytes, bit-identical positions, with growing and shrinking track counts
    (blocks with one track, with none, with 35)."""
    import numpy as np
    from pvio_amd import synth
    from pvio_amd.solver import HipImage, klt_track
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    monkeypatch.setenv("PVIO_HIP_LK_UNITS", "0")
    per_track = HipContext(lib=lib)
    monkeypatch.setenv("PVIO_HIP_LK_UNITS", "1")
    monkeypatch.setenv("PVIO_HIP_LK_BLOCKS", str(blocks))
    queue = HipContext(lib=lib)
    img0, img1, p, truth, init = synth.make_image_pair(160, 120, 70)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    imgs = [(HipImage(c, img0), HipImage(c, img1)) for c in (per_track, queue)]
    for n in (50, 7, 70, 1, 33):
        nA, sA, _ = klt_track(per_track, imgs[0][0], imgs[0][1], p[:n], init[:n])
        nB, sB, _ = klt_track(queue, imgs[1][0], imgs[1][1], p[:n], init[:n])
        n0, s0 = oracle.klt_track(P0, P1, p[:n], init[:n])
        assert (sA == sB).all() and (sB == s0).all()
        assert nA.tobytes() == nB.tobytes()
        assert np.abs(nB - n0)[s0 > 0].max() == 0.0
    for c in (per_track, queue):
        c.close()
