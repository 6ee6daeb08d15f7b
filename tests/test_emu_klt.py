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
def test_emulated_lk_forms_are_bit_identical(oracle, blocks, monkeypatch):
    """The three launch forms of the LK search -- k_lk_track_levels (a workgroup per track, a wave per level: the default), k_lk_track (a wave
    per track), k_lk_track_units ((track, level) units from a queue in LDS, eight waves per block, a block's tracks b, b + G, ...) -- and the
    oracle: same status bytes, bit-identical positions, with growing and shrinking track counts (blocks with one track, with none, with 35)."""
    import numpy as np
    from pvio_amd import synth
    from pvio_amd.solver import HipImage, klt_track
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "3")
    levels = HipContext(lib=lib)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "1")
    per_track = HipContext(lib=lib)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "2")
    monkeypatch.setenv("PVIO_HIP_LK_BLOCKS", str(blocks))
    queue = HipContext(lib=lib)
    img0, img1, p, truth, init = synth.make_image_pair(160, 120, 70)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    imgs = [(HipImage(c, img0), HipImage(c, img1)) for c in (per_track, queue, levels)]
    for n in (50, 7, 70, 1, 33):
        nA, sA, _ = klt_track(per_track, imgs[0][0], imgs[0][1], p[:n], init[:n])
        nB, sB, _ = klt_track(queue, imgs[1][0], imgs[1][1], p[:n], init[:n])
        nC, sC, _ = klt_track(levels, imgs[2][0], imgs[2][1], p[:n], init[:n])
        n0, s0 = oracle.klt_track(P0, P1, p[:n], init[:n])
        assert (sA == sB).all() and (sB == s0).all() and (sC == s0).all()
        assert nA.tobytes() == nB.tobytes() and nA.tobytes() == nC.tobytes()
        assert np.abs(nB - n0)[s0 > 0].max() == 0.0
    for c in (per_track, queue, levels):
        c.close()


@pytest.mark.parametrize("size", [(42, 42), (90, 90), (200, 180)])
def test_emulated_lk_forms_on_short_pyramids_and_border_tracks(oracle, size, monkeypatch):
    """The level-per-wave form and the unit queue where their level bookkeeping is exercised: pyramids of 1, 3 and 4 levels (a level is dropped once it is no larger
    than the 21-pixel window), tracks that start ON the image border, outside the 20-pixel gate, with initial guesses outside the image (levels
    whose template or search window leaves the padded image are skipped or end the track): status bytes identical to the oracle and to the
    wave-per-track kernel, positions bit-identical."""
    import numpy as np
    from pvio_amd import synth
    from pvio_amd.solver import HipImage, klt_track
    w, h = size
    subprocess.check_call(["make", "-s", "-C", EMU_DIR, "libpvio_hipemu.so"])
    lib = capi.load(os.path.join(EMU_DIR, "libpvio_hipemu.so"))
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "3")
    levels = HipContext(lib=lib)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "1")
    per_track = HipContext(lib=lib)
    monkeypatch.setenv("PVIO_HIP_LK_FORM", "2")
    monkeypatch.setenv("PVIO_HIP_LK_BLOCKS", "3")
    queue = HipContext(lib=lib)
    img0, img1, p, truth, init = synth.make_image_pair(w, h, 24)
    P0, P1 = oracle.build_pyramid(oracle.clahe(img0)), oracle.build_pyramid(oracle.clahe(img1))
    assert len(P0) == {42: 1, 90: 3, 200: 4}[w]
    rng = np.random.default_rng(w)
    edge = np.array([(0, 0), (w - 1, h - 1), (0.5, h / 2), (w / 2, 0.25), (w - 1, 3), (20, 20), (19.99, h / 2), (w - 20.01, h - 20.01)], np.float32)
    far = (edge + rng.uniform(-60, 60, edge.shape)).astype(np.float32)  # guesses that may lie far outside the image
    prev = np.concatenate([p, edge, edge]).astype(np.float32)
    guess = np.concatenate([init, edge, far]).astype(np.float32)
    imgs = [(HipImage(c, img0), HipImage(c, img1)) for c in (per_track, queue, levels)]
    nA, sA, _ = klt_track(per_track, imgs[0][0], imgs[0][1], prev, guess)
    nB, sB, _ = klt_track(queue, imgs[1][0], imgs[1][1], prev, guess)
    nC, sC, _ = klt_track(levels, imgs[2][0], imgs[2][1], prev, guess)
    n0, s0 = oracle.klt_track(P0, P1, prev, guess)
    assert (sA == sB).all() and (sB == s0).all() and (sC == s0).all()
    assert nA.tobytes() == nB.tobytes() and nA.tobytes() == nC.tobytes()
    assert (s0 > 0).sum() >= (4 if w > 42 else 0) and (s0 == 0).sum() >= 6  # both outcomes occur
    assert (nB[s0 > 0] == n0[s0 > 0]).all()
    for c in (per_track, queue, levels):
        c.close()
