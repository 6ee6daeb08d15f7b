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

