"""Race hunting: the fiber emulator normally runs the threads of a barrier interval in index order (thread 0 first).  Code
that only works because of that order (a flag written by thread 0 and read by the others without a barrier in between --
this found one in the dense kernel) breaks when the order changes: rerun a slice of the emulated suite in reverse and in
interleaved order.  HIPEMU_ORDER is read when the emulator library loads, hence the subprocesses."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
# (the slice grew with the suite: it is kept to what exercises every kernel once -- both window kinds, plane and rotation-prior
# roles, marginalization (the fault paths are single-thread control code) -- so that the two reruns stay under half a minute each)
SLICE = "vio_small or vision_small or vio_plane or vio_rot_prior or marginalize_matches_oracle"


@pytest.mark.parametrize("order", ["reverse", "interleave"])
def test_emulated_kernels_do_not_depend_on_thread_order(order):
    env = dict(os.environ, HIPEMU_ORDER=order)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_emu_ba.py"), os.path.join(HERE, "test_emu_klt.py"),
                        os.path.join(HERE, "test_emu_gftt.py"), "-x", "-q", "-k", SLICE + " or klt or detection", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
