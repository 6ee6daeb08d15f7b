import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _prebuild_for_workers():
    """`pytest -n K`: the test modules build their helper libraries with `make` on first use; K workers doing that at once write the same objects.  The
    controller builds them once before the workers start (what is up to date is left alone; a failure here is left to the tests to report)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    jobs = [(os.path.join(ROOT, "oracle"), ["liboracle.so"]),
            (os.path.join(here, "hipemu"), ["libpvio_hipemu.so", "libpvio_hipemu_counters.so"]),
            (os.path.join(here, "host"), ["libpvio_host_emu.so", "libpvio_chain_hip_emu.so", "libpvio_chain_oracle.so", "libpvio_host.so", "libpvio_chain_hip.so", "pvio_headless"])]
    if os.path.isdir("/root/reference"):  # the reference's own sources, compiled in place (oracle/_ref): only where they exist
        jobs += [(os.path.join(ROOT, "oracle", "ref"), []), (os.path.join(ROOT, "oracle", "ref"), ["dropin"]), (os.path.join(ROOT, "oracle", "ref"), ["headless"])]
    for d, targets in jobs:
        try:
            subprocess.call(["make", "-s", "-C", d] + targets, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except OSError:
            pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    if getattr(config.option, "numprocesses", None) and not hasattr(config, "workerinput"):
        _prebuild_for_workers()


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never ends does not raise) must not hold the box until the run's own limit: every gpu test
    gets a time limit of its own, enforced from a watchdog thread that ends the process (a signal does not reach a thread blocked in
    hipStreamSynchronize).  pytest-timeout is in the image; without it the marker is registered here and ignored."""
    if not config.pluginmanager.hasplugin("timeout"):
        config.addinivalue_line("markers", "timeout: (pytest-timeout absent: ignored)")
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py
