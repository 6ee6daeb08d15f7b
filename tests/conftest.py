import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


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
