import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A hung kernel (or a worker subprocess that never returns) must FAIL its test, not stall the whole run: every test gets a
    15-minute ceiling when pytest-timeout is installed (the slowest one, the benched-batch replication at 1280x960, takes ~60 s)
    unless the command line sets its own --timeout."""
    if not config.pluginmanager.hasplugin("timeout") or getattr(config.option, "timeout", None):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
