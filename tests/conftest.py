import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def td():
    """The HIP library, initialised on GPU 0. Fails loudly if the extension or the GPU is missing."""
    import taxidispatcher_amd as td_mod
    td_mod.init(0)
    yield td_mod
    td_mod.shutdown()


@pytest.fixture()
def general_solver(td):
    """td_assign with the line-metric attempt (td_line.hip) switched off: the general solver alone."""
    was = td.set_line_metric(False)
    yield td
    td.set_line_metric(was)
