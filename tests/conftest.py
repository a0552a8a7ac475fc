import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build host lib + oracle once (seconds; hipcc is NOT invoked here unless the HIP
    library is missing, in which case the C-ABI export test builds it)."""
    import subprocess
    pkg = os.path.join(ROOT, "pbrt-v3-spectral_amd")
    if not os.path.exists(os.path.join(pkg, "libmipt_host.so")):
        subprocess.check_call(["make", "host"], cwd=ROOT)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_pt.so")):
        subprocess.check_call(["make", "-C", "oracle"], cwd=ROOT)
    yield


KILLEROO = os.path.join(ROOT, "scenes", "killeroo-simple.pbrt")
CORNELL = os.path.join(ROOT, "scenes", "cornell-glass.pbrt")


@pytest.fixture(scope="session")
def pt():
    import pbrt_v3_spectral_amd as m
    return m


@pytest.fixture(scope="session")
def ob():
    import oracle_binding as m
    return m
