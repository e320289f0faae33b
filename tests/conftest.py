import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The tests call through the C ABI: build libenarf_hip.so in-tree if a fresh checkout has not done so yet (hipcc
    cross-compiles for gfx950 without a GPU; a missing hipcc surfaces as the build error)."""
    from enarf_gan_amd import build
    if not os.path.exists(build.LIB):
        build.build()
    return build.LIB
