import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "slam-experiments_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


if os.environ.get("SLAMHIP_TEST_LIB"):      # development aid: run the suite against an experiment build (tools/build_exp.sh)
    from slamhip import _lib as _slamhip_lib

    _slamhip_lib.LIB_PATH = os.path.abspath(os.environ["SLAMHIP_TEST_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Make sure libslamhip.so and liboracle.so exist (built in-tree; they travel to the GPU box)."""
    import __graft_entry__ as g

    if not (os.path.exists(os.path.join(PKG, "lib", "libslamhip.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built):
    import slamhip

    if slamhip.device_count() < 1:
        pytest.fail("gpu-marked test but no HIP device is visible (no CPU fallback exists)")
    return slamhip.default_context()
