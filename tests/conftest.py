import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libspllt_hip.so and the oracle once per session if they are missing."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "spllt_amd", "libspllt_hip.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "spllt_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libspllt_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
