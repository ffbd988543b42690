import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(autouse=True)
def _multi_stream_program_on_small_cases(monkeypatch):
    """Most tests look at the MULTI-STREAM program (streams, events, zones, slices) on matrices small
    enough for a test -- problems that the library by default replays as one chain of kernel nodes and
    for which it therefore builds the single-stream program.  The suite switches that rule off;
    tests of the rule itself (and the fuzz, every other seed) switch it back on."""
    if "SPLLT_CHAIN_GRAPH_SERIAL" not in os.environ:
        monkeypatch.setenv("SPLLT_CHAIN_GRAPH_SERIAL", "0")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libspllt_hip.so and the oracle once per session if they are missing."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "spllt_amd", "libspllt_hip.so")):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "spllt_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libspllt_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
