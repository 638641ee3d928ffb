import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# a fresh checkout has no built artefacts (they are git-ignored): build them once
if not os.path.exists(os.path.join(ROOT, "ch-bin_amd", "libchbin_hip.so")):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "ch-bin_amd", "csrc"), "-j4", "-s"], check=False)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
