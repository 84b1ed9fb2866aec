import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (*.so are git-ignored): build what is missing once.
    hipcc cross-compiles gfx950 without a GPU; the oracle needs only gcc."""
    import subprocess
    lib = os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "lib", "libgsrast.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "taichi_3d_gaussian_splatting_amd", "csrc"), "-j4"],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libgsoracle.so")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_torch_helpers.json")) as fh:
        return json.load(fh)
