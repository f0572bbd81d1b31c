import os
import sys

import pytest
# torch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and the product library is
# linked against /opt/rocm's: whichever is loaded first serves the whole process.  The tests that hand device memory from torch
# to the library (the distributed path does the same: raytracing_folder_amd/dist.py imports torch first) need torch's own,
# so it is loaded before anything else -- whatever subset of the tests is selected.
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library and the
    checker once, exactly as __graft_entry__.build() does.  Where they already exist nothing is rebuilt."""
    import subprocess
    lib = os.path.join(ROOT, "raytracing_folder_amd", "lib", "librt_mi355x.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "raytracing_folder_amd", "csrc")], check=False)
    if not os.path.exists(orc):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=False)


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name))
    return load
