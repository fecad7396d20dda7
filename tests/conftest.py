import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # always ask make: a no-op when the binaries are newer than their sources (the prebuilt .so
    # files travel to the GPU box with their mtimes), a rebuild after an edit -- tests must never
    # run against a stale library.  xdist workers skip it (the controller already did it).
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return
    subprocess.run(["make", "-s", "-C", ROOT, "lib", "viewer"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
