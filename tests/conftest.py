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


# ESC_STAGE_BVH culls triangle meshes with proven bounds (the default path's lists and groups) unless
# ESC_RENDER_BVH_HEURISTIC_PADS / $ESC_BVH_TREE=1 asks for the tree and its heuristic triangle pads.
# Tests that take the `bvh_tree` fixture run both ways.
def pytest_generate_tests(metafunc):
    if "bvh_tree" in metafunc.fixturenames:
        metafunc.parametrize("bvh_tree", [False, True], ids=["proven", "tree"], indirect=True)


@pytest.fixture
def bvh_tree(request, monkeypatch):
    on = bool(getattr(request, "param", False))
    if on:
        monkeypatch.setenv("ESC_BVH_TREE", "1")
    return on
