import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(rel):
    return os.path.join(SCENES, rel)


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def hip_ctx():
    """One rt_ctx on cuda:0 for the whole GPU session.  Fails (never skips) if the HIP library or
    the device is missing: a silent fallback would void the parity claims."""
    from raytracer_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
