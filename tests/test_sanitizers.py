"""The host code under AddressSanitizer + UBSan (`make asan-host`): tools/sanitize_host.cpp loads every
scene file at several animation times from four threads at once (asset cache, keyframes, OBJ and PNG readers),
renders a small picture of each through the oracle, and round-trips it through the PNG and APNG writers.  A report of
either sanitizer fails the run.  (`make tsan` runs the same program under ThreadSanitizer, `make asan-bvh` the host
BVH builder of librt_hip.so: both were run clean this round; they take too long for this tier.)"""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_loader_png_and_oracle_are_clean_under_asan_and_ubsan():
    run = subprocess.run(["make", "asan-host"], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "0 failed" in run.stdout and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
