"""raytracer_amd/csrc/rt_divide.h: division by a divisor that is fixed for a launch, as a high multiply and two shifts
(start_sample takes a sample id apart with it).  The header compiles for the host; tools/divide_check.cpp compares it
with the hardware division for 6 238 divisors (every one up to 4 096, the neighbours of every power of two, 2^32 - 1,
pseudo-random ones) against 79 million numerators."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_division_by_multiplication_is_exact(tmp_path):
    exe = tmp_path / "divide_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "divide_check.cpp")])
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and "-> bad=0" in run.stdout, run.stdout + run.stderr
