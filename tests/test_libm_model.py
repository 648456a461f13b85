"""raytracer_amd/csrc/rt_libm.h restates glibc 2.35's float asinf / atanf / atan2f (the routines behind
the sphere texture coordinates, src/objects.cpp:47) for the device.  The header also compiles for the
host; here it is compared bit for bit with the host libm on a reduced sweep (every 61st float for the
one-argument functions, 16e6 operand pairs for atan2f).  `tools/libm_model_check 1` is the full sweep
(every float; 1.6e9 pairs), recorded in DESIGN.md."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_libm_restatement_is_bit_exact_against_glibc(tmp_path):
    exe = tmp_path / "libm_model_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", str(exe),
                           os.path.join(ROOT, "tools", "libm_model_check.cpp"), "-lpthread"])
    run = subprocess.run([str(exe), "61", "2000000"], capture_output=True, text=True)
    assert run.returncode == 0 and "-> bad=0" in run.stdout, run.stdout + run.stderr
