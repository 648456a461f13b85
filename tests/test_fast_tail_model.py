"""The fast Julia distance tail (raytracer_amd/csrc/rt_intersect.h: julia_distance_tail_fast) as a
CPU model: same operation sequence in C doubles with fma(), hardware seeds emulated by perturbed
float seeds.  Checks, against glibc's evaluation of the reference's expression
(src/objects.cpp:134), that whenever the fast path decides it decides the same float."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_tail_model_agrees_with_the_reference_expression(tmp_path):
    exe = tmp_path / "fast_tail_model"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tools", "fast_tail_model.c"), "-lm"])
    # operands as the march produces them (escaped points, |z|^2 in 1e10..1e20)
    out = subprocess.run([str(exe), "3000000", "march"], capture_output=True, text=True, check=True).stdout
    assert "bad=0" in out, out
    # fallback (exact path) stays rare on such operands
    rate = float(out.split("(")[-1].split(")")[0])
    assert rate < 1e-4, out
    # arbitrary operands, including infinities, zeros and denormals: wrong answers are never allowed
    out = subprocess.run([str(exe), "3000000"], capture_output=True, text=True, check=True).stdout
    assert "bad=0" in out, out
    # the march's lean tail (julia_distance_tail_lean, falling back to the fast one): same bars
    out = subprocess.run([str(exe), "3000000", "march", "lean"], capture_output=True, text=True, check=True).stdout
    assert "bad=0" in out, out
    assert float(out.split("(")[-1].split(")")[0]) < 1e-4, out
    out = subprocess.run([str(exe), "3000000", "any", "lean"], capture_output=True, text=True, check=True).stdout
    assert "bad=0" in out, out
