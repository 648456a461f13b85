"""unorm8 (raytracer_amd/csrc/rt_intersect.h): the kernels read RGBA8 texels and form Color(UColor) = u8 / 255.0f
(reference src/types.h:272) as x*r corrected by one fma residual step, r = fl(1/255).  The same operation sequence in C
must equal the IEEE quotient for all 256 inputs (exhaustive)."""
import subprocess

SRC = r"""
#include <math.h>
#include <stdio.h>
int main(void) {
    int bad = 0, bad_mul_only = 0;
    const float r = 1.0f / 255.0f;
    for (int i = 0; i < 256; i++) {
        const float x = (float)i, want = x / 255.0f;
        const float a = x * r;
        const float e = fmaf(-a, 255.0f, x);
        const float q = fmaf(e, r, a);
        bad += q != want;
        bad_mul_only += a != want;
    }
    printf("bad=%d mul_only_bad=%d\n", bad, bad_mul_only);
    return bad != 0;
}
"""


def test_u8_over_255_by_multiplication(tmp_path):
    src = tmp_path / "unorm8.c"
    src.write_text(SRC)
    exe = tmp_path / "unorm8"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert out.startswith("bad=0 "), out
    assert "mul_only_bad=0" not in out      # the correction step is needed
