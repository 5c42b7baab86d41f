"""convert.c:131-132 scales a UC8 sample as ((float)v - 127.5f) / 127.5f.  The kernels that take raw UC8 samples through the DC block
(readsb_amd/csrc/kernels.hip: uc8_scaled) avoid the division: product with the rounded reciprocal, one fused correction step.  That this
is the correctly rounded quotient for every one of the 256 sample values is checked here, on the host, with the C library's fmaf."""
import os
import subprocess

SRC = r"""
#include <math.h>
#include <stdio.h>
int main(void) {
    const float c = 1.0f / 127.5f;
    int bad = 0;
    for (int v = 0; v < 256; ++v) {
        const float n = (float)v - 127.5f;
        const float want = n / 127.5f;
        float q = n * c;
        const float r = fmaf(-q, 127.5f, n);
        q = fmaf(r, c, q);
        if (q != want) ++bad;
    }
    printf("%d\n", bad);
    return 0;
}
"""


def test_division_free_uc8_scaling_is_exact(tmp_path):
    src = tmp_path / "uc8div.c"
    src.write_text(SRC)
    exe = tmp_path / "uc8div"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    assert subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip() == "0"
