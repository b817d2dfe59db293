"""rpt_amd/csrc/math_converged.h — the range-converged atan / acos / atan2 / exp / sincos the kernels call — against
include/rpt_math.h (the fdlibm restatements the oracle evaluates), on the host: the header is plain C++, IEEE f64 without
FMA contraction on both sides, so the host comparison says what the device computes (tests/test_gpu_parity.py repeats it
there through rptgpu_eval_math).  Every range boundary of the five functions +- 3 ulps with both signs, the special values,
all pairs of those through atan2, and 5 x 10^6 random arguments per function."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_converged_math_has_the_bits_of_the_fdlibm_restatement(tmp_path):
    exe = str(tmp_path / "math_converged_check")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-o", exe,
                    os.path.join(HERE, "cpp", "math_converged_check.cpp"), "-lm"], check=True)
    out = subprocess.run([exe, "5000000"], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "ok" and int(out[1]) > 25_000_000, out


def test_the_kernels_call_the_converged_forms():
    # no call of the plain restatements left in the kernels' sources (rpt_log has no converged form: its cases share
    # their one division already)
    src = os.path.join(HERE, "..", "rpt_amd", "csrc", "kernels")
    for name in sorted(os.listdir(src)):
        text = open(os.path.join(src, name)).read()
        for fn in ("rpt_atan(", "rpt_atan2(", "rpt_acos(", "rpt_exp(", "rpt_sincos_pio2("):
            assert fn not in text, (name, fn)
