"""The C++ host mirror (include/rpt.hpp) drives the same C ABI: the reference's example programs
transcribed to C++ (examples/*.cpp) must produce, on the GPU, exactly the frame the Python host
mirror and the oracle produce for the same scene, size, seed and sample count."""
import os
import subprocess

import numpy as np
import pytest

from test_golden import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "examples", "build")


def ensure_built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])


def test_examples_build_and_fail_loudly_without_gpu(tmp_path, gpu_available):
    ensure_built()
    if gpu_available:
        pytest.skip("GPU present")
    r = subprocess.run([os.path.join(BUILD, "sphere"), "8", "8", "1", "1", "1", str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,bounces,spp,seed", [("sphere", 2, 8, 101), ("cornell", 8, 8, 102)])
def test_cpp_example_equals_golden(tmp_path, name, bounces, spp, seed):
    ensure_built()
    prefix = str(tmp_path / name)
    r = subprocess.run([os.path.join(BUILD, name), "64", "36", str(bounces), str(spp), str(seed), prefix],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Msamples/s" in r.stdout
    img = np.fromfile(prefix + ".f64", dtype=np.float64).reshape(-1, 3)
    assert (img == load(name)["image"]).all()
    with open(prefix + ".ppm", "rb") as f:
        assert f.readline() == b"P6\n"
