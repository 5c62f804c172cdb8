"""The in-range cores of the fp64 square root, reciprocal and division (csrc/device_math.hpp: sqrt_core / rcp_core / div_core) against the
compiler's own expansions of sqrt(x), 1.0 / x and n / d on the GPU: tools/fastmath_check.hip compares the bits over random in-range arguments and the range
bounds.  (ransac_lm_kernel's round 0 and the minimal solver's SVD rely on the equality; arguments outside the range make the RANSAC start over with the
standard functions: tests/test_gpu_ransac.py.)"""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "rs-aware-differential-sfm_amd")


@pytest.mark.gpu
def test_function_cores_equal_the_compilers_expansions_bit_for_bit(tmp_path):
    exe = os.path.join(str(tmp_path), "fastmath_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(PKG, "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "fastmath_check.hip"), "-o", exe])
    p = subprocess.run([exe, "128", "4"], capture_output=True, text=True, timeout=600)  # 4 x (128 M + 2 x 16 M) samples
    assert p.returncode == 0, p.stdout + p.stderr
    assert "sqrt mismatches 0, 1/(1+sqrt) mismatches 0, 1/y mismatches 0, n/d mismatches 0, out-of-domain samples 0" in p.stdout
