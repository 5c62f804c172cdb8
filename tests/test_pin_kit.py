"""The pinning kit (tools/pin_reference) cannot meet the reference in this image; what CAN be checked here is its plumbing: the
exporter, the .pin container, the importer's layout and every comparison of tests/test_reference_golden.py, run against a stand-in
fixture built from the oracle (tools/pin_reference/selfcheck.py) -- on the CPU for the oracle's half, on the GPU for the HIP half."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFCHECK = os.path.join(ROOT, "tools", "pin_reference", "selfcheck.py")


def _run(marker, expect):
    p = subprocess.run([sys.executable, SELFCHECK, "-m", marker], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "plumbing ok" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]
    assert "%d passed" % expect in p.stdout, p.stdout[-1500:]  # none skipped


def test_pin_kit_plumbing_oracle_half():
    _run("not gpu", 30)  # 5 cases x (minimal solver, depth solve, RANSAC, 2 refinements, the first LM step in ulps)


@pytest.mark.gpu
def test_pin_kit_plumbing_hip_half():
    """the HIP path against the ORACLE's outputs in the reference fixture's layout: the parity the GPU suite asserts everywhere, through
    the code that will meet the real fixture"""
    _run("gpu", 50)  # 5 cases x (minimal solver, depth solve, RANSAC, 2 refinements) x 2 arithmetics of the HIP path


def test_pin_kit_inputs_export(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_reference", "export_inputs.py"), str(tmp_path)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tools", "pin_reference"))
    import numpy as np
    import pinio

    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))
    a = pinio.read(str(tmp_path / "deepflow_k0.pin"))
    assert np.array_equal(a["q"], g["deepflow_k0/q"]) and np.array_equal(a["samples"], g["deepflow_k0/samples"]) and a["samples"].dtype == np.int32
    assert a["tolerance"][0] == 0.05 and a["use_k"][0] == 0


def test_mirror_check_compiles(tmp_path):
    """tools/pin_reference/mirror_check.cpp -- the reference's six call sites (main.cc:437-457, errorMeasure.cpp:104-152) written with the
    reference's own argument expressions -- must at least be WELL-FORMED against the mirror's headers.  The image has no Eigen, so the
    mirror's stand-in types answer to the name `Eigen` here (tests/cpp/eigen_shim): a compile check of host/*.h, not a pin."""
    src = os.path.join(ROOT, "tools", "pin_reference", "mirror_check.cpp")
    shim = os.path.join(ROOT, "tests", "cpp", "eigen_shim")
    p = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", shim, src], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-4000:]
    # both of the reference's RansacValues constructors (minimal.h:67-75) and its two Velocities constructors (minimal.h:45-50) exist
    probe = tmp_path / "ctors.cpp"
    probe.write_text(
        '#include <Eigen/Dense>\n#include "%s/rs-aware-differential-sfm_amd/host/minimal.h"\n'
        "int main() {\n"
        "  Eigen::Array3Xd inl(3, 4); Eigen::VectorXd b(4); Eigen::Vector3d w, v;\n"
        "  RansacValues r5(4, inl, b, w, v); RansacValues r7(4, inl, b, b, w, v, 0.5);\n"
        "  Velocities v2(w, v); Velocities v3(w, v, 0.25);\n"
        "  return (r5.k == 0 && r5.alpha_k.size() == 4 && r7.k == 0.5 && v2.k == 0 && v3.k == 0.25) ? 0 : 1;\n}\n" % ROOT
    )
    p = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", shim, str(probe)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-4000:]
