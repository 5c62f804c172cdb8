"""CPU: the radius-factorised refinement's kernels must not need SCRATCH memory (a private segment).  A pass with a private segment running on
several streams at once -- the lanes of the sequence solve -- hung the queue and corrupted spilled values on the MI355X boxes (round 6,
tools/seq_determinism_probe.py); the compiler decides about spills, so the build is checked: hipcc -S of the translation unit, every
refine_rf kernel's .private_segment_fixed_size and .vgpr_spill_count must be 0."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_refine_rf_kernels_have_no_private_segment(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    out = tmp_path / "rf.s"
    src = os.path.join(ROOT, "rs-aware-differential-sfm_amd", "csrc", "refine_rf_kernels.hip")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), src, "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    txt = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*refine_rf\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", txt)
    assert len(kernels) >= 12, len(kernels)  # pass <6|7> x <first|later> x <zsum|no>, stage <6|7>, row <6|7>
    bad = [(n, ps, sp) for n, ps, sp in kernels if int(ps) != 0 or int(sp) != 0]
    assert not bad, bad


def test_minimal_solver_kernels_have_no_private_segment(tmp_path):
    """the k estimation's matrices are register arrays indexed by compile-time constants; one run-time index would move them to scratch (and
    the solver shares the GPU with the other lanes of a sequence solve)"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    out = tmp_path / "m9.s"
    src = os.path.join(ROOT, "rs-aware-differential-sfm_amd", "csrc", "minimal9_kernels.hip")
    p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), src, "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    txt = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S*minimal9\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", txt)
    assert len(kernels) == 3, kernels  # minimal9_kernel<false|true>, minimal9_flatten_kernel
    bad = [(n, ps, sp) for n, ps, sp in kernels if int(ps) != 0 or int(sp) != 0]
    assert not bad, bad
