"""CPU: the C-ABI library loads and exports every symbol include/rsdsfm.h declares (no compute calls)."""
import os

import pytest


@pytest.mark.parametrize("arith", ["reference", "fused"])
def test_library_exports_every_declared_symbol(rsdsfm, arith):
    """both builds of the library (the product; fused = opt-in) export the whole ABI and say which one they are"""
    lib = rsdsfm.load_library(arith=arith)
    names = rsdsfm.declared_symbols()
    assert len(names) >= 70
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert b"gfx950" in lib.rsdsfm_version()
    assert lib.rsdsfm_fused_arithmetic() == (1 if arith == "fused" else 0)
    v = lib.rsdsfm_version()
    # the string says what the DEFAULT path is (analytic trajectory / radius-factorised refinement, guarded) and what mode 1 selects in this build
    assert b"analytic LM trajectory" in v and b"radius-factorised refinement" in v and b"rsdsfm_set_lm_arithmetic(1)" in v
    assert (b"no fused multiply-add" in v) == (arith == "reference") and (b"FUSED per-pixel model" in v) == (arith == "fused")


@pytest.mark.parametrize("path", ["LIB_PATH", "LIB_PATH_FUSED"])
def test_code_object_is_gfx950_only(rsdsfm, path):
    """The fat binary carries gfx950 code objects only (no multi-arch / compatibility builds)."""
    import re

    rsdsfm.load_library(arith="fused" if path.endswith("FUSED") else "reference")
    blob = open(getattr(rsdsfm, path), "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_library_does_not_need_rccl_to_load(rsdsfm):
    """RCCL is resolved with dlopen on the first rsdsfm_dist_* call: the library itself links neither librccl nor torch"""
    import subprocess

    out = subprocess.run(["readelf", "-d", rsdsfm.LIB_PATH], capture_output=True, text=True).stdout
    needed = [ln.split("[")[1].split("]")[0] for ln in out.splitlines() if "(NEEDED)" in ln]
    assert not any("rccl" in n or "nccl" in n or "torch" in n for n in needed), needed
    assert any("amdhip64" in n for n in needed)


def test_tiled_slab_bounds_is_host_only(rsdsfm):
    """slab geometry of the column-tiled solve (no GPU needed): contiguous, ordered, covering, stride = ceil(cols / ranks)"""
    for cols, n in ((3840, 8), (250, 3), (7, 4), (7, 5), (1, 1), (5, 8)):
        prev_end, per = 0, -(-cols // n)
        for r in range(n):
            c0, sc, stride = rsdsfm.tiled_slab_bounds(cols, n, r)
            assert stride == per and c0 == min(cols, r * per) == prev_end and 0 <= sc <= per
            prev_end = c0 + sc
        assert prev_end == cols
    with pytest.raises(rsdsfm.RsdsfmError):
        rsdsfm.tiled_slab_bounds(10, 2, 2)


def test_tiled_shard_bounds_is_host_only(rsdsfm):
    """shards of the row-tiled depth solve: contiguous, covering, even starts (16-byte aligned 8-byte arrays); the Python driver's
    dist.shard_bounds is the same rule"""
    for n, nr in ((8294400, 8), (30150, 4), (7, 5), (7, 2), (0, 3), (1, 1), (9, 9)):
        bounds, per = rsdsfm.dist.shard_bounds(n, nr)
        prev = 0
        for r in range(nr):
            i0, cnt, stride = rsdsfm.tiled_shard_bounds(n, nr, r)
            assert (i0, i0 + cnt) == bounds[r] and stride == per and stride % 2 == 0 and i0 % 2 == 0 or i0 == n
            assert i0 == prev
            prev = i0 + cnt
        assert prev == n
    with pytest.raises(rsdsfm.RsdsfmError):
        rsdsfm.tiled_shard_bounds(10, 0, 0)


def test_create_fails_loudly_without_gpu(rsdsfm):
    """No CPU fallback: on a box without a HIP device the context cannot be created."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(rsdsfm.RsdsfmError):
        rsdsfm.Solver(0)


def test_null_context_is_rejected(rsdsfm):
    lib = rsdsfm.load_library()
    assert lib.rsdsfm_synchronize(None) < 0
    assert lib.rsdsfm_create(None, 0, None) < 0
