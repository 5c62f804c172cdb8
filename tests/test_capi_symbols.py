"""CPU: the C-ABI library loads and exports every symbol include/rsdsfm.h declares (no compute calls)."""
import os

import pytest


def test_library_exports_every_declared_symbol(rsdsfm):
    lib = rsdsfm.load_library()
    names = rsdsfm.declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert b"gfx950" in lib.rsdsfm_version()


def test_code_object_is_gfx950_only(rsdsfm):
    """The fat binary carries gfx950 code objects only (no multi-arch / compatibility builds)."""
    import re

    rsdsfm.load_library()
    blob = open(rsdsfm.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


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
