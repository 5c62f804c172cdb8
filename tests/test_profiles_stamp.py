"""profiles/counters.json is tied to the kernel sources it was collected from (profiles/source_hash.py): bench.py must not price a
kernel with instruction counts of an older build of that kernel."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_stale_sources_are_detected(tmp_path, monkeypatch):
    import source_hash

    now = source_hash.source_hashes()
    assert {"ransac_kernels.hip", "device_math.hpp", "lm_common.hpp", "rsdsfm_internal.hpp", "build.py", "rsdsfm.h"} <= set(now)
    k = "ransac_lm_kernel<true, 3, 2, true>"
    assert source_hash.stale_files(k, now) == []
    assert "ransac_kernels.hip" in source_hash.files_of(k) and "depth_kernels.hip" not in source_hash.files_of(k)
    old = dict(now, **{"ransac_kernels.hip": "0" * 16})
    assert source_hash.stale_files(k, old) == ["ransac_kernels.hip"]
    assert source_hash.stale_files("depth_lm_batch_kernel", old) == []  # another translation unit: not affected
    assert source_hash.stale_files(k, None)  # no stamp at all = stale

    # bench.py's reader: counts when the stamp matches, {"stale": [...]} when it does not
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    for name in ("source_hash.py",):
        (prof / name).write_text(open(os.path.join(ROOT, "profiles", name)).read().replace('os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', repr(ROOT)))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    sys.modules.pop("source_hash", None)
    (prof / "counters.json").write_text(json.dumps({k: {"SQ_INSTS_VALU": 1.0}, "_meta": {"sources": now}}))
    assert b._counters(k) == {"SQ_INSTS_VALU": 1.0}
    (prof / "counters.json").write_text(json.dumps({k: {"SQ_INSTS_VALU": 1.0}, "_meta": {"sources": old}}))
    assert b._counters(k) == {"stale": ["ransac_kernels.hip"]}
    assert b._counters("no_such_kernel") is None
    sys.modules.pop("source_hash", None)


def test_committed_counters_carry_a_stamp_or_are_reported_stale():
    """the committed file either matches the sources (fresh collection) or bench.py says so on its line -- never a silent mismatch"""
    import source_hash

    data = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    stamp = (data.get("_meta") or {}).get("sources")
    stale = source_hash.stale_files("ransac_lma_kernel<2, true>", stamp)
    b = _bench()
    got = b._counters("ransac_lma_kernel<2, true>")
    assert (got == {"stale": stale}) if stale else ("SQ_INSTS_VALU_ADD_F64" in got)
