"""Merges rocprofv3 --pmc passes into profiles/counters.json: per kernel, the MEAN per-launch value of every collected counter.

    python profiles/pmc_to_json.py profiles/counters.json [--suffix :fused] <dir-or-csv> [<dir-or-csv> ...]

Each argument is a *_counter_collection.csv (or a directory searched for them).  Kernel names are normalised (no `void`, no
namespace, no parameter list; template arguments kept), e.g. `ransac_lm_kernel<true>`.  Units as rocprofv3 reports them
(FETCH_SIZE / WRITE_SIZE in KB; SQ_INSTS_* in wave-instructions).  bench.py reads this file for `roofline`."""
import collections
import csv
import glob
import json
import os
import re
import sys


def norm(name):
    name = name.strip().strip('"').replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)  # parameter list
    name = name.replace("void ", "").replace("rsdsfm::", "")
    return name.strip()


def main():
    out_path = sys.argv[1]
    args = sys.argv[2:]
    suffix = ""
    if args and args[0] == "--suffix":
        suffix, args = args[1], args[2:]
    files = []
    for a in args:
        files += [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        per_dispatch = collections.defaultdict(float)  # (dispatch id, kernel, counter) -> sum over the dimensions reported
        for r in csv.DictReader(open(f)):
            key = (r.get("Dispatch_Id"), norm(r["Kernel_Name"]), r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
        for (_, k, cn), v in per_dispatch.items():
            acc[k][cn].append(v)
    res = {}
    if os.path.exists(out_path):
        try:
            res = json.load(open(out_path))
        except Exception:
            res = {}
    for k, d in acc.items():
        if k.startswith("__amd") or k.startswith("at::") or "Cijk" in k:  # runtime fill / copy kernels and torch's own kernels
            continue
        e = res.setdefault(k + suffix, {})
        for cn, v in d.items():
            e[cn] = sum(v) / len(v)
        e["launches_sampled"] = max(len(v) for v in d.values())
    # stamp: the sources these counts belong to (bench.py refuses to price a kernel whose sources have changed since)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import source_hash

    res["_meta"] = {"sources": source_hash.source_hashes(), "note": "sha256[:16] of the sources at collection time; see profiles/source_hash.py"}
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote %s: %d kernels" % (out_path, len(res)))


if __name__ == "__main__":
    main()
