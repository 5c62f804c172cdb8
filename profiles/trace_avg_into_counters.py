"""Adds `trace_avg_us` (and `trace_launches`) of one kernel to a counters.json: the mean duration of its launches in a rocprofv3 --kernel-trace
output directory, counting only launches longer than a threshold (a refinement pass of a solve that has already ended leaves at once).
    python profiles/trace_avg_into_counters.py counters.json <trace dir> "<kernel name as in counters.json>" <min us>"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_to_json import norm  # noqa: E402


def durations(trace_dir):
    out = []
    for db in glob.glob(os.path.join(trace_dir, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, start, end in con.execute("select name, start, end from kernels"):  # (the view profiles/summarize_rocpd.py reads)
            out.append((norm(name), (end - start) / 1e3))
    return out


def main():
    path, trace_dir, kernel, min_us = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    all_ds = [us for n, us in durations(trace_dir) if n == kernel]
    ds = [us for us in all_ds if us > min_us]
    data = json.load(open(path))
    if kernel in data and ds:
        data[kernel]["trace_avg_us"] = sum(ds) / len(ds)
        data[kernel]["trace_launches"] = len(ds)
        # (the counters of this file are means over ALL launches of the kernel; the launches of a solve that has already ended leave at once and
        # count next to nothing: per FULL launch the counters are the stored means / this fraction)
        data[kernel]["trace_full_launch_fraction"] = len(ds) / len(all_ds)
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
        print("%s: %d launches > %.0f us, mean %.2f us" % (kernel, len(ds), min_us, sum(ds) / len(ds)))
    else:
        print("no entry / no launches for", kernel, len(ds))


if __name__ == "__main__":
    main()
