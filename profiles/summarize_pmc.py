"""Per-kernel median / mean of one rocprofv3 --pmc counter (the *_counter_collection.csv of a single-counter pass)."""
import collections
import csv
import statistics
import sys

agg = collections.defaultdict(list)
name = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k].append(float(r["Counter_Value"]))
    name = r["Counter_Name"]
print("# counter %s (raw units as reported: FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE needs x2 on gfx950 for wide coalesced streams)" % name)
print("%-60s %8s %14s %14s %14s" % ("kernel", "calls", "median", "mean", "max"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s %8d %14.3f %14.3f %14.3f" % (k[:60], len(v), statistics.median(v), sum(v) / len(v), max(v)))
