"""Prints the sub-records of a bench.py JSON line in a readable form (usage: python tools/show_bench.py <file with the line>)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "| ms_per_step", d["ms_per_step"])
for k, v in d.items():
    if isinstance(v, dict) and k != "config":
        print(k, json.dumps(v)[:700])
