#!/usr/bin/env python3
"""Instruction counts per basic block of one kernel in a hipcc -S listing:  python tools/isa_loop_count.py file.s <substring of the kernel name> [print]"""
import re
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(s) if re.match(r"^_Z\w+:", l) and key in l.split(":")[0]][0]
end = [i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end")][0]
lines = [l.strip() for l in s[start + 1:end] if l.strip() and not l.strip().startswith(";")]
labels = [(i, l) for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)]
for (i, l), (j, _) in zip(labels, labels[1:] + [(len(lines), "")]):
    ins = [x for x in lines[i + 1:j] if not x.startswith(".")]
    print("%-10s instr %4d  valu %4d  f64 %4d  %s" % (l.split(":")[0], len(ins), sum(x.startswith("v_") for x in ins), sum("f64" in x.split()[0] for x in ins), "loop" if "Loop" in l else ""))
    if len(sys.argv) > 3 and "Loop" in l:
        print("\n".join("    " + x for x in ins))
