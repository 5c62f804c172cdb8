"""Prints the kernel timeline of ONE whole solve from a rocprofv3 --kernel-trace rocpd database: start offset, duration and the
idle gap in front of every kernel (usage: python tools/timeline.py <dir with *_results.db> [index of the solve])."""
import glob
import sqlite3
import sys

f = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cur = sqlite3.connect(f).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
# a solve starts with the minimal solver's launch (which carries the flatten of a dense frame) or, on the older path, with the flatten
starts = [i for i, r in enumerate(rows) if "minimal9_flatten_kernel" in r[0]]
if len(starts) < k + 2:
    starts = [i for i, r in enumerate(rows) if "flatten_tile_kernel<0>" in r[0] or "flatten_tile_kernel<false>" in r[0]]
if len(starts) < k + 2:
    starts = [i for i, r in enumerate(rows) if "flatten_tile_kernel" in r[0]][::2]
a, b = starts[k], starts[k + 1]
t0 = rows[a][1]
prev_end = t0
tot_busy = 0
for name, s, e in rows[a:b]:
    nm = name.split("(")[0].replace("void ", "").replace("rsdsfm::", "")
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm[:60]))
    tot_busy += e - s
    prev_end = max(prev_end, e)
print("solve span %.1f us, kernels busy %.1f us, idle %.1f us" % ((rows[b][1] - t0) / 1e3, tot_busy / 1e3, (rows[b][1] - t0 - tot_busy) / 1e3))
