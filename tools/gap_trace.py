"""What fills the idle time between two whole solves: from a rocprofv3 --hip-trace --kernel-trace rocpd database prints, for a few
consecutive solves, the end of the last kernel of solve i, the HIP API calls the host makes until the first kernel of solve i + 1
starts, and that start (all relative to the last kernel's end).
usage: python tools/gap_trace.py <dir with *_results.db> [first solve] [count]"""
import glob
import sqlite3
import sys

f = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 3
con = sqlite3.connect(f)
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "--schema" in sys.argv:
    for n in names:
        cols = [c[1] for c in cur.execute("pragma table_info('%s')" % n)]
        print(n, cols)
    sys.exit(0)
kern = list(cur.execute("select name, start, end from kernels order by start"))
api_view = "regions" if "regions" in names else [n for n in names if "region" in n][0]
cols = [c[1] for c in cur.execute("pragma table_info('%s')" % api_view)]
api = list(cur.execute("select name, start, end from %s order by start" % api_view))
starts = [i for i, r in enumerate(kern) if "minimal9_flatten_kernel" in r[0]]
for k in range(k0, k0 + cnt):
    last = kern[starts[k + 1] - 1]
    nxt = kern[starts[k + 1]]
    t0 = last[2]
    print("solve %d: last kernel %s ends at 0; next solve's first kernel starts at +%.1f us" % (k, last[0].split("(")[0][-28:], (nxt[1] - t0) / 1e3))
    for name, s, e in api:
        if e < t0 - 30000 or s > nxt[1] + 2000:
            continue
        print("   %+8.1f .. %+8.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, name))
