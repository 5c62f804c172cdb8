export TMPDIR=/tmp
REPO=$PWD
run() { x=$1; rm -rf /tmp/pmc_x$x; (cd /tmp && RSDSFM_DEPTH_X=$x rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_x$x -o p --output-format csv -- python3 $REPO/bench.py --workload depth --streams 1 --steps 6 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/pmc_x$x.err); python3 - <<PY
import csv, glob, collections
ctr = collections.defaultdict(lambda: collections.defaultdict(float)); name = {}
for f in glob.glob('/tmp/pmc_x$x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ctr[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value']); name[r['Dispatch_Id']] = r['Kernel_Name']
dur = {}
for f in glob.glob('/tmp/pmc_x$x/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
rows = [(dur[d], c['GRBM_GUI_ACTIVE']) for d, c in ctr.items() if 'depth_lm_batch' in name[d] and 'decide' not in name[d] and d in dur]
rows = rows[len(rows)//4:]
if rows:
    us = sum(r[0] for r in rows) / len(rows); cyc = sum(r[1] for r in rows) / len(rows)
    print('X=$x launches %d  avg %.2f us  GRBM_GUI_ACTIVE %.0f  -> %.3f GHz if the counter sums 8 XCDs, %.3f GHz if 1' % (len(rows), us, cyc, cyc / 8 / us * 1e-3, cyc / us * 1e-3))
else:
    print('X=$x: no rows', len(ctr), len(dur)); print(open('/tmp/pmc_x$x.err').read()[-600:])
PY
}
run 0; run 50; run 51; run 43
