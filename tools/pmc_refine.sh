#!/bin/bash
# PMC counters of the refinement's pass kernel inside ordinary whole solves (separate rocprofv3 passes, counters only):
#   bash tools/pmc_refine.sh   -> gpurun_out/pmc_refine/*.txt
set -u
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_refine; mkdir -p $OUT
run() { name=$1; shift; ctrs=$1; shift; rm -rf /tmp/pmc_$name; (cd /tmp && rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o p --output-format csv -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-side-records --no-cpu-baseline > /dev/null 2>&1); python3 - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
rows = []
for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % name, recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get('Kernel_Name', '')
    if 'refine_rf_pass' in k or 'ransac_lma_kernel' in k:
        kk = 'rf_pass<%s>' % k.split('refine_rf_pass_kernel')[1][:12] if 'refine_rf_pass' in k else 'ransac_lma'
        agg[kk][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    print(k, {c: (len(v), sum(v) / len(v), max(v)) for c, v in d.items()})
PY
}
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" > $OUT/a.txt 2>&1
run b "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" > $OUT/b.txt 2>&1
run c "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" > $OUT/c.txt 2>&1
cat $OUT/a.txt $OUT/b.txt $OUT/c.txt
