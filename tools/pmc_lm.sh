export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out/r03
run() { name=$1; shift; rm -rf /tmp/pmc_$name; (cd /tmp && rocprofv3 --pmc $@ -d /tmp/pmc_$name -o p --output-format csv -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-side-records --no-cpu-baseline > /dev/null 2>/tmp/pmc_$name.err); python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_$name/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r['Dispatch_Id'], r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])] += float(r['Counter_Value'])
    for (_, k, c), v in per.items():
        acc[k][c].append(v)
for k, d in acc.items():
    if 'ransac_lm_kernel' in k or 'refine_schur' in k or 'minimal9' in k:
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
tail -2 /tmp/pmc_$name.err | cut -c1-200; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS
run c SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run d SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES
