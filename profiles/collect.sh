#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh            -> gpurun_out/final/{bench_*.json, trace_*.txt, pmc_*/...csv}
# rocprofv3 gets `python3 bench.py ...` directly after `--`; PMC passes are separate runs with one counter each.
set -u
OUT=$PWD/gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
b() { name=$1; shift; python3 bench.py "$@" 2>/dev/null | tail -1 > $OUT/bench_$name.json; echo "bench $name: $(cut -c1-160 $OUT/bench_$name.json)"; }
b depth
b depth_1stream --streams 1 --no-cpu-baseline
b depth_1pair --batch 1 --streams 1 --no-cpu-baseline
b cf --workload depth_closed_form
b full --workload full
b tiled --workload tiled
b tiled_full --workload tiled_full
b rectify --workload rectify
b true_flow --workload true_flow
b metrics --workload metrics
t() { name=$1; shift; rm -rf /tmp/tr_$name; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o p -- python3 $OLDPWD/bench.py "$@" > /dev/null 2>&1); python3 profiles/summarize_rocpd.py /tmp/tr_$name > $OUT/trace_$name.txt; echo "trace $name: $(wc -l < $OUT/trace_$name.txt) lines"; }
t depth_1stream --streams 1 --steps 32 --warmup 2 --no-cpu-baseline
t depth --steps 64 --warmup 3 --no-cpu-baseline
t full --workload full --steps 100 --no-cpu-baseline
t rectify --workload rectify --no-cpu-baseline
t true_flow --workload true_flow --no-cpu-baseline
t metrics --workload metrics --no-cpu-baseline
p() { name=$1; ctr=$2; shift 2; rm -rf /tmp/pmc_${name}_$ctr; (cd /tmp && rocprofv3 --pmc $ctr -d /tmp/pmc_${name}_$ctr -o p --output-format csv -- python3 $OLDPWD/bench.py "$@" > /dev/null 2>&1); mkdir -p $OUT/pmc_$name; f=$(find /tmp/pmc_${name}_$ctr -name '*counter_collection.csv' | head -1); python3 profiles/summarize_pmc.py "$f" > $OUT/pmc_$name/${ctr}.txt; echo "pmc $name $ctr: $(wc -l < $OUT/pmc_$name/${ctr}.txt) kernels"; }
for c in FETCH_SIZE WRITE_SIZE; do
  p depth_batch4 $c --streams 1 --steps 8 --warmup 1 --no-cpu-baseline
  p rectify $c --workload rectify --steps 50 --no-cpu-baseline
  p true_flow $c --workload true_flow --steps 20 --no-cpu-baseline
done
