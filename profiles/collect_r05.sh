#!/bin/bash
# Regenerates the raw material of profiles/r05_* on a GPU box (run from the repo root through gpurun), ONE pass at the end of the round:
#   bash profiles/collect_r05.sh [quick]   -> gpurun_out/r05c/{bench_*.json, trace_*.txt, counters.json, ...};  python profiles/install_r05.py
# rocprofv3 gets `python3 bench.py ...` directly after `--`; PMC passes are separate runs (kernel-trace / stats only elsewhere).
# counters.json is stamped with the hashes of the kernel sources (profiles/source_hash.py): bench.py refuses stale counts.
set -u
OUT=$PWD/gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
b() { name=$1; shift; python3 bench.py "$@" 2>$OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; echo "bench $name: $(cut -c1-200 $OUT/bench_$name.json)"; }
t() { name=$1; shift; rm -rf /tmp/tr_$name; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o p -- python3 $REPO/bench.py "$@" > /dev/null 2>&1); python3 profiles/summarize_rocpd.py /tmp/tr_$name > $OUT/trace_$name.txt; echo "trace $name: $(wc -l < $OUT/trace_$name.txt) lines"; }
p() { name=$1; shift; ctrs=$1; shift; rm -rf /tmp/pmc_$name; (cd /tmp && rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o p --output-format csv -- python3 $REPO/bench.py "$@" > /dev/null 2>&1); }
PROF="--no-side-records --no-cpu-baseline"
rm -f $OUT/counters.json
# counters first (bench.py reads profiles/counters.json for its roofline records)
p full_insts "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU" --steps 6 --warmup 2 $PROF
p full_fetch FETCH_SIZE --steps 6 --warmup 2 $PROF
p full_write WRITE_SIZE --steps 6 --warmup 2 $PROF
python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_full_insts /tmp/pmc_full_fetch /tmp/pmc_full_write
for ctr in FETCH_SIZE WRITE_SIZE; do
  p depth_$ctr $ctr --workload depth --streams 1 --steps 6 --warmup 1 --no-cpu-baseline
  python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_depth_$ctr
done
if [ "${1:-}" != "quick" ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    p rect_$ctr $ctr --workload rectify --steps 30 --no-cpu-baseline
    p tflow_$ctr $ctr --workload true_flow --steps 10 --no-cpu-baseline
    python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_rect_$ctr /tmp/pmc_tflow_$ctr
  done
  p lma_occ "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU" --steps 6 --warmup 2 $PROF
  python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_lma_occ
fi
cp $OUT/counters.json profiles/counters.json
t full --steps 100 $PROF
python3 tools/timeline.py /tmp/tr_full 40 > $OUT/timeline_full.txt
b full
python3 tools/lma_time.py 60 > $OUT/lma_time.txt 2>/dev/null
if [ "${1:-}" != "quick" ]; then
  t sequence --steps 3 --warmup 1 --no-cpu-baseline --sequence-only
  b full_fused --arith fused --no-side-records
  b depth --workload depth
  b depth_batch4 --workload depth --batch 4
  b tiled_full --workload tiled_full
  b tiled --workload tiled
  b metrics --workload metrics
  b rectify --workload rectify
  b true_flow --workload true_flow
  t depth --workload depth --streams 1 --steps 60 --no-cpu-baseline
  t tiled_full --workload tiled_full --steps 20
  for n in 2 4 8; do RSDSFM_SHARE_GPU=1 python3 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_shared_gpu_rccl$n.err | tail -1 > $OUT/bench_shared_gpu_rccl$n.json; echo "shared $n: $(cut -c1-160 $OUT/bench_shared_gpu_rccl$n.json)"; done
  python3 tools/seq_sweep.py > $OUT/seq_sweep.txt 2>/dev/null
  python3 tools/solve_times.py > $OUT/solve_times.txt 2>/dev/null
fi
