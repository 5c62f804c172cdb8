#!/bin/bash
# Regenerates the raw material of profiles/r06_* on a GPU box (run from the repo root through gpurun), ONE pass at the end of the round:
#   bash profiles/collect_r06.sh [quick]   -> gpurun_out/r06c/{bench_*.json, trace_*.txt, counters.json, ...};  python profiles/install_r06.py
# rocprofv3 gets `python3 bench.py ...` / `python3 tools/...` directly after `--`; PMC passes are separate runs (counters only).  Every step
# runs under `timeout`.  counters.json is stamped with the hashes of the kernel sources (profiles/source_hash.py): bench.py refuses stale counts.
set -u
OUT=$PWD/gpurun_out/r06c
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
b() { name=$1; shift; timeout 600 python3 bench.py "$@" 2>$OUT/bench_$name.err | tail -1 > $OUT/bench_$name.json; echo "bench $name: $(cut -c1-200 $OUT/bench_$name.json)"; }
t() { name=$1; shift; rm -rf /tmp/tr_$name; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$name -o p -- python3 $REPO/"$@" > /dev/null 2>&1); python3 profiles/summarize_rocpd.py /tmp/tr_$name > $OUT/trace_$name.txt; echo "trace $name: $(wc -l < $OUT/trace_$name.txt) lines"; }
p() { name=$1; shift; ctrs=$1; shift; rm -rf /tmp/pmc_$name; (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o p --output-format csv -- python3 $REPO/bench.py "$@" > /dev/null 2>&1); }
PROF="--no-side-records --no-cpu-baseline"
rm -f $OUT/counters.json
# counters first (bench.py reads profiles/counters.json for its roofline / refine_pass records)
p full_insts "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU" --steps 6 --warmup 2 $PROF
p full_fetch FETCH_SIZE --steps 6 --warmup 2 $PROF
p full_write WRITE_SIZE --steps 6 --warmup 2 $PROF
python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_full_insts /tmp/pmc_full_fetch /tmp/pmc_full_write
for ctr in FETCH_SIZE WRITE_SIZE; do
  p depth_$ctr $ctr --workload depth --streams 1 --steps 6 --warmup 1 --no-cpu-baseline
  python3 profiles/pmc_to_json.py $OUT/counters.json /tmp/pmc_depth_$ctr
done
t full bench.py --steps 100 $PROF
python3 tools/timeline.py /tmp/tr_full 40 > $OUT/timeline_full.txt
# the launch durations of the refinement's pass (the launches that ran their loop: > 12 us) into counters.json beside its counters
python3 profiles/trace_avg_into_counters.py $OUT/counters.json /tmp/tr_full "refine_rf_pass_kernel<6, false, false>" 12.0
cp $OUT/counters.json profiles/counters.json
b full
timeout 300 python3 tools/lma_time.py 60 > $OUT/lma_time.txt 2>/dev/null
timeout 400 python3 tools/lma_time.py sweep 30 > $OUT/lma_T_sweep.txt 2>/dev/null
RSDSFM_RF_STAMPS=1 timeout 200 python3 tools/refine_rf_phases.py > $OUT/refine_phases.txt 2>/dev/null
SLOTS_ACCEL=1 RSDSFM_RF_STAMPS=1 timeout 200 python3 tools/refine_rf_phases.py >> $OUT/refine_phases.txt 2>/dev/null
# the same stamps from the timing build whose loop performs its memory accesses only (tools/build_rfproxy.sh, made before the call; wrong results by construction)
if [ -f rs-aware-differential-sfm_amd/librsdsfm_hip_rfproxy.so ]; then (echo "# RF_LOADS_ONLY=1 timing build: the NP = 6 loop's loads and store without its arithmetic (every solve is sent back by a guard; read the loop phase)"; RSDSFM_LIB=$PWD/rs-aware-differential-sfm_amd/librsdsfm_hip_rfproxy.so RSDSFM_RF_STAMPS=1 timeout 200 python3 tools/refine_rf_phases.py 2>/dev/null) > $OUT/refine_phases_loads_only.txt; fi
timeout 300 python3 tools/refine_slots.py > $OUT/refine_slots.txt 2>/dev/null
SLOTS_ACCEL=1 timeout 300 python3 tools/refine_slots.py >> $OUT/refine_slots.txt 2>/dev/null
timeout 120 ./tools/xfer_probe > $OUT/xfer_probe.txt 2>&1
timeout 120 python3 tools/host_boundary_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/host_boundary_probe.txt
RSDSFM_XFER_TRACE=1 timeout 120 python3 tools/host_boundary_probe.py 40 2>&1 | python3 tools/xfer_trace_stats.py > $OUT/host_boundary_phases.txt
# the k estimation's sections (a diagnostic build of the library: tools/build_ksec.sh, made before the call)
if [ -f rs-aware-differential-sfm_amd/librsdsfm_hip_ksec.so ]; then RSDSFM_LIB=$PWD/rs-aware-differential-sfm_amd/librsdsfm_hip_ksec.so timeout 200 python3 tools/k_sections.py 40 2>/dev/null > $OUT/k_sections.txt; fi
t accel tools/accel_solves.py 30
timeout 120 python3 tools/accel_solves.py 60 > $OUT/accel_solves.txt 2>/dev/null
if [ "${1:-}" != "quick" ]; then
  t sequence bench.py --steps 3 --warmup 1 --no-cpu-baseline --sequence-only
  b full_fused --arith fused --no-side-records
  b depth --workload depth
  b depth_batch4 --workload depth --batch 4
  b tiled_full --workload tiled_full
  b tiled --workload tiled
  b metrics --workload metrics
  b rectify --workload rectify
  b true_flow --workload true_flow
  t depth bench.py --workload depth --streams 1 --steps 60 --no-cpu-baseline
  t tiled_full bench.py --workload tiled_full --steps 20
  for n in 2 4 8; do RSDSFM_SHARE_GPU=1 timeout 600 python3 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline 2>$OUT/bench_shared_gpu_rccl$n.err | tail -1 > $OUT/bench_shared_gpu_rccl$n.json; echo "shared $n: $(cut -c1-160 $OUT/bench_shared_gpu_rccl$n.json)"; done
  timeout 400 python3 tools/seq_sweep.py > $OUT/seq_sweep.txt 2>/dev/null
  timeout 400 python3 tools/solve_times.py > $OUT/solve_times.txt 2>/dev/null
  for i in 1 2 3; do timeout 120 python3 tools/seq_determinism_probe.py 0 3 10 2>/dev/null | tail -2; done > $OUT/seq_determinism.txt
  (timeout 600 python3 tools/soak.py 2>&1 | grep -v amdgpu.ids; timeout 200 python3 tools/leak_check.py 2>&1 | grep -v amdgpu.ids | tail -3) > $OUT/soak.txt
fi
ls $OUT
