#!/bin/bash
# the randomised campaigns of round 6 on the final library (run from the repo root through gpurun; logs -> gpurun_out/r06f/, copied to profiles/r06_fuzz_*.log)
OUT=$PWD/gpurun_out/r06f
mkdir -p $OUT
run() { name=$1; shift; ( time timeout 1200 python3 "$@" ) > $OUT/$name.log 2>&1; echo "== $name"; grep -v Warning $OUT/$name.log | grep -v "^  " | tail -${TAILN:-6} | cut -c1-400; }
# seeds: SEEDS="<fuzz_gpu a> <fuzz_gpu b> <fuzz_frames> <fuzz_tiled>" (first set 70602 70603 9 61, second set 70604 70605 10 62)
set -- ${1:-50000} ${2:-60000} ${3:-2500} ${SEEDS:-70602 70603 9 61}
run fuzz_$4 tests/fuzz_gpu.py $1 $4
run fuzz_$5 tests/fuzz_gpu.py $1 $5
run fuzz_frames_$6 tests/fuzz_frames.py $2 $6
run fuzz_tiled_$7 tests/fuzz_tiled.py $3 $7
