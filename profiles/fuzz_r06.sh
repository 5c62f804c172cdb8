#!/bin/bash
# the randomised campaigns of round 6 on the final library (run from the repo root through gpurun; logs -> gpurun_out/r06f/, copied to profiles/r06_fuzz_*.log)
OUT=$PWD/gpurun_out/r06f
mkdir -p $OUT
run() { name=$1; shift; ( time timeout 1200 python3 "$@" ) > $OUT/$name.log 2>&1; echo "== $name"; grep -v Warning $OUT/$name.log | grep -v "^  " | tail -${TAILN:-6} | cut -c1-400; }
run fuzz_70602 tests/fuzz_gpu.py ${1:-50000} 70602
run fuzz_70603 tests/fuzz_gpu.py ${1:-50000} 70603
run fuzz_frames_9 tests/fuzz_frames.py ${2:-60000} 9
run fuzz_tiled_61 tests/fuzz_tiled.py ${3:-2000} 61
