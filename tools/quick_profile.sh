set -u
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/q1; mkdir -p $OUT
python3 bench.py --steps 40 --warmup 5 --no-side-records --no-cpu-baseline 2>$OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json
rm -rf /tmp/tr_full; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/tr_full -o p -- python3 $REPO/bench.py --steps 60 --no-side-records --no-cpu-baseline > /dev/null 2>&1)
python3 profiles/summarize_rocpd.py /tmp/tr_full > $OUT/trace_full.txt; head -30 $OUT/trace_full.txt
python3 tools/timeline.py /tmp/tr_full 40 > $OUT/timeline_full.txt; cat $OUT/timeline_full.txt
