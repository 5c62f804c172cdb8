export TMPDIR=/tmp
REPO=$PWD; OUT=$PWD/gpurun_out/c1; mkdir -p $OUT
timeout 300 python3 tools/lma_time.py sweep 30 > $OUT/lma_T_sweep.txt 2>&1; cat $OUT/lma_T_sweep.txt | cut -c1-260
timeout 120 python3 tools/accel_solves.py 40 2>&1 | tail -2
rm -rf /tmp/tr_accel; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/tr_accel -o p -- python3 $REPO/tools/accel_solves.py 30 > /dev/null 2>&1)
python3 profiles/summarize_rocpd.py /tmp/tr_accel > $OUT/trace_accel.txt; head -24 $OUT/trace_accel.txt | cut -c1-200
