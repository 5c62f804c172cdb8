export TMPDIR=/tmp
REPO=$PWD
rm -rf /tmp/tr_full; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/tr_full -o p -- python3 $REPO/bench.py --steps 100 --no-side-records --no-cpu-baseline > /dev/null 2>&1)
python3 tools/timeline.py /tmp/tr_full 40 > gpurun_out/timeline_slots.txt
