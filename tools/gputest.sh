export RSDSFM_TEST_BARRIER_TIMEOUT=8
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 200 -p no:cacheprovider > gpurun_out/gputest.log 2>&1
grep -E "FAILED|ERROR|passed|failed" gpurun_out/gputest.log | tail -40
