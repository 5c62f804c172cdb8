"""Copies what `profiles/collect_r05.sh` wrote under gpurun_out/r05c/ into the tracked profiles/r05_* files (run from the repo
root): bench lines as they are, kernel-trace summaries behind a header line naming the command, counters.json (stamped)."""
import glob
import os
import shutil

F, P = "gpurun_out/r05c", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    shutil.copy(f, P + "/r05_" + os.path.basename(f))
for f in glob.glob(F + "/trace_*.txt"):
    name = os.path.basename(f)[len("trace_"):-len(".txt")]
    body = [ln for ln in open(f).read().splitlines(True) if not ln.startswith("#")]
    head = "# rocprofv3 --kernel-trace --stats -- python3 bench.py (workload / flags: %s, see profiles/collect_r05.sh); per-kernel durations, round 5\n" % name
    open(P + "/r05_" + os.path.basename(f), "w").write(head + "".join(body))
shutil.copy(F + "/counters.json", P + "/counters.json")
for name in ("timeline_full.txt", "lma_time.txt", "seq_sweep.txt", "solve_times.txt"):
    if os.path.exists(F + "/" + name):
        shutil.copy(F + "/" + name, P + "/r05_" + name)
print("installed %d bench lines, %d traces" % (len(glob.glob(F + "/bench_*.json")), len(glob.glob(F + "/trace_*.txt"))))
