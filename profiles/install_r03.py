"""Copies what `profiles/collect_r03.sh` wrote under gpurun_out/r03c/ into the tracked profiles/r03_* files (run from the repo
root): bench lines as they are, kernel-trace summaries behind a header line naming the command, counters.json (stamped)."""
import glob
import os
import shutil

F, P = "gpurun_out/r03c", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    shutil.copy(f, P + "/r03_" + os.path.basename(f))
for f in glob.glob(F + "/trace_*.txt"):
    name = os.path.basename(f)[len("trace_"):-len(".txt")]
    body = [ln for ln in open(f).read().splitlines(True) if not ln.startswith("#")]
    head = "# rocprofv3 --kernel-trace --stats -- python3 bench.py (workload / flags: %s, see profiles/collect_r03.sh); per-kernel durations, round 3\n" % name
    open(P + "/r03_" + os.path.basename(f), "w").write(head + "".join(body))
shutil.copy(F + "/counters.json", P + "/counters.json")
print("installed %d bench lines, %d traces" % (len(glob.glob(F + "/bench_*.json")), len(glob.glob(F + "/trace_*.txt"))))
