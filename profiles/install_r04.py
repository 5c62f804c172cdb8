"""Copies what `profiles/collect_r04.sh` wrote under gpurun_out/r04c/ into the tracked profiles/r04_* files (run from the repo
root): bench lines as they are, kernel-trace summaries behind a header line naming the command, counters.json (stamped)."""
import glob
import os
import shutil

F, P = "gpurun_out/r04c", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    shutil.copy(f, P + "/r04_" + os.path.basename(f))
for f in glob.glob(F + "/trace_*.txt"):
    name = os.path.basename(f)[len("trace_"):-len(".txt")]
    body = [ln for ln in open(f).read().splitlines(True) if not ln.startswith("#")]
    head = "# rocprofv3 --kernel-trace --stats -- python3 bench.py (workload / flags: %s, see profiles/collect_r04.sh); per-kernel durations, round 4\n" % name
    open(P + "/r04_" + os.path.basename(f), "w").write(head + "".join(body))
shutil.copy(F + "/counters.json", P + "/counters.json")
for name in ("timeline_full.txt", "depth_clock_probe.txt", "svd_spread.txt", "refine_slots.txt"):
    if os.path.exists(F + "/" + name):
        shutil.copy(F + "/" + name, P + "/r04_" + name)
print("installed %d bench lines, %d traces" % (len(glob.glob(F + "/bench_*.json")), len(glob.glob(F + "/trace_*.txt"))))
