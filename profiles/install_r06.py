"""Copies what `profiles/collect_r06.sh` wrote under gpurun_out/r06c/ into the tracked profiles/r06_* files (run from the repo
root): bench lines as they are, kernel-trace summaries behind a header line naming the command, counters.json (stamped), the tools' text outputs."""
import glob
import os
import shutil

F, P = "gpurun_out/r06c", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    if os.path.getsize(f) > 0:
        shutil.copy(f, P + "/r06_" + os.path.basename(f))
for f in glob.glob(F + "/trace_*.txt"):
    name = os.path.basename(f)[len("trace_"):-len(".txt")]
    body = [ln for ln in open(f).read().splitlines(True) if not ln.startswith("#")]
    if len(body) < 2:
        continue
    head = "# rocprofv3 --kernel-trace --stats -- python3 <command of `%s` in profiles/collect_r06.sh>; per-kernel durations, round 6\n" % name
    open(P + "/r06_" + os.path.basename(f), "w").write(head + "".join(body))
if os.path.exists(F + "/counters.json"):
    shutil.copy(F + "/counters.json", P + "/counters.json")
for name in ("timeline_full.txt", "lma_time.txt", "lma_T_sweep.txt", "refine_phases.txt", "refine_phases_loads_only.txt", "refine_slots.txt", "xfer_probe.txt", "host_boundary_probe.txt", "host_boundary_phases.txt", "k_sections.txt", "accel_solves.txt",
             "seq_sweep.txt", "solve_times.txt", "seq_determinism.txt", "soak.txt"):
    if os.path.exists(F + "/" + name) and os.path.getsize(F + "/" + name) > 0:
        shutil.copy(F + "/" + name, P + "/r06_" + name)
print("installed %d bench lines, %d traces" % (len(glob.glob(F + "/bench_*.json")), len(glob.glob(F + "/trace_*.txt"))))
