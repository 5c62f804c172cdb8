"""Copies what `profiles/collect_r02.sh` wrote under gpurun_out/r02/ into the tracked profiles/r02_* files (run from the repo
root): bench lines as they are, kernel-trace summaries behind the two header lines of the existing files, counters.json."""
import glob
import os
import shutil

F, P = "gpurun_out/r02", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    shutil.copy(f, P + "/r02_" + os.path.basename(f))
for f in glob.glob(F + "/trace_*.txt"):
    dst = P + "/r02_" + os.path.basename(f)
    head = [ln for ln in open(dst).read().splitlines(True)[:2] if ln.startswith("#")] if os.path.exists(dst) else []
    body = [ln for ln in open(f).read().splitlines(True) if not ln.startswith("#")]
    open(dst, "w").write("".join(head + body))
shutil.copy(F + "/counters.json", P + "/counters.json")
print("installed %d bench lines, %d traces" % (len(glob.glob(F + "/bench_*.json")), len(glob.glob(F + "/trace_*.txt"))))
