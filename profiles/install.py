"""Copies what `profiles/collect.sh` wrote under gpurun_out/final/ into the tracked profiles/r01_* files (run from the repo root)."""
import glob
import os
import shutil

F, P = "gpurun_out/final", "profiles"
for f in glob.glob(F + "/bench_*.json"):
    shutil.copy(f, P + "/r01_" + os.path.basename(f))
head = """# rocprofv3 --kernel-trace --stats, MI355X, round 1 FINAL code (collected by profiles/collect.sh, summarised from the rocpd output by profiles/summarize_rocpd.py; microseconds)
# depth_lm_batch_kernel = launch 0 of the batched fast path: the speculative streaming pass over 4 independent pairs per launch
#   (the dominant kernel of the headline workload); depth_lm_decide_apply_batch_kernel = its follow-up (decision + apply).
# depth_lm_kernel<1> / depth_lm_decide_apply_kernel = the same for one pair per launch (the run's one-pair-at-a-time reference loop).
# The other kernels in sections (1) and (2) belong to bench.py's side measurements (whole solves: full_solve / full_solve_batched).
# Earlier summaries: profiles/r01_history/.
"""
sections = [("depth_1stream", "## (1) ROOFLINE REFERENCE -- batches on ONE stream:\n## cd /tmp && rocprofv3 --kernel-trace --stats -d ... -- python3 bench.py --streams 1 --steps 32 --warmup 2 --no-cpu-baseline\n"),
            ("depth", "\n## (2) DEFAULT MODE (4 pairs per launch on each of 2 streams):\n## cd /tmp && rocprofv3 --kernel-trace --stats -d ... -- python3 bench.py --steps 64 --warmup 3 --no-cpu-baseline\n"),
            ("full", "\n## (3) WHOLE SOLVE: cd /tmp && rocprofv3 --kernel-trace --stats -d ... -- python3 bench.py --workload full --steps 100 --no-cpu-baseline\n")]
with open(P + "/r01_kernel_trace_final.txt", "w") as out:
    out.write(head)
    for name, title in sections:
        out.write(title + open("%s/trace_%s.txt" % (F, name)).read())
for n in ("rectify", "true_flow", "metrics"):
    with open("%s/r01_%s_kernel_trace.txt" % (P, n), "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats of: python3 bench.py --workload %s --no-cpu-baseline (profiles/collect.sh; microseconds)\n" % n)
        out.write(open("%s/trace_%s.txt" % (F, n)).read())
for n in ("depth_batch4", "rectify", "true_flow"):
    os.makedirs("%s/r01_pmc_%s" % (P, n), exist_ok=True)
    for f in glob.glob("%s/pmc_%s/*.txt" % (F, n)):
        shutil.copy(f, "%s/r01_pmc_%s/" % (P, n))
print("installed; update profiles/traffic.json from the PMC medians if they changed")
