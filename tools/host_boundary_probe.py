import sys, time, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0,'.')
import numpy as np, rsdsfm
d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
q,u,a,ak = d["q"],d["u"],d["alpha"],d["alpha_k"]
n=len(q)
s = rsdsfm.Solver(0)
outputs={}; ref_out=np.empty((n,3))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    t0=time.perf_counter()
    r = s.ransac(q,u,a,ak,False,50,0.05,seed=11+i,outputs=outputs)
    t1=time.perf_counter()
    out = s.non_linear_refinement(u,r["inliers"],r["alpha"],r["alpha_k"],r["v"],r["w"],r["k"],False,tag=r["tag"],out=ref_out[:r["num_inliers"]])
    t2=time.perf_counter()
    print(i, "ransac %.2f ms refine %.2f ms hits %d" % ((t1-t0)*1e3,(t2-t1)*1e3, s.refine_cache_hits()), flush=True)
for i in range(3):
    t0=time.perf_counter()
    r = s.ransac(q,u,a,ak,False,50,0.05,seed=11+i)
    t1=time.perf_counter()
    out = s.non_linear_refinement(u,r["inliers"],r["alpha"],r["alpha_k"],r["v"],r["w"],r["k"],False)
    t2=time.perf_counter()
    print("fresh", i, "ransac %.2f ms refine %.2f ms" % ((t1-t0)*1e3,(t2-t1)*1e3), flush=True)
s.close()
print("done")
