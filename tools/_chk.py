import json
d=json.loads(open("gpurun_out/bench_slots.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_selective","value_sequence","ransac_restarts","ms_per_step")})
print({k:(round(v["median_ms_per_solve"],3), round(v["min_ms_per_solve"],3), v.get("refine_iterations")) for k,v in d["regimes"].items()})
t=d["tiled_full"]; print({k:t.get(k) for k in ("ms_per_solve","collectives_min_med_max","host_syncs_min_med_max")})
print(d["roofline"]["frac"], d["roofline"].get("shader_clock_mhz"), d["roofline"].get("counters_stale"))
