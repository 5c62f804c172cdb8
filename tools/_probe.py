import importlib.util, os, sys
import torch
def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(path, "__init__.py"), submodule_search_locations=[path])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m); return m
a = load("ab_probe/rs-aware-differential-sfm_amd", "pkg_probe")
d = a.synth.make_config(5)
rows, cols = d["rows"], d["cols"]
img = torch.from_numpy(d["flow_img"]).cuda()
dm = torch.empty((cols, rows), dtype=torch.float64, device="cuda")
with a.Solver(0) as s:
    for i in range(6):
        s.solve_frame_dev(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), trials=50, tol=0.05, seed=1 + i)
        s.synchronize()
