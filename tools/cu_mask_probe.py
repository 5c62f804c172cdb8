"""Probe: do CU-masked HIP streams (hipExtStreamCreateWithCUMask) partition the MI355X the way a space-shared pipeline would need?
Runs the whole solve (1280x720, T = 50) on contexts bound to streams with different CU masks and prints the median solve time and the
duration of the dominant kernel; then two contexts on disjoint masks solving concurrently from two host threads."""
import ctypes as C
import sys
import threading
import time

sys.path.insert(0, ".")
import numpy as np
import torch

import rsdsfm

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
dev = torch.device("cuda", 0)
d = rsdsfm.synth.make_config(5, seed=0x5EED0005)
rows, cols = d["rows"], d["cols"]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st.value


def bench(stream, n=40, tag=""):
    img = torch.from_numpy(d["flow_img"]).to(dev)
    dm = torch.empty((cols, rows), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with rsdsfm.Solver(0, stream=stream) as s:
        call = s.prepared_frame_solve(img.data_ptr(), rows, cols, d["K"], d["gamma"], dm.data_ptr(), trials=50, tol=0.05)
        s.set_profiling(True)
        for i in range(4):
            call(1 + i)
        ts, ks = [], []
        for i in range(n):
            t0 = time.perf_counter()
            call(10 + i)
            ts.append(time.perf_counter() - t0)
            ks.append(s.profile_last_ms("ransac_lm_round0"))
        return np.median(ts) * 1e3, np.median(ks)


ALL = (1 << 256) - 1
for name, bits in (("all 256", ALL), ("low 192 bits", (1 << 192) - 1), ("low 128 bits", (1 << 128) - 1), ("low 64 bits", (1 << 64) - 1),
                   ("every 4th bit (64)", sum(1 << i for i in range(0, 256, 4))), ("3 of every 4 bits (192)", sum(1 << i for i in range(256) if i % 4 != 3))):
    ms, k = bench(masked_stream(bits))
    print("mask %-24s: solve %.3f ms, ransac_lm_kernel %.3f ms" % (name, ms, k), flush=True)

# two contexts on disjoint masks, concurrently
for name, a, b in (("192 / 64 (low / high bits)", (1 << 192) - 1, ALL ^ ((1 << 192) - 1)), ("3-of-4 / every-4th", sum(1 << i for i in range(256) if i % 4 != 3), sum(1 << i for i in range(3, 256, 4))),
                   ("all / all", ALL, ALL)):
    out = [None, None]
    sts = [masked_stream(a), masked_stream(b)]
    bar = threading.Barrier(2)

    def work(j):
        torch.cuda.set_device(0)
        bar.wait()
        out[j] = bench(sts[j], n=30)

    th = [threading.Thread(target=work, args=(j,)) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("concurrent %-28s: A solve %.3f ms (lm %.3f), B solve %.3f ms (lm %.3f)" % (name, out[0][0], out[0][1], out[1][0], out[1][1]), flush=True)
