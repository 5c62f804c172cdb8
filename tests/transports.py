"""Test-only collectives for rsdsfm_dist_set_transport: several logical ranks of the native column-tiled solve on ONE GPU.

ThreadTransport: N ranks = N host threads of one process (one solver context each); GlooTransport: N processes sharing the GPU,
exchanging through torch.distributed (gloo, host memory).  Both stage through blocking hipMemcpy calls -- correctness
vehicles, not fast paths; production ranks use RCCL (rsdsfm_dist_init)."""
import ctypes
import threading

import numpy as np

_D2H, _D2D, _H2D = 2, 3, 1


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    return hip


class ThreadTransport:
    def __init__(self, nranks, timeout=120.0):
        self.n, self.hip = nranks, _hip()
        self.barrier = threading.Barrier(nranks, timeout=timeout)
        self.slots = [None] * nranks

    def callbacks(self, rank):
        hip, n = self.hip, self.n

        def all_gather(send, recv, nbytes, stream):
            try:
                assert hip.hipStreamSynchronize(stream) == 0
                self.slots[rank] = send
                self.barrier.wait()  # every rank's contribution is complete and published
                for r in range(n):
                    dst = recv + r * nbytes
                    if dst != self.slots[r]:
                        assert hip.hipMemcpy(dst, self.slots[r], nbytes, _D2D) == 0
                # a device-to-device hipMemcpy may return before the copy has finished, and the contexts' streams do not wait for the null
                # stream: without this the kernels a rank enqueues next could read `recv` too early (seen once as one rank's refinement
                # summary differing from the others')
                assert hip.hipDeviceSynchronize() == 0
                self.barrier.wait()  # nobody reuses its send buffer before all have read it
                return 0
            except Exception:  # a broken barrier / failed copy must surface as an error code, not as a hang
                self.barrier.abort()
                return 1

        def all_reduce(buf, count, stream):
            try:
                assert hip.hipStreamSynchronize(stream) == 0
                self.slots[rank] = buf
                self.barrier.wait()
                acc, tmp = np.zeros(count), np.empty(count)
                for r in range(n):  # rank order
                    assert hip.hipMemcpy(tmp.ctypes.data, self.slots[r], 8 * count, _D2H) == 0
                    acc += tmp
                self.barrier.wait()  # all have read every buffer
                assert hip.hipMemcpy(buf, acc.ctypes.data, 8 * count, _H2D) == 0
                self.barrier.wait()
                return 0
            except Exception:
                self.barrier.abort()
                return 1

        return all_gather, all_reduce


class GlooTransport:
    """ranks = processes of an initialised torch.distributed group (any backend that moves CPU tensors)"""

    def __init__(self, dist, torch):
        self.dist, self.torch, self.hip = dist, torch, _hip()
        self.n, self.rank = dist.get_world_size(), dist.get_rank()

    def callbacks(self):
        hip, torch, dist, n = self.hip, self.torch, self.dist, self.n

        def all_gather(send, recv, nbytes, stream):
            try:
                assert hip.hipStreamSynchronize(stream) == 0
                mine = torch.empty(nbytes, dtype=torch.uint8)
                assert hip.hipMemcpy(mine.data_ptr(), send, nbytes, _D2H) == 0
                outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(n)]
                dist.all_gather(outs, mine)
                full = torch.cat(outs)
                assert hip.hipMemcpy(recv, full.data_ptr(), nbytes * n, _H2D) == 0
                return 0
            except Exception:
                return 1

        def all_reduce(buf, count, stream):
            try:
                assert hip.hipStreamSynchronize(stream) == 0
                t = torch.empty(count, dtype=torch.float64)
                assert hip.hipMemcpy(t.data_ptr(), buf, 8 * count, _D2H) == 0
                dist.all_reduce(t)  # one non-zero term per entry: exact in any order
                assert hip.hipMemcpy(buf, t.data_ptr(), 8 * count, _H2D) == 0
                return 0
            except Exception:
                return 1

        return all_gather, all_reduce
