"""CAFE scripts on several GPUs of one node: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), every rank runs the same commands through the host driver; a rank scores its
chunk-aligned block of the family table and ONE all_gather per objective call combines the partial
sums (cafe_amd/distributed.py).  The score is bit-identical for any number of ranks, so every rank's
Nelder-Mead takes the same decisions and no broadcast is needed.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m cafe_amd.multi_gpu script.sh
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from . import distributed as D
from .shell import CafeShell

EXCHANGE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_int))
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong)


class MultiGpuShell(CafeShell):
    """CafeShell whose objective is sharded over the ranks of a torch.distributed group."""

    def __init__(self, torch, dist, device_index, log_path="stdout", device="cuda", native=False):
        super().__init__(device_index, log_path)
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device
        self.native = native
        if native:
            # the exchange lives behind the C ABI (cafehost_init_comm): torch.distributed only carries the
            # communicator id from rank 0 to the others; no callback is registered
            ident = [self.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            self.init_comm(self.rank, self.world, ident[0])
            self._cb = self._ag = None
            self._wired = True
            return
        self._check(self._L.cafehost_set_shard(self._h, self.rank, self.world))
        if device == "cuda":
            self._check(self._L.cafehost_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._cb = None
        self._wired = False

        def allgather(_user, mine, nbytes_mine, out, slot):
            # report / Monte-Carlo null: one fixed-slot all_gather of this rank's block of results
            try:
                buf = torch.zeros(slot, dtype=torch.uint8, device=self.device)
                if nbytes_mine:
                    src = np.ctypeslib.as_array(C.cast(mine, C.POINTER(C.c_uint8)), shape=(nbytes_mine,))
                    buf[:nbytes_mine] = torch.from_numpy(src.copy()).to(self.device)
                gathered = torch.empty(slot * self.world, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(gathered, buf)
                dst = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(slot * self.world,))
                dst[:] = gathered.cpu().numpy()
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("allgather failed:", e, file=sys.stderr, flush=True)
                return -1

        self._ag = ALLGATHER_FN(allgather)
        self._check(self._L.cafehost_set_allgather(self._h, C.cast(self._ag, C.c_void_p), None))

    def _wire(self):
        """Size and register the exchange buffers (needs tree + table)."""
        self._check(self._L.cafehost_upload(self._h))
        lo, hi, nc = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.cafehost_shard_bounds(self._h, C.byref(lo), C.byref(hi), C.byref(nc)))
        mine = self.torch.tensor([lo.value, hi.value], dtype=self.torch.int64, device=self.device)
        allb = self.torch.zeros(2 * self.world, dtype=self.torch.int64, device=self.device)
        self.dist.all_gather_into_tensor(allb, mine)
        b = allb.cpu().numpy().reshape(self.world, 2)
        self.bounds = [(int(x), int(y)) for x, y in b]
        self.slots = max(1, max((h - l + D.CHUNK - 1) // D.CHUNK for l, h in self.bounds))
        self.packed, p_chunks, p_fz = D.packed_buffer(self.torch, self.slots, self.device)
        self.gathered = self.torch.zeros((self.slots + 1) * self.world, dtype=self.torch.float64, device=self.device)
        self.gathered_host = None
        if self.device == "cuda":
            self.gathered_host = self.torch.zeros((self.slots + 1) * self.world, dtype=self.torch.float64).pin_memory()

        shell = self

        class _Fetch:   # what exchange_packed needs of an engine: the session's device context does the fetch
            def fetch_small(self, device_ptr, n_doubles):
                out = C.c_void_p()
                shell._check(shell._L.cafehost_fetch_small(shell._h, C.c_void_p(device_ptr), C.c_ulong(8 * n_doubles),
                                                           C.byref(out)))
                return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(n_doubles,))

        fetch = _Fetch() if self.device == "cuda" else None

        def exchange(_user, fz_out):
            score, fz = D.exchange_packed(self.dist, self.torch, self.packed, self.gathered, self.slots, self.bounds,
                                           self.gathered_host, engine=fetch)
            fz_out[0] = -1 if fz == D.NO_ZERO else fz
            return score

        self._cb = EXCHANGE_FN(exchange)
        self._check(self._L.cafehost_set_exchange(self._h, C.cast(self._cb, C.c_void_p), None, C.c_void_p(p_chunks),
                                                  C.c_void_p(p_fz)))
        self._wired = True

    def dispatch(self, line):
        if self.native:
            return super().dispatch(line)
        cmd = line.strip().split(" ")[0] if line.strip() else ""
        if cmd in ("load", "tree"):
            self._wired = False
            self._check(self._L.cafehost_set_exchange(self._h, None, None, None, None))
        if cmd in ("lambda", "lambdamu") and not self._wired:
            self._wire()
        return super().dispatch(line)


def main(argv=None):
    argv = argv or sys.argv[1:]
    if not argv:
        raise SystemExit("usage: python -m cafe_amd.multi_gpu script.sh")
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CAFE_BACKEND", "nccl")
    same = os.environ.get("CAFE_SAME_DEVICE") == "1"  # debugging on a 1-GPU box
    dev = 0 if same else local_rank
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend=backend)
    rank = dist.get_rank()
    sh = MultiGpuShell(torch, dist, dev, "stdout" if rank == 0 else os.devnull,
                       native=os.environ.get("CAFE_NATIVE_COMM") == "1")
    with open(argv[0]) as f:
        for line in f:
            if sh.dispatch(line) == 1:
                break
    if rank == 0:
        print("params", list(sh.params), "score", sh.score, "iterations", sh.iterations, "evaluations",
              sh.evaluations, "search_s", sh.search_seconds, flush=True)
    sh.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
