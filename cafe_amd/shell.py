"""Python handle on the host driver (include/cafehost.h): CAFE's command language for the hot path,
every objective evaluation running on the GPU."""
import ctypes as C

import numpy as np

from . import _lib


class CafeShell:
    def __init__(self, device=0, log_path="stdout"):
        self._L = _lib.load()
        h = C.c_void_p()
        if self._L.cafehost_create(C.byref(h), int(device), (log_path or "stdout").encode()) != 0:
            raise _lib.CafeHipError(self._L.cafehost_last_error().decode("utf-8", "replace"))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.cafehost_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise _lib.CafeHipError(self._L.cafehost_last_error().decode("utf-8", "replace"))
        return rc

    def dispatch(self, line):
        """cafe_shell_dispatch_command (cafe/cafe_commands.cpp:504-536)."""
        return self._check(self._L.cafehost_dispatch(self._h, line.encode()))

    def set_option(self, key, value):
        """cafehost_set_option: "speculate", "timing", or any switch of the device context (include/cafehip.h)."""
        self._check(self._L.cafehost_set_option(self._h, str(key).encode(), ("" if value is None else str(value)).encode()))

    def run_script(self, path):
        return self._check(self._L.cafehost_run_script(self._h, path.encode()))

    # ---- native communicator (include/cafehost.h): RCCL behind the C ABI ---------------------------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """ncclGetUniqueId through the library: bytes for rank 0 to hand to the other ranks."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        self._check(self._L.cafehost_comm_unique_id(buf))
        return buf.raw

    def init_comm(self, rank, world, unique_id):
        """Join the communicator: from here on every table this session loads is sharded and every objective
        evaluation ends in one ncclAllGather on the session's stream."""
        assert len(unique_id) == self.COMM_ID_BYTES
        self._check(self._L.cafehost_init_comm(self._h, int(rank), int(world), C.c_char_p(unique_id)))

    def speculation_stats(self):
        """(batched passes, points evaluated in them, objective calls served from them)."""
        a, b, c = C.c_long(), C.c_long(), C.c_long()
        self._check(self._L.cafehost_speculation_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def lookahead_stats(self):
        """dict: announcements, points announced, evaluations that found their matrices on the device, sets built ahead."""
        out = (C.c_long * 4)()
        self._check(self._L.cafehost_lookahead_stats(self._h, out))
        return dict(zip(("announcements", "points", "hits", "built"), [int(x) for x in out]))

    def exchange_stats(self):
        sec, calls = C.c_double(), C.c_long()
        self._check(self._L.cafehost_exchange_stats(self._h, C.byref(sec), C.byref(calls)))
        return sec.value, calls.value

    @property
    def params(self):
        n = self._L.cafehost_num_params(self._h)
        out = np.zeros(max(n, 1))
        self._L.cafehost_get_params(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n)
        return out[:n]

    @property
    def score(self):
        return self._L.cafehost_last_score(self._h)

    @property
    def iterations(self):
        return self._L.cafehost_search_iterations(self._h)

    @property
    def evaluations(self):
        return self._L.cafehost_num_evaluations(self._h)

    @property
    def search_seconds(self):
        return self._L.cafehost_search_seconds(self._h)

    @property
    def poisson_lambda(self):
        return self._L.cafehost_poisson_lambda(self._h)

    def trace(self):
        """Objective calls of the last command: rows of (params..., score)."""
        n = self._L.cafehost_num_params(self._h)
        rows = self._L.cafehost_num_evaluations(self._h)
        out = np.zeros((max(rows, 1), n + 1))
        got = self._L.cafehost_get_trace(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), rows)
        return out[:got]
