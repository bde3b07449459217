"""Host-side mirror of the reference's per-evaluation interface over the C ABI.

Names follow the reference: ``reset_birthdeath_cache`` (cafe/cafe_main.c:319-326),
``get_posterior`` (cafe/lambda.cpp:691-724), ``init_family_size`` (cafe/cafe_family.c:357-364).
Everything numerical happens in libcafehip.so on the GPU; this file only marshals arrays.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


@dataclass
class FamilySizeRange:
    """family_size_range, libtree/family.h:10-15."""
    min: int
    max: int
    root_min: int
    root_max: int


def init_family_size(max_count):
    """init_family_size, cafe/cafe_family.c:357-364."""
    m = int(max_count)
    # rint() rounds half to even, as Python's round() does
    return FamilySizeRange(0, m + max(50, m // 5), 1, max(30, int(round(m * 1.25))))


class Engine:
    """One GPU context (one process per GPU)."""

    def __init__(self, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        _lib.check(self._L.cafehip_create(C.byref(h), int(device)))
        self._h = h
        self.n_nodes = 0
        self.F = 0
        self.range = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.cafehip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------------------
    def set_option(self, key, value):
        """cafehip_set_option: run-time switches (see include/cafehip.h); value is converted with str()."""
        _lib.check(self._L.cafehip_set_option(self._h, str(key).encode(), ("" if value is None else str(value)).encode()))

    def set_stream(self, hip_stream_handle):
        """Run on the caller's stream; 0 is the HIP default stream (torch's default stream handle)."""
        _lib.check(self._L.cafehip_set_stream(self._h, C.c_void_p(hip_stream_handle or 0)))

    def set_tree(self, parent, left, right, branchlength):
        p = np.ascontiguousarray(parent, np.int32)
        l = np.ascontiguousarray(left, np.int32)
        r = np.ascontiguousarray(right, np.int32)
        b = np.ascontiguousarray(branchlength, np.float64)
        _lib.check(self._L.cafehip_set_tree(self._h, len(p), _i(p), _i(l), _i(r), _d(b)))
        self.n_nodes = len(p)

    def set_families(self, counts, rng, ref=None):
        c = np.ascontiguousarray(counts, np.int32)
        if c.ndim != 2:
            raise ValueError("counts must be F x n_leaves")
        refp = None
        if ref is not None:
            ref = np.ascontiguousarray(ref, np.int32)
            refp = _i(ref)
        _lib.check(self._L.cafehip_set_families(self._h, c.shape[0], c.shape[1], _i(c), refp, rng.min, rng.max,
                                                rng.root_min, rng.root_max))
        self.F = c.shape[0]
        self.range = rng

    def last_setup_ms(self):
        """Host milliseconds of the last set_families: dict(dedup, compression_plan, upload_alloc, total)."""
        ms = (C.c_double * 4)()
        _lib.check(self._L.cafehip_last_setup_ms(self._h, ms))
        return {"dedup": ms[0], "compression_plan": ms[1], "upload_alloc": ms[2], "total": ms[3]}

    def set_error_model(self, errormatrix, leaf_has_model=None):
        if errormatrix is None:
            _lib.check(self._L.cafehip_set_error_model(self._h, 0, None, None))
            return
        e = np.ascontiguousarray(errormatrix, np.float64)
        assert e.ndim == 2 and e.shape[0] == e.shape[1]
        lh = None
        if leaf_has_model is not None:
            lh_arr = np.ascontiguousarray(leaf_has_model, np.uint8)
            lh = lh_arr.ctypes.data_as(C.POINTER(C.c_uint8))
        _lib.check(self._L.cafehip_set_error_model(self._h, e.shape[0] - 1, _d(e), lh))

    # ---- per evaluation ----------------------------------------------------------------
    def get_posterior(self, node_lambda, node_mu, prior, per_family=False):
        """reset_birthdeath_cache + get_posterior.  Returns (score, first_zero_family[, max_lik,
        argmax_root, max_post])."""
        nl = np.ascontiguousarray(node_lambda, np.float64)
        nm = np.ascontiguousarray(node_mu, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        R = self.range.root_max - self.range.root_min + 1
        if len(nl) != self.n_nodes or len(nm) != self.n_nodes or len(pr) < R:
            raise ValueError("node_lambda/node_mu need n_nodes entries and prior >= R entries")
        if not per_family:
            # the objective path of a search / of bench.py: raw addresses and reused result cells (building three
            # ctypes pointer objects per call costs more than the launches' own CPU time at small tables)
            fast = self._fast_eval()
            self._fz.value = -1
            _lib.check(fast(self._h, nl.__array_interface__["data"][0], nm.__array_interface__["data"][0],
                            pr.__array_interface__["data"][0], self._score_ref, self._fz_ref, None, None, None))
            return self._score.value, self._fz.value
        score = C.c_double()
        fz = C.c_int32(-1)
        if per_family:
            ml = np.zeros(self.F)
            mp = np.zeros(self.F)
            am = np.zeros(self.F, np.int32)
            _lib.check(self._L.cafehip_eval_posterior(self._h, _d(nl), _d(nm), _d(pr), C.byref(score), C.byref(fz),
                                                      _d(ml), _i(am), _d(mp)))
            return score.value, fz.value, ml, am, mp
        _lib.check(self._L.cafehip_eval_posterior(self._h, _d(nl), _d(nm), _d(pr), C.byref(score), C.byref(fz),
                                                  None, None, None))
        return score.value, fz.value

    @staticmethod
    def address_of(array):
        """Raw address of a C-contiguous float64 array for get_posterior_at (the caller keeps the array alive)."""
        a = np.ascontiguousarray(array, np.float64)
        if a is not array and not (isinstance(array, np.ndarray) and a.ctypes.data == array.ctypes.data):
            raise ValueError("address_of needs a C-contiguous float64 array (a converted copy would not outlive the call)")
        return a.ctypes.data

    def get_posterior_at(self, addr_lambda, addr_mu, addr_prior, sharded=False):
        """get_posterior / get_posterior_sharded on addresses taken once with address_of: a loop over many parameter
        sets (bench.py, a likelihood surface) then spends its Python time on one foreign call per evaluation -- the
        array checks and address look-ups of the plain wrapper cost ~5 us, 4 % of a configs[1] evaluation."""
        if sharded:
            f = getattr(self, "_fast_sh", None)
            if f is None:
                self._bind_fast_sharded()
                f = self._fast_sh
            _lib.check(f(self._h, addr_lambda, addr_mu, addr_prior, self._sscore_ref, self._sfz_ref))
            return self._sscore.value, self._sfz.value
        fast = self._fast_eval()
        self._fz.value = -1
        _lib.check(fast(self._h, addr_lambda, addr_mu, addr_prior, self._score_ref, self._fz_ref, None, None, None))
        return self._score.value, self._fz.value

    # ---- matrices ahead of time (cafehip_prefetch_matrices) ------------------------------------------------------
    PREFETCH_NOW, PREFETCH_BEHIND_NEXT_EVALUATION = 0, 1

    def prefetch_matrices(self, node_lambdas, node_mus, when=0):
        """Announce parameter sets ([n_sets, n_nodes] or [n_nodes]) that may be evaluated next: their matrices are built on
        a second stream and a later get_posterior of one of them launches no matrix build (same bits)."""
        nl = np.atleast_2d(np.ascontiguousarray(node_lambdas, np.float64))
        nm = np.atleast_2d(np.ascontiguousarray(node_mus, np.float64))
        assert nl.shape == nm.shape and nl.shape[1] == self.n_nodes
        _lib.check(self._L.cafehip_prefetch_matrices(self._h, nl.shape[0], _d(nl), _d(nm), int(when)))

    def prefetch_matrices_at(self, n_sets, addr_lambda, addr_mu, when=1):
        """The same on raw addresses (address_of), one foreign call."""
        f = getattr(self, "_fast_pf", None)
        if f is None:
            vp = C.c_void_p
            f = self._fast_pf = C.CFUNCTYPE(C.c_int, vp, C.c_int, vp, vp, C.c_int)(("cafehip_prefetch_matrices", self._L))
        _lib.check(f(self._h, n_sets, addr_lambda, addr_mu, when))

    def prearm_stats(self):
        out = (C.c_long * 3)()
        _lib.check(self._L.cafehip_prearm_stats(self._h, out))
        return {"used": int(out[0]), "let_go": int(out[1]), "expired": int(out[2])}

    def matrix_cache_stats(self):
        out = (C.c_long * 8)()
        _lib.check(self._L.cafehip_matrix_cache_stats(self._h, out))
        keys = ("announced", "built", "hits", "misses", "replaced", "hits_that_waited", "build_launches", "entries")
        return dict(zip(keys, [int(v) for v in out]))

    def _bind_fast_sharded(self):
        vp = C.c_void_p
        self._fast_sh = C.CFUNCTYPE(C.c_int, vp, vp, vp, vp, vp, vp)(("cafehip_eval_posterior_sharded", self._L))
        self._sscore, self._sfz = C.c_double(), C.c_int32(-1)
        self._sscore_ref, self._sfz_ref = C.addressof(self._sscore), C.addressof(self._sfz)

    def _fast_eval(self):
        f = getattr(self, "_fast", None)
        if f is None:
            vp = C.c_void_p
            proto = C.CFUNCTYPE(C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp)
            f = self._fast = proto(("cafehip_eval_posterior", self._L))
            self._score, self._fz = C.c_double(), C.c_int32(-1)
            self._score_ref, self._fz_ref = C.addressof(self._score), C.addressof(self._fz)
        return f

    def get_posterior_sequence(self, node_lambdas, node_mus, prior, sharded=False):
        """n evaluations one after the other inside ONE foreign call (cafehip_eval_posterior_sequence): node_lambdas /
        node_mus are [n, n_nodes].  Returns (scores[n], first_zero[n])."""
        nl = np.ascontiguousarray(node_lambdas, np.float64)
        nm = np.ascontiguousarray(node_mus, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        assert nl.ndim == 2 and nl.shape == nm.shape and nl.shape[1] == self.n_nodes
        scores = np.zeros(nl.shape[0])
        fz = np.zeros(nl.shape[0], np.int32)
        _lib.check(self._L.cafehip_eval_posterior_sequence(self._h, nl.shape[0], _d(nl), _d(nm), _d(pr), _d(scores), _i(fz), 1 if sharded else 0))
        return scores, fz

    def get_posterior_multi(self, node_lambdas, node_mus, prior):
        """Several objective evaluations in one pass: node_lambdas / node_mus are [n_sets, n_nodes].
        Returns (scores[n_sets], first_zero[n_sets])."""
        nl = np.ascontiguousarray(node_lambdas, np.float64)
        nm = np.ascontiguousarray(node_mus, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        assert nl.ndim == 2 and nl.shape == nm.shape and nl.shape[1] == self.n_nodes
        scores = np.zeros(nl.shape[0])
        fz = np.zeros(nl.shape[0], np.int32)
        _lib.check(self._L.cafehip_eval_posterior_multi(self._h, nl.shape[0], _d(nl), _d(nm), _d(pr), _d(scores), _i(fz)))
        return scores, fz

    def clustered_posterior(self, node_lambdas, node_mus, weights, prior, per_family=False):
        """cafe_get_clustered_posterior (k-cluster model): node rates [K, n_nodes], weights [K].
        Returns (score, first_zero, membership_sums[K][, MAP[F], p_z[F, K]])."""
        nl = np.ascontiguousarray(node_lambdas, np.float64)
        nm = np.ascontiguousarray(node_mus, np.float64)
        w = np.ascontiguousarray(weights, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        K = nl.shape[0]
        assert nl.shape == nm.shape == (K, self.n_nodes) and w.shape == (K,)
        score, fz = C.c_double(), C.c_int32(-1)
        memb = np.zeros(K)
        fmap = np.zeros(self.F) if per_family else None
        pz = np.zeros((self.F, K)) if per_family else None
        _lib.check(self._L.cafehip_eval_clustered_posterior(self._h, K, _d(nl), _d(nm), _d(w), _d(pr), C.byref(score),
                                                            C.byref(fz), _d(memb), _d(fmap) if per_family else None,
                                                            _d(pz) if per_family else None))
        if per_family:
            return score.value, fz.value, memb, fmap, pz
        return score.value, fz.value, memb

    def launch_info(self):
        wg, cu = C.c_int32(), C.c_int32()
        _lib.check(self._L.cafehip_launch_info(self._h, C.byref(wg), C.byref(cu)))
        return wg.value, cu.value

    def last_tables_ms(self):
        """With timing on: the part of last_kernel_ms()[1] spent in the k2c_nodes launches (compressed subtrees)."""
        v = C.c_double()
        _lib.check(self._L.cafehip_last_tables_ms(self._h, C.byref(v)))
        return v.value

    def last_issued_flops(self):
        """(walk, tables): matrix-instruction flops issued by the last objective evaluation's pruning."""
        w, t = C.c_double(), C.c_double()
        _lib.check(self._L.cafehip_last_issued_flops(self._h, C.byref(w), C.byref(t)))
        return w.value, t.value

    def eval_posterior_async(self, node_lambda, node_mu, prior, d_chunk_sums_ptr, d_first_zero_ptr):
        nl = np.ascontiguousarray(node_lambda, np.float64)
        nm = np.ascontiguousarray(node_mu, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        _lib.check(self._L.cafehip_eval_posterior_async(self._h, _d(nl), _d(nm), _d(pr), C.c_void_p(d_chunk_sums_ptr),
                                                        C.c_void_p(d_first_zero_ptr)))

    # ---- multi-GPU (one process per GPU; include/cafehip.h) ----------------------------------------------
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        _lib.check(self._L.cafehip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == self.COMM_ID_BYTES
        _lib.check(self._L.cafehip_comm_init(self._h, int(rank), int(world), C.c_char_p(unique_id)))

    def comm_set_blocks(self, bounds):
        """bounds: [(lo, hi)] of every rank's block of the global table (cafe_amd.distributed.shard_bounds)."""
        lo = np.ascontiguousarray([b[0] for b in bounds], np.int32)
        hi = np.ascontiguousarray([b[1] for b in bounds], np.int32)
        _lib.check(self._L.cafehip_comm_set_blocks(self._h, _i(lo), _i(hi)))

    def comm_allgather(self, mine, slot_bytes):
        """All-gather of host blocks (bytes): returns world slots of slot_bytes each, in rank order."""
        info = self.comm_info()
        mine = bytes(mine)
        out = C.create_string_buffer(slot_bytes * info["world"])
        _lib.check(self._L.cafehip_comm_allgather(self._h, C.c_char_p(mine), len(mine), out, slot_bytes))
        return [out.raw[r * slot_bytes:(r + 1) * slot_bytes] for r in range(info["world"])]

    def get_posterior_sharded(self, node_lambda, node_mu, prior):
        """One objective evaluation of the sharded table: (score, first_zero_global), the same on every rank."""
        nl = np.ascontiguousarray(node_lambda, np.float64)
        nm = np.ascontiguousarray(node_mu, np.float64)
        pr = np.ascontiguousarray(prior, np.float64)
        f = getattr(self, "_fast_sh", None)
        if f is None:
            self._bind_fast_sharded()
            f = self._fast_sh
        _lib.check(f(self._h, nl.__array_interface__["data"][0], nm.__array_interface__["data"][0],
                     pr.__array_interface__["data"][0], self._sscore_ref, self._sfz_ref))
        return self._sscore.value, self._sfz.value

    def comm_info(self):
        """dict(rank, world, mode, exchange_ms, host_seconds, calls); mode: 0 none, 1 rccl, 2 direct."""
        r, w, m = C.c_int32(), C.c_int32(), C.c_int32()
        ms, hs, n = C.c_double(), C.c_double(), C.c_long()
        _lib.check(self._L.cafehip_comm_info(self._h, C.byref(r), C.byref(w), C.byref(m), C.byref(ms), C.byref(hs), C.byref(n)))
        return {"rank": r.value, "world": w.value, "mode": {0: "none", 1: "rccl", 2: "direct"}[m.value],
                "exchange_ms": ms.value, "host_seconds": hs.value, "calls": n.value}

    def comm_status(self):
        """What the communicator's set-up found (cafehip_comm_status): the mode the ranks agreed on after the functional
        probe, what this rank mapped and heard, and what RCCL itself reports -- for logs and bench records."""
        w = np.zeros(16, np.int32)
        ms = C.c_double()
        _lib.check(self._L.cafehip_comm_status(self._h, _i(w), C.byref(ms)))
        name = {0: "none", 1: "rccl", 2: "direct"}
        return {"comm_world": int(w[0]), "mode_agreed_at_init": name[int(w[1])], "mode": name[int(w[2])],
                "direct_ok_on_every_rank": bool(w[3]), "peers_mapped": int(w[4]), "peers_heard_by_probe": int(w[5]),
                "rccl_initialised": bool(w[6]), "rccl_ranks": int(w[7]), "probe_injected_mute": bool(w[8]),
                "host_paced_repolls": int(w[9]), "probe_ms": ms.value}

    def comm_resync(self):
        _lib.check(self._L.cafehip_comm_resync(self._h))

    def num_chunks(self):
        return self._L.cafehip_num_chunks(self._h)

    def reset_birthdeath_cache(self, node_lambda, node_mu):
        nl = np.ascontiguousarray(node_lambda, np.float64)
        nm = np.ascontiguousarray(node_mu, np.float64)
        _lib.check(self._L.cafehip_reset_birthdeath_cache(self._h, _d(nl), _d(nm)))

    def set_exact_matrices(self, on=True):
        """Matrix builds in the reference's per-term arithmetic (report phase) instead of the product form."""
        _lib.check(self._L.cafehip_set_exact_matrices(self._h, 1 if on else 0))

    def get_matrix(self, node):
        S = self._L.cafehip_matrix_size(self._h)
        out = np.zeros((S, S))
        s_out = C.c_int()
        _lib.check(self._L.cafehip_get_matrix(self._h, int(node), _d(out), C.byref(s_out)))
        return out

    def eval_root_likelihoods(self, counts, root_lo, root_hi, col_max):
        c = np.ascontiguousarray(counts, np.int32)
        lo = np.ascontiguousarray(root_lo, np.int32)
        hi = np.ascontiguousarray(root_hi, np.int32)
        cm = np.ascontiguousarray(col_max, np.int32)
        n = int((hi - lo + 1).sum())
        out = np.zeros(n)
        _lib.check(self._L.cafehip_eval_root_likelihoods(self._h, c.shape[0], _i(c), _i(lo), _i(hi), _i(cm), _d(out)))
        return out

    def viterbi(self, counts, root_lo, root_hi, col_max):
        """cafe_tree_viterbi for a batch of rows -> node sizes [B, n_nodes]."""
        c = np.ascontiguousarray(counts, np.int32)
        lo = np.ascontiguousarray(root_lo, np.int32)
        hi = np.ascontiguousarray(root_hi, np.int32)
        cm = np.ascontiguousarray(col_max, np.int32)
        out = np.zeros((c.shape[0], self.n_nodes), np.int32)
        _lib.check(self._L.cafehip_viterbi(self._h, c.shape[0], _i(c), _i(lo), _i(hi), _i(cm), _i(out)))
        return out

    def fetch_small(self, device_ptr, n_doubles):
        """Device doubles (e.g. the output of a collective on this engine's stream) -> numpy view of the engine's
        pinned host mirror, without a copy command or a stream synchronisation (cafehip_fetch_small)."""
        out = C.c_void_p()
        _lib.check(self._L.cafehip_fetch_small(self._h, C.c_void_p(device_ptr), C.c_size_t(8 * n_doubles), C.byref(out)))
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(n_doubles,))

    def enable_timing(self, on=True):
        _lib.check(self._L.cafehip_enable_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = (C.c_double * 3)()
        _lib.check(self._L.cafehip_last_kernel_ms(self._h, ms))
        return list(ms)

    def last_batch_ms(self):
        """Pruning-launch duration of the last eval_root_likelihoods call (timing enabled)."""
        ms = C.c_double()
        _lib.check(self._L.cafehip_last_batch_ms(self._h, C.byref(ms)))
        return ms.value

    def describe(self):
        return self._L.cafehip_describe(self._h).decode()
