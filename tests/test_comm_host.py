"""The host half of the native communicator (cafe_amd/csrc/comm.hpp) with several PROCESSES on CPU: rendezvous by
id through a POSIX shared-memory segment, mailboxes, the sense-reversing barrier and the all-gather of ragged host
blocks in fixed slots that the report phase uses (cafehip_comm_allgather).  No GPU: the device half (peer buffers,
the exchange inside the score kernel) is covered by tests/test_gpu_comm.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 5])
def test_rendezvous_barrier_and_host_allgather(tmp_path, world):
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    procs = []
    for r in range(world):
        out = tmp_path / ("r%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "host", str(r), str(world),
                                             str(idfile), str(out)], cwd=ROOT)))
    for out, p in procs:
        assert p.wait(timeout=300) == 0
        res = json.load(open(out))
        assert res["ok"], res


def test_a_missing_rank_fails_the_call_instead_of_hanging(tmp_path, monkeypatch):
    # world 2 but only rank 0 shows up: the rendezvous gives up (the time-out is shortened through the environment)
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    out = tmp_path / "r0.json"
    env = dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "host", "0", "2", str(idfile), str(out)],
                       cwd=ROOT, env=env, timeout=120)
    assert p.returncode == 0
    res = json.load(open(out))
    assert not res["ok"] and "rendezvous timed out" in res["err"], res


def test_bulk_random_draws_continue_glibc_random_r_stream():
    # the Monte-Carlo null draws its uniforms in one loop over the generator's state (cafe_host.cpp GlibcRand::fill_raw):
    # the same values as random_r, whatever is drawn singly before and after, across several wraps of the 31-word state
    from cafe_amd import _lib
    L = _lib.load()
    for seed, before, bulk, after in ((10, 0, 1000, 5), (10, 7, 100003, 64), (1, 33, 31, 2), (12345, 1, 0, 3), (7, 0, 1, 0)):
        assert L.cafehost_rng_selftest(seed, before, bulk, after) == 0, L.cafehost_last_error()
