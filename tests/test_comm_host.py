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


def _mode_ranks(tmp_path, outcomes):
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    world = len(outcomes)
    procs = []
    for r, (probe_ok, rccl_ok) in enumerate(outcomes):
        out = tmp_path / ("m%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "mode", str(r), str(world),
                                             str(idfile), str(out), str(probe_ok), str(rccl_ok)], cwd=ROOT,
                                            env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="30"))))
    res = []
    for out, p in procs:
        assert p.wait(timeout=300) == 0
        res.append(json.load(open(out)))
    return res


@pytest.mark.parametrize("outcomes,expected", [
    ([(1, 1), (1, 1), (1, 1)], 2),          # every probe heard every peer: direct
    ([(1, 1), (0, 1), (1, 1)], 1),          # rank 1 mapped its peers but did not hear them: EVERY rank goes to RCCL
    ([(1, 0), (1, 1), (0, 1)], 0),          # ... and where one rank cannot join RCCL either: every rank reports "none"
    ([(0, 1), (0, 1)], 1),
    ([(1, 0)], 2),                          # one rank: its own store must come back
    ([(1, 1), (0, -1), (1, 1)], 0),         # rank 1 rules RCCL out locally (option comm=direct in ONE process): nobody enters the
                                            # collective join -- the ranks still run the same rounds and agree on "none" 
])
def test_every_rank_lands_in_the_same_exchange_mode(tmp_path, outcomes, expected):
    # "mapped" is not "reachable" (VERDICT r03 #1): the ranks agree on ONE mode at set-up from their probe outcomes --
    # an unreachable peer on one rank can never surface as a 120 s wait inside another rank's first evaluation
    res = _mode_ranks(tmp_path, outcomes)
    assert [r["rc"] for r in res] == [0] * len(outcomes), res
    assert [r["mode"] for r in res] == [expected] * len(outcomes), res


def test_a_failing_rank_names_itself_to_the_others(tmp_path):
    # a block larger than its slot is refused where it is offered (it used to overflow into the next slot), and the
    # other ranks hear THAT message at once instead of a barrier time-out
    idfile = tmp_path / "id"
    idfile.write_bytes(os.urandom(128))
    procs = []
    for r in range(3):
        out = tmp_path / ("o%d.json" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "oversize", str(r), "3",
                                             str(idfile), str(out), "1"], cwd=ROOT, env=dict(os.environ, CAFEHIP_COMM_TIMEOUT_S="60"))))
    res = []
    for out, p in procs:
        assert p.wait(timeout=300) == 0
        res.append(json.load(open(out)))
    assert all(r["rc"] != 0 for r in res), res
    assert "does not fit" in res[1]["err"]
    for r in (res[0], res[2]):
        assert "rank 1 failed" in r["err"] and "does not fit" in r["err"] and r["seconds"] < 30, r


def test_cleanup_removes_the_names_of_a_killed_job(tmp_path):
    # a rank that waits for a rendezvous which never completes is killed: the launcher removes the segment by id
    from cafe_amd import _lib
    import ctypes as C
    import glob
    import time
    L = _lib.load()
    uid = os.urandom(128)
    idfile = tmp_path / "id"
    idfile.write_bytes(uid)
    before = set(glob.glob("/dev/shm/cafehip_*"))
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), "host", "0", "2", str(idfile), str(tmp_path / "x.json")], cwd=ROOT)
    for _ in range(300):
        time.sleep(0.1)
        if set(glob.glob("/dev/shm/cafehip_*")) - before:
            break
    p.kill()
    p.wait()
    left = set(glob.glob("/dev/shm/cafehip_*")) - before
    assert len(left) == 1, left
    assert L.cafehip_comm_cleanup(C.c_char_p(uid)) == 1
    assert not (set(glob.glob("/dev/shm/cafehip_*")) - before)


def test_bulk_random_draws_continue_glibc_random_r_stream():
    # the Monte-Carlo null draws its uniforms in one loop over the generator's state (cafe_host.cpp GlibcRand::fill_raw):
    # the same values as random_r, whatever is drawn singly before and after, across several wraps of the 31-word state
    from cafe_amd import _lib
    L = _lib.load()
    for seed, before, bulk, after in ((10, 0, 1000, 5), (10, 7, 100003, 64), (1, 33, 31, 2), (12345, 1, 0, 3), (7, 0, 1, 0)):
        assert L.cafehost_rng_selftest(seed, before, bulk, after) == 0, L.cafehost_last_error()
