"""CPU: the N > 1 structure without torch -- launch-shape detection, the thread-per-GPU group, the
process-per-GPU group over the file store (what `torch.distributed.run ... bench.py` uses), block sharding and
the gather of fixed-size records.  No GPU: ranks compute their shard with the oracle, the exchange runs on the
host backend (the RCCL backend has the same interface; it is exercised on the GPU box with one rank)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_plan_launch_shapes():
    from squigglekit_amd import multigpu
    assert multigpu.plan(1, {}) == ("single", 0, 0, 1)
    assert multigpu.plan(8, {}) == ("threads", 0, 0, 8)            # a bare `python bench.py --gpus 8`
    env = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3", "MASTER_PORT": "29500"}
    assert multigpu.plan(4, env) == ("process", 3, 3, 4)            # under torch.distributed.run
    assert multigpu.plan(1, env) == ("process", 3, 3, 4)            # the launcher's world size wins
    assert multigpu.plan(2, {"WORLD_SIZE": "1", "RANK": "0"}) == ("threads", 0, 0, 2)
    assert multigpu.launch_env({"WORLD_SIZE": "x"}) is None


def test_bench_cli_defaults_finish_fast():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse([])
    assert (a.gpus, a.workload, a.scaling, a.reads, a.samples, a.motif) == (1, "motifseq", "strong", 1_000_000, 4000, 200)      # C4 as BASELINE.json words it
    rows = bench.strided_rows(1_000_000, 8192)
    assert rows[0] == 0 and rows[-1] == 999_999 and 8000 <= len(rows) <= 8192
    assert np.unique(rows // 250_000).size == 4                      # every chunk of the screening path is sampled


def test_thread_group_host_exchange_and_errors():
    from squigglekit_amd import multigpu
    g = multigpu.ThreadGroup([0, 1, 2], rccl=False, bind=False)
    assert g.backend == "host"

    def body(comm):
        assert comm.ranks_seen() == 3
        comm.barrier()
        got = comm.allgather_host(np.array([comm.rank * 10.0, 1.5]))
        got2 = comm.allgather_host(np.arange(4, dtype=np.int32) + comm.rank)
        return got.tolist(), got2[:, 0].tolist()

    res = g.run(body)
    assert all(r == ([[0.0, 1.5], [10.0, 1.5], [20.0, 1.5]], [0, 1, 2]) for r in res)

    def boom(comm):
        if comm.rank == 1:
            raise ValueError("rank 1 failed")
        comm.barrier()                                               # must not hang: the barrier is aborted
    g2 = multigpu.ThreadGroup([0, 1], rccl=False, bind=False)
    with pytest.raises(ValueError, match="rank 1 failed"):
        g2.run(boom)
    assert g2.run(lambda comm: comm.allgather_host(np.array([comm.rank])).ravel().tolist()) == [[0, 1], [0, 1]], \
        "a failed run must leave the group usable (the aborted barrier is reset)"
    with pytest.raises(ValueError):
        multigpu.ThreadGroup([0, 0], rccl=False, bind=False)


def test_oversubscribed_group_and_vote():
    """Ranks sharing one device (the dry run of the N > 1 path on a one-GPU box): own context slot each, host backend;
    all_ok() is the vote taken before a device collective so that a failed rank strands nobody inside it."""
    from squigglekit_amd import multigpu
    assert multigpu.oversubscribed({}) is None and multigpu.oversubscribed({"SK_OVERSUBSCRIBE": "0"}) is None
    assert multigpu.oversubscribed({"SK_OVERSUBSCRIBE": "1"}) == 0
    assert multigpu.oversubscribed({"SK_OVERSUBSCRIBE": "d5"}) == 5
    g = multigpu.ThreadGroup([0, 0, 0], bind=False, oversubscribe=True)
    assert g.shared and g.backend == "host" and "share" in g.why_host

    def body(comm):
        a = comm.all_ok(True)
        b = comm.all_ok(comm.rank != 2)
        return a, b, comm.ranks_seen()
    assert g.run(body) == [(True, False, 3)] * 3
    assert multigpu.launch_env({"WORLD_SIZE": "2", "RANK": "zero"}) is None
    assert multigpu.plan(2, {"SK_FORCE_PROCESS_SHAPE": "1", "RANK": "x"}) == ("threads", 0, 0, 2)


def test_file_store_refuses_a_directory_others_can_write(tmp_path):
    from squigglekit_amd import multigpu
    bad = tmp_path / "open"
    bad.mkdir()
    os.chmod(bad, 0o777)
    with pytest.raises(RuntimeError, match="private"):
        multigpu.FileStore(str(bad), 0, 1)
    target = tmp_path / "real"
    target.mkdir()
    link = tmp_path / "link"
    os.symlink(target, link)
    with pytest.raises(RuntimeError, match="private"):
        multigpu.FileStore(str(link), 0, 1)
    ok = multigpu.FileStore(str(tmp_path / "fresh"), 0, 1)
    assert (os.stat(ok.path).st_mode & 0o077) == 0


def test_sharded_host_gather_equals_unsharded(ora):
    """run_sharded's contract: fn(lo, hi) per rank over the block split, results written into views of one
    host array == the unsharded job."""
    from squigglekit_amd import multigpu, sharding, synth
    motif = synth.synthetic_motif(40)
    sig = synth.squiggle_batch(23, 600, 99, motif=motif)
    lens = np.full(23, 600, dtype=np.int32)
    want = ora.motifseq_batch_i16(sig, lens, motif)
    out = np.zeros_like(want)
    g = multigpu.ThreadGroup([0, 1, 2], rccl=False, bind=False)
    seen = []

    def body(comm):
        lo, hi = sharding.shard_bounds(23, comm.rank, comm.world)
        seen.append((lo, hi))
        out[lo:hi] = ora.motifseq_batch_i16(sig[lo:hi], lens[lo:hi], motif)
    g.run(body)
    assert out.tobytes() == want.tobytes()
    assert sorted(seen) == [(0, 8), (8, 16), (16, 23)]


def test_run_sharded_reruns_a_failed_ranks_block_on_the_survivors(ora, monkeypatch, capsys):
    """A rank whose shard raises a library error (a device that dropped out) does not abort the job: its block is cut up
    over the surviving devices and run again; invalid arguments and an all-ranks failure are still raised."""
    from squigglekit_amd import _lib, multigpu, synth
    motif = synth.synthetic_motif(40)
    sig = synth.squiggle_batch(23, 600, 99, motif=motif)
    lens = np.full(23, 600, dtype=np.int32)
    want = ora.motifseq_batch_i16(sig, lens, motif)
    out = np.zeros_like(want)
    for key in ((0, 1, 2), (0, 2), (1,)):                        # (host-logic groups: no device to bind here)
        multigpu._groups[key] = multigpu.ThreadGroup(key, rccl=False, bind=False)
    calls = []

    def shard(lo, hi, comm):
        calls.append((lo, hi))
        out[lo:hi] = ora.motifseq_batch_i16(sig[lo:hi], lens[lo:hi], motif)

    def fault(rank):
        if rank == 1:
            raise _lib.SquiggleKitError(-3, "hipStreamSynchronize failed: device lost (injected)")
    monkeypatch.setattr(multigpu, "_fault_hook", fault)
    g = multigpu.run_sharded([0, 1, 2], 23, shard)
    monkeypatch.setattr(multigpu, "_fault_hook", None)
    assert out.tobytes() == want.tobytes()
    assert sorted(calls) == [(0, 8), (8, 12), (12, 16), (16, 23)]      # rank 1's block 8..15 redone by devices 0 and 2
    assert g.devices == [0, 2] and (0, 1, 2) not in multigpu._groups
    assert "device 1 (rank 1) failed on reads 8..15" in capsys.readouterr().err
    # every rank failing, or the caller's arguments being wrong: raised
    multigpu._groups[(0, 2)] = multigpu.ThreadGroup((0, 2), rccl=False, bind=False)
    monkeypatch.setattr(multigpu, "_fault_hook", lambda rank: (_ for _ in ()).throw(_lib.SquiggleKitError(-3, "all gone")))
    with pytest.raises(_lib.SquiggleKitError):
        multigpu.run_sharded([0, 2], 23, shard)
    # the caller's arguments being wrong (SK_ERR_INVALID = -1) on ONE rank: raised at once, nothing is re-run
    multigpu._groups[(0, 2)] = multigpu.ThreadGroup((0, 2), rccl=False, bind=False)
    del calls[:]

    def bad_argument(rank):
        if rank == 1:
            raise _lib.SquiggleKitError(_lib.SK_ERR_INVALID, "bad argument")
    monkeypatch.setattr(multigpu, "_fault_hook", bad_argument)
    with pytest.raises(_lib.SquiggleKitError) as ei:
        multigpu.run_sharded([0, 2], 23, shard)
    assert ei.value.code == _lib.SK_ERR_INVALID and sorted(calls) == [(0, 12)]
    capsys.readouterr()
    # a device that is gone (SK_ERR_NO_DEVICE = -2, "device lost"): resharded like any other device failure
    for key in ((0, 1, 2), (0, 1)):
        multigpu._groups[key] = multigpu.ThreadGroup(key, rccl=False, bind=False)
    del calls[:]
    out[:] = np.zeros_like(want)

    def lost(rank):
        if rank == 2:
            raise _lib.SquiggleKitError(_lib.SK_ERR_NO_DEVICE, "device lost (injected)")
    monkeypatch.setattr(multigpu, "_fault_hook", lost)
    g = multigpu.run_sharded([0, 1, 2], 23, shard)
    monkeypatch.setattr(multigpu, "_fault_hook", None)
    assert out.tobytes() == want.tobytes() and g.devices == [0, 1]
    assert "device 2 (rank 2) failed on reads 16..22" in capsys.readouterr().err
    assert (_lib.SK_ERR_INVALID, _lib.SK_ERR_NO_DEVICE, _lib.SK_ERR_HIP, _lib.SK_ERR_NOMEM) == (-1, -2, -3, -4)
    multigpu._groups.clear()


_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import oracle as ora
from squigglekit_amd import multigpu, sharding, synth
from squigglekit_amd._lib import HIT_DTYPE
shape, rank, local, world = multigpu.plan(int(sys.argv[2]))
assert shape == "process" and world == int(sys.argv[2])
total = int(sys.argv[3])
motif = synth.synthetic_motif(40)
sig = synth.squiggle_batch(total, 600, 4321, motif=motif)            # every rank can regenerate the job
lo, hi = sharding.shard_bounds(total, rank, world)
lens = np.full(hi - lo, 600, dtype=np.int32)
local_hits = ora.motifseq_batch_i16(sig[lo:hi], lens, motif)          # this rank's shard only
sizes = sharding.shard_sizes(total, world)
with multigpu.ProcessGroup(rank, local, world, bind=False) as comm:
    assert comm.backend == "host" and comm.ranks_seen() == world
    comm.barrier()
    padded = np.zeros(max(sizes), dtype=HIT_DTYPE)
    padded[:hi - lo] = local_hits
    allr = comm.allgather_host(padded)                                # [world, pad] records, rank order
    tmax = comm.allgather_host(np.array([float(rank)])).max()
    full = np.concatenate([allr[r, :sizes[r]] for r in range(world)])
    if rank == 0:
        want = ora.motifseq_batch_i16(sig, np.full(total, 600, dtype=np.int32), motif)
        print("RESULT", full.tobytes() == want.tobytes(), full.size, tmax)
"""


@pytest.mark.parametrize("total", [17, 32])
def test_process_group_file_store_world2(total, tmp_path):
    """Two processes with the launcher's environment (as torch.distributed.run sets it), no torch: the rendezvous
    directory, the exchange and the assembled result.  The directory starts out holding what a crashed earlier job
    with the same address / port left behind: none of it may be taken for this job's."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    stale = tmp_path / ("sk_rdzv_%d_127.0.0.1_29%03d_none" % (os.getuid(), total + 100))
    stale.mkdir()
    (stale / "session").write_bytes(b"deadbeefdeadbeef|00|aaaaaaaaaaaaaaaa|bbbbbbbbbbbbbbbb")
    (stale / "hello.0").write_bytes(b"aaaaaaaaaaaaaaaa")
    (stale / "hello.1").write_bytes(b"bbbbbbbbbbbbbbbb")
    (stale / "ack.1").write_bytes(b"deadbeefdeadbeef")
    (stale / "ag.deadbeefdeadbeef.1.0").write_bytes(b"stale")
    (stale / "ag.deadbeefdeadbeef.1.1").write_bytes(b"stale")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29%03d" % (total + 100), SK_COMM="host", TMPDIR=str(tmp_path))
        env.pop("SK_RDZV_DIR", None)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, "2", str(total)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert outs[0][0].split() == ["RESULT", "True", str(total), "1.0"]
    assert not [d for d in os.listdir(tmp_path) if d.startswith("sk_rdzv_")]      # the store cleaned up after itself


def test_bounded_comm_init_gives_up(monkeypatch):
    """A communicator that never comes up (wedged peer) must not hang the launch: the helper is abandoned after
    SK_COMM_TIMEOUT and the caller falls back to the host gather; a quick one passes its result (or its exception)."""
    import time
    from squigglekit_amd import multigpu
    monkeypatch.setenv("SK_COMM_TIMEOUT", "1")
    t0 = time.monotonic()
    done, why = multigpu._bounded(lambda: time.sleep(5), "ncclCommInitAll")
    assert not done and "ncclCommInitAll" in why and time.monotonic() - t0 < 3
    assert multigpu._bounded(lambda: 7, "x") == (True, 7)
    try:
        multigpu._bounded(lambda: 1 // 0, "x")
    except ZeroDivisionError:
        pass
    else:
        raise AssertionError("the helper's exception must reach the caller")


def test_cpulist_parsing():
    from squigglekit_amd import multigpu
    assert multigpu._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert multigpu._parse_cpulist("\n") == set()
    assert multigpu._parse_cpulist("5") == {5}
