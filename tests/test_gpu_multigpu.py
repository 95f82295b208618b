"""GPU (one device is all the test box has): the RCCL side of the multi-GPU path with a one-rank communicator --
library-loaded librccl, ncclCommInitAll / ncclCommInitRank, the device all-gather of hit records, and
bench.py's self-launched gather path.  The N > 1 structure itself is covered on CPU (test_multigpu_host.py,
test_distributed_gloo.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_thread_group_rccl_one_rank(gpu):
    from squigglekit_amd import multigpu
    g = multigpu.ThreadGroup([0])
    assert g.backend == "rccl", g.why_host

    def body(comm):
        assert comm.ranks_seen() == 1                                 # ncclCommCount
        comm.barrier()
        return comm.allgather_host(np.array([3.5, -1.0])).tolist()
    assert g.run(body) == [[[3.5, -1.0]]]
    g.close()
    assert g.backend == "host"


def test_motifseq_sharded_rccl_gather_equals_plain_call(gpu, ora):
    from squigglekit_amd import api, multigpu, synth
    motif = synth.synthetic_motif(120, seed=9)
    sig = synth.squiggle_batch(300, 3000, 2468, motif=motif)
    lens = np.full(300, 3000, dtype=np.int32)
    lens[::7] = 1777
    plain = api.motifseq_batch(sig, lens, motif)
    got, info = multigpu.motifseq_sharded(sig, lens, motif, 0, 0, 1200, [0], gather="rccl")
    assert info["backend"] == "rccl" and info["gather"] == "rccl" and info["shards"] == [300]
    assert got.tobytes() == plain.tobytes()
    want = ora.motifseq_batch_i16(sig[:40], lens[:40], motif)
    assert np.array_equal(got["dist"][:40], want["dist"]) and np.array_equal(got["end"][:40], want["end"])
    multigpu.close_groups()
    # devices=[0] through the product API (single device: plain path on that device)
    again = api.motifseq_batch(sig, lens, motif, devices=[0])
    assert again.tobytes() == plain.tobytes()
    segs, nsegs = api.segment_batch(sig, lens - 1, devices=[0])
    segs1, nsegs1 = api.segment_batch(sig, lens - 1)
    assert np.array_equal(segs, segs1) and np.array_equal(nsegs, nsegs1)


def test_process_group_rccl_unique_id_one_rank(gpu, tmp_path, monkeypatch):
    """The process-per-GPU shape's RCCL bootstrap (ncclGetUniqueId -> file store -> ncclCommInitRank), world 1."""
    from squigglekit_amd import multigpu
    monkeypatch.setenv("SK_RDZV_DIR", str(tmp_path / "store"))
    with multigpu.ProcessGroup(0, 0, 1) as comm:
        assert comm.backend == "rccl" and comm.ranks_seen() == 1
        assert comm.allgather_host(np.array([7], dtype=np.int64)).tolist() == [[7]]
    assert not (tmp_path / "store").exists()


@pytest.mark.parametrize("scaling,backend", [("weak", "rccl"), ("strong", "rccl"), ("weak", "host")])
def test_bench_gather_path_bare_python(gpu, scaling, backend):
    """`python bench.py` run plainly (no launcher), with the per-step gather forced on at one rank: over RCCL, and by
    host concatenation (what a node without a usable librccl gets; SK_COMM=host)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-comm", "--reads", "20000", "--steps", "2",
           "--warmup", "1", "--cpu-seconds", "2", "--no-extras", "--scaling", scaling]
    env = dict(os.environ, SK_COMM=backend)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert lines[-1].startswith("{"), "the JSON line must be the last thing on stdout: %r" % lines[-3:]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == scaling
    assert line["config"]["gather_backend"] == backend and line["config"]["ranks_seen"] == 1
    assert line["parity"]["dist_bit_identical"] and line["parity"]["start_end_exact"]
    assert line["parity"]["ranks_checked"] == 1 and line["parity"]["every_rank_ok"]       # read out of the gather
    assert "torch" not in p.stderr


def _check_multi_rank_line(line, world):
    assert line["n_gpus"] == world and line["config"]["ranks_seen"] == world
    assert line["config"]["gather_backend"] == "host" and line["config"]["oversubscribed"]
    par = line["parity"]
    assert par["ranks_checked"] == world and par["every_rank_ok"], par
    assert par["dist_bit_identical"] and par["start_end_exact"]
    assert [e["rank"] for e in par["per_rank"]] == list(range(world))
    assert all(e["reads"] >= 64 and e["dist_bit_identical"] and e["start_end_exact"] for e in par["per_rank"])
    assert par["per_rank"][0]["regenerated_equals_resident"]
    assert "gathered buffer" in par["source"]
    # what makes a first real multi-GPU run readable from the one line: who took part, how the gather ran, every
    # rank's own step time and ingest rate
    assert line["ranks_seen"] == world and line["gather_backend"] == "host"
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == world and all(v > 0 for v in pr["ms_per_step"])
    assert max(pr["ms_per_step"]) <= line["ms_per_step"] * 1.001
    assert pr["h2d_GBps"] is None or (len(pr["h2d_GBps"]) == world and all(v > 0 for v in pr["h2d_GBps"]))


@pytest.mark.parametrize("world,scaling,reads", [(2, "weak", 20000), (3, "strong", 30001)])
def test_bench_world_gt1_oversubscribed_threads(gpu, world, scaling, reads):
    """bench.py's world > 1 code on the one-GPU box: `--gpus N --ranks-on-device 0` (one host thread per rank, every
    rank its own context slot on device 0, host-backend gather).  rank_body's shard split, the padded gather, the
    weak run's strong re-slice, the every-rank end-to-end leg -- and rank 0 checks a sample of EVERY rank's shard
    out of the gathered buffer against the oracle."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--ranks-on-device", "0", "--reads",
           str(reads), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--scaling", scaling]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    _check_multi_rank_line(line, world)
    assert line["scaling"] == scaling and line["config"]["launch"].startswith("one process, one host thread")
    if scaling == "weak":
        assert line["config"]["total_reads"] == reads * world and line["strong_scaling"]["total_reads"] == reads
    else:
        # the headline at N > 1 is C4 as BASELINE.json words it (reads in TOTAL); the weak curve is the extra
        assert line["config"]["total_reads"] == reads
        assert line["weak_scaling"]["total_reads"] == reads * world and line["weak_scaling"]["value"] > 0
        assert len(line["weak_scaling"]["per_rank_ms_per_step"]) == world
        assert sum(e["reads"] for e in line["parity"]["per_rank"]) >= 3 * 64
    assert line["end_to_end"]["motifseq_pinned_reads_per_s"] > 0 and len(line["per_rank"]["h2d_GBps"]) == world
    assert "sensitivity" not in line and "secondary" not in line          # N = 1 extras stay out


def test_bench_world2_oversubscribed_process_per_rank(gpu):
    """The driver's N > 1 launch line, two ranks on the one GPU (SK_OVERSUBSCRIBE=1): process-per-GPU shape, file-store
    rendezvous, host-backend gather, every rank's sample checked by rank 0."""
    pytest.importorskip("torch")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "20000",
           "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"]
    env = dict(os.environ, SK_OVERSUBSCRIBE="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    line = json.loads(lines[0])
    _check_multi_rank_line(line, 2)
    assert line["config"]["launch"].startswith("one process per GPU")
    assert line["scaling"] == "strong" and line["config"]["total_reads"] == 20000        # the default: BASELINE's wording
    assert line["weak_scaling"]["total_reads"] == 40000


def test_bench_refuses_more_ranks_than_gpus(gpu):
    """`bench.py --gpus N` with fewer than N GPUs and no SK_OVERSUBSCRIBE: one line of reason, exit code 2 -- never a
    silent run of N ranks on one device that would print a nonsense scaling point."""
    have = gpu.load().sk_device_count()
    env = {k: v for k, v in os.environ.items() if k != "SK_OVERSUBSCRIBE"}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--reads", "4096", "--steps", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 2 and p.stdout.strip() == "", (p.returncode, p.stdout[-300:])
    msg = [ln for ln in p.stderr.strip().splitlines() if ln.startswith("bench.py:")]
    assert len(msg) == 1 and "SK_OVERSUBSCRIBE" in msg[0] and "%d GPU" % have in msg[0]


def test_product_api_two_ranks_on_one_device(gpu, ora, monkeypatch):
    """api.motifseq_batch(devices=[0, 0]) under SK_OVERSUBSCRIBE: the sharded product path with two feeder threads."""
    from squigglekit_amd import api, multigpu, synth
    monkeypatch.setenv("SK_OVERSUBSCRIBE", "1")
    multigpu.close_groups()
    motif = synth.synthetic_motif(150, seed=4)
    sig = synth.squiggle_batch(501, 3000, 97531, motif=motif)
    lens = np.full(501, 3000, dtype=np.int32)
    lens[::5] = 2222
    plain = api.motifseq_batch(sig, lens, motif)
    got = api.motifseq_batch(sig, lens, motif, devices=[0, 0])
    assert got.tobytes() == plain.tobytes()
    got3, info = multigpu.motifseq_sharded(sig, lens, motif, 0, 0, 1200, [0, 0, 0], gather="rccl")
    assert info["shards"] == [167, 167, 167] and info["backend"] == "host" and info["gather"].startswith("host")
    assert got3.tobytes() == plain.tobytes()
    segs, nsegs = api.segment_batch(sig, lens - 1, devices=[0, 0])
    segs1, nsegs1 = api.segment_batch(sig, lens - 1)
    assert np.array_equal(segs, segs1) and np.array_equal(nsegs, nsegs1)
    # the pA (float64) block routes of the tools -- the reference's default input kind -- shard the same way (round 5;
    # before, --gpus N left them on one GPU without a word): a ragged float64 batch with the per-read -n cut, raw rows
    # through the on-device pA conversion, every motif against a ragged batch
    pa = [np.round((sig[r, :lens[r]].astype(np.int64) + 16.0) * (1493.94 / 8192.0), 2) for r in range(501)]
    flat = np.concatenate(pa)
    off = np.concatenate([[0], np.cumsum([x.size for x in pa])]).astype(np.int64)
    cut = np.minimum(lens, 2500).astype(np.int32)
    one = api.segment_ragged_f64(flat, off, cut)
    two = api.segment_ragged_f64(flat, off, cut, devices=[0, 0, 0])
    assert np.array_equal(one[0], two[0]) and np.array_equal(one[1], two[1]) and one[1].sum() > 100
    calib = np.tile(np.array([8192.0, 16.0, 1493.94]), (501, 1))
    one = api.segment_batch_pa(sig, lens, calib)
    two = api.segment_batch_pa(sig, lens, calib, devices=[0, 0])
    assert np.array_equal(one[0], two[0]) and np.array_equal(one[1], two[1]) and one[1].sum() > 100
    motifs = [motif, motif[20:120]]
    one = api.motifseq_multi_ragged_f64(flat, off, motifs)
    two = api.motifseq_multi_ragged_f64(flat, off, motifs, devices=[0, 0])
    assert [h.tobytes() for h in one] == [h.tobytes() for h in two]
    # a rank that fails is not the end of the job (round 5): its block is re-run on the surviving device(s)
    def fault(rank):
        if rank == 1:
            raise gpu.SquiggleKitError(-3, "device lost (injected)")
    monkeypatch.setattr(multigpu, "_fault_hook", fault)
    got = api.motifseq_batch(sig, lens, motif, devices=[0, 0, 0])
    monkeypatch.setattr(multigpu, "_fault_hook", None)
    assert got.tobytes() == plain.tobytes()
    multigpu.close_groups()
    gpu.init(0)                                                       # back to the plain binding for later tests


def test_bench_under_torch_distributed_run_one_rank(gpu):
    """The driver's N > 1 launch line with one rank: `python -m torch.distributed.run ... bench.py`.  The launcher
    only provides the environment; bench.py takes the process-per-GPU shape (forced here, WORLD_SIZE being 1), does
    the ncclUniqueId rendezvous through the file store and gathers over RCCL -- without importing torch."""
    pytest.importorskip("torch")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", "20000",
           "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--no-extras", "--force-comm"]
    env = dict(os.environ, SK_FORCE_PROCESS_SHAPE="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    line = json.loads(lines[0])
    assert line["config"]["launch"].startswith("one process per GPU")
    assert line["config"]["gather_backend"] == "rccl" and line["config"]["ranks_seen"] == 1
    assert line["parity"]["dist_bit_identical"]


@pytest.mark.gpu
def test_rank_thread_binds_to_the_gpus_cpus(gpu):
    """A rank's feeder thread is pinned to the CPUs next to its GPU (local_cpulist of the device's PCI function), only
    that thread, and only when that narrows anything; SK_NUMA_BIND=0 leaves the affinity alone."""
    import threading
    from squigglekit_amd import multigpu
    L = gpu.load()
    buf = C.create_string_buffer(64)
    assert L.sk_device_pci_bus_id(buf, 64) == 0 and buf.value.count(b":") == 2
    before = os.sched_getaffinity(0)
    out = {}

    def body():
        gpu.init(0)
        out["use"] = multigpu.bind_thread_near_device()
        out["after"] = os.sched_getaffinity(0)
    t = threading.Thread(target=body)
    t.start()
    t.join()
    assert os.sched_getaffinity(0) == before                     # the calling thread keeps its own
    if out["use"] is not None:
        assert out["after"] == out["use"] and out["use"] < before
    else:
        assert out["after"] == before
