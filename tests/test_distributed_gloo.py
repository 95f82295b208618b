"""CPU, world_size 2, gloo: the N > 1 structure of the job -- block sharding of reads, no data-path
collective, one gather of fixed-size hit records -- gives the same answer in the same order as
the unsharded job.  Each rank computes its shard with the oracle (the GPU is not needed to test
the sharding/gather logic; on the GPU box the same helpers run over RCCL inside bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as ora
    from squigglekit_amd import sharding, synth
    from squigglekit_amd._lib import HIT_DTYPE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    motif = synth.synthetic_motif(40)
    sig = synth.squiggle_batch(total, 600, 4321, motif=motif)          # every rank can regenerate the job
    lo, hi = sharding.shard_bounds(total, rank, world)
    lens = np.full(hi - lo, 600, dtype=np.int32)
    local = ora.motifseq_batch_i16(sig[lo:hi], lens, motif)            # this rank's shard only
    assert local.dtype == HIT_DTYPE
    sizes = sharding.shard_sizes(total, world)
    t = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy())
    parts = sharding.gather_records(dist, t, world, pad_to=max(sizes) * HIT_DTYPE.itemsize)
    full = sharding.assemble_hits(parts, sizes, HIT_DTYPE)
    dist.barrier()
    if rank == 0:
        want = ora.motifseq_batch_i16(sig, np.full(total, 600, dtype=np.int32), motif)
        q.put((full.tobytes() == want.tobytes(), int(full.size)))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [17, 32])
def test_sharded_job_equals_unsharded(total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    same, n = q.get(timeout=10)
    assert same and n == total


def test_shard_bounds_cover_everything():
    from squigglekit_amd import sharding
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(total, world)
