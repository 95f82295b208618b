"""Read sharding across GPUs and the one exchange the path has: the final gather of the
fixed-size result records.

Reads are independent in both scripts (the per-read loops carry no state:
segmenter.py:189-230, MotifSeq.py:261-298), so a job is a contiguous block split of the
reads, one process per GPU, and no data-path collective.  The gather is done by whatever
`torch.distributed`-shaped object the launcher hands in (backend "nccl" == RCCL on ROCm,
"gloo" in the CPU tests); this module itself imports neither torch nor any collective
library -- it only sees the `dist` handle and tensors it is given.
"""
import numpy as np

HIT_BYTES = 24          # sizeof(sk_hit)


def shard_bounds(total, rank, world):
    """Contiguous block split: rank r owns reads [lo, hi).  Sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(total, world):
    return [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]


def gather_records(dist, local, world, pad_to=None):
    """All-gather byte tensors of per-rank records into rank order.

    local : 1-D uint8 tensor (device for nccl, cpu for gloo), this rank's records
    pad_to: bytes every rank pads to (ranks may own one record more or less); default = len(local)
    Returns a list of `world` uint8 tensors (each `pad_to` long; trim with shard_sizes)."""
    n = int(local.numel())
    pad_to = n if pad_to is None else int(pad_to)
    if pad_to < n:
        raise ValueError("pad_to smaller than the local buffer")
    if pad_to != n:
        buf = local.new_zeros(pad_to)
        buf[:n] = local
    else:
        buf = local
    outs = [buf.new_empty(pad_to) for _ in range(world)]
    dist.all_gather(outs, buf)
    return outs


def assemble_hits(parts, sizes, dtype):
    """Concatenate gathered byte tensors (one per rank) into one record array in read order."""
    chunks = []
    for t, k in zip(parts, sizes):
        raw = t.cpu().numpy()[: k * dtype.itemsize]
        chunks.append(np.frombuffer(raw.tobytes(), dtype=dtype))
    return np.concatenate(chunks) if chunks else np.zeros(0, dtype=dtype)
