#!/usr/bin/env python3
"""Parity at scale: the whole MotifSeq hot path (filter + medmad + subsequence DTW) on a GPU against the
oracle on the host's cores, read for read, on a C4-shaped batch far larger than the unit tests use.

    python tools/parity_at_scale.py [reads=200000] [threads=64] [samples=4000] [motif=200]
    python tools/parity_at_scale.py segmenter [reads=200000] [threads=64] [samples=4000]

Exit code 0 only if start, end, n are equal and the distances bit-identical for every read
(segmenter: every read's segment list equal).
(Test infrastructure: the oracle is the checker here, never the product path.)"""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
from squigglekit_amd import api, synth          # noqa: E402
from oracle import oracle as ora                 # noqa: E402


def segmenter(argv):
    R = int(argv[0]) if len(argv) > 0 else 200000
    T = int(argv[1]) if len(argv) > 1 else 64
    M = int(argv[2]) if len(argv) > 2 else 4000
    sig = synth.squiggle_batch(R, M, synth.SEED_C2)
    lens = np.full(R, M - 1, dtype=np.int32)                  # Num = -1
    t0 = time.perf_counter()
    segs, nsegs = api.segment_batch(sig, lens)
    t_gpu = time.perf_counter() - t0
    per = (R + T - 1) // T
    parts = [(i, min(R, i + per)) for i in range(0, R, per)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        res = list(ex.map(lambda ab: ora.segment_batch_i16(sig[ab[0]:ab[1]], lens[ab[0]:ab[1]],
                                                           max_segs=segs.shape[1]), parts))
    t_cpu = time.perf_counter() - t0
    osegs = np.concatenate([x[0] for x in res])
    onsegs = np.concatenate([x[1] for x in res])
    same_n = np.array_equal(nsegs, onsegs)
    keep = np.arange(segs.shape[1])[None, :] < nsegs[:, None]
    same_s = same_n and np.array_equal(segs[keep], osegs[keep])
    print("segmenter, reads %d x %d: segment counts equal %s, boundaries equal %s (%d segments); "
          "GPU call (host buffers) %.2f s, oracle on %d threads %.1f s"
          % (R, M, same_n, same_s, int(nsegs.sum()), t_gpu, T, t_cpu))
    sys.exit(0 if same_s else 1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "segmenter":
        return segmenter(sys.argv[2:])
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    motif = synth.synthetic_motif(N)
    sig = synth.squiggle_batch(R, M, synth.SEED_C4, motif=motif)
    lens = np.full(R, M, dtype=np.int32)
    t0 = time.perf_counter()
    got = api.motifseq_batch(sig, lens, motif)
    t_gpu = time.perf_counter() - t0
    per = (R + T - 1) // T
    parts = [(i, min(R, i + per)) for i in range(0, R, per)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:            # ctypes calls release the GIL
        want = np.concatenate(list(ex.map(
            lambda ab: ora.motifseq_batch_i16(sig[ab[0]:ab[1]], lens[ab[0]:ab[1]], motif, scale_mode=0), parts)))
    t_cpu = time.perf_counter() - t0
    same_se = np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"])
    same_n = np.array_equal(got["n"], want["n"])
    same_d = np.array_equal(got["dist"], want["dist"])
    print("reads %d x %d, motif %d: start/end equal %s, n equal %s, dist bit-identical %s; "
          "GPU call (host buffers) %.2f s, oracle on %d threads %.1f s"
          % (R, M, N, same_se, same_n, same_d, t_gpu, T, t_cpu))
    sys.exit(0 if (same_se and same_n and same_d) else 1)


if __name__ == "__main__":
    main()
