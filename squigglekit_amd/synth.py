"""Deterministic synthetic squiggle batches (SURVEY.md section 8(d) recipe).

Pure numpy, host side; used by tests, tools/gen_golden.py and bench.py to make
inputs of the shapes BASELINE.json names.  No reference code involved: the
reference ships no generator (its only data is example/test.fast5).

Squiggle model: event levels ~ N(500, 80) raw units, dwell 1 + Poisson(8)
samples, additive N(0, 8) noise, rounded and clipped to int16; a stall plateau
near the start, with p = 0.5 a second plateau later in the read, and four spike
samples from {-5, 0, 950, 1100} so the outlier filter has work to do.
"""
import numpy as np

SEED_C2 = 20260927   # segmenter 10 000 x 4 000
SEED_C3 = 20260928   # MotifSeq 10 000 x 4 000 vs example model
SEED_C4 = 20260929   # MotifSeq 1 M x 4 000 (+ rank)
SEED_C5 = 20260930   # MotifSeq 100 k x 20 000 (+ rank)

SPIKES = np.array([-5, 0, 950, 1100], dtype=np.int16)


def squiggle_batch(n_reads, n_samples, seed, motif=None, chunk=8192):
    """int16 [n_reads, n_samples] synthetic squiggles.

    motif: optional float vector (normalised units); round(motif*93.4 + 511) is
    implanted at a random offset in ~50 % of the reads (positive controls).
    """
    out = np.empty((n_reads, n_samples), dtype=np.int16)
    rng = np.random.default_rng(seed)
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        out[lo:hi] = _chunk(rng, hi - lo, n_samples, motif)
    return out


def _chunk(rng, R, M, motif):
    E = M // 4 + 16                                   # events: mean dwell 9 >> 4
    levels = rng.normal(500.0, 80.0, size=(R, E))
    dwell = 1 + rng.poisson(8.0, size=(R, E))
    starts = np.cumsum(dwell, axis=1)                 # first sample of event k+1
    mark = np.zeros((R, M + 1), dtype=np.int32)
    np.put_along_axis(mark, np.minimum(starts, M), 1, axis=1)
    # several events can clip to column M; that column is dropped
    ev = np.cumsum(mark[:, :M], axis=1)
    sig = np.take_along_axis(levels, ev, axis=1)
    sig += rng.normal(0.0, 8.0, size=(R, M))

    col = np.arange(M)[None, :]
    # stall plateau
    s0 = rng.integers(0, 60, size=(R, 1))
    l0 = rng.integers(100, 600, size=(R, 1))
    m0 = (col >= s0) & (col < s0 + l0)
    sig = np.where(m0, rng.normal(505.0, 12.0, size=(R, M)), sig)
    # optional second plateau (positions scale with read length past 4 000)
    scale = max(1, M // 4000)
    has2 = rng.random(size=(R, 1)) < 0.5
    s1 = rng.integers(1200 * scale, 3400 * scale, size=(R, 1))
    l1 = rng.integers(160, 500, size=(R, 1))
    m1 = has2 & (col >= s1) & (col < s1 + l1)
    sig = np.where(m1, rng.normal(495.0, 10.0, size=(R, M)), sig)

    sig = np.clip(np.rint(sig), -32768, 32767).astype(np.int16)

    if motif is not None:
        mot = np.clip(np.rint(np.asarray(motif, dtype=np.float64) * 93.4 + 511.0),
                      -32768, 32767).astype(np.int16)
        N = mot.size
        if N < M:
            hit = rng.random(size=R) < 0.5
            off = rng.integers(0, M - N, size=R)
            rows = np.nonzero(hit)[0]
            idx = off[rows, None] + np.arange(N)[None, :]
            sig[rows[:, None], idx] = mot[None, :]

    # four spikes per read
    pos = rng.integers(0, M, size=(R, 4))
    val = SPIKES[rng.integers(0, 4, size=(R, 4))]
    np.put_along_axis(sig, pos, val, axis=1)
    return sig


def synthetic_motif(n_points, seed=7):
    """A seeded k-mer-like level sequence expanded by dwell (normalised units),
    shaped like what MotifSeq.py:354-379 builds from a scrappie model."""
    rng = np.random.default_rng(seed)
    vals = []
    while len(vals) < n_points:
        level = float(np.float32(rng.normal(0.0, 1.1)))
        vals.extend([level] * int(rng.integers(6, 13)))
    return np.asarray(vals[:n_points], dtype=np.float64)


def drna_reads(n_reads, seed, min_len=6000, max_len=40000):
    """Ragged dRNA-like int16 reads for the dRNA adapter segmenter (dRNA_segmenter.py): a low
    adapter stretch, a tight poly(A) plateau, then an event-level body; a few outlier spikes."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_reads):
        n = int(rng.integers(min_len, max_len))
        la = int(rng.integers(1500, 6000))
        lp = int(rng.integers(300, 3000))
        body = squiggle_batch(1, max(8, n - la - lp), int(rng.integers(1, 2 ** 31)))[0].astype(np.float64) + 30.0
        sig = np.concatenate([rng.normal(430.0, 25.0, la), rng.normal(560.0, 8.0, lp), body])[:n]
        # short excursions above the band inside the adapter (exercise err / prev_err / merging)
        for _k in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, max(1, la - 40)))
            sig[p:p + int(rng.integers(1, 30))] = 640.0
        pos = rng.integers(0, sig.size, 6)
        sig[pos] = SPIKES[rng.integers(0, 4, 6)].astype(np.float64) + rng.integers(0, 2, 6) * 300
        out.append(np.clip(np.rint(sig), -32768, 32767).astype(np.int16))
    return out


def pattern_reads(rng, R, M):
    """Reads whose in-band mask is built from pieces chosen to stress the jumping walk (k_seg_walk4): quiet stalls,
    alternating stretches (never quiet, never E + 1 out-of-band samples in a row: no anchor for thousands of samples),
    long out-of-band holes, noise, 70-in / 10-out trains (isolated quiet entries), plus a few dropped samples."""
    sig = np.empty((R, M), dtype=np.int16)
    for r in range(R):
        bits = []
        total = 0
        while total < M:
            kind = rng.integers(0, 8) if r % 4 else rng.choice([0, 1, 1, 4, 7])
            ln = int(rng.integers(20, 2600 if kind == 1 else 700))
            if kind == 0:                                        # a stall: few out-of-band samples
                b = (rng.random(ln) > rng.choice([0.0, 0.01, 0.03])).astype(np.uint8)
            elif kind == 1:                                      # alternating, period 2 .. 5
                per = int(rng.integers(2, 6))
                b = (np.arange(ln) % per != 0).astype(np.uint8) if rng.random() < 0.5 else (np.arange(ln) % per == 0).astype(np.uint8)
            elif kind == 2:                                      # a hole
                b = np.zeros(ln, dtype=np.uint8)
            elif kind == 3:                                      # noise
                b = (rng.random(ln) < rng.uniform(0.3, 0.9)).astype(np.uint8)
            elif kind == 4:                                      # trains of in-band samples between short gaps
                on, off = int(rng.integers(30, 140)), int(rng.integers(1, 12))
                b = ((np.arange(ln) % (on + off)) < on).astype(np.uint8)
            elif kind == 5:                                      # event-like: runs of random length
                b = np.repeat(rng.random(ln // 6 + 1) < 0.6, rng.integers(3, 14, ln // 6 + 1))[:ln].astype(np.uint8)
            elif kind == 6:                                      # exactly E, E + 1, E + 2 out-of-band samples between runs
                gap = int(rng.integers(4, 9))
                on = int(rng.integers(5, 60))
                b = ((np.arange(ln) % (on + gap)) < on).astype(np.uint8)
            else:                                                # a near-miss anchor: a run with its whole error budget left, a gap
                # of 4 .. 7 samples (E + 1 = 6 makes an anchor), two entries and more that are neither quiet nor hold
                # an anchor, then a stall -- the sample behind the gap is the newest candidate for the stall's anchor
                on, off = int(rng.integers(2, 6)), int(rng.integers(1, 3))
                alt = ((np.arange(int(rng.integers(130, 420))) % (on + off)) < on).astype(np.uint8)
                stall = np.ones(int(rng.integers(140, 420)), dtype=np.uint8)
                stall[rng.integers(0, stall.size, int(rng.integers(0, 4)))] = 0
                b = np.concatenate([np.ones(int(rng.integers(8, 60)), dtype=np.uint8),
                                    np.zeros(int(rng.integers(4, 8)), dtype=np.uint8), alt, stall])
            bits.append(b)
            total += len(b)
        b = np.concatenate(bits)[:M]
        far = np.where(np.arange(M) % 2 == 0, 300, 700)
        x = np.where(b == 1, 500 + rng.integers(-12, 13, M), far + rng.integers(-12, 13, M))
        if r % 3 == 0:                                           # dropped samples: the squeeze moves every later position
            k = int(rng.integers(1, 40))
            x[rng.integers(0, M, k)] = rng.choice([0, 950, -7])
        sig[r] = x
    return sig
