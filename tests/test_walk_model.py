"""CPU: the claims k_seg_walk4 (csrc/sk_segstat.hip) rests on, checked on a sample-level Python model against the
oracle's get_segs -- independent of the kernel's bit tricks.  The model walks in RAW coordinates (dropped samples are
skipped and counted), notes the stretches of quiet 64-sample entries with their anchors exactly as the statistics
kernel's hints define them, and after the first segment jumps from stretch to stretch.  If the jumps could skip a run
that matters, or the drop bookkeeping moved a boundary, its segments would differ from the oracle's on the filtered
signal."""
import numpy as np
import pytest


def _hints(O, Z, K, E1):
    """Stretches [ka, kb] of quiet entries, each with the newest anchor in entries <= ka - 2 (else sample 0) and the
    samples dropped before the anchor's entry."""
    n = len(O)
    nent = (n + 63) // 64
    quiet = [int(Z[64 * e:64 * e + 64].sum()) < E1 for e in range(nent)]
    drops_before = [int((~K[:64 * e]).sum()) for e in range(nent + 1)]
    anchors = []                                                 # per entry: newest anchor inside it, or None
    for e in range(nent):
        best = None
        for i in range(64 * e, min(64 * e + 64, n)):
            if O[i] and i >= E1 and Z[i - E1:i].all():           # E1 raw predecessors all kept and out of band
                best = i
        anchors.append(best)
    out = []
    e = 0
    while e < nent:
        if quiet[e] and (e == 0 or not quiet[e - 1]):
            kb = e
            while kb + 1 < nent and quiet[kb + 1]:
                kb += 1
            a = 0
            for k in range(e - 2, -1, -1):
                if anchors[k] is not None:
                    a = anchors[k]
                    break
            out.append((a, kb, drops_before[a // 64]))
            e = kb + 1
        else:
            e += 1
    return out


def _jump_walk(inband, kept, error, window, seg_dist, first_len, jumps=True):
    """k_seg_walk4's algorithm, one sample at a time where the kernel takes 64: segments in FILTERED coordinates."""
    O, Z, K = inband & kept, ~inband & kept, kept
    n, E1 = len(O), max(error, 0) + 1
    stretches = _hints(O, Z, K, E1) if jumps else []
    segs, last_end, thr = [], 0, min(window, first_len)
    pos, item = 0, 0
    dropped_before = np.concatenate([[0], np.cumsum(~K)])
    while pos < n:
        if jumps and segs:
            while item < len(stretches) and stretches[item][1] * 64 + 63 < pos:
                item += 1
            if item == len(stretches):
                break
            a, _kb, d = stretches[item]
            assert d == dropped_before[64 * (a // 64)]
            if a > pos:
                pos = a
        nxt = np.flatnonzero(O[pos:])
        if nxt.size == 0:
            break
        s = pos + int(nxt[0])
        zs = np.flatnonzero(Z[s:])
        if zs.size < E1:
            break                                                # the run is still open at the end: dropped (:466)
        z = s + int(zs[E1 - 1])
        c = int(K[s:z].sum())                                    # kept samples of [s, z)
        ones = np.flatnonzero(O[s:z])
        prev_err = int(Z[s + int(ones[-1]) + 1:z].sum())         # out-of-band samples since the last in-band one
        if c >= thr:
            zf = z - int(dropped_before[z])
            start, end = zf - c, zf - prev_err
            if segs and start - last_end < seg_dist:
                segs[-1][1] = end
            else:
                segs.append([start, end])
            last_end, thr = end, window
        pos = z + 1
    return [tuple(x) for x in segs]


def _masks(rng, n):
    """In-band / kept masks with stalls, alternating stretches, holes, trains and noise, and a few dropped samples."""
    from squigglekit_amd import synth
    x = synth.pattern_reads(rng, 1, n)[0].astype(np.float64)
    kept = (x > 0) & (x < 900)
    f = x[kept]
    med, sd = np.median(f), f.std()
    inband = (x < med + 0.75 * sd) & (x > med - 0.75 * sd)
    return x, inband, kept


def _crafted(rng, E):
    """A signal whose mask puts gaps of exactly E - 1 .. E + 2 out-of-band samples right in front of long stalls, behind
    a run that has not used any of its error budget: what tells an anchor (E + 1 in a row) from a near miss."""
    bits = [np.ones(int(rng.integers(160, 260)), bool), np.zeros(int(rng.integers(60, 300)), bool)]
    for _ in range(int(rng.integers(3, 7))):
        on, off = int(rng.integers(2, 6)), int(rng.integers(1, 3))
        ln = int(rng.integers(100, 500))
        alt = (np.arange(ln) % (on + off)) < on                             # never quiet, never E + 1 in a row
        run = np.ones(int(rng.integers(8, 60)), bool)                       # a run with its whole budget left ...
        gap = np.zeros(int(rng.integers(max(E - 1, 1), E + 3)), bool)       # ... meets E - 1 .. E + 2 of them ...
        # ... right in front of the stall, or two entries and more before it (only then can the sample behind the gap
        # become the stall's anchor)
        bits += [alt, run, gap] if rng.random() < 0.4 else [run, gap, alt]
        stall = np.ones(int(rng.integers(140, 420)), bool)                  # ... in front of a stall
        stall[rng.integers(0, stall.size, int(rng.integers(0, 4)))] = False
        bits.append(stall)
        if rng.random() < 0.5:
            bits.append(np.zeros(int(rng.integers(E + 1, 80)), bool))
    b = np.concatenate(bits)[:4096]
    n = b.size
    x = np.where(b, 500.0 + rng.integers(-5, 6, n), np.where(np.arange(n) % 2 == 0, 300.0, 700.0))
    x[rng.integers(0, n, int(rng.integers(0, 6)))] = 1000.0                 # dropped samples
    kept = (x > 0) & (x < 900)
    f = x[kept]
    med, sd = np.median(f), f.std()
    return x, (x < med + 0.75 * sd) & (x > med - 0.75 * sd), kept


@pytest.mark.parametrize("seed", range(4))
def test_jumping_walk_model_on_near_miss_anchors(ora, seed):
    rng = np.random.default_rng(77 + seed)
    total = 0
    for _ in range(40):
        for kw in (dict(), dict(error=3), dict(error=8, corrector=30)):
            p = ora.SegParams(**kw)
            x, inband, kept = _crafted(rng, p.error)
            want = ora.get_segs(x[kept], p) or []
            got = _jump_walk(inband, kept, p.error, p.window, p.seg_dist, int(np.ceil(p.window * p.stall_len)))
            assert got == [tuple(s) for s in want], (seed, kw)
            total += len(got)
    assert total > 100


@pytest.mark.parametrize("seed", range(6))
def test_jumping_walk_model_equals_get_segs(ora, seed):
    rng = np.random.default_rng(1000 + seed)
    total = jumped = 0
    for _ in range(25):
        n = int(rng.choice([700, 2048, 4000, 4096]))
        x, inband, kept = _masks(rng, n)
        for kw in (dict(), dict(window=127, stall_len=0.05), dict(error=0), dict(error=9, corrector=40, seg_dist=0),
                   dict(window=300, stall_len=1.5)):
            p = ora.SegParams(**kw)
            want = ora.get_segs(x[kept], p) or []
            first_len = int(np.ceil(p.window * p.stall_len))
            got = _jump_walk(inband, kept, p.error, p.window, p.seg_dist, first_len)
            assert got == [tuple(s) for s in want], (seed, n, kw)
            plain = _jump_walk(inband, kept, p.error, p.window, p.seg_dist, first_len, jumps=False)
            assert plain == got
            total += len(got)
            jumped += len(_hints(inband & kept, ~inband & kept, kept, p.error + 1))
    assert total > 20 and jumped > 20                            # the cases do produce segments and stretches to jump between
