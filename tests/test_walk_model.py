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


# ---------------------------------------------------------------------------------------------------------------
# k_drna_walk_runs (csrc/sk_segment.hip): the dRNA slow5-branch scan by pieces, trip for trip as the kernel takes them
# ---------------------------------------------------------------------------------------------------------------
def _drna_pieces(mask, error, no_err_thresh, w, window, seg_dist):
    """The kernel's loop on a Python bool mask (True = a < top): 64-sample windows at the lane's own position, a piece cut
    at the re-arming sample / at no_err_thresh / at the window's end, close at the (budget + 1)-th out-of-band sample."""
    n = len(mask)
    kw0 = (max(window, w) + w - 1) // w * w
    prev = err = prev_err = start = last_end = refill = 0
    segs = []
    pos = 0
    while pos < n:
        V = min(64, n - pos)
        win = mask[pos:pos + V]
        if not prev:
            ones = np.flatnonzero(win)
            o = int(ones[0]) if ones.size else V
            if segs and o > 0 and pos + o - 1 - last_end > seg_dist:
                break                                            # adapter found (:152)
            if o == V:
                pos += V
                continue
            pos += o
            win = win[o:]
            V -= o
            prev, start, err, prev_err = 1, pos, 0, 0
            refill = start + kw0 - 1
        L = min(V, refill - pos + 1)
        counted = pos >= no_err_thresh
        if not counted:
            L = min(L, no_err_thresh - pos)
        piece = win[:L]
        zs = np.flatnonzero(~piece)
        tol = max(error - err, 0) if counted else (64 if err < error else 0)
        if zs.size > tol:
            q = int(zs[tol])
            before = np.flatnonzero(piece[:q])
            if before.size:
                prev_err = int((~piece[int(before[-1]) + 1:q]).sum()) if counted else 0
            else:
                prev_err += tol if counted else 0
            i = pos + q
            if i - start >= window:
                end = i - prev_err
                if segs and start - last_end < seg_dist:
                    segs[-1][1] = end
                else:
                    segs.append([start, end])
                last_end = end
            prev = err = prev_err = 0
            pos = i + 1
        else:
            ones = np.flatnonzero(piece)
            if ones.size:
                prev_err = int((~piece[int(ones[-1]) + 1:]).sum()) if counted else 0
            else:
                prev_err += int(zs.size) if counted else 0
            err += int(zs.size) if counted else 0
            pos += L
            if pos - 1 == refill:
                err -= 1
                refill += w
    return segs


@pytest.mark.parametrize("seed", range(4))
def test_drna_scan_by_pieces_model_equals_oracle(ora, seed):
    from squigglekit_amd import synth
    rng = np.random.default_rng(500 + seed)
    reads = synth.drna_reads(6, 900 + seed, min_len=3000, max_len=16000) + \
        [x for x in synth.pattern_reads(rng, 6, int(rng.choice([3000, 9000])))]
    # the stop test at its boundary: a segment, exactly seg_dist idle out-of-band samples, another segment
    reads.append(np.r_[np.full(300, 400), np.full(51, 700), np.full(200, 400), np.full(6000, 700)].astype(np.int16))
    total = 0
    for kw in (dict(), dict(error=0), dict(w=64, window=500, seg_dist=50), dict(no_err_thresh=100000, error=1),
               dict(error=9, w=100, window=250, seg_dist=10, std_scale=1.5), dict(window=0, seg_dist=0, w=65),
               dict(no_err_thresh=777, w=128, window=64, error=3, t_start=0, t_end=9000),
               dict(no_err_thresh=0, error=0, seg_dist=50, window=100), dict(no_err_thresh=0, error=0, seg_dist=49, window=100)):
        p = ora.DrnaParams(**kw)
        for x in reads:
            f = ora.scale_outliers(x.astype(float), 0, 1200)
            want, top = ora.drna_segs(f, p, max_segs=16384)
            got = _drna_pieces(f < top, p.error, p.no_err_thresh, p.w, p.window, p.seg_dist)
            assert got == want, (seed, kw, len(f))
            total += len(got)
    assert total > 50
