"""GPU: two host threads bound to the SAME context slot (SURVEY 8(b) "Threading": per-device context guarded by a mutex).

Every compute / upload / download entry point of csrc/sk_api.hip holds its context's lock (sk_ctx_guard, csrc/sk_common.h)
from sk_cur() to its return, so calls of different threads on one slot take turns: the scratch buffers one call's
sk_reserve may free, the event slots and the "last call" counters belong to one call at a time.  Before round 6 only
sk_init_slot / sk_shutdown took a lock and two threads on one slot corrupted each other's records (round 4's advisor
finding; it had been fixed by ordering in motifseq_cli.py, not in the library).

Reference: the loops these calls replace are single threaded (/root/reference/MotifSeq.py:261-298,
/root/reference/segmenter.py:189-230); what every record must equal is their per-read result, i.e. the oracle's."""
import threading

import numpy as np
import pytest

from conftest import oracle_motifseq_threaded

pytestmark = pytest.mark.gpu

ITER = 200


def test_two_threads_on_one_slot_take_turns(gpu, ora):
    from squigglekit_amd import _lib, api, synth
    motif = synth.synthetic_motif(200)
    rng = np.random.default_rng(606)
    # MotifSeq thread: two batch shapes alternate, so that sk_reserve regrows / the kernels' scratch layout changes
    # between calls while the other thread's call may be in flight
    shapes = [(700, 4000), (300, 6000)]
    m_in, m_want = [], []
    for k, (R, M) in enumerate(shapes):
        sig = synth.squiggle_batch(R, M, 77 + k, motif=motif)
        lens = rng.integers(M // 2, M + 1, R).astype(np.int32)
        m_in.append((sig, lens))
        m_want.append(oracle_motifseq_threaded(ora, sig, lens, motif))
    # segmenter thread: float64 pA reads (ragged), two batch shapes as well
    s_in, s_want = [], []
    for k, (R, M) in enumerate([(96, 3999), (40, 9000)]):
        raw = synth.squiggle_batch(R, M, 177 + k)
        reads = [np.round((raw[r, :int(n)].astype(np.int64) + 16.0) * (1493.94 / 8192.0), 2)
                 for r, n in enumerate(rng.integers(M // 2, M + 1, R))]
        s_in.append(reads)
        s_want.append([ora.get_segs(ora.scale_outliers(x, 0, 900)) for x in reads])

    errors = []
    start = threading.Barrier(2)

    def motifseq_thread():
        try:
            _lib.init(0)                                      # the same slot as the other thread
            start.wait(120)
            for it in range(ITER):
                k = it & 1
                got = api.motifseq_batch(m_in[k][0], m_in[k][1], motif, scale="medmad")
                want = m_want[k]
                same = ((got["start"] == want["start"]) & (got["end"] == want["end"]) & (got["n"] == want["n"])
                        & ((got["dist"] == want["dist"]) | (np.isnan(got["dist"]) & np.isnan(want["dist"]))))
                if not same.all():
                    errors.append("MotifSeq iteration %d: %d of %d records differ from the oracle" % (it, int((~same).sum()), same.size))
                    return
        except BaseException as e:                            # noqa: BLE001 -- reported by the main thread
            errors.append("MotifSeq thread: %r" % (e,))

    def segmenter_thread():
        try:
            _lib.init(0)
            start.wait(120)
            for it in range(ITER):
                k = it & 1
                got = api.segment_reads_f64(s_in[k])
                if got != s_want[k]:
                    bad = [r for r in range(len(got)) if got[r] != s_want[k][r]]
                    errors.append("segmenter iteration %d: reads %s differ from the oracle" % (it, bad[:8]))
                    return
        except BaseException as e:                            # noqa: BLE001
            errors.append("segmenter thread: %r" % (e,))

    ts = [threading.Thread(target=motifseq_thread), threading.Thread(target=segmenter_thread)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a thread is stuck (deadlock between context locks?)"
    assert not errors, errors
    g = api.last_dtw_guard()
    assert g["alarm"] == 0, g
