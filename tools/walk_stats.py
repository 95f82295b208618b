#!/usr/bin/env python3
"""What the segmenter's walk has to do on the bench's C2 data, counted on the CPU (numpy, no GPU): candidate runs per
read, how many of them are long enough to be reported, anchors (in-band samples behind E + 1 out-of-band ones), stretches
of quiet 64-sample entries -- the figures DESIGN.md 4.0b quotes for k_seg_walk4.

    python tools/walk_stats.py [reads=400]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from squigglekit_amd import synth  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    M, E1, window, first = 4000, 6, 150, 38
    sig = synth.squiggle_batch(R, M, synth.SEED_C2)
    runs_n, long38, long150, anchors, stretches, inband = [], [], [], [], [], []
    for r in range(R):
        x = sig[r, :M - 1].astype(np.float64)
        x = x[(x > 0) & (x < 900)]
        med, sd = np.median(x), x.std()
        B = (x < med + 0.75 * sd) & (x > med - 0.75 * sd)
        n = B.size
        zpos = np.flatnonzero(~B)
        i, c = 0, []
        while i < n:                                             # the chain of candidate runs (segmenter.py:429-464, error < corrector)
            nxt = np.flatnonzero(B[i:])
            if nxt.size == 0:
                break
            s = i + int(nxt[0])
            k = int(np.searchsorted(zpos, s))
            if k + E1 - 1 >= zpos.size:
                break
            z = int(zpos[k + E1 - 1])
            c.append(z - s)
            i = z + 1
        c = np.array(c)
        zr = np.concatenate([[0], np.cumsum(~B)])
        a = sum(1 for p in range(E1, n) if B[p] and zr[p] - zr[p - E1] == E1)
        n64 = n // 64
        q = (~B)[:n64 * 64].reshape(n64, 64).sum(1) < E1
        runs_n.append(c.size); long38.append(int((c >= first).sum())); long150.append(int((c >= window).sum()))
        anchors.append(a); stretches.append(int((q & ~np.concatenate([[False], q[:-1]])).sum())); inband.append(B.mean())
    print("reads %d x %d samples (seed C2): in band %.2f of the samples" % (R, M - 1, np.mean(inband)))
    print("candidate runs per read %.1f, of them >= %d samples: %.1f, >= %d samples: %.2f"
          % (np.mean(runs_n), first, np.mean(long38), window, np.mean(long150)))
    print("anchors per read %.1f; stretches of quiet entries per read %.2f (max %d)"
          % (np.mean(anchors), np.mean(stretches), np.max(stretches)))


if __name__ == "__main__":
    main()
