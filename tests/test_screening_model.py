"""CPU: the numbers the DTW screening certificate rests on, derived by brute force instead of by comment
(csrc/sk_sdtwq.hip:8-29, :285-320).  No GPU, no library: exact rational arithmetic against a Python model
of what pass Q computes.

The default DTW path replaces mlpy's exact FP64 sweep (/root/reference/MotifSeq.py:437-439) by a 32-bit fixed-point one
(1 unit = 2^-22) and certifies an exact window from it.  That is exact only if
  (1) every sample IMAGE -- rint(fma(x, qa, qb)) for int16 reads, rint((x - c) * qa) for float64 reads -- is within
      1/2 + imgerr units of the exact normalised value (x - c) / s, with imgerr = 3.8e-7 + |c| qa 1.2e-16 (5.7e-7 for
      the difference form) as k_sdtw_q assumes when it decides whether a read may be screened, and
  (2) every cell of the fixed-point cost matrix is then within E = N + n + 2 units of the exact one, so that
      (Dq - E) is a lower bound and every column that can hold the exact minimum lies within 2 E of the screening
      minimum (the candidate rule of pass P).
(1) is checked against exact rationals over the extremes the kernel comment names; (2) on small matrices cell by cell,
with the kernel's own saturating recurrence.  The rolling-mean branch has the same kind of test (tests/test_drna.py)."""
from fractions import Fraction

import numpy as np

QSCALE = 4194304.0            # 2^22           (csrc/sk_sdtw_dev.h)
QLIM = 400.0
QSAFE = 0xF0000000           # a minimum at or above this may have saturated: exact pass


def fma(a, b, c):
    """one correctly rounded a * b + c (float(Fraction) rounds to nearest even)"""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def image_terms(center, scale):
    inv_scale = 1.0 / scale                       # k_sdtw_q: const double inv_scale = 1.0 / scale
    qa = inv_scale * QSCALE                       # exact (a power of two)
    qb = -center * qa
    return qa, qb


def test_sample_image_error_bound_int16_fma_form():
    """|fma(x, qa, qb) - (x - c) / s * 2^22| <= 3.8e-7 + |c| qa 1.2e-16 units whenever |t| < 400 * 2^22 -- random reads
    and the corners: the largest level over the smallest MAD, |t| at the limit, half-integer medians"""
    rng = np.random.default_rng(1)
    worst = 0.0
    cases = []
    for _ in range(4000):
        c = float(rng.integers(-65536, 65537)) / 2.0                      # np.median of integers: k / 2
        mad = float(rng.integers(1, 4000)) / 2.0
        cases.append((c, mad * 1.4826))
    cases += [(32767.0, 0.5 * 1.4826), (-32768.0, 0.5 * 1.4826), (30000.0, 1.4826), (511.0, 63.0 * 1.4826),
              (0.5, 0.5 * 1.4826), (32767.5, 100.0 * 1.4826), (-0.5, 3000.0 * 1.4826)]
    checked = 0
    for c, s in cases:
        qa, qb = image_terms(c, s)
        bound = 3.8e-7 + abs(c) * qa * 1.2e-16
        lim = QLIM * s                                                    # |x - c| beyond this leaves the fixed-point range
        xs = set(int(v) for v in rng.integers(-32768, 32768, 12))
        for d in (lim, -lim, lim * 0.999, -lim * 0.999, 0.0, 1.0, -1.0):  # the range's ends, the centre
            xs.add(int(max(-32768, min(32767, round(c + d)))))
        for x in xs:
            t = fma(float(x), qa, qb)
            if not abs(t) < QLIM * QSCALE:
                continue                                                  # (the kernel sends such a read to the exact pass)
            exact = (Fraction(x) - Fraction(c)) / Fraction(s) * Fraction(QSCALE)
            err = abs(Fraction(t) - exact)
            assert err <= Fraction(bound), (c, s, x, float(err), bound)
            worst = max(worst, float(err) / bound)
            checked += 1
    assert checked > 40000 and worst > 0.05                               # the bound is tight to about a factor of 20, not vacuous


def test_sample_image_error_bound_float64_difference_form():
    """float64 reads: t = fl(fl(x - c) * qa), three roundings of a value below 400 * 2^22: <= 5.7e-7 units -- including the
    near-constant reads whose fma form was the round-4 hole (c / s ~ 1e14)"""
    rng = np.random.default_rng(2)
    checked = 0
    for k in range(3000):
        if k % 3 == 0:                                                    # near-constant: spread 1e-14 of the level
            c = float(rng.uniform(50, 1000))
            s = float(rng.uniform(0.5, 4)) * 2.0 ** -40
        elif k % 3 == 1:                                                  # pA-like
            c = round(float(rng.uniform(40, 200)), 2)
            s = float(rng.uniform(1, 40))
        else:
            c = float(rng.normal(0, 1e4))
            s = float(abs(rng.normal(0, 50)) + 1e-3)
        qa, _ = image_terms(c, s)
        for _ in range(10):
            x = c + float(rng.uniform(-QLIM, QLIM)) * s * float(rng.choice([1.0, 0.99999, 1e-3]))
            t = (x - c) * qa
            if not abs(t) < QLIM * QSCALE:
                continue
            exact = (Fraction(x) - Fraction(c)) / Fraction(s) * Fraction(QSCALE)
            assert abs(Fraction(t) - exact) <= Fraction(5.7e-7), (c, s, x)
            checked += 1
    assert checked > 20000


def qimg(t):
    """rint(t) as the kernel makes it (adding 1.5 * 2^52), biased to unsigned"""
    return int(np.rint(t)) + 0x80000000


def screening_matrix(xq, yq):
    """pass Q's recurrence: nw = min(|xq - yq| + min3(diag, left, up), 2^32 - 1), row 0 free to start anywhere"""
    N, n = len(xq), len(yq)
    INF = 0xFFFFFFFF
    D = [[0] * n for _ in range(N)]
    for j in range(n):
        for i in range(N):
            c = abs(xq[i] - yq[j])
            if i == 0:
                best = 0
            elif j == 0:
                best = D[i - 1][0]
            else:
                best = min(D[i - 1][j - 1], D[i][j - 1], D[i - 1][j])
            D[i][j] = min(c + best, INF)
    return D


def exact_matrix(x, y):
    """mlpy's subsequence cost matrix in float64 (oracle/sk_oracle.c restates it); the tests compare in rationals"""
    N, n = len(x), len(y)
    D = np.zeros((N, n))
    for j in range(n):
        for i in range(N):
            c = abs(x[i] - y[j])
            if i == 0:
                D[i, j] = c
            elif j == 0:
                D[i, j] = c + D[i - 1, 0]
            else:
                D[i, j] = c + min(D[i - 1, j - 1], D[i, j - 1], D[i - 1, j])
    return D


def test_every_screening_cell_is_within_E_of_the_exact_one_and_the_candidate_rule_holds():
    """|Dq - D * 2^22| <= E = N + n + 2 in EVERY cell, hence Dq - E is a lower bound and the exact argmin of the last row
    lies among the columns within 2 E of the screening minimum -- on small matrices, values on the worst rounding spots
    (k + 1/2 units), ties, large levels, with images carrying the full evaluation error the guard admits (1 / E)."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for case in range(400):
        N, n = int(rng.integers(1, 10)), int(rng.integers(1, 36))
        E = N + n + 2
        kind = case % 4
        if kind == 0:                                                     # half-unit values: every image rounds by 1/2
            x = (rng.integers(-2000, 2000, N) + 0.5) / QSCALE * float(rng.choice([1, 4097]))
            y = (rng.integers(-2000, 2000, n) + 0.5) / QSCALE * float(rng.choice([1, 4097]))
        elif kind == 1:                                                   # ties
            x, y = rng.integers(-3, 4, N).astype(float), rng.integers(-3, 4, n).astype(float)
        elif kind == 2:                                                   # the range's edge
            x, y = rng.uniform(-399.9, 399.9, N), rng.uniform(-399.9, 399.9, n)
        else:
            x, y = rng.normal(0, 1, N), rng.normal(0, 1, n)
        # sample images with an evaluation error of up to 1 / E units on top of the rounding (what the guard admits)
        ev = rng.uniform(-1.0 / E, 1.0 / E, n)
        xq = [qimg(v * QSCALE) for v in x]
        yq = [qimg(v * QSCALE + e) for v, e in zip(y, ev)]
        Dq = screening_matrix(xq, yq)
        D = exact_matrix(x, y)
        for i in range(N):
            for j in range(n):
                # a lower bound always (a cost that saturates at 2^32 - 1 only gets smaller than the truth) ...
                assert Dq[i][j] - E <= D[i, j] * QSCALE, (case, N, n, i, j, Dq[i][j], D[i, j] * QSCALE)
                if Dq[i][j] < QSAFE:                                      # ... and within E wherever it cannot have saturated
                    d = abs(Dq[i][j] - D[i, j] * QSCALE)
                    assert d <= E, (case, N, n, i, j, Dq[i][j], D[i, j] * QSCALE)
                    worst = max(worst, d / E)
        last_q = np.array(Dq[N - 1], dtype=np.int64)
        if last_q.min() < QSAFE:                                          # (else the kernel takes the exact pass: `b < QSAFE`)
            jstar = int(np.argmin(D[N - 1]))
            assert last_q[jstar] <= last_q.min() + 2 * E, (case, "the exact argmin is not a candidate column")
    assert worst > 0.2                                                     # E is not a loose formality either


def test_the_build_keeps_every_double_operation_a_single_rounding():
    """The bounds above count roundings one by one (and the bit-exact parity with numpy / mlpy needs the same): the
    library must be built without FMA contraction and without fast-math -- explicit __builtin_fma calls only."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = open(os.path.join(root, "squigglekit_amd", "csrc", "Makefile")).read()
    for var in ("CXXFLAGS", "HOSTFLAGS"):
        flags = re.search(r"^%s\s*=\s*((?:.*\\\n)*.*)$" % var, mk, re.M).group(1)
        assert "-ffp-contract=off" in flags and "-fno-fast-math" in flags, var
    assert "-ffast-math" not in mk and "-ffp-contract=fast" not in mk
