"""pytest configuration: `gpu` marker, shared paths and fixtures.

`-m "not gpu"` runs here (no GPU): oracle vs reference goldens, host logic, ABI
symbols.  `-m gpu` runs on the MI355X box: HIP path vs oracle / goldens, always
through the C ABI (squigglekit_amd._lib), never through a CPU fallback.
"""
import gzip
import json
import os
import sys

import numpy as np
import pytest

os.environ["SK_TUNING"] = "1"        # the library reads its tuning switches only with this set (tests flip them)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


def load_golden(name):
    path = os.path.join(GOLD, name)
    if name.endswith(".gz"):
        with gzip.open(path, "rt") as fh:
            return json.load(fh)
    with open(path) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def ora():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def example_read():
    from squigglekit_amd.blow5 import read_blow5
    return next(read_blow5(os.path.join(GOLD, "example_0.blow5")))


@pytest.fixture(scope="session")
def example_model():
    """The example scrappie model expanded like MotifSeq.read_synth_model (163 points)."""
    d = load_golden("motifseq_cli.json.gz")
    return np.array(d["model_expanded"]["values"], dtype=np.float64)


@pytest.fixture(scope="session")
def gpu():
    """Bind the HIP library to device 0; fail loudly if it cannot be done."""
    from squigglekit_amd import _lib
    _lib.init(0)
    return _lib


def oracle_motifseq_threaded(ora, sig, lens, motif, scale_mode=0, threads=None):
    """The oracle over a batch, reads split over host threads (its ctypes calls release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    R = sig.shape[0]
    T = max(1, min(threads or (os.cpu_count() or 1), 64, R))
    per = (R + T - 1) // T
    parts = [(i, min(R, i + per)) for i in range(0, R, per)]
    with ThreadPoolExecutor(T) as ex:
        res = list(ex.map(lambda ab: ora.motifseq_batch_i16(sig[ab[0]:ab[1]], lens[ab[0]:ab[1]], motif,
                                                            scale_mode=scale_mode), parts))
    return np.concatenate(res)


def strided_rows(total, want, run=4):
    """About `want` row indices spread over [0, total): runs of `run` consecutive rows at evenly spaced
    positions, always including the first and the last rows of the range."""
    nruns = max(2, want // run)
    starts = np.unique(np.linspace(0, max(0, total - run), nruns).astype(np.int64))
    idx = (starts[:, None] + np.arange(run)[None, :]).ravel()
    return np.unique(idx[idx < total])


def download_rows(L, d_base, row_bytes, rows, dtype, row_items):
    """Rows `rows` of a device array [*, row_bytes] -> numpy [len(rows), row_items]."""
    import ctypes as C
    out = np.empty((len(rows), row_items), dtype=dtype)
    base = d_base if isinstance(d_base, int) else C.cast(d_base, C.c_void_p).value
    one = np.empty(row_items, dtype=dtype)
    for k, r in enumerate(rows):
        rc = L.sk_dev_download(one.ctypes.data_as(C.c_void_p), C.c_void_p(base + int(r) * row_bytes), one.nbytes)
        assert rc == 0, L.sk_last_error()
        out[k] = one
    return out
