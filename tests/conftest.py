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
