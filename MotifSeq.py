#!/usr/bin/env python3
"""MotifSeq.py -- MI355X drop-in for SquiggleKit's MotifSeq.py (same flags, same output).
Thin launcher; the tool lives in squigglekit_amd/motifseq_cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from squigglekit_amd import _warm  # noqa: E402
_warm.start()                      # the GPU context comes up while numpy and the tool are being imported
from squigglekit_amd.motifseq_cli import main  # noqa: E402

if __name__ == "__main__":
    main()
    _warm.fast_exit(0)             # (sys.exit inside main() leaves the ordinary way)
