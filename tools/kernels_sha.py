#!/usr/bin/env python3
"""sha-256 (first 16 hex digits) over the HIP kernel sources, file names included -- what a measurement under profiles/
was taken on.  bench.py compares the stamp of profiles/traffic_*.json with the sources it runs on and marks the
figure `stale` when they differ (there is no .git on the GPU box to ask)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_sha(root=ROOT):
    d = os.path.join(root, "squigglekit_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernels_sha())
