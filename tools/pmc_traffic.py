#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into HBM bytes per read per kernel.

    pmc_traffic.py <reads_per_call> <calls> <fetch_dir> <write_dir> [label] > profiles/traffic_<workload>.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE reports exactly half of the bytes
of a coalesced streaming read (TCC_EA0_RDREQ counted at 64 B per 128-B request), so the read side
is doubled.  (Calibrated here: k_sdtw reads each filtered int16 sample once -- 2*n B/read -- and the
doubled counter gives 7.9 KB/read for n = 3 996.)  Collected in separate --pmc passes with
--kernel-trace only.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernels_sha import kernels_sha


def load(d, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                out[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    reads, calls = int(sys.argv[1]), int(sys.argv[2])
    fetch = load(sys.argv[3], "FETCH_SIZE")
    write = load(sys.argv[4], "WRITE_SIZE")
    label = sys.argv[5] if len(sys.argv) > 5 else ""
    res = {"label": label, "reads_per_call": reads, "calls": calls, "kernels_sha": kernels_sha(),
           "note": "bytes = FETCH_SIZE*1024*2 (gfx950 half-count correction) + WRITE_SIZE*1024; "
                   "per call = sum over the launches one hot-path call makes (chunks), averaged over calls",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "rocclr" in k or "synth" in k:
            continue
        f = fetch.get(k, [0.0])
        w = write.get(k, [0.0])
        res["kernels"][short(k)] = {
            "launches_seen": len(f),
            "fetch_bytes_per_launch": sum(f) / len(f) * 1024 * 2,
            "write_bytes_per_launch": sum(w) / len(w) * 1024,
            "fetch_bytes_total": sum(f) * 1024 * 2,
            "write_bytes_total": sum(w) * 1024,
        }
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
