#!/usr/bin/env python3
"""Per-kernel means of a rocprofv3 --pmc pass (csv) -> JSON.

    pmc_sq.py <pmc_dir> [label] > profiles/rNN_sq_<workload>.json

Used for the SQ pass of tools/profile_round.sh (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
SQ_WAVES, SQ_INSTS_LDS, SQ_INSTS_SALU, SQ_WAIT_INST_ANY + GRBM_GUI_ACTIVE), collected with --kernel-trace only.
Derived per kernel:  valu_per_wave = SQ_INSTS_VALU / SQ_WAVES;  valu_busy_at_4_cycles = 4 * SQ_INSTS_VALU /
(kernel cycles * 1024 SIMDs) with kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter comes back summed over the 8
XCDs: 36.0 M for a 2.08 ms kernel) -- the share of all SIMD issue cycles the launch's vector instructions would
fill at 4 cycles each (tools/ubench/valu_rate: 2.5-2.7 cycles for add / and / xor / bitop3 / mov, 4.2-4.6 for the
rest), so a kernel made of the cheap ones can read above 1.  SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this
stack (one count per instruction), so it says nothing about issue cycles."""
import collections
import csv
import glob
import json
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {"label": label, "note": "means per dispatch; SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles",
           "kernels": {}}
    for k, cs in sorted(acc.items()):
        if "rocclr" in k or "synth" in k:
            continue
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["dispatches"] = max(len(v) for v in cs.values())
        if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
            m["valu_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if m.get("SQ_INSTS_VALU") and m.get("GRBM_GUI_ACTIVE"):
            m["kernel_cycles"] = m["GRBM_GUI_ACTIVE"] / 8.0
            m["valu_busy_at_4_cycles"] = 4.0 * m["SQ_INSTS_VALU"] / (m["kernel_cycles"] * 1024.0)
        out["kernels"][k] = m
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
