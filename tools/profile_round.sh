#!/bin/bash
# Regenerate the measurement artefacts kept under profiles/ on a GPU box:
#   tools/profile_round.sh <tag>      (run from the repo root; writes gpurun_out/prof_<tag>/)
# bench lines (default flags), rocprofv3 --kernel-trace --stats summaries of the same command, and
# HBM traffic from separate --pmc passes (FETCH_SIZE / WRITE_SIZE, kernel trace only -- no other
# trace domains together with counters).
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for WL in motifseq segmenter; do
    python "$R/bench.py" --workload $WL > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err"
    rocprofv3 --kernel-trace --stats -d "$OUT/kt_$WL" -- python "$R/bench.py" --workload $WL --steps 3 --warmup 1 \
        --cpu-seconds 0 > "$OUT/kt_$WL.log" 2>&1
    DB=$(find "$OUT/kt_$WL" -name '*_results.db' | head -1)
    python "$R/tools/rocprof_summary.py" "$DB" "bench.py --workload $WL --steps 3 --warmup 1 ($TAG)" \
        > "$OUT/${WL}_kernel_stats.txt"
    for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_$WL" -- \
            python "$R/bench.py" --workload $WL --steps 2 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_${C}_$WL.log" 2>&1
    done
    python "$R/tools/pmc_traffic.py" 1000000 2 "$OUT/pmc_FETCH_SIZE_$WL" "$OUT/pmc_WRITE_SIZE_$WL" \
        "bench.py --workload $WL --steps 2 --warmup 0 ($TAG, 1M reads/call)" > "$OUT/traffic_$WL.json"
    rm -rf "$OUT/kt_$WL" "$OUT/pmc_FETCH_SIZE_$WL" "$OUT/pmc_WRITE_SIZE_$WL"
done
ls -la "$OUT"
