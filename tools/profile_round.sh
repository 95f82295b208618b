#!/bin/bash
# Regenerate the measurement artefacts kept under profiles/ on a GPU box:
#   tools/profile_round.sh <tag>      (run from the repo root; writes gpurun_out/prof_<tag>/)
# bench lines (default flags), rocprofv3 --kernel-trace --stats summaries of the same command, HBM traffic from
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE) and two SQ passes (VALU instruction counts / issue and wait
# cycles) -- counters always with --kernel-trace only, never with other trace domains -- the VALU issue-cost
# microbenchmark the DTW roof rests on, the command-line tools end to end, and the multi-rank dry run.
export SK_TUNING=1        # the library reads its tuning switches only with this set
set -u
TAG=${1:-r05}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
"$R/tools/ubench/valu_rate" > "$OUT/valu_rate.txt" 2>&1
"$R/tools/ubench/hbm_stream" 8 > "$OUT/hbm_stream.txt" 2>&1
python "$R/bench.py" --full-json "$OUT/bench_motifseq.json" > "$OUT/bench_motifseq.lines" 2> "$OUT/bench_motifseq.err"
python "$R/bench.py" --workload segmenter --no-extras --steps 20 --warmup 3 --full-json "$OUT/bench_segmenter.json" > "$OUT/bench_segmenter.lines" 2> "$OUT/bench_segmenter.err"   # (a pass is 2.4 ms: the first two after the buffers are allocated run 5-10 % slower)
python "$R/bench.py" --reads 10000 --motif 163 --no-extras --steps 20 --warmup 3 --full-json "$OUT/bench_c3_10k_x_163pt.json" > "$OUT/bench_c3_10k_x_163pt.lines" 2>/dev/null
python "$R/bench.py" --reads 100000 --samples 20000 --motif 500 --no-extras --cpu-seconds 6 --full-json "$OUT/bench_c5_100k_x_20000_x_500pt.json" > "$OUT/bench_c5_100k_x_20000_x_500pt.lines" 2>/dev/null
python "$R/bench.py" --workload segmenter --reads 10000 --no-extras --steps 20 --warmup 3 --full-json "$OUT/bench_c2_10k_segmenter.json" > "$OUT/bench_c2_10k_segmenter.lines" 2>/dev/null
python "$R/bench.py" --gpus 2 --ranks-on-device 0 --reads 200000 --steps 3 --warmup 1 --cpu-seconds 2 --full-json "$OUT/bench_2ranks_on_one_gpu.json" > "$OUT/bench_2ranks_on_one_gpu.lines" 2>/dev/null
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
for WL in motifseq segmenter; do
    export SK_SEG_CHUNKS=1      # kernels timed one after the other (the default overlaps the walk with the statistics)
    rocprofv3 --kernel-trace --stats -d "$OUT/kt_$WL" -- python "$R/bench.py" --workload $WL --steps 3 --warmup 1 \
        --cpu-seconds 0 --no-extras > "$OUT/kt_$WL.log" 2>&1
    DB=$(find "$OUT/kt_$WL" -name '*_results.db' | head -1)
    python "$R/tools/rocprof_summary.py" "$DB" "bench.py --workload $WL --steps 3 --warmup 1 ($TAG)" \
        > "$OUT/${WL}_kernel_stats.txt"
    for C in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_$WL" -- \
            python "$R/bench.py" --workload $WL --steps 2 --warmup 0 --cpu-seconds 0 --no-extras > "$OUT/pmc_${C}_$WL.log" 2>&1
    done
    python "$R/tools/pmc_traffic.py" 1000000 2 "$OUT/pmc_FETCH_SIZE_$WL" "$OUT/pmc_WRITE_SIZE_$WL" \
        "bench.py --workload $WL --steps 2 --warmup 0 ($TAG, 1M reads/call)" > "$OUT/traffic_$WL.json"
    i=1
    for SQ in "$SQ1" "$SQ2"; do
        rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$OUT/pmc_sq${i}_$WL" -- \
            python "$R/bench.py" --workload $WL --steps 2 --warmup 0 --cpu-seconds 0 --no-extras > "$OUT/pmc_sq${i}_$WL.log" 2>&1
        python "$R/tools/pmc_sq.py" "$OUT/pmc_sq${i}_$WL" "bench.py --workload $WL --steps 2 --warmup 0 ($TAG, pass $i)" > "$OUT/sq${i}_$WL.json"
        rm -rf "$OUT/pmc_sq${i}_$WL"
        i=$((i+1))
    done
    rm -rf "$OUT/kt_$WL" "$OUT/pmc_FETCH_SIZE_$WL" "$OUT/pmc_WRITE_SIZE_$WL"
    unset SK_SEG_CHUNKS
done
# the paths the headline does not take (bench.py other_paths: float64 pA segmenter, 4 000- and 20 000-sample reads, float64
# medmad, int16 zscale, four motifs): the bench block, a kernel-trace summary and the FETCH / WRITE passes of the same command
python "$R/bench.py" --only-other-paths --steps 3 --cpu-seconds 0 --full-json "$OUT/bench_other_paths.json" > "$OUT/bench_other_paths.lines" 2> "$OUT/bench_other_paths.err"
rocprofv3 --kernel-trace --stats -d "$OUT/kt_other" -- python "$R/bench.py" --only-other-paths --steps 1 --warmup 0 \
    --cpu-seconds 0 > "$OUT/kt_other.log" 2>&1
DB=$(find "$OUT/kt_other" -name '*_results.db' | head -1)
python "$R/tools/rocprof_summary.py" "$DB" "bench.py --only-other-paths --steps 1 --warmup 0 ($TAG)" > "$OUT/other_paths_kernel_stats.txt"
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_other" -- \
        python "$R/bench.py" --only-other-paths --steps 1 --warmup 0 --cpu-seconds 0 > "$OUT/pmc_${C}_other.log" 2>&1
done
python "$R/tools/pmc_traffic.py" 500000 1 "$OUT/pmc_FETCH_SIZE_other" "$OUT/pmc_WRITE_SIZE_other" \
    "bench.py --only-other-paths --steps 1 --warmup 0 ($TAG; per-launch bytes; the float64 kernels run on 500 000 reads x 3 999 samples, 50 000 x 19 999 and 25 000 x 36 977, the dRNA ones on 250 000 reads, 4 launches each)" > "$OUT/traffic_other_paths.json"
rm -rf "$OUT/kt_other" "$OUT/pmc_FETCH_SIZE_other" "$OUT/pmc_WRITE_SIZE_other"
python "$R/tools/cli_throughput.py" 200000 1000000 > "$OUT/cli_throughput.txt" 2>&1
(cd "$R" && python tools/parity_at_scale.py 400000 128 && python tools/parity_at_scale.py segmenter 1000000 128) > "$OUT/parity_at_scale.txt" 2>&1
(cd "$R" && tools/check.sh) > "$OUT/cpu_suite.txt" 2>&1
ls -la "$OUT"
