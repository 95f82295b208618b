#!/bin/bash
# After tools/profile_round.sh <tag> has run on a GPU box (gpurun merges gpurun_out/prof_<tag>/ back): copy what is kept
# under profiles/ -- the bench lines, the rocprofv3 summaries, the counter passes, CLI and parity logs as <tag>_<name>,
# and the two traffic files bench.py reads (profiles/traffic_*.json, stamped with the kernel sources they were measured on).
#   tools/copy_profiles.sh r06 [source-dir-tag]      (source defaults to the tag: gpurun_out/prof_<source>)
set -eu
TAG=${1:-r06}
O=gpurun_out/prof_${2:-$TAG}
for f in bench_2ranks_on_one_gpu.json bench_c2_10k_segmenter.json bench_c3_10k_x_163pt.json bench_c5_100k_x_20000_x_500pt.json \
         bench_motifseq.json bench_other_paths.json bench_segmenter.json cli_throughput.txt cpu_suite.txt hbm_stream.txt \
         motifseq_kernel_stats.txt other_paths_kernel_stats.txt parity_at_scale.txt segmenter_kernel_stats.txt \
         sq1_motifseq.json sq1_segmenter.json sq2_motifseq.json sq2_segmenter.json traffic_motifseq.json \
         traffic_other_paths.json traffic_segmenter.json valu_rate.txt sq1_other_paths.json sq2_other_paths.json; do
    if [ -s "$O/$f" ]; then cp "$O/$f" "profiles/${TAG}_$f"; else echo "missing: $O/$f" >&2; fi
done
cp "$O/traffic_motifseq.json" profiles/traffic_motifseq.json
cp "$O/traffic_segmenter.json" profiles/traffic_segmenter.json
# the counter passes bench_extras.py reads for the other_paths rooflines (tools/prof_other_sq.sh writes the SQ ones)
for f in traffic_other_paths.json sq1_other_paths.json; do [ -s "$O/$f" ] && cp "$O/$f" "profiles/$f"; done
echo "kernel sources now: $(python tools/kernels_sha.py)   stamped: $(grep -o '"kernels_sha": "[0-9a-f]*"' profiles/traffic_motifseq.json)"
