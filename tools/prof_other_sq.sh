#!/bin/bash
# SQ counter passes (VALU instruction counts / busy, waits) and HBM traffic of the paths the headline does not take:
#   tools/prof_other_sq.sh <tag>     -> gpurun_out/prof_<tag>/{sq1,sq2,traffic}_other_paths.json, other_paths_kernel_stats.txt
# counters always with --kernel-trace only (never with other trace domains)
set -u
TAG=${1:-r06}
R=$(pwd); OUT=$R/gpurun_out/prof_$TAG; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp SK_TUNING=1
CMD="python $R/bench.py --only-other-paths --steps 1 --warmup 0 --cpu-seconds 0"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=1
for SQ in "$SQ1" "$SQ2"; do
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$OUT/pmc_sq${i}_other" -- $CMD > "$OUT/pmc_sq${i}_other.log" 2>&1
  python "$R/tools/pmc_sq.py" "$OUT/pmc_sq${i}_other" "bench.py --only-other-paths --steps 1 --warmup 0 ($TAG, pass $i)" > "$OUT/sq${i}_other_paths.json"
  rm -rf "$OUT/pmc_sq${i}_other"
  i=$((i+1))
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_other" -- $CMD > "$OUT/pmc_${C}_other.log" 2>&1
done
python "$R/tools/pmc_traffic.py" 1 1 "$OUT/pmc_FETCH_SIZE_other" "$OUT/pmc_WRITE_SIZE_other" "bench.py --only-other-paths --steps 1 --warmup 0 ($TAG; per-launch bytes)" > "$OUT/traffic_other_paths.json"
rm -rf "$OUT/pmc_FETCH_SIZE_other" "$OUT/pmc_WRITE_SIZE_other"
rocprofv3 --kernel-trace --stats -d "$OUT/kt_other" -- $CMD > "$OUT/kt_other.log" 2>&1
DB=$(find "$OUT/kt_other" -name '*_results.db' | head -1)
python "$R/tools/rocprof_summary.py" "$DB" "bench.py --only-other-paths --steps 1 --warmup 0 ($TAG)" > "$OUT/other_paths_kernel_stats.txt"
rm -rf "$OUT/kt_other"
python - <<EOF
import json
for f in ("sq1_other_paths","sq2_other_paths","traffic_other_paths"):
    d=json.load(open("$OUT/%s.json"%f))
    for k,v in d.get("kernels",{}).items():
        print(f[:3], k[:70], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("valu_per_wave","valu_busy_at_4_cycles","SQ_WAVES","kernel_cycles","fetch_bytes_per_launch","write_bytes_per_launch","launches_seen","SQ_ACTIVE_INST_VALU","SQ_WAIT_ANY","SQ_WAVE_CYCLES","dispatches")})
EOF
