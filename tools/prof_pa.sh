set -u
R=$(pwd); OUT=$R/gpurun_out/r6/prof_pa; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp SK_TUNING=1
for route in pa i16; do python $R/tools/bench_pa_long.py 50000 20000 5 $route; python $R/tools/bench_pa_long.py 25000 36978 5 $route; done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python $R/tools/bench_pa_long.py 50000 20000 2 pa > $OUT/pmc_$C.log 2>&1
done
python $R/tools/pmc_traffic.py 50000 3 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE "bench_pa_long 50000 x 20000 pa (3 calls)" > $OUT/traffic_pa_20k.json 2>&1
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=1
for SQ in "$SQ1" "$SQ2"; do
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/pmc_sq$i -- python $R/tools/bench_pa_long.py 50000 20000 2 pa > $OUT/pmc_sq$i.log 2>&1
  python $R/tools/pmc_sq.py $OUT/pmc_sq$i "bench_pa_long 50000 x 20000 pa, pass $i" > $OUT/sq${i}_pa_20k.json 2>&1
  i=$((i+1))
done
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq1 $OUT/pmc_sq2
python - <<EOF
import json
for f in ("traffic_pa_20k","sq1_pa_20k","sq2_pa_20k"):
    try:
        d=json.load(open("$OUT/%s.json"%f))
        for k,v in d.get("kernels",{}).items():
            if "seg_stats" in k or "walkL" in k: print(f, k[:60], {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    except Exception as e: print(f, "ERR", e, open("$OUT/%s.json"%f).read()[:300])
EOF
