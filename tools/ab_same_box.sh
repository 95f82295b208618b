# usage (on a GPU box, repo root): bash tools/ab_same_box.sh <tag> [bench args]  -- alternates a baseline build of the library (copy it to squigglekit_amd/libsk_alt_base.so first; picked up through SK_LIB_PATH) and the current one, twice each: boxes differ by a few per cent, so only runs on one box compare
export SK_TUNING=1        # the library reads its tuning switches only with this set
TAG=$1; shift
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export SK_LIB_PATH=$R/squigglekit_amd/libsk_alt_base.so; else unset SK_LIB_PATH; fi
    python bench.py --steps 5 --warmup 1 --no-extras --cpu-seconds 0 "$@" --full-json $OUT/ab_${v}_$rep.json > $OUT/ab_${v}_$rep.lines 2> $OUT/ab_${v}_$rep.err
  done
done
unset SK_LIB_PATH
python - <<PY
import json
for v in ("base","new"):
    for rep in (1,2):
        try:
            d=json.load(open("$OUT/ab_%s_%d.json"%(v,rep))); p=d["roofline"]["valu"].get("passes_ms_per_call",{})
            print(v, rep, "%.2f ms"%d["ms_per_step"], {k:round(x,2) for k,x in p.items()}, d["parity"].get("dist_bit_identical"), d["parity"].get("start_end_exact"))
        except Exception as e: print(v, rep, "ERR", e)
PY
