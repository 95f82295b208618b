# One MotifSeq step dispatch by dispatch (rocprofv3 --kernel-trace): which kernels run beside which, where the tail goes.
# usage (GPU box, repo root): bash tools/step_timeline.sh
export SK_TUNING=1        # the library reads its tuning switches only with this set
R=$(pwd); OUT=$R/gpurun_out/r3v; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > $OUT/kt.log 2>&1
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
rows=[r for r in rows if "sdtw" in r["Kernel_Name"] or "prep" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows[-14:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:28]
    print("%-28s start %9.3f ms  dur %8.3f ms  stream/queue %s" % (nm,(s-t0)/1e6,(e-s)/1e6,r.get("Queue_Id","")))
PY
rm -rf $OUT/kt
