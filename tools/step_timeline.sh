# One MotifSeq step dispatch by dispatch (rocprofv3 --kernel-trace): which kernels run beside which, where the tail goes.
# usage (GPU box, repo root): bash tools/step_timeline.sh [reads per call] [tag]
export SK_TUNING=1        # the library reads its tuning switches only with this set
READS=${1:-1000000}; TAG=${2:-timeline}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > $OUT/kt.log 2>&1
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - <<PY > $OUT/timeline_$READS.txt
import csv
rows=list(csv.DictReader(open("$F")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last step: from the last k_sdtw_q launch on
qi=[i for i,r in enumerate(rows) if "k_sdtw_q" in r["Kernel_Name"]]
rows=rows[qi[-1]-4:]
t0=int(rows[4]["Start_Timestamp"])
print("one MotifSeq step of $READS reads, dispatch by dispatch (ms from the start of pass Q)")
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    nm=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:44]
    print("%-44s start %9.3f  end %9.3f  dur %8.3f ms  queue %s  grid %s" % (nm,(s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,r.get("Queue_Id",""),r.get("Grid_Size_X", r.get("Grid_Size",""))))
PY
cat $OUT/timeline_$READS.txt
rm -rf $OUT/kt
