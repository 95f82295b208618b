import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    if d.get("extra")=="other_paths":
        for k,v in d["other_paths"].items():
            if isinstance(v,dict):
                rf=v.get("roofline",{})
                print("%-28s %8.3f ms frac %s kern %s redone %s speedup %s parity %s" % (k, v.get("ms_per_step",0), (round(rf["frac"],3) if rf.get("frac") else None), {a:round(b,3) for a,b in (v.get("kernel_ms") or {}).items()}, v.get("reads_redone_from_float64"), v.get("speedup_vs_float64_route"), [x for x in (v.get("parity") or {}).values()]))
            else: print(k, v)
