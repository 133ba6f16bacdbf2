import json,sys
for f in sys.argv[1:]:
    l=json.loads([x for x in open(f).read().strip().splitlines() if x.startswith("{")][-1])
    print(f, l["config"]["workload"][:12], l["ms_per_step"], l["roofline"]["frac"])
    for k,v in l.get("secondary",{}).items():
        if isinstance(v,dict) and "roofline" in v: print("   ", k, v.get("ms_per_step"), v["roofline"]["frac"])
