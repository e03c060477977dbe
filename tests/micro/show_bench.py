import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("K3", d["ms_per_step"], d["ms_per_step_steady"], d["value"], d["kernels_ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for k,v in d.get("configs",{}).items(): print(k, v.get("ms_per_step"), v.get("ms_per_step_steady"), v.get("value"), v.get("kernels_ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("note"))
