#!/usr/bin/env python3
"""prints the essentials of a bench.py JSON line: value, roofline, kernel groups, other configurations"""
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "ms/chunk", d["config"]["ms_per_chunk"])
print({k: (v["frac"], v["hbm_frac"]) for k, v in d["config"]["kernel_groups"].items()})
oc = d["config"].get("other_configs")
if isinstance(oc, dict):
    for k, v in oc.items():
        print(k, v.get("value"), v.get("unit"), v.get("frac"), v.get("failed", ""))
else:
    print("other_configs:", oc)
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
