#!/usr/bin/env python3
"""usage: trace_latency.py <hip_api_trace.csv> <kernel_trace.csv> [<memory_copy_trace.csv>]: timeline of the LAST detect + compute pair."""
import csv
import sys

api = list(csv.DictReader(open(sys.argv[1])))
ker = list(csv.DictReader(open(sys.argv[2])))
ev = []
for r in api:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for r in ker:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", r["Kernel_Name"].split("(")[0][:40]))
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "")))
ev.sort()
# the last k_detect launch marks the last detect call; start two uploads before it
kd = [e for e in ev if e[2] == "kernel" and e[3].startswith("k_detect")]
t_last = kd[-1][0]
t_prev = kd[-2][0]
period = t_last - t_prev
lo = t_last - period // 3
sel = [e for e in ev if lo <= e[0] <= lo + period]
t0 = sel[0][0]
for s, e, kind, name in sel:
    print("%9.1f %8.1f  %-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, kind, name))
print("period %.1f us" % (period / 1e3))
