"""Reads a rocprofv3 kernel trace csv: which kernels overlapped in time, and for how long (ms)."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "")
    if n.startswith("k_describe<"):
        return "k_describe_s" + n.split("<")[1].split(",")[1].strip()
    return n.split("(")[0].split("<")[0]


rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = []
for r in rows:
    n = short(r["Kernel_Name"])
    if n.startswith("k_"):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0]
busy = defaultdict(float)
pair = defaultdict(float)
for i, a in enumerate(ev):
    busy[a[2]] += (a[1] - a[0]) / 1e6
    for b in ev[i + 1:]:
        if b[0] >= a[1]:
            break
        pair[tuple(sorted((a[2], b[2])))] += (min(a[1], b[1]) - b[0]) / 1e6
span = (max(e[1] for e in ev) - t0) / 1e6
print("span %.2f ms, sum of kernel durations %.2f ms, queues %s" % (span, sum(busy.values()), sorted(set(e[3] for e in ev))))
for k, v in sorted(busy.items(), key=lambda x: -x[1]):
    print("  %-28s %8.3f ms" % (k, v))
print("overlaps (ms of simultaneous execution):")
for k, v in sorted(pair.items(), key=lambda x: -x[1])[:25]:
    print("  %-28s %-28s %8.3f" % (k[0], k[1], v))
if len(sys.argv) > 2:
    for e in ev[:int(sys.argv[2])]:
        print("%10.3f %10.3f  %-26s q%s s%s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, e[2], e[3], e[4]))
