#!/usr/bin/env python3
"""Averages SQ counters per kernel from one or more rocprofv3 counter_collection.csv files.
Usage: pmc_sq.py <csv> [<csv> ...] [--kernel k_detect]"""
import csv


def kernel_name(n):
    """k_foo / void k_foo<1>(...) -> k_foo"""
    n = n.strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0].split("<")[0]
import sys
from collections import defaultdict

want = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--kernel":
        want = args.pop(0)
    else:
        files.append(a)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = kernel_name(r["Kernel_Name"])
        if not k.startswith("k_") or (want and k != want):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, "launches", max(len(v) for v in acc[k].values()))
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.1f" % (c, sum(v) / len(v)))
