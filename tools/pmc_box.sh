#!/bin/bash
# usage (GPU box): tools/pmc_box.sh <tag> : L2 hit / miss and L1->L2 requests per dispatch of the describe-shaped microbenchmark
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o /tmp/mbg $GRAFT_REPO_ROOT/tools/microbench_gather.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $out/p1 -o pass --output-format csv -- /tmp/mbg box > $out/box_pmc.json 2> $out/p1.log
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $out/p2 -o pass --output-format csv -- /tmp/mbg box > /dev/null 2> $out/p2.log
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = [json.loads(l.strip().rstrip(',')) for l in open(out + "/box_pmc.json") if l.startswith('{"radius"')]
disp = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_box" not in r["Kernel_Name"]: continue
        disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
# two passes: dispatch ids restart per process; group by order inside each pass
per_pass = {}
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    ids = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_box" not in r["Kernel_Name"]: continue
        ids.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    per_pass[f] = [ids[k] for k in sorted(ids)]
merged = None
for f, lst in per_pass.items():
    if merged is None: merged = [dict(x) for x in lst]
    else:
        for a, b in zip(merged, lst): a.update(b)
# 3 launches per configuration row
with open(out + "/box_counters.txt", "w") as fo:
    for i, r in enumerate(rows):
        c = merged[3 * i + 2]
        samples = None
        hit = c.get("TCC_HIT_sum", 0); miss = c.get("TCC_MISS_sum", 0); req = c.get("TCP_TCC_READ_REQ_sum", 0)
        line = "r%3d band %3d bytes %d valu %4d W %d : %6.1f samples/ns | L2 hit %.3f | tag accesses/L2 req %.2f | req latency %.0f clk" % (
            r["radius"], r["band_rows"], r["byte_gathers"], r["valu_steps"], r["waves_per_simd"], r["samples_per_ns_chip"],
            hit / max(hit + miss, 1), c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / max(req, 1), c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(req, 1))
        print(line); fo.write(line + "\n")
PY
rm -rf $out/p1 $out/p2
