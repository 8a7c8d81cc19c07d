#!/bin/bash
# usage (GPU box): bash tools/pmc_lds_patch.sh <tag> [filler] : tools/microbench_lds_patch.hip - rates, then the vector-L1 /
# L2 / LDS / VALU counters of every (class, variant) row per SAMPLE -> gpurun_out/<tag>/lds_patch.json (+ .txt)
tag=$1; filler=${2:-300}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mblds $GRAFT_REPO_ROOT/tools/microbench_lds_patch.hip \
      $GRAFT_REPO_ROOT/ethzasl_brisk_amd/csrc/brisk_pattern.cpp || exit 1
python3 $GRAFT_REPO_ROOT/tools/gen_lds_patch_input.py /tmp/lds_in.bin > $out/input.txt || exit 1
cd /tmp && export TMPDIR=/tmp
/tmp/mblds /tmp/lds_in.bin $filler > $out/rates.json 2> $out/rates.err || { cat $out/rates.err; exit 1; }
i=0
for set in "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $out/p$i -o pass --output-format csv -- /tmp/mblds /tmp/lds_in.bin $filler > $out/p$i.json 2> $out/p$i.log
done
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
doc = json.load(open(out + "/rates.json"))
rows = doc["rows"]
merged = None
for p in sorted(glob.glob(out + "/p[0-9]")):
    ids = collections.OrderedDict()
    for f in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_gather" not in r["Kernel_Name"] and "k_lds" not in r["Kernel_Name"]:
                continue
            ids.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    lst = [ids[k] for k in sorted(ids)]
    if merged is None:
        merged = [dict(x) for x in lst]
    else:
        for a, b in zip(merged, lst):
            a.update(b)
lines = []
for i, r in enumerate(rows):
    c = merged[3 * i + 2] if merged and len(merged) >= 3 * i + 3 else {}
    samples = r["keypoints"] * 132.0
    r["per_sample"] = {k: round(v / samples, 3) for k, v in c.items()}
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    r["l2_hit_rate"] = round(hit / max(hit + miss, 1), 3)
    ps = r["per_sample"]
    lines.append("%-14s %-18s waves/CU %2d  %6.2f samples/ns  %.3f us/kp/CU | per sample: TCP accesses %5.2f  L2 req %5.2f  L2 hit %.3f  VALU %5.2f  LDS inst %5.2f  bank-conflict cyc %5.2f" % (
        r["class"], r["variant"], r["waves_per_cu"], r["samples_per_ns_chip"], r["us_per_keypoint_cu"], ps.get("TCP_TOTAL_ACCESSES_sum", 0),
        ps.get("TCP_TCC_READ_REQ_sum", 0), r["l2_hit_rate"], ps.get("SQ_INSTS_VALU", 0), ps.get("SQ_INSTS_LDS", 0), ps.get("SQ_LDS_BANK_CONFLICT", 0)))
json.dump(doc, open(out + "/lds_patch.json", "w"), indent=1)
open(out + "/lds_patch.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $out/p[0-9]
