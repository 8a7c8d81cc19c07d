#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
HBM bytes per launch for every engine kernel, with the gfx950 corrections of that guide:
  * counters are in KiB;
  * FETCH_SIZE under-reports wide coalesced streaming reads (exactly 1/2 for 16 B/lane).  Our kernels read with
    4 B/lane dword loads, an uncalibrated width, so the read side was calibrated in this repo on kernels of the same
    access width with an exactly known byte count: k_copy_layer0 (factor 1.9992) and k_integral_bandsums (1.9995),
    both since folded into k_pyramid_even.  The factor 2.0 is therefore applied to every kernel's FETCH_SIZE.
Also sums the kernels into the groups of SURVEY 8(d) that bench.py reports (`groups`, bytes per launch) and names the
kernel revision (brisk_hip_kernel_revision) the passes were measured on: bench.py uses the file only for that revision.
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <frames_per_launch> <w> <h> [out.json [kernel_revision]]
"""
import csv


def kernel_name(n):
    """k_foo / void k_foo<1>(...) -> k_foo"""
    n = n.strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0].split("<")[0]
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[kernel_name(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    frames, w, h = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    corr = 2.0
    out = {"frames_per_launch": frames, "fetch_correction": corr, "image": [w, h], "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, wr = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"fetch_raw": f, "fetch_corrected": f * corr, "write": wr, "hbm_bytes_per_launch": f * corr + wr}
    group_of = {"k_pyramid_fused": "pyramid", "k_pyramid_even": "pyramid", "k_pyramid_odd": "pyramid", "k_pyramid_level": "pyramid", "k_detect": "detect",
                "k_score_blocks": "nms", "k_classify_refine": "nms", "k_classify_refine_direct": "nms", "k_tie_resolve": "nms",
                "k_finalize": "nms", "k_finalize_large": "nms", "k_smap_clear": "nms", "k_order_candidates": "nms",
                "k_ordered_keypoints": "nms", "k_nms": "nms", "k_integral_final": "integral", "k_desc_prepare": "describe",
                "k_describe": "describe", "k_dp_count": "describe", "k_dp_scan": "describe", "k_dp_scatter": "describe",
                "k_uf_rank": "nms", "k_uf_decide": "nms", "k_uniformity_seq": "nms", "k_bucketing": "nms"}
    groups = {}
    for k, v in out["kernels"].items():
        g = group_of.get(k)
        if g:
            groups[g] = groups.get(g, 0.0) + v["hbm_bytes_per_launch"]
    out["groups"] = groups
    out["kernel_revision"] = sys.argv[7] if len(sys.argv) > 7 else None
    out["hbm_bytes_per_launch"] = out["kernels"].get("k_detect", {}).get("hbm_bytes_per_launch")  # (descriptor-only runs have no detector)
    out["hbm_bytes_all_kernels_per_launch"] = sum(v["hbm_bytes_per_launch"] for v in out["kernels"].values())
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 6:
        open(sys.argv[6], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
