"""GPU box: k_describe launch knobs (debug bits 8-11 = keypoints per ticket, 12-15 = workgroups per CU) through bench.py."""
import json
import os
import subprocess
import sys

runs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
bpcs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
stage0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # workgroups per CU of the orientation stage (0: same)
orders = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
extra = sys.argv[5:]
for order in orders:
  for run in runs:
    for bpc in bpcs:
        flags = (run << 8) | (bpc << 12) | (stage0 << 20) | ((order & 15) << 4) | ((order >> 4) << 29) | int(os.environ.get("SWEEP_EXTRA_FLAGS", "0"), 0)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-host-fed", "--steps", "3", "--warmup", "1",
                              "--inner", "8", "--debug-flags", str(flags)] + extra, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            st = d["config"]["stage_ms_per_chunk"]
            print("order %d run %d wg/CU %d : k_describe %.3f ms  chunk %.3f ms  %.0f frames/s" % (order, run, bpc, st["k_describe"], d["config"]["ms_per_chunk"], d["value"]), flush=True)
        except Exception as e:  # noqa
            print("order %d run %d wg/CU %d : failed %s %s" % (order, run, bpc, e, out.stderr[-400:]), flush=True)
