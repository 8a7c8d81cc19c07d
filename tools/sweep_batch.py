"""GPU box: stage times per frame as a function of the chunk size (does a chunk that fits the Infinity Cache describe faster?)."""
import json
import subprocess
import sys

flags = sys.argv[1] if len(sys.argv) > 1 else "0"
for batch in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,24,32,48,64,128,256").split(",")]:
    inner = max(1, 2048 // batch)
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-host-fed", "--no-other-configs", "--steps", "3", "--warmup", "1", "--batch", str(batch),
                          "--inner", str(inner), "--debug-flags", flags], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        st = d["config"]["stage_ms_per_chunk"]
        print("batch %3d : %6.0f frames/s | us per frame: %s" % (batch, d["value"], "  ".join("%s %.2f" % (k.replace("k_", ""), v * 1e3 / batch) for k, v in st.items())), flush=True)
    except Exception as e:  # noqa
        print("batch %d failed %s %s" % (batch, e, out.stderr[-300:]), flush=True)
