"""GPU box: frames/s as a function of the chunk size and the number of internal stream slices per chunk."""
import json
import subprocess
import sys

for batch in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128").split(",")]:
    row = []
    for streams in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]:
        inner = max(1, 2048 // batch)
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-host-fed", "--no-other-configs", "--steps", "3", "--warmup", "1",
                              "--batch", str(batch), "--inner", str(inner), "--streams", str(streams)], capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            row.append("%d slices %6.0f (%s)" % (streams, d["value"], d["config"].get("stream_slices")))
        except Exception as e:  # noqa
            row.append("%d slices failed" % streams)
    print("batch %3d : %s" % (batch, "   ".join(row)), flush=True)
