hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mblds tools/microbench_lds_patch.hip ethzasl_brisk_amd/csrc/brisk_pattern.cpp || exit 1
python3 tools/gen_lds_patch_input.py /tmp/lds_in.bin > /dev/null || exit 1
mkdir -p gpurun_out/il2
/tmp/mblds /tmp/lds_in.bin 300 gather > gpurun_out/il2/rates.json
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/il2/rates.json"))
for r in d["rows"]: print("%-16s %-12s kp %7d  %7.3f ms  %6.2f samples/ns  checksum %d" % (r["class"], r["variant"], r["keypoints"], r["ms"], r["samples_per_ns_chip"], r["checksum"]))
PY
