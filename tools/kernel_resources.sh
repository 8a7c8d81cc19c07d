#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [pattern]   - VGPR / SGPR / spill / LDS / occupancy of the kernels of one source
cd "$(dirname "$0")/.." || exit 1
mkdir -p /tmp/kres
f=${1:-brisk_kernels.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -simplifycfg-sink-common=false \
  -DBRISK_KERNEL_REV='"x"' $BRISK_HIPCC_EXTRA -c ethzasl_brisk_amd/csrc/$f -o /tmp/kres/k.o -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage 2> /tmp/kres/res.txt
python3 - "$2" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
t = open("/tmp/kres/res.txt").read()
if "error" in t: print(t[:3000])
for m in re.finditer(r"Function Name: (\S+).*?\n(.*?)LDS Size \[bytes/block\]: (\d+)", t, re.S):
    name, body, lds = m.group(1), m.group(2), m.group(3)
    if pat and pat not in name: continue
    g = lambda k: re.search(k + r": (\d+)", body).group(1)
    print("%-60s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %s" % (name[:60], g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), lds))
PY
