#!/bin/bash
# usage: bash tools/detect_phase_isa.sh : instruction counts of k_detect between the phase stamps of the DT_TIMING build
# (VALU / SALU / LDS / VMEM per phase of one wave's straight-line code; phase B is a loop: counted once)
cd /tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -simplifycfg-sink-common=false -DDT_TIMING -save-temps -c \
  -o /tmp/bk_dt.o ${GRAFT_REPO_ROOT:-/root/repo}/ethzasl_brisk_amd/csrc/brisk_kernels.hip 2> /dev/null
sed -n '/^_Z8k_detect/,/s_endpgm/p' /tmp/brisk_kernels-hip-amdgcn-amd-amdhsa-gfx950.s | python3 -c "
import sys, re, collections
phase = -1
cnt = collections.defaultdict(lambda: collections.Counter())
for line in sys.stdin:
    t = line.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op = t.split()[0]
    if op == 's_memtime': phase += 1; continue
    k = 'VALU' if op.startswith('v_') else 'LDS' if op.startswith('ds_') else 'VMEM' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'SALU' if op.startswith('s_') else 'other'
    if op in ('v_perm_b32',): cnt[phase]['v_perm'] += 1
    if op.startswith('v_pk_'): cnt[phase]['v_pk'] += 1
    cnt[phase][k] += 1
names = ['(prologue)', 'decode + staging', 'barrier', 'window reads', 'pre-gate', 'compaction', 'barrier', 'phase B + epilogue']
for p in sorted(cnt):
    c = cnt[p]
    print('%-20s VALU %4d (v_perm %3d, v_pk %3d)  SALU %4d  LDS %3d  VMEM %3d' % (names[p + 1] if p + 1 < len(names) else p, c['VALU'], c['v_perm'], c['v_pk'], c['SALU'], c['LDS'], c['VMEM']))
"
