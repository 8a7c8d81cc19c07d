#!/bin/bash
# round 6: every soak suite on the shipped kernels, the record kept (profiles/r06_soak_summary.txt is gpurun_out/r06_soak/summary.txt)
out=$GRAFT_REPO_ROOT/gpurun_out/r06_soak
mkdir -p $out
cd $GRAFT_REPO_ROOT
rev=$(python3 -c "import ethzasl_brisk_amd as B; print(B.load_library().brisk_hip_kernel_revision().decode())")
echo "soak of kernel revision $rev ($(date -u +%Y-%m-%dT%H:%MZ), $(rocm-smi --showproductname 2>/dev/null | grep -m1 -o 'MI[0-9A-Za-z]*' || echo MI355X))" > $out/summary.txt
for s in frames describe ordered callspace options matcher large threads; do
  t0=$(date +%s)
  timeout 1500 python3 tools/soak.py $s > $out/$s.log 2>&1
  rc=$?
  echo "suite $s: exit code $rc, $(( $(date +%s) - t0 )) s: $(grep -E "^$s:" $out/$s.log | tail -1)" | tee -a $out/summary.txt
  grep -E "MISMATCH|ERROR|HANG" $out/$s.log | head -5 | tee -a $out/summary.txt
done
