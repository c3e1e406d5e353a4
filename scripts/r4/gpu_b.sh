#!/bin/bash
# round 4, call B: what does the flow path do at 2 tiles?  (short: core dumps off, 25 s per case, full stderr)
ulimit -c 0
mkdir -p gpurun_out/r4b
cd /root/repo
for n in 200; do
  echo "=== n=$n flow"; BSFM_CHOL_REPS=1 timeout 25 python scripts/r4/flow_check.py child $n 1 2>&1 | tail -25
  echo "exit: $?"
done
dmesg 2>/dev/null | tail -15
