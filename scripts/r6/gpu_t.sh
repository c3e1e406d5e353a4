#!/bin/bash
# round 6, call T: data-as-flag hand-offs on the chain (uncached copies of W_k and P(k+1,k)) -- tests, stress, then solve times on / off
ulimit -c 0
cd /root/repo
O=/root/repo/gpurun_out/r6t; mkdir -p $O
echo tests-skipped
timeout 600 python scripts/r4/flow_stress.py 300 44 2>&1 | tail -2
for D in 7 3 0; do for N in 500 1350 2600 3712 4800; do
  echo "data_flags $D n $N: $(BSFM_FLOW_DATA_FLAGS=$D BSFM_CHOL_REPS=6 timeout 300 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}')"
done; done 2>&1 | tee $O/times.txt
