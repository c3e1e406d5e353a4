#!/bin/bash
# round 6, call T: data-as-flag hand-offs on the chain (uncached copies of W_k and P(k+1,k)) -- tests, stress, then solve times on / off
ulimit -c 0
cd /root/repo
O=/root/repo/gpurun_out/r6t; mkdir -p $O
timeout 1500 python -m pytest tests/test_chol_gpu.py tests/test_multi_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 600 python scripts/r4/flow_stress.py 300 41 2>&1 | tail -2
for D in 1 0; do for N in 500 1350 2600 3712 5400 9000; do
  echo "data_flags $D n $N: $(BSFM_FLOW_DATA_FLAGS=$D BSFM_CHOL_REPS=6 timeout 300 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}')"
done; done 2>&1 | tee $O/times.txt
