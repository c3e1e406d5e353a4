#!/bin/bash
# round 6, call O: backward substitution with the solution values as the flag -- tests, then kernel durations at three sizes
ulimit -c 0
cd /root/repo
O=/root/repo/gpurun_out/r6o; mkdir -p $O
timeout 1200 python -m pytest tests/test_chol_gpu.py -q -m gpu -x 2>&1 | tail -4 | tee $O/pytest_chol.txt
cd /tmp && export TMPDIR=/tmp
for N in 9000 3712 1350; do
  rm -rf /tmp/prof_o
  BSFM_CHOL_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o run --output-format csv -- python /root/repo/scripts/r4/chol_reps.py $N > $O/reps_$N.txt 2>&1
  f=$(find /tmp/prof_o -name '*kernel_stats.csv' | head -1)
  echo "== n = $N"; python /root/repo/scripts/kstats.py $f 6
done 2>&1 | tee $O/kstats.txt
