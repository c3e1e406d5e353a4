#!/bin/bash
# round 6, call A: bring-up of the dynamic bulk (k_chol_dyn): accuracy + device time against the static order at several sizes, trace at 71 tile columns, the Cholesky tests
ulimit -c 0
cd /root/repo
O=gpurun_out/r6a; mkdir -p $O
export BSFM_FLOW_SPIN_MS=300
for n in 256 384 1024 1799 3712 9000; do
  for s in dynamic static; do
    echo "== n=$n sched=$s"
    BSFM_FLOW_SCHED=$s BSFM_CHOL_REPS=5 timeout 120 python scripts/r4/chol_reps.py $n 2>&1 | grep -v "^$" | tail -7
  done
done 2>&1 | tee $O/sizes.txt
BSFM_CHOL_REPS=2 BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/dyn_trace.txt timeout 200 python scripts/r4/chol_reps.py 9000 2>&1 | tail -3
python scripts/r6/dyn_trace_stats.py /tmp/dyn_trace.txt > $O/trace_9000.txt 2>&1; head -40 $O/trace_9000.txt | cut -c1-600
gzip -c /tmp/dyn_trace.txt > $O/dyn_trace_9000.txt.gz
unset BSFM_FLOW_SPIN_MS
timeout 1500 python -m pytest tests/test_chol_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_chol.txt
