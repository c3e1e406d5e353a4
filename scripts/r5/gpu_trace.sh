#!/bin/bash
# task trace of the dataflow Cholesky: chain timeline + POTRF phases (n given as $1, default 3712 = 29 tile columns)
ulimit -c 0
cd /root/repo
O=gpurun_out/r5trace; mkdir -p $O
N=${1:-3712}; TAG=${2:-base}
BSFM_CHOL_REPS=3 BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/flow_trace.txt timeout 300 python scripts/r4/chol_reps.py $N 2>&1 | tail -4
python scripts/r4/trace_stats.py /tmp/flow_trace.txt > $O/trace_${N}_$TAG.txt 2>&1
[ -n "${RAW:-}" ] && python scripts/r5/critical_path.py /tmp/flow_trace.txt $RAW > $O/critical_${N}_$TAG.txt 2>&1
head -12 $O/trace_${N}_$TAG.txt | cut -c1-400
grep "^#P" $O/trace_${N}_$TAG.txt | tail -4
