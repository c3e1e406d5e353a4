#!/bin/bash
# round 6, call K: critical path of the chain in the tail of the static order (columns 50 - 66 of 71), raw task trace kept
ulimit -c 0
cd /root/repo
O=gpurun_out/r6k; mkdir -p $O
BSFM_CHOL_REPS=3 BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/flow_trace.txt timeout 300 python scripts/r4/chol_reps.py 9000 2>&1 | tail -4
python scripts/r4/trace_stats.py /tmp/flow_trace.txt > $O/trace_9000.txt 2>&1
python scripts/r5/critical_path.py /tmp/flow_trace.txt 40 52 55 58 60 62 64 > $O/critical_9000.txt 2>&1
gzip -c /tmp/flow_trace.txt > $O/flow_trace_9000.txt.gz
head -60 $O/critical_9000.txt | cut -c1-330
