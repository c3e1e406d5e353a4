#!/bin/bash
# round 6, call L: least-laxity priority of the static order (BSFM_FLOW_LAXITY), n = 9 000
ulimit -c 0
cd /root/repo
O=gpurun_out/r6l; mkdir -p $O
N=9000
run() { echo "== $*"; env "$@" BSFM_CHOL_REPS=6 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}'; echo; }
{ run X=1; run BSFM_FLOW_BAND=1; run BSFM_FLOW_BAND=2; run BSFM_FLOW_BAND=3; run BSFM_FLOW_BAND=6; run BSFM_FLOW_BAND=2 BSFM_FLOW_LAXITY=2; run BSFM_FLOW_BAND=3 BSFM_FLOW_LAXITY=2; run BSFM_FLOW_LAXITY=2; run BSFM_FLOW_LAXITY=1.5; run BSFM_FLOW_LAXITY=2.5; run BSFM_FLOW_LAXITY=3; run BSFM_FLOW_BAND=2 BSFM_FLOW_ADAPT=0;
  run X=1; } 2>&1 | tee $O/laxity2.txt
BSFM_FLOW_BAND=2 BSFM_CHOL_REPS=3 BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/flow_trace.txt timeout 300 python scripts/r4/chol_reps.py 9000 2>&1 | tail -2
python scripts/r4/trace_stats.py /tmp/flow_trace.txt > $O/trace_band2.txt 2>&1; head -30 $O/trace_band2.txt | cut -c1-300
