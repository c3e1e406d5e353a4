#!/bin/bash
# round 6, call B: the dynamic bulk with the whole-matrix scan: sizes, parameter variants at 71 tile columns, trace
ulimit -c 0
cd /root/repo
O=gpurun_out/r6b; mkdir -p $O
export BSFM_FLOW_SPIN_MS=300
run() { echo "== $*"; env "$@" BSFM_CHOL_REPS=5 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep -v "^$" | tail -5 | tr '\n' ' '; echo; }
for N in 1024 1799 3712 5400; do run BSFM_FLOW_SCHED=dynamic; run BSFM_FLOW_SCHED=static; done 2>&1 | tee $O/sizes.txt
N=9000
{ run BSFM_FLOW_SCHED=static; run BSFM_FLOW_SCHED=dynamic; run BSFM_FLOW_NPMAX=8; run BSFM_FLOW_NPMAX=6; run BSFM_FLOW_NPMAX=2; run BSFM_DYN_HALVES=0; run BSFM_DYN_HALVES=3;
  run BSFM_FLOW_CHAIN_WGS=17; run BSFM_FLOW_CHAIN_WGS=27; run BSFM_DYN_SCAN=4; } 2>&1 | tee $O/variants_9000.txt
BSFM_CHOL_REPS=2 BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=/tmp/dyn_trace.txt timeout 200 python scripts/r4/chol_reps.py 9000 2>&1 | tail -3
python scripts/r6/dyn_trace_stats.py /tmp/dyn_trace.txt > $O/trace_9000.txt 2>&1; head -40 $O/trace_9000.txt | cut -c1-700
gzip -c /tmp/dyn_trace.txt > $O/dyn_trace_9000.txt.gz
