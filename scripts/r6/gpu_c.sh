#!/bin/bash
# round 6, call C: the static order's knobs on the round-5 kernel (panels per visit, lazy merging, urgent zone) at 71 tile columns
ulimit -c 0
cd /root/repo
O=gpurun_out/r6c; mkdir -p $O
export BSFM_FLOW_SCHED=static
N=9000
run() { echo "== $*"; env "$@" BSFM_CHOL_REPS=6 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}'; echo; }
{ run X=1; run BSFM_FLOW_NPMAX=6; run BSFM_FLOW_NPMAX=8; run BSFM_FLOW_LAZY=2; run BSFM_FLOW_NPMAX=8 BSFM_FLOW_LAZY=2; run BSFM_FLOW_NPMAX=8 BSFM_FLOW_LAZY=4; run BSFM_FLOW_NPMAX=6 BSFM_FLOW_LAZY=3;
  run BSFM_FLOW_URGENT=2; run BSFM_FLOW_NPMAX=8 BSFM_FLOW_URGENT=2; run BSFM_FLOW_ADAPT=0; run BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3; run BSFM_FLOW_TUPD128=20,27 BSFM_FLOW_THAND=3 BSFM_FLOW_TPOTRF=30;
  run BSFM_FLOW_SLOTS=440; run BSFM_FLOW_SLOTS=400 BSFM_FLOW_NPMAX=8; run X=1; } 2>&1 | tee $O/static_knobs_9000.txt
