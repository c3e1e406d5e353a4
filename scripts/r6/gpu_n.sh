#!/bin/bash
# round 6, call N: more knobs of the static order on top of laxity 2: panels per half-tile visit, simulated slots, urgent zone
ulimit -c 0
cd /root/repo
O=gpurun_out/r6n; mkdir -p $O
N=9000
run() { echo "== $*"; env "$@" BSFM_CHOL_REPS=6 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}'; echo; }
{ run X=1; run BSFM_FLOW_NPHALF=3; run BSFM_FLOW_NPHALF=4; run BSFM_FLOW_NPHALF=8; run BSFM_FLOW_SLOTS=490; run BSFM_FLOW_SLOTS=530; run BSFM_FLOW_URGENT=2; run BSFM_FLOW_URGENT=2 BSFM_FLOW_NPHALF=4;
  run BSFM_FLOW_ADAPT=64; run BSFM_FLOW_ADAPT=256; run BSFM_FLOW_LAZY=3; run BSFM_FLOW_LAZY=1; run BSFM_FLOW_CHAIN_WGS=13; run BSFM_FLOW_CHAIN_WGS=21; run X=1; } 2>&1 | tee $O/knobs2.txt
