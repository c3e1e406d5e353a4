#!/bin/bash
# estimated task durations of the list scheduler (chol_flow_sched.h FlowParams) against the measured makespan, n = $1 (default 9000)
cd /root/repo
N=${1:-9000}
run() { echo -n "$1: "; env $2 BSFM_CHOL_REPS=5 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [34]" | sed "s/.*rep [34]://" | tr '\n' ' '; echo; }
run "defaults" "X=1"
run "potrf 30" "BSFM_FLOW_TPOTRF=30"
run "potrf 30, hand 3" "BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3"
run "all measured" "BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26"
run "all measured, hand 1.5" "BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=1.5 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26"
run "all measured, np 3" "BSFM_FLOW_NPMAX=3 BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26"
run "all measured, urgent 2" "BSFM_FLOW_URGENT=2 BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26"
run "all measured, adapt 0" "BSFM_FLOW_ADAPT=0 BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26"
