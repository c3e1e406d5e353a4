#!/bin/bash
cd /root/repo
N=${1:-9000}
for L in 0 3 6 10; do echo -n "lookahead $L: "; BSFM_FLOW_LOOKAHEAD=$L BSFM_CHOL_REPS=5 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [34]" | sed "s/.*rep [34]://" | tr '\n' ' '; echo; done
echo -n "lookahead 6, measured durations: "; BSFM_FLOW_LOOKAHEAD=6 BSFM_FLOW_TPOTRF=30 BSFM_FLOW_THAND=3 BSFM_FLOW_TCHAIN=4.2,19,4.6,2.4 BSFM_FLOW_TUPD64=3.6,13.3 BSFM_FLOW_TUPD128=15,26 BSFM_CHOL_REPS=5 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [34]" | sed "s/.*rep [34]://" | tr '\n' ' '; echo
