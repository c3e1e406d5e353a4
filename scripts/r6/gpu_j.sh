#!/bin/bash
# round 6, call J: chain workgroups sharing their CUs (experiment), camera table as its own kernel again at the headline size
ulimit -c 0
cd /root/repo
O=gpurun_out/r6j; mkdir -p $O
N=9000
run() { echo "== $*"; env "$@" BSFM_CHOL_REPS=6 timeout 120 python scripts/r4/chol_reps.py $N 2>&1 | grep "rep [1-5]" | awk '{printf "%s ", $(NF-1)}'; echo; }
{ run X=1; run BSFM_FLOW_CHAIN_SHARED=1; run BSFM_FLOW_CHAIN_SHARED=1 BSFM_FLOW_CHAIN_WGS=27; run BSFM_FLOW_CHAIN_SHARED=1 BSFM_FLOW_CHAIN_WGS=33; run X=1;
  N=3712; run X=1; run BSFM_FLOW_CHAIN_SHARED=1; N=5400; run X=1; run BSFM_FLOW_CHAIN_SHARED=1; } 2>&1 | tee $O/chain_shared.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['phases_ms'])" | tee $O/bench.txt
SMALL_NO_REF=1 timeout 600 python scripts/small_problem_latency.py 2>&1 | cut -c1-200 | tee $O/latency.txt
