#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3k
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-220
echo "--- small problems, phase timing forced ON (round-2 behaviour)"; BSFM_PHASE_TIMING=1 timeout 600 python scripts/small_problem_latency.py 2>&1 | grep cams | tee $OUT/small_timing_on.txt
echo "--- small problems, default"; timeout 600 python scripts/small_problem_latency.py 2>&1 | grep cams | tee $OUT/small_default.txt
