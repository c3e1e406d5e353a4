#!/bin/bash
# round 6, call D: GPU suite after the small-problem fusions, small-problem traces + latency, speculation threshold A/B
ulimit -c 0
cd /root/repo
O=gpurun_out/r6d; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.txt
bash scripts/r6/gpu_small.sh fused 2>&1 | tail -60
cp gpurun_out/r6small/*fused* $O/ 2>/dev/null
echo "== speculate on for every size"; BSFM_SPECULATE=1 SMALL_NO_REF=1 timeout 600 python scripts/small_problem_latency.py 2>&1 | cut -c1-260 | tee $O/latency_speculate_all.txt
