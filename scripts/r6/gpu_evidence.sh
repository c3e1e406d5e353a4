#!/bin/bash
# round 6: the round's evidence in one call -- GPU suite, smoke, rocprofv3 kernel stats + PMC passes of the bench command (scripts/profile_round.sh),
# the driver's bench command, small-problem latency + launch traces, the matcher's own profile, distributed Cholesky timing, match-file sample check
ulimit -c 0
cd /root/repo
TAG=${1:-r06_cfg3_fd_final}
O=gpurun_out/r6ev; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $O/smoke.txt
bash scripts/profile_round.sh $TAG 2>&1 | tail -20
cp gpurun_out/prof/${TAG}_* gpurun_out/prof/LATEST $O/ 2>/dev/null
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
bash scripts/r6/gpu_small.sh final 2>&1 | tail -32; cp gpurun_out/r6small/*final* $O/ 2>/dev/null
timeout 900 python scripts/r6/dist_chol_timing.py 9000 2 2>&1 | tee $O/dist_chol_timing.txt
timeout 900 python scripts/r6/dist_chol_timing.py 9000 4 2>&1 | tail -6 | tee -a $O/dist_chol_timing.txt
timeout 1500 python scripts/r6/match_file_sample_check.py 200 2>&1 | tail -2 | tee $O/match_file_sample_check.txt
