#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4d
SIZES=200,384,1799,3840,9000 MODES=flow,streams TRACE_OUT=gpurun_out/r4d/flow_trace_9000.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | tee gpurun_out/r4d/flow_check.txt | cut -c1-330
timeout 900 python -m pytest tests/test_chol_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r4d/pytest_chol.txt
