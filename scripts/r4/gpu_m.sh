#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4m
SIZES=384,900,1799,3600,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250
BSFM_FLOW_TPOTRF=56 SIZES=9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250 | sed 's/^/tpotrf56 /'
SIZES=9000 MODES=flow TRACE_OUT=gpurun_out/r4m/flow_trace_9000.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-60
timeout 600 python -m pytest tests/test_chol_gpu.py tests/test_ba_gpu.py -x -q 2>&1 | tail -2
