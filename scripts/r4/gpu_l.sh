#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4l
for n in 200 1799; do timeout 30 scripts/r4/_build/flow_dbg $n 2>&1 | tail -1; done
SIZES=384,1799,3600,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250
SIZES=3600 MODES=flow TRACE_OUT=gpurun_out/r4l/flow_trace_3600.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-60
grep '^#P' gpurun_out/r4l/flow_trace_3600.txt | sed -n '10,12p'
timeout 300 python -m pytest tests/test_chol_gpu.py -x -q 2>&1 | tail -2
