#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4f
for n in 200 1799; do timeout 30 scripts/r4/_build/flow_dbg $n 2>&1 | tail -3; done
SIZES=384,1799,3840,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-250
BSFM_FLOW_CHAIN_WGS=0 SIZES=3840,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-250 | sed 's/^/onequeue /'
SIZES=9000 MODES=flow TRACE_OUT=gpurun_out/r4f/flow_trace_9000.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-150
SIZES=3840 MODES=flow TRACE_OUT=gpurun_out/r4f/flow_trace_3840.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-150
