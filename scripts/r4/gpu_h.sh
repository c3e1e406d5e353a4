#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4h
for n in 200 1799; do timeout 30 scripts/r4/_build/flow_dbg $n 2>&1 | tail -1; done
SIZES=384,900,1799,3600,9000 MODES=flow,streams CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250
for cw in 8 24; do BSFM_FLOW_CHAIN_WGS=$cw SIZES=3600,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250 | sed "s/^/chainwgs=$cw /"; done
for u in 0 2; do BSFM_FLOW_URGENT=$u SIZES=9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250 | sed "s/^/urgent=$u /"; done
BSFM_FLOW_NPMAX=6 SIZES=9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250 | sed "s/^/npmax=6 /"
SIZES=3600 MODES=flow TRACE_OUT=gpurun_out/r4h/flow_trace_3600.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-80
timeout 300 python -m pytest tests/test_chol_gpu.py -x -q 2>&1 | tail -2
