#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4e
echo "=== harness 384"; timeout 40 scripts/r4/_build/flow_dbg 384 1 2>&1 | cut -c1-200 | grep -E 'BAD|STUCK|bad prefixes|marks' | head
echo "=== harness 200"; timeout 40 scripts/r4/_build/flow_dbg 200 1 2>&1 | cut -c1-200 | grep -E 'BAD|STUCK|bad prefixes|marks' | head
SIZES=384,3840 MODES=flow TRACE_OUT=gpurun_out/r4e/flow_trace_3840.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-250
grep '^#P' gpurun_out/r4e/flow_trace_3840.txt | sed -n '1,3p;20,24p'
SIZES=1799,3840,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-250
SIZES=9000 MODES=flow TRACE_OUT=gpurun_out/r4e/flow_trace_9000.txt CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-250
