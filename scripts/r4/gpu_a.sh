#!/bin/bash
# round 4, call A: bring-up of the tile-dataflow Cholesky
mkdir -p gpurun_out/r4a
TRACE_OUT=gpurun_out/r4a/flow_trace_9000.txt CASE_TIMEOUT=150 timeout 1500 python scripts/r4/flow_check.py 2>&1 | tee gpurun_out/r4a/flow_check.txt | cut -c1-400
