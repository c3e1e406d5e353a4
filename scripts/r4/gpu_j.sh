#!/bin/bash
ulimit -c 0
cd /root/repo
for lz in 0 2 3 4; do BSFM_FLOW_LAZY=$lz SIZES=3600,9000 MODES=flow CASE_TIMEOUT=60 timeout 600 python scripts/r4/flow_check.py 2>&1 | cut -c1-20,100-250 | sed "s/^/lazy=$lz /"; done
BSFM_TEST_HUGE=1 timeout 900 python -m pytest tests/test_chol_gpu.py -x -q -k more_than_240 2>&1 | tail -3
