#!/bin/bash
ulimit -c 0
cd /root/repo
O=gpurun_out/r6f; mkdir -p $O
{ python scripts/r6/known_traj.py; BSFM_CHOL=streams python scripts/r6/known_traj.py; } 2>&1 | tee $O/known_traj.txt
timeout 600 python -m pytest tests/test_boundary_link.py tests/test_ba_gpu.py -q -m gpu -x 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], d['phases_ms'])"
