#!/bin/bash
# device block cache: the whole GPU suite (every test file builds and tears down problems in ONE process: reuse of dirty blocks is exercised),
# run_sfm wall time on Bundler-sized problems with the cache and without
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3s_pytest_gpu.log 2>&1; tail -4 gpurun_out/r3s_pytest_gpu.log
(echo "## cache on (default)"; SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py; echo "## BSFM_DEVCACHE_MB=0"; BSFM_DEVCACHE_MB=0 SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py) 2>&1 | tee gpurun_out/r3s_small.txt | cut -c1-200
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3s_bench.json 2>gpurun_out/r3s_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3s_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["final_cost"], d["config"]["problem_create_ms"], (d.get("end_to_end_run_sfm") or {}).get("warm_call",{}).get("phases_ms"))
PY
