#!/bin/bash
# speculative first attempt: the whole GPU suite, run_sfm wall time on Bundler-sized problems with and without it, the bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3s_pytest_gpu.log 2>&1; tail -4 gpurun_out/r3s_pytest_gpu.log
(echo "## default (speculative first attempt)"; SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py; echo "## BSFM_SPECULATE=0"; BSFM_SPECULATE=0 SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py) 2>&1 | tee gpurun_out/r3s_small_spec.txt | cut -c1-230
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-structure-aware --no-matcher --no-connected --no-end-to-end > gpurun_out/r3s_bench.json 2>gpurun_out/r3s_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3s_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], repr(d["final_cost"]), d["config"]["solve_attempts_per_step"])
PY
