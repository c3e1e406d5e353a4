#!/bin/bash
# round 4 closing evidence on HEAD: GPU suite, smoke, Bundler-sized problems, rocprofv3 kernel stats + PMC passes + task trace, the driver's bench command
ulimit -c 0
cd /root/repo
TAG=${1:-final}
mkdir -p gpurun_out/r4z
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r4z/pytest_gpu_$TAG.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r4z/pytest_gpu_$TAG.txt
(echo "## default (tile-dataflow Cholesky)"; SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py; echo "## BSFM_CHOL=streams (rounds 1-3 schedule)"; BSFM_CHOL=streams SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py) 2>&1 | tee gpurun_out/r4z/small_problem_latency_$TAG.txt | cut -c1-120
bash scripts/profile_round.sh r04_cfg3_fd_$TAG 2>&1 | tail -6 | cut -c1-160
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4z/bench_default_$TAG.json 2> gpurun_out/r4z/bench_default_$TAG.err; tail -2 gpurun_out/r4z/bench_default_$TAG.err
python - $TAG <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r4z/bench_default_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["solve_attempts_per_step"], d["phases_ms"], d["roofline"]["frac"], d["roofline"]["whole_factorisation"], d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cached"))
c=d["connected_scene"]; print(c["ms_per_step"], c["phases_ms"], c["envelope_solver"])
print(d["structure_aware"]["ms_per_step"], d.get("matcher", {}).get("value"), d.get("matcher", {}).get("roofline", {}).get("frac"), d.get("end_to_end_run_sfm", {}).get("warm_call", {}).get("phases_ms"))
PY
