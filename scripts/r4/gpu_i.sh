#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4i
(echo "## default (tile-dataflow Cholesky)"; SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py; echo "## BSFM_CHOL=streams (rounds 1-3 schedule)"; BSFM_CHOL=streams SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py) 2>&1 | tee gpurun_out/r4i/small_problem_latency.txt | cut -c1-200
bash scripts/profile_round.sh r04_cfg3_fd_a 2>&1 | tail -16
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4i/bench_default.json 2> gpurun_out/r4i/bench_default.err; tail -2 gpurun_out/r4i/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4i/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["phases_ms"], d["roofline"]["frac"], d["roofline"]["whole_factorisation"], d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cached"))
print({k: d["connected_scene"].get(k) for k in ("ms_per_step",)}, d["connected_scene"]["envelope_solver"])
print(d.get("matcher", {}).get("value"), d.get("matcher", {}).get("roofline", {}).get("frac"), d.get("end_to_end_run_sfm"))
PY
