#!/bin/bash
# the whole GPU suite + smoke + the driver's bench command
ulimit -c 0
cd /root/repo
O=gpurun_out/r5full; mkdir -p $O
TAG=${1:-a}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu_$TAG.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/pytest_gpu_$TAG.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_$TAG.json 2> $O/bench_default_$TAG.err; tail -2 $O/bench_default_$TAG.err
python - $O/bench_default_$TAG.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["solve_attempts_per_step"], d["phases_ms"], d["roofline"]["frac"], d["roofline"]["frac_useful"], d["cpu_baseline"].get("value"))
print("dense_valued", d.get("dense_valued_S"))
c=d["connected_scene"]; print("connected", c["ms_per_step"], c["phases_ms"], c["envelope_solver"])
print(d["structure_aware"]["ms_per_step"], d.get("matcher", {}).get("value"), d.get("matcher", {}).get("roofline", {}).get("frac"), d.get("end_to_end_run_sfm", {}).get("warm_call", {}).get("phases_ms"))
PY
