#!/bin/bash
# round-3 closing check: the whole GPU suite, smoke(), the driver's bench command (timed), the default bench run (timed)
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3o_pytest_gpu.log 2>&1; tail -4 gpurun_out/r3o_pytest_gpu.log
t1=$(date +%s); echo "pytest -m gpu: $((t1-t0)) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t2=$(date +%s); echo "smoke: $((t2-t1)) s"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3o_bench_driver.json 2> gpurun_out/r3o_bench_driver.err
t3=$(date +%s); echo "driver bench: $((t3-t2)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3o_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["solve_attempts_per_step"], d["phases_ms"])
print("matcher", d.get("matcher",{}).get("value"), (d.get("matcher",{}).get("roofline") or {}).get("frac"))
print("e2e", (d.get("end_to_end_run_sfm") or {}).get("warm_call",{}).get("wall_s"), "connected", (d.get("connected_scene") or {}).get("ms_per_step"), ((d.get("connected_scene") or {}).get("envelope_solver") or {}).get("ms_per_step"))
print("cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cached"))
PY
