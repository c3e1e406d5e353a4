#!/bin/bash
# round 6, call E: GPU suite after the second batch of small-problem fusions + pinned tests, small-problem traces / latency, a short headline bench
ulimit -c 0
cd /root/repo
O=gpurun_out/r6e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.txt
bash scripts/r6/gpu_small.sh fused2 2>&1 | tail -45
cp gpurun_out/r6small/*fused2* $O/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected > $O/bench_short.json 2> $O/bench_short.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6e/bench_short.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "phases", d["phases_ms"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_useful", "avg_launch_ms")})
print("hbm", {k: (v.get("ms"), v.get("frac_of_8TBps"), v.get("counter_over_algorithmic")) for k, v in d["hbm_kernels"].items() if isinstance(v, dict)})
PY
