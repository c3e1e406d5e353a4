#!/bin/bash
ulimit -c 0
cd /root/repo
mkdir -p gpurun_out/r4g
timeout 1200 python -m pytest tests/test_chol_gpu.py tests/test_ba_gpu.py tests/test_cfg3_gpu.py -x -q 2>&1 | tail -6 | tee gpurun_out/r4g/pytest_subset.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-structure-aware --no-matcher --no-end-to-end > gpurun_out/r4g/bench.json 2>gpurun_out/r4g/bench.err; tail -3 gpurun_out/r4g/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4g/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("phases_ms"), d.get("roofline",{}).get("frac"), d.get("connected_scene"))
PY
