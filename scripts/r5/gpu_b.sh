#!/bin/bash
# round 5, call B: row kernel v2 (everything indexed from LDS) -- parity, then A/B timing
ulimit -c 0
cd /root/repo
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_schur_rows_gpu.py tests/test_index.py -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_rows.txt
timeout 600 python -m pytest tests/test_cfg3_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_cfg3.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware"
for cfg in "rows_default:" "rows_off_wps3:BSFM_SCHUR_ROWS=0 BSFM_SCHUR_WPS=3" "rows_t1024:BSFM_SCHUR_ROW_TRIMAX=1024" "rows_L64:BSFM_SCHUR_ROW_L=64" "rows_L128:BSFM_SCHUR_ROW_L=128"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 400 $B > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - $O/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["schur"]; c = d.get("connected_scene", {})
    print(sys.argv[2], "ms/step", d["ms_per_step"], "schur", s["ms"], "prep", s["prep_ms"], "rows", s["rows_kernel_ms"], "tasks", s["tasks_kernel_ms"], s["row_kernel"],
          "create", d["config"]["problem_create_ms"], "| connected", c.get("ms_per_step"), c.get("phases_ms", {}).get("schur"), "env", c.get("envelope_solver", {}).get("ms_per_step"), c.get("envelope_solver", {}).get("schur_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
