#!/bin/bash
# diagnosis of the row kernel: which part of a pass costs the time
ulimit -c 0
cd /root/repo
O=gpurun_out/r5d; mkdir -p $O
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matcher --no-end-to-end --no-dense-valued --no-structure-aware --no-connected"
for cfg in "$@"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 400 $B > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - $O/bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d["schur"]
    print(sys.argv[2], "ms/step", d["ms_per_step"], "schur", s["ms"], "prep", s["prep_ms"], "rows", s["rows_kernel_ms"], "tasks", s["tasks_kernel_ms"], s["row_kernel"])
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
