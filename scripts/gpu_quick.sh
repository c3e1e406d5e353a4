#!/bin/bash
# Generic quick GPU call: $1 = pytest selection (may be empty), $2.. = extra bench arguments; output under gpurun_out/quick/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/quick
mkdir -p $OUT
cd $ROOT
if [ -n "${1:-}" ]; then
  timeout 1500 python -m pytest $1 -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -8 $OUT/pytest.log
fi
shift
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-matcher --no-structure-aware "$@" > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms/step", d["ms_per_step"], "phases", d["phases_ms"], "create", d["config"]["problem_create_ms"], d["config"]["index_build_device_ms"])
print("schur", d["schur"]["ms"], d["schur"]["TFLOPs"], "connected", {k: d.get("connected_scene", {}).get(k) for k in ("ms_per_step", "phases_ms", "reduced_camera_blocks", "schur_tasks")})
PY
tail -3 $OUT/bench.err
