#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3j
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log | cut -c1-220
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "phases", d["phases_ms"])
print("connected", d["connected_scene"]["ms_per_step"], d["connected_scene"]["envelope_solver"])
print("matcher", d["matcher"]["value"], d["matcher"]["roofline"]["frac"])
PY
tail -3 $OUT/bench.err
