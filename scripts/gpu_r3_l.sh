#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3l
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_chol_gpu.py -m gpu -q -p no:cacheprovider -x --tb=short 2>&1 | tail -15 | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log | cut -c1-220
echo "--- small problems, narrow steps ON"; SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py 2>&1 | grep cams | cut -c1-200 | tee $OUT/small_narrow.txt
echo "--- small problems, narrow steps OFF"; BSFM_NARROW=0 SMALL_NO_REF=1 timeout 300 python scripts/small_problem_latency.py 2>&1 | grep cams | cut -c1-200 | tee $OUT/small_wide.txt
for nv in 8 0; do
BSFM_NARROW=$nv timeout 600 python bench.py --steps 20 --warmup 5 --no-matcher --no-end-to-end --no-cpu-baseline > $OUT/bench_narrow$nv.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_narrow$nv.json"))
print("NARROW=$nv value", d["value"], "ms/step", d["ms_per_step"], "solve", d["phases_ms"]["solve"], "connected", d["connected_scene"]["ms_per_step"], d["connected_scene"]["phases_ms"]["solve"], "envelope", d["connected_scene"]["envelope_solver"]["ms_per_step"], d["connected_scene"]["envelope_solver"]["solve_ms"])
PY
done
