#!/bin/bash
# round 3, call H: envelope solver (tests + bench), multi-GPU hardening regression, full suite
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3h
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_chol_gpu.py -m gpu -q -p no:cacheprovider -x -k "envelope" > $OUT/pytest_env.log 2>&1; tail -15 $OUT/pytest_env.log | cut -c1-220
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log | cut -c1-220
timeout 600 python bench.py --steps 20 --warmup 5 --no-matcher --no-end-to-end --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "phases", d["phases_ms"])
print("structure_aware", {k: d["structure_aware"].get(k) for k in ("ms_per_step", "solve_ms", "schur_ms", "final_cost_rel_diff_vs_dense")})
print("connected", json.dumps(d.get("connected_scene"), indent=1))
PY
tail -5 $OUT/bench.err
