#!/bin/bash
# round 3, call G: device CRS as the first device work of a process (was flaky), full suite, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3g
mkdir -p $OUT
cd $ROOT
for i in 1 2 3 4 5 6 7 8; do timeout 120 python -m pytest tests/test_index.py -m gpu -q -p no:cacheprovider -k "device_crs or run_sfm_through" 2>&1 | tail -1; done
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log | cut -c1-200
timeout 600 python bench.py --steps 20 --warmup 5 --no-matcher > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "phases", d["phases_ms"])
print("hbm", {k: (v["ms"], v["frac_of_8TBps"]) for k, v in d["hbm_kernels"].items() if isinstance(v, dict)})
print("schur", d["schur"]["ms"], d["schur"]["frac_of_fp64_peak"])
print("structure_aware", {k: d["structure_aware"].get(k) for k in ("ms_per_step", "schur_ms", "final_cost_rel_diff_vs_dense")})
print("connected", {k: d.get("connected_scene", {}).get(k) for k in ("ms_per_step", "phases_ms")})
print("e2e", d["end_to_end_run_sfm"]["warm_call"])
PY
tail -5 $OUT/bench.err
