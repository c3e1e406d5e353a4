#!/bin/bash
# round 3, call A: the full GPU parity suite (new: config-3 three iterations, forced failure branches, reference-pinned ray angles,
# device-side vmask -> CRS), the stagger microbenchmark of the bulk kernel, the default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 scripts/_bin/ubench_syrk_stagger > $OUT/stagger.txt 2>&1; echo "stagger rc=$?"
cat $OUT/stagger.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "phases", d["phases_ms"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms")}, d["roofline"]["whole_factorisation"])
print("structure_aware", d.get("structure_aware"))
print("end_to_end", json.dumps(d.get("end_to_end_run_sfm"), indent=1))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), (d.get("cpu_baseline", {}).get("live_sample") or {}).get("value"))
PY
tail -5 $OUT/bench.err
