#!/bin/bash
# round 3, call E: Schur task kernel -- triples per task x workgroups per CU; kernel times from rocprofv3 --kernel-trace --stats
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_index.py tests/test_cfg3_gpu.py tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end"
for cfg in "128 3" "64 3" "96 3" "160 3" "64 4" "128 4" "64 2" "128 2" "32 3" "192 3"; do
  set -- $cfg
  rm -rf /tmp/p_stats
  BSFM_SCHUR_CHUNK=$1 BSFM_SCHUR_WPS=$2 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st --output-format csv -- $B > /tmp/b.json 2> /tmp/st.err
  python - "$1" "$2" <<'PY'
import csv, glob, sys, json
f = glob.glob("/tmp/p_stats/**/*kernel_stats.csv", recursive=True)[0]
t = {}
for r in csv.DictReader(open(f)):
    for k in ("k_schur_tasks", "k_schur_prep", "k_schur_assemble", "k_jacobian", "k_zero_lower"):
        if k in r["Name"]: t[k] = float(r["AverageNs"]) / 1e3
d = json.load(open("/tmp/b.json"))
print("chunk", sys.argv[1], "wps", sys.argv[2], {k: round(v, 1) for k, v in t.items()}, "schur phase ms", d["phases_ms"]["schur"], "ms/step", d["ms_per_step"])
PY
done 2>&1 | tee $OUT/schur_sweep.txt
