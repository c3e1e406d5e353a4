#!/bin/bash
# round 3, call F: Schur task kernel split timing (no gathers / no matrix instructions / gathers + core only), point-major prep kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r3f
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_index.py tests/test_cfg3_gpu.py tests/test_ba_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end"
for cfg in "160 3 0" "160 4 0" "160 3 1" "160 3 2" "160 3 3" "192 4 0"; do
  set -- $cfg
  rm -rf /tmp/p_stats
  BSFM_SCHUR_CHUNK=$1 BSFM_SCHUR_WPS=$2 BSFM_SCHUR_DBG=$3 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o st --output-format csv -- $B > /tmp/b.json 2> /tmp/st.err
  python - "$1" "$2" "$3" <<'PY'
import csv, glob, sys, json
f = glob.glob("/tmp/p_stats/**/*kernel_stats.csv", recursive=True)[0]
t = {}
for r in csv.DictReader(open(f)):
    for k in ("k_schur_tasks", "k_schur_prep", "k_schur_assemble", "k_jacobian", "k_residual", "k_point_blocks", "k_cam_blocks<", "k_backsub"):
        if k in r["Name"]: t[k] = float(r["AverageNs"]) / 1e3
print("chunk", sys.argv[1], "wps", sys.argv[2], "dbg", sys.argv[3], {k: round(v, 1) for k, v in t.items()})
PY
done 2>&1 | tee $OUT/schur_split.txt
