#!/bin/bash
# Round 5: FETCH_SIZE, L2 hits / misses and L1 (TCP) request counts of the Schur kernels on both config-3 scenes (two populations by grid size).
ulimit -c 0
cd /root/repo
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-matcher --no-structure-aware --no-end-to-end --no-dense-valued"
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr " " "_")
  rm -rf /tmp/pc_$tag
  timeout 280 rocprofv3 --pmc $c -d /tmp/pc_$tag -o c --output-format csv -- $CMD > /dev/null 2> /tmp/pc.err
  f=$(find /tmp/pc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "k_schur_tasks" in r["Kernel_Name"] or "k_schur_prep" in r["Kernel_Name"]:
        acc[(r["Counter_Name"], r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
for (name, kn, grid), v in sorted(acc.items()):
    print(f"{name:30s} {kn:30s} grid {grid:9d}: {len(v):3d} launches, mean {sum(v)/len(v):.4g}")
PY
done
