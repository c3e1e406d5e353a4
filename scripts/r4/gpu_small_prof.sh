#!/bin/bash
# kernel trace of the Bundler-sized problems (what fills an LM iteration at 50 / 200 cameras)
ulimit -c 0
cd /root/repo
export TMPDIR=/tmp
for m in ${SIZES:-50 200}; do
  rm -rf /tmp/sp$m
  SMALL_ONLY=$m SMALL_NO_REF=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sp$m -o sp --output-format csv -- python scripts/small_problem_latency.py 2>&1 | grep cams | cut -c1-200
  f=$(find /tmp/sp$m -name '*kernel_stats.csv' | head -1)
  mkdir -p gpurun_out/r4z; cp "$f" gpurun_out/r4z/small_${m}_kernel_stats.csv
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms in {calls} launches")
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):6.2f}%')
PY
done
