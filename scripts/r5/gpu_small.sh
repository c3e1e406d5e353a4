#!/bin/bash
# kernel mix of the Bundler-sized problems (rocprofv3 --kernel-trace --stats), round 5
ulimit -c 0
cd /root/repo
O=gpurun_out/r5small; mkdir -p $O
for m in ${1:-14 50}; do
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_small && SMALL_NO_REF=1 SMALL_ONLY=$m timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_small -o st --output-format csv -- python /root/repo/scripts/small_problem_latency.py > /tmp/small_$m.out 2>/tmp/small_$m.err; cp $(find /tmp/p_small -name "*kernel_stats.csv" | head -1) /root/repo/$O/r05_small_${m}cams_kernel_stats.csv)
  cut -c1-200 /tmp/small_$m.out
  python - $O/r05_small_${m}cams_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lm = [r for r in rows if int(r["Calls"]) >= 60]
n_it = None
for r in lm:
    if "k_jacobian" in r["Name"]: n_it = int(r["Calls"])
print("iterations traced:", n_it)
acc = 0.0
for r in sorted(lm, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    per_it = float(r["TotalDurationNs"]) / n_it / 1e3
    acc += per_it
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:7.2f} us  per iteration {per_it:7.2f} us')
print("sum per iteration (us):", round(acc, 1))
PY
done
