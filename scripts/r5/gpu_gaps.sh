#!/bin/bash
# idle time between the kernels of one LM iteration of bench.py's headline (rocprofv3 --kernel-trace), round 5
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_gap
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_gap -o g --output-format csv -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-matcher --no-connected --no-structure-aware --no-end-to-end --no-dense-valued ${BENCH_EXTRA:-} > /dev/null 2>/tmp/gap.err
python - $(find /tmp/p_gap -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last iterations: from the last but one k_jacobian to the last
idx = [i for i, r in enumerate(rows) if "k_jacobian" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f'{(s - t0) / 1e3:9.1f} us  +{gap:7.1f} gap  {(e - s) / 1e3:8.1f} us  {r["Kernel_Name"][:90]}')
    if r is not rows[b]: busy += e - s
    prev_end = max(prev_end or 0, e)
print("iteration", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "us, kernels", busy / 1e3, "us")
PY
