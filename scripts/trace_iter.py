import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
jac = [i for i, r in enumerate(rows) if 'k_jacobian' in r['Kernel_Name']]
i0, i1 = jac[-2], jac[-1]
t0, t1 = int(rows[i0]['Start_Timestamp']), int(rows[i1]['Start_Timestamp'])
print("iteration span (us):", (t1 - t0) / 1e3)
by = collections.defaultdict(lambda: [0, 0.0])
busy_end = t0; gap = 0.0
for r in rows[i0:i1]:
    n = r['Kernel_Name'].split('(')[0].replace('bsfm::', '').replace('void ', '')
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    by[n][0] += 1; by[n][1] += d
    s = int(r['Start_Timestamp'])
    if s > busy_end: gap += s - busy_end
    busy_end = max(busy_end, int(r['End_Timestamp']))
print("idle gaps (no kernel running) us:", gap / 1e3)
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{n[:40]:40s} n={c:4d} total={d/1e3:9.1f} us")
