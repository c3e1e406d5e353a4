import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# focus on the last factorisation: find last k_potrf_diag run of 71
diag = [i for i, r in enumerate(rows) if 'k_potrf_diag' in r['Kernel_Name']]
last = diag[-71:] if len(diag) >= 71 else diag
i0 = last[0]; t0 = int(rows[i0]['Start_Timestamp'])
sel = [r for r in rows[i0:] ]
end = max(int(r['End_Timestamp']) for r in sel if any(k in r['Kernel_Name'] for k in ('k_bwd_step', 'k_bwd_persistent')))
print("factor+solve span (us):", (end - t0) / 1e3)
byname = collections.defaultdict(list)
for r in sel:
    if int(r['Start_Timestamp']) > end: break
    n = r['Kernel_Name'].split('(')[0].replace('bsfm::', '')
    byname[n].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Queue_Id', '?')))
for n, v in byname.items():
    d = [b - a for a, b, _ in v]
    print(f"{n:28s} n={len(v):4d} total={sum(d)/1e3:9.1f}us avg={sum(d)/len(d)/1e3:8.2f} first={d[0]/1e3:8.2f} last={d[-1]/1e3:8.2f}")
# chain timeline for a few steps
dg = byname.get('k_potrf_diag', [])
for k in (1, 2, 10, 30, 50, 69):
    if k < len(dg):
        print(f"diag[{k}] start {dg[k][0]/1e3:9.1f} end {dg[k][1]/1e3:9.1f}  (prev diag end {dg[k-1][1]/1e3:9.1f})  step period {(dg[k][0]-dg[k-1][0])/1e3:7.1f} us")
