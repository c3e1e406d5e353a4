"""The bulk stream of one dense factorisation from a rocprofv3 kernel trace: every k_syrk_update launch with its grid, duration, the gap
to the end of the previous bulk launch and the ideal duration of its workgroups at the stand-alone rate.
usage: python scripts/trace_bulk.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
jac = [i for i, r in enumerate(rows) if 'k_jacobian' in r['Kernel_Name']] + [len(rows)]
best = []
for a, b in zip(jac, jac[1:]):
    seg = [r for r in rows[a:b] if 'k_syrk_update' in r['Kernel_Name']]
    if len(seg) >= len(best):
        best = seg
        allseg = rows[a:b]
diag = [r for r in allseg if 'k_potrf_diag' in r['Kernel_Name']]
t0 = int(diag[0]['Start_Timestamp'])
us = lambda t: (int(t) - t0) / 1e3
prev_end = None
tot_busy = tot_gap = 0.0
print("launch  start_us  dur_us  gap_us  workgroups  rounds(512)  us_per_round")
for i, r in enumerate(best):
    s, e = us(r['Start_Timestamp']), us(r['End_Timestamp'])
    wg = int(r.get('Grid_Size', 0)) // max(int(r.get('Workgroup_Size', 512)), 1) if r.get('Grid_Size') else 0
    gap = s - prev_end if prev_end is not None else 0.0
    rounds = wg / 512.0
    print("%4d  %9.1f  %7.1f  %6.1f  %6d  %6.2f  %7.1f" % (i, s, e - s, gap, wg, rounds, (e - s) / max(rounds, 1e-9) if wg else 0))
    tot_busy += e - s; tot_gap += max(gap, 0.0)
    prev_end = e
print("bulk busy %.1f us, gaps between consecutive bulk launches %.1f us, last diag end %.1f us" % (tot_busy, tot_gap, us(diag[-1]['End_Timestamp'])))
