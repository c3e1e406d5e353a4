#!/usr/bin/env python3
"""Summary of a per-task trace of the tile-dataflow Cholesky (BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=...)."""
import sys
import numpy as np
path = sys.argv[1]
rows = [l for l in open(path) if not l.startswith('#')]
d = np.loadtxt(rows)
tk, ty, i, j, p0, npn, part, tt, tr, td, xcc, wg, queue = d.T
ms = td.max() - tt.min()
nw = len(np.unique(wg))
print(f'tasks {len(d)}, makespan {ms:.1f} us, workgroups that ran tasks {nw}')
names = ['POTRF', 'TRSM32', 'TRSM64', 'UPD32', 'UPD64', 'UPD128', 'FTRSM', 'FUPD']
for t in range(8):
    m = ty == t
    if not m.any(): continue
    run = td[m] - tr[m]; wait = tr[m] - tt[m]
    print(f'{names[t]:7s} n={m.sum():6d} run med {np.median(run):7.2f} mean {run.mean():7.2f} p90 {np.percentile(run, 90):7.2f} | wait med {np.median(wait):7.2f} mean {wait.mean():8.2f} | share of slot-time: busy {run.sum() / nw / ms:.3f} waiting {wait.sum() / nw / ms:.3f}')
    if t in (3, 4, 5):
        print('        by panels per visit: ' + ', '.join(f'np={n}: {int((m & (npn == n)).sum())} x {np.median((td - tr)[m & (npn == n)]):.1f}' for n in range(1, 9) if (m & (npn == n)).any()))
m = ty == 0; o = np.argsort(j[m]); pd = td[m][o]
print('POTRF completion gaps (us):', np.round(np.diff(pd)).astype(int).tolist())
print('POTRF run (us):', np.round((td[m] - tr[m])[o]).astype(int).tolist())
print('POTRF waited after its ticket (us):', np.round((tr[m] - tt[m])[o]).astype(int).tolist())
edges = np.arange(0, ms + 500, 500)
print('bulk workgroups, busy / waiting fraction per 500 us:')
b = queue == 0
nb = len(np.unique(wg[b]))
print('  ' + ' '.join(f'{(np.clip(td[b], a, c) - np.clip(tr[b], a, c)).sum() / (nb * (c - a)):.2f}/{(np.clip(tr[b], a, c) - np.clip(tt[b], a, c)).sum() / (nb * (c - a)):.2f}' for a, c in zip(edges[:-1], edges[1:])))
for l in open(path):
    if l.startswith('#P') and l.split()[1] in ('1:', '5:', '40:', '65:'): print(l.strip())
