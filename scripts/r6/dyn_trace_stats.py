#!/usr/bin/env python3
"""Summary of a per-task trace of the DYNAMIC tile-dataflow Cholesky (BSFM_FLOW_TRACE=1 BSFM_FLOW_TRACE_FILE=..., chol_dyn.hip.h):
per role durations, time from scan to claim, slot-time split of the bulk workgroups (busy / scanning with nothing to claim), panels per visit,
chain timeline."""
import sys
import numpy as np
path = sys.argv[1]
rows = [l for l in open(path) if not l.startswith('#')]
d = np.loadtxt(rows)
q, ty, i, j, p0, npn, part, tt, tr, td, xcc, wg, queue, idle = d.T
ms = td.max() - tt[tt > 0].min()
names = ['POTRF', 'TRSM32', 'TRSM64', 'UPD32', 'UPD64', 'UPD128', 'FTRSM', 'FUPD']
bulk = queue == 0
nb = len(np.unique(wg[bulk]))
print(f'records {len(d)}, makespan {ms:.1f} us, bulk workgroups that ran tasks {nb}, chain workgroups {len(np.unique(wg[~bulk]))}')
for q_ in (0, 1, 3):
    for t in range(8):
        m = (ty == t) & (queue == q_)
        if not m.any(): continue
        run = td[m] - tr[m]; wait = tr[m] - tt[m]
        print(f'{"bulk " if q_ == 0 else "chain" if q_ == 1 else "potrf"} {names[t]:7s} n={m.sum():6d} run med {np.median(run):7.2f} mean {run.mean():7.2f} p90 {np.percentile(run, 90):7.2f} | '
              f'scan/ticket -> start med {np.median(wait):7.2f} mean {wait.mean():8.2f}')
        if t in (4, 5, 7):
            print('        by panels per visit: ' + ', '.join(f'np={n}: {int((m & (npn == n)).sum())} x {np.median((td - tr)[m & (npn == n)]):.1f}' for n in range(1, 17) if (m & (npn == n)).any()))
# slot-time of the bulk workgroups: busy = claimed -> done; scan = last successful scan -> claimed; the rest = idle scans / sleeping
busy = (td[bulk] - tr[bulk]).sum(); scan = (tr[bulk] - tt[bulk]).sum()
print(f'bulk slot-time ({nb} workgroups x {ms:.0f} us): busy {busy / nb / ms:.3f}, successful scan+claim {scan / nb / ms:.3f}, idle {1 - (busy + scan) / nb / ms:.3f}')
edges = np.arange(0, ms + 500, 500)
t0 = tt[tt > 0].min()
print('bulk workgroups, busy / scan fraction per 500 us:')
print('  ' + ' '.join(f'{(np.clip(td[bulk] - t0, a, c) - np.clip(tr[bulk] - t0, a, c)).sum() / (nb * (c - a)):.2f}/{(np.clip(tr[bulk] - t0, a, c) - np.clip(tt[bulk] - t0, a, c)).sum() / (nb * (c - a)):.2f}' for a, c in zip(edges[:-1], edges[1:])))
m = ty == 0; o = np.argsort(j[m]); pd = td[m][o]
print('POTRF completion gaps (us):', np.round(np.diff(pd)).astype(int).tolist())
print('POTRF run (us):', np.round((td[m] - tr[m])[o]).astype(int).tolist())
print('POTRF waited after its ticket (us):', np.round((tr[m] - tt[m])[o]).astype(int).tolist())
u = (ty == 5) | (ty == 4)
print(f'update visits {int(u.sum())}: tile products {int(((npn * np.where(ty == 5, 2, 1))[u]).sum() / 2)}, mean panels per visit {npn[u].mean():.2f}, visits after an idle scan {(idle[u & bulk] > 0).mean():.2f}')
for l in open(path):
    if l.startswith('#P') and l.split()[1] in ('1:', '5:', '40:', '65:'): print(l.strip())
