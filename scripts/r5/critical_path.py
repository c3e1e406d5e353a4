#!/usr/bin/env python3
"""Walks the critical path of a traced dataflow-Cholesky launch backwards from POTRF(k): for every task, the predecessor that finished last, and
whether the task started when that predecessor finished (dependency-bound) or later (its workgroup took the ticket late: slot-bound).
usage: critical_path.py raw_trace.txt k [k ...]"""
import sys
import numpy as np
rows = [l for l in open(sys.argv[1]) if not l.startswith('#')]
d = np.loadtxt(rows)
q, ty, ti, tj, p0, npn, part, tt, tr, td = [d[:, c] for c in range(10)]
ty = ty.astype(int); ti = ti.astype(int); tj = tj.astype(int); p0 = p0.astype(int); npn = npn.astype(int)
names = ['POTRF', 'TRSM32', 'TRSM64', 'UPD32', 'UPD64', 'UPD128', 'FTRSM', 'FUPD']
T = ti[ty == 0].max() + 1
def idx(mask): return np.nonzero(mask)[0]
def producers_of_panel(i, p):      # tasks that write P_ip
    return idx(((ty == 1) | (ty == 2)) & (ti == i) & (tj == p))
def updates_of_tile(i, j):
    return idx(((ty == 3) | (ty == 4) | (ty == 5)) & (ti == i) & (tj == j))
def preds(x):
    t, i, j = ty[x], ti[x], tj[x]
    if t == 0: return list(updates_of_tile(i, i))                       # POTRF(k): j holds k? (i == j == k)
    if t in (1, 2): return list(idx((ty == 0) & (ti == j))) + list(updates_of_tile(i, j))
    if t in (3, 4, 5):
        out = []
        for p in range(p0[x], p0[x] + npn[x]):
            out += list(producers_of_panel(i, p)) + list(producers_of_panel(j, p))
        out += [u for u in updates_of_tile(i, j) if td[u] <= tr[x] + 1e-9 and u != x]
        return out
    return []
for k in map(int, sys.argv[2:]):
    x = idx((ty == 0) & (ti == k))[0]
    print(f"--- POTRF({k}): ticket {tt[x]:.1f} start {tr[x]:.1f} done {td[x]:.1f}")
    for depth in range(12):
        ps = [p for p in preds(x) if td[p] <= tr[x] + 0.5]
        if not ps: break
        b = max(ps, key=lambda p: td[p])
        slack = tr[x] - td[b]
        late_ticket = tt[x] - td[b]
        print(f"  <- {names[ty[b]]}({ti[b]},{tj[b]}) p0 {p0[b]} np {npn[b]} part {int(part[b])}: ticket {tt[b]:.1f} start {tr[b]:.1f} done {td[b]:.1f} (ran {td[b] - tr[b]:.1f}, waited {tr[b] - tt[b]:.1f});  successor started {slack:.1f} us later" + (f"  [successor's ticket was taken {late_ticket:.1f} us AFTER this finished: slot-bound]" if late_ticket > 1.0 else ""))
        x = b
