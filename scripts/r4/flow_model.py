#!/usr/bin/env python3
"""Model of the tile-dataflow Cholesky (round 4): event-driven list scheduling of the task graph
   POTRF(k) / TRSM(i,k) / UPD(i,j,p0..p1) on NSLOT workgroup slots, with estimated task durations.
   Prints the simulated makespan for a dense T x T tile matrix -- the design model the round-3 verdict asked for
   BEFORE building (item 1: 'if a model shows it cannot beat 8.4 ms, write the model down').
"""
import heapq, sys, argparse, os
ap = argparse.ArgumentParser()
ap.add_argument('--T', type=int, default=71)
ap.add_argument('--slots', type=int, default=512)
ap.add_argument('--npmax', type=int, default=4)
ap.add_argument('--diag', type=float, default=42.0)
ap.add_argument('--hand', type=float, default=1.5)
ap.add_argument('--band', type=int, default=0, help='envelope: tile rows below the diagonal (0 = dense)')
ap.add_argument('--policy', default='edf')
ap.add_argument('--c0', type=float, default=14.0)
ap.add_argument('--c1', type=float, default=35.0)
a = ap.parse_args()
T, NS = a.T, a.slots
last = [min(T - 1, k + a.band) if a.band else T - 1 for k in range(T)]
def dur_upd128(n): return a.c0 + a.c1 * n
U64A, U64B = float(os.environ.get("U64A", 10)), float(os.environ.get("U64B", 20))
def dur_upd64(n): return U64A + U64B * n
D_T, TRSM0_T, UPD32_T, TRSM64_T = a.diag, 6.6, 7.3, float(os.environ.get("TRSM64", 28))
H = a.hand
# state
ver = {}         # tile (i,j) -> number of panels applied
pready = {}      # (i,k) -> time P_ik ready
wready = {}      # k -> time
busy_tile = set()
for j in range(T):
    for i in range(j, last[j] + 1): ver[(i, j)] = 0
def need(i, j):   # panels that apply to tile (i,j): p < j with last[p] >= i
    return [p for p in range(j) if last[p] >= i]
needs = {t: need(*t) for t in ver}
done_d = [False] * T; started_d = [False] * T
trsm_started = set()
events = []   # (time, kind, payload)
free = NS; now = 0.0
nfin = 0
total_busy = 0.0
def tile_time(i, j): return tile_t.get((i, j), 0.0)
tile_t = {}
def ready_tasks():
    """yield (prio, task) for all ready tasks; small T so brute force with pruning per call is ok"""
    out = []
    return out
# To stay fast: maintain candidate sets incrementally.  Simpler: at each scheduling point scan the 'frontier' columns.
import collections
chain_k = 0
def try_schedule():
    global free
    progressed = True
    while free > 0 and progressed:
        progressed = False
        best = None
        # 1. POTRF
        for k in range(T):
            if started_d[k]: continue
            t = (k, k)
            if t in busy_tile: break
            if ver[t] == len(needs[t]) and tile_time(k, k) <= now:
                best = (0, k, ('D', k)); 
            break
        if best is None:
            # 2. TRSM: tile (i,k) final & W_k ready
            cand = []
            for k in range(T):
                if not done_d[k]:
                    break
                for i in range(k + 1, last[k] + 1):
                    if (i, k) in trsm_started: continue
                    t = (i, k)
                    if t in busy_tile: continue
                    if ver[t] == len(needs[t]) and tile_time(i, k) <= now and wready[k] <= now:
                        cand.append(((1 if i == k + 1 else 2), k * 1000 + i, ('TRSM', i, k)))
            # 3. UPD
            # frontier: chain position = first k without D done
            kf = 0
            while kf < T and done_d[kf]: kf += 1
            for (i, j), v in ver.items():
                if (i, j) in busy_tile: continue
                nd = needs[(i, j)]
                if v >= len(nd): continue
                if tile_time(i, j) > now: continue
                # how many consecutive panels are ready
                n = 0
                while v + n < len(nd) and n < (a.npmax if not (i == j or j <= kf + 1) else 2):
                    p = nd[v + n]
                    if pready.get((i, p), 1e30) <= now and pready.get((j, p), 1e30) <= now: n += 1
                    else: break
                if n == 0: continue
                urgent = (i == j and j <= kf + 1)
                col_urgent = (j <= kf + 1)
                if urgent: pr = 1
                elif col_urgent: pr = 2
                else:
                    pr = 3
                    if a.policy == 'lazy':
                        # wait for a fuller merge unless the column is close to the chain
                        slack = j - kf
                        remaining = len(nd) - v
                        if n < min(a.npmax, remaining) and slack > 3 + remaining // a.npmax: continue
                cand.append((pr, j * 1000 + i, ('UPD', i, j, v, n, urgent, col_urgent)))
            if cand:
                best = min(cand)
        if best is None: break
        task = best[2]
        free -= 1; progressed = True
        if task[0] == 'D':
            k = task[1]; started_d[k] = True; busy_tile.add((k, k))
            heapq.heappush(events, (now + D_T, 'D', k))
        elif task[0] == 'TRSM':
            _, i, k = task; trsm_started.add((i, k)); busy_tile.add((i, k))
            d = TRSM0_T if i == k + 1 else TRSM64_T
            heapq.heappush(events, (now + d, 'TRSM', (i, k)))
        else:
            _, i, j, v, n, urgent, col_urgent = task
            busy_tile.add((i, j))
            d = (UPD32_T + 3 * (n - 1)) if urgent else (dur_upd64(n) if col_urgent else dur_upd128(n))
            heapq.heappush(events, (now + d, 'UPD', (i, j, n, d)))
            stats[n] += 1
stats = collections.Counter()
dtimes = []
try_schedule()
def complete(kind, pl):
    global free, total_busy
    if kind == 'WAKE': return
    free += 1
    if kind == 'D':
        k = pl; done_d[k] = True; wready[k] = now + H; busy_tile.discard((k, k)); dtimes.append(now)
    elif kind == 'TRSM':
        i, k = pl; pready[(i, k)] = now + H; busy_tile.discard((i, k))
    else:
        i, j, n, d = pl; ver[(i, j)] += n; tile_t[(i, j)] = now + H; busy_tile.discard((i, j)); total_busy += d
    heapq.heappush(events, (now + H + 1e-6, 'WAKE', None))
while events:
    now, kind, pl = heapq.heappop(events)
    complete(kind, pl)
    while events and events[0][0] <= now + 1e-9:
        _, kind, pl = heapq.heappop(events); complete(kind, pl)
    try_schedule()
print(f"T={T} slots={NS} npmax={a.npmax} policy={a.policy}: makespan {now/1000:.3f} ms; chain period avg {(dtimes[-1]-dtimes[0])/(len(dtimes)-1):.1f} us; "
      f"UPD busy {total_busy/1000/NS:.3f} ms/slot; merge histogram {dict(stats)}")
import os
if os.environ.get('GAPS'):
    g = [round(dtimes[i+1]-dtimes[i]) for i in range(len(dtimes)-1)]
    print('D-completion gaps:', g)
