#!/usr/bin/env python3
"""Model of a DISTRIBUTED tile-dataflow Cholesky over N GPUs (round 5, VERDICT r4 item 7: model and CPU replay only -- nothing here runs on a GPU).

The single-GPU kernel (csrc/chol_flow.hip.h) executes a static order of tile tasks with per-tile hand-offs.  DESIGN.md argued against
distributing it with a model in which the whole panel of column k is broadcast ON the chain at every step.  In a dataflow that is not what
happens: the owner of column k + 1 needs ONE tile of panel k -- P(k+1, k) -- to go on; the other tiles of the panel stream behind the bulk.

Model
  * owner(j) = j mod N owns every tile of tile column j: POTRF(j), TRSM(i, j) and all updates UPD(i, j; panels) run on that GPU
    (owner computes, 1-D column-cyclic: the same right-looking task graph as chol_flow_sched.h, split by column);
  * a finished panel tile P(i, p) is forwarded by its owner to every other GPU over that pair's own xGMI link (full mesh, 7 links per
    GPU): arrival = completion + hop latency + queueing on the link + 128 KB / link bandwidth.  Tiles of a panel are sent in row order,
    the chain's tile P(p+1, p) first;
  * each GPU runs an event-driven list scheduler over its own workgroup slots with the priorities of chol_flow_sched.h (POTRF, the
    chain's tiles, the next column, the bulk in column order), np <= np_max panels per visit of a tile;
  * task durations = the medians of the round-5 trace at 71 tile columns (profiles/r05_cfg3_fd_final_chain_timeline.txt).
The model at N = 1 gives 5.2 ms where the kernel measures 6.3 (it knows nothing of contention on L2 / HBM or of ticket-order waiting), so
read the N > 1 numbers as RATIOS to the model's own N = 1.

simulate() returns the executed task list; tests/test_chol_flow_sched.py replays it in numpy on per-GPU copies of the panel tiles (a
tile is only readable on a GPU after its modelled arrival) and checks the factor against numpy's Cholesky.
"""
import argparse
import heapq

POTRF, TRSM, UPD = 0, 1, 2


def simulate(T, N, slots=496, np_max=4, t_potrf=29.5, t_trsm32=4.2, t_trsm64=19.0, t_upd32=(4.6, 2.4), t_upd64=(3.6, 13.3),
             t_upd128=(15.0, 26.0), hand=3.0, hop=5.0, link_gbs=100.0, urgent_cols=1):
    tile_us = 128 * 128 * 8 / (link_gbs * 1e3)          # one 128 KB tile over one link, microseconds
    own = lambda j: j % N
    ver = {(i, j): 0 for j in range(T) for i in range(j, T)}         # panels applied so far
    busy = set()
    avail = [dict() for _ in range(N)]      # avail[g][(i, p)] = time P(i, p) (or W(p) as (p, p)) is readable on GPU g
    link_free = {}                          # (src, dst) -> time the link is free again
    free = [slots] * N
    fin = {}                                # (i, j) -> completion of POTRF / TRSM
    started_final = set()
    front = 0
    events = []                             # (time, seq, kind, payload)
    seq = [0]
    now = 0.0
    tasks = []                              # (start, end, gpu, type, i, j, p0, np)

    def push(t, kind, payload):
        seq[0] += 1
        heapq.heappush(events, (t, seq[0], kind, payload))

    def panels_ready(g, i, j, v):
        n = 0
        while v + n < j and n < np_max:
            p = v + n
            a, b = avail[g].get((i, p)), avail[g].get((j, p))
            if a is not None and a <= now and b is not None and b <= now:
                n += 1
            else:
                break
        return n

    def candidates(g):
        out = []
        for j in range(g, T, N):
            for i in range(j, T):
                t = (i, j)
                if t in busy or t in started_final:
                    continue
                if ver[t] < j:
                    n = panels_ready(g, i, j, ver[t])
                    if n > 0:
                        urgent = j <= front + urgent_cols
                        cls = 1 if (i == j and urgent) else 2 if urgent else 3
                        out.append((cls, j, i, UPD, n))
                else:
                    if i == j:
                        out.append((0, j, i, POTRF, 0))
                    else:
                        w = avail[g].get((j, j))
                        if w is not None and w <= now:
                            out.append((1 if i == j + 1 else 2, j, i, TRSM, 0))
        out.sort()
        return out

    def schedule(g):
        for cls, j, i, kind, n in candidates(g):
            if free[g] <= 0:
                break
            if kind == POTRF:
                parts, dur = 1, t_potrf
            elif kind == TRSM:
                parts, dur = (16, t_trsm32) if i == j + 1 else (2, t_trsm64)
            else:
                if cls == 1:
                    n = min(n, 3); parts, dur = 10, t_upd32[0] + t_upd32[1] * (n - 1)
                elif cls == 2:
                    n = min(n, 2); parts, dur = 2, t_upd64[0] + t_upd64[1] * n
                else:
                    parts, dur = 1, t_upd128[0] + t_upd128[1] * n
            if parts > free[g]:
                break
            free[g] -= parts
            busy.add((i, j))
            if kind != UPD:
                started_final.add((i, j))
            p0 = ver[(i, j)]
            tasks.append((now, now + dur, g, kind, i, j, p0, n))
            push(now + dur, 0, (g, parts))
            push(now + dur, 1, (g, kind, i, j, n))

    for g in range(N):
        schedule(g)
    while events:
        now = events[0][0]
        touched = set()
        while events and events[0][0] <= now + 1e-9:
            _, _, kind, pl = heapq.heappop(events)
            if kind == 0:
                free[pl[0]] += pl[1]; touched.add(pl[0])
            elif kind == 1:
                g, k, i, j, n = pl
                busy.discard((i, j))
                if k == UPD:
                    ver[(i, j)] += n
                    push(now + hand, 3, g)
                else:
                    fin[(i, j)] = now
                    if k == POTRF:
                        while front < T and (front, front) in fin:
                            front += 1
                    # readable on the owner after the local hand-off; forwarded to the others (the inverse diagonal factor stays at home:
                    # the TRSMs of a column run where its POTRF ran)
                    push(now + hand, 2, (g, i, j))
                    if k == TRSM:
                        for d in range(N):
                            if d == g:
                                continue
                            key = (g, d)
                            start = max(now, link_free.get(key, 0.0))
                            link_free[key] = start + tile_us
                            push(start + tile_us + hop, 2, (d, i, j))
            elif kind == 2:
                g, i, j = pl
                avail[g][(i, j)] = now; touched.add(g)
            else:
                touched.add(pl)
        for g in (touched if touched else range(N)):
            schedule(g)
    assert len(fin) == T * (T + 1) // 2, "the model got stuck"
    return tasks, now


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, nargs="*", default=[71, 141])
    ap.add_argument("--N", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--link", type=float, default=100.0, help="GB/s per xGMI link (one direction)")
    ap.add_argument("--hop", type=float, default=5.0, help="latency of one forwarded tile, microseconds")
    a = ap.parse_args()
    for T in a.T:
        base = None
        for N in a.N:
            tasks, ms = simulate(T, N, link_gbs=a.link, hop=a.hop)
            base = base or ms
            busy = sum((e - s) * {POTRF: 1, TRSM: 2, UPD: 1}[k] for s, e, g, k, *_ in tasks)
            print(f"T = {T:3d} tile columns, N = {N}: simulated makespan {ms / 1e3:6.3f} ms  ({base / ms:4.2f} x the model's N = 1), "
                  f"{len(tasks)} tasks, busiest link {max([0.0] + [0.0]):.0f}" if False else
                  f"T = {T:3d} tile columns, N = {N}: simulated makespan {ms / 1e3:6.3f} ms  ({base / ms:4.2f} x the model's N = 1), {len(tasks)} tasks, "
                  f"slot-time busy {busy / (ms * N * 496):4.2f}")


if __name__ == "__main__":
    main()
