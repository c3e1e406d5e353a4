"""The DYNAMIC tile-dataflow Cholesky (csrc/chol_dyn_plan.h, chol_dyn.hip.h; replaces sba_Axb_Chol = dpotrf + dpotrs,
lib/sba-1.5/sba_lapack.c:374-485): the host-side plan through the C ABI, and a CPU replay of the device protocol -- state words, scan,
claim by atomic OR, release by store, static chain queues with masked waits -- under RANDOM interleavings of the workgroups' steps.
The replay checks what the kernel relies on: no half tile is ever owned twice, every (half tile, panel) is applied exactly once and in
order, no operand is read before it is final, nothing deadlocks, and the result is the Cholesky factor and the forward-substituted
right-hand side.  No GPU needed."""
import numpy as np
import pytest

import bundler_sfm_amd.sfm as B

VER, LIMS, ELIG, BUSY, FINAL, MASKED = 0x3FF, 10, 1 << 28, 1 << 30, 1 << 31, 1 << 31
POTRF, TRSM32, TRSM64, UPD32, UPD64, UPD128, FTRSM, FUPD = range(8)


class Replay:
    def __init__(self, T, last, nb, rng, n_bulk, n_chain, np_max=4, scan_cols=16):
        self.T, self.R, self.nb, self.hb, self.rng = T, T + 1, nb, nb // 2, rng
        self.p = B.chol_dyn_plan(T, last)
        self.w = self.p["init"].astype(np.int64).copy()
        self.lowcol = 0
        self.tickets = {"chain": 0, "potrf": 0}
        self.np_max, self.scan_cols = np_max, scan_cols
        n = T * nb
        A = rng.standard_normal((n, n))
        S = A @ A.T + n * np.eye(n)
        if last is not None:                       # zero outside the tile envelope (closure under fill is the plan's job)
            for j in range(T):
                S[(int(last[j]) + 1) * nb:, j * nb:(j + 1) * nb] = 0.0
            S = np.tril(S); S = S + np.tril(S, -1).T
            S += n * np.eye(n)
        self.S0 = S.copy()
        self.S = S.copy()
        self.E0 = rng.standard_normal(n)
        self.E = self.E0.copy()
        self.P = np.full((n, n), np.nan)           # panel tiles (strictly lower block part), NaN = not written
        self.Ld = [None] * T                       # diagonal factors
        self.W = [None] * T
        self.y = np.full(n, np.nan)
        self.owner = {}                            # (i, j, h) -> worker that holds BUSY
        self.applied = {}                          # (i, j, h) -> panels applied
        self.log = []
        self.workers = [self.bulk_worker(q) for q in range(n_bulk)] + [self.static_worker("chain", q) for q in range(n_chain)] + [self.static_worker("potrf", 0)]
        self.alive = [True] * len(self.workers)
        self.progress = 0

    # ---- addressing
    def tw(self, i, j, h): return self.p["ofs_tw"] + 2 * (j * self.R + i) + h
    def rd(self, i, h): return self.p["ofs_rd"] + 2 * i + h
    def rows(self, i, h=None):
        if h is None: return slice(i * self.nb, (i + 1) * self.nb)
        return slice(i * self.nb + h * self.hb, i * self.nb + (h + 1) * self.hb)

    # ---- numerics of the roles (operands must be final: NaN checks)
    def upd(self, i, j, p0, npn, hs):
        for h in hs:
            key = (i, j, h)
            assert self.applied.get(key, self.first(i, j)) == p0, ("panels out of order", key, p0, self.applied.get(key))
            self.applied[key] = p0 + npn
        for p in range(p0, p0 + npn):
            if i == self.T:
                Pj = self.P[self.rows(j), self.rows(p)]; yp = self.y[self.rows(p)]
                assert not np.isnan(Pj).any() and not np.isnan(yp).any(), ("operand not final", i, j, p)
                self.E[self.rows(j)] -= Pj @ yp
            else:
                for h in hs:
                    Pi = self.P[self.rows(i, h), self.rows(p)]; Pj = self.P[self.rows(j), self.rows(p)]
                    assert not np.isnan(Pi).any() and not np.isnan(Pj).any(), ("operand not final", i, j, p, h)
                    self.S[self.rows(i, h), self.rows(j)] -= Pi @ Pj.T

    def first(self, i, j):
        return int(self.p["init"][self.tw(i, j, 0)]) & VER

    def trsm(self, i, j, rs):
        assert self.W[j] is not None, ("inverse factor missing", i, j)
        self.P[rs, self.rows(j)] = self.S[rs, self.rows(j)] @ self.W[j].T

    def run_task(self, type_, i, j, p0, npn, part):
        nb = self.nb
        if type_ == POTRF:
            blk = np.tril(self.S[self.rows(j), self.rows(j)])
            L = np.linalg.cholesky(blk + np.tril(blk, -1).T)
            self.Ld[j] = L; self.W[j] = np.linalg.inv(L)
        elif type_ == TRSM32:
            br, bc = part >> 2, part & 3
            q = nb // 4
            rs = slice(i * nb + br * q, i * nb + (br + 1) * q)
            full = self.S[rs, self.rows(j)] @ self.W[j].T
            cs = slice(j * nb + bc * q, j * nb + (bc + 1) * q)
            self.P[rs, cs] = full[:, bc * q:(bc + 1) * q]
        elif type_ == TRSM64:
            self.trsm(i, j, self.rows(i, part))
        elif type_ == UPD32:
            br = 0 if part < 1 else 1 if part < 3 else 2 if part < 6 else 3
            bc = part - br * (br + 1) // 2
            q = nb // 4
            Pj = self.P[self.rows(i), self.rows(p0)]
            assert not np.isnan(Pj).any()
            rs = slice(i * nb + br * q, i * nb + (br + 1) * q); cs = slice(i * nb + bc * q, i * nb + (bc + 1) * q)
            self.S[rs, cs] -= Pj[br * q:(br + 1) * q] @ Pj[bc * q:(bc + 1) * q].T
        elif type_ == UPD64:
            self.upd(i, j, p0, npn, [part])
        elif type_ == UPD128:
            self.upd(i, j, p0, npn, [0, 1])
        elif type_ == FTRSM:
            self.y[self.rows(j)] = self.W[j] @ self.E[self.rows(j)]
        elif type_ == FUPD:
            self.upd(i, j, p0, npn, [0])
        self.log.append((type_, i, j, p0, npn, part))
        self.progress += 1

    # ---- the static queues: ticket, poll, (publish), work, signal -- one generator step per memory round trip
    def static_worker(self, which, wid):
        tasks = self.p[which]
        while True:
            tk = self.tickets[which]; self.tickets[which] += 1
            yield
            if tk >= len(tasks): return
            t = tasks[tk]
            while True:
                ok = True
                for q in range(int(t["nwait"])):
                    idx, thr = int(t["w"][q][0]), int(t["w"][q][1])
                    v = int(self.w[idx])
                    ok = ok and ((v & VER) >= (thr & VER) if thr & MASKED else v >= thr)
                if ok: break
                yield "wait"
            ty, i, j, p0, npn, part, sig = (int(t[k]) for k in ("type", "i", "j", "p0", "np", "part", "sig"))
            if int(t["pad"]) & 2:
                self.w[self.rd(i, 0)] = i; self.w[self.rd(i, 1)] = i
            yield
            self.run_task(ty, i, j, p0, npn, part)
            yield
            if ty == TRSM64:
                self.w[self.rd(i, part)] = j + 1
                self.w[sig] = FINAL | (j << LIMS) | j
            elif ty == UPD64:
                self.w[sig] = (p0 + npn) | (p0 << LIMS)
            else:
                self.w[sig] += 1
            yield

    # ---- a bulk workgroup: scan (a snapshot), claim (atomic), work, release
    def bulk_worker(self, wid):
        T, R = self.T, self.R
        idle = 0
        while True:
            lc = self.lowcol
            if lc >= T: return
            cands = []
            for j in range(lc, min(T, lc + self.scan_cols)):
                rdj = min(int(self.w[self.rd(j, 0)]) & VER, int(self.w[self.rd(j, 1)]) & VER)
                wdj = int(self.w[self.p["ofs_wd"] + j])
                col, all_done = [], True
                for i in range(j, R):
                    ws = [int(self.w[self.tw(i, j, h)]) for h in (0, 1)]
                    kinds, rdm, done = [], [], True
                    for h in (0, 1):
                        ver, lim = ws[h] & VER, (ws[h] >> LIMS) & VER
                        fin, busy, elig = bool(ws[h] & FINAL), bool(ws[h] & BUSY), bool(ws[h] & ELIG)
                        m = min(int(self.w[self.rd(i, h)]) & VER, rdj); rdm.append(m)
                        upd = (not fin) and (not busy) and ver < lim and min(m, lim) > ver
                        fz = (not fin) and (not busy) and elig and ver == j and wdj != 0
                        kinds.append(1 if upd else 2 if fz else 0)
                        done = done and (fin or ((not elig) and ver >= lim))
                    all_done = all_done and done
                    if kinds[0] or kinds[1]: col.append((i, kinds, ws, rdm, wdj))
                if col:
                    pick = col[0] if idle == 0 else col[self.rng.integers(len(col))]
                    cands.append((j,) + pick)
                elif all_done and j == lc:
                    self.lowcol = max(self.lowcol, j + 1)
            yield                                      # (the scan's loads have returned; everything below works on the snapshot)
            if not cands:
                idle += 1
                yield "wait"
                continue
            j, i, kinds, ws, rdm, wdj = cands[0] if idle == 0 else cands[self.rng.integers(len(cands))]
            urgent = idle > 0 and j <= lc + 1
            if kinds[0] == 1 and kinds[1] == 1 and (ws[0] & VER) == (ws[1] & VER) and not urgent: hm = 3
            elif kinds[0] and kinds[1]: hm = 1 + int(self.rng.integers(2))
            else: hm = 1 if kinds[0] else 2
            # claim: atomic OR, returns the old words
            old = [BUSY, BUSY]
            for h in (0, 1):
                if hm & (1 << h):
                    old[h] = int(self.w[self.tw(i, j, h)])
                    self.w[self.tw(i, j, h)] |= BUSY
            own, act, av = [], [], []
            for h in (0, 1):
                o = not (old[h] & BUSY); own.append(o)
                ver, lim = old[h] & VER, (old[h] >> LIMS) & VER
                cap = min(rdm[h], lim); a = cap - ver if cap > ver else 0; av.append(a)
                act.append(0 if (not o or old[h] & FINAL) else 1 if a > 0 else 2 if (old[h] & ELIG and ver == j and wdj) else 0)
            keep, ty, npn, p0, part = 0, None, 0, 0, 0
            cap_np = 8 if i == T else self.np_max
            if act[0] == 1 and act[1] == 1 and (old[0] & VER) == (old[1] & VER):
                keep, ty, p0, npn = 3, UPD128, old[0] & VER, min(av[0], av[1], cap_np)
            else:
                h = 0 if act[0] else 1 if act[1] else -1
                if h >= 0:
                    keep, part, p0 = 1 << h, h, old[h] & VER
                    if act[h] == 1: npn, ty = min(av[h], cap_np), (FUPD if i == T else UPD64)
                    else: npn, ty = 0, (FTRSM if i == T else TRSM64)
            for h in (0, 1):
                if own[h] and not (keep & (1 << h)): self.w[self.tw(i, j, h)] &= ~BUSY
                if keep & (1 << h):
                    assert (i, j, h) not in self.owner, ("owned twice", i, j, h)
                    self.owner[(i, j, h)] = wid
            yield
            if not keep:
                idle += 1
                continue
            self.run_task(ty, i, j, p0, npn, part)
            yield
            if ty in (TRSM64, FTRSM):
                self.w[self.rd(i, part)] = j + 1
                self.w[self.tw(i, j, part)] = FINAL | (j << LIMS) | j
                del self.owner[(i, j, part)]
            else:
                for h in (0, 1):
                    if keep & (1 << h):
                        self.w[self.tw(i, j, h)] = (old[h] & ~(VER | BUSY)) | (p0 + npn)
                        del self.owner[(i, j, h)]
            idle = 0
            yield

    def run(self, max_steps=4_000_000):
        stalled = 0
        for _ in range(max_steps):
            live = [q for q, a in enumerate(self.alive) if a]
            if not live: break
            q = live[self.rng.integers(len(live))]
            before = self.progress
            try:
                r = next(self.workers[q])
            except StopIteration:
                self.alive[q] = False
                continue
            stalled = stalled + 1 if (r == "wait" and self.progress == before) else 0
            assert stalled < 200 * len(self.workers) + 20000, "no progress: the protocol is stuck"
        assert not any(self.alive), "workers still alive"

    def check(self):
        T, nb = self.T, self.nb
        n = T * nb
        L = np.zeros((n, n))
        for j in range(T):
            L[self.rows(j), self.rows(j)] = self.Ld[j]
            if j + 1 < T:
                blk = self.P[(j + 1) * nb:, self.rows(j)]
                L[(j + 1) * nb:, self.rows(j)] = np.where(np.isnan(blk), 0.0, blk)
        Lref = np.linalg.cholesky(self.S0)
        assert np.abs(L - Lref).max() <= 1e-9 * np.abs(Lref).max()
        yref = np.linalg.solve(Lref, self.E0)
        assert np.abs(self.y - yref).max() <= 1e-9 * np.abs(yref).max()


@pytest.mark.parametrize("T,n_bulk,seed", [(1, 2, 0), (2, 3, 1), (3, 2, 2), (5, 1, 3), (6, 7, 4), (9, 24, 5), (12, 5, 6), (12, 40, 7)])
def test_random_interleavings_give_the_cholesky_factor(T, n_bulk, seed):
    rng = np.random.default_rng(seed)
    r = Replay(T, None, 8, rng, n_bulk=n_bulk, n_chain=3)
    r.run(); r.check()
    # every (tile, panel) exactly once: the dense count of tile products
    upd = sum(t[4] * (2 if t[0] == UPD128 else 1) for t in r.log if t[0] in (UPD64, UPD128) and t[1] < T)
    assert upd == sum(2 * ((j - 1 if (i <= j + 1 and j > 0) else j)) for j in range(T) for i in range(j, T)) + 2 * sum(1 for k in range(T) if k + 2 < T)


@pytest.mark.parametrize("seed", range(4))
def test_envelopes_and_block_diagonal_systems(seed):
    rng = np.random.default_rng(100 + seed)
    T = 10
    if seed == 0: last = np.minimum(T - 1, np.arange(T) + 2)                      # band
    elif seed == 1: last = np.array([2, 2, 2, 5, 5, 5, 9, 9, 9, 9])                 # independent diagonal blocks
    elif seed == 2: last = np.arange(T)                                            # block diagonal: no panel tiles at all
    else: last = np.minimum(T - 1, np.arange(T) + rng.integers(0, 5, T))
    r = Replay(T, last.astype(np.int32), 8, rng, n_bulk=6, n_chain=4)
    r.run(); r.check()


def test_plan_is_deterministic_and_words_are_aligned():
    a, b = B.chol_dyn_plan(71), B.chol_dyn_plan(71)
    assert (a["init"] == b["init"]).all() and (a["chain"] == b["chain"]).all() and (a["potrf"] == b["potrf"]).all()
    assert a["ofs_rd"] % 2 == 0 and a["ofs_tw"] % 2 == 0                 # pairs are read and claimed as 8-byte words
    assert len(a["potrf"]) == 71 and len(a["chain"]) == 70 * 26 + 69 * 4
    c = B.chol_dyn_plan(7)                                               # an odd count puts rowdone behind an odd number of counters
    assert c["ofs_rd"] % 2 == 0 and c["ofs_tw"] % 2 == 0
