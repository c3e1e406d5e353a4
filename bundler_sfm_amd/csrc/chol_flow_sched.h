// chol_flow_sched.h -- static schedule of the tile-dataflow Cholesky (round 4).  Host only, no HIP.
//
// The reduced camera solve replaces sba_Axb_Chol = dpotrf("U") + dpotrs (lib/sba-1.5/sba_lapack.c:374-485).  Rounds 1-3 ran
// the tiled factorisation as ~350 launches on three streams tied together by ~280 stream events; what bounded it was not the
// matrix cores but those events (7-18 us each) and the launch-wide barriers ("Dense Cholesky").  Round 4 runs it as a
// DATAFLOW: every tile operation is a task,
//
//      POTRF(k)              L_kk = chol(S_kk), W_k = inv(L_kk)                              (one workgroup, the serial chain)
//      TRSM(i,k)             P_ik = S_ik W_k^T                                               (i > k; row T = the right-hand side: y_k)
//      UPD(i,j,p0,np)        S_ij -= sum_{p = p0}^{p0+np-1} P_ip P_jp^T                      (i >= j > p; np panels in ONE pass over the tile)
//
// tasks are handed to resident workgroups in a fixed order by an atomic ticket, and each task waits on monotonic per-tile
// counters for exactly the tasks it depends on (flag >= threshold), then signals its own tile's counter.  Because every task's
// dependencies come EARLIER in the order, it is a topological order of the task graph and the scheme cannot deadlock however many
// workgroups are resident.  The order is served as TWO queues, each a subsequence of it: the chain (POTRF, first panel tile, next
// diagonal tile) by a few workgroups that have a CU to themselves, everything else by the rest -- the earliest unfinished task of
// the whole order is always at the head of its queue with all of its dependencies done, so the argument carries over.
//
// This file builds that order: an event-driven simulation of greedy list scheduling on `slots` workgroup slots with estimated task
// durations.  Priorities: the chain (POTRF, the first panel tile, the next diagonal tile) first; then the panel rows and the column
// the chain needs next, as 64-row halves so that twice as many workgroups share them; then the bulk, earliest column first,
// each visit applying ALL panels that are ready for the tile (up to np_max) -- a tile the bulk has fallen behind on is caught up in one
// pass over C with a long accumulation (the 128-step update is bound by its C traffic), which the old
// launch-per-panel schedule could never do.  The simulated start order is the ticket order; at run time the spin-waits absorb whatever
// the estimates got wrong.  The result depends only on (nblk, envelope, parameters): the summation order of every tile is fixed, so
// the factorisation stays bit-identical from run to run.
//
// The right-hand side rides along as tile row T (one row instead of 128): TRSM(T,k) is y_k = W_k E_k and UPD(T,j,..) is
// E_j -= sum_p P_jp y_p, with the same dependency rules -- the forward substitution needs no code of its own in the scheduler.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
#include <queue>
#include <algorithm>
#include <functional>

namespace bsfm {

enum FlowTaskType : uint8_t {
    FT_POTRF = 0,     // diagonal tile k = i = j
    FT_TRSM32 = 1,    // first panel tile (i = k + 1), 16 parts of 32 x 32
    FT_TRSM64 = 2,    // panel tile, 2 parts of 64 rows
    FT_UPD32 = 3,     // diagonal tile on the chain, 10 parts (lower-triangle 32 x 32 blocks)
    FT_UPD64 = 4,     // urgent column, 2 parts of 64 rows
    FT_UPD128 = 5,    // bulk, one part
    FT_FTRSM = 6,     // y_k = W_k E_k            (row T)
    FT_FUPD = 7,      // E_j -= sum_p P_jp y_p    (row T)
    FT_NTYPES = 8
};

struct FlowWait { uint32_t idx, thr; };

struct FlowTask {          // 40 bytes, read with scalar loads by the device
    uint8_t type, np, part, nwait;
    uint16_t i, j, p0, pad;      // pad: queue (1 = chain, 0 = bulk)
    uint32_t sig;          // counter this task increments when it is done
    FlowWait w[3];
};
static_assert(sizeof(FlowTask) == 40, "FlowTask layout is shared with the device");

struct FlowParams {
    int slots = 512;          // resident workgroups assumed by the simulation (2 per CU x 256)
    int np_max = 8;           // panels per bulk visit (round 6: 8 with lazy_cols = 2, 6.33 -> 6.23 ms at 71 tile columns; 4 / 0 before)
    int np_max_rhs = 8;       // ... for the right-hand-side row
    int np_max_half = 2;      // ... for the 64-row halves of the urgent column (a tile that reaches the urgent column with a backlog catches up in halves)
    // estimated durations, microseconds: medians of the per-task trace of the n = 9 000 solve on MI355X (profiles/r04_flow_task_durations.txt)
    double t_potrf = 42.0, t_trsm32 = 4.5, t_trsm64 = 17.0, t_upd32 = 5.0, t_upd32_per = 2.5;
    double t_upd64_0 = 7.0, t_upd64_per = 12.5, t_upd128_0 = 12.0, t_upd128_per = 27.0;
    double t_ftrsm = 10.0, t_fupd_0 = 5.0, t_fupd_per = 8.0;
    double t_hand = 0.5;      // completion -> visible to a dependent (measured 0.3-0.5)
    int urgent_cols = 1;      // columns up to (chain front + urgent_cols) are served in halves / blocks
    int adaptive_halves = 128; // > 0: the column the chain needs next is served in 64-row halves only while the bulk has no backlog (fewer than this many
                              // ready tasks beyond the free slots): a half runs at 50 TFLOP/s-equivalent against 53-73 for a whole tile, and latency only
                              // matters when the chain is the bound.  6.47 -> 6.40 ms at 71 tile columns; 0 = always halves
    int lookahead = 0;        // > 0: the LOOKAHEAD TRIANGLE (round 5) -- tiles (i, j) with front < j <= i <= front + lookahead are kept current: every panel is
                              // applied to them as soon as it exists, in 64-row halves, whatever the backlog of the bulk.  Those are the tiles the next
                              // `lookahead` columns of the chain read (row i of the factor up to column i); kept current, the chain runs ahead of the
                              // bulk sweeps instead of in step with them (scripts/r5/critical_path.py: at 71 tile columns POTRF(38) waited 145 us for
                              // the last two visits of tile (38, 37), whole-tile visits of four panels that had been queued behind the bulk)
    double laxity = 2.0;      // > 0 (round 6; default 2.0, 0 = the column order of rounds 4-5): LEAST LAXITY instead of "lowest column first" for the bulk.  A tile is a SEQUENTIAL job -- one workgroup at a time,
                              // ~30 us per panel -- so a tile that is B panels behind must be started B * 30 us before the chain reaches its column, not when
                              // every lower column has run dry: the traced launch had tile (50, 49) 27 panels behind with the front at column 36, and POTRF(50)
                              // then waited 160 us for five serial 8-panel visits of it (profiles/r06_chain_critical_path.txt).  The bulk's priority key becomes
                              // column - laxity * (panels the tile is behind): laxity = columns of head start per panel of backlog
    int band_rows = 0;        // > 0 (round 6): tiles within this many tile rows of the diagonal -- (j, j), (j + 1, j), ... : what the chain itself reads when it reaches column
                              // j -- are kept current in EVERY column (priority class of the urgent column, ordered by column), not only next to the front
    int lazy_cols = 2;        // > 0: a bulk tile further than this many columns ahead of the chain is only visited once TWO panels are ready
                              // for it (or its last one): a one-panel visit moves 393 KB for 4.2 Mflop and is HBM-bound
};

struct FlowSchedule {
    int T = 0;                           // tile columns; row T is the right-hand side
    std::vector<int> last;               // last[k]: last tile row of column k's envelope (closed under fill)
    std::vector<FlowTask> tasks;         // ticket order
    std::vector<int> stage_start;        // ticket of POTRF(k)
    int nflags = 0;                      // (T + 1) * T counters: tile (i, j) -> i * T + j
    double sim_us = 0.0;                 // simulated makespan
    double upd_tiles = 0.0;              // tile products in UPD128/UPD64/UPD32 tasks (flop accounting: x 2 * 128^3)
    double trsm_tiles = 0.0;
    long long count[FT_NTYPES] = { 0 };
};

inline uint32_t flow_flag(int T, int i, int j) { return (uint32_t)(i * T + j); }

// Builds the schedule.  last_in: envelope (empty = dense).  Returns 0, or -1 if the simulation got stuck (a bug).
inline int flow_build_schedule(int T, const std::vector<int>& last_in, const FlowParams& prm, FlowSchedule& out)
{
    out = FlowSchedule();
    out.T = T;
    out.nflags = (T + 1) * T;
    std::vector<int> last((size_t)T);
    for (int k = 0; k < T; ++k) last[k] = (int)last_in.size() >= T ? std::min(T - 1, std::max(k, last_in[k])) : T - 1;
    // closure under fill: a panel that reaches row r also reaches every column up to r, whose own panels then reach at least as far
    for (int p = 0; p < T; ++p)
        for (int j = p + 1; j <= last[p]; ++j) last[j] = std::max(last[j], last[p]);
    out.last = last;
    const int R = T + 1;      // rows incl. the right-hand side
    auto reach = [&](int p, int r) { return r == T || last[p] >= r; };
    auto exists = [&](int i, int j) { return j < T && i >= j && (i == T || i <= last[j]); };

    struct Tile {
        int first = 0;        // first panel that applies (panels first .. j-1 apply: contiguous, see closure)
        int ver = 0;          // next panel to apply
        uint32_t cum = 0;     // signals issued so far on this tile's counter
        bool busy = false, fin = false, started_final = false;
        bool last_u32 = false; // the last visit was a one-panel UPD32 with panel j - 1 (diagonal tiles)
        double t_ready = 0.0; // time the last modification becomes visible
        double p_ready = -1.0;   // time P (or W for diagonal tiles) becomes visible; < 0: not yet
        uint32_t p_thr = 0;      // counter value that means "P / W ready"
    };
    std::vector<Tile> tiles((size_t)R * T);
    auto tl = [&](int i, int j) -> Tile& { return tiles[(size_t)i * T + j]; };
    for (int j = 0; j < T; ++j)
        for (int i = j; i < R; ++i) {
            if (!exists(i, j)) continue;
            Tile& t = tl(i, j);
            int f = j;
            for (int p = 0; p < j; ++p) if (reach(p, i) && reach(p, j)) { f = p; break; }
            t.first = f; t.ver = f;
        }
    // events
    struct Ev { double t; int kind; int a, b, c; };     // kind 0: slots freed (a = count); 1: tile (a,b) modification visible; 2: P/W of tile (a,b) visible
    auto evcmp = [](const Ev& x, const Ev& y) { return x.t > y.t; };
    std::priority_queue<Ev, std::vector<Ev>, decltype(evcmp)> events(evcmp);
    int free_slots = prm.slots;
    double now = 0.0;
    int front = 0;            // first column whose POTRF has not completed
    std::vector<char> potrf_done((size_t)T, 0);

    // ready queue: (class, column, row) ascending; entries are re-validated when popped
    struct Cand { int cls, col, row, kind; double key; };   // kind 0: POTRF / TRSM (finalise), 1: UPD; key: the column, or the column less the laxity bonus
    auto ccmp = [](const Cand& x, const Cand& y) {
        if (x.cls != y.cls) return x.cls > y.cls;
        if (x.key != y.key) return x.key > y.key;
        if (x.col != y.col) return x.col > y.col;
        return x.row > y.row;
    };
    std::priority_queue<Cand, std::vector<Cand>, decltype(ccmp)> ready(ccmp);

    auto panels_ready = [&](int i, int j, int v) -> int {      // consecutive panels v.. whose operands are visible now
        Tile& t = tl(i, j);
        (void)t;
        const int cap = i == T ? prm.np_max_rhs : prm.np_max;
        int n = 0;
        while (v + n < j && n < cap) {
            const int p = v + n;
            const Tile& a = tl(i, p); const Tile& b = tl(j, p);
            if (a.p_ready >= 0.0 && a.p_ready <= now && b.p_ready >= 0.0 && b.p_ready <= now) ++n; else break;
        }
        return n;
    };
    // panels the tile could apply right now without the per-visit cap (its backlog), for the laxity key
    auto backlog = [&](int i, int j, int v) -> int {
        int n = 0;
        while (v + n < j) {
            const int p = v + n;
            const Tile& a = tl(i, p); const Tile& b = tl(j, p);
            if (a.p_ready >= 0.0 && a.p_ready <= now && b.p_ready >= 0.0 && b.p_ready <= now) ++n; else break;
        }
        return n;
    };
    auto upd_key = [&](int cls, int i, int j, int v) -> double {
        if (prm.laxity <= 0.0 || cls != 3 || i == T) return (double)j;
        return (double)j - prm.laxity * (double)backlog(i, j, v);
    };
    auto in_triangle = [&](int i, int j) -> bool { return prm.lookahead > 0 && i < T && i <= front + prm.lookahead; };      // (j <= i always)
    auto classify_upd = [&](int i, int j) -> int {
        const bool urgent_col = j <= front + prm.urgent_cols;
        if (i == T) return urgent_col ? 2 : 3;
        if (i == j && urgent_col) return 1;
        if (urgent_col || in_triangle(i, j)) return 2;
        if (prm.band_rows > 0 && i - j <= prm.band_rows) return 2;
        return 3;
    };
    auto consider = [&](int i, int j) {                 // push whatever the tile can do now
        if (!exists(i, j)) return;
        Tile& t = tl(i, j);
        if (t.busy || t.fin || t.started_final) return;
        if (t.t_ready > now) return;
        if (t.ver < j) {
            const int nr = panels_ready(i, j, t.ver);
            const int need = (prm.lazy_cols > 0 && i != T && j > front + prm.lazy_cols) ? std::min(2, j - t.ver) : 1;
            if (nr >= need) { const int cls_ = classify_upd(i, j); ready.push({ cls_, j, i, 1, upd_key(cls_, i, j, t.ver) }); }
        } else {
            // final: POTRF (diagonal) or TRSM
            if (i == j) ready.push({ 0, j, i, 0, (double)j });
            else {
                const Tile& d = tl(j, j);
                if (d.p_ready >= 0.0 && d.p_ready <= now) ready.push({ i == j + 1 ? 1 : 2, j, i, 0, (double)j });
            }
        }
    };
    auto emit = [&](uint8_t type, int i, int j, int p0, int np, int part, uint32_t sig, const FlowWait* w, int nw) {
        FlowTask k{};
        k.type = type; k.np = (uint8_t)np; k.part = (uint8_t)part; k.nwait = (uint8_t)nw;
        k.i = (uint16_t)i; k.j = (uint16_t)j; k.p0 = (uint16_t)p0; k.sig = sig;
        k.pad = (type == FT_POTRF || type == FT_TRSM32 || type == FT_UPD32) ? 1 : 0;      // queue: 1 = the chain's own workgroups, 0 = bulk
        for (int q = 0; q < 3; ++q) k.w[q] = q < nw ? w[q] : FlowWait{ 0u, 0u };
        out.tasks.push_back(k);
        out.count[type]++;
    };

    for (int j = 0; j < T; ++j) for (int i = j; i < R; ++i) consider(i, j);
    long long remaining = 0;
    for (int j = 0; j < T; ++j) for (int i = j; i < R; ++i) if (exists(i, j)) ++remaining;   // tiles not yet finalised
    out.stage_start.assign((size_t)T, 0);

    auto schedule_now = [&]() {
        while (!ready.empty() && free_slots > 0) {
            const Cand c = ready.top();
            const int i = c.row, j = c.col;
            Tile& t = tl(i, j);
            // re-validate
            bool ok = !t.busy && !t.fin && !t.started_final && t.t_ready <= now;
            int n = 0;
            if (ok && c.kind == 1) {
                ok = t.ver < j;
                if (ok) {
                    n = panels_ready(i, j, t.ver);
                    const int need = (prm.lazy_cols > 0 && i != T && j > front + prm.lazy_cols) ? std::min(2, j - t.ver) : 1;
                    ok = n >= need;
                }
            }
            if (ok && c.kind == 0) {
                ok = t.ver >= j;
                if (ok && i != j) { const Tile& d = tl(j, j); ok = d.p_ready >= 0.0 && d.p_ready <= now; }
            }
            if (ok && c.kind == 1 && classify_upd(i, j) != c.cls) { ready.pop(); const int cls_ = classify_upd(i, j); ready.push({ cls_, j, i, 1, upd_key(cls_, i, j, t.ver) }); continue; }
            if (!ok) { ready.pop(); continue; }
            // parts and duration
            uint8_t type; int parts; double dur;
            if (c.kind == 0) {
                if (i == j) { type = FT_POTRF; parts = 1; dur = prm.t_potrf; }
                else if (i == T) { type = FT_FTRSM; parts = 1; dur = prm.t_ftrsm; }
                else if (i == j + 1) { type = FT_TRSM32; parts = 16; dur = prm.t_trsm32; }
                else { type = FT_TRSM64; parts = 2; dur = prm.t_trsm64; }
            } else {
                const int cls = classify_upd(i, j);
                if (i == T) { type = FT_FUPD; parts = 1; dur = prm.t_fupd_0 + prm.t_fupd_per * n; }
                else if (cls == 1) { n = std::min(n, 3); type = FT_UPD32; parts = 10; dur = prm.t_upd32 + prm.t_upd32_per * (n - 1); }
                else if (cls == 2 && (in_triangle(i, j) || !(prm.adaptive_halves && (long long)ready.size() > (long long)free_slots + prm.adaptive_halves))) {
                    n = std::min(n, prm.np_max_half); type = FT_UPD64; parts = 2; dur = prm.t_upd64_0 + prm.t_upd64_per * n;
                }
                else { type = FT_UPD128; parts = 1; dur = prm.t_upd128_0 + prm.t_upd128_per * n; }
            }
            if (parts > free_slots) break;            // the head of the queue waits for room (it has the highest priority)
            ready.pop();
            free_slots -= parts;
            const uint32_t me = flow_flag(T, i, j);
            FlowWait w[3]; int nw = 0;
            if (t.cum > 0) w[nw++] = { me, t.cum };
            if (c.kind == 0) {
                if (i != j) { const Tile& d = tl(j, j); w[nw++] = { flow_flag(T, j, j), d.p_thr }; }
                if (i == j) out.stage_start[(size_t)j] = (int)out.tasks.size();
                // (np of a POTRF task: 1 = the tile's last update was a one-panel UPD32 with the chain's panel -- the kernel may take the tile from that update's
                //  second copy instead of waiting for its counter; chol_flow.hip.h, FlowArgs::Du)
                for (int part = 0; part < parts; ++part) emit(type, i, j, 0, (type == FT_POTRF && t.last_u32) ? 1 : 0, part, me, w, nw);
                t.started_final = true; t.busy = true;
                t.p_thr = t.cum + (uint32_t)parts;
                t.cum += (uint32_t)parts;
                if (i != j && i != T) out.trsm_tiles += 1.0;
                events.push({ now + dur, 0, parts, 0, 0 });
                events.push({ now + dur + prm.t_hand, 2, i, j, 0 });
            } else {
                const int pl = t.ver + n - 1;
                { const Tile& a = tl(i, pl); w[nw++] = { flow_flag(T, i, pl), a.p_thr }; }
                if (i != j) { const Tile& b = tl(j, pl); w[nw++] = { flow_flag(T, j, pl), b.p_thr }; }
                for (int part = 0; part < parts; ++part) emit(type, i, j, t.ver, n, part, me, w, nw);
                t.last_u32 = type == FT_UPD32 && n == 1 && pl == j - 1;
                t.busy = true;
                t.cum += (uint32_t)parts;
                t.ver += n;
                if (i != T) out.upd_tiles += (double)n * (type == FT_UPD32 ? 10.0 / 16.0 : 1.0);
                events.push({ now + dur, 0, parts, 0, 0 });
                events.push({ now + dur + prm.t_hand, 1, i, j, 0 });
            }
        }
    };

    schedule_now();
    while (!events.empty()) {
        now = events.top().t;
        while (!events.empty() && events.top().t <= now + 1e-9) {
            const Ev e = events.top(); events.pop();
            if (e.kind == 0) { free_slots += e.a; continue; }
            const int i = e.a, j = e.b;
            Tile& t = tl(i, j);
            t.busy = false;
            if (e.kind == 1) {
                t.t_ready = now;
                consider(i, j);
            } else {
                t.fin = true; t.p_ready = now; --remaining;
                if (i == j) {
                    potrf_done[(size_t)j] = 1;
                    while (front < T && potrf_done[(size_t)front]) ++front;
                    for (int r = j + 1; r < R; ++r) consider(r, j);                      // TRSMs of column j
                    for (int c2 = j + 1; c2 <= std::min(T - 1, front + std::max(prm.urgent_cols, prm.lazy_cols)); ++c2)     // urgency may have changed
                        for (int r = c2; r < R; ++r) consider(r, c2);
                    if (prm.lookahead > 0)                                                  // the row that has just entered the triangle
                        for (int r = std::max(j + 1, front + 1); r <= std::min(T - 1, front + prm.lookahead); ++r)
                            for (int c2 = std::max(j + 1, front + prm.urgent_cols + 1); c2 <= r; ++c2) consider(r, c2);
                } else {
                    // P_ij ready: row operand of tiles (i, c), j < c <= i; column operand of tiles (r, i), r >= i
                    for (int c2 = j + 1; c2 <= std::min(i, T - 1); ++c2) consider(i, c2);
                    if (i < T) for (int r = i; r < R; ++r) consider(r, i);
                }
            }
        }
        schedule_now();
    }
    out.sim_us = now;
    if (remaining != 0) {
        fprintf(stderr, "[bsfm] flow scheduler stuck: %lld tiles not finalised (T = %d)\n", remaining, T);
        return -1;
    }
    return 0;
}

// Structural check (also run by tests/test_chol_flow_sched.py through the C ABI): every wait of a task must be satisfiable by
// tasks with EARLIER tickets -- the property that makes ticket order deadlock-free.  Returns the number of violations.
inline long long flow_check_schedule(const FlowSchedule& s)
{
    std::vector<uint32_t> cnt((size_t)s.nflags, 0u);
    long long bad = 0;
    for (const FlowTask& t : s.tasks) {
        for (int q = 0; q < t.nwait; ++q)
            if (t.w[q].idx >= (uint32_t)s.nflags || cnt[t.w[q].idx] < t.w[q].thr) ++bad;
        if (t.sig >= (uint32_t)s.nflags) ++bad; else cnt[t.sig]++;
    }
    return bad;
}

}  // namespace bsfm
