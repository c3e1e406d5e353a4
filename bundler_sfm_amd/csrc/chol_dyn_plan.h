// chol_dyn_plan.h -- host-side plan of the DYNAMIC tile-dataflow Cholesky (round 6).  Host only, no HIP.
//
// Same factorisation as chol_flow_sched.h (sba_Axb_Chol = dpotrf("U") + dpotrs, lib/sba-1.5/sba_lapack.c:374-485, called from
// lib/sba-1.5/sba_levmar.c:1368), same tile roles, but the BULK of it is no longer served in a ticket order simulated on the host.
// Round 5's trace showed 22 % of the bulk workgroups' slot-time spent holding a ticket whose dependencies were not ready, because
// the simulated durations do not see the contention of the real launch.  Now:
//
//   * every half tile (64 rows x 128 columns of a tile (i, j), or the one right-hand-side row of column j) owns ONE 32-bit state word
//         bits 0-9  ver   number of panels applied so far = next panel to apply
//         bits 10-19 lim  panels [first, lim) are the bulk's to apply (lim = j, or j - 1 where the chain applies the last one itself)
//         bit 28    the bulk also finalises the half (TRSM64 / FTRSM) once ver == j and the diagonal tile's inverse exists
//         bit 30    BUSY  claimed by a workgroup
//         bit 31    FINAL the panel tile (or y_j) is written
//     and every tile row i a pair of monotone counters rowdone[i][h] = "P_ic is final for every c < rowdone[i][h]" (rows of half h);
//   * a free bulk workgroup SCANS the state words of the lowest unfinished columns (one wave per column, one lane per tile row), picks
//     a half tile whose next panels exist -- min(rowdone[i][h], rowdone[j][0], rowdone[j][1], lim) > ver -- claims it with an atomic OR of the
//     BUSY bit, applies ALL panels that are ready (up to np_max) in one pass over the tile, and releases it with a store of the new word.
//     Two halves of a tile with the same state are claimed together (one 128 x 128 pass).  Nothing is handed over through queues: the
//     state words ARE the ready set, lowest column first is the priority, and a claim that finds the work gone is given back;
//   * the CHAIN stays a static ticket list served by workgroups that have a CU to themselves and poll their dependencies directly (a
//     hand-off costs one load round trip): POTRF(k); the sixteen 32 x 32 blocks of TRSM(k+1, k); the ten blocks of the LAST update of
//     the next diagonal tile; and -- new -- the second-order chain TRSM64(k+2, k) and UPD64(k+2, k+1; panel k), the two tasks that decide
//     when tile (k+2, k+1) is ready for the chain's next step.  Which panel of which tile is applied by which role is fixed here, on
//     the host, and every role applies the panels of a tile in ascending order through the same accumulation chains, whatever the batching:
//     the factor is bit-identical from run to run and for any number of workgroups (tests/test_chol_gpu.py).
//
// This file builds the static part (chain tasks in the FlowTask format of chol_flow_sched.h, with "masked" waits on state words) and the
// initial image of the state words; tests/test_chol_dyn_plan.py replays the protocol on the CPU with random interleavings.
#pragma once
#include "chol_flow_sched.h"

namespace bsfm {

constexpr uint32_t DYN_VER_MASK = 0x3ffu;
constexpr uint32_t DYN_LIM_SHIFT = 10;
constexpr uint32_t DYN_ELIG = 1u << 28;      // the bulk finalises this half (TRSM64 of rows i >= j + 3, FTRSM of the right-hand side)
constexpr uint32_t DYN_BUSY = 1u << 30;
constexpr uint32_t DYN_FINAL = 1u << 31;
constexpr uint32_t DYN_WAIT_MASKED = 1u << 31;      // FlowWait.thr: compare (word & DYN_VER_MASK) >= (thr & DYN_VER_MASK)
constexpr int DYN_MAX_TILES = 1023;
// FlowTask.pad of a chain task (bit 0 = "chain queue" as before):
constexpr uint16_t DYN_PAD_CHAIN = 1;
constexpr uint16_t DYN_PAD_PUBROW = 2;       // once its waits have passed the task publishes rowdone[i][0..1] = i (the row's last panel tile, written by the 16 TRSM32 parts)

struct DynPlan {
    int T = 0;
    std::vector<int> last, frow;             // envelope (closed under fill); first column of row i's panel tiles
    std::vector<FlowTask> chain, potrf;      // static queues, ticket order
    // layout of the launch's word array behind the 8 control words (offsets relative to flags = sync + 8)
    uint32_t ofs_c32 = 0, ofs_c10 = 0, ofs_wd = 0, ofs_rd = 0, ofs_tw = 0, nwords = 0;
    std::vector<uint32_t> init;              // initial image of [0, nwords)
    double upd_tiles = 0.0, trsm_tiles = 0.0;      // tile products the launch performs (flop accounting: x 2 * 128^3)
    long long n_halves = 0;                  // half tiles the bulk has work on
};

inline uint32_t dyn_tw(const DynPlan& p, int i, int j, int h) { return p.ofs_tw + 2u * ((uint32_t)j * (uint32_t)(p.T + 1) + (uint32_t)i) + (uint32_t)h; }
inline uint32_t dyn_rd(const DynPlan& p, int i, int h) { return p.ofs_rd + 2u * (uint32_t)i + (uint32_t)h; }

// last_in: envelope (last tile row of each column; empty = dense).  Returns 0, -1 on a size the word format cannot hold.
inline int dyn_build_plan(int T, const std::vector<int>& last_in, DynPlan& out)
{
    out = DynPlan();
    if (T <= 0 || T > DYN_MAX_TILES) return -1;
    out.T = T;
    std::vector<int> last((size_t)T);
    for (int k = 0; k < T; ++k) last[k] = (int)last_in.size() >= T ? std::min(T - 1, std::max(k, last_in[k])) : T - 1;
    for (int p = 0; p < T; ++p)
        for (int j = p + 1; j <= last[p]; ++j) last[j] = std::max(last[j], last[p]);      // closure under fill (as flow_build_schedule)
    out.last = last;
    const int R = T + 1;
    std::vector<int> frow((size_t)R, 0);
    for (int i = 0; i < T; ++i) { int f = i; for (int c = 0; c < i; ++c) if (last[c] >= i) { f = c; break; } frow[i] = f; }
    frow[T] = 0;
    out.frow = frow;
    auto exists = [&](int i, int j) { return j < T && i >= j && (i == T || i <= last[j]); };
    auto first = [&](int i, int j) { return std::min(j, std::max(frow[i], frow[j])); };

    out.ofs_c32 = 0; out.ofs_c10 = (uint32_t)T; out.ofs_wd = 2u * T; out.ofs_rd = 3u * T; out.ofs_rd += out.ofs_rd & 1u;      // pairs are 8-byte aligned (flags = sync + 8 is)
    out.ofs_tw = out.ofs_rd + 2u * R;
    out.nwords = out.ofs_tw + 2u * (uint32_t)T * (uint32_t)R;
    out.init.assign(out.nwords, 0u);
    for (int i = 0; i < R; ++i) { out.init[dyn_rd(out, i, 0)] = (uint32_t)frow[i]; out.init[dyn_rd(out, i, 1)] = i == T ? DYN_VER_MASK : (uint32_t)frow[i]; }
    for (int j = 0; j < T; ++j)
        for (int i = j; i < R; ++i) {
            if (!exists(i, j)) { out.init[dyn_tw(out, i, j, 0)] = out.init[dyn_tw(out, i, j, 1)] = DYN_FINAL | ((uint32_t)j << DYN_LIM_SHIFT) | (uint32_t)j; continue; }
            const int f = first(i, j);
            // the chain applies panel j - 1 itself to the diagonal tile (UPD32) and to tile (j + 1, j) (the static UPD64 behind TRSM64(j + 1, j - 1))
            const bool chain_last = i < T && i <= j + 1 && f < j;
            const int lim = chain_last ? j - 1 : j;
            const bool elig = i == T || i >= j + 3;
            const uint32_t w = (uint32_t)f | ((uint32_t)lim << DYN_LIM_SHIFT) | (elig ? DYN_ELIG : 0u);
            out.init[dyn_tw(out, i, j, 0)] = w;
            out.init[dyn_tw(out, i, j, 1)] = i == T ? (DYN_FINAL | w) : w;      // the right-hand side is one "half"
            if (lim > f || elig) out.n_halves += i == T ? 1 : 2;
            if (i < T) {
                out.upd_tiles += (double)(lim - f);                             // bulk passes compute the whole tile, also on the diagonal
                if (chain_last) out.upd_tiles += i == j ? 10.0 / 16.0 : 1.0;
                if (i > j) out.trsm_tiles += 1.0;
            }
        }
    auto emit = [&](std::vector<FlowTask>& q, uint8_t type, int i, int j, int p0, int np, int part, uint32_t sig, uint16_t pad, std::initializer_list<FlowWait> ws) {
        FlowTask k{};
        k.type = type; k.np = (uint8_t)np; k.part = (uint8_t)part; k.nwait = (uint8_t)ws.size();
        k.i = (uint16_t)i; k.j = (uint16_t)j; k.p0 = (uint16_t)p0; k.pad = pad; k.sig = sig;
        int q2 = 0; for (const FlowWait& w : ws) k.w[q2++] = w;
        q.push_back(k);
    };
    auto masked = [](uint32_t v) { return DYN_WAIT_MASKED | v; };
    for (int k = 0; k < T; ++k) {
        const bool has_upd32 = k > 0 && exists(k, k - 1);          // panel k - 1 applies to the diagonal tile
        if (has_upd32) emit(out.potrf, FT_POTRF, k, k, 0, 0, 0, out.ofs_wd + k, DYN_PAD_CHAIN, { FlowWait{ out.ofs_c10 + (uint32_t)k, 10u } });
        else emit(out.potrf, FT_POTRF, k, k, 0, 0, 0, out.ofs_wd + k, DYN_PAD_CHAIN, {});
        if (k + 1 >= T || !exists(k + 1, k)) continue;
        const bool s1 = k + 2 < T && exists(k + 2, k);
        // the second-order chain first: it is the longer of the two paths to the next column (19 + 17 us against 4 + 5)
        if (s1)
            for (int h = 0; h < 2; ++h)
                emit(out.chain, FT_TRSM64, k + 2, k, 0, 0, h, dyn_tw(out, k + 2, k, h), DYN_PAD_CHAIN,
                     { FlowWait{ out.ofs_wd + (uint32_t)k, 1u }, FlowWait{ dyn_tw(out, k + 2, k, h), masked((uint32_t)k) } });
        for (int part = 0; part < 16; ++part)
            emit(out.chain, FT_TRSM32, k + 1, k, 0, 0, part, out.ofs_c32 + k, DYN_PAD_CHAIN,
                 { FlowWait{ out.ofs_wd + (uint32_t)k, 1u }, FlowWait{ dyn_tw(out, k + 1, k, 0), masked((uint32_t)k) }, FlowWait{ dyn_tw(out, k + 1, k, 1), masked((uint32_t)k) } });
        for (int part = 0; part < 10; ++part)
            emit(out.chain, FT_UPD32, k + 1, k + 1, k, 1, part, out.ofs_c10 + k + 1, (uint16_t)(DYN_PAD_CHAIN | (part == 0 ? DYN_PAD_PUBROW : 0)),
                 { FlowWait{ out.ofs_c32 + (uint32_t)k, 16u }, FlowWait{ dyn_tw(out, k + 1, k + 1, 0), masked((uint32_t)k) }, FlowWait{ dyn_tw(out, k + 1, k + 1, 1), masked((uint32_t)k) } });
        if (s1)
            for (int h = 0; h < 2; ++h)
                emit(out.chain, FT_UPD64, k + 2, k + 1, k, 1, h, dyn_tw(out, k + 2, k + 1, h), DYN_PAD_CHAIN,
                     { FlowWait{ out.ofs_c32 + (uint32_t)k, 16u }, FlowWait{ dyn_rd(out, k + 2, h), masked((uint32_t)(k + 1)) }, FlowWait{ dyn_tw(out, k + 2, k + 1, h), masked((uint32_t)k) } });
    }
    return 0;
}

}  // namespace bsfm
