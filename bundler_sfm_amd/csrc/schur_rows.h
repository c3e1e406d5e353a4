// schur_rows.h -- plan of the row-wise Schur kernel (round 5; schur.hip.h: k_schur_rows).  Host only, no HIP.
//
// The Schur complement S_jk -= sum_i Y_ij W_ik^T (lib/sba-1.5/sba_levmar.c:1195-1302) is accumulated from co-visibility triples
// (record of (i,j), record of (i,k)).  Rounds 1-4 cut every block (j,k) into tasks of <= 192 triples, one wave each, and every task
// gathered BOTH records of every triple in 16-byte chunks: 400 bytes per triple, 11.4 GB through L1 per launch at 1 000 cameras /
// 5 M observations, of which the j side (A_ij, C_ij: 208 bytes) is fetched again for every partner camera k of the same point.
// The row kernel turns that around: a workgroup owns (camera j, a SEGMENT of <= L consecutive records of j in camera-major order),
// streams that segment ONCE into LDS -- contiguous, coalesced, no gather -- together with the triple lists of everything it is going to
// do, and its waves then walk the partner blocks (j,k): only the k side (192 bytes) is gathered per triple, the j side is an LDS
// operand shared by all partners, and no wave ever waits for an index load again.
//
// That pays when a (block, segment) VISIT holds enough triples to fill the kernel's passes of 16: a block is DENSE when it has at
// least `dense_min` triples per segment of camera j on average, and a camera's row takes part only when its dense blocks give a
// workgroup at least `wg_min` triples per segment (a workgroup with one short visit would stream its segment for nothing).  Sparse
// blocks (two cameras that share a few percent of their points) stay with the task kernel of rounds 3-4 (k_schur_tasks); both
// write partial sums into one slot space and k_schur_assemble adds a block's slots in slot order, so S stays bit-identical from
// run to run whatever the launch order.
//
// This file builds the plan from what the index construction knows (index_build.hip):
//   stage A (host)    segments per camera, dense flags, the candidate visits (block b, segment s) of the dense blocks
//   stage B (device)  k_row_visits: the sub-range of block b's triple list whose j-side record lies in segment s
//                     (the triples of a block are in point order = ascending record position: two binary searches)
//   stage C (host)    workgroups = the non-empty visits of one (j, s), cut where the LDS budget ends (tri_max padded triples, ROW_PMAX
//                     pieces); their passes of 16 triples are dealt to the NW waves as contiguous runs of equal length (a visit that
//                     straddles two waves becomes two PIECES with a partial sum each); slots in (block, segment, part) order; launch
//                     order (slice of the camera's record list, camera in breadth-first numbering, segment) laid out so that every
//                     XCD gets one contiguous stretch; the copy list from which the device builds the kernel's own triple array
//                     (workgroup after workgroup, wave after wave, every piece padded to whole passes, j side as slab row).
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>

#if defined(__HIPCC__)
#define BSFM_HD __host__ __device__
#else
#define BSFM_HD
#endif

namespace bsfm {

// Stage B, one candidate visit: the triples [lo, lo + cnt) of block [t0, t1) whose j-side record (tri_x: .x of the triple, ascending
// inside a block) lies in [r0, r1).  Shared by the device kernel (index_build.hip: k_row_visits) and the host replay of the tests.
template <typename GetX>
BSFM_HD inline void row_visit_range(int t0, int t1, int r0, int r1, GetX tri_x, int& lo_out, int& cnt_out)
{
    int lo = t0, hi = t1;
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (tri_x(mid) < r0) lo = mid + 1; else hi = mid; }
    const int first = lo;
    hi = t1;
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (tri_x(mid) < r1) lo = mid + 1; else hi = mid; }
    lo_out = first; cnt_out = lo - first;
}

constexpr int ROW_NW = 4;            // waves per workgroup of k_schur_rows
constexpr int ROW_LMAX = 128;        // records per segment at most
constexpr int ROW_PASS = 16;         // triples per pass (= SCM_PASS of schur.hip.h)
constexpr int ROW_PMAX = 32;         // pieces per workgroup at most (LDS copy of their headers)
constexpr int ROW_DEAD = 1 << 16;    // .x of a padding entry of the kernel's triple array (the low 16 bits stay a valid slab row)

struct RowPiece { int npass, diag, out, pad; };              // passes of 16 (padded) triples, diagonal block?, slot of its partial sum
// segment [rec0, rec0 + nrec) of camera-major records; the workgroup's entries of the row triple array are [tri0, tri0 + 16 npass);
// its pieces [piece0, piece0 + npieces); wave w owns passes [wpass[w], wpass[w + 1]) (wpass[NW] = npass) and the pieces from wpiece[w]
struct RowWG { int rec0, nrec, tri0, npass, piece0, npieces, wpass[ROW_NW], wpiece[ROW_NW], pad[2]; };
static_assert(sizeof(RowWG) == 64, "RowWG layout is shared with the device");
struct RowFill { int src, count, dst, rec0; };               // copy triples[src .. src + count) to row entries [dst ..), padded to whole passes

struct RowPlanParams {
    int L = 96;               // records per segment (multiple of 16, <= ROW_LMAX)
    int dense_min = 24;       // a block goes to the row kernel when it has >= dense_min triples per segment of camera j on average
    int wg_min = 0;           // ... and camera j's dense blocks together >= wg_min triples per segment (0: 2.5 L)
    int tri_max = 0;          // padded triples per workgroup at most (LDS; 0: 12 L)
};
inline int row_tri_max(const RowPlanParams& p) { return p.tri_max > 0 ? p.tri_max : 12 * p.L; }

struct RowPlanA {                        // stage A
    std::vector<int> nseg;               // per camera (0 for constrained / unseen cameras)
    std::vector<int> visbase;            // nblk + 1: first candidate visit of block b (a sparse block has none)
    std::vector<unsigned char> dense;    // nblk
    int nvisits = 0, ndense = 0;
};

// blk_j: camera j of the blocks in block order (j <= k, rows consecutive); counts: triples per block; camptr: m + 1
inline void row_plan_stage_a(int m, int mcon, const std::vector<int>& blk_j, const std::vector<int>& counts, const std::vector<int>& camptr,
                             const RowPlanParams& prm, RowPlanA& a)
{
    const int nblk = (int)blk_j.size();
    a.nseg.assign((size_t)m, 0);
    for (int j = mcon; j < m; ++j) a.nseg[j] = (camptr[j + 1] - camptr[j] + prm.L - 1) / prm.L;
    a.visbase.assign((size_t)nblk + 1, 0);
    a.dense.assign((size_t)nblk, 0);
    std::vector<long long> row_dense((size_t)m, 0);
    for (int b = 0; b < nblk; ++b) {
        const int ns = a.nseg[blk_j[b]];
        if (ns > 0 && (long long)counts[b] >= (long long)prm.dense_min * ns) { a.dense[b] = 1; row_dense[blk_j[b]] += counts[b]; }
    }
    const long long wg_min = prm.wg_min > 0 ? prm.wg_min : (5LL * prm.L) / 2;
    a.ndense = 0;
    long long v = 0;
    for (int b = 0; b < nblk; ++b) {
        a.visbase[b] = (int)v;
        const int j = blk_j[b];
        if (a.dense[b] && row_dense[j] < wg_min * a.nseg[j]) a.dense[b] = 0;
        if (a.dense[b]) { v += a.nseg[j]; ++a.ndense; }
    }
    a.visbase[nblk] = (int)v;
    a.nvisits = (int)v;
}

struct RowPlan {
    std::vector<RowWG> wgs;              // launch order (XCD-striped)
    std::vector<RowPiece> pieces;
    std::vector<RowFill> fills;          // one per piece, same order
    std::vector<int> blk_row0;           // nblk + 1: slots of block b are [blk_row0[b], blk_row0[b+1]) (+ slot_base, already in RowPiece::out)
    int nslots = 0;
    long long ntri = 0;                  // entries of the row triple array (padded)
    long long triples = 0, passes = 0;   // covered by the plan
};

// vis_lo / vis_cnt: stage B's result per candidate visit (first triple, number of triples); blk_k: for the diagonal flag;
// rank: breadth-first number of camera j - mcon (launch order; may be empty = identity); slot_base: first slot of the row pieces in
// the partial-sum buffer (= number of task slots of the task kernel).  Returns 0, or -1 when the row triple array would pass 2^31 entries.
inline int row_plan_stage_c(int m, int mcon, const std::vector<int>& blk_j, const std::vector<int>& blk_k, const std::vector<int>& camptr,
                            const RowPlanParams& prm, const RowPlanA& a, const std::vector<int>& vis_lo, const std::vector<int>& vis_cnt,
                            const std::vector<int>& rank, int slot_base, RowPlan& out)
{
    const int nblk = (int)blk_j.size();
    out = RowPlan();
    out.blk_row0.assign((size_t)nblk + 1, 0);
    // rows of the block list: blocks of camera j are consecutive
    std::vector<int> row0((size_t)m + 1, nblk);
    for (int b = nblk - 1; b >= 0; --b) row0[blk_j[b]] = b;
    for (int j = m - 1; j >= 0; --j) if (row0[j] == nblk) row0[j] = row0[j + 1];
    row0[m] = nblk;
    const int tri_max = std::max(row_tri_max(prm), ROW_PASS * ((ROW_LMAX + ROW_PASS - 1) / ROW_PASS));
    const int vis_max = ROW_PMAX - (ROW_NW - 1);          // a visit that straddles waves adds a piece per boundary
    // ---- pieces per workgroup; workgroups in (j, s) order; slots are numbered afterwards in (block, segment, part) order
    struct TmpPiece { int visit, part, start, count, npass, diag; };
    struct TmpWG { int j, s, piece0, npieces, npass, wpass[ROW_NW], wpiece[ROW_NW]; };
    std::vector<TmpPiece> tp;
    std::vector<TmpWG> tw;
    std::vector<int> nparts((size_t)a.nvisits + 1, 0);
    std::vector<int> cur;                                 // blocks of the workgroup being formed
    auto close_wg = [&](int j, int s) {
        if (cur.empty()) return;
        long long total = 0;
        for (int b : cur) total += (vis_cnt[a.visbase[b] + s] + ROW_PASS - 1) / ROW_PASS;
        TmpWG w; w.j = j; w.s = s; w.piece0 = (int)tp.size(); w.npass = (int)total;
        for (int q = 0; q < ROW_NW; ++q) { w.wpass[q] = (int)((long long)q * total / ROW_NW); w.wpiece[q] = -1; }
        long long done = 0;                               // passes dealt so far; wave q owns passes [q total / NW, (q + 1) total / NW)
        for (int b : cur) {
            const int v = a.visbase[b] + s, cnt = vis_cnt[v];
            const int np = (cnt + ROW_PASS - 1) / ROW_PASS;
            int p = 0, part = nparts[v];
            while (p < np) {
                int wave = 0;
                while (wave + 1 < ROW_NW && (long long)(wave + 1) * total / ROW_NW <= done + p) ++wave;
                const long long wend = (long long)(wave + 1) * total / ROW_NW;          // first pass of the next wave
                const int take = (int)std::min<long long>(np - p, wend - (done + p));
                TmpPiece pc; pc.visit = v; pc.part = part++; pc.start = vis_lo[v] + p * ROW_PASS; pc.npass = take;
                pc.count = std::min(take * ROW_PASS, cnt - p * ROW_PASS); pc.diag = blk_j[b] == blk_k[b] ? 1 : 0;
                if (w.wpiece[wave] < 0) w.wpiece[wave] = (int)tp.size() - w.piece0;
                tp.push_back(pc);
                p += take;
            }
            nparts[v] = part;
            done += np;
            out.triples += cnt; out.passes += np;
        }
        w.npieces = (int)tp.size() - w.piece0;
        for (int q = ROW_NW - 1, nxt = w.npieces; q >= 0; --q) { if (w.wpiece[q] < 0) w.wpiece[q] = nxt; else nxt = w.wpiece[q]; }   // a wave without passes
        tw.push_back(w);
        cur.clear();
    };
    for (int j = mcon; j < m; ++j) {
        const int b0 = row0[j], b1 = row0[j + 1];
        for (int s = 0; s < a.nseg[j]; ++s) {
            long long tri = 0;
            for (int b = b0; b < b1; ++b) {
                if (!a.dense[b]) continue;
                const int cnt = vis_cnt[a.visbase[b] + s];
                if (cnt <= 0) continue;
                const int padded = (cnt + ROW_PASS - 1) / ROW_PASS * ROW_PASS;
                if (!cur.empty() && (tri + padded > tri_max || (int)cur.size() >= vis_max)) { close_wg(j, s); tri = 0; }
                cur.push_back(b); tri += padded;
            }
            close_wg(j, s);
        }
    }
    // ---- slots: (block, segment, part) order = candidate-visit order
    std::vector<int> slot0((size_t)a.nvisits + 1, 0);
    for (int v = 0; v < a.nvisits; ++v) slot0[v + 1] = slot0[v] + nparts[v];
    out.nslots = slot0[a.nvisits];
    for (int b = 0; b <= nblk; ++b) out.blk_row0[b] = slot0[a.visbase[b]];
    // ---- launch order: (slice of the camera's record list, breadth-first number of the camera, segment)
    int maxseg = 1;
    for (int j = mcon; j < m; ++j) maxseg = std::max(maxseg, a.nseg[j]);
    std::vector<int> ord(tw.size());
    for (size_t q = 0; q < ord.size(); ++q) ord[q] = (int)q;
    auto key = [&](int q) {
        const TmpWG& w = tw[q];
        const long long slice = (long long)w.s * maxseg / std::max(1, a.nseg[w.j]);
        const long long r = rank.empty() ? (long long)(w.j - mcon) : (long long)rank[w.j - mcon];
        return (slice << 40) | (r << 16) | (long long)(w.s & 0xffff);
    };
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return key(x) < key(y); });
    // every XCD (workgroup index % 8) gets one contiguous stretch of that order (as k_launch_order does for the task kernel)
    const int nwg = (int)tw.size();
    out.wgs.assign((size_t)nwg, RowWG());
    out.pieces.resize(tp.size());
    out.fills.resize(tp.size());
    int pc_out = 0;
    long long tri_out = 0;
    for (int wg = 0; wg < nwg; ++wg) {
        const int x = wg & 7;
        int next = wg >> 3;
        for (int r = 0; r < x; ++r) next += nwg > r ? (nwg - r + 7) / 8 : 0;
        const TmpWG& w = tw[ord[next]];
        RowWG h;
        h.rec0 = camptr[w.j] + w.s * prm.L;
        h.nrec = std::min(prm.L, camptr[w.j + 1] - h.rec0);
        if (tri_out + (long long)w.npass * ROW_PASS > 0x7fffffffLL) return -1;
        h.tri0 = (int)tri_out; h.npass = w.npass; h.piece0 = pc_out; h.npieces = w.npieces; h.pad[0] = h.pad[1] = 0;
        for (int q = 0; q < ROW_NW; ++q) { h.wpass[q] = w.wpass[q]; h.wpiece[q] = w.wpiece[q]; }
        for (int t = 0; t < w.npieces; ++t) {
            const TmpPiece& pc = tp[(size_t)w.piece0 + t];
            RowPiece o; o.npass = pc.npass; o.diag = pc.diag; o.out = slot_base + slot0[pc.visit] + pc.part; o.pad = 0;
            RowFill f; f.src = pc.start; f.count = pc.count; f.dst = (int)tri_out; f.rec0 = h.rec0;
            out.pieces[(size_t)pc_out] = o; out.fills[(size_t)pc_out] = f; ++pc_out;
            tri_out += (long long)pc.npass * ROW_PASS;
        }
        out.wgs[wg] = h;
    }
    out.ntri = tri_out;
    return 0;
}

}  // namespace bsfm
