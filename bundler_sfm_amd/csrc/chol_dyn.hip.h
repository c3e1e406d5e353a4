// chol_dyn.hip.h -- the tile-dataflow Cholesky with a DYNAMIC bulk (round 6): the roles of chol_flow.hip.h, served from state words
// instead of a host-simulated ticket order.  Protocol and word layout: chol_dyn_plan.h.  Replaces sba_Axb_Chol = dpotrf("U") + dpotrs
// (lib/sba-1.5/sba_lapack.c:374-485, called from lib/sba-1.5/sba_levmar.c:1368).
//
// Visibility rules are those of chol_flow.hip.h (same invariants, same static_asserts): write-once buffers (Pc, Linv, y) are stored
// write-through and read cached only after the producer's word has been seen; rewritten data (tiles of S, E) only with agent-scope
// accesses.  The producer's word here is rowdone[i][h] (panel tiles of row i), the wd counter (inverse diagonal factor) or the state
// word itself (a tile of S between two passes): stored / incremented with agent-scope operations AFTER every wave of the workgroup has
// drained its stores and the workgroup has synchronised.  A claim is an agent-scope atomic OR on the state word: whoever sees BUSY clear
// in the returned value owns the half until it stores the word again, so ver only ever changes under ownership.
#pragma once
#include "chol_flow.hip.h"
#include "chol_dyn_plan.h"

namespace bsfm {

struct DynExtra {
    unsigned ofs_rd, ofs_tw, ofs_wd;     // word offsets behind flags = sync + 8
    unsigned np_max, np_max_rhs;
    unsigned trace_cap;                  // records the trace buffer holds (0 = no trace)
    unsigned halves_cols;                // columns up to lowcol + halves_cols are served in 64-row halves when the workgroup came out of an idle scan
    unsigned scan_rounds;                // columns scanned per pass = 8 * scan_rounds
};

constexpr int DYN_SCAN_MAX_ROUNDS = 9;      // 72 columns per pass: every column of a system of up to 72 tile columns
// control words (sync[0..8)): [0] claims made (test hook), [1] time-out, [2] chain ticket, [3] chain CU claims, [4] POTRF ticket, [5] lowcol, [6] trace records
constexpr int DYN_W_CLAIMS = 0, DYN_W_TIMEOUT = 1, DYN_W_CHAIN = 2, DYN_W_CUCLAIM = 3, DYN_W_POTRF = 4, DYN_W_LOWCOL = 5, DYN_W_TRACE = 6;

__device__ __forceinline__ unsigned dyn_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dyn_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long dyn_ld2(const unsigned* p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int dyn_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned dyn_uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// One candidate of one scanned column, as the scanning wave leaves it in LDS (all fields wave-uniform).
struct DynCand { unsigned valid, col, row, hmask, w0, w1, rd0, rd1, wd; };      // rdh = min(rowdone[row][h], rowdone[col][0], rowdone[col][1]); w = words as scanned

template <int WPS>
__global__ __launch_bounds__(512, WPS) void k_chol_dyn(FlowArgs a_param, DynExtra x_param)
{
    constexpr int V = WPS;
    const FlowKWords ka = (FlowKWords)__builtin_amdgcn_kernarg_segment_ptr();
    const FlowArgs a_seg = flow_kernel_args(ka);
    const FlowArgs& a = V == 2 ? a_seg : a_param;
    const FlowArgs* const a_in = V == 2 ? nullptr : &a_param;
    const DynExtra x = x_param;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ unsigned s_ticket;
    __shared__ int s_abort;
    __shared__ int s_role;
    __shared__ DynCand s_cand[8 * DYN_SCAN_MAX_ROUNDS];
    __shared__ unsigned s_low, s_stop;
    __shared__ unsigned s_task[10];         // type, i, j, p0, np, part, hmask(owned), valid, the owned halves' words as claimed
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = dyn_uni(tid >> 6);
    unsigned* const flags = a.sync + 8;
    const int T = dyn_uni(a.T);
    const int R = T + 1;
    // ---- role: 0 = bulk (dynamic), 1 = chain queue, 3 = POTRF queue, 2 = leave (second workgroup on a chain CU) -- as k_chol_flow
    if (wave == 0) {
        int role = 0;
        const unsigned n_cw = dyn_uni(a.n_chain_wgs);
        if (n_cw > 0u) {
            unsigned hw = 0, xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned key = ((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
            unsigned* cu_count = flags + a.nflags + key;
            unsigned* cu_state = flags + a.nflags + FLOW_CU_KEYS + key;
            unsigned slot = 0;
            if (lane == 0) slot = atomicAdd(cu_count, 1u);
            slot = dyn_uni(slot);
            if (slot == 0u) {
                unsigned r = 0;
                if (lane == 0) r = atomicAdd(a.sync + DYN_W_CUCLAIM, 1u);
                r = dyn_uni(r);
                role = r < n_cw ? 1 : 0;
                if (r == 0u && a.n_potrf > 0u) role = 3;
                if (lane == 0) __hip_atomic_store(cu_state, role ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (slot == 1u) {
                unsigned st = 0;
                for (int it = 0; it < 4096 && st == 0u; ++it) {
                    st = dyn_uni(__hip_atomic_load(cu_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (st == 0u) __builtin_amdgcn_s_sleep(1);
                }
                role = st == 2u ? 2 : 0;
            }
        }
        if (lane == 0) s_role = role;
    }
    __syncthreads();
    const int role = dyn_uni(s_role);
    if (role == 2) return;

    if (role != 0) {
        // ================================================================ the static queues (chain, POTRF): ticket, poll, work, signal
        const FlowTask* const queue = role == 3 ? a.potrf_tasks : a.chain_tasks;
        const unsigned t_end = dyn_uni(role == 3 ? a.n_potrf : a.n_chain);
        unsigned* const ticket = a.sync + (role == 3 ? DYN_W_POTRF : DYN_W_CHAIN);
        for (;;) {
            long long st0 = 0;
            if (wave == 0) {
                unsigned t = 0;
                if (lane == 0) t = atomicAdd(ticket, 1u);
                t = dyn_uni(t);
                if (lane == 0) { s_ticket = t; s_abort = 0; }
                if (a.trace) st0 = wall_clock64();
            }
            __syncthreads();
            const unsigned tk = dyn_uni(s_ticket);
            if (tk >= t_end) return;
            const FlowTask* tp = queue + tk;
            const uint32_t w0 = *reinterpret_cast<const uint32_t*>(tp);
            const uint32_t w1 = *(reinterpret_cast<const uint32_t*>(tp) + 1);
            const uint32_t w2 = *(reinterpret_cast<const uint32_t*>(tp) + 2);
            const int type = dyn_uni((int)(w0 & 255u)), np = dyn_uni((int)((w0 >> 8) & 255u));
            const int part = dyn_uni((int)((w0 >> 16) & 255u)), nwait = dyn_uni((int)(w0 >> 24));
            const int ti = dyn_uni((int)(w1 & 0xffffu)), tj = dyn_uni((int)(w1 >> 16));
            const int p0 = dyn_uni((int)(w2 & 0xffffu));
            const unsigned pad = dyn_uni(w2 >> 16);
            const unsigned sig = dyn_uni(tp->sig);
            long long st1 = 0;
            if (wave == 0) {
                const long long t_begin = wall_clock64();
                int ab = 0;
                unsigned idx[3], thr[3], msk[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int qq = q < nwait ? q : 0;
                    idx[q] = dyn_uni(tp->w[qq].idx);
                    const unsigned th = q < nwait ? dyn_uni(tp->w[qq].thr) : 0u;
                    msk[q] = (th & DYN_WAIT_MASKED) ? DYN_VER_MASK : 0xffffffffu;
                    thr[q] = th & ~DYN_WAIT_MASKED;
                }
                unsigned spins = 0;
                while (nwait > 0) {
                    const unsigned s0 = dyn_ld(flags + idx[0]) & msk[0];
                    const unsigned s1 = dyn_ld(flags + idx[1]) & msk[1];
                    const unsigned s2 = dyn_ld(flags + idx[2]) & msk[2];
                    const int ready = dyn_uni((int)(s0 >= thr[0] && s1 >= thr[1] && s2 >= thr[2]));
                    if (ready) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0u) {
                        const unsigned tmo = dyn_uni(dyn_ld(a.sync + DYN_W_TIMEOUT));
                        const int late = dyn_uni((int)(wall_clock64() - t_begin > a.spin_limit));
                        if (tmo != 0u || late) { ab = 1; break; }
                    }
                }
                if (ab && lane == 0) { dyn_st(a.sync + DYN_W_TIMEOUT, 1u); s_abort = 1; }
                if (!ab && (pad & DYN_PAD_PUBROW) && lane == 0) {        // the row's last panel tile is complete (16 TRSM32 parts): tell the bulk
                    dyn_st(flags + x.ofs_rd + 2u * (unsigned)ti, (unsigned)ti);
                    dyn_st(flags + x.ofs_rd + 2u * (unsigned)ti + 1u, (unsigned)ti);
                }
                if (a.trace) st1 = wall_clock64();
            }
            __syncthreads();
            if (dyn_uni(s_abort)) return;
            switch (type) {
            case FT_POTRF:  flow_potrf(FlowTag<V>(), ka, a_in, tj, lds); break;
            case FT_TRSM32: flow_tile32<false>(FlowTag<V>(), ka, a_in, ti, tj, 0, 0, part, lds); break;
            case FT_TRSM64: flow_trsm64(FlowTag<V>(), ka, a_in, ti, tj, 64 * part, lds); break;
            case FT_UPD32:  flow_tile32<true>(FlowTag<V>(), ka, a_in, ti, tj, p0, np, part, lds); break;
            case FT_UPD64:  flow_upd<64>(FlowTag<V>(), ka, a_in, ti, tj, p0, np, 64 * part, lds); break;
            default: break;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if (type == FT_TRSM64) {               // static TRSM64(k+2, k): the half is final, its rows of the panel tile exist
                    dyn_st(flags + x.ofs_rd + 2u * (unsigned)ti + (unsigned)part, (unsigned)tj + 1u);
                    dyn_st(flags + sig, DYN_FINAL | ((unsigned)tj << DYN_LIM_SHIFT) | (unsigned)tj);
                } else if (type == FT_UPD64) {         // static UPD64(k+2, k+1; panel k): the half now holds every panel (lim stays j - 1: nothing left for the bulk)
                    dyn_st(flags + sig, (unsigned)(p0 + np) | ((unsigned)p0 << DYN_LIM_SHIFT));
                } else {
                    __hip_atomic_fetch_add(flags + sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (a.trace) {
                    const unsigned rec = atomicAdd(a.sync + DYN_W_TRACE, 1u);
                    if (rec < x.trace_cap) {
                        unsigned xcc = 0;
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                        long long* r = a.trace + 6 * (size_t)rec;
                        r[0] = (long long)type | ((long long)ti << 8) | ((long long)tj << 20) | ((long long)p0 << 32) | ((long long)np << 44) | ((long long)part << 52);
                        r[1] = st0; r[2] = st1; r[3] = wall_clock64();
                        r[4] = (long long)(xcc & 15u) | ((long long)blockIdx.x << 8) | ((long long)role << 32);
                        r[5] = 0;
                    }
                }
            }
        }
    }

    // ==================================================================== the bulk: scan, claim, work, release
    const unsigned wg_hash = (unsigned)blockIdx.x * 2654435761u;
    unsigned idle_rounds = 0;          // scans without a claim since the last task (0: the workgroup is in a backlog, > 0: it has been waiting for work)
    unsigned attempt = 0;
    long long t_idle0 = 0;
    if (wave == 0) t_idle0 = wall_clock64();
    // lowcol and the time-out word are read TOGETHER with the scan's loads and used by the next pass (one round trip fewer per task):
    // lowcol only grows, so an old value costs a column or two of wasted scanning, never correctness
    if (tid == 0) { s_low = 0u; s_stop = 0u; }
    __syncthreads();
    for (;;) {
        long long st0 = 0;
        if (wave == 0 && a.trace) st0 = wall_clock64();
        if (dyn_uni(s_stop)) return;
        const int lowcol = (int)dyn_uni(s_low);
        const int ncols = min(T - lowcol, 8 * min((int)x.scan_rounds, DYN_SCAN_MAX_ROUNDS));
        const int nrounds = (ncols + 7) / 8;
        unsigned lc_next = 0u, tmo_next = 0u;
        if (wave == 0) { lc_next = dyn_ld(a.sync + DYN_W_LOWCOL); tmo_next = dyn_ld(a.sync + DYN_W_TIMEOUT); }
        // ---- scan: wave w looks at columns lowcol + w + 8 r; one lane per tile row (rows j .. j + 63 first), both halves of a tile in one 8-byte load.
        // The first chunk of every column of the pass is fetched before any of them is looked at.
        unsigned long long tw_pre[DYN_SCAN_MAX_ROUNDS], rd_pre[DYN_SCAN_MAX_ROUNDS], rdj_pre[DYN_SCAN_MAX_ROUNDS];
        unsigned wd_pre[DYN_SCAN_MAX_ROUNDS];
#pragma unroll
        for (int r = 0; r < DYN_SCAN_MAX_ROUNDS; ++r) {
            const int j = min(lowcol + wave + 8 * r, T - 1);
            const int ic = min(j + lane, R - 1);
            tw_pre[r] = dyn_ld2(flags + x.ofs_tw + 2u * ((unsigned)j * (unsigned)R + (unsigned)ic));
            rd_pre[r] = dyn_ld2(flags + x.ofs_rd + 2u * (unsigned)ic);
            rdj_pre[r] = dyn_ld2(flags + x.ofs_rd + 2u * (unsigned)j);
            wd_pre[r] = dyn_ld(flags + x.ofs_wd + (unsigned)j);
        }
        if (tid < 8 * DYN_SCAN_MAX_ROUNDS) s_cand[tid].valid = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < DYN_SCAN_MAX_ROUNDS; ++r) {
            const int j = lowcol + wave + 8 * r;
            if (r >= nrounds || j >= T) continue;
            const unsigned rdj = dyn_uni(min((unsigned)rdj_pre[r] & DYN_VER_MASK, (unsigned)(rdj_pre[r] >> 32) & DYN_VER_MASK));
            const unsigned wdj = dyn_uni(wd_pre[r]);
            const bool urgent = idle_rounds > 0u && j <= lowcol + (int)x.halves_cols;
            bool all_done = true;
            unsigned found = 0u;
            for (int base = j; base < R && !found; base += 64) {
                const int i = base + lane;
                const bool in = i < R;
                const int ic = in ? i : R - 1;
                unsigned long long tw2 = tw_pre[r], rd2 = rd_pre[r];
                if (base != j) {      // (systems of more than 63 tile rows below the diagonal: the further chunks of a column, one round trip each)
                    tw2 = dyn_ld2(flags + x.ofs_tw + 2u * ((unsigned)j * (unsigned)R + (unsigned)ic));
                    rd2 = dyn_ld2(flags + x.ofs_rd + 2u * (unsigned)ic);
                }
                const unsigned w[2] = { (unsigned)tw2, (unsigned)(tw2 >> 32) };
                const unsigned rdi[2] = { (unsigned)rd2 & DYN_VER_MASK, (unsigned)(rd2 >> 32) & DYN_VER_MASK };
                unsigned kind[2], rdm[2];
                bool done = true;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned ver = w[h] & DYN_VER_MASK, lim = (w[h] >> DYN_LIM_SHIFT) & DYN_VER_MASK;
                    const bool fin = (w[h] & DYN_FINAL) != 0u, busy = (w[h] & DYN_BUSY) != 0u, elig = (w[h] & DYN_ELIG) != 0u;
                    rdm[h] = min(rdi[h], rdj);
                    const bool upd = in && !fin && !busy && ver < lim && min(rdm[h], lim) > ver;
                    const bool fz = in && !fin && !busy && elig && ver == (unsigned)j && wdj != 0u;
                    kind[h] = upd ? 1u : fz ? 2u : 0u;
                    done = done && (!in || fin || (!elig && ver >= lim));
                }
                all_done = all_done && (__builtin_amdgcn_ballot_w64(!done) == 0ull);
                const unsigned long long m = __builtin_amdgcn_ballot_w64(kind[0] != 0u || kind[1] != 0u);
                if (m != 0ull) {
                    // lowest row first while the workgroup is in a backlog; spread by workgroup when it comes out of an idle scan (everybody
                    // who was waiting sees the same new work at the same moment)
                    const unsigned rot = idle_rounds > 0u ? ((wg_hash >> 16) + attempt) & 63u : 0u;
                    const unsigned long long mr = rot ? ((m >> rot) | (m << (64u - rot))) : m;
                    const unsigned sel = ((unsigned)__builtin_ctzll(mr) + rot) & 63u;
                    const unsigned k0 = dyn_uni((unsigned)__shfl((int)kind[0], (int)sel, 64)), k1 = dyn_uni((unsigned)__shfl((int)kind[1], (int)sel, 64));
                    const unsigned sw0 = dyn_uni((unsigned)__shfl((int)w[0], (int)sel, 64)), sw1 = dyn_uni((unsigned)__shfl((int)w[1], (int)sel, 64));
                    const unsigned sr0 = dyn_uni((unsigned)__shfl((int)rdm[0], (int)sel, 64)), sr1 = dyn_uni((unsigned)__shfl((int)rdm[1], (int)sel, 64));
                    unsigned hm;
                    if (k0 == 1u && k1 == 1u && ((sw0 ^ sw1) & DYN_VER_MASK) == 0u && !urgent) hm = 3u;          // both halves, one pass
                    else if (k0 != 0u && k1 != 0u) hm = ((wg_hash >> 8) + attempt) & 1u ? 2u : 1u;             // either half: spread
                    else hm = k0 != 0u ? 1u : 2u;
                    if (lane == 0) {
                        DynCand c; c.valid = 1u; c.col = (unsigned)j; c.row = (unsigned)(base + (int)sel); c.hmask = hm; c.w0 = sw0; c.w1 = sw1; c.rd0 = sr0; c.rd1 = sr1; c.wd = wdj;
                        s_cand[wave + 8 * r] = c;
                    }
                    found = 1u;
                }
            }
            if (!found && all_done && j == lowcol && lane == 0) atomicMax(a.sync + DYN_W_LOWCOL, (unsigned)(j + 1));
        }
        if (wave == 0) {
            // the next pass's lowcol / stop decision
            lc_next = dyn_uni(lc_next); tmo_next = dyn_uni(tmo_next);
            int stop = (tmo_next != 0u || lc_next >= (unsigned)T) ? 1 : 0;
            if (!stop && idle_rounds > 0u && (idle_rounds & 15u) == 0u) {
                const int late = dyn_uni((int)(wall_clock64() - t_idle0 > a.spin_limit));
                if (late) { if (lane == 0) dyn_st(a.sync + DYN_W_TIMEOUT, 1u); stop = 1; }
            }
            if (lane == 0) { s_low = max(lc_next, (unsigned)lowcol); s_stop = (unsigned)stop; }
        }
        __syncthreads();
        // ---- choose (wave 0) and claim
        if (wave == 0) {
            unsigned valid = 0u;
            {
                const int nslots = 8 * nrounds;
                static_assert(8 * DYN_SCAN_MAX_ROUNDS <= 128, "two ballots cover the candidate slots");
                const unsigned v0 = lane < nslots ? s_cand[lane < nslots ? lane : 0].valid : 0u;
                const unsigned v1 = lane + 64 < nslots ? s_cand[lane + 64 < nslots ? lane + 64 : 0].valid : 0u;
                const unsigned long long vm0 = __builtin_amdgcn_ballot_w64(v0 != 0u), vm1 = __builtin_amdgcn_ballot_w64(v1 != 0u);
                if ((vm0 | vm1) != 0ull) {
                    // slots are in column order.  Backlog: the lowest column.  Out of an idle scan: any of them, by workgroup
                    unsigned pick = vm0 != 0ull ? (unsigned)__builtin_ctzll(vm0) : 64u + (unsigned)__builtin_ctzll(vm1);
                    if (idle_rounds > 0u) {
                        const unsigned n0 = (unsigned)__builtin_popcountll(vm0), n = n0 + (unsigned)__builtin_popcountll(vm1);
                        unsigned q = ((wg_hash >> 20) + attempt) % n;
                        unsigned long long t = q < n0 ? vm0 : vm1;
                        const unsigned ofs = q < n0 ? 0u : 64u;
                        if (q >= n0) q -= n0;
                        while (q--) t &= t - 1ull;
                        pick = ofs + (unsigned)__builtin_ctzll(t);
                    }
                    valid = 1u;
                    const DynCand c = s_cand[pick];
                    const unsigned j = dyn_uni(c.col), i = dyn_uni(c.row), hm = dyn_uni(c.hmask);
                    unsigned* pair = flags + x.ofs_tw + 2u * (j * (unsigned)R + i);
                    unsigned o0 = DYN_BUSY, o1 = DYN_BUSY;       // words as the claim found them (BUSY set = not ours)
                    if (hm == 3u) {
                        unsigned long long old = 0ull;
                        if (lane == 0) old = __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(pair), (unsigned long long)DYN_BUSY | ((unsigned long long)DYN_BUSY << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        o0 = dyn_uni((unsigned)old); o1 = dyn_uni((unsigned)(old >> 32));
                    } else {
                        unsigned old = 0u;
                        if (lane == 0) old = __hip_atomic_fetch_or(pair + (hm == 2u ? 1 : 0), DYN_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        old = dyn_uni(old);
                        if (hm == 2u) o1 = old; else o0 = old;
                    }
                    // what the owned halves can do, from the words the claim returned (ver is exact under ownership) and the scanned counters (monotone: at worst too small)
                    const unsigned rds[2] = { dyn_uni(c.rd0), dyn_uni(c.rd1) };
                    const unsigned ow[2] = { o0, o1 };
                    unsigned own[2], act[2], av[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        own[h] = (ow[h] & DYN_BUSY) == 0u ? 1u : 0u;
                        const unsigned ver = ow[h] & DYN_VER_MASK, lim = (ow[h] >> DYN_LIM_SHIFT) & DYN_VER_MASK;
                        const bool fin = (ow[h] & DYN_FINAL) != 0u;
                        const unsigned cap = min(rds[h], lim);
                        av[h] = cap > ver ? cap - ver : 0u;
                        act[h] = !own[h] || fin ? 0u : av[h] > 0u ? 1u : ((ow[h] & DYN_ELIG) && ver == j && dyn_uni(c.wd) != 0u) ? 2u : 0u;
                    }
                    unsigned keep = 0u, type = 0u, np = 0u, p0 = 0u, part = 0u;
                    const unsigned cap_np = i == (unsigned)T ? x.np_max_rhs : x.np_max;
                    if (act[0] == 1u && act[1] == 1u && ((ow[0] ^ ow[1]) & DYN_VER_MASK) == 0u) {
                        keep = 3u; type = FT_UPD128; p0 = ow[0] & DYN_VER_MASK; np = min(min(av[0], av[1]), cap_np);
                    } else {
                        const int h = act[0] != 0u ? 0 : act[1] != 0u ? 1 : -1;
                        if (h >= 0) {
                            keep = 1u << h; part = (unsigned)h; p0 = ow[h] & DYN_VER_MASK;
                            if (act[h] == 1u) { np = min(av[h], cap_np); type = i == (unsigned)T ? FT_FUPD : FT_UPD64; }
                            else { np = 0u; type = i == (unsigned)T ? FT_FTRSM : FT_TRSM64; }
                        }
                    }
                    // give back what was claimed but is not used
                    if (lane == 0) {
                        if (own[0] && !(keep & 1u)) __hip_atomic_fetch_and(pair, ~DYN_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (own[1] && !(keep & 2u)) __hip_atomic_fetch_and(pair + 1, ~DYN_BUSY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_task[0] = type; s_task[1] = i; s_task[2] = j; s_task[3] = p0; s_task[4] = np; s_task[5] = part; s_task[6] = keep;
                        s_task[7] = keep != 0u ? 1u : 0u; s_task[8] = ow[0]; s_task[9] = ow[1];
                    }
                }
            }
            if (!valid && lane == 0) s_task[7] = 0u;
            if (dyn_uni(valid) == 0u) {
                // nothing to do anywhere in the window: wait a little (longer the longer it lasts) before looking again
                const unsigned n = idle_rounds < 8u ? 1u : idle_rounds < 64u ? 4u : 16u;
                for (unsigned q = 0; q < n; ++q) __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
        ++attempt;
        if (dyn_uni(s_task[7]) == 0u) { ++idle_rounds; continue; }
        const int type = (int)dyn_uni(s_task[0]), ti = (int)dyn_uni(s_task[1]), tj = (int)dyn_uni(s_task[2]);
        const int p0 = (int)dyn_uni(s_task[3]), np = (int)dyn_uni(s_task[4]), part = (int)dyn_uni(s_task[5]);
        const unsigned keep = dyn_uni(s_task[6]);
        long long st1 = 0;
        if (wave == 0 && a.trace) st1 = wall_clock64();
        if (a.stall_ticket >= 0) {            // TEST HOOK (BSFM_FLOW_TEST_STALL=k): the k-th claim of the launch is never released -- what a starved launch looks like
            __shared__ unsigned s_claim;
            if (tid == 0) s_claim = atomicAdd(a.sync + DYN_W_CLAIMS, 1u);
            __syncthreads();
            if ((int)dyn_uni(s_claim) == a.stall_ticket) { idle_rounds = 1u; if (wave == 0) t_idle0 = wall_clock64(); continue; }
        }
        switch (type) {
        case FT_TRSM64: flow_trsm64(FlowTag<V>(), ka, a_in, ti, tj, 64 * part, lds); break;
        case FT_UPD64:  flow_upd<64>(FlowTag<V>(), ka, a_in, ti, tj, p0, np, 64 * part, lds); break;
        case FT_UPD128: flow_upd<128>(FlowTag<V>(), ka, a_in, ti, tj, p0, np, 0, lds); break;
        case FT_FTRSM:  flow_ftrsm(a, tj, lds); break;
        case FT_FUPD:   flow_fupd(a, tj, p0, np); break;
        default: break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) {
            unsigned* pair = flags + x.ofs_tw + 2u * ((unsigned)tj * (unsigned)R + (unsigned)ti);
            if (type == FT_TRSM64 || type == FT_FTRSM) {
                dyn_st(flags + x.ofs_rd + 2u * (unsigned)ti + (unsigned)part, (unsigned)tj + 1u);
                dyn_st(pair + part, DYN_FINAL | ((unsigned)tj << DYN_LIM_SHIFT) | (unsigned)tj);
            } else {
                // the word again, with the new version and without BUSY: lim and the eligibility bit are constants of the half
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (keep & (1u << h)) dyn_st(pair + h, (s_task[8 + h] & ~(DYN_VER_MASK | DYN_BUSY)) | (unsigned)(p0 + np));
            }
            if (a.trace) {
                const unsigned rec = atomicAdd(a.sync + DYN_W_TRACE, 1u);
                if (rec < x.trace_cap) {
                    unsigned xcc = 0;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    long long* r = a.trace + 6 * (size_t)rec;
                    r[0] = (long long)type | ((long long)ti << 8) | ((long long)tj << 20) | ((long long)p0 << 32) | ((long long)np << 44) | ((long long)part << 52);
                    r[1] = st0; r[2] = st1; r[3] = wall_clock64();
                    r[4] = (long long)(xcc & 15u) | ((long long)blockIdx.x << 8);
                    r[5] = (long long)idle_rounds;
                }
            }
        }
        idle_rounds = 0u;
        if (wave == 0) t_idle0 = wall_clock64();
    }
}

// ------------------------------------------------------------------------------------------------ host side
__global__ __launch_bounds__(256) void k_dyn_begin(unsigned* __restrict__ sync, const unsigned* __restrict__ init, unsigned nwords, unsigned sync_words,
                                                   int* __restrict__ bflags, int nbflags, double* __restrict__ etmp, const double* __restrict__ E, int n, int ld)
{
    const unsigned stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned q = t0; q < sync_words; q += stride) sync[q] = (q >= 8u && q < 8u + nwords) ? init[q - 8u] : 0u;
    for (unsigned q = t0; q < (unsigned)nbflags; q += stride) bflags[q] = 0;
    for (unsigned q = t0; q < (unsigned)ld; q += stride) etmp[q] = q < (unsigned)n ? E[q] : 0.0;
}

struct DynWorkspace {
    int nblk = 0;
    std::vector<int> env_key;
    DynPlan plan;
    FlowTask* d_tasks = nullptr;           // chain queue, then POTRF queue
    unsigned* d_init = nullptr;            // initial image of the state words
    unsigned* d_sync = nullptr;
    size_t sync_words = 0;
    long long* d_trace = nullptr; unsigned trace_cap = 0;
    int np_max = 4, np_max_rhs = 8, halves_cols = 1, scan_rounds = DYN_SCAN_MAX_ROUNDS;
    int chain_wgs = 19;
};

inline void dyn_free(DynWorkspace& d)
{
    bsfm::dev_free(d.d_tasks, true); bsfm::dev_free(d.d_init, true); bsfm::dev_free(d.d_sync, true);
    if (d.d_trace) (void)hipFree(d.d_trace);
    d = DynWorkspace();
}

// Prepares the dynamic launch for nblk tile columns and the envelope of w (shares the panel-tile buffer, the events and the test hooks of the FlowWorkspace).
inline int dyn_prepare(FlowWorkspace& f, DynWorkspace& d, int nblk, const std::vector<int>& env_rows)
{
    std::vector<int> key;
    if ((int)env_rows.size() >= nblk) { key.resize((size_t)nblk); for (int k = 0; k < nblk; ++k) key[k] = k + env_rows[k]; }
    if (d.d_tasks && d.nblk == nblk && d.env_key == key) return 0;
    bsfm::dev_free(d.d_tasks, true); d.d_tasks = nullptr;
    bsfm::dev_free(d.d_init, true); d.d_init = nullptr;
    bsfm::dev_free(d.d_sync, true); d.d_sync = nullptr;
    if (f.nblk != nblk) { bsfm::dev_free(f.pc, true); f.pc = nullptr; }
    if (const char* e = getenv("BSFM_FLOW_WGS")) f.wgs = std::max(2, atoi(e));
    if (const char* e = getenv("BSFM_FLOW_TRACE")) f.trace = atoi(e) != 0;
    f.spin_limit = FLOW_SPIN_LIMIT_TICKS; f.stall_ticket = -1; f.stall_bwd_col = -1;
    if (const char* e = getenv("BSFM_FLOW_SPIN_MS")) f.spin_limit = std::max(1LL, (long long)atoll(e)) * 100000LL;
    if (const char* e = getenv("BSFM_FLOW_TEST_STALL")) f.stall_ticket = atoi(e);
    if (const char* e = getenv("BSFM_FLOW_TEST_STALL_BWD")) f.stall_bwd_col = atoi(e);
    if (dyn_build_plan(nblk, key, d.plan) != 0) return -1;
    d.np_max = 4; d.np_max_rhs = 8; d.halves_cols = 1; d.scan_rounds = DYN_SCAN_MAX_ROUNDS;
    if (const char* e = getenv("BSFM_FLOW_NPMAX")) d.np_max = std::max(1, std::min(16, atoi(e)));
    if (const char* e = getenv("BSFM_DYN_HALVES")) d.halves_cols = std::max(0, atoi(e));
    if (const char* e = getenv("BSFM_DYN_SCAN")) d.scan_rounds = std::max(1, std::min(DYN_SCAN_MAX_ROUNDS, atoi(e)));
    // chain workgroups: POTRF alone + 16 for the blocks of TRSM32 + 2 for the halves of the second-order chain that run beside them
    d.chain_wgs = 19;
    if (const char* e = getenv("BSFM_FLOW_CHAIN_WGS")) d.chain_wgs = std::max(2, atoi(e));
    d.chain_wgs = std::min(d.chain_wgs, std::max(2, f.wgs / 4));
    f.chain_wgs = d.chain_wgs;
    {
        int lat_tiles = 38;
        if (const char* e = getenv("BSFM_FLOW_LATENCY_TILES")) lat_tiles = atoi(e);
        f.latency_build = nblk <= lat_tiles;
    }
    f.nblk = nblk; f.env_key = key;      // (the static queues of f are not built: f.d_tasks stays as it was)
    d.nblk = nblk; d.env_key = key;
    const size_t nt = d.plan.chain.size() + d.plan.potrf.size();
    if (bsfm::dev_alloc((void**)&d.d_tasks, std::max<size_t>(1, nt) * sizeof(FlowTask)) != hipSuccess) return -1;
    if (!d.plan.chain.empty() && hipMemcpy(d.d_tasks, d.plan.chain.data(), d.plan.chain.size() * sizeof(FlowTask), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemcpy(d.d_tasks + d.plan.chain.size(), d.plan.potrf.data(), d.plan.potrf.size() * sizeof(FlowTask), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (bsfm::dev_alloc((void**)&d.d_init, (size_t)d.plan.nwords * sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemcpy(d.d_init, d.plan.init.data(), (size_t)d.plan.nwords * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) return -1;
    d.sync_words = 8 + (size_t)d.plan.nwords + 2 * (size_t)FLOW_CU_KEYS;
    if (bsfm::dev_alloc((void**)&d.d_sync, d.sync_words * sizeof(unsigned)) != hipSuccess) return -1;
    if (!f.pc) {
        const size_t ntile = std::max<size_t>(1, (size_t)nblk * (size_t)(nblk - 1) / 2);
        if (bsfm::dev_alloc((void**)&f.pc, ntile * FLOW_TL * sizeof(double)) != hipSuccess) return -1;
    }
    if (d.d_trace) { (void)hipFree(d.d_trace); d.d_trace = nullptr; }
    d.trace_cap = 0;
    if (f.trace) {
        // every (half tile, panel) can be a visit of its own, plus the finalisations and the chain's tasks
        const size_t cap = (size_t)(2.0 * (d.plan.upd_tiles + d.plan.trsm_tiles)) + 4 * (size_t)nblk * (size_t)(nblk + 1) + nt + 64;
        if (hipMalloc((void**)&d.d_trace, (cap * 6 + 40 * (size_t)nblk) * sizeof(long long)) != hipSuccess) return -1;
        (void)hipMemset(d.d_trace, 0, (cap * 6 + 40 * (size_t)nblk) * sizeof(long long));
        d.trace_cap = (unsigned)cap;
    }
    f.flops = (d.plan.upd_tiles + d.plan.trsm_tiles) * 2.0 * POTRF_NB * POTRF_NB * POTRF_NB;
    if (!f.k0) { (void)hipEventCreate(&f.k0); (void)hipEventCreate(&f.k1); }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_dyn<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(FLOW_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_dyn<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(FLOW_LDS_DOUBLES * sizeof(double))) != hipSuccess) return -1;
    return 0;
}

inline void dyn_dump_trace(FlowWorkspace& f, DynWorkspace& d, hipStream_t st)
{
    const char* path = getenv("BSFM_FLOW_TRACE_FILE");
    if (!f.trace || !d.d_trace || !path) return;
    (void)hipStreamSynchronize(st);
    unsigned ctl[8];
    if (hipMemcpy(ctl, d.d_sync, sizeof ctl, hipMemcpyDeviceToHost) != hipSuccess) return;
    const size_t nrec = std::min<size_t>(ctl[DYN_W_TRACE], d.trace_cap);
    std::vector<long long> h(nrec * 6);
    if (nrec && hipMemcpy(h.data(), d.d_trace, nrec * 6 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    FILE* fp = fopen(path, "w");
    if (!fp) return;
    long long t0 = nrec ? h[1] : 0;
    for (size_t q = 0; q < nrec; ++q) if (h[6 * q + 1] > 0) t0 = std::min(t0, h[6 * q + 1]);
    fprintf(fp, "# record type i j p0 np part  t_scan t_claimed t_done (us since the first scan)  xcc wg queue idle_scans   (%u records, capacity %u)\n", ctl[DYN_W_TRACE], d.trace_cap);
    for (size_t q = 0; q < nrec; ++q) {
        const long long* r = h.data() + 6 * q;
        fprintf(fp, "%zu %lld %lld %lld %lld %lld %lld %.2f %.2f %.2f %lld %lld %lld %lld\n", q, r[0] & 255, (r[0] >> 8) & 4095, (r[0] >> 20) & 4095, (r[0] >> 32) & 4095,
                (r[0] >> 44) & 255, (r[0] >> 52) & 255, (r[1] - t0) * 0.01, (r[2] - t0) * 0.01, (r[3] - t0) * 0.01, r[4] & 15, (r[4] >> 8) & 0xffffff, r[4] >> 32, r[5]);
    }
    std::vector<long long> ph(40 * (size_t)f.nblk);
    if (hipMemcpy(ph.data(), d.d_trace + 6 * (size_t)d.trace_cap, ph.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
        fprintf(fp, "# POTRF phases per column (us since entry, stamped by the factor wave)\n");
        for (int k = 0; k < f.nblk; ++k) {
            const long long* q = ph.data() + 40 * (size_t)k;
            fprintf(fp, "#P %d: %.2f (first block factored %.2f) |", k, (q[2] - q[1]) * 0.01, (q[3] - q[1]) * 0.01);
            for (int s2 = 0; s2 < 8; ++s2) fprintf(fp, " [%.2f] %.2f %.2f |", (q[4 + 4 * s2] - q[1]) * 0.01, (q[5 + 4 * s2] - q[1]) * 0.01, s2 < 7 ? (q[7 + 4 * s2] - q[1]) * 0.01 : (q[6 + 4 * s2] - q[1]) * 0.01);
            fprintf(fp, " %.2f %.2f \n", (q[36] - q[1]) * 0.01, (q[37] - q[1]) * 0.01);
        }
    }
    fclose(fp);
}

// Solves S x = E like flow_solve, with the dynamic bulk.  info: 0, dpotrf's k, or POTRF_INFO_TIMEOUT.
inline int dyn_solve(PotrfWorkspace& w, FlowWorkspace& f, DynWorkspace& d, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    const int nblk = (n + POTRF_NB - 1) / POTRF_NB;
    if (dyn_prepare(f, d, nblk, w.env_rows) != 0) return -1;
    hipLaunchKernelGGL(k_dyn_begin, dim3((unsigned)std::min<size_t>(64, (std::max<size_t>(d.sync_words, (size_t)ld) + 255) / 256)), dim3(256), 0, st,
                       d.d_sync, (const unsigned*)d.d_init, d.plan.nwords, (unsigned)d.sync_words, w.bflags, w.nblk + 1, w.etmp, E, n, ld);
    FlowArgs a;
    memset(&a, 0, sizeof a);
    a.S = S; a.ld = ld; a.n_total = n; a.T = nblk; a.Pc = f.pc; a.Linv = w.linv; a.E = w.etmp; a.y = w.y;
    a.tasks = d.d_tasks; a.n_bulk = 0;
    a.chain_tasks = d.d_tasks; a.n_chain = (unsigned)d.plan.chain.size();
    a.potrf_tasks = d.d_tasks + d.plan.chain.size(); a.n_potrf = (unsigned)d.plan.potrf.size();
    a.n_chain_wgs = (unsigned)d.chain_wgs; a.sync = d.d_sync; a.nflags = d.plan.nwords; a.info = d_info;
    a.trace = f.trace ? d.d_trace : nullptr; a.ptrace_ofs = (unsigned)(6 * (size_t)d.trace_cap);
    a.spin_limit = f.spin_limit; a.stall_ticket = f.stall_ticket;
    DynExtra x;
    x.ofs_rd = d.plan.ofs_rd; x.ofs_tw = d.plan.ofs_tw; x.ofs_wd = d.plan.ofs_wd;
    x.np_max = (unsigned)d.np_max; x.np_max_rhs = (unsigned)d.np_max_rhs; x.trace_cap = d.trace_cap;
    x.halves_cols = (unsigned)d.halves_cols; x.scan_rounds = (unsigned)d.scan_rounds;
    const size_t lds_bytes = FLOW_LDS_DOUBLES * sizeof(double);
    const bool timed = w.timing && f.k0;
    if (timed) {
        if (f.kern_pending) { float ms = 0.f; if (hipEventElapsedTime(&ms, f.k0, f.k1) == hipSuccess && ms >= 0.f) { f.kern_ms += ms; f.kern_cnt++; } f.kern_pending = false; }
        (void)hipEventRecord(f.k0, st);
    }
    const size_t work = (size_t)d.plan.n_halves + d.plan.chain.size() + d.plan.potrf.size();
    if (f.latency_build) {
        const unsigned grid = (unsigned)std::min<size_t>((size_t)f.wgs / 2, work + (size_t)d.chain_wgs);
        hipLaunchKernelGGL(k_chol_dyn<2>, dim3(grid), dim3(512), lds_bytes, st, a, x);
    } else {
        const unsigned grid = (unsigned)std::min<size_t>((size_t)f.wgs, work + 2 * (size_t)d.chain_wgs);
        hipLaunchKernelGGL(k_chol_dyn<4>, dim3(grid), dim3(512), lds_bytes, st, a, x);
    }
    if (timed) { (void)hipEventRecord(f.k1, st); f.kern_pending = true; }
    const bool env = (int)w.env_rows.size() >= nblk && w.d_last != nullptr;
    for (int first = 0; first < nblk; first += POTRF_MAX_TILES)
        hipLaunchKernelGGL(k_bwd_flow, dim3(std::min(POTRF_MAX_TILES, nblk - first)), dim3(256), 0, st, (const double*)f.pc, nblk, first,
                           (const double*)w.linv, (const double*)w.y, w.xs, w.bflags, w.bflags + w.nblk, (const int*)(env ? w.d_last : nullptr),
                           f.spin_limit, f.stall_bwd_col);
    hipLaunchKernelGGL(k_flow_end, dim3((unsigned)std::min(64, (n + 255) / 256)), dim3(256), 0, st, (const unsigned*)d.d_sync, (const int*)(w.bflags + w.nblk),
                       d_info, (const double*)w.xs, x_out, n);
    if (f.trace) dyn_dump_trace(f, d, st);
    return 0;
}

inline int flow_solve_dispatch(PotrfWorkspace& w, double* S, int ld, int n, const double* E, double* x_out, int* d_info, hipStream_t st)
{
    if (!w.flow) w.flow = new FlowWorkspace();
    FlowWorkspace& f = *w.flow;
    if (w.dist_comm) {      // the ranks of a communicator factor the replicated system together (chol_flow.hip.h: FlowDist)
        if (!f.dist) { f.dist = new FlowDist(); f.dist->comm = w.dist_comm; }
        return flow_solve_dist(w, f, *f.dist, S, ld, n, E, x_out, d_info, st);
    }
    if (f.dynamic < 0) {
        // Round 6, measured (profiles/r06_dynamic_bulk_*.txt): the dynamic bulk is correct and bit-identical to itself for any number of
        // workgroups, but at 71 tile columns it takes 8.9 - 13 ms against 6.2 ms for the static order -- every row's TRSM -> UPD chain pays
        // the discovery latency (a scan of shared state words: 8 - 12 us under the contention of 470 scanning workgroups, against ~3 us for a
        // pre-assigned waiter), once per tile column, and that throttles the front.  The static order stays the default; this is the opt-in.
        f.dynamic = 0;
        if (const char* e = getenv("BSFM_FLOW_SCHED")) f.dynamic = strcmp(e, "dynamic") == 0;
    }
    if (!f.dynamic) return flow_solve(w, f, S, ld, n, E, x_out, d_info, st);
    if (!f.dyn) f.dyn = new DynWorkspace();
    return dyn_solve(w, f, *static_cast<DynWorkspace*>(f.dyn), S, ld, n, E, x_out, d_info, st);
}
inline void flow_release(PotrfWorkspace& w)
{
    if (!w.flow) return;
    if (w.flow->dyn) { DynWorkspace* d = static_cast<DynWorkspace*>(w.flow->dyn); dyn_free(*d); delete d; w.flow->dyn = nullptr; }
    flow_free(*w.flow); delete w.flow; w.flow = nullptr;
}

}  // namespace bsfm
