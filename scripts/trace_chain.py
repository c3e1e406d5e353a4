"""Per-step timeline of the Cholesky chain from a rocprofv3 kernel trace (last LM iteration in the file).
usage: python scripts/trace_chain.py <kernel_trace.csv> [first_step last_step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('bsfm::', '').replace('void ', '')
jac = [i for i, r in enumerate(rows) if 'k_jacobian' in r['Kernel_Name']] + [len(rows)]
# the LAST complete LM iteration of the trace that holds a full factorisation (r01's v7 / v8 runs ended on a converged problem
# whose last segment had none: this script then died with an IndexError and the traceback was committed as "evidence")
seg, diag = [], []
for a, b in reversed(list(zip(jac, jac[1:]))):
    cand = rows[a:b]
    d = [r for r in cand if name(r) in ('k_potrf_diag', 'k_potrf_diag_a')]      # chain tiles (the split chain's inverse kernel is k_potrf_diag_b)
    if len(d) > len(diag):
        seg, diag = cand, d
    if len(diag) >= 2 and len(d) == len(diag) and cand is not seg:
        break
if len(diag) < 2:
    sys.exit("no factorisation with >= 2 tile columns in this trace")
t0 = int(diag[0]['Start_Timestamp'])
us = lambda t: (int(t) - t0) / 1e3
print("solve: first diag start -> last diag end: %.1f us, %d diag tiles" % (us(diag[-1]['End_Timestamp']), len(diag)))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 6)
lo, hi = max(0, min(lo, len(diag) - 2)), max(1, min(hi, len(diag) - 1))
per = [us(b['Start_Timestamp']) - us(a['Start_Timestamp']) for a, b in zip(diag, diag[1:])]
print("chain period (us) by step:", " ".join("%.0f" % p for p in per))
for k in range(lo, hi):
    a, b = int(diag[k]['Start_Timestamp']), int(diag[k + 1]['End_Timestamp'])
    print("--- step %d" % k)
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s >= a and s < int(diag[k + 1]['Start_Timestamp']) + 1:
            print("  %-18s start %9.1f  dur %7.1f  grid %6s  queue %s" % (name(r)[:18], us(s), (e - s) / 1e3, r.get('Grid_Size', '?'), r.get('Queue_Id', '?')))
