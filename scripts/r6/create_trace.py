"""Kernel-by-kernel list of what one run_sfm call on a 14-camera problem launches BEFORE its first LM kernel (index build, uploads):
rocprofv3 --kernel-trace --memory-copy-trace? (kernel trace only here) -- python scripts/r6/create_trace.py run | show <csv>"""
import csv, os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import bundler_sfm_amd as B
    m, n = int(sys.argv[2]), int(sys.argv[3])
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    opt = B.default_options(verbose=0)
    for _ in range(3):
        c2 = B.copy_cameras(s["cams"]); p2 = s["pts"].copy()
        B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, c2, p2, eps2=1e-12, options=opt)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    jac = [i for i, r in enumerate(rows) if 'k_jacobian' in r['Kernel_Name']]
    # the last call: kernels between the previous call's last kernel and this call's first Jacobian
    # find the start of the last call = first kernel after the largest gap before the last burst of Jacobians
    calls = [0]
    for i in range(1, len(rows)):
        if int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']) > 300000: calls.append(i)
    i0 = calls[-1]
    i1 = next(i for i in jac if i >= i0)
    t0 = int(rows[i0]['Start_Timestamp'])
    print("last call: %d kernels before the first Jacobian, %.1f us from the first of them to it" % (i1 - i0, (int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
    prev = t0
    for r in rows[i0:i1 + 1]:
        nme = r['Kernel_Name'].split('(')[0].replace('bsfm::', '').replace('void ', '').replace('(anonymous namespace)::', '')
        s_, e_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print("  %-60s start %7.1f dur %6.1f gap %6.1f" % (nme[:60], (s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - prev) / 1e3))
        prev = max(prev, e_)
