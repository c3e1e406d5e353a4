"""One LM iteration of a Bundler-sized problem, launch by launch (rocprofv3 kernel trace): what a 14 / 50-camera iteration is made of.
usage (on the GPU box): rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python scripts/small_iter_trace.py run <cams> <pts>
       python scripts/small_iter_trace.py show <kernel_trace.csv>"""
import csv, os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bundler_sfm_amd as B
    m, n = int(sys.argv[2]), int(sys.argv[3])
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    opt = B.default_options(verbose=0)
    for _ in range(2):
        c2 = B.copy_cameras(s["cams"]); p2 = s["pts"].copy()
        B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, c2, p2, eps2=1e-12, options=opt)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    jac = [i for i, r in enumerate(rows) if 'k_jacobian' in r['Kernel_Name']]
    i0, i1 = jac[-3], jac[-2]
    t0 = int(rows[i0]['Start_Timestamp'])
    print("iteration: %.1f us, %d launches" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3, i1 - i0))
    prev_end = t0
    for r in rows[i0:i1]:
        n = r['Kernel_Name'].split('(')[0].replace('bsfm::', '').replace('void ', '').replace('(anonymous namespace)::', '')
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print("  %-34s start %7.1f  dur %6.1f  gap %5.1f" % (n[:34], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = max(prev_end, e)
