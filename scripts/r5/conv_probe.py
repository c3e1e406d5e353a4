import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import bundler_sfm_amd as B
M, N, DEG = 1000, 500000, 10
s = B.synth_ba(M, N, DEG)
for rep in range(2):
    pb = B.Problem(N, M, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"],
                   options=B.default_options(jacobian=B.JAC_FD, verbose=0, itmax=40, opts=[1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2]))
    rc, info = pb.solve()
    print(rep, rc, list(info))
    pb.close()
