"""Soak: the headline scene solved to convergence several times through the resident API; iterations, final cost (must be the same bits every time)
and the number of dataflow launches that had to be repeated on the stream-ordered schedule (must be 0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bundler_sfm_amd as B
m, n = 1000, 500000
s = B.synth_ba(m, n, 10)
costs = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    pb = B.Problem(n, m, s["rowptr"], s["colidx"], s["proj"], s["cams"], s["pts"], options=B.default_options(verbose=0))
    t = time.perf_counter(); rc, info = pb.solve(); B.lib.bsfm_device_synchronize(); t = time.perf_counter() - t
    fb = pb.phase_ms("flow_fallbacks")
    costs.append(info[1])
    print(f"run {rep}: rc {rc} iterations {int(info[5])} stop {int(info[6])} systems {int(info[9])} final cost {info[1]!r} fallbacks {fb} wall {t:.3f} s", flush=True)
    pb.close()
print("bit-identical final cost across runs:", len(set(costs)) == 1)
