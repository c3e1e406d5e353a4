#!/usr/bin/env python3
"""GPU bring-up check of the tile-dataflow Cholesky: accuracy vs numpy at several orders, device time per solve, for the
   default (one launch), chunked (a launch per tile column) and legacy (three-stream) paths.  Each case runs in its own process
   under a timeout so that a stuck kernel costs seconds, not the box."""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child(n, reps):
    import bundler_sfm_amd.sfm as B
    rng = np.random.default_rng(n)
    F = rng.standard_normal((n, 48))
    d = np.exp(rng.uniform(np.log(1e-1), np.log(1e1), n))
    A = (F * np.linspace(0.5, 2.0, 48)) @ F.T
    A[np.diag_indices(n)] += d
    b = rng.standard_normal(n)
    os.environ["BSFM_CHOL_REPS"] = str(reps)
    t0 = time.time()
    rc, x = B.dense_chol_solve(A, b)
    dt = time.time() - t0
    r = A @ x - b
    ref = np.linalg.solve(A, b) if n <= 4000 else None
    out = {"n": n, "rc": int(rc), "backward": float(np.abs(r).max() / (np.abs(A).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())),
           "forward": float(np.abs(x - ref).max() / np.abs(ref).max()) if ref is not None else None, "wall_s": round(dt, 2),
           "finite": bool(np.isfinite(x).all())}
    print("RESULT " + json.dumps(out), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(int(sys.argv[2]), int(sys.argv[3])); sys.exit(0)

cases = []
sizes = [int(v) for v in os.environ.get("SIZES", "200,256,384,1024,1799,3840,9000").split(",")]
modes = os.environ.get("MODES", "flow,chunk1,streams").split(",")
for n in sizes:
    for mode in modes:
        env = dict(os.environ)
        env.pop("BSFM_CHOL", None); env.pop("BSFM_FLOW_CHUNK", None)
        if mode == "streams": env["BSFM_CHOL"] = "streams"
        if mode.startswith("chunk"): env["BSFM_FLOW_CHUNK"] = mode[5:]
        if mode == "flow" and n == sizes[-1] and os.environ.get("TRACE_OUT"):
            env["BSFM_FLOW_TRACE"] = "1"; env["BSFM_FLOW_TRACE_FILE"] = os.environ["TRACE_OUT"]
        reps = 1 if (mode == "flow" and env.get("BSFM_FLOW_TRACE")) else 4
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n), str(reps)], env=env, capture_output=True, text=True,
                               timeout=int(os.environ.get("CASE_TIMEOUT", "180")))
            res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            times = [l.split(": ")[-1] for l in p.stderr.splitlines() if "dense_chol_solve n =" in l]
            err = [l for l in p.stderr.splitlines() if "dense_chol_solve n =" not in l][-6:]
            print(f"n={n} mode={mode} exit={p.returncode} {res[0] if res else 'NO RESULT'} times={times} {' | '.join(err) if (p.returncode or not res) else ''}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"n={n} mode={mode} TIMEOUT after {time.time()-t0:.0f}s", flush=True)
