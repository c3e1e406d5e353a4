#!/usr/bin/env python3
"""Device time of the DISTRIBUTED dataflow Cholesky between `world` processes sharing cuda:0 (ipc transport), next to one process with the same total
and with the per-rank number of workgroups: what the protocol (peer-mapped panel tiles, remote counters, filtered task lists) costs when the hardware
is the same.  usage: python scripts/r6/dist_chol_timing.py [n=9000] [world=2]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
worker = f"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, {ROOT!r})
import bundler_sfm_amd as B, bundler_sfm_amd.sfm as S
n = {n}
rng = np.random.default_rng(5)
Z = np.zeros((n, n))
for g in range((n + 89) // 90):
    lo, hi = 90 * g, min(n, 90 * (g + 1))
    G = rng.standard_normal((hi - lo, hi - lo + 8)); Z[lo:hi, lo:hi] = G @ G.T + (hi - lo) * np.eye(hi - lo)
b = rng.standard_normal(n)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    B.lib.bsfm_comm_create_from_env.restype = C.c_void_p
    c = C.c_void_p(B.lib.bsfm_comm_create_from_env())
    rc, x = S.dense_chol_solve_dist(c, Z, b)
    B.lib.bsfm_comm_destroy(c)
else:
    rc, x = S.dense_chol_solve(Z, b)
print("rc", rc, "residual", np.abs(Z @ x - b).max())
"""
def run(envs, nproc):
    tmp = tempfile.mkdtemp()
    procs = []
    for rank in reversed(range(nproc)):
        env = dict(os.environ, BSFM_CHOL_REPS="5", HSA_ENABLE_IPC_MODE_LEGACY="0", **envs)
        if nproc > 1:
            env.update(RANK=str(rank), WORLD_SIZE=str(nproc), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29871", BSFM_COMM_TRANSPORT="ipc",
                       BSFM_COMM_ID_FILE=os.path.join(tmp, "job.id"), BSFM_COMM_TIMEOUT_S="120")
        procs.append(subprocess.Popen([sys.executable, "-c", worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for k, (o, e) in enumerate(outs):
        times = [l.split(": ")[-1] for l in e.splitlines() if "dense_chol_solve n =" in l]
        other = [l for l in e.splitlines() if "dense_chol_solve n =" not in l][-3:]
        print(f"   process {nproc - 1 - k}: {o.strip()}  times {times} {' | '.join(other)}")
print(f"n = {n}: one process, 512 workgroups"); run({}, 1)
print(f"n = {n}: one process, {512 // world} workgroups (what one rank of the distributed run launches)"); run({"BSFM_FLOW_WGS": str(512 // world)}, 1)
print(f"n = {n}: {world} processes x {512 // world} workgroups, distributed (tile column j on rank j mod {world})"); run({"BSFM_FLOW_WGS": str(512 // world)}, world)
