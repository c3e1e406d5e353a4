#!/usr/bin/env python
"""CPU baseline AT THE HEADLINE CONFIG: the reference's own sba_motstr_levmar (oracle/_ref/libsfmref_timings.so = lib/sba-1.5 compiled
-DTIMINGS, lib/sba-1.5/sba_levmar.c:49-53, with the vendored CLAPACK; gcc -O3; ONE thread -- the reference has no threading) on the
synthetic 1 000 cameras / 500 000 points / 5 000 000 observations scene of BASELINE.json configs[2], forward-difference Jacobian
(run_sfm's own mode), itmax = 3.  Minutes per iteration, so bench.py cannot hold it inside its default run: this script is run
once per round ON THE GPU BOX'S HOST (gpurun) and its JSON is committed as profiles/rNN_cpu_baseline_cfg3.json, which bench.py
reports as `cpu_baseline` (marked "cached") next to a small live sample.

    python scripts/cpu_baseline_cfg3.py [out.json] [itmax] [cams points]
"""
import ctypes as C
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run(m=1000, n=500000, itmax=3):
    """The reference's sba_motstr_levmar (-DTIMINGS build) on the seeded scene; returns the record (phases from its own prints)."""
    import oracle_util as O
    import bundler_sfm_amd as B            # host-only helpers: seeded scene generator, dense vmask
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libsfmref_timings.so")
    if not os.path.exists(lib_path):
        raise RuntimeError(lib_path + " missing (make -C oracle ref, build container)")
    lib = C.CDLL(lib_path)
    s = B.synth_ba(m, n, 10)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    cams = O.copy_cams(s["cams"]); pts = np.array(s["pts"], np.float64, copy=True)
    info = np.zeros(10); p = np.zeros(m * 9 + 3 * n); secs = C.c_double()
    dp = C.POINTER(C.c_double)
    fn = lib.ref_sba_motstr
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, dp, C.c_int, C.c_int, C.c_int, C.POINTER(O.CameraParams), dp, C.c_int, C.c_int,
                   dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp]
    # the -DTIMINGS prints go to stdout: capture fd 1 in a file
    tmp = tempfile.NamedTemporaryFile(prefix="sba_timings_", suffix=".txt", delete=False)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(tmp.fileno(), 1)
    t0 = time.time()
    try:
        rc = fn(n, m, 0, vm.ctypes.data_as(C.c_char_p), O._d(s["proj"]), 1, 1, 1, cams, O._d(pts), 0, 0, None, 0.0, 1e-12, itmax, 0, 0,
                O._d(info), O._d(p), None, None, None, None, C.byref(secs))
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved, 1); os.close(saved)
    wall = time.time() - t0
    text = open(tmp.name).read()
    os.unlink(tmp.name)
    phases = {}
    for name, val in re.findall(r"\[sba_motstr_levmar_x\] (.+?) took ([0-9.]+)s", text):
        phases.setdefault(name, []).append(float(val))
    its = max(int(info[5]), 1)
    return {
        "what": "reference sba_motstr_levmar (lib/sba-1.5, -DTIMINGS, vendored CLAPACK, gcc -O3), FD Jacobian, 1 thread",
        "config": {"cameras": m, "points": n, "observations": int(s["rowptr"][-1]), "itmax": itmax},
        "host_cpu": cpu_model(), "host_cpus": os.cpu_count(), "cores_used": 1,
        "rc": rc, "iterations": int(info[5]), "stop": int(info[6]), "linear_systems": int(info[9]),
        "initial_cost": info[0], "final_cost": info[1],
        "wall_s": round(wall, 2), "sba_s": round(secs.value, 2),
        "ms_per_iteration": round(1e3 * secs.value / its, 1), "iterations_per_s": round(its / secs.value, 6),
        "phases_s_per_call": {k: [round(v, 3) for v in vals] for k, vals in phases.items()},
        "phases_s_mean": {k: round(sum(vals) / len(vals), 3) for k, vals in phases.items()},
    }


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "cpu_baseline_cfg3.json")
    itmax = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    m = int(sys.argv[3]) if len(sys.argv) > 4 else 1000
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
    try:
        out = run(m, n, itmax)
    except RuntimeError as exc:
        sys.exit(str(exc))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
