"""Drop-in boundary, compiled for real: oracle/_ref/run_sfm_link_test is a C translation unit that
#includes the REFERENCE'S lib/sfm-driver/sfm.h and was linked with -lbsfm_hip ahead of the reference's sfm.c built with
-Drun_sfm=run_sfm_cpu_reference (oracle/Makefile, oracle/run_sfm_link_test.c).  `run_sfm` therefore resolves to the GPU
library, sfm_project_final / run_sfm_cpu_reference to the reference's objects, in one process."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "run_sfm_link_test")


def test_link_test_binary_resolves_run_sfm_to_the_gpu_library():
    """CPU-side: the binary exists (built where /root/reference is) and its run_sfm is an UNDEFINED symbol satisfied by
    libbsfm_hip.so, while run_sfm_cpu_reference and sfm_project_final are defined inside it (the reference's objects)."""
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/run_sfm_link_test not built (needs /root/reference)")
    nm = subprocess.run(["nm", "-D", "--undefined-only", EXE], capture_output=True, text=True).stdout
    assert re.search(r"\bU run_sfm\b", nm)
    defined = subprocess.run(["nm", EXE], capture_output=True, text=True).stdout
    assert re.search(r"\bT run_sfm_cpu_reference\b", defined) and re.search(r"\bT sfm_project_final\b", defined)
    assert not re.search(r"\bT run_sfm\b", defined)
    ldd = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "libbsfm_hip.so" in ldd and "bundler_sfm_amd" in ldd


@pytest.mark.gpu
def test_run_sfm_called_from_c_through_the_reference_header():
    assert os.path.exists(EXE), "oracle/_ref/run_sfm_link_test missing (make -C oracle ref in the build container)"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    gpu = out[out.index("== gpu") + 6:out.index("== cpu")].strip().splitlines()
    cpu = out[out.index("== cpu") + 6:out.index("== end")]
    # the two summary lines of run_sfm (lib/sfm-driver/sfm.c:872-873), nothing else, in this order and format
    assert len(gpu) == 2, gpu
    m0 = re.fullmatch(r"\[run_sfm\] Number of iterations: (\d+)", gpu[0])
    m1 = re.fullmatch(r"info\[6\] = (\d+\.\d{3})", gpu[1])
    assert m0 and m1, gpu
    c0 = re.search(r"\[run_sfm\] Number of iterations: (\d+)", cpu)
    c1 = re.search(r"info\[6\] = (\d+\.\d{3})", cpu)
    assert c0 and c1
    assert abs(int(m0.group(1)) - int(c0.group(1))) <= 1          # stop rule 4 (eps4 = 0) may differ by one iteration, DESIGN.md section 6
    vals = {k: float(v) for k, v in re.findall(r"^(before|after_gpu|after_cpu|max_rel_focal_diff): ([0-9.eE+-]+)$", out, re.M)}
    assert vals["after_gpu"] < 0.05 * vals["before"] and vals["after_cpu"] < 0.05 * vals["before"]
    # both runs take 24 iterations and stop on rule 8 (4 % rule) short of the minimum, so the last iterate carries the rounding history of
    # the whole trajectory: measured 5.3e-7 with the 16 x 16 x 4 Schur tiles of rounds 3-4, 1.2e-6 with the 4 x 4 x 4 blocks of round 5
    # (another summation order of the same triples); the cameras agree to 1e-7 either way (max_rel_focal_diff below)
    assert abs(vals["after_gpu"] - vals["after_cpu"]) <= 5e-6 * vals["after_cpu"]
    assert vals["max_rel_focal_diff"] <= 1e-5
    assert "scales: 1 1" in out                                   # f_scale / k_scale reset on exit (sfm.c:918-921)


@pytest.mark.gpu
def test_run_sfm_boundary_at_a_fixed_iteration_index(gpu_bsfm):
    """Beside the converged comparison above (whose last iterate carries the rounding history of 24 iterations and a stop rule that may fire one
    iteration apart): the same boundary cut at a FIXED iteration count -- bsfm_run_sfm_ex with itmax = 12 against the reference's
    sba_motstr_levmar (oracle/_ref, run_sfm's own packing and options, sfm.c:649-811) with itmax = 12 -- counters equal, cost to 1e-9,
    points and focal lengths to 1e-7 (a bar that moves with the summation order pins nothing)."""
    import numpy as np
    import oracle_util as O
    B = gpu_bsfm
    m, n = 12, 300
    s = B.synth_ba(m, n, 6)
    vm = B.dense_vmask(n, m, s["rowptr"], s["colidx"])
    ref = O.ref_sba(n, m, vm, s["proj"], s["cams"], s["pts"], itmax=12, jac_mode=0)
    cams = B.copy_cameras(s["cams"]); pts = s["pts"].copy()
    rc, info = B.run_sfm(n, m, 0, vm, s["proj"], 1, 0, 1, 1, cams, pts, eps2=1e-12, options=B.default_options(verbose=0, itmax=12))
    assert rc == 12 and list(info[5:10]) == list(ref["info"][5:10])
    assert abs(info[1] - ref["info"][1]) <= 1e-9 * ref["info"][1]
    rp = ref["p"]
    assert np.abs(pts - rp[9 * m:]).max() <= 1e-7 * np.abs(rp[9 * m:]).max()
    f_ref = np.array([rp[9 * j + 6] / 0.001 for j in range(m)])
    f_gpu = np.array([cams[j].f for j in range(m)])
    assert np.abs(f_gpu - f_ref).max() <= 1e-7 * np.abs(f_ref).max()
