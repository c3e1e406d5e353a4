#!/usr/bin/env python3
"""VGPRs and scratch of every FUNCTION of csrc/solver.hip whose name contains one of the given substrings (default: flow_) -- the non-inlined
roles of the 128-VGPR build of the dataflow Cholesky are functions of their own, and the code-object notes that scripts/kernel_resources.py
reads only know whole kernels.  Source: the `.set <fn>.num_vgpr / .private_seg_size` symbols of the device assembly.
  python scripts/role_resources.py [substr ...]        (exit code 1 if the POTRF factor role needs scratch: its A1 is the chain's critical path)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subs = sys.argv[1:] or ["flow_"]
asm = "/tmp/bsfm_solver_dev.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-x", "hip", "-S",
                os.path.join(ROOT, "bundler_sfm_amd", "csrc", "solver.hip"), "-o", asm, "--cuda-device-only"], check=True, stderr=subprocess.DEVNULL)
vg, sc = {}, {}
for l in open(asm):
    m = re.match(r"\s*\.set \.?L?(_Z\w+)\.(num_vgpr|private_seg_size), (\d+)", l)
    if m:
        (vg if m.group(2) == "num_vgpr" else sc)[m.group(1)] = int(m.group(3))
bad = 0
for name in sorted(vg):
    if not any(s in name for s in subs):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:110]:110s} VGPRs {vg[name]:3d}  scratch {sc.get(name, 0):4d} B/lane")
    if "flow_potrf_factor" in name and sc.get(name, 0) != 0:
        bad = 1
sys.exit(bad)
