#!/usr/bin/env python3
"""Prints VGPRs / AGPRs / spills / scratch / static LDS of every kernel in an object of bundler_sfm_amd/csrc/_build whose name contains one of the
given substrings:  python scripts/kernel_resources.py solver.o k_schur k_chol_flow"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "bundler_sfm_amd", "csrc", "_build", sys.argv[1])
subs = sys.argv[2:]
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, stderr=subprocess.DEVNULL)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
for block in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    if subs and not any(s in name for s in subs):
        continue
    get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
    agpr = int(re.match(r"\s*(\d+)", block).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{dem[:110]:110s} vgpr {get('vgpr_count'):3d} agpr {agpr:3d} sgpr {get('sgpr_count'):3d} spill {get('vgpr_spill_count'):3d} scratch {get('private_segment_fixed_size'):5d} lds {get('group_segment_fixed_size'):6d}")
