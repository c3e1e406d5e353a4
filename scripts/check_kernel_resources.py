#!/usr/bin/env python3
"""Build-time guard: kernels whose correctness or speed depends on their register allocation must not spill.

  * k_match_bound / k_match_l2 (match_l2.hip) issue their fragment loads with inline-asm global_load and hand-counted
    s_waitcnt: a spilled, copied or re-allocated prefetch register would silently produce wrong matches.  No scratch, no
    VGPR spills, and the register budget of __launch_bounds__(256, 2) respected.
  * k_chol_flow (chol_flow.hip.h) must keep its 128-VGPR budget (two 512-thread workgroups per CU) and must not spill VGPRs in
    its own body; its scratch is the callee-save area of the role functions only (bounded).

Reads the code-object metadata of bundler_sfm_amd/csrc/_build/*.o (llvm-objcopy .hip_fatbin -> clang-offload-bundler ->
llvm-readelf --notes).  Exit status 1 with a message when a rule is broken; __graft_entry__.build() runs it after make."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
RULES = {   # kernel-name substring -> (object, max VGPRs, max VGPR spills, max scratch bytes per lane)
    "k_match_bound": ("match_l2.o", 256, 0, 0),
    "k_match_l2": ("match_l2.o", 256, 0, 0),
    "k_chol_flowILi4": ("solver.o", 128, 48, 336),      # throughput build: two workgroups per CU; its scratch is the kernel body's own state around the role
                                                        # calls (44 VGPRs, 288 bytes at the end of round 5) -- the roles themselves, the POTRF factor
                                                        # wave's included, use none (scripts/role_resources.py, profiles/r05_flow_role_resources.txt)
    "k_chol_flowILi2": ("solver.o", 256, 0, 0),         # latency build (one workgroup per CU): nothing in scratch at all
    "k_flow_solve_one": ("solver.o", 128, 0, 0),        # one-tile systems: the same tile factorisation, one launch
}


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, stderr=subprocess.DEVNULL)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
        out[name.group(1)] = dict(vgpr=get("vgpr_count"), spill=get("vgpr_spill_count"), scratch=get("private_segment_fixed_size"))
    return out


def main():
    build = os.path.join(ROOT, "bundler_sfm_amd", "csrc", "_build")
    cache, bad, seen = {}, [], set()
    for sub, (obj, max_v, max_spill, max_scr) in RULES.items():
        path = os.path.join(build, obj)
        if not os.path.exists(path):
            bad.append(f"{path} missing (run make first)"); continue
        ks = cache.setdefault(obj, kernels_of(path))
        for name, k in ks.items():
            if sub not in name:
                continue
            seen.add(sub)
            if k["vgpr"] > max_v or k["spill"] > max_spill or k["scratch"] > max_scr:
                bad.append(f"{name}: {k['vgpr']} VGPRs (max {max_v}), {k['spill']} VGPR spills (max {max_spill}), "
                           f"{k['scratch']} bytes of scratch per lane (max {max_scr})")
    for sub in RULES:
        if sub not in seen:
            bad.append(f"no kernel matching '{sub}' found")
    if bad:
        sys.stderr.write("kernel resource check FAILED:\n  " + "\n  ".join(bad) + "\n")
        return 1
    print("kernel resource check ok:", ", ".join(sorted(seen)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
