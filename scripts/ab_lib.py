import sys, shutil, subprocess, os
# A/B different builds of the library on the chol9000 configuration (fresh process per variant)
for lib in sys.argv[1:]:
    shutil.copy(lib, "bundler_sfm_amd/libbsfm_hip.so")
    out = subprocess.run([sys.executable, "scripts/gpu_check3.py"], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "solve=" in l][-1]
    print(os.path.basename(lib), line.strip())
