"""bench.py's accounting tables, checked on the CPU against the committed profiler summaries (VERDICT r5 "bench hygiene"): ONE algorithmic byte
count per kernel (bench.algorithmic_bytes, quoted by DESIGN.md section 4), a phase made of two kernels compared with the SUM of their counters,
and no phase whose counted traffic is less than what its algorithm must move (beyond what the 32 MB of L2 can hold of the previous kernel's
output)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_streaming_phase_is_compared_with_all_of_its_kernels():
    alg = bench.algorithmic_bytes(5.03e6, 5e5, 9, 9000)
    assert alg["backsub"]["kernels"] == ["k_backsub_obs", "k_backsub"]            # the two-pass back-substitution is two kernels
    assert alg["backsub"]["bytes"] > alg["backsub"]["one_pass_bound_bytes"]       # and moves more than a perfect one-pass gather would
    for ph, ent in alg.items():
        assert ent["bytes"] > 0 and ent["kernels"], ph


def test_counted_traffic_is_not_below_the_algorithmic_bytes():
    """With the committed PMC summary of the headline command (profiles/LATEST): counter / algorithmic >= 0.93 for every phase -- a ratio
    below 1 by more than the L2's 32 MB means the algorithmic figure forgot a stream (round 5 printed 0.64 for the back-substitution)."""
    d, name = bench.pmc_summary()
    assert d is not None, "no committed PMC summary"
    alg = bench.algorithmic_bytes(5029998, 500000, 9, 9000)
    seen = 0
    for ph, ent in alg.items():
        ctrs = [bench.pmc_traffic(k) for k in ent["kernels"]]
        if not all(ctrs):
            continue
        if ph == "schur_prep" and bench._kernel_entry(d, "k_zero_lower_tiles"):
            continue      # (a summary from before round 6: the tile clearing was a kernel of its own)
        seen += 1
        ratio = sum(ctrs) / ent["bytes"]
        floor = 1.0 - 32e6 * len(ent["kernels"]) / ent["bytes"] - 0.02
        assert ratio >= floor, (ph, name, round(ratio, 3), round(floor, 3))
        assert ratio <= 1.7, (ph, name, round(ratio, 3))
    assert seen >= 4


def test_latest_pointer_names_existing_files():
    ptr = json.load(open(os.path.join(ROOT, "profiles", "LATEST")))
    for key in ("pmc_traffic", "kernel_stats"):
        assert os.path.exists(os.path.join(ROOT, "profiles", ptr[key])), ptr[key]
