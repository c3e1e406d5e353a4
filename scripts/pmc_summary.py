"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [more counter csvs ...]
Every counter found in the extra CSVs (e.g. SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE, MfmaUtil) is
reported per kernel as the mean per launch under "counters".
Corrections, exactly as the guide's HBM section prescribes for gfx950: both counters are in KB (x1024); FETCH_SIZE
tallies 128-byte requests at 64 B for wide coalesced reads, so fetched bytes = 2 x FETCH_SIZE (the kernels reported here
read with 16-byte-per-lane loads); WRITE_SIZE is taken as is (uncalibrated per the guide)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("bsfm::", "").replace("void ", "")
        key = (r.get("Dispatch_Id"), name)
        a = acc[name]
        if key not in seen:
            seen.add(key); a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def all_counters(path, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [set(), 0.0]))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("bsfm::", "").replace("void ", "")
        a = acc[name][r["Counter_Name"]]
        a[0].add(r.get("Dispatch_Id")); a[1] += float(r["Counter_Value"])
    for name, cs in acc.items():
        k = out["kernels"].setdefault(name, {})
        for cn, (ids, tot) in cs.items():
            k.setdefault("counters", {})[cn] = tot / max(len(ids), 1)


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch", "correction": "fetch = 2 x FETCH_SIZE x 1024, write = WRITE_SIZE x 1024", "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(name, (0, 0.0)); nw, vw = write.get(name, (0, 0.0))
        f = 2.0 * 1024.0 * vf / nf if nf else None
        w = 1024.0 * vw / nw if nw else None
        out["kernels"][name] = {"launches": max(nf, nw), "fetch_bytes": f, "write_bytes": w,
                                "traffic_bytes": (f or 0.0) + (w or 0.0)}
    for extra in sys.argv[4:]:
        try:
            all_counters(extra, out)
        except Exception as ex:       # a pass that produced nothing must not lose the others
            print("skipped", extra, ex)
    syrk = [v for k, v in out["kernels"].items() if k.startswith("k_syrk_update")]
    c = (max(syrk, key=lambda v: v.get("launches", 0)) if syrk else {}).get("counters", {})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"] > 0:
        # both are summed over the SIMDs/CUs of the device: busy matrix-pipe cycles per busy CU cycle (4 SIMDs per CU)
        out["k_syrk_update_mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        if k.startswith(("k_syrk_update", "k_schur_tasks", "k_jacobian")):
            print(k, v)


if __name__ == "__main__":
    main()
