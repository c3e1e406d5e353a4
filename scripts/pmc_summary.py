"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
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
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k in ("k_syrk_update", "k_schur_tasks_v2<9>", "k_jacobian<9, true>"):
        if k in out["kernels"]:
            print(k, out["kernels"][k])


if __name__ == "__main__":
    main()
