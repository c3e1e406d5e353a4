"""Concise per-kernel table from a rocprofv3 *_kernel_stats.csv: name, calls, average us, total ms."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    n = r["Name"].split("(")[0].replace("bsfm::", "").replace("void ", "")
    print("%-34s calls %5s  avg %9.1f us  total %8.2f ms" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
