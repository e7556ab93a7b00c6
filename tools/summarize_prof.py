#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to the library's own kernels (plus a total row for
everything else) so that the summary committed under profiles/ stays readable."""
import csv
import sys


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    keep, other_ns, other_calls = [], 0, 0
    for r in rows:
        name = r["Name"]
        if "anonymous namespace" in name and "at::native" not in name:
            short = name.split("::")[1].split("(")[0]
            keep.append([short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                         r["MaxNs"], r["StdDev"]])
        else:
            other_ns += int(r["TotalDurationNs"])
            other_calls += int(r["Calls"])
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerows(keep)
        w.writerow(["(all torch / runtime kernels)", other_calls, other_ns, "", "", "", "", ""])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
