#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-launch mean of each counter for the
step kernel.  Usage: pmc_summary.py out.json dir1 dir2 ..."""
import csv, glob, json, sys
from collections import defaultdict


def main(out, dirs):
    acc = defaultdict(list)
    for d in dirs:
        for f in glob.glob(f"{d}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "env_step" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: sum(v) / len(v) for k, v in acc.items()}
    res["launches_sampled"] = {k: len(v) for k, v in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
