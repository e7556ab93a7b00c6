#!/usr/bin/env python3
"""Summaries of rocprofv3's sqlite output (the default format of ROCm 7.2).

  db_summary.py stats  out.csv  run.db             per-kernel launch statistics of the library's kernels
  db_summary.py pmc    out.json run1.db run2.db …  per-launch counter means of the step kernel
"""
import csv, json, sqlite3, statistics, sys
from collections import defaultdict


def short(name):
    if "anonymous namespace" in name and "at::native" not in name:
        return name.split("::")[1].split("(")[0]
    return None


def stats(out, dbp):
    db = sqlite3.connect(dbp)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else "kernel_name"
    per = defaultdict(list)
    other = [0, 0]
    for name, dur in db.execute(f"select {namec}, duration from kernels"):
        s = short(name)
        if s:
            per[s].append(dur)
        else:
            other[0] += 1
            other[1] += dur
    tot = sum(sum(v) for v in per.values()) + other[1]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100 * sum(v) / tot, 2), min(v), max(v),
                        round(statistics.pstdev(v), 1)])
        w.writerow(["(all torch / runtime kernels)", other[0], other[1], "", round(100 * other[1] / tot, 2), "", "", ""])
    print(open(out).read())


STEP_KERNELS = ("env_step_front_kernel", "solve_order_kernel", "env_step_solve_kernel", "env_step_big_kernel")


def pmc(out, dbs):
    """Counter totals per step (= one marl_env_step call: front + order + solve launches), and per kernel."""
    res = {"per_step": {}, "per_kernel_launch": {}}
    for dbp in dbs:
        db = sqlite3.connect(dbp)
        rows = list(db.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"))
        per = defaultdict(lambda: defaultdict(float))      # counter -> kernel -> total (instances summed)
        disp = defaultdict(set)
        for name, cname, did, val in rows:
            k = short(name)
            if k is None or not k.startswith(STEP_KERNELS):
                continue
            per[cname][k] += val
            disp[k].add(did)
        steps = max((len(v) for k, v in disp.items() if k.startswith(("env_step_solve_kernel", "env_step_big_kernel"))),
                    default=0)
        for cname, byk in per.items():
            res["per_step"][cname] = sum(byk.values()) / steps
            res["per_kernel_launch"][cname] = {k: v / len(disp[k]) for k, v in byk.items()}
        res["steps_sampled"] = steps
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    {"stats": lambda: stats(sys.argv[2], sys.argv[3]), "pmc": lambda: pmc(sys.argv[2], sys.argv[3:])}[sys.argv[1]]()
