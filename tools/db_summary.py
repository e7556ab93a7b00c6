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


STEP_KERNELS = ("env_step_front_kernel", "solve_order_kernel", "env_step_solve_kernel", "env_step_solve_prio_kernel",
                "env_step_solve_lanczos_kernel", "env_step_big_kernel", "env_step_lanczos_rows_kernel", "env_step_team_kernel")
SOLVE_KERNELS = ("env_step_solve_kernel", "env_step_solve_prio_kernel", "env_step_solve_lanczos_kernel",
                 "env_step_lanczos_rows_kernel", "env_step_team_kernel")


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
        # one step = one launch of the kernel that holds the solves (the front kernel itself in the one-launch schedule)
        steps = max((len(v) for k, v in disp.items() if k.startswith(SOLVE_KERNELS)), default=0) or \
            max((len(v) for k, v in disp.items() if k.startswith("env_step_front_kernel")), default=1)
        for cname, byk in per.items():
            res["per_step"][cname] = sum(byk.values()) / steps
            res["per_kernel_launch"][cname] = {k: v / len(disp[k]) for k, v in byk.items()}
        res["steps_sampled"] = steps
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


def summary(out, stats_csv, pmc_json=None):
    """What bench.py reads back (profiles/summary_c<n>.json): average launch duration per kernel, the duration
    of the eigen-solve launch, and from the counter passes HBM bytes per step (FETCH_SIZE doubled as prescribed
    for gfx950, + WRITE_SIZE; both are reported in KiB) and VALU busy per kernel
    (SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32): quad-cycles over the busy cycles of 32 SIMDs per counter instance)."""
    import os
    res = {"git_head": os.environ.get("PROFILE_GIT_HEAD", "unknown"), "command": os.environ.get("PROFILE_COMMAND", ""),
           "kernel_avg_us": {}}
    for row in csv.DictReader(open(stats_csv)):
        if row["AverageNs"]:
            res["kernel_avg_us"][row["Name"]] = float(row["AverageNs"]) / 1e3
    solve = [v for k, v in res["kernel_avg_us"].items() if k.startswith(SOLVE_KERNELS)]
    if solve:
        res["solve_kernel_us"] = sum(solve)        # N > 64: the Lanczos launch and the team launch after it
    if pmc_json:
        p = json.load(open(pmc_json))
        ps = p["per_step"]
        if "FETCH_SIZE" in ps and "WRITE_SIZE" in ps:
            res["hbm_bytes_per_step"] = (2.0 * ps["FETCH_SIZE"] + ps["WRITE_SIZE"]) * 1024.0
        pk = p["per_kernel_launch"]
        if "SQ_ACTIVE_INST_VALU" in pk and "SQ_BUSY_CYCLES" in pk:
            res["valu_busy_pct"] = {k: round(100.0 * pk["SQ_ACTIVE_INST_VALU"][k] * 4 / (pk["SQ_BUSY_CYCLES"][k] * 32), 1)
                                    for k in pk["SQ_ACTIVE_INST_VALU"] if pk["SQ_BUSY_CYCLES"].get(k)}
        res["steps_sampled"] = p.get("steps_sampled")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    {"stats": lambda: stats(sys.argv[2], sys.argv[3]), "pmc": lambda: pmc(sys.argv[2], sys.argv[3:]),
     "summary": lambda: summary(*sys.argv[2:5])}[sys.argv[1]]()
