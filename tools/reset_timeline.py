#!/usr/bin/env python3
"""Kernel timeline around the auto-reset of a batch stepped as parts on streams of their own.
  rocprofv3 --kernel-trace -d DIR -o r -- python3 tools/reset_timeline.py run     (steps 3 parts across one auto-reset)
  python3 tools/reset_timeline.py show DIR/.../r_results.db                      (prints the launches around the reset)"""
import os, sys
if sys.argv[1] == "run":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from marl_amd import BatchedJammerEnv, EnvConfig
    E, M = 8192, 8
    dev = torch.device("cuda", 0)
    pool = torch.empty(16, E, M, 3, device=dev).uniform_(-5, 5)
    pool[..., 2] = torch.randint(0, 4, (16, E, M), device=dev).float()
    shards = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=50, M=M), seed=1, device=dev, num_shards=shards)
    env.reset()
    for i in range(190):
        env.step(pool[i % 16], episode_info=False)
    torch.cuda.synchronize()
    env.fork()
    for i in range(20):
        env.step(pool[i % 16], episode_info=False)
    env.join()
    torch.cuda.synchronize()
else:
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    print(cols)
    namec = "name" if "name" in cols else "kernel_name"
    qc = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    sc = "stream_id" if "stream_id" in cols else None
    sel = f"select {namec}, start, end" + (f", {qc}" if qc else ", 0") + (f", {sc}" if sc else ", 0") + " from kernels order by start"
    rows = [(n.split("::")[1].split("(")[0][:28] if "anonymous namespace" in n else n[:28], s, e, q, st) for n, s, e, q, st in db.execute(sel)]
    idx = [i for i, r in enumerate(rows) if r[0].startswith("env_reset_wave_kernel")]
    first_masked = idx[1] if len(idx) > 1 else idx[0]          # idx[0] is the reset of the whole batch at the start
    t0 = rows[max(0, first_masked - 30)][1]
    for r in rows[max(0, first_masked - 30):first_masked + 45]:
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} us  {(r[2]-r[1])/1e3:7.1f}  q{r[3]} s{r[4]}  {r[0]}")
