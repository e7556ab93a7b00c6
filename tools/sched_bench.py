#!/usr/bin/env python3
"""Step throughput per schedule (N=50, M=8): env-steps/s, and equality of the outputs across schedules."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

scheds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["split", "fused"]
for E in (8192, 16384, 65536):
    pool = torch.empty(16, E, 8, 3, device="cuda").uniform_(-5, 5)
    pool[..., 2] = torch.randint(0, 4, (16, E, 8), device="cuda").float()
    ref = None
    for sched in scheds:
        env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=50, M=8, step_schedule=sched), seed=1, auto_reset=True)
        env.reset()
        for i in range(30):
            env.step(pool[i % 16])
        torch.cuda.synchronize()
        steps = 300 if E <= 16384 else 80
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(pool[i % 16])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        chk = (float(env.reward.double().sum()), float(env.lambda2.sum()), float(env.views["enemy_x"].sum()))
        same = "" if ref is None else ("  == first" if chk == ref else "  DIFFERS from first")
        ref = ref or chk
        print(f"E={E:6d} {sched:7s} {dt / steps * 1e6:9.1f} us/step  {E * steps / dt:12.0f} env-steps/s{same}", flush=True)
        del env
