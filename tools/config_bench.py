#!/usr/bin/env python3
"""Throughput of the full step at the BASELINE config sizes (one GPU): env-steps/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

for name, E, N, M, steps in (("C2 4096 x N=10 M=4", 4096, 10, 4, 300), ("C3 8192 x N=50 M=8", 8192, 50, 8, 200),
                             ("C3x8 65536 x N=50 M=8", 65536, 50, 8, 50), ("C4 8192 x N=200 M=16", 8192, 200, 16, 12),
                             ("C5 16384 x N=50 M=8", 16384, 50, 8, 100)):
    env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1, auto_reset=True)
    env.reset()
    pool = torch.empty(16, E, M, 3, device="cuda").uniform_(-5, 5)
    pool[..., 2] = torch.randint(0, 4, (16, E, M), device="cuda").float()
    for i in range(5):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:26s} {dt / steps * 1e3:9.3f} ms/step  {E * steps / dt:12.0f} env-steps/s", flush=True)
    del env
