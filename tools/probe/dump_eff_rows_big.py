#!/usr/bin/env python3
"""The same as dump_eff_rows.py for BASELINE config 4's swarm (N = 200, M = 16; 1 024 envs): gpurun_out/eff_rows_big.npy,
[steps][E][N][W] uint32 words."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from marl_amd import BatchedJammerEnv, EnvConfig

E, N, M = 1024, 200, 16
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1234, debug_bits=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
keep = []
for t in range(60):
    act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(act)
    if t % 20 == 19:
        keep.append((env.adj_bits & ~env.jam_bits).cpu().numpy().view(np.uint32))
        print(t, float(env.lambda2.mean()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/eff_rows_big.npy", np.stack(keep))
