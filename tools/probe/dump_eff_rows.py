#!/usr/bin/env python3
"""Dumps the un-jammed adjacency rows of BASELINE config 3 (8 192 envs, N = 50, M = 8) at a few steps of an episode
to gpurun_out/eff_rows.npy ([steps][E][N] uint64), for offline statistics of the eigenproblems the solve launch gets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from marl_amd import BatchedJammerEnv, EnvConfig

E, N, M = 8192, 50, 8
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1234, debug_bits=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
keep = []
for t in range(120):
    act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(act)
    if t % 20 == 19:
        eff = (env.adj_bits & ~env.jam_bits).cpu().numpy().view(np.uint32).astype(np.uint64)     # [E][N][2] words
        keep.append(eff[..., 0] | (eff[..., 1] << np.uint64(32)))
        print(t, float(env.lambda2.mean()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/eff_rows.npy", np.stack(keep))
