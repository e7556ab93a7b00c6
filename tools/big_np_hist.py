#!/usr/bin/env python3
"""Sizes n' of the eigenproblems the team solver gets at N = 200, M = 16 (BASELINE config 4): histogram over a few steps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marl_amd import BatchedJammerEnv, EnvConfig

E, N, M = 8192, 200, 16
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1234)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
ld = (N + 7) // 8 * 8
hist = np.zeros(N + 1, dtype=np.int64)
for t in range(12):
    act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(act)
    torch.cuda.synchronize()
    base = env.state.workspace + E * ld * ld * 8                  # the pending counts follow the matrices
    off = base - env._buf.data_ptr()
    pend = env._buf[off:off + 4 * E].view(torch.int32).cpu().numpy()
    hist += np.bincount(np.clip(pend, 0, N), minlength=N + 1)
tot = hist.sum()
print("envs x steps", tot, " no team solve (n' <= 64 or final):", hist[0] / tot)
for lo, hi in ((65, 96), (97, 128), (129, 160), (161, 192), (193, 200)):
    print(f"  n' {lo:3d}..{hi:3d}: {hist[lo:hi + 1].sum() / tot:.3f}")
w = np.arange(N + 1, dtype=np.float64) ** 3
print("share of sum n'^3 in n' <= 128:", (hist[:129] * w[:129]).sum() / (hist * w).sum())
