#!/usr/bin/env python3
"""Per-section cycles of env_step_front_kernel (library built with -DMARL_EXP_TIMING, MARL_ENV_LIB=...):
reward_terms then holds the cycles of: 0 loads + action decode + enemy move, 1 cluster table / re-clustering,
2 adjacency + jam rows, 3 lambda2 front (shortcuts, compaction), 4 reward, observation, stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

E = 8192
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=50, M=8), seed=1, auto_reset=False)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
names = ["loads+action+enemy", "clusters", "pair rows", "lambda2 front", "reward+obs+stores"]
acc = torch.zeros(5, dtype=torch.float64)
acc10 = torch.zeros(5, dtype=torch.float64)
n = 0
for t in range(1, 40):
    act = (torch.rand(E, 8, 3, device="cuda", generator=g) - 0.5) * 10
    act[:, :, 2] = torch.randint(0, 4, (E, 8), device="cuda", generator=g).float()
    _, _, _, _, info = env.step(act)
    sec = info["reward_terms"].double().mean(0).cpu()
    if t % 10 == 0:
        acc10 += sec
    else:
        acc += sec; n += 1
print("mean cycles per wave (steps without re-clustering):")
for k in range(5):
    print(f"   {names[k]:22s} {acc[k]/n:9.0f}")
print(f"   {'sum':22s} {acc.sum()/n:9.0f}")
print("re-clustering steps:", [round(float(x) / 3) for x in acc10])
