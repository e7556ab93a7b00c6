#!/usr/bin/env python3
"""The masked reset launch alone: microseconds per reset of a whole batch (HIP events around `reps` calls).
  python3 tools/reset_bench.py [E] [N] [M]        (MARL_ENV_LIB picks the library build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig
E, N, M = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8192), (2, 50), (3, 8)))
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1, device="cuda:0")
mask = torch.ones(E, dtype=torch.bool, device="cuda:0")
env.reset()
for m in (None, mask):
    for _ in range(3):
        env.reset(mask=m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        env.reset(mask=m)
    e1.record()
    torch.cuda.synchronize()
    print(f"E={E} N={N} M={M} {'masked' if m is not None else 'whole '} reset: {e0.elapsed_time(e1) * 50:.1f} us")
