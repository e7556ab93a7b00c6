#!/usr/bin/env python3
"""Cost of the auto-reset step inside a short timed region: 20 steps after a device synchronisation with the
auto-reset of every env at step 10, on fresh batches; one stream against three parts on streams of their own, with and
without the pre-reset copies of the episode values (wall clock per step, ms)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

E, M = 8192, 8
dev = torch.device("cuda", 0)
pool = torch.empty(16, E, M, 3, device=dev).uniform_(-5, 5)
pool[..., 2] = torch.randint(0, 4, (16, E, M), device=dev).float()
for shards, ep in ((1, True), (3, True), (3, False)):
    out = []
    for trial in range(8):
        env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=50, M=M), seed=1 + trial, device=dev, num_shards=shards)
        env.reset()
        for i in range(190):
            env.step(pool[i % 16], episode_info=ep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.fork()
        for i in range(20):
            env.step(pool[i % 16], episode_info=ep)
        env.join()
        torch.cuda.synchronize()
        out.append("%.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
        del env
    print(f"shards={shards} episode_info={ep}:", " ".join(out))
