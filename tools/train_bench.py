#!/usr/bin/env python3
"""End-to-end PPO rollouts + updates on one GPU (BASELINE configs[4] shard: N=50, M=8, device rollout
buffer): env-steps/s of rollout collection (policy inference + env step), of the update, and overall."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import EnvConfig
from marl_amd.trainer import BatchedTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=2048)      # 16 384 over 8 GPUs
ap.add_argument("--T", type=int, default=64)
ap.add_argument("--rollouts", type=int, default=3)
ap.add_argument("--no-graph", action="store_true")
a = ap.parse_args()
tr = BatchedTrainer(EnvConfig.fast_preset(N=50, M=8), num_envs=a.envs, rollout_length=a.T, seed=0, use_graph=not a.no_graph)
print("batch_size", tr.agent.batch_size, flush=True)
tr.collect_rollout(); print("warm-up rollout done", flush=True)
tr.update(); print("warm-up update done", flush=True)
tr.collect_rollout(); tr.update(); print("second rollout (graph capture) done", flush=True)
torch.cuda.synchronize()
tc = tu = 0.0
for _ in range(a.rollouts):
    t0 = time.perf_counter(); tr.collect_rollout(); torch.cuda.synchronize(); t1 = time.perf_counter()
    tr.update(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tc += t1 - t0; tu += t2 - t1
n = a.rollouts * a.envs * a.T
print(f"envs={a.envs} T={a.T}: collect {n/tc:,.0f} env-steps/s ({tc/a.rollouts/a.T*1e3:.3f} ms per step), "
      f"update {tu/a.rollouts:.3f} s per rollout, overall {n/(tc+tu):,.0f} env-steps/s")
