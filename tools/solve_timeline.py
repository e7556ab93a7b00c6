#!/usr/bin/env python3
"""Timeline of the eigen-solve launch, block by block.  Needs a library built with -DMARL_EXP_BLOCKTIME
(MARL_ENV_LIB=...): there every solve block writes, in place of its env's reward terms, the hardware id of
its wave, the size of its problem and its start / end clock.  Prints, per SIMD, when the last block ended,
how much block time the SIMD carried, and what the launch would take if the load were spread evenly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
cfg = EnvConfig.fast_preset(N=50, M=8, step_schedule=os.environ.get("MARL_BENCH_SCHEDULE", "auto"))
env = BatchedJammerEnv(E, cfg, seed=1234, device=dev, auto_reset=True)
env.reset()
g = torch.Generator(device=dev).manual_seed(4321)
pool = torch.empty(16, E, 8, 3, device=dev)
pool[..., :2] = (torch.rand(16, E, 8, 2, device=dev, generator=g) - 0.5) * 10.0
pool[..., 2] = torch.randint(0, 4, (16, E, 8), device=dev, generator=g).float()
for i in range(57):
    env.step(pool[i % 16])
env.reward_terms.zero_()
env.step(pool[9])
torch.cuda.synchronize()
w = env.reward_terms.cpu().numpy().view(np.uint32)
ok = (w[:, 1] >> 24) == 0xAB
w = w[ok]
print(f"{ok.sum()} envs solved in the launch of {E}")
hw, xcc, npv = w[:, 0], w[:, 1] & 0xF, (w[:, 1] >> 8) & 0xFF
t0, t1, blk = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64), w[:, 4]
# pairs share a block: keep one record per block
_, first = np.unique(blk, return_index=True)
hw, xcc, npv, t0, t1, blk = hw[first], xcc[first], npv[first], t0[first], t1[first], blk[first]
dur = (t1 - t0) & 0xFFFFFFFF
simd = (xcc.astype(np.int64) << 16) | (hw & 0xFF30)
print(f"{len(blk)} blocks on {len(np.unique(simd))} SIMDs; block durations: n'=50 median {np.median(dur[npv == 50]):.0f} ticks, "
      f"all: mean {dur.mean():.0f}, max {dur.max()}")
# clocks are per XCD: normalise by the first start within the XCD
span_all = []
for x in np.unique(xcc):
    m = xcc == x
    base = t0[m].min()
    end = ((t1[m] - base) & 0xFFFFFFFF)
    span_all.append(end.max())
    sk = simd[m]
    ends = np.array([end[sk == s].max() for s in np.unique(sk)])
    load = np.array([dur[m][sk == s].sum() for s in np.unique(sk)])
    cnt = np.array([(sk == s).sum() for s in np.unique(sk)])
    print(f"XCD {x}: {m.sum()} blocks, span {end.max()} ticks; last end per SIMD: min {ends.min()} median {np.median(ends):.0f} max {ends.max()}; "
          f"block-time per SIMD: min {load.min()} mean {load.mean():.0f} max {load.max()}; blocks per SIMD {cnt.min()}..{cnt.max()}")
print("launch span (max over XCDs):", max(span_all), "ticks")
# duration against size, for blocks that started in the first round (all slots busy) and later ones
for lo, hi in ((2, 8), (9, 16), (17, 24), (25, 32), (33, 40), (41, 49), (50, 50)):
    m = (npv >= lo) & (npv <= hi)
    if m.any():
        print(f"n' {lo:2d}..{hi:2d}: {m.sum():5d} blocks, duration mean {dur[m].mean():8.0f}  min {dur[m].min():8d}  max {dur[m].max():8d}")
np.save("gpurun_out/solve_timeline.npy", np.stack([blk, xcc, hw, npv, t0, t1]))
