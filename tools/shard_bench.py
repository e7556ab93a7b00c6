#!/usr/bin/env python3
"""One batch of E envs on one stream against S shards of E/S envs on S streams (N=50, M=8): us per step of the whole batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = 400
for S in (1, 2, 3, 4):
    Es = E // S
    envs = [BatchedJammerEnv(Es, EnvConfig.fast_preset(N=50, M=8), seed=1, auto_reset=True, env_offset=k * Es) for k in range(S)]
    pools = [torch.empty(16, Es, 8, 3, device="cuda").uniform_(-5, 5) for _ in range(S)]
    for p in pools:
        p[..., 2] = torch.randint(0, 4, (16, Es, 8), device="cuda").float()
    streams = [torch.cuda.Stream() for _ in range(S)]
    for env in envs:
        env.reset()
    torch.cuda.synchronize()

    FJ = len(sys.argv) > 2 and sys.argv[2] == "fj"      # fork from / join to the current stream every step

    def run(n):
        cur = torch.cuda.current_stream()
        for i in range(n):
            for k in range(S):
                if FJ:
                    streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    envs[k].step(pools[k][i % 16])
            if FJ:
                for k in range(S):
                    cur.wait_stream(streams[k])
    run(30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"E={E} shards={S}: {dt / steps * 1e6:8.1f} us/step  {Es * S * steps / dt:12.0f} env-steps/s", flush=True)
    del envs
