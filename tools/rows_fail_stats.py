#!/usr/bin/env python3
"""How many of the eigenproblems beyond 64 vertices the Lanczos iteration on bit rows leaves to the team solver
(BASELINE config 4: N = 200, M = 16): per step, the envs pending after the front launch and after the iteration."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marl_amd import BatchedJammerEnv, EnvConfig, _lib

E, N, M = 8192, 200, 16
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1234, debug_bits=True)
env.reset()
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
ld = (N + 7) // 8 * 8
nb = lib.marl_lambda2_workspace_bytes(N, E)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
out = torch.zeros(E, dtype=torch.float64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = np.zeros(3, dtype=np.int64)
for t in range(12):
    act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(act)
    eff = (env.adj_bits & ~env.jam_bits).contiguous()
    counts = []
    for sel in (_lib.LAMBDA2_DIRECT, _lib.LAMBDA2_LANCZOS):
        _lib.check(lib.marl_lambda2_from_bits(N, E, eff.data_ptr(), out.data_ptr(), sel, ws.data_ptr(), st))
        torch.cuda.synchronize()
        pend = ws[E * ld * ld * 8:E * ld * ld * 8 + 4 * E].view(torch.int32)
        counts.append(int((pend > 0).sum()))
    tot += np.array([E, counts[0], counts[1]])
    print(f"step {t + 1}: pending after the front launch {counts[0]}, left by the iteration {counts[1]}")
print(f"of {tot[0]} env-steps: {tot[1]} eigenproblems beyond 64 vertices, {tot[2]} left to the team solver "
      f"({100.0 * tot[2] / max(tot[1], 1):.2f} %)")
