#!/usr/bin/env python3
"""Solver probe on the benchmark's graphs: the lambda2 stage op on the jammed graphs of 8 192 envs x (N=50, M=8) after
some random steps, timed with the Lanczos iteration on and off; with a library built with -DMARL_EXP_LANCZOS_STATS
(MARL_ENV_LIB=...) the value returned per env encodes steps + 100 checks + 10000 fallback instead, and their
distribution over the problem size is printed."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marl_amd import BatchedJammerEnv, EnvConfig, _lib

E, N, M = int(os.environ.get("E", 8192)), 50, 8
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1234, auto_reset=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(60):
    a = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    a[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(a)
lib, v = env.lib, env.views
W = 2
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
adj = torch.zeros(E, N, W, dtype=torch.int32, device="cuda"); jam = torch.zeros_like(adj)
_lib.check(lib.marl_adjacency_bits(C.byref(env.params), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(), v["enemy_band"].data_ptr(), adj.data_ptr(), st))
_lib.check(lib.marl_jam_mask_bits(C.byref(env.params), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(), v["enemy_band"].data_ptr(),
                                  v["jam_x"].data_ptr(), v["jam_y"].data_ptr(), v["jam_band"].data_ptr(), jam.data_ptr(), st))
eff = (adj & ~jam).contiguous()
out = {}
for name, rounds in (("lanczos", 0x200), ("direct", 0x100)):
    o = torch.zeros(E, dtype=torch.float64, device="cuda")
    call = lambda: _lib.check(lib.marl_lambda2_from_bits(N, E, eff.data_ptr(), o.data_ptr(), rounds, None, st))
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    out[name] = o.cpu().numpy()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per call of {E} graphs", flush=True)
# problem sizes: touched vertices of connected, incomplete graphs
A = np.zeros((E, N, N), dtype=bool)
w = eff.cpu().numpy().view(np.uint32)
for j in range(N):
    A[:, :, j] = (w[:, :, j // 32] >> np.uint32(j % 32)) & 1
npv = (A.sum(2) < N - 1).sum(1)
if os.environ.get("STATS"):
    s = out["lanczos"]
    k = (s % 100).astype(int); ch = ((s // 100) % 100).astype(int); fb = (s // 10000).astype(int)
    direct_l2 = out["direct"]
    solved = (direct_l2 > 1e-9) & (npv >= 20) & (direct_l2 < N - 1e-9)
    print("solved by lanczos path:", solved.sum(), "of", E)
    for lo, hi in ((20, 33), (33, 40), (40, 48), (48, 51)):
        m = solved & (npv >= lo) & (npv < hi)
        if m.any():
            print(f"np {lo}..{hi-1}: n={m.sum()} steps mean {k[m].mean():.1f} p90 {np.percentile(k[m], 90):.0f} max {k[m].max()} checks mean {ch[m].mean():.2f} max {ch[m].max()} fallbacks {fb[m].sum()}")
else:
    d = np.abs(out["lanczos"] - out["direct"])
    print("max |lanczos - direct| =", d.max(), " mean lambda2", out["direct"].mean())
    print("np histogram (solved graphs):", np.histogram(npv[(out['direct'] > 1e-9) & (out['direct'] < N - 1e-9)], bins=[0, 2, 10, 20, 33, 40, 48, 51])[0])
if os.environ.get("TICKS"):
    s = out["lanczos"]
    solved = (out["direct"] > 1e-9) & (npv >= 33) & (out["direct"] < N - 1e-9)
    for lo, hi in ((33, 40), (40, 48), (48, 51)):
        m = solved & (npv >= lo) & (npv < hi)
        if m.any():
            print(f"np {lo}..{hi-1}: n={m.sum()} ticks mean {s[m].mean():.0f} p90 {np.percentile(s[m], 90):.0f} max {s[m].max():.0f}")
if os.environ.get("WORST"):
    d = np.abs(out["lanczos"] - out["direct"])
    for e in np.argsort(-d)[:6]:
        print("env", e, "np", npv[e], "lanczos", out["lanczos"][e], "direct", out["direct"][e])
