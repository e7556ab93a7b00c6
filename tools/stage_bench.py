#!/usr/bin/env python3
"""Times the separately callable stage kernels on realistic states taken from a running batch
(N=50, M=8, fast preset, random actions).  Prints microseconds per env for each stage."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import BatchedJammerEnv, EnvConfig, _lib

E, N, M = int(os.environ.get("E", 8192)), int(os.environ.get("N", 50)), int(os.environ.get("M", 8))
cfg = EnvConfig.fast_preset(N=N, M=M)
env = BatchedJammerEnv(E, cfg, seed=7, auto_reset=True, debug_bits=True)
env.reset()
for t in range(37):
    act = torch.empty(E, M, 3, device="cuda").uniform_(-5, 5)
    act[..., 2] = torch.randint(0, 4, (E, M), device="cuda").float()
    env.step(act)
lib, P, v = env.lib, env.params, env.views
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
eff = (env.adj_bits & ~env.jam_bits).contiguous()
W = (N + 31) // 32
out = torch.zeros(E, dtype=torch.float64, device="cuda")
bits = torch.zeros(E, N, W, dtype=torch.int32, device="cuda")
labels = torch.zeros(E, N, dtype=torch.int32, device="cuda")
cx = torch.zeros(E, N + 1, dtype=torch.float64, device="cuda"); cy = torch.zeros_like(cx)
csz = torch.zeros(E, N + 1, dtype=torch.int32, device="cuda")
ncl = torch.zeros(E, dtype=torch.int32, device="cuda"); nce = torch.zeros_like(ncl)


def timeit(name, fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:28s} {ms*1e3:9.1f} us/launch  {ms*1e6/E:8.1f} ns/env")


timeit("lambda2_from_bits", lambda: lib.marl_lambda2_from_bits(N, E, eff.data_ptr(), out.data_ptr(), 0, None, st()))
full = env.adj_bits.contiguous()
timeit("lambda2 (complete graphs)", lambda: lib.marl_lambda2_from_bits(N, E, full.data_ptr(), out.data_ptr(), 0, None, st()))
timeit("adjacency_bits", lambda: lib.marl_adjacency_bits(C.byref(P), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(),
                                                         v["enemy_band"].data_ptr(), bits.data_ptr(), st()))
timeit("jam_mask_bits", lambda: lib.marl_jam_mask_bits(C.byref(P), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(),
                                                       v["enemy_band"].data_ptr(), v["jam_x"].data_ptr(), v["jam_y"].data_ptr(),
                                                       v["jam_band"].data_ptr(), bits.data_ptr(), st()))
timeit("dbscan", lambda: lib.marl_dbscan(C.byref(P), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(), labels.data_ptr(),
                                         cx.data_ptr(), cy.data_ptr(), csz.data_ptr(), ncl.data_ptr(), nce.data_ptr(), st()))
act = torch.empty(E, M, 3, device="cuda").uniform_(-5, 5)
env.auto_reset = False
timeit("env_step (whole)", lambda: env.step(act))
# graph statistics of the realistic batch
effc = eff.cpu().numpy().view("uint32")
import numpy as np
deg = np.zeros((E, N), dtype=np.int64)
for w in range(W):
    x = effc[:, :, w]
    deg += np.array([bin(int(t)).count("1") for t in x.ravel()]).reshape(E, N)
complete = (deg == N - 1).all(axis=1)
touched = (deg < N - 1).sum(axis=1)
lam = out.cpu().numpy()
print("complete graphs: %.3f  mean touched vertices (deg<N-1): %.1f  quartiles %s" %
      (complete.mean(), touched.mean(), np.percentile(touched, [10, 25, 50, 75, 90])))
