#!/usr/bin/env python3
"""How much smaller is the eigenproblem when the complement graph is split into connected components?
Runs the benchmark workload for a while, then compares n'^3 with the cubes of the components that can
hold the largest eigenvalue (upper bound min(n_c, 2*dmax_c) above the best lower bound dmax+1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.sparse.csgraph import connected_components
from marl_amd import BatchedJammerEnv, EnvConfig

E, N, M = 2048, 50, 8
env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=1, auto_reset=True, debug_bits=True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
for t in range(57):
    act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
    act[:, :, 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
    env.step(act)
torch.cuda.synchronize()
def unpack(bits):
    w = bits.cpu().numpy().view(np.uint32).astype(np.uint64)
    j = np.arange(N)
    return ((w[:, :, j // 32] >> (j % 32).astype(np.uint64)) & 1).astype(bool)
adj, jam = unpack(env.adj_bits), unpack(env.jam_bits)
eff = adj & ~jam
full = cand = 0.0
sizes = []
for e in range(E):
    G = eff[e]
    deg = G.sum(1)
    if deg.min() == N - 1:
        continue
    if connected_components(G, directed=False)[0] > 1:
        continue
    touched = np.where(deg < N - 1)[0]
    C = ~G[np.ix_(touched, touched)]
    np.fill_diagonal(C, False)
    ncomp, lab = connected_components(C, directed=False)
    cd = C.sum(1)
    lb = cd.max() + 1
    tot = 0
    big = 0
    for c in range(ncomp):
        idx = lab == c
        nc, dm = idx.sum(), cd[idx].max()
        if min(nc, 2 * dm) > lb or (dm + 1 == lb and min(nc, 2 * dm) >= lb and nc > 2):
            tot += nc ** 3
            big = max(big, nc)
    full += len(touched) ** 3
    cand += tot
    sizes.append((len(touched), big, ncomp))
s = np.array(sizes)
print("solved envs:", len(s), " sum n'^3:", full, " sum candidate-component cubes:", cand, " ratio:", cand / full)
print("n' quartiles:", np.percentile(s[:, 0], [25, 50, 75, 100]), " largest candidate component quartiles:",
      np.percentile(s[:, 1], [25, 50, 75, 100]), " components (median):", np.median(s[:, 2]))
