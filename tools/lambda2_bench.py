#!/usr/bin/env python3
"""Times marl_lambda2_from_bits on synthetic graphs: n' (vertices touching a missing edge) and rounds sweep."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from marl_amd import _lib
from oracle import jammer_oracle as orc

lib = _lib.load()
E = 8192
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(N, npv, rounds, p_missing=0.3):
    rng = np.random.RandomState(1)
    base = []
    for g in range(64):
        A = ~np.eye(N, dtype=bool)
        idx = rng.permutation(N)[:npv]
        if npv >= 2:
            sub = np.triu(rng.rand(npv, npv) < p_missing, 1)
            sub = sub | sub.T
            # make sure every chosen vertex misses at least one edge
            for a in range(npv):
                if not sub[a].any():
                    b = (a + 1) % npv
                    sub[a, b] = sub[b, a] = True
            A[np.ix_(idx, idx)] &= ~sub
        base.append(orc.pack_rows(A).view(np.int32))
    bits = torch.from_numpy(np.stack(base)).cuda().repeat(E // 64, 1, 1).contiguous()
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    for _ in range(3):
        lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), rounds, None, st())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), rounds, None, st())
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 * 1e3


for N in (50,):
    for npv in (8, 16, 24, 32, 40, 50):
        t8 = run(N, npv, 8)
        t1 = run(N, npv, 1)
        print(f"N={N} n'={npv:3d}: rounds=8 {t8:8.1f} us ({t8*1e3/E:6.1f} ns/env)   rounds=1 {t1:8.1f} us   sturm/round ~{(t8-t1)/7:6.1f} us")
