#!/usr/bin/env python3
"""Per-section cycle counts of the Householder loop.  Needs a library built with -DMARL_EXP_TIMING
(MARL_ENV_LIB=...): there lambda2 = N - cycles spent in section (env index % 8), summed over the steps.
Sections: 0 column prep + LDS publish, 1 matvec, 2 norm reduction, 3 sqrt/rcp/p, 4 K reduction + q,
5 LDS publish q, 6 update, 7 loop tail."""
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
NAMES = ["column_of", "aii + norm reduce", "sqrt + publish u", "matvec", "rcp + p", "K reduce + q + publish", "update", "tail sync"]

for npv in (16, 32, 50):
    N = 50
    rng = np.random.RandomState(1)
    base = []
    for g in range(64):
        A = ~np.eye(N, dtype=bool)
        idx = rng.permutation(N)[:npv]
        sub = np.triu(rng.rand(npv, npv) < 0.3, 1)
        sub = sub | sub.T
        for a in range(npv):
            if not sub[a].any():
                b = (a + 1) % npv
                sub[a, b] = sub[b, a] = True
        A[np.ix_(idx, idx)] &= ~sub
        base.append(orc.pack_rows(A).view(np.int32))
    bits = torch.from_numpy(np.stack(base)).cuda().repeat(E // 64, 1, 1).contiguous()
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    for _ in range(3):
        lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), 1, None, st())
    torch.cuda.synchronize()
    cyc = (N - out.cpu().numpy())
    steps = npv - 2
    tot = 0
    print(f"n'={npv}: cycles per Householder step (mean over waves)")
    for k in range(8):
        c = cyc[k::8].mean() / steps
        tot += c
        print(f"   {NAMES[k]:16s} {c:8.0f}")
    print(f"   {'sum':16s} {tot:8.0f}")
