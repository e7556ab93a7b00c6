#!/usr/bin/env python3
"""Per-section cycles of the team solver's Householder step (library built with -DMARL_EXP_TIMING):
lambda2 = N - cycles of section (env % 8).  0 loop top, 1 swap + norm partial + publish v, 2 barrier 1,
3 sqrt / rcp, 4 matvec, 5 p, K partial, barrier 2, 6 update."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marl_amd import _lib
from oracle import jammer_oracle as orc
lib = _lib.load()
N, E = 200, 2048
names = ["loop top", "swap + norm partial", "barrier 1", "sqrt/rcp", "matvec",
         "p, K partial, barrier 2", "update", "-"]
for k in (100, 200):
    rng = np.random.RandomState(0)
    A = ~np.eye(N, dtype=bool)
    idx = rng.permutation(N)[:k]
    sub = np.triu(rng.rand(k, k) < 0.05, 1); sub = sub | sub.T
    for a in range(k):
        if not sub[a].any():
            sub[a, (a + 1) % k] = sub[(a + 1) % k, a] = True
    A[np.ix_(idx, idx)] &= ~sub
    bits = torch.from_numpy(orc.pack_rows(A).view(np.int32)[None]).cuda().repeat(E, 1, 1).contiguous()
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    ws = torch.zeros(lib.marl_lambda2_workspace_bytes(N, E), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        _lib.check(lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), 1, ws.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    cyc = N - out.cpu().numpy()
    tot = 0
    print(f"n'={k}: cycles per step (wave 0)")
    for s in range(7):
        c = cyc[s::8].mean() / (k - 2); tot += c
        print(f"   {names[s]:26s} {c:8.0f}")
    print(f"   {'sum':26s} {tot:8.0f}")
