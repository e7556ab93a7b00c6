#!/usr/bin/env python3
"""marl_lambda2_from_bits at N = 200 on graphs like the step's (complete minus jammed links)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from marl_amd import _lib
from oracle import jammer_oracle as orc
lib = _lib.load()
N, E = 200, 2048
rng = np.random.RandomState(0)
base = []
for g in range(32):
    A = ~np.eye(N, dtype=bool)
    k = rng.randint(70, 200)
    idx = rng.permutation(N)[:k]
    sub = np.triu(rng.rand(k, k) < 0.05, 1); sub = sub | sub.T
    A[np.ix_(idx, idx)] &= ~sub
    base.append(orc.pack_rows(A).view(np.int32))
bits = torch.from_numpy(np.stack(base)).cuda().repeat(E // 32, 1, 1).contiguous()
out = torch.zeros(E, dtype=torch.float64, device="cuda")
ws = torch.zeros(lib.marl_lambda2_workspace_bytes(N, E), dtype=torch.uint8, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(2):
    _lib.check(lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), st()))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), st())
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
ref = []
for g in range(32):
    A = orc.unpack_rows(base[g].view(np.uint32), N)
    ref.append(orc.lambda2_of_laplacian(orc.laplacian(A)))
err = np.abs(out[:32].cpu().numpy() - np.array(ref)).max()
print(f"N={N} E={E}: {ms:.2f} ms per launch, {ms*1e3/E:.2f} us per env; max |err| vs LAPACK {err:.2e}")
