"""marl_lambda2_from_bits_static -- the entry point that allocates nothing -- against
marl_lambda2_from_bits (same kernels, caller-provided workspace): equal results bit for bit, on two
calls in a row of which the second has a smaller input (the static scratch keeps the first call's
contents), for the three solver paths that use scratch (team solver, matrix in the workspace) and the
wave solver that uses none; sizes beyond the static buffer are refused."""
import ctypes as C

import numpy as np
import pytest
import torch

from marl_amd import _lib
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _graphs(n, E, seed):
    rng = np.random.RandomState(seed)
    mats, want = [], []
    for e in range(E):
        A = np.triu(rng.rand(n, n) < rng.uniform(0.2, 0.97), 1)
        A = A | A.T
        mats.append(orc.pack_rows(A).view(np.int32))
        want.append(orc.lambda2_of_laplacian(orc.laplacian(A)))
    return torch.as_tensor(np.stack(mats)).cuda(), np.array(want)


def _with_workspace(lib, n, E, bits):
    out = torch.full((E,), -1.0, dtype=torch.float64, device="cuda")
    nb = lib.marl_lambda2_workspace_bytes(n, E)
    ws = torch.zeros(max(nb, 8), dtype=torch.uint8, device="cuda")
    _lib.check(lib.marl_lambda2_from_bits(n, E, bits.data_ptr(), out.data_ptr(), 0, ws.data_ptr() if nb else None,
                                          _stream()))
    torch.cuda.synchronize()
    return out


def _static(lib, n, E, bits):
    out = torch.full((E,), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    _lib.check(lib.marl_lambda2_from_bits_static(n, E, bits.data_ptr(), out.data_ptr(), 0, _stream()),
               "marl_lambda2_from_bits_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free_before          # no device memory was taken by the call
    return out


@pytest.mark.parametrize("n", [200, 224, 100, 50])
def test_static_scratch_entry_point_equals_workspace_entry_point(n):
    lib = _lib.load()
    for E, seed in ((24, 1), (7, 2)):           # second call: smaller input over the first call's leftovers
        bits, want = _graphs(n, E, seed + n)
        ref = _with_workspace(lib, n, E, bits)
        got = _static(lib, n, E, bits)
        assert torch.equal(got, ref), (n, E)
        assert np.max(np.abs(got.cpu().numpy() - want) / np.maximum(1.0, np.abs(want))) < 2e-9


def test_static_scratch_entry_point_rejects_oversized_requests():
    lib = _lib.load()
    n = 256
    E = 1
    while lib.marl_lambda2_workspace_bytes(n, E) <= _lib.STATIC_SCRATCH_BYTES:
        E += 1
    bits = torch.zeros(E, n, 8, dtype=torch.int32, device="cuda")
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    assert lib.marl_lambda2_from_bits_static(n, E, bits.data_ptr(), out.data_ptr(), 0, _stream()) == _lib.MARL_ETOOBIG
    assert b"static" in lib.marl_strerror(_lib.MARL_ETOOBIG)
    assert lib.marl_lambda2_from_bits_static(n, E - 1, bits.data_ptr(), out.data_ptr(), 0, _stream()) == _lib.MARL_OK
    torch.cuda.synchronize()
    assert lib.marl_lambda2_from_bits_static(257, 1, bits.data_ptr(), out.data_ptr(), 0, _stream()) == _lib.MARL_EUNSUPPORTED
    assert lib.marl_lambda2_from_bits_static(50, 0, bits.data_ptr(), out.data_ptr(), 0, _stream()) == _lib.MARL_EINVAL
