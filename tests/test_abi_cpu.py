"""CPU-side checks of the C ABI: the library loads, exports every symbol
include/marl_env.h declares, the ctypes struct mirrors have the C sizes, and the
host-only entry points (params init, state carving, argument validation) behave.
No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from marl_amd import _lib
from marl_amd.env import EnvConfig
from oracle import jammer_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "marl_env.h")).read()
    declared = set(re.findall(r"\b(marl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_sizes_match_header_layout():
    # 10 int32 + 20 double + 12 double + 4 float
    assert C.sizeof(_lib.MarlEnvParams) == 10 * 4 + 32 * 8 + 12 * 4
    assert C.sizeof(_lib.MarlEnvState) == 8 + 34 * 8
    assert C.sizeof(_lib.MarlStepIO) == 13 * 8
    assert C.sizeof(_lib.MarlPolicyIO) == 9 * 8
    assert C.sizeof(_lib.MarlResetIO) == 11 * 8
    assert C.sizeof(_lib.MarlPpoIO) == 20 * 8 + 6 * 4
    assert C.sizeof(_lib.MarlPpoApply) == 9 * 8 + 3 * 8 + 8 * 4  # 7 x 4 bytes, padded


@pytest.mark.parametrize("cfg", [EnvConfig.fast_preset(), EnvConfig(),
                                 EnvConfig(arena_size=6000.0, P_jammer_dbm=30.0, P_jam_thresh_dbm=-45.0)])
def test_params_thresholds_reproduce_reference_comparison(cfg):
    """d2 <= comm_d2_max must equal the reference's float64 chain P_R >= P_sens
    (physics/fspl.py:199-206) for doubles on both sides of the threshold."""
    p = cfg.to_params()
    for b in range(4):
        f = orc.FREQUENCY_BANDS[b]
        for d2max, P, thr in ((p.comm_d2_max[b], p.P_tx_w, p.P_sens_w), (p.jam_d2_max[b], p.P_jam_w, p.P_thr_w)):
            assert d2max > 0
            cands = [d2max]
            x = d2max
            for _ in range(8):
                x = np.nextafter(x, np.inf)
                cands.append(x)
            x = d2max
            for _ in range(8):
                x = np.nextafter(x, 0.0)
                cands.append(x)
            cands += [d2max * (1 + 1e-12), d2max * (1 - 1e-12), 0.0, d2max * 4, d2max / 4]
            for d2 in cands:
                ref = bool(orc.received_power_watts(P, np.sqrt(np.float64(d2)), f) >= thr)
                assert (d2 <= d2max) == ref, (b, d2, d2max)
        R = orc.compute_jam_range(p.P_jam_w, p.P_thr_w, f)
        assert p.R_jam[b] == R
        assert abs(np.sqrt(p.jam_d2_max[b]) / R - 1) < 1e-12


def test_unreachable_threshold_gives_no_links():
    cfg = EnvConfig(P_tx_dbm=-300.0, P_sens_dbm=300.0)
    p = cfg.to_params()
    assert all(p.comm_d2_max[b] == -1.0 for b in range(4))


def test_state_carving_is_disjoint_and_aligned():
    lib = _lib.load()
    p = EnvConfig.fast_preset(N=50, M=8).to_params()
    E = 37
    nbytes = lib.marl_state_bytes(C.byref(p), E)
    assert nbytes > 0
    st = _lib.MarlEnvState()
    base = 0x10000000
    _lib.check(lib.marl_state_bind(C.byref(st), C.byref(p), E, C.c_void_p(base)))
    assert st.workspace is None                       # N <= 64 needs no lambda2 scratch
    assert st.solve_rows is not None and st.solve_meta is not None and st.solve_partial is not None
    ptrs = sorted(getattr(st, n) for n in _lib._STATE_ARRAYS if n != "workspace")
    assert ptrs[0] == base and all(q % 256 == 0 for q in ptrs)
    assert len(set(ptrs)) == len(ptrs) and ptrs[-1] < base + nbytes
    # N = 200: one 200 x 200 float64 matrix per env on top (+ the pending sizes and the team solver's list of 1 024 envs)
    p2 = EnvConfig.fast_preset(N=200, M=16).to_params()
    n2 = lib.marl_state_bytes(C.byref(p2), E)
    assert E * 200 * 200 * 8 <= lib.marl_lambda2_workspace_bytes(200, E) <= E * 200 * 200 * 8 + 8192 and n2 > E * 200 * 200 * 8
    assert lib.marl_lambda2_workspace_bytes(1000, 2) == 2 * 1024 * 1024 * 8 and lib.marl_lambda2_workspace_bytes(1025, 1) == 0
    _lib.check(lib.marl_state_bind(C.byref(st), C.byref(p2), E, C.c_void_p(base)))
    assert st.workspace is not None and st.workspace + E * 200 * 200 * 8 <= base + n2
    assert st.solve_rows is None                      # the three-launch step is for N <= 64
    assert lib.marl_state_bind(C.byref(st), C.byref(p), E, C.c_void_p(base + 8)) == _lib.MARL_EINVAL


def test_argument_validation_without_launch():
    lib = _lib.load()
    bad = EnvConfig(N=300)
    with pytest.raises(_lib.MarlError):
        bad.to_params()
    p = EnvConfig.fast_preset().to_params()
    st = _lib.MarlEnvState()
    io = _lib.MarlStepIO()
    assert lib.marl_env_step(C.byref(p), C.byref(st), C.byref(io), None) == _lib.MARL_EINVAL
    assert lib.marl_env_step(None, None, None, None) == _lib.MARL_EINVAL
    assert lib.marl_env_step_timed(C.byref(p), C.byref(st), C.byref(io), None, None, None) == _lib.MARL_EINVAL   # no events
    assert lib.marl_reward_obs(C.byref(p), 4, None, None) == _lib.MARL_EINVAL
    # a `rounds` word: rounds in the low byte, at most one solver selector, nothing unknown
    for bad in (-1, _lib.LAMBDA2_DIRECT | _lib.LAMBDA2_LANCZOS, 0x800, 0x10000):
        assert lib.marl_lambda2_from_bits(10, 1, C.cast(1 << 12, _lib._pu32), C.cast(1 << 12, _lib._pd), bad, None, None) \
            == _lib.MARL_EINVAL, hex(bad)
    assert lib.marl_lambda2_from_bits(0, 1, None, None, 0, None, None) == _lib.MARL_EINVAL
    assert lib.marl_gae(0, 1, None, None, None, None, None, 0.99, 0.95, None, None, None) == _lib.MARL_EINVAL
    assert b"invalid" in lib.marl_strerror(_lib.MARL_EINVAL)


def test_product_path_has_no_oracle_dependency():
    """Nothing under marl_amd/ may import or call the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "marl_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "jammer_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
