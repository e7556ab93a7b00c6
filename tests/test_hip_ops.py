"""torch.ops.marl_amd.* custom ops: same results as the direct C-ABI calls / the oracle, and the
fake implementations give the right shapes (opcheck-style)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from marl_amd import BatchedJammerEnv, EnvConfig
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def test_stage_ops_match_reference_vectors():
    import marl_amd.ops  # noqa: F401  registers torch.ops.marl_amd
    cases = [(p, d) for p, d in gu.stage_cases() if p.N == 50 and p.arena_size == 200.0 and p.eps == 25.0]
    assert cases
    p = cases[0][0]
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).cuda()
    x = dev(np.stack([d["epos"][:, 0] for _, d in cases]), torch.float64)
    y = dev(np.stack([d["epos"][:, 1] for _, d in cases]), torch.float64)
    band = dev([int(d["band"]) for _, d in cases], torch.int32)
    jx = dev(np.stack([d["jpos"][:, 0] for _, d in cases]), torch.float64)
    jy = dev(np.stack([d["jpos"][:, 1] for _, d in cases]), torch.float64)
    jb = dev(np.stack([d["jb"] for _, d in cases]), torch.int32)
    adj = torch.ops.marl_amd.adjacency_bits(x, y, band, p.P_tx_dbm, p.P_sens_dbm)
    jam = torch.ops.marl_amd.jam_mask_bits(x, y, band, jx, jy, jb, p.P_jammer_dbm, p.P_jam_thresh_dbm)
    lam = torch.ops.marl_amd.lambda2_from_bits((adj & ~jam).contiguous(), 0)
    labels, cx, cy, csz, ncl, nce = torch.ops.marl_amd.dbscan_labels(x, y, p.eps, p.min_samples, p.arena_size)
    for e, (_, d) in enumerate(cases):
        assert np.array_equal(adj[e].cpu().numpy().view(np.uint32), d["adj"])
        assert np.array_equal(jam[e].cpu().numpy().view(np.uint32), d["jam"])
        assert gu.lam_close(float(lam[e]), float(d["lam"]))
        assert np.array_equal(labels[e].cpu().numpy(), d["labels"])
        assert int(nce[e]) == int(d["ncent"])


def test_gae_op_and_fake_shapes():
    import marl_amd.ops  # noqa: F401
    z = gu.load("edges.npz")
    T, E = 64, 3
    rep = lambda a: torch.from_numpy(np.repeat(a[:, None], E, 1).astype(np.float32)).cuda().contiguous()
    adv, ret = torch.ops.marl_amd.gae(rep(z["gae_rew"]), rep(z["gae_val"]), rep(z["gae_done"]),
                                      torch.full((E,), float(z["gae_last_value"]), device="cuda"),
                                      torch.zeros(E, device="cuda"), 0.99, 0.95)
    assert np.array_equal(ret[:, 1].cpu().numpy(), z["gae_ret"])
    bits = torch.zeros(5, 50, 2, dtype=torch.int32, device="meta")
    assert torch.ops.marl_amd.lambda2_from_bits(bits, 0).shape == (5,)
    xm = torch.zeros(5, 50, dtype=torch.float64, device="meta")
    assert torch.ops.marl_amd.adjacency_bits(xm, xm, torch.zeros(5, dtype=torch.int32, device="meta")).shape == (5, 50, 2)


def test_env_step_op_mutates_registered_state():
    from marl_amd import ops
    cfg = EnvConfig.fast_preset(N=10, M=4)
    a = BatchedJammerEnv(32, cfg, seed=4, auto_reset=False)
    b = BatchedJammerEnv(32, cfg, seed=4, auto_reset=False)
    a.reset(); b.reset()
    h = ops.register_env(b)
    obs = torch.zeros(32, 4, 5, device="cuda"); rew = torch.zeros(32, device="cuda")
    tr = torch.zeros(32, dtype=torch.uint8, device="cuda")
    for t in range(6):
        act = (torch.rand(32, 4, 3, device="cuda") - 0.5) * 10
        oa, ra, _, _, _ = a.step(act)
        torch.ops.marl_amd.env_step(b._buf, h, act, None, obs, rew, tr)
        assert torch.equal(oa, obs) and torch.equal(ra, rew)
        assert torch.equal(a.views["enemy_x"], b.views["enemy_x"])
    pa = torch.ops.marl_amd.reward_per_agent(b._buf, h, b.lambda2.clone())
    assert pa.shape == (32, 4) and torch.equal(pa, a.reward_per_agent())


def _pack_bits(adj):
    """bool [E,N,N] -> the int32 words [E,N,ceil(N/32)] of the adjacency-bit layout (bit k of word w: vertex 32 w + k)."""
    E, N, _ = adj.shape
    W = (N + 31) // 32
    padded = np.zeros((E, N, 32 * W), dtype=np.uint8)
    padded[:, :, :N] = adj
    return np.packbits(padded.reshape(E, N, W, 32), axis=-1, bitorder="little").view(np.uint32).reshape(E, N, W)


@pytest.mark.parametrize("N", [257, 300, 640, 1024])
def test_lambda2_of_large_graphs_equals_lapack(N):
    """compute_lambda2 beyond the env kernels' 256 vertices (communication_graph.py:219-286, the size range of its sparse
    branch): sparse, dense, disconnected, complete and path graphs against numpy's eigvalsh.  The reference's own
    fixtures stop at N = 200, so the expected values here are LAPACK's on the same Laplacian (its dense branch)."""
    import marl_amd.ops  # noqa: F401
    rng = np.random.RandomState(N)
    adj = np.zeros((6, N, N), dtype=bool)
    for e, p in enumerate((2.5 * np.log(N) / N, 0.3, 0.5 * np.log(N) / N)):      # connected sparse, dense, disconnected
        upper = np.triu(rng.rand(N, N) < p, 1)
        adj[e] = upper | upper.T
    adj[3] = ~np.eye(N, dtype=bool)                                               # complete: lambda2 = N
    i = np.arange(N - 1)
    adj[4, i, i + 1] = adj[4, i + 1, i] = True                                    # path: already tridiagonal
    adj[5, : N // 2, : N // 2] = ~np.eye(N // 2, dtype=bool)                      # a clique and isolated vertices
    bits = torch.from_numpy(_pack_bits(adj).view(np.int32)).cuda()
    got = torch.ops.marl_amd.lambda2_from_bits(bits, 0).cpu().numpy()
    for e in range(adj.shape[0]):
        a = adj[e].astype(np.float64)
        want = np.linalg.eigvalsh(np.diag(a.sum(1)) - a)[1]
        assert abs(got[e] - want) <= 1e-9 * max(1.0, abs(want)), (e, got[e], want)
    assert abs(got[3] - N) < 1e-9 and abs(got[2]) < 1e-9 and abs(got[5]) < 1e-9


def test_lambda2_size_limits():
    from marl_amd import _lib
    lib = _lib.load()
    out = torch.zeros(1, dtype=torch.float64, device="cuda")
    bits = torch.zeros(1, 1025, 33, dtype=torch.int32, device="cuda")
    assert lib.marl_lambda2_from_bits(1025, 1, bits.data_ptr(), out.data_ptr(), 0, None, None) == _lib.MARL_EUNSUPPORTED
    assert lib.marl_lambda2_from_bits(300, 1, bits.data_ptr(), out.data_ptr(), 0, None, None) == _lib.MARL_EINVAL   # no workspace
