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
