"""The fused PPO minibatch step (marl_ppo_rows + row products) against torch autograd on the same networks:
losses, statistics and every parameter gradient (ppo_agent.py:296-331)."""
import numpy as np
import pytest
import torch

from marl_amd.agents import PPOAgent, TrainPacker, pack_train_weights

pytestmark = pytest.mark.gpu


def make_rows(agent, n, seed, scale_heads=4.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():                                   # spread the heads so that clamps and clips are hit
        for head in (agent.actor.mu_head, agent.actor.log_std_head, agent.actor.band_head):
            head.weight.mul_(scale_heads * 30)
        for m in (agent.actor.trunk[1], agent.actor.trunk[4]):
            m.weight.add_(0.3 * torch.randn(m.weight.shape, device="cuda", generator=g))
            m.bias.add_(0.3 * torch.randn(m.bias.shape, device="cuda", generator=g))
    obs = torch.randn(n, 5, device="cuda", generator=g)
    act = torch.randn(n, 3, device="cuda", generator=g) * 6
    act[:, 2] = torch.randint(0, 4, (n,), device="cuda", generator=g).float()
    with torch.no_grad():
        lp, _ = agent.actor.evaluate_actions(obs, act)
    old = lp + 0.3 * torch.randn(n, device="cuda", generator=g)          # ratios on both sides of the clip range
    adv = torch.randn(n, device="cuda", generator=g)
    adv[::7] = 0.0
    ret = torch.randn(n, device="cuda", generator=g) * 3
    return obs, act, old, adv, ret


@pytest.mark.parametrize("hidden,B,n", [(128, 4096, 9000), (64, 1000, 1000), (128, 37, 64), (128, 65536, 70000)])
def test_fused_gradients_equal_autograd(hidden, B, n):
    agent = PPOAgent(device="cuda", hidden_dim=hidden, M=4, batch_size=B, seed=3, graph_update=False, fused_update=True)
    assert agent.fused_update
    obs, act, old, adv, ret = make_rows(agent, n, 11)
    idx = torch.randperm(n, device="cuda")[:B].contiguous()
    ref = agent._forward_backward(obs[idx], act[idx], old[idx], adv[idx], ret[idx]).clone()
    params = list(agent.actor.named_parameters()) + [("critic." + k, v) for k, v in agent.critic.named_parameters()]
    want = {k: p.grad.clone() for k, p in params}
    for _, p in params:
        p.grad.zero_()
    got = agent._fused_forward_backward(obs, act, old, adv, ret, idx)
    torch.cuda.synchronize()
    assert torch.allclose(got, ref, rtol=2e-4, atol=1e-6), (got, ref)
    assert float(ref[3]) > 0.05                               # the clip range is exercised
    for k, p in params:
        w, g = want[k], p.grad
        scale = float(w.abs().max()) + 1e-12
        err = float((g - w).abs().max()) / scale
        assert err < 2e-3, (k, err, scale)


def test_train_packer_is_the_packing():
    agent = PPOAgent(device="cuda", hidden_dim=128, M=4, seed=5)
    packer = TrainPacker(agent.actor, agent.critic)
    assert torch.equal(packer.pack(), pack_train_weights(agent.actor, agent.critic))
    with torch.no_grad():
        for p in agent.actor.parameters():
            p.add_(0.25)
    assert torch.equal(packer.pack(), pack_train_weights(agent.actor, agent.critic))


def test_fused_update_follows_eager_update():
    """Whole optimise() calls: fused against eager torch on the same rows and permutations."""
    n, B = 8192, 1024
    agents = [PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=2, seed=9, graph_update=False, fused_update=f)
              for f in (False, True)]
    agents[1].actor.load_state_dict(agents[0].actor.state_dict())
    agents[1].critic.load_state_dict(agents[0].critic.state_dict())
    g = torch.Generator(device="cuda").manual_seed(2)
    obs = torch.randn(n, 5, device="cuda", generator=g)
    act = torch.randn(n, 3, device="cuda", generator=g) * 3
    act[:, 2] = torch.randint(0, 4, (n,), device="cuda", generator=g).float()
    with torch.no_grad():
        old, _ = agents[0].actor.evaluate_actions(obs, act)
    adv = torch.randn(n, device="cuda", generator=g)
    ret = torch.randn(n, device="cuda", generator=g)
    perms = [np.random.RandomState(e).permutation(n) for e in range(2)]
    before = {k: v.clone() for k, v in agents[0].actor.state_dict().items()}
    stats = [a.optimise(obs, act, old, adv, ret, permutations=perms) for a in agents]
    for k in stats[0]:
        assert abs(stats[0][k] - stats[1][k]) <= 2e-3 * max(1.0, abs(stats[0][k])), (k, stats)
    for (k, a), b in zip(agents[0].actor.state_dict().items(), agents[1].actor.state_dict().values()):
        moved = float((a - before[k]).abs().max())
        assert float((a - b).abs().max()) <= 0.02 * moved + 1e-7, k
