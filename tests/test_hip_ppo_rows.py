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
    agent = PPOAgent(device="cuda", hidden_dim=hidden, M=4, batch_size=B, seed=3, fused_update=True)
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


def test_apply_is_clip_and_adam():
    """marl_ppo_apply against clip_grad_norm_ + torch.optim.Adam (ppo_agent.py:324-331) on the same gradients: norms
    above and below the clip threshold, three steps, then the packed weights it leaves for the next row pass."""
    agents = [PPOAgent(device="cuda", M=4, seed=4, fused_update=f) for f in (False, True)]
    eager, fused = agents
    fused.actor.load_state_dict(eager.actor.state_dict())
    fused.critic.load_state_dict(eager.critic.state_dict())
    z = fused._fused_scratch(64)
    nets = [list(a.actor.parameters()) + list(a.critic.parameters()) for a in agents]
    g = torch.Generator(device="cuda").manual_seed(8)
    for step, scale in enumerate((1.0, 1e-4, 0.05)):
        for pe, pf in zip(*nets):
            grad = scale * torch.randn(pe.shape, device="cuda", generator=g)
            pe.grad = grad.clone()
            pf.grad.copy_(grad)
        eager._apply_gradients()
        fused._fused_apply()
        torch.cuda.synchronize()
        for pe, pf in zip(*nets):
            assert torch.allclose(pf, pe, rtol=0, atol=2e-7), (step, float((pf - pe).abs().max()))
            opt = fused.actor_optimizer if any(pf is q for q in fused.actor.parameters()) else fused.critic_optimizer
            ref = eager.actor_optimizer if opt is fused.actor_optimizer else eager.critic_optimizer
            assert float(opt.state[pf]["step"]) == float(ref.state[pe]["step"]) == step + 1
            for key in ("exp_avg", "exp_avg_sq"):               # the clip factors differ by the rounding of two norms
                got, want = opt.state[pf][key], ref.state[pe][key]
                assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), (step, key)
        assert torch.equal(z["packed"], pack_train_weights(fused.actor, fused.critic))


def test_fused_agent_resumes_an_eager_checkpoint(tmp_path):
    """A checkpoint written after eager updates, loaded into an agent that steps through marl_ppo_apply: the next
    update starts from the saved moments and step counts (ppo_agent.py:386-458)."""
    n, B = 4096, 1024
    eager = PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=1, seed=12, fused_update=False)
    g = torch.Generator(device="cuda").manual_seed(6)
    obs = torch.randn(n, 5, device="cuda", generator=g)
    act = torch.randn(n, 3, device="cuda", generator=g) * 3
    act[:, 2] = torch.randint(0, 4, (n,), device="cuda", generator=g).float()
    with torch.no_grad():
        old, _ = eager.actor.evaluate_actions(obs, act)
    adv = torch.randn(n, device="cuda", generator=g)
    ret = torch.randn(n, device="cuda", generator=g)
    perms = [np.random.RandomState(3).permutation(n)]
    eager.optimise(obs, act, old, adv, ret, permutations=perms)
    eager.save(tmp_path)
    fused = PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=1, seed=13, fused_update=True)
    fused.optimise(obs, act, old, adv, ret, permutations=perms)          # its flat state exists before the load
    fused.load(tmp_path)
    before = {k: v.clone() for k, v in eager.actor.state_dict().items()}
    for a in (eager, fused):
        a.optimise(obs, act, old, adv, ret, permutations=perms)
    for (k, a), b in zip(eager.actor.state_dict().items(), fused.actor.state_dict().values()):
        moved = float((a - before[k]).abs().max())
        assert float((a - b).abs().max()) <= 0.02 * moved + 1e-7, k
    steps = {float(st["step"]) for opt in (fused.actor_optimizer, fused.critic_optimizer) for st in opt.state.values()}
    assert steps == {8.0}
    fused.save(tmp_path / "again")
    third = PPOAgent.from_checkpoint(tmp_path / "again", device="cuda")
    for a, b in zip(fused.critic.state_dict().values(), third.critic.state_dict().values()):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [8192, 8500])
def test_fused_update_follows_eager_update(n):
    """Whole optimise() calls: fused against eager torch on the same rows and permutations (n = 8500: a short last
    minibatch per epoch)."""
    B = 1024
    agents = [PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=2, seed=9, fused_update=f)
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
