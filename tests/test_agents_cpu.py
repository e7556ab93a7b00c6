"""CPU tests of the agent side (pure torch): network shapes / parameter counts of the
reference architecture, the PPO optimise loop, checkpoint round trip, and the multi-rank
path (flat gradient all-reduce, cross-rank advantage normalisation, common KL early stop)
with two gloo processes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from marl_amd.agents import Actor, Critic, FlatGradSync, PPOAgent, normalise_advantages


def test_parameter_counts_match_reference_architecture():
    # actor 768+256+16512+256+258+258+516 = 18824 (agents/actor.py:86-100); critic 17409 (critic.py:71-77)
    assert sum(p.numel() for p in Actor().parameters()) == 18824
    assert sum(p.numel() for p in Critic().parameters()) == 17409


def test_actor_sample_and_evaluate_shapes():
    torch.manual_seed(0)
    a = Actor(log_std_min=-1.0, log_std_max=0.5)
    obs = torch.rand(7, 4, 5)
    act, logp, ent = a.sample(obs)
    assert act.shape == (7, 4, 3) and logp.shape == (7, 4) and ent.shape == (7, 4)
    assert act[..., :2].abs().max() <= 5.0 and set(act[..., 2].unique().tolist()) <= {0.0, 1.0, 2.0, 3.0}
    lp2, ent2 = a.evaluate_actions(obs.reshape(-1, 5), act.reshape(-1, 3))
    clamped = act[..., :2].abs().max(dim=-1).values.reshape(-1) >= 5.0
    assert torch.allclose(lp2[~clamped], logp.reshape(-1)[~clamped], atol=1e-5)
    det, _, _ = a.sample(obs, deterministic=True)
    det2, _, _ = a.sample(obs, deterministic=True)
    assert torch.equal(det, det2)
    c = Critic()
    assert c.forward_pooled(obs.mean(dim=1)).shape == (7, 1) and c(obs).shape == (7, 1)


def _fake_rollout(n, seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand(n, 5, generator=g)
    act = torch.cat([torch.randn(n, 2, generator=g) * 2, torch.randint(0, 4, (n, 1), generator=g).float()], dim=1)
    old = torch.randn(n, generator=g) * 0.1 - 3.0
    adv = torch.randn(n, generator=g)
    ret = torch.randn(n, generator=g)
    return obs, act, old, adv, ret


def test_optimise_runs_and_improves_value_fit():
    agent = PPOAgent(device="cpu", batch_size=64, n_epochs=4, seed=3)
    obs, act, old, adv, ret = _fake_rollout(512, 1)
    with torch.no_grad():
        old = agent.actor.evaluate_actions(obs, act)[0]
        before = ((agent.critic.forward_pooled(obs).squeeze(-1) - ret) ** 2).mean().item()
    stats = agent.optimise(obs, act, old, adv, ret)
    assert set(stats) == {"actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl"}
    assert all(np.isfinite(v) for v in stats.values())
    with torch.no_grad():
        after = ((agent.critic.forward_pooled(obs).squeeze(-1) - ret) ** 2).mean().item()
    assert after < before


def test_numpy_single_env_api_and_checkpoint_roundtrip(tmp_path):
    agent = PPOAgent(device="cpu", M=4, seed=0)
    obs = np.random.RandomState(0).rand(4, 5).astype(np.float32)
    a, lp, v = agent.get_action(obs)
    assert a.shape == (4, 3) and lp.shape == (4,) and isinstance(v, float)
    agent.total_timesteps, agent.total_updates = 123, 4
    agent.save(tmp_path / "ck")
    other = PPOAgent.from_checkpoint(tmp_path / "ck", device="cpu")
    assert other.total_timesteps == 123 and other.total_updates == 4
    for p, q in zip(agent.actor.parameters(), other.actor.parameters()):
        assert torch.equal(p, q)
    d1, _, _ = agent.get_action(obs, deterministic=True)
    d2, _, _ = other.get_action(obs, deterministic=True)
    assert np.array_equal(d1, d2)


def test_schedule_setters():
    agent = PPOAgent(device="cpu")
    agent.set_learning_rate(1e-5, 2e-5)
    assert agent.actor_optimizer.param_groups[0]["lr"] == 1e-5 and agent.critic_optimizer.param_groups[0]["lr"] == 2e-5
    agent.set_entropy_coef(0.123)
    assert agent.c2 == 0.123
    agent.set_training_mode(False)
    assert not agent.actor.training


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # ranks start different, the agent broadcasts rank 0
    agent = PPOAgent(device="cpu", batch_size=4096, n_epochs=3)
    obs, act, old, adv, ret = _fake_rollout(256, 5)
    half = slice(rank * 128, (rank + 1) * 128)
    with torch.no_grad():
        old = agent.actor.evaluate_actions(obs, act)[0]
    adv_n = normalise_advantages(adv[half] * 3 + 1)    # moments must come from both ranks
    stats = agent.optimise(obs[half], act[half], old[half], adv[half], ret[half])
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out["params_equal"] = bool(torch.equal(gathered[0], gathered[1]))
        out["flat"] = flat.clone()
        out["stats"] = stats
    out[f"adv{rank}"] = adv_n
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out["params_equal"]
    # reference: one process, whole batch, same initial weights (seed 100 = rank 0's seed)
    torch.manual_seed(100)
    agent = PPOAgent(device="cpu", batch_size=4096, n_epochs=3)
    obs, act, old, adv, ret = _fake_rollout(256, 5)
    with torch.no_grad():
        old = agent.actor.evaluate_actions(obs, act)[0]
    stats = agent.optimise(obs, act, old, adv, ret)
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    assert torch.allclose(out["flat"], flat, atol=2e-5, rtol=1e-4)
    for k in stats:
        assert abs(stats[k] - out["stats"][k]) < 1e-4, k
    whole = normalise_advantages(adv * 3 + 1)
    assert torch.allclose(torch.cat([out["adv0"], out["adv1"]]), whole, atol=1e-5)


def test_flat_grad_sync_is_noop_without_process_group():
    m = torch.nn.Linear(3, 2)
    m(torch.ones(1, 3)).sum().backward()
    g = m.weight.grad.clone()
    FlatGradSync([m]).sync()
    assert torch.equal(m.weight.grad, g)


# ---------------------------------------------------------------- the reference's own update, replayed
def _load_ppo_golden():
    import golden_util as gu
    z = gu.load("ppo_update.npz")
    T, M, B, epochs = (int(x) for x in z["hyper"])
    return z, T, M, B, epochs


def _state_from(z, prefix, module):
    sd = {k: torch.from_numpy(z[f"{prefix}_{k}"].copy()) for k in module.state_dict().keys()}
    module.load_state_dict(sd)


def test_optimise_reproduces_reference_update():
    """One PPOAgent.update() of the reference (agents/ppo_agent.py:250-373; T=64, M=4, 2 epochs of 4
    minibatches) recorded by tests/golden/make_golden.py: same initial weights, same buffer, same
    shuffles -> same weights after the 8 optimiser steps and the same five statistics."""
    z, T, M, B, epochs = _load_ppo_golden()
    agent = PPOAgent(device="cpu", M=M, rollout_length=T, batch_size=B, n_epochs=epochs)
    _state_from(z, "a0", agent.actor)
    _state_from(z, "c0", agent.critic)
    obs = torch.from_numpy(z["buf_obs"]).reshape(-1, 5)
    act = torch.from_numpy(z["buf_actions"]).reshape(-1, 3)
    logp = torch.from_numpy(z["buf_log_probs"]).reshape(-1)
    adv = torch.from_numpy(z["adv_norm"]).reshape(-1)
    ret = torch.from_numpy(z["returns"]).reshape(-1)
    stats = agent.optimise(obs, act, logp, adv, ret, permutations=z["perms"])
    for k, v in agent.actor.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(z[f"a1_{k}"]), rtol=1e-5, atol=1e-7), k
        assert not torch.equal(v, torch.from_numpy(z[f"a0_{k}"])), k
    for k, v in agent.critic.state_dict().items():
        assert torch.allclose(v, torch.from_numpy(z[f"c1_{k}"]), rtol=1e-5, atol=1e-7), k
    want = dict(zip(("actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl"), z["stats"]))
    for k in want:
        assert abs(stats[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, stats[k], want[k])
    # the recorded log-probs are the ones this actor assigns to the recorded actions
    agent2 = PPOAgent(device="cpu", M=M)
    _state_from(z, "a0", agent2.actor)
    with torch.no_grad():
        lp, _ = agent2.actor.evaluate_actions(obs, act)
    unclamped = act[:, :2].abs().max(dim=1).values < 5.0
    assert torch.allclose(lp[unclamped], logp[unclamped], atol=1e-5)


def test_reference_checkpoint_loads_and_deploy_export_matches_minimal_actor(tmp_path):
    """ppo_agent.pt written by the reference's PPOAgent.save (float learning rates, capturable=False,
    numpy scalars in training_stats) loads with the weights-only loader; the exported actor state dict
    has exactly the keys and shapes of the reference's deployment network (deploy.py:33-92)."""
    import golden_util as gu
    z, T, M, B, epochs = _load_ppo_golden()
    import os
    agent = PPOAgent.from_checkpoint(os.path.join(gu.GOLDEN, "ref_ckpt"), device="cpu")
    assert agent.rollout_length == T and agent.batch_size == B and agent.total_updates == 1
    assert agent.total_timesteps == T and len(agent.training_stats["actor_loss"]) == 1
    assert isinstance(agent.training_stats["approx_kl"][0], float)
    for k, v in agent.actor.state_dict().items():
        assert torch.equal(v, torch.from_numpy(z[f"a1_{k}"])), k
    st = agent.actor_optimizer.state[next(iter(agent.actor.parameters()))]
    assert float(st["step"]) == 2 * (T * M // B) and st["exp_avg"].abs().sum() > 0
    # it keeps training after the load
    obs, act, old, adv, ret = _fake_rollout(256, 2)
    with torch.no_grad():
        old = agent.actor.evaluate_actions(obs, act)[0]
    assert all(np.isfinite(v) for v in agent.optimise(obs, act, old, adv, ret).values())
    sd = agent.export_actor(tmp_path / "deployment" / "actor_state_dict.pt")
    loaded = torch.load(tmp_path / "deployment" / "actor_state_dict.pt", weights_only=True)
    keys = [str(k) for k in z["deploy_keys"]]
    assert list(loaded.keys()) == keys
    for k, shp in zip(keys, z["deploy_shapes"]):
        assert list(loaded[k].shape) == [int(x) for x in shp[:loaded[k].dim()]], k
    # own checkpoints round-trip through the same loader with plain-float statistics
    agent.save(tmp_path / "own")
    again = PPOAgent.from_checkpoint(tmp_path / "own", device="cpu")
    assert again.training_stats == agent.training_stats


def test_saved_optimizer_state_loads_into_a_plain_adam_as_the_reference_does(tmp_path):
    """PPOAgent.load of the reference (ppo_agent.py:430-435) hands the saved optimiser state to load_state_dict
    of a plain torch.optim.Adam and goes on stepping: the file must hold float learning rates and
    capturable=False / fused=None groups, whatever device wrote it."""
    agent = PPOAgent(device="cpu", M=4, batch_size=64, n_epochs=1, seed=2)
    agent.optimise(*_fake_rollout(256, 1))
    agent.save(tmp_path / "ck")
    ck = torch.load(tmp_path / "ck" / "ppo_agent.pt", map_location="cpu", weights_only=True)
    for key, net in (("actor_optimizer_state_dict", Actor()), ("critic_optimizer_state_dict", Critic())):
        g = ck[key]["param_groups"][0]
        assert isinstance(g["lr"], float) and g["capturable"] is False and g["fused"] is None
        net.load_state_dict(ck[key.replace("_optimizer", "")])
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        opt.load_state_dict(ck[key])
        before = [p.detach().clone() for p in net.parameters()]
        for p in net.parameters():
            p.grad = torch.ones_like(p)
        opt.step()
        assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters()))
        assert float(next(iter(opt.state.values()))["step"]) == float(agent.actor_optimizer.state[next(iter(agent.actor.parameters()))]["step"]) + 1
