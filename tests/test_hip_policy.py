"""GPU tests of marl_policy_act (the rollout loop's policy inference, one launch for actor, critic, draws and
log-probabilities) against the torch networks it mirrors, and of the trainer pieces built on it: in-kernel
episode statistics, evaluation, checkpoint / resume."""
import json
import os

import numpy as np
import pytest
import torch

from marl_amd import BatchedJammerEnv, EnvConfig
from marl_amd.agents import PPOAgent
from marl_amd.trainer import BatchedTrainer, HISTORY_KEYS

pytestmark = pytest.mark.gpu


def _agent(hidden, M, seed=3):
    a = PPOAgent(device="cuda", M=M, hidden_dim=hidden, num_envs=1, rollout_length=2, seed=seed)
    # make every parameter non-trivial (LayerNorm weights are 1, biases 0 after init)
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for p in list(a.actor.parameters()) + list(a.critic.parameters()):
            p.add_(0.05 * torch.randn(p.shape, device="cuda", generator=g))
        for head in (a.actor.mu_head, a.actor.log_std_head, a.actor.band_head):
            head.weight.mul_(6.0)           # spread the means, reach the log-std clamps, uneven band logits
    a._packed_stale = True
    return a


@pytest.mark.parametrize("hidden,E,M", [(128, 37, 8), (64, 130, 4), (128, 1, 1), (128, 4096, 8)])
def test_policy_kernel_matches_torch_networks(hidden, E, M):
    agent = _agent(hidden, M)
    g = torch.Generator(device="cuda").manual_seed(1)
    obs = torch.rand(E, M, 5, device="cuda", generator=g)
    act = torch.empty(E, M, 3, device="cuda")
    logp = torch.empty(E, M, device="cuda")
    val = torch.empty(E, device="cuda")
    tick = torch.arange(E, dtype=torch.int32, device="cuda")
    with torch.no_grad():
        mu, log_std, logits = agent.actor(obs)
        want_val = agent.critic.forward_pooled(obs.mean(dim=1)).squeeze(-1)
    # deterministic: velocity mean (clamped), arg-max band, the critic's value
    agent.act_kernel(obs, act, logp, val, tick=tick, deterministic=True)
    torch.cuda.synchronize()
    assert torch.allclose(act[..., :2], mu.clamp(-5, 5), rtol=1e-4, atol=1e-4)
    top2 = logits.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(act[..., 2][clear].long(), logits.argmax(-1)[clear])
    assert torch.allclose(val, want_val, rtol=1e-4, atol=1e-4)
    # stochastic: the kernel's log-probability is the one the torch actor assigns to the kernel's own draw
    agent.act_kernel(obs, act, logp, val, tick=tick)
    torch.cuda.synchronize()
    assert float(act[..., :2].abs().max()) <= 5.0 and set(act[..., 2].unique().tolist()) <= {0.0, 1.0, 2.0, 3.0}
    inside = act[..., :2].abs().max(dim=-1).values < 5.0          # clamped draws: the unclamped value is not in `act`
    with torch.no_grad():
        lp, _ = agent.actor.evaluate_actions(obs.reshape(-1, 5), act.reshape(-1, 3))
    lp = lp.reshape(E, M)
    assert inside.float().mean() > 0.3
    assert torch.allclose(logp[inside], lp[inside], rtol=1e-4, atol=2e-4), float((logp - lp)[inside].abs().max())
    assert torch.allclose(val, want_val, rtol=1e-4, atol=1e-4)
    # a later step clock draws other actions; the same clock and call repeat them
    a1 = act.clone()
    agent.act_kernel(obs, act, logp, val, tick=tick)
    assert torch.equal(act, a1)
    agent.act_kernel(obs, act, logp, val, tick=tick + 1)
    assert not torch.equal(act, a1)
    agent.act_kernel(obs, act, logp, val)                 # no clock: the call counter keys the draws
    a2 = act.clone()
    agent.act_kernel(obs, act, logp, val)
    assert not torch.equal(act, a2)


def test_policy_kernel_draw_statistics():
    """(v - mu) / sigma of the unclamped draws is N(0, 1); band frequencies follow softmax(logits)."""
    E, M = 8192, 8
    agent = PPOAgent(device="cuda", M=M, num_envs=1, rollout_length=2, seed=5)     # fresh init: small means, sigma ~ 0.6
    with torch.no_grad():
        agent.actor.band_head.bias.copy_(torch.tensor([0.0, 1.0, -1.0, 0.5], device="cuda"))
    agent._packed_stale = True
    obs = torch.rand(E, M, 5, device="cuda")
    act = torch.empty(E, M, 3, device="cuda")
    logp = torch.empty(E, M, device="cuda")
    agent.act_kernel(obs, act, logp, None, tick=torch.zeros(E, dtype=torch.int32, device="cuda"))
    with torch.no_grad():
        mu, log_std, logits = agent.actor(obs)
    z = (act[..., :2] - mu) / log_std.exp()
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z[..., 0] * z[..., 1]).mean())) < 0.01                      # the two components are independent
    freq = torch.bincount(act[..., 2].long().reshape(-1), minlength=4).double() / (E * M)
    want = torch.softmax(logits, -1).double().mean(dim=(0, 1))
    assert float((freq - want).abs().max()) < 0.01
    # agents of one env and envs of one batch draw independently
    zz = z[..., 0]
    assert abs(float((zz[:, 0] * zz[:, 1]).mean())) < 0.05 and abs(float((zz[:-1, 0] * zz[1:, 0]).mean())) < 0.05


def test_library_minibatch_step_equals_torch_step_at_small_and_large_minibatches():
    """Every GPU agent takes the minibatch step through the library (marl_ppo_rows / products / apply), whatever
    the minibatch size; one epoch of it gives the weights and statistics of the torch autograd step, at the
    reference's 256 rows and at 8 192 rows."""
    for B, n in ((256, 2560), (8192, 32768)):
        a = PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=1, seed=1, fused_update=False)
        b = PPOAgent(device="cuda", M=4, batch_size=B, n_epochs=1, seed=1)
        assert b.fused_update and not a.fused_update
        g = torch.Generator(device="cuda").manual_seed(2)
        obs = torch.rand(n, 5, device="cuda", generator=g)
        act = torch.cat([torch.randn(n, 2, device="cuda", generator=g),
                         torch.randint(0, 4, (n, 1), device="cuda", generator=g).float()], 1)
        adv, ret = torch.randn(n, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)
        with torch.no_grad():
            old = a.actor.evaluate_actions(obs, act)[0]
        perm = [torch.randperm(n, generator=torch.Generator().manual_seed(3)).numpy()]
        sa = a.optimise(obs, act, old, adv, ret, permutations=perm)
        sb = b.optimise(obs, act, old, adv, ret, permutations=perm)
        for p, q in zip(list(a.actor.parameters()) + list(a.critic.parameters()),
                        list(b.actor.parameters()) + list(b.critic.parameters())):
            moved = 10 * 3e-3          # at most lr-sized moves per step
            assert float((p.detach() - q.detach()).abs().max()) <= 2e-2 * moved
        for k in sa:
            assert abs(sa[k] - sb[k]) <= 2e-4 * max(1.0, abs(sa[k])), k


def test_episode_statistics_from_the_step_kernels():
    """The four sums the step kernels accumulate (episodes, return, lambda2 reduction, length) against the
    same quantities recomputed from the rollout buffer and the env's outputs."""
    E, T, LEN = 96, 30, 7
    for N, M in ((20, 4), (100, 6)):
        cfg = EnvConfig.fast_preset(N=N, M=M, max_steps=LEN)
        tr = BatchedTrainer(cfg, num_envs=E, rollout_length=T, seed=3, use_graph=False)
        env, agent = tr.env, tr.agent
        obs, _ = env.reset()
        tr._obs = obs.clone()
        stats = torch.zeros(4, dtype=torch.float64, device="cuda")
        ret = torch.zeros(E, dtype=torch.float64, device="cuda")
        want = torch.zeros(4, dtype=torch.float64, device="cuda")
        for t in range(T):
            actions, rew_out, done_out, obs_out = agent.act_into_buffer(tr._obs, tick=env.views["tick"])
            lam0 = env.views["lambda2_initial"].clone()
            o, r, _, trunc, info = env.step(actions, out_obs=obs_out, out_reward=rew_out, out_truncated=done_out,
                                            episode_stats=stats)
            agent.commit_step()
            tr._obs.copy_(o)
            ret += r.double()
            d = trunc.double()
            red = 100.0 * (1.0 - info["lambda2_current"] / lam0)
            want += torch.stack([d.sum(), (ret * d).sum(), (red * d).sum(), (d * LEN).sum()])
            ret *= 1.0 - d
            assert torch.allclose(env.views["episode_return"], ret, rtol=1e-12, atol=1e-12)
        assert float(stats[0]) == E * (T // LEN) == float(want[0]) and float(stats[3]) == float(want[3])
        assert torch.allclose(stats, want, rtol=1e-9, atol=1e-9)


def test_trainer_logs_checkpoints_evaluates_and_resumes(tmp_path):
    """Class-level trainer features of the reference (training/trainer.py:326-458, 522-558; training/metrics.py:
    152-315): CSV / history / final_stats with the reference's column names, checkpoints with ppo_agent.pt +
    trainer_state.json + the deployment actor, evaluation, and a resume that continues bit-identically:
    two rollouts in one go == one rollout, save, load into fresh objects, one more rollout."""
    cfg = EnvConfig.fast_preset(N=20, M=4, max_steps=15)
    mk = lambda out: BatchedTrainer(cfg, num_envs=64, rollout_length=20, total_timesteps=10 ** 9, seed=11,
                                    agent_kwargs=dict(n_epochs=2), output_dir=out, save_interval=1, eval_interval=2,
                                    eval_episodes=1, use_graph=False)
    a = mk(str(tmp_path / "a"))
    fin = a.train(max_rollouts=2, log=None)
    assert fin["total_rollouts"] == 2 and fin["total_timesteps"] == 2 * 20 * 64
    out = tmp_path / "a"
    rows = open(out / "training_log.csv").read().strip().splitlines()
    assert rows[0].split(",") == list(HISTORY_KEYS) and len(rows) == 3
    hist = json.load(open(out / "history.json"))
    assert set(hist) == set(HISTORY_KEYS) and hist["timestep"] == [1280, 2560]
    # per-step info of the rollout loop (training/trainer.py:239-245, 293): real values, not the defaults
    assert all(-80.0 < x < 0.0 for x in hist["avg_jamming_power_dbm"]) and all(0.0 <= x <= 1.0 for x in hist["band_match_rate"])
    terms, power, buf = a._terms, a._power, a.agent.buffer
    assert abs(hist["avg_jamming_power_dbm"][-1] - float(power.mean())) < 1e-9
    assert abs(hist["band_match_rate"][-1] - float(terms[..., 1].double().mean())) < 1e-9
    # the band-match term of step t is the mean of feature 4 of the observation it produced, unless the env started over
    keep = ~buf.dones_u8.bool()
    assert torch.allclose(terms[:, :, 1][keep], buf.observations[1:, :, :, 4].mean(-1)[keep], atol=1e-6) and bool(keep.any())
    assert set(json.load(open(out / "final_stats.json"))) >= {"total_timesteps", "total_episodes", "total_rollouts",
                                                                 "best_lambda2_reduction", "time_elapsed", "fps"}
    for name in ("rollout_1", "rollout_2", "latest", "final"):
        d = out / "checkpoints" / name
        assert (d / "ppo_agent.pt").exists() and (d / "trainer_state.json").exists() and (d / "actor_state_dict.pt").exists()
        assert (d / "env_state_0.pt").exists()
    st = json.load(open(out / "checkpoints" / "rollout_1" / "trainer_state.json"))
    assert st["total_rollouts"] == 1 and st["total_timesteps"] == 1280 and "best_reduction" in st
    assert st["recent_reductions"] == list(a._reductions)[:1]
    ev = a.evaluate(n_episodes=2, num_envs=32)
    assert ev["episodes"] == 64 and ev["episode_length_mean"] == 15 and 0.0 <= ev["fragmentation_rate"] <= 1.0
    assert np.isfinite(ev["reward_mean"]) and ev["lambda2_reduction_std"] >= 0.0
    # resume from the first checkpoint in fresh objects and redo the second rollout
    c = mk(str(tmp_path / "c"))                 # the run to compare with: one rollout, then a second without any save
    c.train(max_rollouts=1, log=None)
    c.collect_rollout()
    c.update()
    b = mk(str(tmp_path / "b"))
    b.collect_rollout()                         # scrambles its own state first: the load must overwrite all of it
    b.load_checkpoint(str(out / "checkpoints" / "rollout_1"))      # also restores the device generator's state
    assert b.total_rollouts == 1 and b.total_timesteps == 1280
    b.collect_rollout()
    b.update()
    for k in ("enemy_x", "enemy_vy", "jam_x", "jam_y", "jam_band", "labels", "step_count", "tick", "episode_return"):
        assert torch.equal(b.env.views[k], c.env.views[k]), k
    for p, q in zip(list(b.agent.actor.parameters()) + list(b.agent.critic.parameters()),
                    list(c.agent.actor.parameters()) + list(c.agent.critic.parameters())):
        assert torch.equal(p, q)
    # a run that resumes in the same output directory keeps its log: rows are appended, the window of the
    # convergence test comes back
    e = mk(str(out))
    e.load_checkpoint(str(out / "checkpoints" / "latest"))
    assert list(e._reductions) == list(a._reductions) and e.total_rollouts == 2
    e.train(max_rollouts=3, log=None)
    rows = open(out / "training_log.csv").read().strip().splitlines()
    assert rows[0].split(",") == list(HISTORY_KEYS) and len(rows) == 4
    assert [int(float(r.split(",")[0])) for r in rows[1:]] == [1280, 2560, 3840]
    assert json.load(open(out / "history.json"))["timestep"] == [1280, 2560, 3840]
    # and the convergence test (training/trainer.py:422-458)
    t = mk(str(tmp_path / "d"))
    t.convergence_window, t.target_reduction = 3, 50.0
    t._reductions = type(t._reductions)([10.0, 20.0], maxlen=3)
    assert not t.check_convergence()
    t._reductions.append(60.0)
    assert not t.check_convergence() and t.best_reduction == 30.0
    t._reductions.extend([70.0, 80.0])
    assert t.check_convergence()
    t.disable_early_convergence = True
    assert not t.check_convergence()


def test_rollout_graph_replay_matches_eager_rollout():
    """The captured rollout (policy kernel + step kernels + masked reset, T steps) replayed == the same rollout
    launched eagerly: same buffer contents, same env state, same episode sums."""
    cfg = EnvConfig.fast_preset(N=50, M=8, max_steps=9)
    mk = lambda graph: BatchedTrainer(cfg, num_envs=128, rollout_length=16, seed=5, use_graph=graph,
                                      agent_kwargs=dict(n_epochs=1))
    a, b = mk(False), mk(True)
    for r in range(3):
        ra, rb = a.collect_rollout(), b.collect_rollout()
        if r >= 1:
            assert b._graph is not None
        assert ra["episodes_completed"] == rb["episodes_completed"] > 0
        assert abs(ra["mean_reward"] - rb["mean_reward"]) < 1e-9 and abs(ra["mean_lambda2_reduction"] - rb["mean_lambda2_reduction"]) < 1e-9
        assert ra["mean_jamming_power_dbm"] == rb["mean_jamming_power_dbm"] and ra["band_match_rate"] == rb["band_match_rate"]
        assert -80.0 < ra["mean_jamming_power_dbm"] < 0.0
        for name in ("observations", "actions", "log_probs", "rewards", "dones_u8", "values"):
            assert torch.equal(getattr(a.agent.buffer, name), getattr(b.agent.buffer, name)), (r, name)
        assert torch.equal(a.env.views["enemy_x"], b.env.views["enemy_x"])
        torch.manual_seed(100 + r); a.update()         # the shuffles come from the global device generator
        torch.manual_seed(100 + r); b.update()
        for p, q in zip(a.agent.actor.parameters(), b.agent.actor.parameters()):
            assert torch.equal(p, q)
