"""GPU tests of the device rollout buffer and the batched trainer: GAE through the C ABI
against the reference's reverse scan (oracle), in-place transition writes by the step
kernel, and two rollouts + PPO updates end to end."""
import numpy as np
import pytest
import torch

from marl_amd import BatchedJammerEnv, EnvConfig
from marl_amd.agents import DeviceRolloutBuffer, PPOAgent
from marl_amd.trainer import BatchedTrainer
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def test_device_buffer_gae_matches_reference_scan():
    T, E, M = 48, 33, 4
    rng = np.random.RandomState(2)
    buf = DeviceRolloutBuffer(T, E, M)
    rew = rng.normal(size=(T, E)).astype(np.float32)
    val = rng.normal(size=(T, E)).astype(np.float32)
    don = rng.rand(T, E) < 0.08
    for t in range(T):
        buf.add(np.zeros((E, M, 5), np.float32), np.zeros((E, M, 3), np.float32), np.zeros((E, M), np.float32),
                rew[t], don[t], val[t])
    assert len(buf) == T and buf.full
    lv = rng.normal(size=E).astype(np.float32)
    ld = rng.rand(E) < 0.5
    buf.compute_returns_and_advantages(torch.from_numpy(lv), torch.from_numpy(ld))
    adv_all = np.zeros((T, E), np.float32)
    for e in range(E):
        adv, ret = orc.gae(rew[:, e], val[:, e], don[:, e].astype(np.float32), float(lv[e]), bool(ld[e]), 0.99, 0.95)
        adv_all[:, e] = adv
        assert np.array_equal(buf.returns[:, e].cpu().numpy(), ret), e
    want = (adv_all - adv_all.mean()) / (adv_all.std() + 1e-8)
    np.testing.assert_allclose(buf.advantages.cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    obs, act, lp, adv, ret, v = buf.get_samples()
    assert obs.shape == (T * E * M, 5) and adv.shape == (T * E * M,)
    assert torch.equal(adv.reshape(T, E, M)[:, :, 0], buf.advantages)


def test_step_kernel_writes_transitions_in_place():
    E, T = 64, 12
    cfg = EnvConfig.fast_preset(N=10, M=4, max_steps=5)
    env = BatchedJammerEnv(E, cfg, seed=2, auto_reset=True)
    agent = PPOAgent(M=4, rollout_length=T, num_envs=E, device="cuda", seed=1)
    obs, _ = env.reset()
    for t in range(T):
        actions, rew_out, done_out, obs_out = agent.act_into_buffer(obs)
        obs, reward, _, trunc, info = env.step(actions, out_obs=obs_out, out_reward=rew_out, out_truncated=done_out)
        assert obs.data_ptr() == agent.buffer.observations[t + 1].data_ptr()
        assert reward.data_ptr() == agent.buffer.rewards[t].data_ptr()
        agent.commit_step()
        assert bool(trunc.all()) == ((t + 1) % 5 == 0)
    b = agent.buffer
    assert b.full and torch.all(b.dones_u8[4] == 1) and torch.all(b.dones_u8[3] == 0)
    assert float(b.rewards.abs().sum()) > 0 and float(b.observations[1:].abs().sum()) > 0
    stats = agent.update(obs, trunc)
    assert all(np.isfinite(v) for v in stats.values()) and len(agent.buffer) == 0


def test_batched_trainer_two_rollouts():
    cfg = EnvConfig.fast_preset(N=10, M=4, max_steps=20)
    tr = BatchedTrainer(cfg, num_envs=128, rollout_length=24, total_timesteps=10 ** 9, seed=0,
                        agent_kwargs=dict(batch_size=4096, n_epochs=2))
    out = tr.train(max_rollouts=2, log=None)
    assert out["total_rollouts"] == 2 and out["total_timesteps"] == 2 * 24 * 128
    assert tr.total_episodes == 2 * 128          # 48 steps of 20-step episodes: two ends per env
    assert np.isfinite(tr.agent.training_stats["actor_loss"][-1])


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """The N>1 code path of bench.py (rank sharding by env_offset, barrier, MAX-over-ranks timing,
    one JSON line from rank 0) with two gloo ranks on the one visible GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "5", "--envs-per-gpu", "512", "--backend", "gloo",
           "--share-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["value"] > 0 and abs(d["value"] - 2 * 512 * 20 / (d["ms_per_step"] * 20e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks of its own (here both on the one
    visible GPU, over gloo) and rank 0 prints one line for the two of them."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--envs-per-gpu", "512", "--backend", "gloo", "--share-device"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and "cpu_baseline" not in d
    assert len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"])
    assert abs(d["value"] - 2 * 512 * 20 / (d["ms_per_step"] * 20e-3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_barrier_and_timing_exchange_over_rccl():
    """The barrier and the exchange of the ranks' times as a multi-GPU run issues them -- backend nccl (RCCL), device
    tensors -- in a group of one rank, which is what one GPU can hold."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["MASTER_PORT"] = "29741"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
           "--envs-per-gpu", "512", "--force-dist", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and len(d["per_rank_ms_per_step"]) == 1
    assert d["per_rank_ms_per_step"][0] == pytest.approx(d["ms_per_step"])


def test_auto_reset_launch_is_skipped_only_when_nothing_can_be_done():
    """The masked auto-reset launch is left out while no step counter can have reached max_steps (host-side
    bounds).  An env that keeps the bounds and one that always launches must stay identical, also after
    single envs were reset out of lockstep."""
    E, T = 64, 7
    cfg = EnvConfig.fast_preset(N=12, M=3, max_steps=T)
    a = BatchedJammerEnv(E, cfg, seed=5, auto_reset=True)
    b = BatchedJammerEnv(E, cfg, seed=5, auto_reset=True)
    b._track_counts = False
    a.reset()
    b.reset()
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    for t in range(40):
        act = torch.empty(E, 3, 3, device="cuda")
        act[..., :2] = (torch.rand(E, 3, 2, device="cuda", generator=g) * 2 - 1) * 5
        act[..., 2] = torch.randint(0, 4, (E, 3), device="cuda", generator=g).float()
        oa, ra, _, ta, ia = a.step(act)
        ob, rb, _, tb, ib = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(ta, tb), t
        assert torch.equal(a.views["step_count"], b.views["step_count"]), t
        assert torch.equal(ia["episode_step"], ib["episode_step"]), t
        assert int(a.views["step_count"].max()) < T
        if t in (3, 17):                      # some envs start over early: the others must still truncate on time
            m = torch.zeros(E, dtype=torch.bool, device="cuda")
            m[t::5] = True
            a.reset(mask=m)
            b.reset(mask=m)
    assert a._count_bounds == [0, T - 1]          # out of lockstep: the launch stays in from then on


@pytest.mark.parametrize("fin,fout", [(5, 128), (128, 128), (128, 4), (128, 1)])
def test_tall_linear_gradients_match_nn_linear(fin, fout):
    """TallLinear splits the weight-gradient GEMM over row chunks: same values as nn.Linear."""
    from marl_amd.agents import TallLinear, _split_k
    rows = 32768
    assert _split_k(rows) == 64 and _split_k(1000) == 2 and _split_k(300) == 1 and _split_k(2048) == 8
    torch.manual_seed(0)
    a = TallLinear(fin, fout).cuda()
    b = torch.nn.Linear(fin, fout).cuda()
    b.load_state_dict(a.state_dict())
    x1 = torch.randn(rows, fin, device="cuda", requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = torch.randn(rows, fout, device="cuda")
    y1, y2 = a(x1), b(x2)
    y1.backward(g)
    y2.backward(g)
    assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-5, atol=1e-5)
    scale = float(b.weight.grad.abs().max())
    assert float((a.weight.grad - b.weight.grad).abs().max()) <= 2e-5 * scale
    assert float((a.bias.grad - b.bias.grad).abs().max()) <= 2e-5 * float(b.bias.grad.abs().max())


# ---------------------------------------------------------------- the reference's PPO update on the device
def _ppo_golden():
    import golden_util as gu
    z = gu.load("ppo_update.npz")
    T, M, B, epochs = (int(x) for x in z["hyper"])
    return z, T, M, B, epochs


def _load_state(z, prefix, module):
    module.load_state_dict({k: torch.from_numpy(z[f"{prefix}_{k}"].copy()) for k in module.state_dict().keys()})


@pytest.mark.parametrize("fused", [False, True])
def test_update_reproduces_reference_update_on_device(fused):
    """PPOAgent.update of the reference (agents/ppo_agent.py:250-373) recorded with its weights before and
    after: the device agent (E = 1) is given the same weights, the same 64 transitions through
    DeviceRolloutBuffer.add, the same shuffles; GAE runs in marl_gae, the eight optimiser steps on the GPU
    (torch autograd + torch.optim, and the library's marl_ppo_* step).  Statistics to 1e-4.  Weights: Adam divides
    the gradient by its running magnitude, so in these first steps an element whose gradient is near zero
    moves by up to the learning rate in a direction that float32 summation order decides; the bar is
    therefore on the update itself -- every element within 1 % of the tensor's largest change, and the
    mean deviation below 0.1 % of it."""
    z, T, M, B, epochs = _ppo_golden()
    agent = PPOAgent(device="cuda", M=M, rollout_length=T, batch_size=B, n_epochs=epochs, num_envs=1,
                     fused_update=fused)
    assert agent.fused_update == fused
    _load_state(z, "a0", agent.actor)
    _load_state(z, "c0", agent.critic)
    for t in range(T):
        agent.store_transition(z["buf_obs"][t][None], z["buf_actions"][t][None], z["buf_log_probs"][t][None],
                               z["buf_rewards"][t:t + 1], z["buf_dones"][t:t + 1] > 0, z["buf_values"][t:t + 1])
    stats = agent.update(torch.from_numpy(z["last_obs"]).cuda()[None], torch.zeros(1, dtype=torch.bool, device="cuda"),
                         permutations=z["perms"])
    for prefix, net in (("a1", agent.actor), ("c1", agent.critic)):
        for k, v in net.state_dict().items():
            want = torch.from_numpy(z[f"{prefix}_{k}"]).cuda()
            moved = float((want - torch.from_numpy(z[f"{prefix[0]}0_{k}"]).cuda()).abs().max())
            err = (v - want).abs()
            assert float(err.max()) <= 1e-2 * moved + 1e-7, (k, float(err.max()), moved)
            assert float(err.mean()) <= 1e-3 * moved + 1e-8, (k, float(err.mean()), moved)
    want = dict(zip(("actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl"), z["stats"]))
    for k in want:
        assert abs(stats[k] - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (k, stats[k], want[k])


def test_gpu_checkpoint_resume_and_reference_checkpoint_then_update(tmp_path):
    """save -> from_checkpoint -> update() on the GPU (the loaded optimiser must again be capturable, with
    device learning rates and step counters), and the same from a checkpoint the reference wrote."""
    import os
    import golden_util as gu
    obs = torch.rand(2048, 5, device="cuda")
    act = torch.cat([torch.randn(2048, 2, device="cuda"), torch.randint(0, 4, (2048, 1), device="cuda").float()], 1)
    adv, ret = torch.randn(2048, device="cuda"), torch.randn(2048, device="cuda")

    def run(agent, n):
        with torch.no_grad():
            old = agent.actor.evaluate_actions(obs, act)[0]
        g = torch.Generator(device="cuda").manual_seed(n)
        return agent.optimise(obs, act, old, adv, ret, generator=g)

    a = PPOAgent(device="cuda", M=4, batch_size=256, n_epochs=1, seed=3)
    run(a, 1)
    a.save(tmp_path / "ck")
    b = PPOAgent.from_checkpoint(tmp_path / "ck", device="cuda")
    assert isinstance(b.actor_optimizer.param_groups[0]["lr"], torch.Tensor)
    assert b.actor_optimizer.param_groups[0]["capturable"] is True
    sa, sb = run(a, 2), run(b, 2)
    for k in sa:
        assert abs(sa[k] - sb[k]) <= 1e-5 * max(1.0, abs(sa[k])), k
    for p, q in zip(a.actor.parameters(), b.actor.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7)
    b.set_learning_rate(1e-5, 2e-5)                       # in place: the library step reads the device scalar
    assert abs(float(b.actor_optimizer.param_groups[0]["lr"]) - 1e-5) < 1e-12
    run(b, 3)
    # GPU checkpoint on the CPU
    c = PPOAgent.from_checkpoint(tmp_path / "ck", device="cpu")
    assert c.actor_optimizer.param_groups[0]["capturable"] is False
    c.optimise(obs.cpu(), act.cpu(), torch.zeros(2048), adv.cpu(), ret.cpu())
    # the reference's file: float lr, capturable False, host step counters, numpy scalars in the statistics
    r = PPOAgent.from_checkpoint(os.path.join(gu.GOLDEN, "ref_ckpt"), device="cuda")
    assert r.actor_optimizer.param_groups[0]["capturable"] is True
    st = r.actor_optimizer.state[next(iter(r.actor.parameters()))]
    assert st["step"].is_cuda and float(st["step"]) == 8.0
    r.batch_size = 256
    out = run(r, 4)
    assert all(np.isfinite(v) for v in out.values())


# ---------------------------------------------------------------- BASELINE configs[4]: rollouts at N=50, M=8
def test_rollout_into_device_buffer_matches_oracle_n50_m8():
    """The collection loop of the batched trainer (policy on the device -> marl_env_step writing the
    transition into the buffer slot -> masked auto-reset) at the training configuration's swarm size,
    checked slot by slot against the oracle fed with the policy's actions and the noise the device
    generator draws; the episodes end inside the rollout, so the post-reset observation and the `done`
    flag of the slot are covered too."""
    E, N, M, T, LEN = 12, 50, 8, 26, 11
    cfg = EnvConfig.fast_preset(N=N, M=M, max_steps=LEN)
    p = orc.EnvParams.fast_preset(N=N, M=M)
    p.max_steps = LEN
    env = BatchedJammerEnv(E, cfg, seed=9, auto_reset=True)
    agent = PPOAgent(M=M, rollout_length=T, num_envs=E, device="cuda", seed=4)
    obs, _ = env.reset()
    torch.cuda.synchronize()
    v = env.views

    def oracle_reset(e, st):
        epos = np.stack([v["enemy_x"][e].cpu().numpy(), v["enemy_y"][e].cpu().numpy()], 1)
        evel = np.stack([v["enemy_vx"][e].cpu().numpy(), v["enemy_vy"][e].cpu().numpy()], 1)
        jpos = np.stack([v["jam_x"][e].cpu().numpy(), v["jam_y"][e].cpu().numpy()], 1)
        return orc.env_reset(p, st, epos, evel, int(v["enemy_band"][e]), jpos, v["jam_band"][e].cpu().numpy())[0]

    states = [orc.EnvState.empty(p) for _ in range(E)]
    for e in range(E):
        np.testing.assert_allclose(obs[e].cpu().numpy(), oracle_reset(e, states[e]), rtol=1e-5, atol=1e-6)
    ends = 0
    for t in range(T):
        noise = env.fill_noise().cpu().numpy()
        actions, rew_out, done_out, obs_out = agent.act_into_buffer(obs)
        obs, reward, _, trunc, info = env.step(actions, out_obs=obs_out, out_reward=rew_out, out_truncated=done_out)
        agent.commit_step()
        torch.cuda.synchronize()
        b = agent.buffer
        act = b.actions[t].cpu().numpy()
        assert act[..., :2].min() >= -5.0 and act[..., :2].max() <= 5.0
        for e in range(E):
            o, r, _, tr, inf = orc.env_step(p, states[e], act[e], noise[e], want_avg_power=False)
            assert abs(float(b.rewards[t, e]) - r) <= 1e-5 * max(1.0, abs(r)), (t, e)
            assert bool(b.dones_u8[t, e]) == tr
            assert abs(float(info["lambda2_current"][e]) - inf["lambda2_current"]) <= 1e-5 * max(1.0, abs(inf["lambda2_current"]))
            if tr:                                        # the slot holds the first observation of the next episode
                ends += 1
                assert int(v["step_count"][e]) == 0
                o = oracle_reset(e, states[e])
            else:
                assert np.array_equal(v["enemy_x"][e].cpu().numpy(), states[e].enemy_pos[:, 0])
                assert np.array_equal(v["labels"][e].cpu().numpy(), states[e].clusters.labels)
            np.testing.assert_allclose(b.observations[t + 1, e].cpu().numpy(), o, rtol=1e-5, atol=1e-6)
    assert ends == 2 * E and agent.buffer.full
    stats = agent.update(obs, trunc)
    assert all(np.isfinite(x) for x in stats.values())
