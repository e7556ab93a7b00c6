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
