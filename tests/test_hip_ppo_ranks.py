"""The library PPO step under torch.distributed: two ranks (two processes on the one GPU, gloo) on half batches end
with identical weights, equal to one process on the whole batch -- gradients go through one flat all-reduce between
marl_ppo_products and the optimiser (agents/ppo_agent.py:296-331 per rank); and a whole training run over a group of
one rank on RCCL, the backend a multi-GPU job uses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from marl_amd.agents import PPOAgent

pytestmark = pytest.mark.gpu

N_ROWS, EPOCHS = 8192, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rows(n, seed):
    g = torch.Generator().manual_seed(seed)
    obs = torch.rand(n, 5, generator=g)
    act = torch.cat([torch.randn(n, 2, generator=g) * 2, torch.randint(0, 4, (n, 1), generator=g).float()], dim=1)
    adv = torch.randn(n, generator=g)
    ret = torch.randn(n, generator=g)
    return [t.cuda() for t in (obs, act, adv, ret)]


def _run(agent, sl):
    obs, act, adv, ret = _rows(N_ROWS, 5)
    with torch.no_grad():
        old = agent.actor.evaluate_actions(obs, act)[0] + 0.1 * torch.sin(torch.arange(N_ROWS, device="cuda").float())
    stats = agent.optimise(obs[sl].contiguous(), act[sl].contiguous(), old[sl].contiguous(), adv[sl].contiguous(),
                           ret[sl].contiguous())
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    return stats, flat


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # ranks start different, the agent broadcasts rank 0's weights
    agent = PPOAgent(device="cuda", batch_size=N_ROWS, n_epochs=EPOCHS, fused_update=True)
    agent.target_kl = 1e9
    half = N_ROWS // world
    stats, flat = _run(agent, slice(rank * half, (rank + 1) * half))
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out["params_equal"] = bool(torch.equal(gathered[0], gathered[1]))
        out["flat"] = flat.cpu()
        out["stats"] = stats
    dist.destroy_process_group()


def test_two_ranks_on_half_batches_equal_one_process_on_the_whole_batch():
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out["params_equal"]
    torch.manual_seed(100)
    agent = PPOAgent(device="cuda", batch_size=N_ROWS, n_epochs=EPOCHS, fused_update=True)
    agent.target_kl = 1e9
    torch.manual_seed(100)
    start = PPOAgent(device="cuda", batch_size=N_ROWS, n_epochs=EPOCHS, fused_update=True)
    before = torch.cat([p.detach().reshape(-1) for p in list(start.actor.parameters()) + list(start.critic.parameters())]).cpu()
    stats, flat = _run(agent, slice(0, N_ROWS))
    moved = (flat.cpu() - before).abs().max()
    assert moved > 1e-4
    assert (out["flat"] - flat.cpu()).abs().max() <= 0.02 * moved + 1e-7
    for k in stats:
        assert abs(stats[k] - out["stats"][k]) < 2e-3 * max(1.0, abs(stats[k])), k


def _trainer_run(out_key, out):
    from marl_amd import EnvConfig
    from marl_amd.trainer import BatchedTrainer
    cfg = EnvConfig.fast_preset(N=10, M=4, max_steps=20)
    tr = BatchedTrainer(cfg, num_envs=128, rollout_length=24, total_timesteps=10 ** 9, seed=0,
                        agent_kwargs=dict(batch_size=1024, n_epochs=2))
    fin = tr.train(max_rollouts=2, log=None)
    ev = tr.evaluate(n_episodes=1, num_envs=64)
    flat = torch.cat([p.detach().reshape(-1) for p in list(tr.agent.actor.parameters()) + list(tr.agent.critic.parameters())])
    out[out_key] = dict(flat=flat.cpu(), episodes=tr.total_episodes, timesteps=fin["total_timesteps"],
                        eval={k: float(v) for k, v in ev.items() if isinstance(v, (int, float))},
                        kl=float(tr.agent.training_stats["approx_kl"][-1]))
    return tr


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out["backend"] = dist.get_backend()
    tr = _trainer_run("rccl", out)
    # the step's own all-reduce is skipped when the group has one rank: the same call on the same vector here
    g = tr.agent._fz["gflat"]
    before = g.clone()
    dist.all_reduce(g)
    torch.cuda.synchronize()
    out["gflat_ok"] = bool(torch.equal(g, before)) and bool(g.abs().sum() > 0) and g.dtype == torch.float32
    dist.destroy_process_group()


def test_training_over_a_one_rank_rccl_group_equals_training_without_a_group():
    """Every collective of training (weight broadcast, the flat gradient all-reduce, advantage moments, the epoch's KL,
    the rollout and evaluation metrics) issued on device tensors through RCCL itself -- a group of one rank is what
    one GPU can hold -- leaves the run as it is without a group."""
    port = _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    assert out["backend"] == "nccl" and out["gflat_ok"]
    _trainer_run("alone", out)
    a, b = out["rccl"], out["alone"]
    assert a["episodes"] == b["episodes"] and a["timesteps"] == b["timesteps"] == 2 * 24 * 128
    assert torch.equal(a["flat"], b["flat"])
    assert a["kl"] == b["kl"] and a["eval"].keys() == b["eval"].keys() and len(a["eval"]) > 0
    for k in a["eval"]:
        assert a["eval"][k] == pytest.approx(b["eval"][k], rel=1e-12, abs=1e-12), k
