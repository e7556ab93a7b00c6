"""The library PPO step under torch.distributed: two ranks (two processes on the one GPU, gloo) on half batches end
with identical weights, equal to one process on the whole batch -- gradients go through one flat all-reduce between
marl_ppo_products and the optimiser (agents/ppo_agent.py:296-331 per rank)."""
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
