#!/usr/bin/env python3
"""Per-step check of the two captured graphs: weights and Adam state of the graphed agent are overwritten
with the eager agent's before every step, so each replay starts from identical state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import agents as ag

B = int(os.environ.get("B", 262144))
if os.environ.get("NOTALL"):
    ag._split_k = lambda rows: 1
a = ag.PPOAgent(device="cuda", M=8, batch_size=B, seed=1, graph_update=False)
b = ag.PPOAgent(device="cuda", M=8, batch_size=B, seed=1, graph_update=True)
g = torch.Generator(device="cuda").manual_seed(5)

def sync_state():
    with torch.no_grad():
        for p, q in zip(list(a.actor.parameters()) + list(a.critic.parameters()),
                        list(b.actor.parameters()) + list(b.critic.parameters())):
            q.copy_(p)
        for oa, ob in ((a.actor_optimizer, b.actor_optimizer), (a.critic_optimizer, b.critic_optimizer)):
            for (pa, sa), (pb, sb) in zip(oa.state.items(), ob.state.items()):
                for k in sa:
                    sb[k].copy_(sa[k])

for step in range(7):
    obs = torch.rand(B, 5, device="cuda", generator=g)
    act = torch.cat([torch.randn(B, 2, device="cuda", generator=g), torch.randint(0, 4, (B, 1), device="cuda", generator=g).float()], 1)
    adv, ret = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
    with torch.no_grad():
        old = a.actor.evaluate_actions(obs, act)[0] + 0.05 * torch.randn(B, device="cuda", generator=g)
    batch = (obs, act, old, adv, ret)
    if step >= 2:
        sync_state()
    oa = a._forward_backward(*batch).clone()
    ga = [p.grad.clone() for p in list(a.actor.parameters()) + list(a.critic.parameters())]
    a._apply_gradients()
    if step < 2:
        ob = b._minibatch_step(*batch).clone()
        b._eager_steps += 1
        continue
    if b._step_graph is None:
        b._graphed_step(batch)          # capture + first replay; redo from the synced state below
        sync_state()
    for d, t in zip(b._step_static, batch):
        d.copy_(t)
    b._step_graph[0].replay()
    torch.cuda.synchronize()
    gb = [p.grad.clone() for p in list(b.actor.parameters()) + list(b.critic.parameters())]
    ob = b._step_out.clone()
    b._step_graph[1].replay()
    torch.cuda.synchronize()
    gd = max(float((x - y).abs().max() / (x.abs().max() + 1e-12)) for x, y in zip(ga, gb))
    with torch.no_grad():
        wd = max(float((p - q).abs().max()) for p, q in zip(list(a.actor.parameters()) + list(a.critic.parameters()),
                                                               list(b.actor.parameters()) + list(b.critic.parameters())))
    print(f"B={B} step {step}: stat diff {float((oa - ob).abs().max()):.3e}  rel grad diff {gd:.3e}  weight diff after Adam {wd:.3e}", flush=True)
