#!/usr/bin/env python3
"""Intermediates of the captured minibatch step vs the eager one (large batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd import agents as ag

B = int(os.environ.get("B", 262144))
ag._split_k = lambda rows: 1
dbg = {}

def fb(self, b_obs, b_act, b_old, b_adv, b_ret, tag):
    log_probs, entropy = self.actor.evaluate_actions(b_obs, b_act)
    values = self.critic.forward_pooled(b_obs).squeeze(-1)
    ratio = torch.exp(log_probs - b_old)
    surr1 = ratio * b_adv
    surr2 = torch.clamp(ratio, 1 - self.clip_eps, 1 + self.clip_eps) * b_adv
    actor_loss = -torch.min(surr1, surr2).mean()
    critic_loss = self.c1 * ((values - b_ret) ** 2).mean()
    entropy_loss = -self._c2_t * entropy.mean()
    for k, v in (("lp0", log_probs), ("ent0", entropy), ("ratio0", ratio)):
        dbg.setdefault(tag + k, torch.zeros_like(v)).copy_(v.detach())
    self.actor_optimizer.zero_grad(set_to_none=False)
    self.critic_optimizer.zero_grad(set_to_none=False)
    (actor_loss + entropy_loss).backward()
    critic_loss.backward()
    with torch.no_grad():
        for k, v in (("lp1", log_probs), ("ent1", entropy), ("ratio1", ratio)):
            dbg.setdefault(tag + k, torch.zeros_like(v)).copy_(v)
        kl = ((ratio - 1) - torch.log(ratio)).mean()
        return torch.stack([actor_loss, critic_loss, entropy.mean(),
                            (torch.abs(ratio - 1) > self.clip_eps).float().mean(), kl]).double()

a = ag.PPOAgent(device="cuda", M=8, batch_size=B, seed=1, graph_update=False)
b = ag.PPOAgent(device="cuda", M=8, batch_size=B, seed=1, graph_update=True)
g = torch.Generator(device="cuda").manual_seed(5)
graph = None
for step in range(5):
    obs = torch.rand(B, 5, device="cuda", generator=g)
    act = torch.cat([torch.randn(B, 2, device="cuda", generator=g), torch.randint(0, 4, (B, 1), device="cuda", generator=g).float()], 1)
    adv, ret = torch.randn(B, device="cuda", generator=g), torch.randn(B, device="cuda", generator=g)
    with torch.no_grad():
        old = a.actor.evaluate_actions(obs, act)[0] + 0.05 * torch.randn(B, device="cuda", generator=g)
    batch = (obs, act, old, adv, ret)
    oa = fb(a, *batch, "a_").clone(); a._apply_gradients()
    if step < 2:
        ob = fb(b, *batch, "w_").clone(); b._apply_gradients()
    else:
        if graph is None:
            static = [torch.empty_like(t) for t in batch]
            out = torch.zeros(5, dtype=torch.float64, device="cuda")
            for k in list(dbg):
                if k.startswith("a_"):
                    dbg["b_" + k[2:]] = torch.zeros_like(dbg[k])
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out.copy_(fb(b, *static, "b_"))
                b._apply_gradients()
        for d, t in zip(static, batch):
            d.copy_(t)
        graph.replay()
        ob = out.clone()
        torch.cuda.synchronize()
        print(f"step {step}: stats eager {[round(float(x),4) for x in oa]} graph {[round(float(x),4) for x in ob]}")
        for k in ("lp0", "ent0", "ratio0", "lp1", "ent1", "ratio1"):
            d = (dbg["a_" + k] - dbg["b_" + k]).abs()
            print(f"    {k}: max diff {float(d.max()):.3e}  mean(graph)={float(dbg['b_' + k].mean()):.5f} mean(eager)={float(dbg['a_' + k].mean()):.5f}", flush=True)
