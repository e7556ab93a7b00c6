#!/usr/bin/env python3
"""Time of the fused PPO minibatch step and of its parts at the config-5 minibatch (262 144 rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from marl_amd.agents import PPOAgent

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = 4 * B
agent = PPOAgent(device="cuda", M=8, batch_size=B, seed=1, fused_update=True)
g = torch.Generator(device="cuda").manual_seed(1)
obs = torch.randn(n, 5, device="cuda", generator=g)
act = torch.randn(n, 3, device="cuda", generator=g)
act[:, 2] = torch.randint(0, 4, (n,), device="cuda", generator=g).float()
with torch.no_grad():
    old, _ = agent.actor.evaluate_actions(obs, act)
adv = torch.randn(n, device="cuda", generator=g)
ret = torch.randn(n, device="cuda", generator=g)
idx = torch.randperm(n, device="cuda")[:B].contiguous()


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("fused forward/backward  %.3f ms" % timed(lambda: agent._fused_forward_backward(obs, act, old, adv, ret, idx)))
print("  + torch clip, Adam    %.3f ms" % timed(lambda: (agent._fused_forward_backward(obs, act, old, adv, ret, idx), agent._apply_gradients())))
print("step as in optimise()   %.3f ms" % timed(lambda: (agent._fused_forward_backward(obs, act, old, adv, ret, idx, stats_row=0), agent._fused_apply()), reps=40))
z = agent._fz
import ctypes as C
from marl_amd import _lib
w = z["packed"]
io = _lib.MarlPpoIO()
io.weights = w.data_ptr(); io.obs, io.actions, io.old_log_probs = obs.data_ptr(), act.data_ptr(), old.data_ptr()
io.advantages, io.returns, io.index = adv.data_ptr(), ret.data_ptr(), idx.data_ptr()
for k in ("xe", "y1e", "y2e", "dz1", "dz2", "actor_sums", "dh", "cy1e", "cy2e", "cdz1", "cdz2", "dv"):
    setattr(io, k, z[k].data_ptr())
io.stats = z["stats"][0].data_ptr()
io.clip_eps, io.c1, io.c2, io.v_max, io.log_std_min, io.log_std_max = 0.2, 0.5, 0.01, 5.0, -1.0, 0.5
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("marl_ppo_rows alone     %.3f ms" % timed(lambda: _lib.check(_lib.load().marl_ppo_rows(B, 128, C.byref(io), st))))
io.index = None
print("  rows in order         %.3f ms" % timed(lambda: _lib.check(_lib.load().marl_ppo_rows(B, 128, C.byref(io), st))))
print("pack                    %.3f ms" % timed(lambda: z["packer"].pack()))
if os.environ.get("PPO_PHASES"):          # library built with the cycle stamps of the phase experiment
    io.index = idx.data_ptr()
    _lib.check(_lib.load().marl_ppo_rows(B, 128, C.byref(io), st)); torch.cuda.synchronize()
    names = ["inputs", "layer 1", "normalise 1", "affine + relu + store y1", "layer 2", "LN 2 + store y2", "heads", "losses",
             "heads^T + LN 2 backward", "store dz2 + W2^T product", "layer 1 again + normalise", "LN 1 backward + store dz1"]
    v = z["dh"].view(-1)[:13].cpu().tolist()
    print("actor tile phases (cycles, one wave; %d tiles):" % int(v[12]))
    for n_, c in zip(names, v):
        print("   %-40s %9.0f" % (n_, c))
    print("   %-40s %9.0f" % ("sum", sum(v[:12])))
