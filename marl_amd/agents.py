"""Actor / Critic networks and the batched PPO agent.

Same architecture, initialisation, losses and quirks as the reference
(agents/actor.py:86-116, agents/critic.py:71-91, agents/ppo_agent.py:193-373);
the two MLPs stay stock PyTorch (36 k parameters: nothing to write a kernel
for).  What changes is the data path: observations, actions, rewards and
values never leave the GPU, E environments are acted on at once, the rollout
lives in `DeviceRolloutBuffer`, GAE runs in libmarl_env.so, and with more than
one rank the gradients of both networks travel in ONE flat all-reduce per
optimiser step (RCCL over xGMI; 145 KB, latency bound).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical, Normal

from . import _lib


def _split_k(rows: int) -> int:
    """Number of row chunks for the weight gradient of a tall batch: a power of two <= 64 that divides
    `rows` and leaves chunks of at least 256 rows (1 = no split)."""
    s = 64
    while s > 1 and (rows % s or rows // s < 256):
        s //= 2
    return s


class _TallLinearFn(torch.autograd.Function):
    """y = x W^T + b for x [B, in] with B in the tens of thousands.  The weight gradient dY^T X is a GEMM
    with a 128 x 128 (or smaller) result and K = B: left to the library as one call it runs on a handful
    of workgroups (100 us at B = 32 768, half of the PPO update's time).  It is issued as a batched GEMM
    over row chunks instead and the partial results are summed: the same arithmetic on all CUs."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        rows = x.shape[0]
        gx = gy @ w if ctx.needs_input_grad[0] else None
        s = _split_k(rows)
        if s > 1:
            gyc, xc = gy.view(s, rows // s, gy.shape[1]), x.view(s, rows // s, x.shape[1])
            gw = torch.bmm(gyc.transpose(1, 2), xc).sum(0)
        else:
            gw = gy.t() @ x
        return gx, gw, gy.sum(0)


class TallLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose backward splits the weight-gradient GEMM
    over row chunks when the batch is tall."""

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda and x.is_contiguous() and torch.is_grad_enabled() and _split_k(x.shape[0]) > 1:
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


class Actor(nn.Module):
    """agents/actor.py:56-247: 5 -> hidden -> hidden (LayerNorm + ReLU) trunk, heads for
    velocity mean (2), clamped log-std (2) and band logits (num_bands)."""

    def __init__(self, obs_dim=5, hidden_dim=128, v_max=5.0, num_bands=4, log_std_min=-2.0, log_std_max=2.0):
        super().__init__()
        self.obs_dim, self.hidden_dim, self.v_max, self.num_bands = obs_dim, hidden_dim, v_max, num_bands
        self.log_std_min, self.log_std_max = log_std_min, log_std_max
        self.trunk = nn.Sequential(TallLinear(obs_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(),
                                   TallLinear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU())
        self.mu_head = TallLinear(hidden_dim, 2)
        self.log_std_head = TallLinear(hidden_dim, 2)
        self.band_head = TallLinear(hidden_dim, num_bands)
        for m in self.modules():                                    # actor.py:105-116
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
                nn.init.constant_(m.bias, 0.0)
        nn.init.orthogonal_(self.mu_head.weight, gain=0.01)
        nn.init.orthogonal_(self.log_std_head.weight, gain=0.01)
        nn.init.constant_(self.log_std_head.bias, -0.5)
        nn.init.orthogonal_(self.band_head.weight, gain=0.01)

    def forward(self, obs):
        f = self.trunk(obs)
        return self.mu_head(f), torch.clamp(self.log_std_head(f), self.log_std_min, self.log_std_max), self.band_head(f)

    def sample(self, obs, deterministic=False, generator: Optional[torch.Generator] = None):
        """actor.py:146-202.  obs [..., obs_dim] -> actions [..., 3], log_prob [...], entropy [...]."""
        mu, log_std, logits = self.forward(obs)
        std = torch.exp(log_std)
        # validate_args=False: the argument checks read a flag back to the host at every call
        vdist, bdist = Normal(mu, std, validate_args=False), Categorical(logits=logits, validate_args=False)
        if deterministic:
            velocity, band = mu, torch.argmax(logits, dim=-1)
        else:
            if generator is None:
                velocity, band = vdist.rsample(), bdist.sample()
            else:
                velocity = mu + std * torch.randn(mu.shape, device=mu.device, dtype=mu.dtype, generator=generator)
                band = torch.multinomial(bdist.probs.reshape(-1, self.num_bands), 1,
                                         generator=generator).reshape(logits.shape[:-1])
        log_prob = vdist.log_prob(velocity).sum(dim=-1) + bdist.log_prob(band)
        entropy = vdist.entropy().sum(dim=-1) + bdist.entropy()
        actions = torch.cat([torch.clamp(velocity, -self.v_max, self.v_max), band.unsqueeze(-1).float()], dim=-1)
        return actions, log_prob, entropy

    def evaluate_actions(self, obs, actions):
        """actor.py:204-247."""
        mu, log_std, logits = self.forward(obs)
        vdist = Normal(mu, torch.exp(log_std), validate_args=False)
        bdist = Categorical(logits=logits, validate_args=False)
        velocity = torch.clamp(actions[..., :2], -self.v_max * 2, self.v_max * 2)
        band = actions[..., 2].long()
        log_prob = vdist.log_prob(velocity).sum(dim=-1) + bdist.log_prob(band)
        entropy = vdist.entropy().sum(dim=-1) + bdist.entropy()
        return log_prob, entropy


class Critic(nn.Module):
    """agents/critic.py:53-134: 5 -> hidden -> hidden -> 1 on the mean-pooled observation."""

    def __init__(self, obs_dim=5, hidden_dim=128):
        super().__init__()
        self.obs_dim, self.hidden_dim = obs_dim, hidden_dim
        self.network = nn.Sequential(TallLinear(obs_dim, hidden_dim), nn.ReLU(), TallLinear(hidden_dim, hidden_dim),
                                     nn.ReLU(), TallLinear(hidden_dim, 1))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, gain=np.sqrt(2))
                nn.init.constant_(m.bias, 0.0)
        nn.init.orthogonal_(self.network[-1].weight, gain=1.0)

    def forward(self, obs):
        if obs.dim() == 3:
            obs = obs.mean(dim=1)
        elif obs.dim() == 1:
            obs = obs.unsqueeze(0)
        return self.network(obs)

    def forward_pooled(self, obs_pooled):
        return self.network(obs_pooled)


class DeviceRolloutBuffer:
    """RolloutBuffer (agents/rollout_buffer.py:38-279) for E environments, resident on the GPU.

    Layout [T, E, M, .]: slot t of `observations`, `rewards`, `dones` can be handed to
    `BatchedJammerEnv.step(out_obs=..., out_reward=..., out_truncated=...)` so that the step
    kernel writes the transition in place.  `compute_returns_and_advantages` runs the
    reference's reverse GAE scan (one lane per env, marl_gae) and its whole-buffer
    advantage normalisation, summed over all ranks when torch.distributed is initialised.
    """

    def __init__(self, buffer_size, num_envs, M=4, obs_dim=5, action_dim=3, gamma=0.99, gae_lambda=0.95,
                 device="cuda"):
        self.buffer_size, self.E, self.M = buffer_size, num_envs, M
        self.obs_dim, self.action_dim, self.gamma, self.gae_lambda = obs_dim, action_dim, gamma, gae_lambda
        self.device = _lib.normalise_device(device)
        T, E = buffer_size, num_envs
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)
        self.observations = z(T + 1, E, M, obs_dim)      # slot t+1 = observation after step t
        self.actions = z(T, E, M, action_dim)
        self.log_probs = z(T, E, M)
        self.rewards = z(T, E)
        self.dones_u8 = z(T, E, dt=torch.uint8)
        self.values = z(T, E)
        self.advantages = z(T, E)
        self.returns = z(T, E)
        self.pos, self.full, self.generator = 0, False, 0

    def reset(self):
        self.pos, self.full = 0, False
        self.generator += 1

    def __len__(self):
        return self.buffer_size if self.full else self.pos

    @property
    def size(self):
        return len(self)

    def add(self, obs, actions, log_probs, reward, done, value):
        """Copying form of rollout_buffer.py:120-149 (the in-place form is `slot`)."""
        t = self.pos
        self.observations[t].copy_(torch.as_tensor(obs, device=self.device).reshape(self.E, self.M, self.obs_dim))
        self.actions[t].copy_(torch.as_tensor(actions, device=self.device).reshape(self.E, self.M, self.action_dim))
        self.log_probs[t].copy_(torch.as_tensor(log_probs, device=self.device).reshape(self.E, self.M))
        self.rewards[t].copy_(torch.as_tensor(reward, device=self.device, dtype=torch.float32).reshape(self.E))
        self.dones_u8[t].copy_(torch.as_tensor(done, device=self.device).reshape(self.E).to(torch.uint8))
        self.values[t].copy_(torch.as_tensor(value, device=self.device, dtype=torch.float32).reshape(self.E))
        self.advance()

    def slot(self):
        """Views of the current slot for in-place writes: (obs_in, actions, log_probs, values,
        reward_out, done_out, obs_out)."""
        t = self.pos
        return (self.observations[t], self.actions[t], self.log_probs[t], self.values[t], self.rewards[t],
                self.dones_u8[t], self.observations[t + 1])

    def advance(self):
        self.pos += 1
        if self.pos == self.buffer_size:
            self.full = True

    def compute_returns_and_advantages(self, last_values: torch.Tensor, last_dones: torch.Tensor):
        """rollout_buffer.py:151-209 per env; normalisation over the whole buffer of all ranks."""
        T = len(self)
        lib = _lib.load()
        dones = self.dones_u8[:T].float().contiguous()
        lv = last_values.to(self.device, torch.float32).reshape(self.E).contiguous()
        ld = last_dones.to(self.device).float().reshape(self.E).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(lib.marl_gae(T, self.E, self.rewards.data_ptr(), self.values.data_ptr(), dones.data_ptr(),
                                    lv.data_ptr(), ld.data_ptr(), float(self.gamma), float(self.gae_lambda),
                                    self.advantages.data_ptr(), self.returns.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "marl_gae")
        self.advantages[:T] = normalise_advantages(self.advantages[:T])

    def get_samples(self):
        """rollout_buffer.py:211-229: flattened (T*E*M, .) views; advantage / return / value are
        shared by the M agents of an env."""
        T, E, M = len(self), self.E, self.M
        obs = self.observations[:T].reshape(-1, self.obs_dim)
        actions = self.actions[:T].reshape(-1, self.action_dim)
        log_probs = self.log_probs[:T].reshape(-1)
        adv = self.advantages[:T].unsqueeze(-1).expand(T, E, M).reshape(-1)
        ret = self.returns[:T].unsqueeze(-1).expand(T, E, M).reshape(-1)
        val = self.values[:T].unsqueeze(-1).expand(T, E, M).reshape(-1)
        return obs, actions, log_probs, adv, ret, val


def _fragment_order(H: int, device):
    """Index tables of the matrix-core layout marl_policy_act uses: unit[s, hl] = the hidden unit that a lane of
    wave half hl holds in its s-th accumulator register (block s // 16, register s % 16)."""
    s = torch.arange(H // 2, device=device)
    b, v = s // 16, s % 16
    hl = torch.arange(2, device=device)
    return (32 * b + 8 * (v // 4) + (v % 4))[:, None] + 4 * hl[None, :]          # [H/2, 2]


def _frag_first(w, b):
    """[H, 5], [H] -> [H/32][3][64]: lane l of step s of block bo carries [W | b][32 bo + l % 32][2 s + l // 32]."""
    dev = w.device
    NB = w.shape[0] // 32
    lane = torch.arange(64, device=dev)
    we = torch.cat([w.detach().float(), b.detach().float()[:, None]], 1)      # [H, 6]
    rows = (32 * torch.arange(NB, device=dev))[:, None, None] + (lane % 32)[None, None, :]
    cols = (2 * torch.arange(3, device=dev))[None, :, None] + (lane // 32)[None, None, :]
    return we[rows, cols].reshape(-1)


def _frag_hidden(w):
    """[32 * blocks, H] (out, in) -> [blocks][H/8][64][4]: lane l of step s of block bo carries
    W[32 bo + l % 32][unit(s, l // 32)]; four steps side by side."""
    dev = w.device
    w = w.detach().float()
    blocks, H = w.shape[0] // 32, w.shape[1]
    NS = H // 2
    lane = torch.arange(64, device=dev)
    unit = _fragment_order(H, dev)                                                                   # [NS, 2]
    rows = (32 * torch.arange(blocks, device=dev))[:, None, None] + (lane % 32)[None, None, :]      # [blocks, 1, 64]
    cols = unit[:, lane // 32][None, :, :]                                                           # [1, NS, 64]
    f = w[rows, cols]                                                                                # [blocks, NS, 64]
    return f.reshape(blocks, NS // 4, 4, 64).permute(0, 1, 3, 2).reshape(-1)


def _frag_heads_t(wh):
    """[8, H] -> [H/32][4][64]: lane l of step s of block b carries W_heads[2 s + l // 32][32 b + l % 32]."""
    dev = wh.device
    wh = wh.detach().float()
    NB = wh.shape[1] // 32
    lane = torch.arange(64, device=dev)
    rows = (2 * torch.arange(4, device=dev))[None, :, None] + (lane // 32)[None, None, :]
    cols = (32 * torch.arange(NB, device=dev))[:, None, None] + (lane % 32)[None, None, :]
    return wh[rows, cols].reshape(-1)


def _heads(actor):
    H = actor.hidden_dim
    w = torch.cat([actor.mu_head.weight, actor.log_std_head.weight, actor.band_head.weight], 0).detach().float()      # [8, H]
    w32 = torch.zeros(32, H, device=w.device)
    w32[:8] = w
    b = torch.cat([actor.mu_head.bias, actor.log_std_head.bias, actor.band_head.bias], 0).detach().float()
    return w, w32, b


def _check_packable(actor, critic):
    H = actor.hidden_dim
    if actor.num_bands != 4 or actor.obs_dim != 5 or critic.hidden_dim != H or H % 32:
        raise ValueError("the packed kernels take obs_dim 5, 4 bands and equal hidden sizes (multiples of 32) only")


def pack_policy_weights(actor: "Actor", critic: "Critic") -> torch.Tensor:
    """Actor and critic parameters as the flat float32 buffer marl_policy_act reads (layout: include/marl_env.h).
    Matrices are stored as the A-operand fragments of v_mfma_f32_32x32x2: for output block bo and step s,
    lane l carries W[32 bo + l % 32][k(s, l // 32)], where k is input 2 s + l // 32 for the first layer (the bias
    as input 5) and the unit the lane's s-th accumulator register holds for the hidden layer and the heads."""
    _check_packable(actor, critic)
    t, n = actor.trunk, critic.network
    _, heads_w32, heads_b = _heads(actor)
    dev = heads_b.device
    parts = [_frag_first(t[0].weight, t[0].bias), t[1].weight.detach(), t[1].bias.detach(),
             _frag_hidden(t[3].weight), t[3].bias.detach(), t[4].weight.detach(), t[4].bias.detach(),
             _frag_hidden(heads_w32), heads_b,
             _frag_first(n[0].weight, n[0].bias), _frag_hidden(n[2].weight), n[2].bias.detach(),
             n[4].weight.detach().reshape(-1), n[4].bias.detach(), torch.zeros(3, device=dev)]
    return torch.cat([p.float().reshape(-1) for p in parts]).contiguous()


def pack_train_weights(actor: "Actor", critic: "Critic") -> torch.Tensor:
    """The buffer marl_ppo_rows reads: the policy layout with the fragments of the transposed matrices added
    (heads^T after the head biases, W2^T at the end of each network; include/marl_env.h)."""
    _check_packable(actor, critic)
    t, n = actor.trunk, critic.network
    heads_w, heads_w32, heads_b = _heads(actor)
    dev = heads_b.device
    parts = [_frag_first(t[0].weight, t[0].bias), t[1].weight.detach(), t[1].bias.detach(),
             _frag_hidden(t[3].weight), t[3].bias.detach(), t[4].weight.detach(), t[4].bias.detach(),
             _frag_hidden(heads_w32), heads_b, _frag_heads_t(heads_w), _frag_hidden(t[3].weight.detach().t()),
             _frag_first(n[0].weight, n[0].bias), _frag_hidden(n[2].weight), n[2].bias.detach(),
             n[4].weight.detach().reshape(-1), n[4].bias.detach(), torch.zeros(3, device=dev),
             _frag_hidden(n[2].weight.detach().t())]
    return torch.cat([p.float().reshape(-1) for p in parts]).contiguous()


class TrainPacker:
    """pack_train_weights as ONE gather: the packing is a fixed permutation (with zero padding) of the flat parameter
    vector, found once by packing stand-in networks whose parameters hold their own flat positions."""

    def __init__(self, actor: "Actor", critic: "Critic"):
        import copy
        self.params = list(actor.parameters()) + list(critic.parameters())
        a2, c2 = copy.deepcopy(actor), copy.deepcopy(critic)
        pos = 1                                           # position 0 of the flat vector is a constant zero (padding)
        with torch.no_grad():
            for q in list(a2.parameters()) + list(c2.parameters()):
                q.copy_(torch.arange(pos, pos + q.numel(), device=q.device, dtype=torch.float32).view_as(q))
                pos += q.numel()
        if pos >= 1 << 24:
            raise ValueError("too many parameters for float32 positions")
        # positions counted from 1 in the parameter vector (marl_ppo_apply's `gather`); self.flat holds that vector from
        # element 4 on (16-byte aligned, padded to whole groups of four), behind the zero that the padding entries read
        at = pack_train_weights(a2, c2).round().long()
        self.n = pos - 1
        self.gather32 = at.to(torch.int32)
        self.gather = torch.where(at > 0, at + 3, at)
        self.flat = torch.zeros(4 + 4 * ((self.n + 3) // 4), dtype=torch.float32, device=at.device)
        self.in_place = False

    def adopt(self):
        """Moves the parameters INTO the flat vector (every .data becomes a view of it): no copy per pack, and an
        optimiser that works on the flat vector updates the modules."""
        o = 4
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + k].view_as(p)
                o += k
        self.in_place = True

    def pack(self) -> torch.Tensor:
        if not self.in_place:
            torch.cat([p.detach().reshape(-1) for p in self.params], out=self.flat[4:4 + self.n])
        return self.flat[self.gather]


def _dist_on() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def normalise_advantages(adv: torch.Tensor) -> torch.Tensor:
    """(A - mean) / (std + 1e-8) with population std over the whole buffer
    (rollout_buffer.py:207-209), the moments summed over every rank."""
    stats = torch.stack([adv.sum(dtype=torch.float64), (adv.double() ** 2).sum(),
                         torch.tensor(float(adv.numel()), dtype=torch.float64, device=adv.device)])
    if _dist_on():
        torch.distributed.all_reduce(stats)
    mean = stats[0] / stats[2]
    var = torch.clamp(stats[1] / stats[2] - mean * mean, min=0.0)
    return ((adv - mean.float()) / (torch.sqrt(var).float() + 1e-8))


class FlatGradSync:
    """One all-reduce for the gradients of several modules: grads are packed into a single
    flat float32 buffer, summed across ranks, divided by the world size and unpacked."""

    def __init__(self, modules):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def sync(self):
        if not _dist_on() or torch.distributed.get_world_size() == 1:
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[o:o + n].zero_()
            else:
                self.flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        torch.distributed.all_reduce(self.flat)
        self.flat.div_(torch.distributed.get_world_size())
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[o:o + n].view_as(p))
            o += n


class PPOAgent:
    """PPOAgent of the reference (agents/ppo_agent.py:66-484) acting on E environments at once.

    get_action(obs [E,M,5]) -> (actions [E,M,3], log_probs [E,M], values [E]) as device tensors;
    a numpy (M,5) observation is treated as a batch of one and returns numpy / float exactly
    like the reference.  update(last_obs, last_done) runs the reference's clipped-surrogate
    loop with its quirks kept: critic on per-agent observations during the update
    (ppo_agent.py:301), two optimisers, grad-norm clip 0.5 each, KL early stop at 0.03.
    """

    def __init__(self, obs_dim=5, hidden_dim=128, M=4, v_max=5.0, num_bands=4, log_std_min=-1.0, log_std_max=0.5,
                 gamma=0.99, gae_lambda=0.95, clip_eps=0.2, lr_actor=3e-4, lr_critic=1e-3, c1=0.5, c2=0.01,
                 max_grad_norm=0.5, rollout_length=2048, batch_size=256, n_epochs=10, device="auto", num_envs=1,
                 seed: Optional[int] = None, fused_update: Optional[bool] = None):
        self.obs_dim, self.hidden_dim, self.M, self.v_max, self.num_bands = obs_dim, hidden_dim, M, v_max, num_bands
        self.gamma, self.gae_lambda, self.clip_eps, self.c1, self.c2 = gamma, gae_lambda, clip_eps, c1, c2
        self.c2_initial, self.max_grad_norm = c2, max_grad_norm
        self.lr_actor_initial, self.lr_critic_initial = lr_actor, lr_critic
        self.rollout_length, self.batch_size, self.n_epochs, self.E = rollout_length, batch_size, n_epochs, num_envs
        self.device = _lib.normalise_device(("cuda" if torch.cuda.is_available() else "cpu") if device == "auto"
                                            else device)
        if seed is not None:
            torch.manual_seed(seed)
        self.actor = Actor(obs_dim, hidden_dim, v_max, num_bands, log_std_min, log_std_max).to(self.device)
        self.critic = Critic(obs_dim, hidden_dim).to(self.device)
        if _dist_on():                       # every rank starts from rank 0's weights
            for p in list(self.actor.parameters()) + list(self.critic.parameters()):
                torch.distributed.broadcast(p.data, src=0)
        # On the GPU the learning rates and the entropy weight live in device memory: marl_ppo_apply reads
        # them there, so a schedule changes them without a host round trip per step.  No torch kernel of the
        # minibatch step is ever captured into a HIP graph (a captured autograd step replayed other gradients
        # than the eager one for minibatches of a few thousand rows and more on this software stack; the
        # library step below is six launches and needs none).
        on_gpu = self.device.type == "cuda"
        mk_lr = (lambda v: torch.tensor(float(v), device=self.device)) if on_gpu else float
        # fused: one launch per optimiser step instead of a dozen (same arithmetic, same state dict)
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr=mk_lr(lr_actor), capturable=on_gpu, fused=on_gpu)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr=mk_lr(lr_critic), capturable=on_gpu, fused=on_gpu)
        self._c2_t = torch.tensor(float(c2), device=self.device)
        self._sync = FlatGradSync([self.actor, self.critic])
        self.buffer = None
        if self.device.type == "cuda":
            self.buffer = DeviceRolloutBuffer(rollout_length, num_envs, M, obs_dim, 3, gamma, gae_lambda, self.device)
        self.total_timesteps, self.total_updates = 0, 0
        self.training_stats: Dict[str, List[float]] = {k: [] for k in
                                                       ("actor_loss", "critic_loss", "entropy", "clip_fraction",
                                                        "approx_kl")}
        self.target_kl = 0.03                                  # ppo_agent.py:284
        self._sample_gen = None
        # Acting through libmarl_env.so (marl_policy_act: both networks, the draws and the log-probabilities in
        # one launch) needs the weights as one packed buffer, refreshed whenever they change.
        self.policy_kernel = self.device.type == "cuda" and hidden_dim in (64, 128) and num_bands == 4 and obs_dim == 5
        self._packed, self._packed_stale, self._act_calls = None, True, 0
        self.seed = 0 if seed is None else int(seed)
        # Minibatch step through libmarl_env.so (marl_ppo_rows: forward, losses and backward of every row in one
        # launch; the sums over rows in a second one).  None = wherever the library has kernels for the networks' shape.
        self.fused_update = self.policy_kernel and (True if fused_update is None else bool(fused_update))
        self._fz = None

    # -- acting (ppo_agent.py:193-225) ----------------------------------------
    @torch.no_grad()
    def get_action(self, obs, deterministic: bool = False):
        single = isinstance(obs, np.ndarray)
        o = torch.from_numpy(obs).float().to(self.device) if single else obs
        if o.dim() == 2:
            o = o.unsqueeze(0)
        actions, log_probs, _ = self.actor.sample(o, deterministic, generator=self._sample_gen)
        values = self.critic.forward_pooled(o.mean(dim=1)).squeeze(-1)
        if single:
            return actions[0].cpu().numpy(), log_probs[0].cpu().numpy(), float(values[0].item())
        return actions, log_probs, values

    def packed_weights(self) -> torch.Tensor:
        if self._packed is None:
            self._packed = pack_policy_weights(self.actor, self.critic)
            n = _lib.load().marl_policy_weight_count(self.hidden_dim)
            if n != self._packed.numel():
                raise _lib.MarlError(f"packed policy weights: {self._packed.numel()} floats, the library expects {n}")
        elif self._packed_stale:
            self._packed.copy_(pack_policy_weights(self.actor, self.critic))      # in place: captured launches read this buffer
        self._packed_stale = False
        return self._packed

    @torch.no_grad()
    def act_kernel(self, obs: torch.Tensor, actions: torch.Tensor, log_probs: torch.Tensor,
                   values: Optional[torch.Tensor], tick: Optional[torch.Tensor] = None, deterministic: bool = False,
                   env_offset: int = 0):
        """get_action (ppo_agent.py:193-225) for obs [E,M,5] through marl_policy_act, results written into the
        given float32 device tensors (actions [E,M,3], log_probs [E,M], values [E] or None)."""
        E, M = obs.shape[0], obs.shape[1]
        for t_, shp in ((obs, (E, M, 5)), (actions, (E, M, 3)), (log_probs, (E, M))):
            if tuple(t_.shape) != shp or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != self.device:
                raise ValueError("act_kernel: contiguous float32 device tensors obs [E,M,5], actions [E,M,3], log_probs [E,M]")
        io = _lib.MarlPolicyIO()
        io.weights, io.obs = self.packed_weights().data_ptr(), obs.data_ptr()
        io.tick = None if tick is None else tick.data_ptr()
        io.actions, io.log_probs = actions.data_ptr(), log_probs.data_ptr()
        io.values = None if values is None else values.data_ptr()
        # draw counter: the envs' step clock when given (it advances with every step, also inside a replayed
        # graph), else the number of calls made so far
        io.seed, io.env_offset, io.deterministic = self.seed, int(env_offset), int(deterministic)
        io.call = 0 if tick is not None else self._act_calls & 0xFFFFFFFF
        self._act_calls += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().marl_policy_act(E, M, self.hidden_dim, float(self.v_max), float(self.actor.log_std_min),
                                                   float(self.actor.log_std_max), C.byref(io),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "marl_policy_act")

    def store_transition(self, obs, actions, log_probs, reward, done, value):
        """ppo_agent.py:227-248 (copying form)."""
        self.buffer.add(obs, actions, log_probs, reward, done, value)
        self.total_timesteps += self.E

    @torch.no_grad()
    def act_into_buffer(self, obs: Optional[torch.Tensor] = None, tick: Optional[torch.Tensor] = None,
                        env_offset: int = 0):
        """Acts on buffer slot t's observation and writes actions / log-probs / values in place;
        returns (actions, reward_out, done_out, next_obs_out) for BatchedJammerEnv.step.  `tick` (the env's
        step clock, views["tick"]) keys the draws of the kernel path on top of the host-side call counter."""
        obs_in, act, logp, val, rew, done, obs_out = self.buffer.slot()
        if obs is not None and obs.data_ptr() != obs_in.data_ptr():
            obs_in.copy_(obs)
        if self.policy_kernel:
            self.act_kernel(obs_in, act, logp, val, tick=tick, env_offset=env_offset)
        else:
            a, lp, v = self.get_action(obs_in)
            act.copy_(a); logp.copy_(lp); val.copy_(v)
        return act, rew, done, obs_out

    def commit_step(self):
        self.buffer.advance()
        self.total_timesteps += self.E

    # -- learning (ppo_agent.py:250-373) ----------------------------------------
    def update(self, last_obs, last_done, permutations=None) -> Dict[str, float]:
        """ppo_agent.py:250-373.  `permutations` (one index array per epoch) replaces the shuffles of
        rollout_buffer.py:251-252, for tests that replay a recorded update of the reference."""
        with torch.no_grad():
            lo = torch.from_numpy(last_obs).float().to(self.device) if isinstance(last_obs, np.ndarray) else last_obs
            if lo.dim() == 2:
                lo = lo.unsqueeze(0)
            last_value = self.critic.forward_pooled(lo.mean(dim=1)).squeeze(-1)
            ld = torch.as_tensor(last_done, device=self.device).reshape(-1).expand(self.E) \
                if not isinstance(last_done, torch.Tensor) else last_done
        self.buffer.compute_returns_and_advantages(last_value, ld)
        obs, actions, logp, adv, ret, _ = self.buffer.get_samples()
        stats = self.optimise(obs, actions, logp, adv, ret, permutations=permutations)
        self.total_updates += 1
        self.buffer.reset()
        for k, v in stats.items():
            self.training_stats[k].append(v)
        return stats

    def _forward_backward(self, b_obs, b_act, b_old, b_adv, b_ret) -> torch.Tensor:
        """Losses and gradients of one minibatch (ppo_agent.py:296-323, 328-329); returns [actor_loss,
        critic_loss, entropy, clip_fraction, approx_kl] as a float64 device tensor."""
        log_probs, entropy = self.actor.evaluate_actions(b_obs, b_act)
        values = self.critic.forward_pooled(b_obs).squeeze(-1)
        ratio = torch.exp(log_probs - b_old)
        surr1 = ratio * b_adv
        surr2 = torch.clamp(ratio, 1 - self.clip_eps, 1 + self.clip_eps) * b_adv
        actor_loss = -torch.min(surr1, surr2).mean()
        critic_loss = self.c1 * ((values - b_ret) ** 2).mean()
        entropy_loss = -self._c2_t * entropy.mean()
        self.actor_optimizer.zero_grad(set_to_none=False)
        self.critic_optimizer.zero_grad(set_to_none=False)
        (actor_loss + entropy_loss).backward()
        critic_loss.backward()
        with torch.no_grad():
            kl = ((ratio - 1) - torch.log(ratio)).mean()
            return torch.stack([actor_loss, critic_loss, entropy.mean(),
                                (torch.abs(ratio - 1) > self.clip_eps).float().mean(), kl]).double()

    def _fused_scratch(self, B: int):
        H = self.hidden_dim
        if self._fz is None:
            packer = TrainPacker(self.actor, self.critic)
            n = _lib.load().marl_ppo_weight_count(H)
            if n != packer.gather.numel():
                raise _lib.MarlError(f"packed training weights: {packer.gather.numel()} floats, the library expects {n}")
            # stats: row 0 for minibatches of batch_size rows, row 1 for the short one at the end of an epoch
            self._fz = {"B": 0, "packer": packer, "stats": torch.zeros(2, 5, dtype=torch.float64, device=self.device)}
            self._fz["gindex"], self._fz["gflat"] = self._gradient_layout()
            packer.adopt()
            nflat = packer.flat.numel() - 4                     # whole groups of four, as marl_ppo_apply reads them
            self._fz.update(packed=torch.empty(n, dtype=torch.float32, device=self.device),
                            m=torch.zeros(nflat, dtype=torch.float32, device=self.device),
                            v=torch.zeros(nflat, dtype=torch.float32, device=self.device),
                            steps=torch.zeros(len(packer.params), dtype=torch.float32, device=self.device))
            self._flatten_optimizer_state()
        z = self._fz
        if z["B"] < B:
            f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.device)
            lib = _lib.load()
            R = 32 * ((B + 31) // 32)                    # the row arrays hold whole tiles of 32 rows
            z.update(B=B, xe=f(R, 8), y1e=f(R, H), y2e=f(R, H), dz1=f(R, H), dz2=f(R, H), dh=f(R, 8),
                     cy1e=f(R, H), cy2e=f(R, H), cdz1=f(R, H), cdz2=f(R, H), dv=f(R),
                     actor_sums=f(lib.marl_ppo_sum_rows(), 4 * H))
            if "partial" not in z:
                z["partial"] = f(lib.marl_ppo_partial_rows(), lib.marl_ppo_partial_floats(H))
        return z

    def _flatten_optimizer_state(self):
        """Adam's moments and step counters of both optimisers as views of three flat vectors (in the order of the flat
        parameter vector), whatever they hold so far copied in: marl_ppo_apply then steps both optimisers in one
        launch while optimizer.state_dict() stays what the reference's checkpoint holds (ppo_agent.py:386-420)."""
        z = self._fz
        params = z["packer"].params
        n_actor_tensors = len(list(self.actor.parameters()))
        o = 0
        with torch.no_grad():
            for i, p in enumerate(params):
                opt = self.actor_optimizer if i < n_actor_tensors else self.critic_optimizer
                st, k = opt.state[p], p.numel()
                m, v, step = z["m"][o:o + k].view_as(p), z["v"][o:o + k].view_as(p), z["steps"][i]
                if "exp_avg" in st:
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    step.copy_(torch.as_tensor(st["step"], dtype=torch.float32))
                else:
                    m.zero_()
                    v.zero_()
                    step.zero_()
                st["step"], st["exp_avg"], st["exp_avg_sq"] = step, m, v
                o += k
        a = _lib.MarlPpoApply()
        a.params, a.grads = z["packer"].flat[4:].data_ptr(), z["gflat"].data_ptr()
        a.exp_avg, a.exp_avg_sq, a.steps = z["m"].data_ptr(), z["v"].data_ptr(), z["steps"].data_ptr()
        ga, gc = self.actor_optimizer.param_groups[0], self.critic_optimizer.param_groups[0]
        for g in (ga, gc):
            if g["betas"] != ga["betas"] or g["eps"] != ga["eps"] or g["weight_decay"] != 0 or g["amsgrad"] or g["maximize"]:
                raise _lib.MarlError("marl_ppo_apply: both optimisers must be plain Adam with the same betas and eps")
        a.lr_actor, a.lr_critic = ga["lr"].data_ptr(), gc["lr"].data_ptr()
        a.gather, a.packed, a.n_packed = z["packer"].gather32.data_ptr(), z["packed"].data_ptr(), z["packed"].numel()
        a.n_actor = sum(p.numel() for p in self.actor.parameters())
        a.n_critic = sum(p.numel() for p in self.critic.parameters())
        a.tensors_actor, a.tensors_critic = n_actor_tensors, len(params) - n_actor_tensors
        a.beta1, a.beta2, a.eps = float(ga["betas"][0]), float(ga["betas"][1]), float(ga["eps"])
        a.max_norm, a.grad_scale = float(self.max_grad_norm), 1.0
        z["apply"] = a

    def _fused_apply(self, grad_scale: float = 1.0):
        """Clip and both Adam steps (ppo_agent.py:324-325, 330-331) on the flat vectors, then the weights laid out
        for the next row pass: marl_ppo_apply."""
        a = self._fz["apply"]
        a.grad_scale = float(grad_scale)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().marl_ppo_apply(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "marl_ppo_apply")

    def _gradient_layout(self):
        """Where every parameter gradient sits in what marl_ppo_products / marl_ppo_rows leave behind (include/marl_env.h):
        an int32 index per element of the flat gradient vector -- >= 0: offset in a workgroup slab, < 0: -(offset in a
        wave's actor_sums row) - 1 -- and that vector, with every .grad pointed into it."""
        H = self.hidden_dim
        sizes = (H * H, H * 32, 32 * H, H * 32, 32 * 32)      # dz2^T y1, dz1^T xe, dh^T y2, dz2^T xe, dh^T xe
        half = sum(sizes)

        def cut(base):
            out, o = [], base
            for n_, shape in zip(sizes, ((H, H), (H, 32), (32, H), (H, 32), (32, 32))):
                out.append(torch.arange(o, o + n_, dtype=torch.int64).view(shape))
                o += n_
            return out

        g2, g1, gh, gb, ghb = cut(0)
        c2, c1, c3, cb, chb = cut(half)
        ln = -(torch.arange(4 * H, dtype=torch.int64) + 1)   # LayerNorm 1 bias, weight, LayerNorm 2 bias, weight (H each)
        t, n = self.actor.trunk, self.critic.network
        where = {t[0].weight: g1[:, :5], t[0].bias: g1[:, 5], t[1].weight: ln[H:2 * H], t[1].bias: ln[:H],
                 t[3].weight: g2, t[3].bias: gb[:, 5], t[4].weight: ln[3 * H:4 * H], t[4].bias: ln[2 * H:3 * H],
                 self.actor.mu_head.weight: gh[0:2], self.actor.mu_head.bias: ghb[0:2, 5],
                 self.actor.log_std_head.weight: gh[2:4], self.actor.log_std_head.bias: ghb[2:4, 5],
                 self.actor.band_head.weight: gh[4:8], self.actor.band_head.bias: ghb[4:8, 5],
                 n[0].weight: c1[:, :5], n[0].bias: c1[:, 5], n[2].weight: c2, n[2].bias: cb[:, 5],
                 n[4].weight: c3[0:1], n[4].bias: chb[0:1, 5]}
        params = list(self.actor.parameters()) + list(self.critic.parameters())
        index = torch.cat([where[p].reshape(-1) for p in params]).to(torch.int32).to(self.device)
        # padded to whole groups of four floats (marl_ppo_apply reads it that way); the padding stays zero
        flat = torch.zeros(4 * ((index.numel() + 3) // 4), dtype=torch.float32, device=self.device)[:index.numel()]
        o = 0
        for p in params:
            p.grad = flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        return index, flat

    def _fused_forward_backward(self, obs, actions, old_log_probs, advantages, returns, idx, stats_row=None):
        """_forward_backward for the rows `idx` of the flat rollout tensors, without gathering them: one marl_ppo_rows
        launch (ppo_agent.py:296-323 per row, on the matrix cores), then every parameter gradient as the product of
        two of the matrices it wrote.  Leaves the gradients in .grad.  A call of its own (stats_row None) lays out
        the current weights first and returns the five statistics; inside optimise() the weights are those
        marl_ppo_apply left, and the sums over the rows are ADDED to row stats_row of the statistics (nothing returned)."""
        B, H = int(idx.numel()), self.hidden_dim
        z = self._fused_scratch(B)
        if stats_row is None:
            z["stats"][0].zero_()
            z["packed"].copy_(z["packer"].pack())
        io = _lib.MarlPpoIO()
        io.weights = z["packed"].data_ptr()
        io.obs, io.actions, io.old_log_probs = obs.data_ptr(), actions.data_ptr(), old_log_probs.data_ptr()
        io.advantages, io.returns, io.index = advantages.data_ptr(), returns.data_ptr(), idx.data_ptr()
        for k in ("xe", "y1e", "y2e", "dz1", "dz2", "actor_sums", "dh", "cy1e", "cy2e", "cdz1", "cdz2", "dv"):
            setattr(io, k, z[k].data_ptr())
        io.stats = z["stats"][stats_row or 0].data_ptr()
        io.clip_eps, io.c1, io.c2, io.v_max = float(self.clip_eps), float(self.c1), float(self.c2), float(self.v_max)
        io.log_std_min, io.log_std_max = float(self.actor.log_std_min), float(self.actor.log_std_max)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().marl_ppo_rows(B, H, C.byref(io), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "marl_ppo_rows")
        lib = _lib.load()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(lib.marl_ppo_products(B, H, C.byref(io), z["partial"].data_ptr(), st), "marl_ppo_products")
            # per-workgroup slabs and per-wave rows -> the flat gradient vector every .grad is a view of
            _lib.check(lib.marl_ppo_reduce(z["gindex"].numel(), z["gindex"].data_ptr(), H, z["partial"].data_ptr(),
                                           z["actor_sums"].data_ptr(), z["gflat"].data_ptr(), st), "marl_ppo_reduce")
        self._keep = idx                                  # the launches read it after this call returns
        return z["stats"][0] / B if stats_row is None else None

    def _apply_gradients(self):
        """Gradient-norm clip and both Adam steps (ppo_agent.py:324-325, 330-331)."""
        nn.utils.clip_grad_norm_(self.actor.parameters(), self.max_grad_norm)
        nn.utils.clip_grad_norm_(self.critic.parameters(), self.max_grad_norm)
        self.actor_optimizer.step()
        self.critic_optimizer.step()

    def _minibatch_step(self, b_obs, b_act, b_old, b_adv, b_ret) -> torch.Tensor:
        """One optimiser step of ppo_agent.py:296-345: gradients, their average over the ranks (one flat
        all-reduce), clip, Adam."""
        out = self._forward_backward(b_obs, b_act, b_old, b_adv, b_ret)
        self._sync.sync()
        self._apply_gradients()
        return out

    def optimise(self, obs, actions, old_log_probs, advantages, returns, generator=None,
                 permutations=None) -> Dict[str, float]:
        """The epoch / minibatch loop of ppo_agent.py:286-354 on flattened device tensors.  Works on
        any device; with torch.distributed initialised, gradients are averaged in one flat
        all-reduce per step and the epoch KL is averaged so that every rank stops together."""
        n = obs.shape[0]
        self._packed_stale = True
        if self.fused_update:
            obs, actions, old_log_probs, advantages, returns = (
                x.to(torch.float32).contiguous() for x in (obs, actions, old_log_probs, advantages, returns))
        keys = ("actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl")
        acc = torch.zeros(len(keys), dtype=torch.float64, device=obs.device)
        steps = 0
        if self.fused_update:
            # the library adds every minibatch's sums over its rows to one of two rows of statistics (full minibatches,
            # the short last one): sum(rows) / B per minibatch, summed over the steps so far, is their weighted sum
            z = self._fused_scratch(min(n, self.batch_size))
            z["stats"].zero_()
            z["packed"].copy_(z["packer"].pack())
            tail = n % self.batch_size
            weight = torch.tensor([1.0 / min(n, self.batch_size), 1.0 / tail if tail and n > self.batch_size else 0.0],
                                  dtype=torch.float64, device=obs.device)
            world = torch.distributed.get_world_size() if _dist_on() else 1
        for epoch in range(self.n_epochs):
            if permutations is not None:
                perm = torch.as_tensor(np.asarray(permutations[epoch]), dtype=torch.long, device=obs.device)
            else:
                perm = torch.randperm(n, device=obs.device, generator=generator)
            epoch_kl = torch.zeros((), dtype=torch.float64, device=obs.device)
            nb = 0
            for s in range(0, n, self.batch_size):
                idx = perm[s:s + self.batch_size]
                if self.fused_update:
                    self._fused_forward_backward(obs, actions, old_log_probs, advantages, returns, idx,
                                                 stats_row=0 if idx.numel() == min(n, self.batch_size) else 1)
                    if world > 1:                                  # the gradients already are one flat vector
                        torch.distributed.all_reduce(z["gflat"])
                    self._fused_apply(1.0 / world)
                    nb += 1
                    steps += 1
                    continue
                batch = (obs[idx], actions[idx], old_log_probs[idx], advantages[idx], returns[idx])
                out = self._minibatch_step(*batch)
                acc += out
                epoch_kl += out[4]
                nb += 1
                steps += 1
            if self.fused_update:
                total = (z["stats"] * weight[:, None]).sum(0)
                epoch_kl = total[4] - acc[4]
                acc = total
            avg_kl = epoch_kl / max(nb, 1)
            if _dist_on():
                torch.distributed.all_reduce(avg_kl)
                avg_kl = avg_kl / torch.distributed.get_world_size()
            if float(avg_kl) > self.target_kl:
                break
        mean = (acc / max(steps, 1))
        if _dist_on():
            torch.distributed.all_reduce(mean)
            mean = mean / torch.distributed.get_world_size()
        return {k: float(v) for k, v in zip(keys, mean.cpu())}

    # -- schedule / mode / checkpoint (ppo_agent.py:375-484) ----------------------
    def set_learning_rate(self, lr_actor, lr_critic):
        for opt, lr in ((self.actor_optimizer, lr_actor), (self.critic_optimizer, lr_critic)):
            for g in opt.param_groups:
                if isinstance(g["lr"], torch.Tensor):
                    g["lr"].fill_(float(lr))          # in place: a captured step reads this tensor
                else:
                    g["lr"] = lr

    def set_entropy_coef(self, c2):
        self.c2 = c2
        self._c2_t.fill_(float(c2))

    def set_training_mode(self, training=True):
        self.actor.train(training)
        self.critic.train(training)

    def _config(self):
        return {k: getattr(self, k) for k in ("obs_dim", "hidden_dim", "M", "v_max", "num_bands", "gamma",
                                              "gae_lambda", "clip_eps", "c1", "c2", "max_grad_norm",
                                              "rollout_length", "batch_size", "n_epochs")}

    def save(self, path):
        """Same file name and keys as the reference checkpoint (ppo_agent.py:386-420); the statistics are
        stored as plain floats so that the file loads with weights_only=True."""
        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        stats = {k: [float(x) for x in v] for k, v in self.training_stats.items()}
        torch.save({"actor_state_dict": self.actor.state_dict(), "critic_state_dict": self.critic.state_dict(),
                    "actor_optimizer_state_dict": self._portable_optimizer_state(self.actor_optimizer),
                    "critic_optimizer_state_dict": self._portable_optimizer_state(self.critic_optimizer),
                    "total_timesteps": self.total_timesteps, "total_updates": self.total_updates,
                    "training_stats": stats, "config": self._config()}, path / "ppo_agent.pt")

    @staticmethod
    def _portable_optimizer_state(opt):
        """state_dict() with the param groups as the reference's plain torch.optim.Adam writes them (float learning
        rate, capturable=False, fused=None): PPOAgent.load of the reference (ppo_agent.py:430-435) hands them to
        load_state_dict of such an optimiser, on CPU parameters too, where capturable / fused groups do not step."""
        sd = opt.state_dict()
        groups = []
        for g in sd["param_groups"]:
            g = dict(g)
            g["lr"] = float(g["lr"])
            g["capturable"], g["fused"] = False, None
            groups.append(g)
        state = {k: {n: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                 for k, st in sd["state"].items()}
        return {"state": state, "param_groups": groups}

    @staticmethod
    def _read_checkpoint(path, map_location):
        """torch.load with the weights-only unpickler.  A checkpoint written by the reference holds
        numpy float64 scalars in `training_stats` (np.mean of the losses, ppo_agent.py:361-371): numpy's
        scalar constructor and dtype classes are allow-listed, nothing else."""
        allowed = [np.dtype, type(np.dtype(np.float64)), type(np.dtype(np.float32)), type(np.dtype(np.int64))]
        try:
            allowed.append(np._core.multiarray.scalar)
        except AttributeError:                       # numpy < 2
            allowed.append(np.core.multiarray.scalar)
        with torch.serialization.safe_globals(allowed):
            return torch.load(Path(path) / "ppo_agent.pt", map_location=map_location, weights_only=True)

    def _adopt_optimizer_state(self, opt, state_dict):
        """load_state_dict, then this agent's own group settings again: a checkpoint from the reference
        or from a CPU run has a float learning rate, capturable=False and step counters on the host,
        none of which a captured Adam step on the GPU accepts (and the reverse for a GPU checkpoint
        loaded on the CPU)."""
        opt.load_state_dict(state_dict)
        on_gpu = self.device.type == "cuda"
        for g in opt.param_groups:
            lr = float(g["lr"])
            g["lr"] = torch.tensor(lr, device=self.device) if on_gpu else lr
            g["capturable"] = on_gpu
            g["fused"] = True if on_gpu else None
        for st in opt.state.values():
            if "step" in st and isinstance(st["step"], torch.Tensor):
                st["step"] = st["step"].to(self.device if on_gpu else "cpu", torch.float32)

    def load(self, path):
        ck = self._read_checkpoint(path, self.device)
        self.actor.load_state_dict(ck["actor_state_dict"])
        self.critic.load_state_dict(ck["critic_state_dict"])
        self._adopt_optimizer_state(self.actor_optimizer, ck["actor_optimizer_state_dict"])
        self._adopt_optimizer_state(self.critic_optimizer, ck["critic_optimizer_state_dict"])
        self.total_timesteps, self.total_updates = int(ck["total_timesteps"]), int(ck["total_updates"])
        self.training_stats = {k: [float(x) for x in v] for k, v in ck["training_stats"].items()}
        if "c2" in ck.get("config", {}):               # the decayed entropy weight is part of a resumed run's state
            self.set_entropy_coef(float(ck["config"]["c2"]))
        if self._fz is not None:                       # the loaded moments into the flat vectors marl_ppo_apply steps
            self._flatten_optimizer_state()
        self._packed_stale = True

    def export_actor(self, path):
        """`actor_state_dict.pt` for the reference's deployment controller (deploy.py:139-188): the state
        dict its MinimalActor (deploy.py:33-92) loads strictly, i.e. trunk, velocity-mean head and band
        head; the log-std head is not part of the deployed network."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        sd = {k: v.detach().cpu().clone() for k, v in self.actor.state_dict().items()
              if not k.startswith("log_std_head.")}
        torch.save(sd, path)
        return sd

    @classmethod
    def from_checkpoint(cls, path, device="auto", num_envs=1):
        ck = cls._read_checkpoint(path, "cpu")
        agent = cls(device=device, num_envs=num_envs, **ck["config"])
        agent.load(path)
        return agent
