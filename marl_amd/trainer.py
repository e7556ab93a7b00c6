"""Batched counterpart of the reference training loop (training/trainer.py:185-294, 460-599):
collect T steps from E lock-stepped environments straight into the device rollout buffer,
run the PPO update, decay learning rate and entropy coefficient as the reference does.
One process per GPU; with torch.distributed initialised each rank owns E environments
(env_offset = rank * E), gradients and episode metrics are all-reduced, the step itself
never communicates."""
from __future__ import annotations

import time
from typing import Any, Dict, Optional

import torch

from . import _lib
from .agents import PPOAgent, _dist_on
from .env import BatchedJammerEnv, EnvConfig


class BatchedTrainer:
    def __init__(self, env_cfg: EnvConfig, num_envs: int, rollout_length: int = 64, total_timesteps: int = 10 ** 7,
                 seed: int = 0, device="cuda", agent_kwargs: Optional[Dict[str, Any]] = None,
                 minibatches_per_epoch: int = 32, use_graph: bool = True):
        rank = torch.distributed.get_rank() if _dist_on() else 0
        self.world = torch.distributed.get_world_size() if _dist_on() else 1
        self.device = _lib.normalise_device(device)
        self.E, self.T = num_envs, rollout_length
        self.total_budget = total_timesteps
        self.env = BatchedJammerEnv(num_envs, env_cfg, seed=seed, device=self.device, auto_reset=True,
                                    env_offset=rank * num_envs)
        # The reference's defaults (2048-step rollout of one env, M = 4, minibatches of 256:
        # training/config.py:86-96) give 32 optimiser steps per epoch; a batched rollout holds E*T*M
        # samples, so the minibatch grows with it and the number of steps per epoch stays the same.
        samples = rollout_length * num_envs * env_cfg.M
        kw = dict(M=env_cfg.M, v_max=env_cfg.v_max, rollout_length=rollout_length, num_envs=num_envs,
                  device=self.device, seed=seed, batch_size=max(256, samples // max(1, minibatches_per_epoch)))
        kw.update(agent_kwargs or {})
        self.agent = PPOAgent(**kw)
        self.total_timesteps = 0          # env-steps over all ranks
        self.total_rollouts = 0
        self.total_episodes = 0
        self._ep_reward = torch.zeros(num_envs, dtype=torch.float64, device=self.device)
        self._obs = None
        self._done = torch.zeros(num_envs, dtype=torch.bool, device=self.device)
        # The T steps of a rollout (policy inference, step kernels, masked reset, bookkeeping: about
        # 40 small launches per step) are captured once into a HIP graph and replayed: the loop is
        # launch-bound from Python, and everything in it already works on fixed device buffers and
        # draws its randomness from device-side counters.
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._graph = None
        self._eager_rollouts = 0
        self._stats = torch.zeros(4, dtype=torch.float64, device=self.device)   # episodes, sum reward, sum reduction %, sum len

    def _rollout_steps(self):
        """trainer.py:185-294 for E envs: T steps, transitions written in place by the step kernels;
        per-episode reward and lambda2 reduction are accumulated on the device.  Touches only
        persistent tensors, so that the launch sequence can be captured and replayed."""
        env, agent = self.env, self.agent
        self._stats.zero_()
        obs = self._obs
        for _ in range(self.T):
            actions, rew_out, done_out, obs_out = agent.act_into_buffer(obs)
            obs, reward, _, truncated, info = env.step(actions, out_obs=obs_out, out_reward=rew_out,
                                                       out_truncated=done_out)
            agent.commit_step()
            self._ep_reward += reward.double()
            lam0 = info["episode_lambda2_initial"]
            red = torch.where(lam0 > 1e-10, 100.0 * (1.0 - info["lambda2_current"] / lam0), torch.zeros_like(lam0))
            d = truncated.double()
            self._stats += torch.stack([d.sum(), (self._ep_reward * d).sum(), (red * d).sum(),
                                        (info["episode_step"].double() * d).sum()])
            self._ep_reward *= (1.0 - d)
        self._obs.copy_(obs)
        self._done.copy_(truncated)

    def collect_rollout(self) -> Dict[str, float]:
        env, agent = self.env, self.agent
        if self._obs is None:
            obs0, _ = env.reset()
            self._obs = obs0.clone()
        if not self.use_graph:
            self._rollout_steps()
        elif self._graph is None and self._eager_rollouts < 1:
            self._rollout_steps()                          # first rollout eagerly: warms up every lazy initialisation
            self._eager_rollouts += 1
        else:
            if self._graph is None:
                pos, steps = agent.buffer.pos, agent.total_timesteps
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._rollout_steps()
                self._graph = g
                # capture runs nothing: put the host-side counters back and replay below
                agent.buffer.pos, agent.buffer.full, agent.total_timesteps = pos, False, steps
            self._graph.replay()
            agent.buffer.pos, agent.buffer.full = self.T, True
            agent.total_timesteps += self.T * self.E
        stats = self._stats.clone()
        if _dist_on():
            torch.distributed.all_reduce(stats)
        self.total_timesteps += self.T * self.E * self.world
        n = max(float(stats[0]), 1.0)
        self.total_episodes += int(stats[0])
        return {"steps": self.T * self.E * self.world, "episodes_completed": int(stats[0]),
                "mean_reward": float(stats[1]) / n, "mean_lambda2_reduction": float(stats[2]) / n,
                "mean_length": float(stats[3]) / n}

    def update(self) -> Dict[str, float]:
        return self.agent.update(self._obs, self._done)

    def train(self, max_rollouts: Optional[int] = None, log=print) -> Dict[str, Any]:
        t0 = time.time()
        while self.total_timesteps < self.total_budget:
            roll = self.collect_rollout()
            upd = self.update()
            self.total_rollouts += 1
            progress = min(self.total_timesteps / self.total_budget, 1.0)           # trainer.py:505-519
            decay = 0.1 + 0.9 * (1.0 - progress)
            self.agent.set_learning_rate(self.agent.lr_actor_initial * decay, self.agent.lr_critic_initial * decay)
            self.agent.set_entropy_coef(self.agent.c2_initial * (0.2 + 0.8 * (1.0 - progress)))
            if log is not None:
                log(f"[rollout {self.total_rollouts}] steps={self.total_timesteps:,} reward={roll['mean_reward']:.3f} "
                    f"l2_red={roll['mean_lambda2_reduction']:.1f}% kl={upd['approx_kl']:.4f} "
                    f"fps={self.total_timesteps / (time.time() - t0):,.0f}")
            if max_rollouts is not None and self.total_rollouts >= max_rollouts:
                break
        elapsed = time.time() - t0
        return {"total_timesteps": self.total_timesteps, "total_episodes": self.total_episodes,
                "total_rollouts": self.total_rollouts, "time_elapsed": elapsed,
                "fps": self.total_timesteps / max(elapsed, 1e-9)}
