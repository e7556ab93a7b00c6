"""Batched counterpart of the reference training loop (training/trainer.py:185-294, 326-458, 460-599):
collect T steps from E lock-stepped environments straight into the device rollout buffer,
run the PPO update, decay learning rate and entropy coefficient, log, checkpoint, evaluate and test
for convergence as the reference does.  One process per GPU; with torch.distributed initialised each
rank owns E environments (env_offset = rank * E), gradients and episode metrics are all-reduced, the
step itself never communicates.

A collection step is two library calls: marl_policy_act (both networks, the draws, the log-probabilities,
written into the buffer slot) and marl_env_step (+ the masked reset), which also keeps the per-episode
return and the episode sums of the reference's rollout statistics on the device.  Nothing in the loop
depends on host-visible results, so the T steps can be captured once as a HIP graph and replayed."""
from __future__ import annotations

import csv
import json
import time
from collections import deque
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _lib
from .agents import PPOAgent, _dist_on
from .env import BatchedJammerEnv, EnvConfig

# column names of training_log.csv / history.json (training/metrics.py:121-135)
HISTORY_KEYS = ("timestep", "episode", "reward", "lambda2_reduction", "episode_length", "band_match_rate",
                "actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl", "time_elapsed",
                "avg_jamming_power_dbm")


class RolloutLog:
    """What MetricsLogger (training/metrics.py:72-315) writes: a row per rollout in training_log.csv, the
    same rows column-wise in history.json, and final_stats.json."""

    def __init__(self, output_dir):
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)
        self.history = {k: [] for k in HISTORY_KEYS}
        self.start_time = time.time()
        self._csv_file = None
        self._csv = None
        self._append = False

    def resume(self):
        """A run that continues from a checkpoint keeps the rows it has: history.json is read back if present and
        training_log.csv is appended to instead of started over."""
        hp = self.output_dir / "history.json"
        if hp.exists():
            with open(hp) as f:
                old = json.load(f)
            self.history = {k: list(old.get(k, [])) for k in HISTORY_KEYS}
            if self.history["time_elapsed"]:
                self.start_time = time.time() - float(self.history["time_elapsed"][-1])
        self._append = (self.output_dir / "training_log.csv").exists()

    def _open(self):
        path = self.output_dir / "training_log.csv"
        self._csv_file = open(path, "a" if self._append else "w", newline="")
        self._csv = csv.DictWriter(self._csv_file, fieldnames=list(HISTORY_KEYS))
        if not self._append:
            self._csv.writeheader()

    def log_rollout(self, **rec):
        if self._csv_file is None:
            self._open()
        rec = {k: rec.get(k, 0.0) for k in HISTORY_KEYS}
        rec["time_elapsed"] = time.time() - self.start_time
        for k, v in rec.items():
            self.history[k].append(v)
        self._csv.writerow(rec)
        self._csv_file.flush()

    def save(self, final_stats: Optional[Dict[str, Any]] = None):
        with open(self.output_dir / "history.json", "w") as f:
            json.dump(self.history, f, indent=2)
        if final_stats is not None:
            with open(self.output_dir / "final_stats.json", "w") as f:
                json.dump(final_stats, f, indent=2)

    def close(self):
        if self._csv_file:
            self._csv_file.close()
            self._csv_file = None
            self._append = True


class BatchedTrainer:
    def __init__(self, env_cfg: EnvConfig, num_envs: int, rollout_length: int = 64, total_timesteps: int = 10 ** 7,
                 seed: int = 0, device="cuda", agent_kwargs: Optional[Dict[str, Any]] = None,
                 minibatches_per_epoch: int = 32, use_graph: bool = True, output_dir: Optional[str] = None,
                 log_interval: int = 1, save_interval: int = 10, eval_interval: int = 10, eval_episodes: int = 5,
                 target_reduction: float = 70.0, convergence_window: int = 50, early_stop_patience: int = 100,
                 disable_early_convergence: bool = False, log_step_info: bool = True):
        rank = torch.distributed.get_rank() if _dist_on() else 0
        self.rank = rank
        self.world = torch.distributed.get_world_size() if _dist_on() else 1
        self.device = _lib.normalise_device(device)
        self.E, self.T = num_envs, rollout_length
        self.env_cfg, self.seed = env_cfg, seed
        self.total_budget = total_timesteps
        self.env = BatchedJammerEnv(num_envs, env_cfg, seed=seed, device=self.device, auto_reset=True,
                                    env_offset=rank * num_envs)
        # What the reference's rollout loop reads from `info` at every step (training/trainer.py:239-245): the mean
        # jamming power at the link midpoints and the band matches.  One slot per step of the rollout, written by the
        # step kernel, reduced once per rollout.  (The reference's env never puts `band_match_rate` into `info`
        # (jammer_env.py:443-455), so its column is always 0.0; here it is the mean of the reward's band-match
        # term, the fraction of jammers on the enemy's band.)
        self.log_step_info = bool(log_step_info)
        self._power = torch.zeros(rollout_length, num_envs, dtype=torch.float64, device=self.device) \
            if self.log_step_info else None
        self._terms = torch.zeros(rollout_length, num_envs, 5, dtype=torch.float32, device=self.device) \
            if self.log_step_info else None
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
        self.best_reduction = 0.0         # training/trainer.py:101-103
        self.no_improvement_count = 0
        self._obs = None
        self._done = torch.zeros(num_envs, dtype=torch.bool, device=self.device)
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._graph = None
        self._eager_rollouts = 0
        # episodes, sum of returns, sum of lambda2 reductions (%), sum of lengths: filled by the step kernels
        self._stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        # cadence and convergence test (training/config.py:116-135)
        self.log_interval, self.save_interval, self.eval_interval = log_interval, save_interval, eval_interval
        self.eval_episodes, self.target_reduction = eval_episodes, target_reduction
        self.convergence_window, self.early_stop_patience = convergence_window, early_stop_patience
        self.disable_early_convergence = disable_early_convergence
        self.output_dir = Path(output_dir) if output_dir is not None else None
        self.logger = RolloutLog(self.output_dir) if (self.output_dir is not None and rank == 0) else None
        # The reference's window holds single episodes (training/trainer.py:440-442); a batched rollout
        # finishes thousands at once, so the window holds one mean per rollout that finished episodes.
        self._reductions = deque(maxlen=max(convergence_window, 1))

    # -- rollout (training/trainer.py:185-294) ----------------------------------
    def _rollout_steps(self):
        """T collection steps; touches only persistent tensors, so the launch sequence can be captured."""
        env, agent = self.env, self.agent
        self._stats.zero_()
        obs = self._obs
        tick = env.views["tick"]
        for t in range(self.T):
            actions, rew_out, done_out, obs_out = agent.act_into_buffer(obs, tick=tick, env_offset=env.env_offset)
            obs, _, _, truncated, _ = env.step(actions, out_obs=obs_out, out_reward=rew_out, out_truncated=done_out,
                                               episode_stats=self._stats, episode_info=False,
                                               out_reward_terms=self._terms[t] if self.log_step_info else None,
                                               out_avg_power=self._power[t] if self.log_step_info else None)
            agent.commit_step()
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
                pos, steps, calls = agent.buffer.pos, agent.total_timesteps, agent._act_calls
                if agent.policy_kernel:
                    agent.packed_weights()                 # packed outside the capture; replays read the same buffer
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._rollout_steps()
                self._graph = g
                # capture runs nothing: put the host-side counters back and replay below
                agent.buffer.pos, agent.buffer.full, agent.total_timesteps = pos, False, steps
                agent._act_calls = calls
            if agent.policy_kernel:
                agent.packed_weights()                     # refreshed in place after an update
            self._graph.replay()
            agent.buffer.pos, agent.buffer.full = self.T, True
            agent.total_timesteps += self.T * self.E
        if self.log_step_info:
            per_step = torch.stack([self._power.sum(), self._terms[..., 1].sum(dtype=torch.float64)])
        else:
            per_step = torch.zeros(2, dtype=torch.float64, device=self.device)
        stats = torch.cat([self._stats, per_step])
        if _dist_on():
            torch.distributed.all_reduce(stats)
        stats = stats.cpu()
        self.total_timesteps += self.T * self.E * self.world
        n = max(float(stats[0]), 1.0)
        samples = float(self.T * self.E * self.world)
        self.total_episodes += int(stats[0])
        return {"steps": self.T * self.E * self.world, "episodes_completed": int(stats[0]),
                "mean_reward": float(stats[1]) / n, "mean_lambda2_reduction": float(stats[2]) / n,
                "mean_length": float(stats[3]) / n,
                # mean over the steps of the rollout (training/trainer.py:293); -80.0 is what the reference reports
                # when it has no value
                "mean_jamming_power_dbm": float(stats[4]) / samples if self.log_step_info else -80.0,
                "band_match_rate": float(stats[5]) / samples}

    def update(self) -> Dict[str, float]:
        return self.agent.update(self._obs, self._done)

    # -- evaluation (training/trainer.py:326-385) -----------------------------------
    @torch.no_grad()
    def evaluate(self, n_episodes: int = 5, deterministic: bool = False, num_envs: int = 256) -> Dict[str, float]:
        """n_episodes episodes on each of `num_envs` fresh environments with the current policy: mean / std of
        the episode return and of the final lambda2 reduction, mean length, fraction of episodes in which the
        graph was fragmented (lambda2 < 1e-10) at some step.  Does not touch the training environments."""
        cfg, agent = self.env_cfg, self.agent
        env = BatchedJammerEnv(num_envs, cfg, seed=self.seed + 7919 + self.total_rollouts, device=self.device,
                               auto_reset=False, env_offset=self.rank * num_envs)
        agent.set_training_mode(False)
        act = torch.empty(num_envs, cfg.M, 3, device=self.device)
        logp = torch.empty(num_envs, cfg.M, device=self.device)
        rets, reds, frag = [], [], []
        band = torch.zeros((), dtype=torch.float64, device=self.device)
        for ep in range(n_episodes):
            obs, info = env.reset()
            lam0 = info["lambda2_initial"].clone()
            ret = torch.zeros(num_envs, dtype=torch.float64, device=self.device)
            fragmented = torch.zeros(num_envs, dtype=torch.bool, device=self.device)
            for t in range(cfg.max_steps):
                if agent.policy_kernel:
                    agent.act_kernel(obs, act, logp, None, tick=env.views["tick"], deterministic=deterministic,
                                     env_offset=env.env_offset)
                    a = act
                else:
                    a, _, _ = agent.get_action(obs, deterministic=deterministic)
                obs, reward, _, truncated, info = env.step(a)
                ret += reward.double()
                fragmented |= info["lambda2_current"] < 1e-10
                band += info["reward_terms"][:, 1].sum(dtype=torch.float64)
            lam = info["lambda2_current"]
            red = torch.where(lam0 > 1e-10, 100.0 * (1.0 - lam / lam0), torch.zeros_like(lam0))
            rets.append(ret); reds.append(red); frag.append(fragmented.double())
        agent.set_training_mode(True)
        rets, reds, frag = (torch.cat(x) for x in (rets, reds, frag))
        moments = torch.stack([torch.tensor(float(rets.numel()), dtype=torch.float64, device=self.device), rets.sum(),
                               (rets ** 2).sum(), reds.sum(), (reds ** 2).sum(), frag.sum(), band])
        if _dist_on():
            torch.distributed.all_reduce(moments)
        m = moments.cpu().numpy()
        n = m[0]
        std = lambda s, q: float(np.sqrt(max(q / n - (s / n) ** 2, 0.0)))
        return {"reward_mean": float(m[1] / n), "reward_std": std(m[1], m[2]),
                "lambda2_reduction_mean": float(m[3] / n), "lambda2_reduction_std": std(m[3], m[4]),
                "episode_length_mean": float(cfg.max_steps), "band_match_rate_mean": float(m[6] / (n * cfg.max_steps)),
                "fragmentation_rate": float(m[5] / n), "episodes": int(n)}

    # -- checkpoints (training/trainer.py:387-420) ----------------------------------
    def save_checkpoint(self, name: str = "checkpoint", directory: Optional[str] = None) -> Path:
        """ppo_agent.pt + trainer_state.json as the reference writes them, the deployment actor
        (actor_state_dict.pt, deploy.py:139-188), and env_state_<rank>.pt: the device state of the E
        environments, the current observation and the generator counters, so that a resumed run continues
        the same trajectory.  Every rank writes its own env state; rank 0 writes the rest."""
        base = Path(directory) if directory is not None else (self.output_dir or Path(".")) / "checkpoints" / name
        base.mkdir(parents=True, exist_ok=True)
        if self.rank == 0:
            self.agent.save(base)
            self.agent.export_actor(base / "actor_state_dict.pt")
            with open(base / "trainer_state.json", "w") as f:
                json.dump({"total_timesteps": self.total_timesteps, "total_episodes": self.total_episodes,
                           "total_rollouts": self.total_rollouts, "best_reduction": self.best_reduction,
                           "no_improvement_count": self.no_improvement_count, "num_envs": self.E,
                           "rollout_length": self.T, "world_size": self.world,
                           "recent_reductions": [float(x) for x in self._reductions]}, f, indent=2)
            if self.logger is not None:
                self.logger.save()                 # history.json beside the CSV, so that a resumed run appends to both
        torch.cuda.synchronize(self.device)
        env = self.env
        bounds = env._count_bounds
        torch.save({"state": env._buf.cpu(), "base_offset": int(env.state.enemy_x) - env._buf.data_ptr(),
                    "obs": None if self._obs is None else self._obs.cpu(), "done": self._done.cpu(),
                    "episode": env._episode, "seed": env.seed, "env_offset": env.env_offset,
                    "count_bounds": [-1, -1] if bounds is None else [int(bounds[0]), int(bounds[1])],
                    "act_calls": self.agent._act_calls, "torch_rng": torch.cuda.get_rng_state(self.device),
                    "E": env.E, "N": env.N, "M": env.M}, base / f"env_state_{self.rank}.pt")
        return base

    def load_checkpoint(self, path: str):
        """Restores what save_checkpoint wrote (the agent in place, counters, environments).  The captured
        rollout graph of this object, if any, stays valid: every buffer is restored in place."""
        base = Path(path)
        self.agent.load(base)
        if self.agent.buffer is not None:
            self.agent.buffer.reset()              # checkpoints are written between rollouts: nothing collected yet
        sp = base / "trainer_state.json"
        if sp.exists():
            with open(sp) as f:
                st = json.load(f)
            self.total_timesteps = st.get("total_timesteps", 0)
            self.total_episodes = st.get("total_episodes", 0)
            self.total_rollouts = st.get("total_rollouts", 0)
            self.best_reduction = st.get("best_reduction", 0.0)
            self.no_improvement_count = st.get("no_improvement_count", 0)
            self._reductions.clear()
            self._reductions.extend(st.get("recent_reductions", []))
        if self.logger is not None:
            self.logger.resume()
        ep = base / f"env_state_{self.rank}.pt"
        if ep.exists():
            ck = torch.load(ep, map_location="cpu", weights_only=True)
            env = self.env
            if (ck["E"], ck["N"], ck["M"]) != (env.E, env.N, env.M):
                raise ValueError("env_state was saved for another batch shape")
            # the arrays sit at the same offsets from the first one whatever the alignment of the allocation
            src_off, dst_off = int(ck["base_offset"]), int(env.state.enemy_x) - env._buf.data_ptr()
            n = min(ck["state"].numel() - src_off, env._buf.numel() - dst_off)
            env._buf[dst_off:dst_off + n].copy_(ck["state"][src_off:src_off + n])
            env._episode, env.seed = int(ck["episode"]), int(ck["seed"])
            cb = ck["count_bounds"]
            env._count_bounds = None if (cb[0] < 0 or not env._track_counts) else [int(cb[0]), int(cb[1])]
            if ck["obs"] is not None:
                if self._obs is None:
                    self._obs = ck["obs"].to(self.device)
                else:
                    self._obs.copy_(ck["obs"])
            self._done.copy_(ck["done"])
            self.agent._act_calls = int(ck["act_calls"])
            torch.cuda.set_rng_state(ck["torch_rng"], self.device)

    # -- convergence test (training/trainer.py:422-458) -------------------------------
    def check_convergence(self) -> bool:
        if self.disable_early_convergence or len(self._reductions) < self.convergence_window:
            return False
        mean_reduction = float(np.mean(self._reductions))
        if mean_reduction > self.best_reduction:
            self.best_reduction = mean_reduction
            self.no_improvement_count = 0
        elif mean_reduction <= self.best_reduction + 1.0:
            self.no_improvement_count += 1
        if mean_reduction >= self.target_reduction:
            return True
        return self.no_improvement_count >= self.early_stop_patience

    # -- main loop (training/trainer.py:460-599) ---------------------------------------
    def train(self, max_rollouts: Optional[int] = None, log=print) -> Dict[str, Any]:
        t0 = time.time()
        while self.total_timesteps < self.total_budget:
            roll = self.collect_rollout()
            self.total_rollouts += 1
            upd = self.update()
            progress = min(self.total_timesteps / self.total_budget, 1.0)           # trainer.py:505-519
            decay = 0.1 + 0.9 * (1.0 - progress)
            self.agent.set_learning_rate(self.agent.lr_actor_initial * decay, self.agent.lr_critic_initial * decay)
            self.agent.set_entropy_coef(self.agent.c2_initial * (0.2 + 0.8 * (1.0 - progress)))
            if roll["episodes_completed"] > 0:
                self._reductions.append(roll["mean_lambda2_reduction"])
            if self.logger is not None:
                self.logger.log_rollout(timestep=self.total_timesteps, episode=self.total_episodes,
                                        reward=roll["mean_reward"], lambda2_reduction=roll["mean_lambda2_reduction"],
                                        episode_length=roll["mean_length"], band_match_rate=roll["band_match_rate"],
                                        avg_jamming_power_dbm=roll["mean_jamming_power_dbm"], **upd)
            if log is not None and self.rank == 0 and self.total_rollouts % max(self.log_interval, 1) == 0:
                log(f"[Rollout {self.total_rollouts}] Steps: {self.total_timesteps:,} | Episodes: {self.total_episodes} | "
                    f"Reward: {roll['mean_reward']:.2f} | L2_red: {roll['mean_lambda2_reduction']:.1f}% | "
                    f"Entropy: {upd['entropy']:.3f} | KL: {upd['approx_kl']:.4f} | "
                    f"FPS: {self.total_timesteps / (time.time() - t0):,.0f}")
            if self.output_dir is not None and self.total_rollouts % max(self.save_interval, 1) == 0:
                self.save_checkpoint(f"rollout_{self.total_rollouts}")
                self.save_checkpoint("latest")
            if self.output_dir is not None and self.total_rollouts % max(self.eval_interval, 1) == 0:
                summary = self.evaluate(n_episodes=self.eval_episodes)
                if log is not None and self.rank == 0:
                    log(f"  [Eval] Reward: {summary['reward_mean']:.2f} | L2_reduction: {summary['lambda2_reduction_mean']:.1f}%")
                if summary["lambda2_reduction_mean"] > self.best_reduction:
                    self.best_reduction = summary["lambda2_reduction_mean"]
                    self.save_checkpoint("best")
            if self.check_convergence():
                break
            if max_rollouts is not None and self.total_rollouts >= max_rollouts:
                break
        elapsed = time.time() - t0
        final = {"total_timesteps": self.total_timesteps, "total_episodes": self.total_episodes,
                 "total_rollouts": self.total_rollouts, "best_lambda2_reduction": self.best_reduction,
                 "time_elapsed": elapsed, "fps": self.total_timesteps / max(elapsed, 1e-9)}
        if self.output_dir is not None:
            self.save_checkpoint("final")
            if self.logger is not None:
                self.logger.save(final)
                self.logger.close()
        return final
