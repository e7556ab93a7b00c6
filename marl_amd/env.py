"""Host side of the batched JammerEnv step.

`BatchedJammerEnv` advances E environments in lockstep on one GPU: all state is
device resident (one caller-owned buffer carved by `marl_state_bind`), `step`
and `reset` each launch one HIP kernel of libmarl_env.so on torch's current
stream and return device tensors without synchronising.

`JammerEnv` is the single-environment facade with the reference's constructor,
`reset()` / `step()` tuples, `info` keys and attributes
(reference: environment/jammer_env.py:94-127, 261-354, 356-457).  It draws the
random numbers on the host from numpy `RandomState` streams in the order the
reference consumes them, so a seed gives the same trajectory as the reference.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import MarlEnvParams, MarlEnvState, MarlResetIO, MarlStepIO

BANDS = [433e6, 915e6, 2.4e9, 5.8e9]          # jammer_env.py:182


def db_to_watts(dbm: float) -> float:
    """physics/fspl.py:50-69"""
    return 10 ** ((dbm - 30) / 10)


@dataclass
class EnvConfig:
    """Constructor arguments of the reference JammerEnv (jammer_env.py:94-127)."""
    M: int = 4
    N: int = 20
    arena_size: float = 200.0
    max_steps: int = 200
    dt: float = 0.5
    v_max: float = 5.0
    v_enemy: float = 2.0
    enemy_mode: str = "random_walk"
    P_tx_dbm: float = 20.0
    P_sens_dbm: float = -90.0
    P_jammer_dbm: float = 30.0
    P_jam_thresh_dbm: float = -70.0
    omega_1: float = 1.0
    omega_2: float = 0.3
    omega_3: float = 0.2
    omega_4: float = 0.1
    omega_5: float = 0.2
    eps: float = 30.0
    min_samples: int = 2
    K_recompute: int = 10
    min_lambda2_initial: float = 0.5
    max_reset_attempts: int = 100
    random_jammer_start: bool = False
    lambda2_rounds: int = 0
    omega_max: float = 0.5          # EnemySwarm default (enemy_swarm.py:60); JammerEnv does not expose it
    step_schedule: str = "auto"     # N <= 64: "auto", "fused" (one launch per step) or "split" (three, longest solve first)
    recluster: bool = True          # False: step() never re-clusters (the table of the last reset stays): "step + reward only"
    solve_priority: bool = True     # split schedule: the issue priority of an eigen-solve follows the work it has left
    reset_shortcut: bool = True     # False (testing): the general reset kernel even where the un-jammed graph is known to be complete
    lambda2_solver: str = "auto"    # split schedule, N > 32: "direct" (Householder), "lanczos" (iteration first); "auto": direct up to N = 64

    @staticmethod
    def fast_preset(N: int = 10, M: int = 4, **kw) -> "EnvConfig":
        """Environment of the reference's `fast` preset (training/config.py:258-272)."""
        base = dict(M=M, N=N, arena_size=200.0, max_steps=200, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0,
                    eps=25.0, min_samples=2, K_recompute=10, random_jammer_start=True)
        base.update(kw)
        return EnvConfig(**base)

    def to_params(self) -> MarlEnvParams:
        if self.enemy_mode not in ("random_walk", "coordinated_turn"):
            raise ValueError("enemy_mode must be 'random_walk' or 'coordinated_turn' (jammer_env.py:103)")
        p = MarlEnvParams()
        p.enemy_mode = _lib.ENEMY_COORDINATED_TURN if self.enemy_mode == "coordinated_turn" else _lib.ENEMY_RANDOM_WALK
        p.omega_max = float(self.omega_max)
        p.step_schedule = {"auto": _lib.SCHEDULE_AUTO, "fused": _lib.SCHEDULE_FUSED, "split": _lib.SCHEDULE_SPLIT}[self.step_schedule]
        p.N, p.M = int(self.N), int(self.M)
        p.max_steps, p.K_recompute, p.min_samples = int(self.max_steps), int(self.K_recompute), int(self.min_samples)
        p.random_jammer_start = int(bool(self.random_jammer_start))
        p.max_reset_attempts = int(self.max_reset_attempts)
        p.lambda2_rounds = int(self.lambda2_rounds) | {"auto": 0, "direct": _lib.LAMBDA2_DIRECT,
                                                       "lanczos": _lib.LAMBDA2_LANCZOS}[self.lambda2_solver]
        p.stage_flags = (0 if self.recluster else _lib.STAGE_NO_RECLUSTER) | (0 if self.solve_priority else _lib.STAGE_NO_SOLVE_PRIORITY) | \
            (0 if self.reset_shortcut else _lib.STAGE_GENERAL_RESET)
        p.arena_size, p.dt, p.v_max, p.v_enemy = float(self.arena_size), float(self.dt), float(self.v_max), float(self.v_enemy)
        p.P_tx_w, p.P_sens_w = db_to_watts(self.P_tx_dbm), db_to_watts(self.P_sens_dbm)
        p.P_jam_w, p.P_thr_w = db_to_watts(self.P_jammer_dbm), db_to_watts(self.P_jam_thresh_dbm)
        for i, w in enumerate((self.omega_1, self.omega_2, self.omega_3, self.omega_4, self.omega_5)):
            p.omega[i] = float(w)
        p.kappa, p.eps = 0.05, float(self.eps)
        p.drift_threshold, p.spread, p.noise_std = 30.0, 10.0, 0.1
        p.min_lambda2_initial = float(self.min_lambda2_initial)
        _lib.check(_lib.load().marl_params_init(C.byref(p)), "marl_params_init")
        return p


_DT = {"f64": torch.float64, "f32": torch.float32, "i32": torch.int32}
_STATE_SPEC = {   # name -> (dtype, shape as function of (E, N, M))
    "enemy_x": ("f64", "EN"), "enemy_y": ("f64", "EN"), "enemy_vx": ("f64", "EN"), "enemy_vy": ("f64", "EN"),
    "enemy_omega": ("f64", "EN"),
    "jam_x": ("f64", "EM"), "jam_y": ("f64", "EM"), "jam_band": ("i32", "EM"),
    "last_vx": ("f32", "EM"), "last_vy": ("f32", "EM"),
    "step_count": ("i32", "E"), "enemy_band": ("i32", "E"), "tick": ("i32", "E"), "lambda2_initial": ("f64", "E"),
    "labels": ("i32", "EN"), "cent_x": ("f64", "EK"), "cent_y": ("f64", "EK"), "cent_size": ("i32", "EK"),
    "n_clusters": ("i32", "E"), "ncent": ("i32", "E"), "prev_x": ("f64", "EK"), "prev_y": ("f64", "EK"),
    "prev_n_clusters": ("i32", "E"), "prev_ncent": ("i32", "E"), "assign": ("i32", "EM"),
    "solve_np": ("i32", "E"), "solve_hist": ("i32", "H"),       # front -> solve hand-over (N <= 64 only)
    "episode_return": ("f64", "E"),
}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BatchedJammerEnv:
    """E reference environments advanced in lockstep on one GPU.

    step(actions) -> (obs, reward, terminated, truncated, info), all device
    tensors of leading dimension E.  With `auto_reset=True` an env whose episode
    was truncated is reset inside the same call and `obs` holds its first
    observation of the next episode, as the reference trainer does
    (training/trainer.py:248-278); the terminal-step metrics stay in `info`
    (`lambda2_current`, `episode_lambda2_initial`).

    `num_shards=S` (N <= 64) steps the batch as S parts on S HIP streams of their own, so that the front
    kernel of one part runs beside the eigen-solves of another; the envs are independent
    (environment/jammer_env.py:380-545), so the results are those of one launch.  step() then does not wait:
    call fork() after producing actions on the current stream (the parts' streams wait for it) and join()
    before reading results on the current stream; step after step may be issued without either.
    """

    AUTO_RESET_EPISODE = 0x5EED5EED

    def __init__(self, num_envs: int, config: Optional[EnvConfig] = None, seed: int = 0, device="cuda",
                 auto_reset: bool = True, env_offset: int = 0, track_avg_power: bool = False,
                 debug_bits: bool = False, write_obs: bool = True, num_shards: int = 1, **cfg_kwargs):
        self.cfg = config if config is not None else EnvConfig(**cfg_kwargs)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.MarlError("BatchedJammerEnv needs a HIP device; there is no CPU implementation")
        self.device = _lib.normalise_device(device)
        self.E, self.N, self.M = int(num_envs), self.cfg.N, self.cfg.M
        self.seed, self.env_offset, self.auto_reset = int(seed), int(env_offset), auto_reset
        self.write_obs = bool(write_obs)      # False: step() leaves the observation out (MarlStepIO.obs = NULL)
        self.params = self.cfg.to_params()
        nbytes = self.lib.marl_state_bytes(C.byref(self.params), self.E)
        if nbytes == 0:
            raise _lib.MarlError("marl_state_bytes: unsupported sizes")
        with torch.cuda.device(self.device):
            self._buf = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            base = (self._buf.data_ptr() + 255) & ~255
            self.state = MarlEnvState()
            _lib.check(self.lib.marl_state_bind(C.byref(self.state), C.byref(self.params), self.E,
                                                C.c_void_p(base)), "marl_state_bind")
            dims = {"E": (self.E,), "EN": (self.E, self.N), "EM": (self.E, self.M), "EK": (self.E, self.N + 1)}
            self.views: Dict[str, torch.Tensor] = {}
            dims["H"] = (1344,)
            for name, (dt, shp) in _STATE_SPEC.items():
                if getattr(self.state, name) is None:
                    continue
                off = getattr(self.state, name) - self._buf.data_ptr()
                shape = dims[shp]
                dtype = _DT[dt]
                nb = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
                self.views[name] = self._buf[off:off + nb].view(dtype).view(shape)
            self.obs = torch.zeros(self.E, self.M, 5, dtype=torch.float32, device=self.device)
            self.reward = torch.zeros(self.E, dtype=torch.float32, device=self.device)
            self._trunc_u8 = torch.zeros(self.E, dtype=torch.uint8, device=self.device)
            self.lambda2 = torch.zeros(self.E, dtype=torch.float64, device=self.device)
            self.reward_terms = torch.zeros(self.E, 5, dtype=torch.float32, device=self.device)
            self.avg_power = torch.zeros(self.E, dtype=torch.float64, device=self.device) if track_avg_power else None
            W = (self.N + 31) // 32
            self.adj_bits = torch.zeros(self.E, self.N, W, dtype=torch.int32, device=self.device) if debug_bits else None
            self.jam_bits = torch.zeros(self.E, self.N, W, dtype=torch.int32, device=self.device) if debug_bits else None
            self._terminated = torch.zeros(self.E, dtype=torch.bool, device=self.device)
            self._shards = []            # (first, count, MarlEnvState, stream) per part when num_shards > 1
            S = int(num_shards)
            if S < 1 or S > self.E:
                raise ValueError("num_shards must be in 1..num_envs")
            if S > 1:
                sb = self.lib.marl_shard_scratch_bytes()
                self._shard_scratch = torch.zeros(S * sb + 256, dtype=torch.uint8, device=self.device)
                sbase = (self._shard_scratch.data_ptr() + 255) & ~255
                self._ep_l0 = torch.zeros(self.E, dtype=torch.float64, device=self.device)
                self._ep_step = torch.zeros(self.E, dtype=torch.int32, device=self.device)
                first = 0
                for k in range(S):
                    count = self.E // S + (1 if k < self.E % S else 0)
                    st = MarlEnvState()
                    _lib.check(self.lib.marl_state_shard(C.byref(self.params), C.byref(self.state), first, count,
                                                         C.c_void_p(sbase + k * sb), C.byref(st)), "marl_state_shard")
                    self._shards.append((first, count, st, torch.cuda.Stream(device=self.device)))
                    first += count
                self._part_streams = (C.c_void_p * S)(*[sh[3].cuda_stream for sh in self._shards])
                self._part_scratch = C.c_void_p(sbase)
                # The reset kernel is the only one here that is built with a private (scratch) segment (68 bytes that its
                # code never touches), and a HIP queue sets scratch up at the first launch that asks for it - some hundred
                # microseconds per queue, which would otherwise fall into the first auto-reset of a rollout.  A masked
                # reset that selects no env (every block returns at once, the state is not touched) pays it here.
                none = torch.zeros(self.E, dtype=torch.uint8, device=self.device)
                rio = MarlResetIO()
                rio.mask, rio.obs = none.data_ptr(), self.obs.data_ptr()
                _lib.check(self.lib.marl_env_reset_parts(C.byref(self.params), C.byref(self.state), C.byref(rio),
                                                         S, self._part_streams, self._part_scratch), "marl_env_reset_parts")
                for sh in self._shards:
                    sh[3].synchronize()
        self._episode = 0
        # Host-side bounds on the envs' step counters (min, max); None = unknown.  While the largest
        # counter cannot have reached max_steps no env is truncated and the masked auto-reset launch is
        # left out.  Known only while every state change goes through this object's eager calls: a HIP
        # graph replays launches without running this code, so the first call under stream capture
        # turns the bookkeeping off for good.
        self._count_bounds: Optional[list] = None
        self._track_counts = True

    # -- parts on their own streams -------------------------------------------
    def fork(self):
        """The parts' streams wait for the work queued on the current stream so far."""
        cur = torch.cuda.current_stream(self.device)
        for _, _, _, stream in self._shards:
            stream.wait_stream(cur)

    def join(self):
        """The current stream waits for the steps queued on the parts' streams so far."""
        cur = torch.cuda.current_stream(self.device)
        for _, _, _, stream in self._shards:
            cur.wait_stream(stream)

    # -- reset ---------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, mask: Optional[torch.Tensor] = None, *,
              enemy_pos=None, enemy_vel=None, enemy_band=None, jam_init=None, jam_band=None,
              enemy_omega=None, out_obs: Optional[torch.Tensor] = None):
        """JammerEnv.reset for the masked envs (all when mask is None).  The
        keyword tensors override the on-device random draws (enemy_pos/enemy_vel
        [E,N,2] f64, enemy_band [E] i32, jam_init [E,M,2] f64, jam_band [E,M] i32; coordinated
        turn: enemy_omega [E,N] f64)."""
        if seed is not None:
            self.seed = int(seed)
            self._episode = 0
        io = MarlResetIO()
        keep = []
        self.join()                     # a reset runs on the current stream, after every step queued so far

        def dev(t, dtype):
            if t is None:
                return None
            t = torch.as_tensor(t, dtype=dtype, device=self.device).contiguous()
            keep.append(t)
            return t.data_ptr()

        if mask is not None:
            io.mask = dev(mask.to(torch.uint8) if isinstance(mask, torch.Tensor) else np.asarray(mask, np.uint8), torch.uint8)
        io.enemy_pos = dev(enemy_pos, torch.float64)
        io.enemy_vel = dev(enemy_vel, torch.float64)
        io.enemy_omega = dev(enemy_omega, torch.float64)
        io.enemy_band = dev(enemy_band, torch.int32)
        io.jam_init = dev(jam_init, torch.float64)
        io.jam_band = dev(jam_band, torch.int32)
        obs = self.obs if out_obs is None else out_obs
        io.obs = obs.data_ptr()
        io.seed, io.episode, io.env_offset = self.seed, self._episode, self.env_offset
        self._episode += 1
        with torch.cuda.device(self.device):
            _lib.check(self.lib.marl_env_reset(C.byref(self.params), C.byref(self.state), C.byref(io), _stream()),
                       "marl_env_reset")
        self._keepalive = keep
        self.fork()
        if mask is None:
            self._count_bounds = [0, 0]
        elif self._count_bounds is not None:
            self._count_bounds[0] = 0
        v = self.views
        info = dict(lambda2_initial=v["lambda2_initial"], lambda2_current=v["lambda2_initial"],
                    enemy_band=v["enemy_band"], n_clusters=v["n_clusters"], step=v["step_count"])
        return obs, info

    # -- step ----------------------------------------------------------------
    def step(self, actions: torch.Tensor, noise: Optional[torch.Tensor] = None, *,
             out_obs: Optional[torch.Tensor] = None, out_reward: Optional[torch.Tensor] = None,
             out_truncated: Optional[torch.Tensor] = None, episode_stats: Optional[torch.Tensor] = None,
             episode_info: bool = True, solve_events: Optional[Tuple[torch.cuda.Event, torch.cuda.Event]] = None,
             out_reward_terms: Optional[torch.Tensor] = None, out_avg_power: Optional[torch.Tensor] = None):
        """One JammerEnv.step per env.  `actions` [E,M,3] float32 on the device.
        `noise` [E,N,2] float64 replaces the on-device enemy noise generator.
        out_* let the caller have results written straight into its own storage
        (e.g. slot t of a rollout buffer; `out_reward_terms` [E,5] float32 and `out_avg_power` [E] float64, the
        info-only mean jamming power of jammer_env.py:510-560, likewise).  `episode_stats` [4] float64: every episode that ends with this
        step adds 1, its return, its lambda2 reduction (%) and its length (training/trainer.py:248-268).
        `episode_info=False` leaves the pre-reset copies `episode_lambda2_initial` / `episode_step` out of
        `info` (two device copies per auto-reset).  `solve_events`: two timing events that the library records
        around the launch holding the eigen-solves (measurement; one launch sequence only, not with parts)."""
        if actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(actions.shape) != (self.E, self.M, 3):
            raise ValueError(f"actions must be [{self.E},{self.M},3], got {tuple(actions.shape)}")
        if noise is not None:
            noise = noise.to(device=self.device, dtype=torch.float64).contiguous()
            if tuple(noise.shape) != (self.E, self.N, 2):
                raise ValueError(f"noise must be [{self.E},{self.N},2]")
        obs = self.obs if out_obs is None else out_obs
        reward = self.reward if out_reward is None else out_reward
        trunc = self._trunc_u8 if out_truncated is None else out_truncated
        stats = None
        if episode_stats is not None:
            if episode_stats.dtype != torch.float64 or episode_stats.numel() < 4 or episode_stats.device != self.device:
                raise ValueError("episode_stats must be a float64 device tensor of 4 elements")
            stats = episode_stats.data_ptr()
        with_obs = self.write_obs or out_obs is not None
        if solve_events is not None and self._shards:
            raise ValueError("solve_events time one launch sequence: not available with num_shards > 1")

        io = MarlStepIO()
        io.actions, io.noise = actions.data_ptr(), _ptr(noise)
        io.obs = obs.data_ptr() if with_obs else None
        io.reward, io.truncated = reward.data_ptr(), trunc.data_ptr()
        terms = self.reward_terms if out_reward_terms is None else out_reward_terms
        avg_power = self.avg_power if out_avg_power is None else out_avg_power
        if terms.dtype != torch.float32 or terms.numel() != self.E * 5 or not terms.is_contiguous():
            raise ValueError(f"out_reward_terms must be contiguous float32 [{self.E},5]")
        if avg_power is not None and (avg_power.dtype != torch.float64 or avg_power.numel() != self.E
                                      or not avg_power.is_contiguous()):
            raise ValueError(f"out_avg_power must be contiguous float64 [{self.E}]")
        io.lambda2, io.reward_terms = self.lambda2.data_ptr(), terms.data_ptr()
        io.avg_power_dbm = _ptr(avg_power)
        io.adj_bits, io.jam_bits = _ptr(self.adj_bits), _ptr(self.jam_bits)
        io.episode_stats = stats
        io.seed, io.env_offset = self.seed, self.env_offset
        with torch.cuda.device(self.device):
            if self._shards:
                _lib.check(self.lib.marl_env_step_parts(C.byref(self.params), C.byref(self.state), C.byref(io),
                                                        len(self._shards), self._part_streams, self._part_scratch),
                           "marl_env_step_parts")
            elif solve_events is not None:
                for ev in solve_events:
                    ev.record()               # torch creates the HIP event on its first record
                _lib.check(self.lib.marl_env_step_timed(C.byref(self.params), C.byref(self.state), C.byref(io), _stream(),
                                                        C.c_void_p(solve_events[0].cuda_event),
                                                        C.c_void_p(solve_events[1].cuda_event)), "marl_env_step_timed")
            else:
                _lib.check(self.lib.marl_env_step(C.byref(self.params), C.byref(self.state), C.byref(io), _stream()),
                           "marl_env_step")
        v = self.views
        info: Dict[str, Any] = dict(lambda2_current=self.lambda2, lambda2_initial=v["lambda2_initial"],
                                    enemy_band=v["enemy_band"], n_clusters=v["n_clusters"],
                                    step=v["step_count"], reward_terms=terms)
        if avg_power is not None:
            info["avg_jamming_power_dbm"] = avg_power
        truncated = trunc.view(torch.bool) if trunc.dtype == torch.uint8 else trunc
        if self._track_counts and torch.cuda.is_current_stream_capturing():
            self._track_counts, self._count_bounds = False, None
        b = self._count_bounds if self._track_counts else None
        if b is not None:
            b[0] += 1
            b[1] += 1
        if self.auto_reset and b is not None and b[1] < self.cfg.max_steps:
            # no env can be truncated yet: nothing to reset, the episode values are the current ones
            info["episode_lambda2_initial"] = v["lambda2_initial"]
            info["episode_step"] = v["step_count"]
        elif self.auto_reset:
            if episode_info and not self._shards:
                info["episode_lambda2_initial"] = v["lambda2_initial"].clone()
                info["episode_step"] = v["step_count"].clone()
            elif episode_info:
                info["episode_lambda2_initial"], info["episode_step"] = self._ep_l0, self._ep_step
            if b is not None:
                # envs at max_steps start over; if even the smallest counter is there, all of them do
                self._count_bounds = [0, 0] if b[0] >= self.cfg.max_steps else [0, self.cfg.max_steps - 1]
            # The draws of an auto-reset are keyed by the env's step clock (`tick`, never rewound) alone: the
            # episode word stays constant, so that the launch is the same whether it is issued eagerly, left
            # out because nothing could be done, or replayed from a captured graph.
            rio = MarlResetIO()
            rio.mask = trunc.data_ptr()
            rio.obs = io.obs
            rio.seed, rio.episode, rio.env_offset = self.seed, self.AUTO_RESET_EPISODE, self.env_offset
            with torch.cuda.device(self.device):
                if self._shards:
                    if episode_info:
                        for first, count, _, stream in self._shards:      # pre-reset copies, on the part's own stream
                            with torch.cuda.stream(stream):
                                self._ep_l0[first:first + count].copy_(v["lambda2_initial"][first:first + count])
                                self._ep_step[first:first + count].copy_(v["step_count"][first:first + count])
                    _lib.check(self.lib.marl_env_reset_parts(C.byref(self.params), C.byref(self.state), C.byref(rio),
                                                             len(self._shards), self._part_streams, self._part_scratch),
                               "marl_env_reset_parts")
                else:
                    _lib.check(self.lib.marl_env_reset(C.byref(self.params), C.byref(self.state), C.byref(rio),
                                                       _stream()), "marl_env_reset")
        return obs, reward, self._terminated, truncated, info

    def reward_per_agent(self, lambda2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """RewardCalculator.compute_per_agent (reward.py:308-360) for the current state and the lambda2 of
        the last step (or the [E] float64 tensor given): [E, M] float32."""
        v = self.views
        out = torch.empty(self.E, self.M, dtype=torch.float32, device=self.device)
        lam = self.lambda2 if lambda2 is None else lambda2.to(self.device, torch.float64).contiguous()
        if tuple(lam.shape) != (self.E,):
            raise ValueError(f"lambda2 must be [{self.E}]")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.marl_reward_per_agent(
                C.byref(self.params), self.E, lam.data_ptr(), v["lambda2_initial"].data_ptr(),
                v["enemy_band"].data_ptr(), v["jam_x"].data_ptr(), v["jam_y"].data_ptr(), v["jam_band"].data_ptr(),
                v["last_vx"].data_ptr(), v["last_vy"].data_ptr(), v["cent_x"].data_ptr(), v["cent_y"].data_ptr(),
                v["n_clusters"].data_ptr(), v["ncent"].data_ptr(), v["assign"].data_ptr(), out.data_ptr(),
                _stream()), "marl_reward_per_agent")
        return out

    def compute_lambda2(self, disrupted: bool = True) -> torch.Tensor:
        """JammerEnv._compute_lambda2 (jammer_env.py:464-508) for the current state of every env, through the
        stage ops: FSPL adjacency, minus the disrupted links when `disrupted`, Laplacian lambda2.  [E] float64."""
        v = self.views
        W = (self.N + 31) // 32
        with torch.cuda.device(self.device):
            adj = torch.zeros(self.E, self.N, W, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.marl_adjacency_bits(C.byref(self.params), self.E, v["enemy_x"].data_ptr(),
                                                    v["enemy_y"].data_ptr(), v["enemy_band"].data_ptr(),
                                                    adj.data_ptr(), _stream()), "marl_adjacency_bits")
            if disrupted:
                jam = torch.zeros_like(adj)
                _lib.check(self.lib.marl_jam_mask_bits(C.byref(self.params), self.E, v["enemy_x"].data_ptr(),
                                                       v["enemy_y"].data_ptr(), v["enemy_band"].data_ptr(),
                                                       v["jam_x"].data_ptr(), v["jam_y"].data_ptr(),
                                                       v["jam_band"].data_ptr(), jam.data_ptr(), _stream()),
                           "marl_jam_mask_bits")
                adj = (adj & ~jam).contiguous()
            out = torch.zeros(self.E, dtype=torch.float64, device=self.device)
            nb = self.lib.marl_lambda2_workspace_bytes(self.N, self.E)
            ws = torch.empty(nb, dtype=torch.uint8, device=self.device) if nb else None
            _lib.check(self.lib.marl_lambda2_from_bits(self.N, self.E, adj.data_ptr(), out.data_ptr(),
                                                       self.params.lambda2_rounds & 0xFF, _ptr(ws), _stream()),
                       "marl_lambda2_from_bits")
        return out

    def fill_noise(self) -> torch.Tensor:
        """The enemy noise the next step() draws when called with noise=None."""
        out = torch.empty(self.E, self.N, 2, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.marl_fill_noise(C.byref(self.params), C.byref(self.state), self.seed,
                                                self.env_offset, out.data_ptr(), _stream()), "marl_fill_noise")
        return out

    def state_numpy(self, e: int = 0) -> Dict[str, np.ndarray]:
        return {k: v[e].cpu().numpy() for k, v in self.views.items()}


# ---------------------------------------------------------------------------
# single-env facade with the reference's API
# ---------------------------------------------------------------------------
class _Box:
    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class _MultiDiscrete:
    """`spaces.MultiDiscrete([num_bands] * M)` as far as callers of the reference read it (`nvec`)."""

    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape

    def sample(self, rng=np.random):
        return np.array([rng.randint(0, n) for n in self.nvec], dtype=np.int64)


class _DictSpace:
    """`spaces.Dict(velocity=Box, band=MultiDiscrete)` (jammer_env.py:239-247): item access and sample()."""

    def __init__(self, **spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def sample(self):
        v = self.spaces["velocity"]
        return {"velocity": np.random.uniform(v.low, v.high, size=v.shape).astype(v.dtype),
                "band": self.spaces["band"].sample()}


class _SwarmView:
    """`env.enemy_swarm.positions / .velocities` of the reference (enemy_swarm.py:96-110)."""

    def __init__(self, env: "JammerEnv"):
        self._env = env
        self._rng = np.random.RandomState()

    @property
    def positions(self) -> np.ndarray:
        v = self._env._b.views
        return np.stack([v["enemy_x"][0].cpu().numpy(), v["enemy_y"][0].cpu().numpy()], axis=1)

    @property
    def velocities(self) -> np.ndarray:
        v = self._env._b.views
        return np.stack([v["enemy_vx"][0].cpu().numpy(), v["enemy_vy"][0].cpu().numpy()], axis=1)


class _ClustererView:
    def __init__(self, env: "JammerEnv"):
        self._env = env

    @property
    def n_clusters(self) -> int:
        return int(self._env._b.views["n_clusters"][0].item())

    @property
    def labels(self) -> np.ndarray:
        return self._env._b.views["labels"][0].cpu().numpy().astype(np.int64)


class JammerEnv:
    """Drop-in for the reference `JammerEnv` (environment/jammer_env.py:59): same
    constructor, same `reset(seed, options)` / `step(action)` return tuples and
    `info` keys.  One environment = a batch of one on the GPU."""

    metadata = {"render_modes": ["human", "rgb_array"]}

    def __init__(self, M=4, N=20, arena_size=200.0, max_steps=200, dt=0.5, v_max=5.0, v_enemy=2.0,
                 enemy_mode="random_walk", P_tx_dbm=20.0, P_sens_dbm=-90.0, P_jammer_dbm=30.0,
                 P_jam_thresh_dbm=-70.0, omega_1=1.0, omega_2=0.3, omega_3=0.2, omega_4=0.1, omega_5=0.2,
                 eps=30.0, min_samples=2, K_recompute=10, min_lambda2_initial=0.5, max_reset_attempts=100,
                 debug_mode=False, random_jammer_start=False, seed=None, render_mode=None, device="cuda"):
        self.cfg = EnvConfig(M=M, N=N, arena_size=arena_size, max_steps=max_steps, dt=dt, v_max=v_max,
                             v_enemy=v_enemy, enemy_mode=enemy_mode, P_tx_dbm=P_tx_dbm, P_sens_dbm=P_sens_dbm,
                             P_jammer_dbm=P_jammer_dbm, P_jam_thresh_dbm=P_jam_thresh_dbm, omega_1=omega_1,
                             omega_2=omega_2, omega_3=omega_3, omega_4=omega_4, omega_5=omega_5, eps=eps,
                             min_samples=min_samples, K_recompute=K_recompute,
                             min_lambda2_initial=min_lambda2_initial, max_reset_attempts=max_reset_attempts,
                             random_jammer_start=random_jammer_start)
        self.M, self.N, self.arena_size, self.max_steps = M, N, arena_size, max_steps
        self.dt, self.v_max, self.v_enemy, self.enemy_mode = dt, v_max, v_enemy, enemy_mode
        self.P_tx_dbm, self.P_sens_dbm = P_tx_dbm, P_sens_dbm
        self.P_jammer_dbm, self.P_jam_thresh_dbm = P_jammer_dbm, P_jam_thresh_dbm
        self.eps, self.min_samples, self.K_recompute = eps, min_samples, K_recompute
        self.min_lambda2_initial, self.max_reset_attempts = min_lambda2_initial, max_reset_attempts
        self.debug_mode, self.random_jammer_start, self.render_mode = debug_mode, random_jammer_start, render_mode
        self.BANDS = list(BANDS)
        self.num_bands = len(self.BANDS)
        self._b = BatchedJammerEnv(1, self.cfg, device=device, auto_reset=False, track_avg_power=True)
        self.R_jam = float(self._b.params.R_jam[2])             # default band 2.4 GHz (jammer_env.py:187-189)
        self.observation_space = _Box(0.0, 1.0, (M, 5), np.float32)
        self.action_space = _DictSpace(velocity=_Box(-v_max, v_max, (M, 2), np.float32),
                                       band=_MultiDiscrete([self.num_bands] * M))
        # RNG streams of the reference: EnemySwarm.__init__ seeds the global numpy
        # generator and its own RandomState (enemy_swarm.py:87-91); the env has a
        # second RandomState (jammer_env.py:250).
        self.enemy_swarm = _SwarmView(self)
        if seed is not None:
            np.random.seed(seed)
            self.enemy_swarm._rng = np.random.RandomState(seed)
        # constructor draws of EnemySwarm (positions, headings) advance its stream
        self.enemy_swarm._rng.uniform(arena_size * 0.1, arena_size - arena_size * 0.1, size=(N, 2))
        self.enemy_swarm._rng.uniform(0, 2 * np.pi, size=N)
        if enemy_mode == "coordinated_turn":               # _init_coordinated_turn_state (enemy_swarm.py:113-114, 133-137)
            self.enemy_swarm._rng.uniform(-self.cfg.omega_max * 0.5, self.cfg.omega_max * 0.5, size=N)
        self.clusterer = _ClustererView(self)
        self._rng = np.random.RandomState(seed)
        self.enemy_band = 2
        self.lambda2_initial = 0.0

    # -- attributes the reference exposes -------------------------------------
    @property
    def step_count(self) -> int:
        return int(self._b.views["step_count"][0].item())

    @property
    def jammer_positions(self) -> np.ndarray:
        v = self._b.views
        return np.stack([v["jam_x"][0].cpu().numpy(), v["jam_y"][0].cpu().numpy()], axis=1)

    @jammer_positions.setter
    def jammer_positions(self, value):
        value = np.asarray(value, dtype=np.float64)
        v = self._b.views
        v["jam_x"][0].copy_(torch.from_numpy(np.ascontiguousarray(value[:, 0])))
        v["jam_y"][0].copy_(torch.from_numpy(np.ascontiguousarray(value[:, 1])))

    @property
    def jammer_bands(self) -> np.ndarray:
        return self._b.views["jam_band"][0].cpu().numpy().astype(np.int32)

    @jammer_bands.setter
    def jammer_bands(self, value):
        self._b.views["jam_band"][0].copy_(torch.from_numpy(np.asarray(value, dtype=np.int32)))

    @property
    def last_velocities(self) -> np.ndarray:
        v = self._b.views
        return np.stack([v["last_vx"][0].cpu().numpy(), v["last_vy"][0].cpu().numpy()], axis=1)

    # -- reset (jammer_env.py:261-354) ------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        if seed is not None:
            self._rng = np.random.RandomState(seed)
        A, N, M = self.arena_size, self.N, self.M
        reset_seed = seed
        srng = self.enemy_swarm
        obs = None
        for attempt in range(self.max_reset_attempts):
            if reset_seed is not None:                                    # enemy_swarm.py:259-260
                srng._rng = np.random.RandomState(reset_seed)
            margin = A * 0.1
            epos = srng._rng.uniform(margin, A - margin, size=(N, 2))     # enemy_swarm.py:266-270
            ang = srng._rng.uniform(0, 2 * np.pi, size=N)
            evel = np.column_stack([self.v_enemy * np.cos(ang), self.v_enemy * np.sin(ang)])
            eomega = None
            if self.enemy_mode == "coordinated_turn":                     # enemy_swarm.py:133-137
                wmax = self.cfg.omega_max
                eomega = srng._rng.uniform(-wmax * 0.5, wmax * 0.5, size=N)
            self.enemy_band = int(self._rng.randint(0, self.num_bands))   # jammer_env.py:287
            # One attempt of the reference's loop = one fit + lambda2_initial.  The jammer draws
            # are only consumed after the loop ends, so probe with placeholder jammers; the
            # cluster tables are then put back so that the final call below redoes the last
            # attempt's fit from the same "previous centroids" the reference would have.
            last = attempt == self.max_reset_attempts - 1
            ok = self._probe_reset(epos, evel, restore_always=last)
            if ok:
                break
            reset_seed = None if seed is None else seed + attempt + 1
        self.R_jam = float(self._b.params.R_jam[self.enemy_band])
        if self.random_jammer_start:
            jinit = self._rng.uniform(0.1 * A, 0.9 * A, size=(M, 2))      # jammer_env.py:324-326
        else:
            jinit = np.array([np.random.uniform(-10.0, 10.0, size=2) for _ in range(M)])  # dbscan_clustering.py:399
        jb = self._rng.randint(0, self.num_bands, size=M).astype(np.int32)  # jammer_env.py:339
        obs_t, _ = self._b.reset(enemy_pos=epos[None], enemy_vel=evel[None],
                                 enemy_band=np.array([self.enemy_band], np.int32),
                                 jam_init=jinit[None], jam_band=jb[None],
                                 enemy_omega=None if eomega is None else eomega[None])
        torch.cuda.synchronize()
        self.lambda2_initial = float(self._b.views["lambda2_initial"][0].item())
        obs = obs_t[0].cpu().numpy()
        info = {"lambda2_initial": self.lambda2_initial, "lambda2_current": self.lambda2_initial,
                "enemy_band": self.enemy_band, "n_clusters": self.clusterer.n_clusters, "step": 0}
        return obs, info

    def _probe_reset(self, epos, evel, restore_always: bool) -> bool:
        """Cluster + lambda2_initial for these enemy draws; True when the reference's loop would break."""
        v = self._b.views
        names = ("cent_x", "cent_y", "cent_size", "n_clusters", "ncent",
                 "prev_x", "prev_y", "prev_n_clusters", "prev_ncent")
        saved = {k: v[k].clone() for k in names}
        self._b.reset(enemy_pos=epos[None], enemy_vel=evel[None],
                      enemy_band=np.array([self.enemy_band], np.int32),
                      jam_init=np.zeros((1, self.M, 2)), jam_band=np.zeros((1, self.M), np.int32))
        ok = float(v["lambda2_initial"][0].item()) >= self.min_lambda2_initial
        if ok or restore_always:
            for k, t in saved.items():
                v[k].copy_(t)
        return ok

    # -- step (jammer_env.py:356-457) -------------------------------------------
    def step(self, action):
        if isinstance(action, dict):
            velocity = np.asarray(action["velocity"])
            bands = np.asarray(action["band"])
            a = np.zeros((self.M, 3), dtype=np.float32)
            a[:, :2] = velocity
            a[:, 2] = bands
        else:
            a = np.asarray(action, dtype=np.float32)
        if self.enemy_mode == "coordinated_turn":                         # enemy_swarm.py:228, one draw per drone
            noise = np.zeros((self.N, 2))
            noise[:, 0] = [self.enemy_swarm._rng.normal(0, 0.1 * 0.5) for _ in range(self.N)]
        else:
            noise = self.enemy_swarm._rng.normal(0, 0.1, size=(self.N, 2))    # enemy_swarm.py:165
        act_t = torch.from_numpy(np.ascontiguousarray(a[None])).to(self._b.device)
        noise_t = torch.from_numpy(noise[None]).to(self._b.device)
        obs_t, rew_t, _, trunc_t, info_t = self._b.step(act_t, noise_t)
        torch.cuda.synchronize()
        lam = float(info_t["lambda2_current"][0].item())
        lam0 = self.lambda2_initial
        raw_reduction = 1.0 - (lam / lam0) if lam0 > 1e-6 else 0.0
        terms = info_t["reward_terms"][0].cpu().numpy().astype(np.float64)
        c = self.cfg
        reward = float(rew_t[0].item())
        breakdown = {"lambda2_reduction": c.omega_1 * terms[0], "band_match": c.omega_2 * terms[1],
                     "proximity": c.omega_3 * terms[2], "energy_penalty": -c.omega_4 * terms[3],
                     "overlap_penalty": -c.omega_5 * terms[4], "total": reward}   # reward.py:398-414
        info = {"lambda2_initial": lam0, "lambda2_current": lam, "lambda2_reduction": raw_reduction,
                "lambda2_reduction_raw": raw_reduction, "enemy_band": self.enemy_band,
                "n_clusters": self.clusterer.n_clusters, "step": self.step_count,
                "reward_components": breakdown, "jammer_positions": self.jammer_positions,
                "enemy_positions": self.enemy_swarm.positions,
                "avg_jamming_power_dbm": float(info_t["avg_jamming_power_dbm"][0].item())}
        return obs_t[0].cpu().numpy(), reward, False, bool(trunc_t[0].item()), info

    def _compute_lambda2(self, disrupted: bool = True) -> float:
        """jammer_env.py:464-508: lambda2 of the current swarm graph, with or without the jammed links."""
        return float(self._b.compute_lambda2(disrupted)[0].item())

    def get_global_observation(self) -> np.ndarray:
        """Mean-pooled observation for the critic (observation.py:335-350)."""
        return self._b.obs[0].mean(dim=0).cpu().numpy().astype(np.float32)

    def render(self):
        return None

    def close(self):
        pass
