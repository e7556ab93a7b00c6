"""Thin PyTorch custom ops (`torch.ops.marl_amd.*`) over the C ABI of libmarl_env.so.

Each op launches one HIP kernel on torch's current stream and returns / mutates device
tensors; fake (meta) implementations are registered so the ops trace under torch.compile.
The stage ops mirror the reference's pure functions:

    adjacency_bits      physics/communication_graph.py:41-105   compute_adjacency_matrix
    jam_mask_bits       physics/jamming.py:151-247              compute_disrupted_links
    lambda2_from_bits   physics/communication_graph.py:184-286  compute_laplacian + compute_lambda2
    dbscan_labels       clustering/dbscan_clustering.py:78-144  DBSCANClusterer.fit
    gae                 agents/rollout_buffer.py:151-204        compute_returns_and_advantages
    reward_per_agent    environment/reward.py:308-396           RewardCalculator.compute_per_agent
    reward_obs          environment/reward.py:106-176, observation.py:264-332   compute + build_vectorized
    env_step            environment/jammer_env.py:356-457       JammerEnv.step (state mutated in place)

Bit matrices are int32 tensors [E, N, ceil(N/32)]: row i, bit j of word j // 32.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import List, Optional, Tuple

import torch

from . import _lib
from .env import BatchedJammerEnv, EnvConfig


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@lru_cache(maxsize=64)
def _params(N: int, M: int, P_tx_dbm: float, P_sens_dbm: float, P_jammer_dbm: float, P_jam_thresh_dbm: float,
            eps: float, min_samples: int, arena_size: float):
    return EnvConfig(N=N, M=M, P_tx_dbm=P_tx_dbm, P_sens_dbm=P_sens_dbm, P_jammer_dbm=P_jammer_dbm,
                     P_jam_thresh_dbm=P_jam_thresh_dbm, eps=eps, min_samples=min_samples,
                     arena_size=arena_size).to_params()


def _check_dev(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError("marl_amd ops need contiguous tensors on the HIP device")


@torch.library.custom_op("marl_amd::adjacency_bits", mutates_args=())
def adjacency_bits(x: torch.Tensor, y: torch.Tensor, band: torch.Tensor, P_tx_dbm: float = 20.0,
                   P_sens_dbm: float = -90.0) -> torch.Tensor:
    """x, y [E,N] float64, band [E] int32 -> adjacency bits."""
    _check_dev(x, y, band)
    E, N = x.shape
    p = _params(N, 1, P_tx_dbm, P_sens_dbm, 30.0, -70.0, 30.0, 2, 200.0)
    out = torch.zeros(E, N, (N + 31) // 32, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().marl_adjacency_bits(C.byref(p), E, x.data_ptr(), y.data_ptr(), band.data_ptr(),
                                               out.data_ptr(), _stream()), "marl_adjacency_bits")
    return out


@adjacency_bits.register_fake
def _(x, y, band, P_tx_dbm=20.0, P_sens_dbm=-90.0):
    E, N = x.shape
    return x.new_empty((E, N, (N + 31) // 32), dtype=torch.int32)


@torch.library.custom_op("marl_amd::jam_mask_bits", mutates_args=())
def jam_mask_bits(x: torch.Tensor, y: torch.Tensor, band: torch.Tensor, jam_x: torch.Tensor, jam_y: torch.Tensor,
                  jam_band: torch.Tensor, P_jammer_dbm: float = 30.0, P_jam_thresh_dbm: float = -70.0) -> torch.Tensor:
    """Enemy x, y [E,N], band [E]; jammer jam_x, jam_y [E,M] float64, jam_band [E,M] int32 -> mask bits."""
    _check_dev(x, y, band, jam_x, jam_y, jam_band)
    E, N = x.shape
    M = jam_x.shape[1]
    p = _params(N, M, 20.0, -90.0, P_jammer_dbm, P_jam_thresh_dbm, 30.0, 2, 200.0)
    out = torch.zeros(E, N, (N + 31) // 32, dtype=torch.int32, device=x.device)
    _lib.check(_lib.load().marl_jam_mask_bits(C.byref(p), E, x.data_ptr(), y.data_ptr(), band.data_ptr(),
                                              jam_x.data_ptr(), jam_y.data_ptr(), jam_band.data_ptr(),
                                              out.data_ptr(), _stream()), "marl_jam_mask_bits")
    return out


@jam_mask_bits.register_fake
def _(x, y, band, jam_x, jam_y, jam_band, P_jammer_dbm=30.0, P_jam_thresh_dbm=-70.0):
    E, N = x.shape
    return x.new_empty((E, N, (N + 31) // 32), dtype=torch.int32)


@torch.library.custom_op("marl_amd::lambda2_from_bits", mutates_args=())
def lambda2_from_bits(bits: torch.Tensor, rounds: int = 0) -> torch.Tensor:
    """bits [E,N,W] int32 -> lambda2 [E] float64."""
    _check_dev(bits)
    E, N, _ = bits.shape
    lib = _lib.load()
    out = torch.zeros(E, dtype=torch.float64, device=bits.device)
    nb = lib.marl_lambda2_workspace_bytes(N, E)
    ws = torch.empty(nb, dtype=torch.uint8, device=bits.device) if nb else None
    _lib.check(lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), rounds,
                                          ws.data_ptr() if ws is not None else None, _stream()),
               "marl_lambda2_from_bits")
    return out


@lambda2_from_bits.register_fake
def _(bits, rounds=0):
    return bits.new_empty((bits.shape[0],), dtype=torch.float64)


@torch.library.custom_op("marl_amd::dbscan_labels", mutates_args=())
def dbscan_labels(x: torch.Tensor, y: torch.Tensor, eps: float, min_samples: int = 2,
                  arena_size: float = 200.0) -> List[torch.Tensor]:
    """x, y [E,N] float64 -> [labels [E,N] i32, cent_x [E,N+1] f64, cent_y, cent_size i32, n_clusters [E], ncent [E]]."""
    _check_dev(x, y)
    E, N = x.shape
    p = _params(N, 1, 20.0, -90.0, 30.0, -70.0, eps, min_samples, arena_size)
    dev = x.device
    labels = torch.zeros(E, N, dtype=torch.int32, device=dev)
    cx = torch.zeros(E, N + 1, dtype=torch.float64, device=dev)
    cy = torch.zeros(E, N + 1, dtype=torch.float64, device=dev)
    csz = torch.zeros(E, N + 1, dtype=torch.int32, device=dev)
    ncl = torch.zeros(E, dtype=torch.int32, device=dev)
    nce = torch.zeros(E, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().marl_dbscan(C.byref(p), E, x.data_ptr(), y.data_ptr(), labels.data_ptr(), cx.data_ptr(),
                                       cy.data_ptr(), csz.data_ptr(), ncl.data_ptr(), nce.data_ptr(), _stream()),
               "marl_dbscan")
    return [labels, cx, cy, csz, ncl, nce]


@dbscan_labels.register_fake
def _(x, y, eps, min_samples=2, arena_size=200.0):
    E, N = x.shape
    i32, f64 = torch.int32, torch.float64
    return [x.new_empty((E, N), dtype=i32), x.new_empty((E, N + 1), dtype=f64), x.new_empty((E, N + 1), dtype=f64),
            x.new_empty((E, N + 1), dtype=i32), x.new_empty((E,), dtype=i32), x.new_empty((E,), dtype=i32)]


@torch.library.custom_op("marl_amd::gae", mutates_args=())
def gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_value: torch.Tensor,
        last_done: torch.Tensor, gamma: float = 0.99, lam: float = 0.95) -> List[torch.Tensor]:
    """rewards, values, dones [T,E] float32; last_value, last_done [E] -> [advantages, returns] [T,E]."""
    _check_dev(rewards, values, dones, last_value, last_done)
    T, E = rewards.shape
    adv, ret = torch.zeros_like(rewards), torch.zeros_like(rewards)
    _lib.check(_lib.load().marl_gae(T, E, rewards.data_ptr(), values.data_ptr(), dones.data_ptr(),
                                    last_value.data_ptr(), last_done.data_ptr(), gamma, lam, adv.data_ptr(),
                                    ret.data_ptr(), _stream()), "marl_gae")
    return [adv, ret]


@gae.register_fake
def _(rewards, values, dones, last_value, last_done, gamma=0.99, lam=0.95):
    return [torch.empty_like(rewards), torch.empty_like(rewards)]


@torch.library.custom_op("marl_amd::reward_per_agent", mutates_args=())
def reward_per_agent(state: torch.Tensor, handle: int, lambda2: torch.Tensor) -> torch.Tensor:
    """Per-jammer reward [E,M] float32 of the registered env's current state for the given lambda2 [E] f64."""
    env = _ENVS[handle]
    if state.data_ptr() != env._buf.data_ptr():
        raise ValueError("state must be the buffer of the registered env")
    _check_dev(lambda2)
    return env.reward_per_agent(lambda2)


@reward_per_agent.register_fake
def _(state, handle, lambda2):
    env = _ENVS[handle]
    return lambda2.new_empty((env.E, env.M), dtype=torch.float32)


@torch.library.custom_op("marl_amd::reward_obs", mutates_args=())
def reward_obs(state: torch.Tensor, handle: int, lambda2: torch.Tensor) -> List[torch.Tensor]:
    """Reward stage + observation stage of the registered env's current state for the given lambda2 [E] f64:
    [reward [E] f32, reward_terms [E,5] f32, obs [E,M,5] f32, avg_jamming_power_dbm [E] f64]."""
    env = _ENVS[handle]
    if state.data_ptr() != env._buf.data_ptr():
        raise ValueError("state must be the buffer of the registered env")
    _check_dev(lambda2)
    v, dev = env.views, env.device
    io = _lib.MarlRewardObsIO()
    rew = torch.empty(env.E, dtype=torch.float32, device=dev)
    terms = torch.empty(env.E, 5, dtype=torch.float32, device=dev)
    obs = torch.empty(env.E, env.M, 5, dtype=torch.float32, device=dev)
    avgp = torch.empty(env.E, dtype=torch.float64, device=dev)
    io.lambda2, io.lambda2_initial = lambda2.data_ptr(), v["lambda2_initial"].data_ptr()
    for name in ("enemy_band", "enemy_x", "enemy_y", "jam_x", "jam_y", "jam_band", "cent_x", "cent_y", "cent_size",
                 "n_clusters", "ncent", "assign"):
        setattr(io, name, v[name].data_ptr())
    io.vx, io.vy = v["last_vx"].data_ptr(), v["last_vy"].data_ptr()
    io.reward, io.reward_terms, io.obs, io.avg_power_dbm = rew.data_ptr(), terms.data_ptr(), obs.data_ptr(), avgp.data_ptr()
    _lib.check(_lib.load().marl_reward_obs(C.byref(env.params), env.E, C.byref(io), _stream()), "marl_reward_obs")
    return [rew, terms, obs, avgp]


@reward_obs.register_fake
def _(state, handle, lambda2):
    env = _ENVS[handle]
    f32 = torch.float32
    return [lambda2.new_empty((env.E,), dtype=f32), lambda2.new_empty((env.E, 5), dtype=f32),
            lambda2.new_empty((env.E, env.M, 5), dtype=f32), lambda2.new_empty((env.E,), dtype=torch.float64)]


# -- the step itself: state lives in a BatchedJammerEnv, addressed by a small integer handle -------
_ENVS = {}


def register_env(env: BatchedJammerEnv) -> int:
    h = len(_ENVS) + 1
    _ENVS[h] = env
    return h


@torch.library.custom_op("marl_amd::env_step", mutates_args=("state", "obs", "reward", "truncated"))
def env_step(state: torch.Tensor, handle: int, actions: torch.Tensor, noise: Optional[torch.Tensor],
             obs: torch.Tensor, reward: torch.Tensor, truncated: torch.Tensor) -> None:
    """JammerEnv.step for every env of the registered batch: `state` is its device state buffer
    (mutated), obs [E,M,5] f32 / reward [E] f32 / truncated [E] u8 receive the results."""
    env = _ENVS[handle]
    if state.data_ptr() != env._buf.data_ptr():
        raise ValueError("state must be the buffer of the registered env")
    auto = env.auto_reset
    env.auto_reset = False
    try:
        env.step(actions, noise, out_obs=obs, out_reward=reward, out_truncated=truncated)
    finally:
        env.auto_reset = auto


@env_step.register_fake
def _(state, handle, actions, noise, obs, reward, truncated):
    return None
