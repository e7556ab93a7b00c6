"""
CPU oracle for the JammerEnv step path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a numpy restatement of the reference environment step
(`src/environment/jammer_env.py:356-457`) and of every function that step
calls, written stage by stage in the order the reference executes them.  It is
the checker the HIP kernels are compared with.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; nothing under `marl_amd/` does, and the product path raises when its HIP
library is missing instead of falling back to this file.

Parity status: PINNED.  Every function below is checked in
`tests/test_oracle_golden.py` against vectors produced by running the
reference's own modules in the build container (`tests/golden/make_golden.py`
is the generating script, `tests/golden/*.npz` the vectors), and against the
closed-form answers the reference's unit tests assert (K_n, path, cycle
Laplacians, dBm/W identities, jam-range formula).

The reference keeps cluster data in Python dicts.  The oracle keeps the same
information in fixed-size arrays, which is the form the GPU state uses:

    cent_xy[KS, 2], cent_size[KS]   KS = N + 1 slots, in the reference's dict
                                    iteration order: clusters 0..n_clusters-1
                                    first, then the noise entry (id -1) when
                                    one exists.
    n_clusters                      number of DBSCAN clusters (ids >= 0)
    ncent                           number of dict entries (n_clusters, +1 if a
                                    noise / fallback entry exists)
    assign[M]                       cluster id a jammer is assigned to;
                                    -1 = the noise pseudo-cluster,
                                    -2 = no assignment (empty dict)

All arithmetic that feeds a threshold test (adjacency, jamming mask, DBSCAN
neighbourhoods) is float64 in the reference's exact operation order, so the
boolean / integer outputs are bit-exact against the reference.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
from scipy import linalg as _sla

# physics/fspl.py:31 and :38-43
SPEED_OF_LIGHT = 3.0e8
FREQUENCY_BANDS = (4.33e8, 9.15e8, 2.40e9, 5.80e9)
NUM_BANDS = 4
ASSIGN_NONE = -2
NOISE_ID = -1


# --------------------------------------------------------------------------
# parameters (constructor arguments of JammerEnv, jammer_env.py:94-127, plus
# the three constants the reference fixes elsewhere)
# --------------------------------------------------------------------------
@dataclass
class EnvParams:
    M: int = 4
    N: int = 20
    arena_size: float = 200.0
    max_steps: int = 200
    dt: float = 0.5
    v_max: float = 5.0
    v_enemy: float = 2.0
    P_tx_dbm: float = 20.0
    P_sens_dbm: float = -90.0
    P_jammer_dbm: float = 30.0
    P_jam_thresh_dbm: float = -70.0
    omega: Tuple[float, float, float, float, float] = (1.0, 0.3, 0.2, 0.1, 0.2)
    eps: float = 30.0
    min_samples: int = 2
    K_recompute: int = 10
    min_lambda2_initial: float = 0.5
    max_reset_attempts: int = 100
    random_jammer_start: bool = False
    kappa: float = 0.05            # reward.py:77
    noise_std: float = 0.1         # enemy_swarm.py:61
    drift_threshold: float = 30.0  # jammer_env.py:398
    spread: float = 10.0           # jammer_env.py:335
    enemy_mode: str = "random_walk"   # jammer_env.py:103, or "coordinated_turn"
    omega_max: float = 0.5         # enemy_swarm.py:60

    @staticmethod
    def fast_preset(N: int = 10, M: int = 4) -> "EnvParams":
        """Physics of the reference's `fast` training preset
        (training/config.py:258-272 over the defaults at :18-61)."""
        return EnvParams(M=M, N=N, arena_size=200.0, max_steps=200, dt=0.5,
                         v_max=5.0, v_enemy=2.0, P_tx_dbm=20.0,
                         P_sens_dbm=-90.0, P_jammer_dbm=20.0,
                         P_jam_thresh_dbm=-40.0, eps=25.0, min_samples=2,
                         K_recompute=10, random_jammer_start=True)


# --------------------------------------------------------------------------
# physics/fspl.py
# --------------------------------------------------------------------------
def db_to_watts(power_dbm: float) -> float:
    """fspl.py:50-69"""
    return 10 ** ((power_dbm - 30) / 10)


def watts_to_db(power_watts):
    """fspl.py:72-91"""
    return 10 * np.log10(power_watts) + 30


def received_power_watts(tx_power_watts, distance, frequency, eps=1e-10):
    """fspl.py:168-206.  Operation order kept: lambda/(4*pi*d), squared, times P."""
    d = np.maximum(distance, eps)
    wavelength = SPEED_OF_LIGHT / frequency
    path_gain = (wavelength / (4 * np.pi * d)) ** 2
    return tx_power_watts * path_gain


def compute_jam_range(jammer_power_watts, jam_threshold_watts, frequency):
    """fspl.py:276-308 (compute_comm_range :241-273 is the same formula)."""
    wavelength = SPEED_OF_LIGHT / frequency
    return (wavelength / (4 * np.pi)) * np.sqrt(jammer_power_watts / jam_threshold_watts)


def pairwise_received_power(positions, tx_power_watts, frequency, eps=1e-10):
    """fspl.py:315-365"""
    diff = positions[:, None, :] - positions[None, :, :]
    distances = np.sqrt(np.sum(diff ** 2, axis=2))
    P = received_power_watts(tx_power_watts, distances, frequency, eps)
    np.fill_diagonal(P, np.inf)
    return P


# --------------------------------------------------------------------------
# physics/communication_graph.py
# --------------------------------------------------------------------------
def adjacency_matrix(positions, tx_power_watts, sensitivity_watts, frequency):
    """communication_graph.py:41-105 -> boolean (N, N), diagonal False."""
    N = positions.shape[0]
    if N == 0:
        return np.zeros((0, 0), dtype=bool)
    if N == 1:
        return np.zeros((1, 1), dtype=bool)
    P = pairwise_received_power(positions, tx_power_watts, frequency)
    A = P >= sensitivity_watts
    np.fill_diagonal(A, False)
    return A


def laplacian(adj_bool):
    """communication_graph.py:158-212: L = diag(rowsum A) - A in float64."""
    A = adj_bool.astype(np.float64)
    return np.diag(np.sum(A, axis=1)) - A


def lambda2_of_laplacian(L) -> float:
    """communication_graph.py:219-286 (dense branch, the one the env takes)."""
    N = L.shape[0]
    if N <= 1:
        return 0.0
    ev = _sla.eigh(L, subset_by_index=[0, 1], eigvals_only=True)
    return float(ev[1]) if len(ev) > 1 else 0.0


# --------------------------------------------------------------------------
# physics/jamming.py
# --------------------------------------------------------------------------
def midpoint_distances(jammer_positions, enemy_positions):
    """jamming.py:37-70 and :73-109 -> (M, N, N) distances jammer k -> midpoint(i, j)."""
    mid = (enemy_positions[:, None, :] + enemy_positions[None, :, :]) / 2
    diff = jammer_positions[:, None, None, :] - mid[None, :, :, :]
    return np.sqrt(np.sum(diff ** 2, axis=3))


def disrupted_links(jammer_positions, jammer_bands, enemy_positions, enemy_band,
                    jammer_power_watts, jam_threshold_watts):
    """jamming.py:151-247 -> boolean (N, N)."""
    M = jammer_positions.shape[0]
    N = enemy_positions.shape[0]
    if M == 0 or N <= 1:
        return np.zeros((N, N), dtype=bool)
    f = FREQUENCY_BANDS[int(enemy_band)]
    d = midpoint_distances(jammer_positions, enemy_positions)
    P = received_power_watts(jammer_power_watts, d, f)
    eff = (P >= jam_threshold_watts) & (np.asarray(jammer_bands) == enemy_band)[:, None, None]
    jam = np.any(eff, axis=0)
    np.fill_diagonal(jam, False)
    return jam


def avg_jamming_power_dbm(jammer_positions, jammer_bands, enemy_positions, enemy_band,
                          jammer_power_watts) -> float:
    """jammer_env.py:510-560 (logging metric)."""
    f = FREQUENCY_BANDS[int(enemy_band)]
    band_mask = np.asarray(jammer_bands) == enemy_band
    if not np.any(band_mask):
        return -80.0
    d = midpoint_distances(jammer_positions, enemy_positions)
    P = received_power_watts(jammer_power_watts, d, f)
    P[~band_mask, :, :] = 0.0
    Pmax = np.max(P, axis=0)
    iu = np.triu_indices_from(Pmax, k=1)
    link = Pmax[iu]
    valid = link[link > 0]
    if len(valid) == 0:
        return -80.0
    return float(watts_to_db(np.mean(valid)))


# --------------------------------------------------------------------------
# clustering/dbscan_clustering.py  (+ the scikit-learn DBSCAN it calls)
# --------------------------------------------------------------------------
def dbscan_labels(positions, eps: float, min_samples: int):
    """Labels of sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(positions)
    (call site dbscan_clustering.py:72,113).

    scikit-learn is a third-party dependency of the reference
    (requirements.txt:9, `scikit-learn>=1.2.0`, unpinned; 1.7.2 in the build
    container).  Its published algorithm: a point is a core point when its
    closed eps-ball (itself included, `sum (x-y)^2 <= eps^2` in float64) holds
    at least min_samples points; clusters are grown one at a time from the
    lowest-index unlabeled core point by a depth-first walk that only expands
    through core points, so (i) cluster ids follow the smallest core index of
    each cluster and (ii) a non-core point gets the id of the first, i.e.
    lowest-numbered, cluster one of whose core points has it as a neighbour.
    Points reached by no cluster keep -1.
    """
    N = positions.shape[0]
    labels = np.full(N, -1, dtype=np.int64)
    if N == 0:
        return labels
    diff = positions[:, None, :] - positions[None, :, :]
    d2 = diff[:, :, 0] * diff[:, :, 0] + diff[:, :, 1] * diff[:, :, 1]
    nb = d2 <= eps * eps
    core = nb.sum(axis=1) >= min_samples
    label = 0
    for s in range(N):
        if labels[s] != -1 or not core[s]:
            continue
        stack = [s]
        while stack:
            i = stack.pop()
            if labels[i] == -1:
                labels[i] = label
                if core[i]:
                    for v in np.nonzero(nb[i])[0]:
                        if labels[v] == -1:
                            stack.append(int(v))
        label += 1
    return labels


@dataclass
class ClusterState:
    """DBSCANClusterer.labels/centroids/n_clusters/_prev_centroids
    (dbscan_clustering.py:72-76) plus JammerEnv.cluster_sizes, as arrays."""
    labels: np.ndarray
    cent_xy: np.ndarray
    cent_size: np.ndarray
    n_clusters: int = 0
    ncent: int = 0
    prev_xy: np.ndarray = None
    prev_n_clusters: int = 0
    prev_ncent: int = 0

    @staticmethod
    def empty(N: int) -> "ClusterState":
        KS = N + 1
        return ClusterState(labels=np.full(N, -1, dtype=np.int64),
                            cent_xy=np.zeros((KS, 2)), cent_size=np.zeros(KS, dtype=np.int64),
                            prev_xy=np.zeros((KS, 2)))

    def cent_ids(self):
        """Cluster id of each populated slot, in dict order."""
        ids = list(range(self.n_clusters))
        if self.ncent > self.n_clusters:
            ids.append(NOISE_ID)
        return ids

    def prev_ids(self):
        ids = list(range(self.prev_n_clusters))
        if self.prev_ncent > self.prev_n_clusters:
            ids.append(NOISE_ID)
        return ids


def cluster_fit(cs: ClusterState, positions, eps, min_samples, arena_size):
    """DBSCANClusterer.fit + get_cluster_sizes + compute_centroids
    (dbscan_clustering.py:78-128, 130-144, 230-260), N >= 2.

    Centroid of a label = np.mean(rows, axis=0): a sequential float64 sum in
    index order divided by the count.  The dict is iterated as set(labels),
    which for small ints is 0..k-1 followed by -1.
    """
    N = positions.shape[0]
    labels = dbscan_labels(positions, eps, min_samples)
    # _prev_centroids = centroids.copy()                      (:116)
    cs.prev_xy = cs.cent_xy.copy()
    cs.prev_n_clusters = cs.n_clusters
    cs.prev_ncent = cs.ncent
    k = int(labels.max()) + 1 if N > 0 else 0
    k = max(k, 0)
    has_noise = bool(np.any(labels == -1))
    cent_xy = np.zeros_like(cs.cent_xy)
    cent_size = np.zeros_like(cs.cent_size)
    ids = list(range(k)) + ([NOISE_ID] if has_noise else [])
    for slot, cid in enumerate(ids):
        rows = positions[labels == cid]
        acc = np.zeros(2)
        for r in rows:                  # axis-0 reduction: sequential, index order
            acc = acc + r
        cent_xy[slot] = acc / len(rows)
        cent_size[slot] = len(rows)
    ncent = len(ids)
    if k == 0:                          # all noise -> fallback centroid (:124-126)
        cent_xy[0] = (arena_size / 2, arena_size / 2)
        cent_size[0] = N                # get_cluster_sizes(): {-1: N}
        ncent = 1
    cs.labels = labels
    cs.cent_xy = cent_xy
    cs.cent_size = cent_size
    cs.n_clusters = k
    cs.ncent = ncent
    return cs


def should_reassign(cs: ClusterState, drift_threshold: float) -> bool:
    """compute_centroid_drift + should_reassign_jammers (dbscan_clustering.py:163-204)."""
    ids = cs.cent_ids()
    pids = cs.prev_ids()
    for slot, cid in enumerate(ids):
        if cid in pids:
            pslot = pids.index(cid)
            diff = cs.cent_xy[slot] - cs.prev_xy[pslot]
            d = math.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
        else:
            d = float("inf")
        if d > drift_threshold:
            return True
    return cs.ncent != cs.prev_ncent


def assign_proportional(M: int, cs: ClusterState):
    """assign_jammers_to_clusters(..., strategy="proportional")
    (dbscan_clustering.py:263-362) -> assign[M] cluster ids."""
    assign = np.full(M, ASSIGN_NONE, dtype=np.int64)
    if M == 0 or cs.ncent == 0:
        return assign
    ids = cs.cent_ids()
    valid = [(s, c) for s, c in enumerate(ids) if c >= 0]
    if not valid:
        valid = list(enumerate(ids))
    total = sum(int(cs.cent_size[s]) for s, _ in valid)
    if total == 0:
        total = len(valid)
    j = 0
    remaining = M
    for i, (s, c) in enumerate(valid):
        size = int(cs.cent_size[s])
        if i == len(valid) - 1:
            count = remaining
        else:
            count = max(1, int(M * size / total))
            count = min(count, remaining)
        assign[j:j + count] = c
        j += count
        remaining -= count
    return assign


def jammer_initial_positions(M, cs: ClusterState, assign, offsets, arena_size):
    """get_jammer_initial_positions (dbscan_clustering.py:365-406); `offsets[j]`
    is the np.random.uniform(-spread, spread, size=2) draw of jammer j."""
    pos = np.zeros((M, 2))
    ids = cs.cent_ids()
    for j in range(M):
        c = int(assign[j])
        if c == ASSIGN_NONE or c not in ids:
            continue
        pos[j] = np.clip(cs.cent_xy[ids.index(c)] + offsets[j], 0, arena_size)
    return pos


# --------------------------------------------------------------------------
# environment/action_space.py, enemy_swarm.py
# --------------------------------------------------------------------------
def apply_action(positions, velocity, dt, v_max, arena_size):
    """ActionHandler.apply_action / clip_velocity (action_space.py:190-240).
    `velocity` keeps its dtype: float32 actions are clipped and multiplied by
    dt in float32, then added to the float64 positions."""
    v = np.clip(velocity, -v_max, v_max)
    return np.clip(positions + v * dt, 0, arena_size)


def clip_bands(bands, num_bands=NUM_BANDS):
    """ActionHandler.clip_bands (action_space.py:202-212)."""
    return np.clip(np.asarray(bands).astype(np.int32), 0, num_bands - 1)


def enemy_step_random_walk(pos, vel, noise, v_enemy, dt, arena_size):
    """EnemySwarm._step_random_walk (enemy_swarm.py:157-189); `noise` is the
    (N, 2) normal(0, 0.1) draw."""
    vel = vel + noise
    speeds = np.sqrt(np.sum(vel * vel, axis=1, keepdims=True))
    speeds = np.maximum(speeds, 1e-6)
    vel = vel / speeds * v_enemy
    new = pos + vel * dt
    for dim in range(2):
        below = new[:, dim] < 0
        new[below, dim] = -new[below, dim]
        vel[below, dim] = -vel[below, dim]
        above = new[:, dim] > arena_size
        new[above, dim] = 2 * arena_size - new[above, dim]
        vel[above, dim] = -vel[above, dim]
    return np.clip(new, 0, arena_size), vel


def enemy_step_coordinated_turn(pos, vel, omega, noise_omega, dt, arena_size, omega_max):
    """EnemySwarm._step_coordinated_turn (enemy_swarm.py:191-242): per-drone turn
    with rate omega, then omega += noise_omega (the N scalar normal(0, 0.05)
    draws, in drone order), clipped; positions wrap modulo the arena.  The
    scalar loop is kept: np.sin / np.cos of a scalar and of an array element may
    round differently."""
    N = pos.shape[0]
    new_pos, new_vel, new_omega = np.empty_like(pos), np.empty_like(vel), np.empty(N)
    for i in range(N):
        x, y = pos[i, 0], pos[i, 1]
        x_dot, y_dot = vel[i, 0], vel[i, 1]
        w = omega[i]
        if abs(w) < 1e-6:
            nx, ny, nxd, nyd = x + x_dot * dt, y + y_dot * dt, x_dot, y_dot
        else:
            sin_wt, cos_wt = np.sin(w * dt), np.cos(w * dt)
            nx = x + (sin_wt / w) * x_dot - ((1 - cos_wt) / w) * y_dot
            nxd = cos_wt * x_dot - sin_wt * y_dot
            ny = y + ((1 - cos_wt) / w) * x_dot + (sin_wt / w) * y_dot
            nyd = sin_wt * x_dot + cos_wt * y_dot
        new_pos[i] = (nx, ny)
        new_vel[i] = (nxd, nyd)
        new_omega[i] = np.clip(w + noise_omega[i], -omega_max, omega_max)
    new_pos = np.mod(new_pos, arena_size)
    return new_pos, new_vel, new_omega


# --------------------------------------------------------------------------
# environment/reward.py, observation.py
# --------------------------------------------------------------------------
def _nearest_slot(p, cs: ClusterState):
    best, best_d = -1, float("inf")
    for s in range(cs.ncent):
        diff = p - cs.cent_xy[s]
        d = math.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
        if d < best_d:
            best_d, best = d, s
    return best, best_d


def reward5(p: EnvParams, lambda2_current, lambda2_initial, jammer_bands, enemy_band,
            jammer_positions, cs: ClusterState, velocities, assign, R_jam):
    """RewardCalculator.compute and helpers (reward.py:106-306).
    Returns (total, (r1, r2, r3, r4, r5))."""
    M = jammer_positions.shape[0]
    if M == 0:
        return 0.0, (0.0, 0.0, 0.0, 0.0, 0.0)
    # :178-199
    if lambda2_initial <= 0:
        r1 = 0.0
    else:
        r1 = float(np.clip(1.0 - (max(0, lambda2_current) / lambda2_initial), 0.0, 1.0))
    # :201-216
    r2 = float(np.sum(np.asarray(jammer_bands) == enemy_band)) / M
    # :218-260
    ids = cs.cent_ids()
    tot = 0.0
    if cs.ncent > 0:
        for j in range(M):
            c = int(assign[j])
            if c != ASSIGN_NONE and c in ids:
                target = cs.cent_xy[ids.index(c)]
            else:
                s, _ = _nearest_slot(jammer_positions[j], cs)
                target = cs.cent_xy[s]
            diff = jammer_positions[j] - target
            dist = math.sqrt(diff[0] * diff[0] + diff[1] * diff[1])
            tot += np.exp(-p.kappa * dist)
    r3 = tot / M
    # :262-280  (float32 when the action is float32)
    speed_sq = np.sum(velocities ** 2, axis=1)
    r4 = float(np.mean(speed_sq / (p.v_max ** 2)))
    # :282-306
    r5 = overlap_fraction(jammer_positions, R_jam)
    w = p.omega
    raw = w[0] * r1 + w[1] * r2 + w[2] * r3 - w[3] * r4 - w[4] * r5
    return float(np.clip(raw, -10.0, 10.0)), (r1, r2, float(r3), r4, r5)


def reward_per_agent(p: EnvParams, lambda2_current, lambda2_initial, jammer_bands, enemy_band,
                     jammer_positions, cs: ClusterState, velocities, assign, R_jam):
    """RewardCalculator.compute_per_agent (reward.py:308-360) with _compute_single_proximity
    (:362-396): the lambda2 and overlap terms are shared, band match, proximity and energy are the
    agent's own; float32 result, no clipping."""
    M = jammer_positions.shape[0]
    if M == 0:
        return np.array([])
    if lambda2_initial <= 0:
        r1 = 0.0
    else:
        r1 = float(np.clip(1.0 - (max(0, lambda2_current) / lambda2_initial), 0.0, 1.0))
    r5 = overlap_fraction(jammer_positions, R_jam)
    ids = cs.cent_ids()
    w = p.omega
    out = np.zeros(M, dtype=np.float32)
    for j in range(M):
        rb = 1.0 if jammer_bands[j] == enemy_band else 0.0
        rp = 0.0
        if cs.ncent > 0:
            c = int(assign[j])
            if c != ASSIGN_NONE and c in ids:
                target = cs.cent_xy[ids.index(c)]
            else:
                s_, _ = _nearest_slot(jammer_positions[j], cs)
                target = cs.cent_xy[s_]
            diff = jammer_positions[j] - target
            rp = np.exp(-p.kappa * math.sqrt(diff[0] * diff[0] + diff[1] * diff[1]))
        re = np.sum(velocities[j] ** 2) / (p.v_max ** 2)
        out[j] = w[0] * r1 + w[1] * rb + w[2] * rp - w[3] * re - w[4] * r5
    return out


def overlap_fraction(jammer_positions, R_jam) -> float:
    """reward.py:282-306 == observation.py:238-262."""
    M = jammer_positions.shape[0]
    if M <= 1:
        return 0.0
    thr = 2 * R_jam
    cnt = 0
    for i in range(M):
        for j in range(i + 1, M):
            diff = jammer_positions[i] - jammer_positions[j]
            if math.sqrt(diff[0] * diff[0] + diff[1] * diff[1]) < thr:
                cnt += 1
    return cnt / (M * (M - 1) / 2)


def build_observation(p: EnvParams, jammer_positions, jammer_bands, cs: ClusterState,
                      enemy_band, R_jam):
    """ObservationBuilder.build_vectorized (observation.py:264-332) -> (M, 5) float32."""
    M = jammer_positions.shape[0]
    obs = np.zeros((M, 5), dtype=np.float32)
    if M == 0:
        return obs
    A = p.arena_size
    if cs.ncent > 0:
        cent = cs.cent_xy[:cs.ncent]
        sizes = cs.cent_size[:cs.ncent]
    else:
        cent = np.array([[A / 2, A / 2]])
        sizes = np.zeros(1, dtype=np.int64)
    d = np.linalg.norm(jammer_positions[:, None, :] - cent[None, :, :], axis=2)
    obs[:, 0] = np.clip(np.min(d, axis=1) / A, 0, 1.0)
    if p.N > 0 and cs.ncent > 0:
        near = np.argmin(d, axis=1)
        for j in range(M):
            obs[j, 1] = int(sizes[near[j]]) / p.N
    if M > 1:
        pw = np.linalg.norm(jammer_positions[:, None, :] - jammer_positions[None, :, :], axis=2)
        np.fill_diagonal(pw, 0)
        obs[:, 2] = np.clip(pw.sum(axis=1) / (M - 1) / A, 0, 1.0)
    else:
        obs[:, 2] = 0.5
    obs[:, 3] = overlap_fraction(jammer_positions, R_jam)
    obs[:, 4] = (np.asarray(jammer_bands) == enemy_band).astype(np.float32)
    return obs


# --------------------------------------------------------------------------
# environment/jammer_env.py : state, reset, step
# --------------------------------------------------------------------------
@dataclass
class EnvState:
    """JammerEnv episode state (jammer_env.py:250-259) + EnemySwarm positions /
    velocities (enemy_swarm.py:96-110) + clusterer state."""
    step_count: int = 0
    enemy_band: int = 2
    lambda2_initial: float = 0.0
    R_jam: float = 0.0
    jammer_positions: np.ndarray = None
    jammer_bands: np.ndarray = None
    last_velocities: np.ndarray = None
    enemy_pos: np.ndarray = None
    enemy_vel: np.ndarray = None
    enemy_omega: np.ndarray = None      # coordinated-turn rate per drone (enemy_swarm.py:133)
    clusters: ClusterState = None
    assign: np.ndarray = None

    @staticmethod
    def empty(p: EnvParams) -> "EnvState":
        return EnvState(jammer_positions=np.zeros((p.M, 2)),
                        jammer_bands=np.zeros(p.M, dtype=np.int32),
                        last_velocities=np.zeros((p.M, 2)),
                        enemy_pos=np.zeros((p.N, 2)), enemy_vel=np.zeros((p.N, 2)),
                        enemy_omega=np.zeros(p.N),
                        clusters=ClusterState.empty(p.N),
                        assign=np.full(p.M, ASSIGN_NONE, dtype=np.int64))


def compute_lambda2(p: EnvParams, st: EnvState, disrupted: bool):
    """JammerEnv._compute_lambda2 (jammer_env.py:464-508).
    Returns (lambda2, adjacency_bool, jammed_bool)."""
    f = FREQUENCY_BANDS[int(st.enemy_band)]
    adj = adjacency_matrix(st.enemy_pos, db_to_watts(p.P_tx_dbm), db_to_watts(p.P_sens_dbm), f)
    jam = np.zeros_like(adj)
    if disrupted:
        jam = disrupted_links(st.jammer_positions, st.jammer_bands, st.enemy_pos,
                              st.enemy_band, db_to_watts(p.P_jammer_dbm),
                              db_to_watts(p.P_jam_thresh_dbm))
        adj = adj & ~jam                                   # jamming.py:309-332
    return lambda2_of_laplacian(laplacian(adj)), adj, jam


def env_reset(p: EnvParams, st: EnvState, enemy_pos, enemy_vel, enemy_band,
              jammer_init, jammer_bands, enemy_omega=None):
    """Deterministic part of JammerEnv.reset (jammer_env.py:261-354) for ONE
    attempt of its retry loop, given the random draws:

      enemy_pos, enemy_vel   EnemySwarm.reset draws (enemy_swarm.py:244-283)
      enemy_band             _rng.randint(0, 4)                        (:287)
      jammer_init            random_jammer_start: the (M,2) uniform positions
                             (:324-326); else the (M,2) uniform(-10,10) offsets
                             of get_jammer_initial_positions
      jammer_bands           _rng.randint(0, 4, size=M)                (:339)
      enemy_omega            coordinated turn only: the uniform(-omega_max/2,
                             omega_max/2, N) draw (enemy_swarm.py:133-137)

    The caller repeats with fresh draws while lambda2_initial <
    min_lambda2_initial (:304-308).  Returns (obs, info)."""
    st.step_count = 0
    st.enemy_pos = np.array(enemy_pos, dtype=np.float64)
    st.enemy_vel = np.array(enemy_vel, dtype=np.float64)
    st.enemy_omega = (np.zeros(p.N) if enemy_omega is None else np.array(enemy_omega, dtype=np.float64))
    st.enemy_band = int(enemy_band)
    f = FREQUENCY_BANDS[st.enemy_band]
    st.R_jam = float(compute_jam_range(db_to_watts(p.P_jammer_dbm),
                                       db_to_watts(p.P_jam_thresh_dbm), f))
    cluster_fit(st.clusters, st.enemy_pos, p.eps, p.min_samples, p.arena_size)
    st.lambda2_initial, _, _ = compute_lambda2(p, st, disrupted=False)
    if p.random_jammer_start:
        st.jammer_positions = np.array(jammer_init, dtype=np.float64)
        st.assign = np.full(p.M, ASSIGN_NONE, dtype=np.int64)
    else:
        st.assign = assign_proportional(p.M, st.clusters)
        st.jammer_positions = jammer_initial_positions(p.M, st.clusters, st.assign,
                                                       np.asarray(jammer_init), p.arena_size)
    st.jammer_bands = np.asarray(jammer_bands).astype(np.int32)
    st.last_velocities = np.zeros((p.M, 2))
    obs = build_observation(p, st.jammer_positions, st.jammer_bands, st.clusters,
                            st.enemy_band, st.R_jam)
    info = dict(lambda2_initial=st.lambda2_initial, lambda2_current=st.lambda2_initial,
                enemy_band=st.enemy_band, n_clusters=st.clusters.n_clusters, step=0)
    return obs, info


def env_reset_with_retries(p: EnvParams, st: EnvState, attempts, jammer_init, jammer_bands):
    """The retry loop of JammerEnv.reset (jammer_env.py:283-316) given the draws of each attempt.

    `attempts` yields (enemy_pos, enemy_vel, enemy_band[, enemy_omega]) in the order the reference
    draws them (attempt k > 0 comes from the streams reseeded with seed + k, or from the running
    streams when reset() was called without a seed).  Every attempt re-clusters (:298), so the
    "previous centroids" a later drift test sees are those of the attempt before the accepted one.
    The loop ends at the first attempt with lambda2_initial >= min_lambda2_initial, or keeps the last
    one after max_reset_attempts (:312-316).  The jammer draws are consumed once, after the loop
    (:321-339).  Returns (obs, info, attempts_used)."""
    used = 0
    for k, a in enumerate(attempts):
        if k >= p.max_reset_attempts:
            break
        used = k + 1
        st.step_count = 0
        st.enemy_pos = np.array(a[0], dtype=np.float64)
        st.enemy_vel = np.array(a[1], dtype=np.float64)
        st.enemy_band = int(a[2])
        st.enemy_omega = np.zeros(p.N) if len(a) < 4 or a[3] is None else np.array(a[3], dtype=np.float64)
        st.R_jam = float(compute_jam_range(db_to_watts(p.P_jammer_dbm), db_to_watts(p.P_jam_thresh_dbm),
                                           FREQUENCY_BANDS[st.enemy_band]))
        cluster_fit(st.clusters, st.enemy_pos, p.eps, p.min_samples, p.arena_size)
        st.lambda2_initial, _, _ = compute_lambda2(p, st, disrupted=False)
        if st.lambda2_initial >= p.min_lambda2_initial:
            break
    if p.random_jammer_start:
        st.jammer_positions = np.array(jammer_init, dtype=np.float64)
        st.assign = np.full(p.M, ASSIGN_NONE, dtype=np.int64)
    else:
        st.assign = assign_proportional(p.M, st.clusters)
        st.jammer_positions = jammer_initial_positions(p.M, st.clusters, st.assign,
                                                       np.asarray(jammer_init), p.arena_size)
    st.jammer_bands = np.asarray(jammer_bands).astype(np.int32)
    st.last_velocities = np.zeros((p.M, 2))
    obs = build_observation(p, st.jammer_positions, st.jammer_bands, st.clusters, st.enemy_band, st.R_jam)
    info = dict(lambda2_initial=st.lambda2_initial, lambda2_current=st.lambda2_initial,
                enemy_band=st.enemy_band, n_clusters=st.clusters.n_clusters, step=0)
    return obs, info, used


def env_step(p: EnvParams, st: EnvState, action, noise, want_avg_power: bool = True, recluster: bool = True):
    """JammerEnv.step (jammer_env.py:356-457) with the (N,2) enemy noise draw
    passed in.  `action` is the (M,3) array form (:376-378).  `recluster=False` leaves the
    K_recompute block (jammer_env.py:394-402) out: the cluster table of the last reset stays."""
    st.step_count += 1
    action = np.asarray(action)
    velocity = action[:, :2]
    bands = action[:, 2].astype(np.int32)
    st.last_velocities = velocity.copy()
    st.jammer_positions = apply_action(st.jammer_positions, velocity, p.dt, p.v_max, p.arena_size)
    st.jammer_bands = clip_bands(bands)
    noise = np.asarray(noise, dtype=np.float64)
    if p.enemy_mode == "coordinated_turn":       # noise[:, 0] = the N turn-rate draws
        st.enemy_pos, st.enemy_vel, st.enemy_omega = enemy_step_coordinated_turn(
            st.enemy_pos, st.enemy_vel, st.enemy_omega, noise[:, 0], p.dt, p.arena_size, p.omega_max)
    else:
        st.enemy_pos, st.enemy_vel = enemy_step_random_walk(
            st.enemy_pos, st.enemy_vel, noise, p.v_enemy, p.dt, p.arena_size)
    reassigned = False
    if recluster and st.step_count % p.K_recompute == 0:
        cluster_fit(st.clusters, st.enemy_pos, p.eps, p.min_samples, p.arena_size)
        if should_reassign(st.clusters, p.drift_threshold):
            st.assign = assign_proportional(p.M, st.clusters)
            reassigned = True
    lam, adj, jam = compute_lambda2(p, st, disrupted=True)
    raw_reduction = 1.0 - (lam / st.lambda2_initial) if st.lambda2_initial > 1e-6 else 0.0
    reward, comps = reward5(p, lam, st.lambda2_initial, st.jammer_bands, st.enemy_band,
                            st.jammer_positions, st.clusters, st.last_velocities,
                            st.assign, st.R_jam)
    truncated = st.step_count >= p.max_steps
    obs = build_observation(p, st.jammer_positions, st.jammer_bands, st.clusters,
                            st.enemy_band, st.R_jam)
    w = p.omega
    # RewardCalculator.get_reward_breakdown (reward.py:398-414): weighted, penalties negative
    breakdown = (w[0] * comps[0], w[1] * comps[1], w[2] * comps[2], -w[3] * comps[3],
                 -w[4] * comps[4])
    info = dict(lambda2_initial=st.lambda2_initial, lambda2_current=lam,
                lambda2_reduction=raw_reduction, enemy_band=st.enemy_band,
                n_clusters=st.clusters.n_clusters, step=st.step_count,
                reward_components=breakdown, reward_terms=comps,
                adjacency=adj, jammed=jam, reassigned=reassigned)
    if want_avg_power:
        info["avg_jamming_power_dbm"] = avg_jamming_power_dbm(
            st.jammer_positions, st.jammer_bands, st.enemy_pos, st.enemy_band,
            db_to_watts(p.P_jammer_dbm))
    return obs, reward, False, truncated, info


# --------------------------------------------------------------------------
# agents/rollout_buffer.py : GAE
# --------------------------------------------------------------------------
def gae(rewards, values, dones, last_value, last_done, gamma, lam):
    """RolloutBuffer.compute_returns_and_advantages (rollout_buffer.py:151-204)
    before the normalisation at :207-209.  float32 arrays of length T.
    Returns (advantages[T], returns[T])."""
    T = len(rewards)
    v = np.zeros(T + 1, dtype=np.float32)
    v[:T] = values
    v[T] = last_value * (1 - float(last_done))
    d = np.zeros(T + 1, dtype=np.float32)
    d[:T] = dones
    d[T] = float(last_done)
    adv = np.zeros(T, dtype=np.float32)
    last = 0
    for t in reversed(range(T)):
        nnt = 1.0 - d[t + 1]
        delta = rewards[t] + gamma * v[t + 1] * nnt - v[t]
        adv[t] = last = delta + gamma * lam * nnt * last
    return adv, adv + np.asarray(values, dtype=np.float32)


# --------------------------------------------------------------------------
# bit packing helpers shared by tests (row i, bit j of word j//32)
# --------------------------------------------------------------------------
def pack_rows(mat_bool):
    N = mat_bool.shape[0]
    W = (N + 31) // 32
    out = np.zeros((N, W), dtype=np.uint32)
    for j in range(mat_bool.shape[1]):
        out[:, j // 32] |= (mat_bool[:, j].astype(np.uint32) << np.uint32(j % 32))
    return out


def unpack_rows(words, N):
    out = np.zeros((N, N), dtype=bool)
    for j in range(N):
        out[:, j] = (words[:, j // 32] >> np.uint32(j % 32)) & 1
    return out
