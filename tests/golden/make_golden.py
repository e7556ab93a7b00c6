#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What is executed is the reference's own, unmodified code, imported from
/root/reference/src:

  * stage vectors   physics.fspl / physics.jamming / physics.communication_graph /
                    clustering.dbscan_clustering and environment/{reward,
                    observation,enemy_swarm,action_space}.py, agents/rollout_buffer.py
                    (the last five loaded by file path so that the package
                    __init__ files, which pull in gymnasium / torch, are not run)
  * trajectories    environment.jammer_env.JammerEnv itself.  That module does
                    `import gymnasium`, which is not installed here and supplies no
                    arithmetic to the step (a base class and space descriptors
                    only), so an in-memory module object with `Env` and
                    `spaces.{Box,Dict,MultiDiscrete,Discrete,Tuple}` attribute
                    holders is registered under that name before the import.

Outputs are data only: inputs and expected outputs as .npz arrays.  Python
dicts of the reference are flattened to arrays in dict-iteration order.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from physics import fspl, jamming, communication_graph as cg          # noqa: E402
from clustering import dbscan_clustering as dbc                       # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


ref_reward = _load("ref_reward", "environment/reward.py")
ref_obs = _load("ref_observation", "environment/observation.py")
ref_swarm = _load("ref_enemy_swarm", "environment/enemy_swarm.py")
ref_action = _load("ref_action_space", "environment/action_space.py")
ref_buffer = _load("ref_rollout_buffer", "agents/rollout_buffer.py")


def _import_jammer_env():
    class _Holder:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k
            self.shape = k.get("shape")

    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:
        def __init__(self, *a, **k):
            pass

    for n in ("Box", "Dict", "MultiDiscrete", "Discrete", "Tuple"):
        setattr(spaces, n, type(n, (_Holder,), {}))
    gym.Env = Env
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces
    from environment.jammer_env import JammerEnv
    return JammerEnv


JammerEnv = _import_jammer_env()

FAST = dict(arena_size=200.0, max_steps=200, dt=0.5, v_max=5.0, v_enemy=2.0,
            P_tx_dbm=20.0, P_sens_dbm=-90.0, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0,
            eps=25.0, min_samples=2, K_recompute=10, random_jammer_start=True)
SIZES = [(5, 2), (10, 4), (50, 8), (200, 16)]


def pack_rows(mat):
    mat = np.asarray(mat).astype(bool)
    N = mat.shape[0]
    W = (N + 31) // 32
    out = np.zeros((N, W), dtype=np.uint32)
    for j in range(mat.shape[1]):
        out[:, j // 32] |= mat[:, j].astype(np.uint32) << np.uint32(j % 32)
    return out


def flatten_clusters(N, centroids, sizes):
    """dict -> (ids[KS], xy[KS,2], size[KS], ncent) in dict-iteration order."""
    KS = N + 1
    ids = np.full(KS, -9, dtype=np.int32)
    xy = np.zeros((KS, 2))
    sz = np.zeros(KS, dtype=np.int32)
    for s, (cid, c) in enumerate(centroids.items()):
        ids[s] = int(cid)
        xy[s] = c
        sz[s] = int(sizes.get(cid, 0))
    return ids, xy, sz, len(centroids)


def flatten_assign(M, assignments):
    a = np.full(M, -2, dtype=np.int32)
    for cid, lst in assignments.items():
        for j in lst:
            a[j] = int(cid)
    return a


# --------------------------------------------------------------------------
# 1. stage vectors
# --------------------------------------------------------------------------
def stage_case(rng, N, M, band, arena, P_tx, P_sens, P_jam, P_thr, eps, min_samples,
               tight=False):
    tx, sens = fspl.db_to_watts(P_tx), fspl.db_to_watts(P_sens)
    pj, pt = fspl.db_to_watts(P_jam), fspl.db_to_watts(P_thr)
    f = fspl.FREQUENCY_BANDS[band]
    if tight:      # a few blobs so that DBSCAN finds clusters and jammers sit on links
        k = max(1, N // 8)
        ctr = rng.uniform(0.15 * arena, 0.85 * arena, size=(k, 2))
        epos = ctr[rng.randint(0, k, size=N)] + rng.normal(0, 0.06 * arena, size=(N, 2))
        epos = np.clip(epos, 0, arena)
    else:
        epos = rng.uniform(0.1 * arena, 0.9 * arena, size=(N, 2))
    jpos = rng.uniform(0.1 * arena, 0.9 * arena, size=(M, 2))
    if tight:      # put half of the jammers right on link midpoints
        for k_ in range(0, M, 2):
            i, j = rng.randint(0, N, size=2)
            jpos[k_] = (epos[i] + epos[j]) / 2 + rng.normal(0, 1.0, size=2)
    jb = rng.randint(0, 4, size=M).astype(np.int32)
    if tight:
        jb[rng.rand(M) < 0.6] = band
    vel = rng.uniform(-6, 6, size=(M, 2)).astype(np.float32)

    A = cg.compute_adjacency_matrix(epos, tx, sens, f)
    J = jamming.compute_disrupted_links(jpos, jb, epos, band, pj, pt)
    Aj = jamming.apply_jamming_to_adjacency(A, J)
    L = cg.compute_laplacian(Aj)
    lam = cg.compute_lambda2(L)
    lam0 = cg.compute_lambda2(cg.compute_laplacian(A))
    deg = np.sum(Aj, axis=1).astype(np.int32)

    cl = dbc.DBSCANClusterer(eps=eps, min_samples=min_samples, arena_size=arena)
    labels, cents = cl.fit(epos)
    sizes = cl.get_cluster_sizes()
    ids, cxy, csz, ncent = flatten_clusters(N, cents, sizes)
    assigns = dbc.assign_jammers_to_clusters(M, cents, sizes, strategy="proportional")
    assign = flatten_assign(M, assigns)
    # half the cases use "no assignment" as after a random-start reset
    use_assign = bool(rng.rand() < 0.5)
    R_jam = fspl.compute_jam_range(pj, pt, f)
    rc = ref_reward.RewardCalculator(v_max=5.0, R_jam=R_jam, arena_size=arena)
    total, comp = rc.compute(lam, lam0, jb, band, jpos, cents, vel,
                             assigns if use_assign else {})
    ob = ref_obs.ObservationBuilder(arena_size=arena, R_jam=R_jam)
    obs = ob.build_vectorized(jpos, jb, cents, sizes, band, N, assigns)

    # _compute_avg_jamming_power, restated from the same reference functions
    mid = jamming.compute_midpoints(epos)
    dist = jamming.compute_distances_to_midpoints(jpos, mid)
    bm = jb == band
    if not np.any(bm):
        avgp = -80.0
    else:
        P = jamming.compute_jamming_power(dist, pj, f)
        P[~bm] = 0.0
        Pm = np.max(P, axis=0)
        lp = Pm[np.triu_indices_from(Pm, k=1)]
        v = lp[lp > 0]
        avgp = float(fspl.watts_to_db(np.mean(v))) if len(v) else -80.0

    return dict(epos=epos, jpos=jpos, jb=jb, vel=vel, band=np.int32(band),
                adj=pack_rows(A), jam=pack_rows(J), deg=deg, lam=np.float64(lam),
                lam0=np.float64(lam0), labels=labels.astype(np.int32), cent_ids=ids,
                cent_xy=cxy, cent_size=csz, ncent=np.int32(ncent),
                n_clusters=np.int32(cl.n_clusters), assign=assign,
                use_assign=np.int32(use_assign), R_jam=np.float64(R_jam),
                reward=np.float64(total),
                comps=np.array([comp.lambda2_reduction, comp.band_match, comp.proximity,
                                comp.energy_penalty, comp.overlap_penalty], dtype=np.float64),
                obs=obs, avgp=np.float64(avgp))


def make_stage():
    rng = np.random.RandomState(20240611)
    out = {}
    meta = []
    idx = 0
    for (N, M) in SIZES:
        ncase = 32 if N <= 50 else 8
        for c in range(ncase):
            band = c % 4
            kind = c % 3
            if kind == 0:      # fast-preset physics, uniform layout
                cfg = dict(arena=200.0, P_tx=20.0, P_sens=-90.0, P_jam=20.0, P_thr=-40.0,
                           eps=25.0, min_samples=2, tight=False)
            elif kind == 1:    # fast-preset physics, blobs + jammers on links, min_samples 3
                cfg = dict(arena=200.0, P_tx=20.0, P_sens=-90.0, P_jam=20.0, P_thr=-40.0,
                           eps=15.0, min_samples=2 if c % 2 else 3, tight=True)
            else:              # large arena: FSPL adjacency itself is non-trivial
                cfg = dict(arena=6000.0, P_tx=20.0, P_sens=-90.0, P_jam=30.0, P_thr=-45.0,
                           eps=400.0, min_samples=2, tight=(c % 2 == 0))
            d = stage_case(rng, N, M, band, **cfg)
            for k, v in d.items():
                out[f"c{idx}_{k}"] = v
            meta.append([N, M, cfg["arena"], cfg["P_tx"], cfg["P_sens"], cfg["P_jam"],
                         cfg["P_thr"], cfg["eps"], cfg["min_samples"]])
            idx += 1
    out["meta"] = np.array(meta, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "stage.npz"), **out)
    print("stage cases:", idx)


# --------------------------------------------------------------------------
# 1b. lambda2 through the reference's sparse branch (communication_graph.py:268-278), N > 256
# --------------------------------------------------------------------------
def make_lambda2_sparse():
    """compute_lambda2(L, use_sparse=True) (scipy eigsh, which='SM') on graphs of 300, 640 and 1024 vertices:
    connected sparse, dense and disconnected random graphs per size; the dense branch's value beside it."""
    out = {}
    k = 0
    for n in (300, 640, 1024):
        rng = np.random.RandomState(7000 + n)
        for dens in (2.5 * np.log(n) / n, 0.3, 0.5 * np.log(n) / n):
            upper = np.triu(rng.rand(n, n) < dens, 1)
            A = (upper | upper.T).astype(np.float64)
            L = cg.compute_laplacian(A)
            out[f"g{k}_n"] = np.int32(n)
            out[f"g{k}_adj"] = pack_rows(A)
            out[f"g{k}_lam_sparse"] = np.float64(cg.compute_lambda2(L, use_sparse=True))
            out[f"g{k}_lam_dense"] = np.float64(cg.compute_lambda2(L))
            k += 1
    out["n_graphs"] = np.int32(k)
    np.savez_compressed(os.path.join(OUT, "lambda2_sparse.npz"), **out)
    print("sparse-branch lambda2 graphs:", k)


# --------------------------------------------------------------------------
# 2. trajectories of the real JammerEnv
# --------------------------------------------------------------------------
def record_reset(env, seed, gstate):
    """State right after env.reset(); the global-RNG offsets are re-drawn from
    the snapshot `gstate` taken before the call."""
    g = np.random.RandomState()
    g.set_state(gstate)
    offsets = np.array([g.uniform(-10.0, 10.0, size=2) for _ in range(env.M)])
    ids, cxy, csz, ncent = flatten_clusters(env.N, env.centroids, env.cluster_sizes)
    extra = {}
    if env.enemy_swarm.mode == "coordinated_turn":
        extra["eomega"] = env.enemy_swarm.state[:, 4].copy()
    return dict(epos=env.enemy_swarm.positions.copy(), evel=env.enemy_swarm.velocities.copy(), **extra,
                band=np.int32(env.enemy_band), jpos=env.jammer_positions.copy(),
                jb=env.jammer_bands.copy(), offsets=offsets, lam0=np.float64(env.lambda2_initial),
                R_jam=np.float64(env.R_jam), labels=env.clusterer.labels.astype(np.int32),
                cent_ids=ids, cent_xy=cxy, cent_size=csz, ncent=np.int32(ncent),
                n_clusters=np.int32(env.clusterer.n_clusters),
                assign=flatten_assign(env.M, env.jammer_assignments))


def make_trajectory(name, N, M, seed, n_steps, policy, **kw):
    np.random.seed(seed + 1000)
    env = JammerEnv(M=M, N=N, seed=seed, **kw)
    arng = np.random.RandomState(seed + 7)
    resets = []
    gs = np.random.get_state()
    obs0, info0 = env.reset(seed=seed)
    r = record_reset(env, seed, gs)
    r["obs"] = obs0
    r["at_step"] = np.int32(0)
    resets.append(r)
    rec = {k: [] for k in ("action", "noise", "obs", "reward", "lam", "trunc", "n_clusters",
                           "labels", "epos", "evel", "jpos", "jb", "comps", "avgp", "adj", "jam",
                           "assign", "cent_ids", "cent_xy", "cent_size", "ncent", "step", "eomega")}
    ct = env.enemy_swarm.mode == "coordinated_turn"
    tx, sens = fspl.db_to_watts(env.P_tx_dbm), fspl.db_to_watts(env.P_sens_dbm)
    pj, pt = fspl.db_to_watts(env.P_jammer_dbm), fspl.db_to_watts(env.P_jam_thresh_dbm)
    for t in range(n_steps):
        a = np.zeros((M, 3), dtype=np.float32)
        if policy == "random":
            a[:, :2] = arng.uniform(-1.2 * env.v_max, 1.2 * env.v_max, size=(M, 2))
            a[:, 2] = arng.uniform(-1.0, 5.0, size=M)
        else:  # "chase": head for random link midpoints on the enemy band (lambda2 drops)
            ep = env.enemy_swarm.positions
            for j in range(M):
                i1, i2 = arng.randint(0, N, size=2)
                tgt = (ep[i1] + ep[i2]) / 2
                a[j, :2] = np.clip((tgt - env.jammer_positions[j]) / env.dt, -7, 7)
            a[:, 2] = env.enemy_band + arng.uniform(0, 0.9, size=M)
            if arng.rand() < 0.3:
                a[arng.randint(0, M), 2] = arng.uniform(-1, 5)
        st = env.enemy_swarm._rng.get_state()
        rr = np.random.RandomState()
        rr.set_state(st)
        if ct:      # N scalar draws of normal(0, std/2), one per drone (enemy_swarm.py:228); second column unused
            noise = np.zeros((N, 2))
            noise[:, 0] = [rr.normal(0, env.enemy_swarm.process_noise_std * 0.5) for _ in range(N)]
        else:
            noise = rr.normal(0, env.enemy_swarm.process_noise_std, size=(N, 2))
        obs, reward, term, trunc, info = env.step(a)
        rec["eomega"].append(env.enemy_swarm.state[:, 4].copy() if ct else np.zeros(N))
        A = cg.compute_adjacency_matrix(env.enemy_swarm.positions, tx, sens,
                                        env.BANDS[env.enemy_band])
        J = jamming.compute_disrupted_links(env.jammer_positions, env.jammer_bands,
                                            env.enemy_swarm.positions, env.enemy_band, pj, pt)
        ids, cxy, csz, ncent = flatten_clusters(N, env.centroids, env.cluster_sizes)
        c = info["reward_components"]
        rec["action"].append(a); rec["noise"].append(noise); rec["obs"].append(obs)
        rec["reward"].append(float(reward)); rec["lam"].append(info["lambda2_current"])
        rec["trunc"].append(bool(trunc)); rec["n_clusters"].append(info["n_clusters"])
        rec["labels"].append(env.clusterer.labels.astype(np.int16))
        rec["epos"].append(env.enemy_swarm.positions.copy())
        rec["evel"].append(env.enemy_swarm.velocities.copy())
        rec["jpos"].append(env.jammer_positions.copy()); rec["jb"].append(env.jammer_bands.copy())
        rec["comps"].append([c["lambda2_reduction"], c["band_match"], c["proximity"],
                             c["energy_penalty"], c["overlap_penalty"]])
        rec["avgp"].append(info["avg_jamming_power_dbm"])
        rec["adj"].append(pack_rows(A)); rec["jam"].append(pack_rows(J))
        rec["assign"].append(flatten_assign(M, env.jammer_assignments))
        rec["cent_ids"].append(ids); rec["cent_xy"].append(cxy); rec["cent_size"].append(csz)
        rec["ncent"].append(ncent); rec["step"].append(info["step"])
        assert not term
        if trunc:
            gs = np.random.get_state()
            o, i_ = env.reset()
            r = record_reset(env, None, gs)
            r["obs"] = o
            r["at_step"] = np.int32(t + 1)
            resets.append(r)
    out = {}
    for k, v in rec.items():
        out[k] = np.array(v)
    for i, r in enumerate(resets):
        for k, v in r.items():
            out[f"reset{i}_{k}"] = v
    out["n_resets"] = np.int32(len(resets))
    cfg = dict(M=M, N=N, arena_size=env.arena_size, max_steps=env.max_steps, dt=env.dt,
               v_max=env.v_max, v_enemy=env.v_enemy, P_tx_dbm=env.P_tx_dbm,
               P_sens_dbm=env.P_sens_dbm, P_jammer_dbm=env.P_jammer_dbm,
               P_jam_thresh_dbm=env.P_jam_thresh_dbm, eps=env.eps, min_samples=env.min_samples,
               K_recompute=env.K_recompute, random_jammer_start=int(env.random_jammer_start),
               coordinated_turn=int(ct), omega_max=env.enemy_swarm.omega_max)
    for k, v in cfg.items():
        out[f"cfg_{k}"] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, f"traj_{name}.npz"), **out)
    lam = np.array(rec["lam"])
    print(f"traj {name}: steps={n_steps} resets={len(resets)} lam[min,max]=({lam.min():.3g},{lam.max():.3g})"
          f" reassigned_steps={int(np.sum(np.any(np.diff(np.array(rec['assign']), axis=0) != 0, axis=1)))}")


def make_trajectories():
    fast = dict(FAST)
    make_trajectory("n5m2_random", 5, 2, 11, 120, "random", **{**fast, "max_steps": 50})
    make_trajectory("n10m4_random", 10, 4, 42, 400, "random", **fast)
    make_trajectory("n10m4_chase", 10, 4, 43, 400, "chase", **fast)
    make_trajectory("n50m8_random", 50, 8, 7, 250, "random", **{**fast, "max_steps": 120})
    make_trajectory("n50m8_chase", 50, 8, 8, 250, "chase", **{**fast, "max_steps": 120})
    make_trajectory("n200m16_chase", 200, 16, 9, 50, "chase", **{**fast, "max_steps": 40})
    # centroid start (random_jammer_start=False) with the env's default physics and eps
    dflt = dict(arena_size=200.0, max_steps=60, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0,
                eps=30.0, random_jammer_start=False)
    make_trajectory("n20m4_centroid", 20, 4, 5, 150, "chase", **dflt)
    # large arena: FSPL adjacency non-trivial, faster swarm so that clusters drift
    big = dict(arena_size=6000.0, max_steps=80, v_max=60.0, v_enemy=40.0, P_jammer_dbm=30.0,
               P_jam_thresh_dbm=-45.0, eps=500.0, K_recompute=5, random_jammer_start=False)
    make_trajectory("n30m6_bigarena", 30, 6, 21, 200, "chase", **big)


def make_per_agent():
    """RewardCalculator.compute_per_agent (reward.py:308-360): not called by the env's step, but part of
    the reward module's public functions."""
    rng = np.random.RandomState(2024)
    out = {}
    idx = 0
    for N, M in ((10, 4), (50, 8), (30, 6), (120, 16)):
        for rep in range(10):
            arena, band = 200.0, int(rng.randint(0, 4))
            k = max(1, N // 8)
            ctr = rng.uniform(0.15 * arena, 0.85 * arena, size=(k, 2))
            epos = np.clip(ctr[rng.randint(0, k, size=N)] + rng.normal(0, 0.06 * arena, size=(N, 2)), 0, arena)
            jpos = rng.uniform(0.1 * arena, 0.9 * arena, size=(M, 2))
            if rep % 3 == 0:
                jpos[1:] = jpos[0] + rng.normal(0, 3.0, size=(M - 1, 2))      # overlapping jammers
            jb = rng.randint(0, 4, size=M).astype(np.int32)
            vel = rng.uniform(-6, 6, size=(M, 2)).astype(np.float32)
            lam0 = float(N)
            lam = float(rng.uniform(-0.1, 1.2) * N)
            if rep == 7:
                lam0 = 0.0
            cl = dbc.DBSCANClusterer(eps=25.0, min_samples=2, arena_size=arena)
            labels, cents = cl.fit(epos)
            sizes = cl.get_cluster_sizes()
            ids, cxy, csz, ncent = flatten_clusters(N, cents, sizes)
            assigns = dbc.assign_jammers_to_clusters(M, cents, sizes, strategy="proportional")
            mode = rep % 3                      # 0: assignments, 1: empty dict, 2: None
            R_jam = fspl.compute_jam_range(fspl.db_to_watts(20.0), fspl.db_to_watts(-40.0),
                                           fspl.FREQUENCY_BANDS[band])
            rc = ref_reward.RewardCalculator(v_max=5.0, R_jam=R_jam, arena_size=arena)
            r = rc.compute_per_agent(lam, lam0, jb, band, jpos, cents, vel,
                                     assigns if mode == 0 else ({} if mode == 1 else None))
            d = dict(N=np.int32(N), M=np.int32(M), band=np.int32(band), jpos=jpos, jb=jb, vel=vel,
                     lam=np.float64(lam), lam0=np.float64(lam0), cent_ids=ids, cent_xy=cxy, cent_size=csz,
                     ncent=np.int32(ncent), n_clusters=np.int32(cl.n_clusters),
                     assign=flatten_assign(M, assigns if mode == 0 else {}), R_jam=np.float64(R_jam),
                     rewards=np.asarray(r, dtype=np.float32))
            for k_, v in d.items():
                out[f"c{idx}_{k_}"] = v
            idx += 1
    out["n_cases"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "per_agent.npz"), **out)
    print("per-agent reward cases:", idx)


def make_ct_trajectories():
    """Second dynamics mode of the enemy swarm (enemy_swarm.py:191-242)."""
    fast = dict(FAST)
    make_trajectory("n10m4_ct", 10, 4, 44, 300, "chase", **{**fast, "max_steps": 120, "enemy_mode": "coordinated_turn"})
    make_trajectory("n50m8_ct", 50, 8, 12, 200, "random", **{**fast, "max_steps": 90, "enemy_mode": "coordinated_turn"})


# --------------------------------------------------------------------------
# 3. edge cases
# --------------------------------------------------------------------------
def make_edges():
    out = {}
    tx, sens = fspl.db_to_watts(20.0), fspl.db_to_watts(-90.0)
    pj, pt = fspl.db_to_watts(20.0), fspl.db_to_watts(-40.0)
    # (a) closed-form Laplacian spectra, computed by the reference's own function
    for n in (2, 3, 4, 5, 6, 10, 50):
        A = np.ones((n, n)) - np.eye(n)
        out[f"lam_K{n}"] = np.float64(cg.compute_lambda2(cg.compute_laplacian(A)))
    for n in (5, 10, 50):
        A = np.zeros((n, n))
        for i in range(n - 1):
            A[i, i + 1] = A[i + 1, i] = 1
        out[f"lam_P{n}"] = np.float64(cg.compute_lambda2(cg.compute_laplacian(A)))
        A[0, n - 1] = A[n - 1, 0] = 1
        out[f"lam_C{n}"] = np.float64(cg.compute_lambda2(cg.compute_laplacian(A)))
    A = np.zeros((6, 6))
    A[0, 1] = A[1, 0] = A[1, 2] = A[2, 1] = A[3, 4] = A[4, 3] = A[4, 5] = A[5, 4] = 1
    out["lam_two_paths"] = np.float64(cg.compute_lambda2(cg.compute_laplacian(A)))
    # (b) random graphs of every density: adjacency bits -> lambda2
    rng = np.random.RandomState(99)
    k = 0
    for n in (5, 10, 24, 50, 64, 200):
        for p_edge in (0.03, 0.1, 0.3, 0.6, 0.9, 0.98):
            A = (rng.rand(n, n) < p_edge)
            A = np.triu(A, 1)
            A = (A | A.T).astype(np.float64)
            out[f"g{k}_adj"] = pack_rows(A)
            out[f"g{k}_n"] = np.int32(n)
            out[f"g{k}_lam"] = np.float64(cg.compute_lambda2(cg.compute_laplacian(A)))
            k += 1
    out["n_graphs"] = np.int32(k)
    # (c) jammer exactly on a midpoint (distance clamps to 1e-10), wrong band, far away
    epos = np.array([[40.0, 50.0], [60.0, 50.0], [150.0, 150.0]])
    for name, jp, jb in (("on_mid", [[50.0, 50.0]], [2]), ("wrong_band", [[50.0, 50.0]], [1]),
                         ("far", [[50.0, 50.0 + 2000.0]], [2])):
        J = jamming.compute_disrupted_links(np.array(jp), np.array(jb), epos, 2, pj, pt)
        out[f"jam_{name}"] = pack_rows(J)
    out["jam_epos"] = epos
    # (d) threshold knife-edge: pairs placed within a few ulp of the comm / jam radius
    f = fspl.FREQUENCY_BANDS[3]
    R = fspl.compute_comm_range(tx, sens, f)
    xs = [np.nextafter(R, 0), R, np.nextafter(R, 1e9), R * (1 - 1e-15), R * (1 + 1e-15),
          R * (1 - 3e-16), R * (1 + 3e-16)]
    pos = np.zeros((len(xs) + 1, 2))
    pos[1:, 0] = xs
    A = cg.compute_adjacency_matrix(pos, tx, sens, f)
    out["knife_pos"] = pos
    out["knife_adj"] = pack_rows(A)
    Rj = fspl.compute_jam_range(pj, pt, f)
    ep2 = np.array([[0.0, 0.0], [2 * Rj, 0.0], [0.0, 2 * np.nextafter(Rj, 0)],
                    [2 * np.nextafter(Rj, 1e9), 2 * np.nextafter(Rj, 1e9)]])
    J = jamming.compute_disrupted_links(np.array([[0.0, 0.0]]), np.array([3]), ep2, 3, pj, pt)
    out["knife_jam_epos"] = ep2
    out["knife_jam"] = pack_rows(J)
    # (e) DBSCAN: all noise, one cluster, single chain, border points with min_samples 3/4
    cases = {
        "all_noise": (np.array([[0.0, 0], [100, 100], [200, 0], [0, 200]]), 10.0, 2),
        "one": (np.array([[0.0, 0], [1, 1], [2, 0], [1, -1]]), 5.0, 2),
        "two": (np.array([[0.0, 0], [5, 5], [100, 100], [105, 105], [50, 50]]), 10.0, 2),
        "chain": (np.column_stack([np.arange(12) * 9.0, np.zeros(12)]), 10.0, 2),
        "border3": (np.array([[0.0, 0], [1, 0], [2, 0], [3.5, 0], [10, 0], [11, 0], [12, 0],
                              [6.75, 0]]), 3.3, 3),
        "exact_eps": (np.array([[0.0, 0], [3, 4], [6, 8], [6, 8 + np.nextafter(5.0, 6)]]), 5.0, 2),
    }
    rng = np.random.RandomState(5)
    for i in range(6):
        cases[f"rand{i}"] = (rng.uniform(0, 100, size=(40, 2)), 9.0 + i, 2 + (i % 4))
    for name, (pts, eps, ms) in cases.items():
        cl = dbc.DBSCANClusterer(eps=eps, min_samples=ms, arena_size=200.0)
        labels, cents = cl.fit(pts)
        ids, cxy, csz, ncent = flatten_clusters(len(pts), cents, cl.get_cluster_sizes())
        out[f"db_{name}_pts"] = pts
        out[f"db_{name}_par"] = np.array([eps, ms], dtype=np.float64)
        out[f"db_{name}_labels"] = labels.astype(np.int32)
        out[f"db_{name}_ids"] = ids
        out[f"db_{name}_xy"] = cxy
        out[f"db_{name}_size"] = csz
        out[f"db_{name}_ncent"] = np.int32(ncent)
    out["db_names"] = np.array(sorted(cases.keys()))
    # (f) enemy reflection off both walls and the speed floor
    sw = ref_swarm.EnemySwarm(N=6, positions=np.array([[0.2, 100.0], [199.9, 100.0], [100.0, 0.1],
                                                       [100.0, 199.95], [0.3, 199.8], [50.0, 50.0]]),
                              v_enemy=2.0, dt=0.5, arena_size=200.0, seed=3)
    sw.velocities = np.array([[-2.0, 0.0], [2.0, 0.0], [0.0, -2.0], [0.0, 2.0], [-1.5, 1.4],
                              [0.0, 0.0]])
    st = sw._rng.get_state()
    rr = np.random.RandomState(); rr.set_state(st)
    noise = rr.normal(0, 0.1, size=(6, 2))
    noise[5] = 0.0          # zero velocity + zero noise -> speed floor 1e-6
    out["refl_pos0"] = sw.positions.copy()
    out["refl_vel0"] = sw.velocities.copy()
    out["refl_noise"] = noise
    sw._rng = types.SimpleNamespace(normal=lambda *a, **k: noise)
    sw.step()
    out["refl_pos1"] = sw.positions.copy()
    out["refl_vel1"] = sw.velocities.copy()
    # (g) action decode
    ah = ref_action.ActionHandler(v_max=5.0, num_bands=4, arena_size=200.0)
    p0 = np.array([[100.0, 100.0], [199.0, 1.0], [0.5, 0.5], [150.25, 20.125]])
    v = np.array([[10.0, 0.0], [5.0, -5.0], [-3.3, -0.7], [4.999, 7.0]], dtype=np.float32)
    out["act_pos0"] = p0
    out["act_vel"] = v
    out["act_pos1_dt05"] = ah.apply_action(p0, v, 0.5)
    out["act_pos1_dt01"] = ah.apply_action(p0, v, 0.1)
    out["act_bands_in"] = np.array([-1.0, 0.0, 3.0, 5.0, 2.9, 0.99, -0.5], dtype=np.float32)
    out["act_bands_out"] = ah.clip_bands(out["act_bands_in"].astype(np.int32))
    # (h) GAE from the reference buffer
    T, M = 64, 4
    rng = np.random.RandomState(17)
    buf = ref_buffer.RolloutBuffer(buffer_size=T, M=M, gamma=0.99, gae_lambda=0.95)
    rew = rng.normal(0, 1, size=T).astype(np.float32)
    val = rng.normal(0, 1, size=T).astype(np.float32)
    don = (rng.rand(T) < 0.1)
    for t in range(T):
        buf.add(np.zeros((M, 5), np.float32), np.zeros((M, 3), np.float32),
                np.zeros(M, np.float32), float(rew[t]), bool(don[t]), float(val[t]))
    # un-normalised advantages: re-run the loop body through a subclass-free path
    buf.compute_returns_and_advantages(0.37, False)
    out["gae_rew"] = rew; out["gae_val"] = val; out["gae_done"] = don.astype(np.float32)
    out["gae_last_value"] = np.float32(0.37)
    out["gae_adv_norm"] = buf.advantages[:, 0].copy()
    out["gae_ret"] = buf.returns[:, 0].copy()
    np.savez_compressed(os.path.join(OUT, "edges.npz"), **out)
    print("edge vectors:", len(out))


# --------------------------------------------------------------------------
# 4. reset retry loop (jammer_env.py:283-316): attempts that fail the lambda2_initial test
# --------------------------------------------------------------------------
def record_retry_case(N, M, seed, reset_seeds, **kw):
    """JammerEnv(seed).reset(seed=...) for every entry of `reset_seeds` (None = continue the streams),
    recording the draws of EVERY attempt (the wrapper around _update_clusters only reads attributes)
    and the state the reset ends in."""
    np.random.seed(seed + 1000)
    env = JammerEnv(M=M, N=N, seed=seed, **kw)
    out = []
    for rs in reset_seeds:
        attempts = []
        orig = env._update_clusters

        def spy():
            extra = {}
            if env.enemy_swarm.mode == "coordinated_turn":
                extra["eomega"] = env.enemy_swarm.state[:, 4].copy()
            attempts.append(dict(epos=env.enemy_swarm.positions.copy(), evel=env.enemy_swarm.velocities.copy(),
                                 band=np.int32(env.enemy_band), **extra))
            orig()
            attempts[-1]["lam0"] = None
        env._update_clusters = spy
        gs = np.random.get_state()
        obs, info = env.reset(seed=rs)
        env._update_clusters = orig
        r = record_reset(env, rs, gs)
        r["obs"] = obs
        pids, pxy, _, pncent = flatten_clusters(N, env.clusterer._prev_centroids, {})
        r.update(prev_ids=pids, prev_xy=pxy, prev_ncent=np.int32(pncent), n_attempts=np.int32(len(attempts)),
                 att_epos=np.array([a["epos"] for a in attempts]), att_evel=np.array([a["evel"] for a in attempts]),
                 att_band=np.array([a["band"] for a in attempts], dtype=np.int32),
                 reset_seed=np.int64(-1 if rs is None else rs))
        out.append(r)
    return env, out


def make_reset_retry():
    """Arena of 6 km with N = 5: at 2.4 / 5.8 GHz (comm range 3.1 / 1.3 km) the FSPL graph is often
    disconnected, so lambda2_initial < 0.5 and the loop draws again with seed + attempt + 1."""
    out = {}
    cases = []
    big = dict(arena_size=6000.0, max_steps=20, P_jammer_dbm=30.0, P_jam_thresh_dbm=-45.0, eps=900.0)
    found = 0
    for seed in range(400):
        for rjs in (True, False):
            if found >= 6:
                break
            env, recs = record_retry_case(5, 2, seed, [seed, None, None], random_jammer_start=rjs, **big)
            na = [int(r["n_attempts"]) for r in recs]
            # keep seeds whose seeded reset needs 2..6 attempts and whose unseeded resets retry too
            if 2 <= na[0] <= 6 and max(na[1:]) >= 2 and (found % 2 == int(rjs)):
                cases.append((seed, rjs, dict(big), 100, recs))
                found += 1
    # exhausted loop: 3 attempts allowed, 40 km arena -> never connected at the drawn bands
    for seed in range(400):
        env, recs = record_retry_case(5, 2, seed, [seed], random_jammer_start=True, max_reset_attempts=3,
                                      **{**big, "arena_size": 60000.0, "eps": 3000.0})
        if int(recs[0]["n_attempts"]) == 3 and float(recs[0]["lam0"]) < 0.5:
            cases.append((seed, True, {**big, "arena_size": 60000.0, "eps": 3000.0}, 3, recs))
            break
    for ci, (seed, rjs, cfg, max_att, recs) in enumerate(cases):
        out[f"c{ci}_seed"] = np.int64(seed)
        out[f"c{ci}_rjs"] = np.int32(rjs)
        out[f"c{ci}_max_attempts"] = np.int32(max_att)
        out[f"c{ci}_nresets"] = np.int32(len(recs))
        for k, v in cfg.items():
            out[f"c{ci}_cfg_{k}"] = np.float64(v)
        for ri, r in enumerate(recs):
            for k, v in r.items():
                out[f"c{ci}_r{ri}_{k}"] = v
        print(f"retry case {ci}: seed={seed} random_start={rjs} attempts={[int(r['n_attempts']) for r in recs]} "
              f"lam0={[float(r['lam0']) for r in recs]}")
    out["n_cases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(OUT, "reset_retry.npz"), **out)


# --------------------------------------------------------------------------
# 5. one PPOAgent.update() of the reference (ppo_agent.py:250-373), and its checkpoint
# --------------------------------------------------------------------------
def make_ppo():
    import torch
    from agents.ppo_agent import PPOAgent as RefPPO
    torch.manual_seed(123)
    T, M, B, EPOCHS = 64, 4, 64, 2
    agent = RefPPO(M=M, rollout_length=T, batch_size=B, n_epochs=EPOCHS, device="cpu")
    out = {}
    for k, v in agent.actor.state_dict().items():
        out[f"a0_{k}"] = v.numpy().copy()
    for k, v in agent.critic.state_dict().items():
        out[f"c0_{k}"] = v.numpy().copy()
    rng = np.random.RandomState(5)
    obs = rng.uniform(0, 1, size=(M, 5)).astype(np.float32)
    for t in range(T):
        a, lp, v = agent.get_action(obs)
        agent.store_transition(obs, a, lp, float(rng.normal(0.3, 0.5)), bool(rng.rand() < 0.06), v)
        obs = np.clip(obs + rng.normal(0, 0.1, size=(M, 5)), 0, 1).astype(np.float32)
    b = agent.buffer
    out.update(buf_obs=b.observations.copy(), buf_actions=b.actions.copy(), buf_log_probs=b.log_probs.copy(),
               buf_rewards=b.rewards.copy(), buf_dones=b.dones.astype(np.float32).copy(), buf_values=b.values.copy(),
               last_obs=obs.copy(), last_done=np.int32(0))
    np.random.seed(777)                       # the stream get_minibatches shuffles with (rollout_buffer.py:251-252)
    rr = np.random.RandomState()
    rr.set_state(np.random.get_state())
    perms = []
    for _ in range(EPOCHS):
        idx = np.arange(T * M)
        rr.shuffle(idx)
        perms.append(idx)
    out["perms"] = np.array(perms, dtype=np.int64)
    # advantages / returns as the update sees them: recomputed on a copy of the buffer
    import copy
    b2 = copy.deepcopy(b)
    with torch.no_grad():
        lv = float(agent.critic.forward_pooled(torch.from_numpy(obs).float().mean(dim=0, keepdim=True)).numpy().squeeze())
    b2.compute_returns_and_advantages(lv, False)
    out.update(last_value=np.float32(lv), adv_norm=b2.advantages.copy(), returns=b2.returns.copy())
    stats = agent.update(obs, False)
    for k, v in agent.actor.state_dict().items():
        out[f"a1_{k}"] = v.numpy().copy()
    for k, v in agent.critic.state_dict().items():
        out[f"c1_{k}"] = v.numpy().copy()
    out["stats"] = np.array([stats[k] for k in ("actor_loss", "critic_loss", "entropy", "clip_fraction", "approx_kl")],
                            dtype=np.float64)
    out["hyper"] = np.array([T, M, B, EPOCHS], dtype=np.int64)
    # deployment actor (deploy.py:33-92): parameter names and shapes a exported state dict must have
    deploy = _load("ref_deploy", "../deploy.py")
    ma = deploy.MinimalActor()
    keys = list(ma.state_dict().keys())
    out["deploy_keys"] = np.array(keys)
    out["deploy_shapes"] = np.array([list(ma.state_dict()[k].shape) + [0] * (2 - ma.state_dict()[k].dim())
                                     for k in keys], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "ppo_update.npz"), **out)
    # the reference's own checkpoint after that update (agents/ppo_agent.py:386-420): a file this script wrote
    agent.save(os.path.join(OUT, "ref_ckpt"))
    print("ppo update:", {k: float(v) for k, v in stats.items()})


if __name__ == "__main__":
    if "--ppo-only" in sys.argv:
        make_ppo()
    elif "--retry-only" in sys.argv:
        make_reset_retry()
    elif "--ct-only" in sys.argv:       # adds the coordinated-turn files without touching the others
        make_ct_trajectories()
    elif "--per-agent-only" in sys.argv:
        make_per_agent()
    elif "--sparse-only" in sys.argv:
        make_lambda2_sparse()
    else:
        make_lambda2_sparse()
        make_per_agent()
        make_stage()
        make_edges()
        make_trajectories()
        make_ct_trajectories()
        make_reset_retry()
        make_ppo()
