"""GPU parity tests for the corners of the path the trajectory goldens do not reach: the reset retry
loop (attempts that fail the lambda2_initial test), the reference's enemy-reflection / speed-floor /
action-decode vectors through the step kernel itself, and invariance of the generator-mode batch under
sharding (the multi-GPU layout, on one GPU)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from marl_amd import BatchedJammerEnv, EnvConfig, JammerEnv, _lib
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def cfg_from(p: orc.EnvParams, **kw) -> EnvConfig:
    return EnvConfig(M=p.M, N=p.N, arena_size=p.arena_size, max_steps=p.max_steps, dt=p.dt, v_max=p.v_max,
                     v_enemy=p.v_enemy, P_tx_dbm=p.P_tx_dbm, P_sens_dbm=p.P_sens_dbm,
                     P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                     min_samples=p.min_samples, K_recompute=p.K_recompute, max_reset_attempts=p.max_reset_attempts,
                     random_jammer_start=p.random_jammer_start, enemy_mode=p.enemy_mode, omega_max=p.omega_max, **kw)


def lam_ok(a, b):
    return abs(a - b) <= 1e-5 * max(abs(b), 1.0)


# ---------------------------------------------------------------- reset retry loop (jammer_env.py:283-316)
def test_facade_reset_retry_loop_reproduces_reference():
    """JammerEnv(seed).reset(seed) and two unseeded resets on configurations where the first attempts
    give lambda2_initial < 0.5: the facade redraws with seed + attempt + 1 (or from the running
    streams) exactly like the reference and ends in its state, previous-centroid table included."""
    for p, seed, recs in gu.retry_cases():
        np.random.seed(seed + 1000)
        env = JammerEnv(M=p.M, N=p.N, seed=seed, arena_size=p.arena_size, max_steps=p.max_steps,
                        P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                        random_jammer_start=p.random_jammer_start, max_reset_attempts=p.max_reset_attempts)
        for ri, r in enumerate(recs):
            rs = int(r["reset_seed"])
            obs, info = env.reset(seed=None if rs < 0 else rs)
            v = env._b.views
            assert np.array_equal(env.enemy_swarm.positions, r["epos"]), (seed, ri)
            assert np.array_equal(env.enemy_swarm.velocities, r["evel"])
            assert info["enemy_band"] == int(r["band"])
            assert lam_ok(info["lambda2_initial"], float(r["lam0"])), (seed, ri)
            assert abs(env.R_jam - float(r["R_jam"])) <= 1e-12 * float(r["R_jam"])
            assert np.array_equal(env.clusterer.labels.astype(np.int32), r["labels"])
            nc, pnc = int(r["ncent"]), int(r["prev_ncent"])
            assert int(v["ncent"][0]) == nc and int(v["prev_ncent"][0]) == pnc, (seed, ri)
            assert np.array_equal(v["cent_x"][0, :nc].cpu().numpy(), r["cent_xy"][:nc, 0])
            assert np.array_equal(v["prev_x"][0, :pnc].cpu().numpy(), r["prev_xy"][:pnc, 0]), (seed, ri)
            assert np.array_equal(v["prev_y"][0, :pnc].cpu().numpy(), r["prev_xy"][:pnc, 1])
            assert int(v["prev_n_clusters"][0]) == int(np.sum(r["prev_ids"][:pnc] >= 0))
            assert np.array_equal(v["assign"][0].cpu().numpy(), r["assign"])
            assert np.array_equal(env.jammer_positions, r["jpos"]), (seed, ri)
            assert np.array_equal(env.jammer_bands, r["jb"])
            np.testing.assert_allclose(obs, r["obs"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("random_start", [True, False])
def test_device_reset_retry_loop_matches_oracle(random_start):
    """The in-kernel retry loop (generator mode).  The draws of attempt k do not depend on how many
    attempts are allowed, so running the same reset with max_reset_attempts = 1, 2, ... exposes the draws
    of every attempt; the oracle's loop fed with them must end where the device loop ends: accepted
    attempt, cluster tables, previous centroids, assignments."""
    E, N, M, KMAX = 256, 5, 2, 8
    base = dict(N=N, M=M, arena_size=6000.0, max_steps=20, P_jammer_dbm=30.0, P_jam_thresh_dbm=-45.0, eps=900.0,
                random_jammer_start=random_start)
    names = ("enemy_x", "enemy_y", "enemy_vx", "enemy_vy", "enemy_band", "lambda2_initial", "labels", "cent_x",
             "cent_y", "cent_size", "ncent", "n_clusters", "prev_x", "prev_y", "prev_ncent", "prev_n_clusters",
             "assign", "jam_x", "jam_y", "jam_band", "step_count")
    snaps = []
    for k in range(1, KMAX + 1):
        env = BatchedJammerEnv(E, EnvConfig(max_reset_attempts=k, **base), seed=77, auto_reset=False)
        obs, _ = env.reset()
        torch.cuda.synchronize()
        snap = {n: env.views[n].cpu().numpy().copy() for n in names}
        snap["obs"] = obs.cpu().numpy().copy()
        snaps.append(snap)
    final = snaps[-1]
    lam0 = final["lambda2_initial"]
    # first accepted attempt per env = first k whose outcome passes (and then never changes again)
    passed = np.stack([s["lambda2_initial"] >= 0.5 for s in snaps])            # [KMAX, E]
    first = np.where(passed.any(0), passed.argmax(0), KMAX - 1)
    assert (first > 0).sum() > E // 10 and (first == 0).sum() > E // 10          # both branches are exercised
    assert np.all((lam0 >= 0.5) | (first == KMAX - 1))
    p = orc.EnvParams(**{k: v for k, v in base.items()}, max_reset_attempts=KMAX)
    for e in range(E):
        a = int(first[e])
        for later in snaps[a:]:                                                  # accepted: more attempts change nothing
            assert np.array_equal(later["enemy_x"][e], snaps[a]["enemy_x"][e])
        attempts = [(np.stack([snaps[k]["enemy_x"][e], snaps[k]["enemy_y"][e]], 1),
                     np.stack([snaps[k]["enemy_vx"][e], snaps[k]["enemy_vy"][e]], 1),
                     int(snaps[k]["enemy_band"][e])) for k in range(a + 1)]
        st = orc.EnvState.empty(p)
        jpos = np.stack([final["jam_x"][e], final["jam_y"][e]], 1)
        obs, info, used = orc.env_reset_with_retries(p, st, attempts, jpos if random_start else np.zeros((M, 2)),
                                                     final["jam_band"][e])
        assert used == a + 1
        assert lam_ok(float(lam0[e]), info["lambda2_initial"])
        nc, pnc = st.clusters.ncent, st.clusters.prev_ncent
        assert int(final["ncent"][e]) == nc and int(final["prev_ncent"][e]) == pnc, e
        assert int(final["n_clusters"][e]) == st.clusters.n_clusters
        assert int(final["prev_n_clusters"][e]) == st.clusters.prev_n_clusters
        assert np.array_equal(final["labels"][e], st.clusters.labels)
        assert np.array_equal(final["cent_x"][e, :nc], st.clusters.cent_xy[:nc, 0])
        assert np.array_equal(final["cent_size"][e, :nc], st.clusters.cent_size[:nc])
        assert np.array_equal(final["prev_x"][e, :pnc], st.clusters.prev_xy[:pnc, 0]), e
        assert np.array_equal(final["prev_y"][e, :pnc], st.clusters.prev_xy[:pnc, 1]), e
        assert np.array_equal(final["assign"][e], st.assign)
        assert int(final["step_count"][e]) == 0
        if random_start:
            np.testing.assert_allclose(final["obs"][e], obs, rtol=1e-5, atol=1e-6)
        else:   # centroid start: the jammer sits within the spread around its cluster's centroid
            for j in range(M):
                c = int(st.assign[j])
                slot = c if c >= 0 else st.clusters.n_clusters
                assert np.all(np.abs(jpos[j] - np.clip(st.clusters.cent_xy[slot], 0, p.arena_size)) <= 10.0 + 1e-9)


@pytest.mark.parametrize("random_start,N,M", [(True, 50, 8), (False, 50, 8), (True, 64, 16), (False, 10, 4)])
def test_reset_shortcut_for_always_complete_graphs_equals_general_reset(random_start, N, M):
    """Where every generated spawn gives the complete graph (arena diagonal below every band's range) the reset
    launch skips the adjacency test and the eigensolver; state, observations and everything downstream are those
    of the general kernel bit for bit, over whole-batch resets, masked auto-resets and a reset of a chosen half."""
    E = 384
    mk = lambda shortcut: BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M, max_steps=13, reset_shortcut=shortcut,
                                                                    random_jammer_start=random_start),
                                           seed=21, auto_reset=True)
    a, b = mk(True), mk(False)
    oa, _ = a.reset()
    ob, _ = b.reset()
    assert torch.equal(oa, ob) and torch.all(a.views["lambda2_initial"] == float(N))
    g = torch.Generator(device="cuda").manual_seed(9)
    half = (torch.arange(E, device="cuda") % 2 == 0)
    for t in range(30):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        ra, rb = a.step(act), b.step(act)
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and torch.equal(ra[3], rb[3]), t
        if t == 20:
            oa, _ = a.reset(mask=half)
            ob, _ = b.reset(mask=half)
            assert torch.equal(oa, ob)
        for k in a.views:
            if k not in ("solve_rows", "solve_np", "solve_slot", "solve_perm", "solve_partial"):
                assert torch.equal(a.views[k], b.views[k]), (t, k)


# ---------------------------------------------------------------- reference edge vectors through the step kernel
def test_reflection_speed_floor_and_action_decode_vectors_through_the_step_kernel():
    """edges.npz refl_* (enemies leaving through every wall and a corner, zero velocity with zero noise ->
    speed floor 1e-6; recorded from EnemySwarm.step) and act_* (ActionHandler.apply_action / clip_bands on
    float32 actions) fed to marl_env_step itself: positions, velocities and bands bit-exact."""
    z = gu.load("edges.npz")
    N = len(z["refl_pos0"])
    cfg = EnvConfig(N=N, M=2, arena_size=200.0, dt=0.5, v_enemy=2.0, random_jammer_start=True)
    env = BatchedJammerEnv(1, cfg, auto_reset=False)
    env.reset(enemy_pos=z["refl_pos0"][None], enemy_vel=z["refl_vel0"][None], enemy_band=np.array([2], np.int32),
              jam_init=np.full((1, 2, 2), 100.0), jam_band=np.zeros((1, 2), np.int32))
    env.step(torch.zeros(1, 2, 3, device="cuda"), torch.from_numpy(z["refl_noise"][None]).cuda())
    torch.cuda.synchronize()
    v = env.views
    assert np.array_equal(v["enemy_x"][0].cpu().numpy(), z["refl_pos1"][:, 0])
    assert np.array_equal(v["enemy_y"][0].cpu().numpy(), z["refl_pos1"][:, 1])
    assert np.array_equal(v["enemy_vx"][0].cpu().numpy(), z["refl_vel1"][:, 0])
    assert np.array_equal(v["enemy_vy"][0].cpu().numpy(), z["refl_vel1"][:, 1])
    # action decode: positions for dt = 0.5 and 0.1
    M = len(z["act_pos0"])
    for dt, key in ((0.5, "act_pos1_dt05"), (0.1, "act_pos1_dt01")):
        cfg = EnvConfig(N=4, M=M, arena_size=200.0, dt=dt, v_max=5.0, random_jammer_start=True)
        env = BatchedJammerEnv(1, cfg, seed=1, auto_reset=False)
        rng = np.random.RandomState(0)
        env.reset(enemy_pos=rng.uniform(20, 180, (1, 4, 2)), enemy_vel=np.zeros((1, 4, 2)) + 1.0,
                  enemy_band=np.array([1], np.int32), jam_init=z["act_pos0"][None], jam_band=np.zeros((1, M), np.int32))
        act = np.zeros((1, M, 3), np.float32)
        act[0, :, :2] = z["act_vel"]
        env.step(torch.from_numpy(act).cuda(), torch.zeros(1, 4, 2, dtype=torch.float64, device="cuda"))
        torch.cuda.synchronize()
        got = np.stack([env.views["jam_x"][0].cpu().numpy(), env.views["jam_y"][0].cpu().numpy()], 1)
        assert np.array_equal(got, z[key]), key
        assert np.array_equal(env.views["last_vx"][0].cpu().numpy(), z["act_vel"][:, 0])       # raw, pre-clip
    # band decode: truncation towards zero, then clip to 0..3
    Mb = len(z["act_bands_in"])
    cfg = EnvConfig(N=4, M=Mb, arena_size=200.0, random_jammer_start=True)
    env = BatchedJammerEnv(1, cfg, seed=1, auto_reset=False)
    env.reset()
    act = np.zeros((1, Mb, 3), np.float32)
    act[0, :, 2] = z["act_bands_in"]
    env.step(torch.from_numpy(act).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(env.views["jam_band"][0].cpu().numpy(), z["act_bands_out"])


# ---------------------------------------------------------------- shard invariance (multi-GPU layout on one GPU)
@pytest.mark.parametrize("N,M,E", [(50, 8, 2048), (10, 4, 512), (100, 12, 96)])
def test_two_shards_equal_one_batch_across_auto_resets(N, M, E):
    """Rank r of a multi-GPU job owns envs [r*E/2, (r+1)*E/2) with env_offset = r*E/2.  Two such shards,
    stepped side by side in generator mode with auto-reset, must reproduce the single batch of E envs
    bit for bit over two episode boundaries: the device draws are keyed by the global env index."""
    T = 9
    cfg = EnvConfig.fast_preset(N=N, M=M, max_steps=T)
    whole = BatchedJammerEnv(E, cfg, seed=31, auto_reset=True)
    halves = [BatchedJammerEnv(E // 2, cfg, seed=31, auto_reset=True, env_offset=r * (E // 2)) for r in range(2)]
    ow, _ = whole.reset()
    oh = [h.reset()[0] for h in halves]
    assert torch.equal(ow, torch.cat(oh))
    g = torch.Generator(device="cuda").manual_seed(5)
    names = ("enemy_x", "enemy_y", "enemy_vx", "jam_x", "jam_band", "labels", "assign", "cent_x", "step_count",
             "enemy_band", "lambda2_initial", "tick")
    for t in range(2 * T + 3):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 11
        act[:, :, 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        ow, rw, _, tw, iw = whole.step(act)
        outs = [h.step(act[r * (E // 2):(r + 1) * (E // 2)].contiguous()) for r, h in enumerate(halves)]
        assert torch.equal(ow, torch.cat([o[0] for o in outs])), t
        assert torch.equal(rw, torch.cat([o[1] for o in outs])), t
        assert torch.equal(tw, torch.cat([o[3] for o in outs])), t
        assert torch.equal(iw["lambda2_current"], torch.cat([o[4]["lambda2_current"] for o in outs])), t
        assert bool(tw.all()) == ((t + 1) % T == 0)
        for n in names:
            assert torch.equal(whole.views[n], torch.cat([h.views[n] for h in halves])), (t, n)
    # checksum of checksums, the form a multi-GPU job would compare
    def digest(env):
        return sum(int(env.views[n].view(torch.uint8).to(torch.int64).sum()) for n in ("enemy_x", "jam_x", "labels"))
    assert digest(whole) == digest(halves[0]) + digest(halves[1])


@pytest.mark.parametrize("N,M,E,S", [(50, 8, 4099, 3), (10, 4, 2048, 2), (33, 5, 1500, 4)])
def test_parts_on_their_own_streams_equal_one_launch(N, M, E, S):
    """num_shards=S steps the batch as S parts on S streams (no waiting between steps); after join() every
    output and the whole state equal those of the single launch, over two auto-resets, with the episode
    statistics summed over the parts."""
    T = 7
    cfg = EnvConfig.fast_preset(N=N, M=M, max_steps=T)
    one = BatchedJammerEnv(E, cfg, seed=77, auto_reset=True, env_offset=11)
    parts = BatchedJammerEnv(E, cfg, seed=77, auto_reset=True, env_offset=11, num_shards=S)
    assert len(parts._shards) == S and sum(c for _, c, _, _ in parts._shards) == E
    o1, _ = one.reset()
    o2, _ = parts.reset()
    assert torch.equal(o1, o2)
    g = torch.Generator(device="cuda").manual_seed(9)
    acts = (torch.rand(2 * T + 2, E, M, 3, device="cuda", generator=g) - 0.5) * 11
    acts[..., 2] = torch.randint(0, 4, (2 * T + 2, E, M), device="cuda", generator=g).float()
    st1 = torch.zeros(4, dtype=torch.float64, device="cuda")
    st2 = torch.zeros(4, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    parts.fork()
    for t in range(acts.shape[0]):
        o1, r1, _, t1, i1 = one.step(acts[t], episode_stats=st1)
        o2, r2, _, t2, i2 = parts.step(acts[t], episode_stats=st2)
        if t % 5 == 4 or t == acts.shape[0] - 1:
            parts.join()
            torch.cuda.synchronize()
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2), t
            assert torch.equal(i1["lambda2_current"], i2["lambda2_current"]), t
            assert torch.equal(i1["reward_terms"], i2["reward_terms"]), t
            for n, v in one.views.items():
                if not n.startswith("solve_") and n != "workspace":
                    assert torch.equal(v, parts.views[n]), (t, n)
    assert st1[0] == 2 * E and st2[0] == st1[0] and st2[3] == st1[3]
    assert torch.allclose(st1, st2, rtol=1e-12)
    # a full reset goes through the current stream and orders itself against the parts
    assert torch.equal(one.reset(seed=5)[0], parts.reset(seed=5)[0])
    o1 = one.step(acts[0])[0]
    o2 = parts.step(acts[0])[0]
    parts.join()
    assert torch.equal(o1, o2)


def test_parts_need_the_wave_kernels():
    with pytest.raises(_lib.MarlError):
        BatchedJammerEnv(64, EnvConfig.fast_preset(N=100, M=4), num_shards=2)
    with pytest.raises(ValueError):
        BatchedJammerEnv(4, EnvConfig.fast_preset(N=10, M=4), num_shards=5)


# ---------------------------------------------------------------- float32 first stage of the step's jamming test
@pytest.mark.parametrize("arena,band,extra", [(200.0, 0, 0), (200.0, 3, 0), (6000.0, 2, 0), (200.0, 0, 58), (200.0, 1, 158),
                                              (200.0, 2, 214)])
def test_step_jamming_mask_is_exact_around_the_float32_guard_band(arena, band, extra):
    """The step evaluates the jamming test in float32 first and settles the pairs inside a guard band around
    the radius with the exact float64 expression.  Link midpoints are planted at R (1 + eps) from a jammer for
    eps from one ulp to 1e-4 on both sides (inside the band, at its edges and outside), with the enemies held
    still (v_enemy = 0) so that the planted geometry is what the step sees: bits equal the reference expression.
    `extra` further enemies at random places bring the swarm to 100, 200 and 256: the kernels for more than 64 enemies
    have a first stage of their own."""
    rng = np.random.RandomState(5)
    eps_list = [0.0, 1.2e-16, -1.2e-16, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 5e-7, -5e-7, 1e-6, -1e-6, 2e-6, -2e-6,
                4e-6, -4e-6, 1e-5, -1e-5, 1e-4, -1e-4]
    N, M, E = 2 * len(eps_list) + extra, 2, 8
    p = orc.EnvParams(M=M, N=N, arena_size=arena, P_jammer_dbm=20.0 if arena < 1000 else 30.0,
                      P_jam_thresh_dbm=-40.0 if arena < 1000 else -45.0, v_enemy=0.0, random_jammer_start=True)
    R = orc.compute_jam_range(orc.db_to_watts(p.P_jammer_dbm), orc.db_to_watts(p.P_jam_thresh_dbm),
                              orc.FREQUENCY_BANDS[band])
    env = BatchedJammerEnv(E, cfg_from(p), auto_reset=False, debug_bits=True)
    epos = np.zeros((E, N, 2))
    jpos = np.zeros((E, M, 2))
    for e in range(E):
        q = rng.uniform(0.3 * arena, 0.7 * arena, size=2)
        jpos[e, 0] = q
        jpos[e, 1] = rng.uniform(0.1 * arena, 0.9 * arena, size=2)
        for t, eps in enumerate(eps_list):
            ang, half = rng.uniform(0, 2 * np.pi), rng.uniform(0.5, 0.2 * R + 0.5)
            mid = q + R * (1.0 + eps) * np.array([np.cos(ang), np.sin(ang)])
            d = half * np.array([-np.sin(ang), np.cos(ang)])
            epos[e, 2 * t], epos[e, 2 * t + 1] = mid + d, mid - d
        epos[e, 2 * len(eps_list):] = rng.uniform(0.0, arena, size=(extra, 2))
        epos[e] = np.clip(epos[e], 0.0, arena)
    jb = np.tile(np.array([band, (band + 1) % 4], np.int32), (E, 1))
    env.reset(enemy_pos=epos, enemy_vel=np.tile(np.array([1.0, 0.0]), (E, N, 1)),
              enemy_band=np.full(E, band, np.int32), jam_init=jpos, jam_band=jb)
    act = np.zeros((E, M, 3), np.float32)
    act[:, :, 2] = jb
    for step in range(2):
        env.step(torch.from_numpy(act).cuda(), torch.zeros(E, N, 2, dtype=torch.float64, device="cuda"))
        torch.cuda.synchronize()
        near = 0
        for e in range(E):
            assert np.array_equal(env.views["enemy_x"][e].cpu().numpy(), epos[e, :, 0])       # held still
            want = orc.disrupted_links(jpos[e], jb[e], epos[e], band, orc.db_to_watts(p.P_jammer_dbm),
                                       orc.db_to_watts(p.P_jam_thresh_dbm))
            got = env.jam_bits[e].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, orc.pack_rows(want)), (arena, band, e)
            dmid = orc.midpoint_distances(jpos[e, :1], epos[e])[0]
            near += int(np.sum(np.abs(dmid / R - 1.0) < 3e-6)) // 2
        assert near >= E * 12                                   # the planted pairs did land inside the band
