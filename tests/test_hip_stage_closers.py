"""GPU parity tests of paths that only benchmarks or callers of the reference's helper methods reach:

  * the "step + reward only" form of the step (MARL_STAGE_NO_RECLUSTER, no observation output) that
    `bench.py --config 2` times, at its batch size, against the oracle step with the re-clustering
    block (jammer_env.py:394-402) left out;
  * marl_reward_obs on the reference's stage vectors (RewardCalculator.compute, reward.py:106-176;
    ObservationBuilder.build_vectorized, observation.py:264-332; _compute_avg_jamming_power,
    jammer_env.py:510-560);
  * JammerEnv._compute_lambda2(disrupted) on the facade (jammer_env.py:464-508);
  * marl_lambda2_from_bits beyond 256 vertices against values of the reference's sparse branch
    (communication_graph.py:268-278).
"""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu
from marl_amd import BatchedJammerEnv, EnvConfig, JammerEnv, _lib
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


# ---------------------------------------------------------------- BASELINE configs[1] as benchmarked
@pytest.mark.parametrize("schedule,shards", [("auto", 1), ("split", 1), ("auto", 2)])
def test_step_and_reward_only_matches_oracle_at_bench_size(schedule, shards):
    """4 096 envs x (N=10, M=4), recluster=False, write_obs=False, host-fed reset and noise: every 8th env is
    stepped by the oracle too (adjacency / jamming bits, positions exact; lambda2, reward to 1e-5), all envs
    must leave the observation buffer and the cluster tables of the reset untouched, over 12 steps (step 10
    is where the full step would re-cluster)."""
    E, N, M, T, every = 4096, 10, 4, 12, 8
    rng = np.random.RandomState(77)
    p = orc.EnvParams.fast_preset(N=N, M=M)
    cfg = EnvConfig.fast_preset(N=N, M=M, recluster=False, step_schedule=schedule)
    env = BatchedJammerEnv(E, cfg, seed=3, auto_reset=False, write_obs=False, debug_bits=True, num_shards=shards)
    A = p.arena_size
    epos = rng.uniform(0.1 * A, 0.9 * A, size=(E, N, 2))
    ang = rng.uniform(0, 2 * np.pi, size=(E, N))
    evel = np.stack([2 * np.cos(ang), 2 * np.sin(ang)], axis=2)
    eband = rng.randint(0, 4, size=E).astype(np.int32)
    jinit = rng.uniform(0.1 * A, 0.9 * A, size=(E, M, 2))
    jb = rng.randint(0, 4, size=(E, M)).astype(np.int32)
    obs0, _ = env.reset(enemy_pos=epos, enemy_vel=evel, enemy_band=eband, jam_init=jinit, jam_band=jb)
    torch.cuda.synchronize()
    obs_after_reset = env.obs.clone()
    tables = {k: env.views[k].clone() for k in ("labels", "cent_x", "cent_y", "cent_size", "n_clusters", "ncent", "assign")}
    picked = list(range(0, E, every))
    states = {}
    for e in picked:
        st = orc.EnvState.empty(p)
        o, _ = orc.env_reset(p, st, epos[e], evel[e], eband[e], jinit[e], jb[e])
        np.testing.assert_allclose(obs0[e].cpu().numpy(), o, rtol=1e-5, atol=1e-6)
        states[e] = st
    for t in range(T):
        act = rng.uniform(-6, 6, size=(E, M, 3)).astype(np.float32)
        act[:, :, 2] = rng.uniform(-1, 5, size=(E, M))
        for e in picked[::2]:                       # jammers steered onto link midpoints on the enemy's band
            st = states[e]
            for j in range(M):
                i1, i2 = rng.randint(0, N, size=2)
                tgt = (st.enemy_pos[i1] + st.enemy_pos[i2]) / 2
                act[e, j, :2] = np.clip((tgt - st.jammer_positions[j]) / p.dt, -7, 7)
                act[e, j, 2] = st.enemy_band + 0.5
        noise = rng.normal(0, 0.1, size=(E, N, 2))
        env.fork()
        _, rew, _, trunc, info = env.step(torch.from_numpy(act).cuda(), torch.from_numpy(noise).cuda())
        env.join()
        torch.cuda.synchronize()
        v = env.views
        ex, ey, jx = v["enemy_x"].cpu().numpy(), v["enemy_y"].cpu().numpy(), v["jam_x"].cpu().numpy()
        adjb = env.adj_bits.cpu().numpy().view(np.uint32)
        jamb = env.jam_bits.cpu().numpy().view(np.uint32)
        lam, r = info["lambda2_current"].cpu().numpy(), rew.cpu().numpy()
        terms = info["reward_terms"].cpu().numpy()
        for e in picked:
            _, ro, _, tru, inf = orc.env_step(p, states[e], act[e], noise[e], want_avg_power=False, recluster=False)
            st = states[e]
            assert np.array_equal(ex[e], st.enemy_pos[:, 0]) and np.array_equal(ey[e], st.enemy_pos[:, 1]), (t, e)
            assert np.array_equal(jx[e], st.jammer_positions[:, 0]), (t, e)
            assert np.array_equal(adjb[e], orc.pack_rows(inf["adjacency"] | inf["jammed"])), (t, e)
            assert np.array_equal(jamb[e], orc.pack_rows(inf["jammed"])), (t, e)
            assert gu.lam_close(float(lam[e]), inf["lambda2_current"]), (t, e)
            assert abs(float(r[e]) - ro) <= 1e-5 * max(1.0, abs(ro)), (t, e)
            np.testing.assert_allclose(terms[e], inf["reward_terms"], rtol=1e-5, atol=2e-6)
            assert bool(trunc[e]) == tru
        assert torch.equal(env.obs, obs_after_reset), t                  # the observation is not written
        for k, want in tables.items():                                   # and nothing re-clusters
            assert torch.equal(v[k], want), (t, k)


def test_step_and_reward_only_schedules_agree_bit_for_bit():
    E, N, M = 4096, 10, 4
    envs = [BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M, recluster=False, step_schedule=s), seed=11,
                             auto_reset=True, write_obs=False, num_shards=k)
            for s, k in (("fused", 1), ("split", 1), ("auto", 2))]
    for env in envs:
        env.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(25):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        outs = []
        for env in envs:
            env.fork()
            _, rew, _, _, info = env.step(act)
            env.join()
            outs.append((rew.clone(), info["lambda2_current"].clone(), env.views["enemy_x"].clone()))
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), t


@pytest.mark.parametrize("N,M,E", [(50, 8, 2048), (10, 4, 512), (100, 8, 64)])
def test_timed_step_equals_the_step_and_times_the_solve_launch(N, M, E):
    """marl_env_step_timed issues the launches of marl_env_step (three-launch, one-launch and wide-team schedules)
    and its two events bracket one of them: same results bit for bit, a positive time below the step's own."""
    envs = [BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M), seed=3, auto_reset=True) for _ in range(2)]
    for env in envs:
        env.reset()
    g = torch.Generator(device="cuda").manual_seed(17)
    for t in range(12):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        whole = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        o0, r0, _, _, i0 = envs[0].step(act)
        whole[0].record()
        o1, r1, _, _, i1 = envs[1].step(act, solve_events=ev)
        whole[1].record()
        torch.cuda.synchronize()
        assert torch.equal(o0, o1) and torch.equal(r0, r1), t
        assert torch.equal(i0["lambda2_current"], i1["lambda2_current"]), t
        assert 0.0 < ev[0].elapsed_time(ev[1]) <= whole[0].elapsed_time(whole[1]), t
    with pytest.raises(ValueError):
        BatchedJammerEnv(64, EnvConfig.fast_preset(N=10, M=4), seed=1, num_shards=2).step(
            torch.zeros(64, 4, 3, device="cuda"), solve_events=ev)


# ---------------------------------------------------------------- reward + observation stage
def _stage_groups():
    groups = {}
    for p, d in gu.stage_cases():
        key = (p.N, p.M, p.arena_size, p.P_jammer_dbm, p.P_jam_thresh_dbm)
        groups.setdefault(key, (p, []))[1].append(d)
    return list(groups.values())


@pytest.mark.parametrize("gi", range(len(_stage_groups())))
def test_reward_obs_stage_matches_reference_vectors(gi):
    """marl_reward_obs on the inputs of the reference's stage vectors: total reward, the five components,
    the observation and the average jamming power as the reference computed them."""
    p, cases = _stage_groups()[gi]
    lib = _lib.load()
    cfg = EnvConfig(M=p.M, N=p.N, arena_size=p.arena_size, P_tx_dbm=p.P_tx_dbm, P_sens_dbm=p.P_sens_dbm,
                    P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm)
    P = cfg.to_params()
    E, N, M = len(cases), p.N, p.M
    stack = lambda k: np.stack([np.asarray(c[k]) for c in cases])
    keep = []

    def dv(a, dtype):
        t = _dev(a, dtype)
        keep.append(t)
        return t.data_ptr()

    io = _lib.MarlRewardObsIO()
    io.lambda2, io.lambda2_initial = dv(stack("lam"), torch.float64), dv(stack("lam0"), torch.float64)
    io.enemy_band = dv(stack("band"), torch.int32)
    io.enemy_x, io.enemy_y = dv(stack("epos")[:, :, 0], torch.float64), dv(stack("epos")[:, :, 1], torch.float64)
    io.jam_x, io.jam_y = dv(stack("jpos")[:, :, 0], torch.float64), dv(stack("jpos")[:, :, 1], torch.float64)
    io.jam_band = dv(stack("jb"), torch.int32)
    io.vx, io.vy = dv(stack("vel")[:, :, 0], torch.float32), dv(stack("vel")[:, :, 1], torch.float32)
    io.cent_x, io.cent_y = dv(stack("cent_xy")[:, :, 0], torch.float64), dv(stack("cent_xy")[:, :, 1], torch.float64)
    io.cent_size = dv(stack("cent_size"), torch.int32)
    io.n_clusters, io.ncent = dv(stack("n_clusters"), torch.int32), dv(stack("ncent"), torch.int32)
    assign = stack("assign").astype(np.int32)
    assign[stack("use_assign") == 0] = orc.ASSIGN_NONE                 # reward computed with an empty assignment dict
    io.assign = dv(assign, torch.int32)
    rew = torch.zeros(E, dtype=torch.float32, device="cuda")
    terms = torch.zeros(E, 5, dtype=torch.float32, device="cuda")
    obs = torch.zeros(E, M, 5, dtype=torch.float32, device="cuda")
    avgp = torch.zeros(E, dtype=torch.float64, device="cuda")
    io.reward, io.reward_terms, io.obs, io.avg_power_dbm = rew.data_ptr(), terms.data_ptr(), obs.data_ptr(), avgp.data_ptr()
    _lib.check(lib.marl_reward_obs(C.byref(P), E, C.byref(io), _stream()), "marl_reward_obs")
    torch.cuda.synchronize()
    for e, c in enumerate(cases):
        want = float(c["reward"])
        assert abs(float(rew[e]) - want) <= 1e-5 * max(1.0, abs(want)), e
        np.testing.assert_allclose(terms[e].cpu().numpy(), c["comps"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(obs[e].cpu().numpy(), c["obs"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(float(avgp[e]), float(c["avgp"]), rtol=1e-9)


def test_reward_obs_op_equals_the_step():
    """torch.ops.marl_amd.reward_obs on a stepped env reproduces what the step itself wrote."""
    from marl_amd import ops
    cfg = EnvConfig.fast_preset(N=50, M=8)
    env = BatchedJammerEnv(64, cfg, seed=9, auto_reset=False, track_avg_power=True)
    env.reset()
    h = ops.register_env(env)
    g = torch.Generator(device="cuda").manual_seed(2)
    for t in range(11):
        act = (torch.rand(64, 8, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (64, 8), device="cuda", generator=g).float()
        obs, rew, _, _, info = env.step(act)
    r, terms, o, avgp = torch.ops.marl_amd.reward_obs(env._buf, h, info["lambda2_current"])
    assert torch.equal(r, rew) and torch.equal(o, obs) and torch.equal(terms, info["reward_terms"])
    assert torch.equal(avgp, info["avg_jamming_power_dbm"])


# ---------------------------------------------------------------- facade: _compute_lambda2
def test_facade_compute_lambda2_unjammed_and_jammed():
    """The reference's own use (tests/test_jammer_env.py:358-371) and the values: unjammed = lambda2_initial of
    the reset, jammed = lambda2_current of the step that produced the state; both equal to the oracle's."""
    env = JammerEnv(N=10, M=4, seed=42)
    env.reset()
    un, ja = env._compute_lambda2(disrupted=False), env._compute_lambda2(disrupted=True)
    assert np.isfinite(un) and np.isfinite(ja)
    assert un == pytest.approx(env.lambda2_initial, abs=1e-9)
    env = JammerEnv(N=30, M=6, seed=5, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0, random_jammer_start=True)
    env.reset(seed=5)
    p = orc.EnvParams(M=6, N=30, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0, random_jammer_start=True)
    rng = np.random.RandomState(0)
    for t in range(5):
        a = rng.uniform(-5, 5, size=(6, 3)).astype(np.float32)
        a[:, 2] = env.enemy_band
        _, _, _, _, info = env.step(a)
    st = orc.EnvState.empty(p)
    st.enemy_pos, st.enemy_band = env.enemy_swarm.positions, env.enemy_band
    st.jammer_positions, st.jammer_bands = env.jammer_positions, env.jammer_bands
    assert gu.lam_close(env._compute_lambda2(disrupted=True), orc.compute_lambda2(p, st, True)[0])
    assert gu.lam_close(env._compute_lambda2(disrupted=False), orc.compute_lambda2(p, st, False)[0])
    assert env._compute_lambda2(True) == pytest.approx(info["lambda2_current"], abs=1e-9)


# ---------------------------------------------------------------- lambda2 beyond 256 vertices
def test_lambda2_of_large_graphs_equals_reference_sparse_branch():
    """tests/golden/lambda2_sparse.npz: compute_lambda2(L, use_sparse=True) of the reference (scipy eigsh,
    which='SM') on graphs of 300, 640 and 1 024 vertices.  On connected graphs that branch and the dense one
    agree to 1e-12 and the solver must match both.  On disconnected graphs the reference's sparse branch is
    itself off (ARPACK returns one copy of the multiple zero eigenvalue and the next distinct one: 0.12 for a
    graph whose lambda2 is 0); there the solver is held to the dense branch, the one the env takes."""
    import marl_amd.ops  # noqa: F401
    z = gu.load("lambda2_sparse.npz")
    for k in range(int(z["n_graphs"])):
        n = int(z[f"g{k}_n"])
        bits = torch.from_numpy(z[f"g{k}_adj"].view(np.int32)[None]).cuda().contiguous()
        got = float(torch.ops.marl_amd.lambda2_from_bits(bits, 0)[0])
        sparse, dense = float(z[f"g{k}_lam_sparse"]), float(z[f"g{k}_lam_dense"])
        assert abs(got - dense) <= 1e-9 * max(1.0, abs(dense)), (n, got, dense)
        if abs(dense) > 1e-9:
            assert abs(got - sparse) <= 1e-9 * max(1.0, abs(sparse)), (n, got, sparse)
