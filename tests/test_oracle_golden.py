"""Pins the CPU oracle (oracle/jammer_oracle.py) to the reference: every stage,
edge case and whole trajectories are compared with vectors the reference itself
produced (tests/golden/make_golden.py).  Boolean / integer outputs bit-exact;
float64 outputs that the oracle computes in the reference's operation order
exactly equal; lambda2 / reward / observation within 1e-5 relative."""
import math

import numpy as np
import pytest

import golden_util as gu
from oracle import jammer_oracle as orc


# ---------------------------------------------------------------- stage vectors
@pytest.mark.parametrize("idx", range(104))
def test_stage_case(idx):
    p, d = gu.stage_cases()[idx]
    band = int(d["band"])
    f = orc.FREQUENCY_BANDS[band]
    tx, sens = orc.db_to_watts(p.P_tx_dbm), orc.db_to_watts(p.P_sens_dbm)
    pj, pt = orc.db_to_watts(p.P_jammer_dbm), orc.db_to_watts(p.P_jam_thresh_dbm)
    A = orc.adjacency_matrix(d["epos"], tx, sens, f)
    J = orc.disrupted_links(d["jpos"], d["jb"], d["epos"], band, pj, pt)
    assert np.array_equal(orc.pack_rows(A), d["adj"])
    assert np.array_equal(orc.pack_rows(J), d["jam"])
    Aj = A & ~J
    assert np.array_equal(Aj.sum(axis=1).astype(np.int32), d["deg"])
    lam = orc.lambda2_of_laplacian(orc.laplacian(Aj))
    lam0 = orc.lambda2_of_laplacian(orc.laplacian(A))
    assert lam == float(d["lam"]) and lam0 == float(d["lam0"])
    cs = orc.ClusterState.empty(p.N)
    orc.cluster_fit(cs, d["epos"], p.eps, p.min_samples, p.arena_size)
    assert np.array_equal(cs.labels.astype(np.int32), d["labels"])
    assert cs.n_clusters == int(d["n_clusters"]) and cs.ncent == int(d["ncent"])
    assert cs.cent_ids() == [int(x) for x in d["cent_ids"][:cs.ncent]]
    assert np.array_equal(cs.cent_xy[:cs.ncent], d["cent_xy"][:cs.ncent])
    assert np.array_equal(cs.cent_size[:cs.ncent], d["cent_size"][:cs.ncent])
    assign = orc.assign_proportional(p.M, cs)
    assert np.array_equal(assign.astype(np.int32), d["assign"])
    R = orc.compute_jam_range(pj, pt, f)
    assert R == float(d["R_jam"])
    use = assign if int(d["use_assign"]) else np.full(p.M, orc.ASSIGN_NONE)
    total, comps = orc.reward5(p, lam, lam0, d["jb"], band, d["jpos"], cs, d["vel"], use, R)
    np.testing.assert_allclose(total, float(d["reward"]), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(comps, d["comps"], rtol=1e-6, atol=1e-12)
    obs = orc.build_observation(p, d["jpos"], d["jb"], cs, band, R)
    assert obs.dtype == np.float32
    np.testing.assert_allclose(obs, d["obs"], rtol=1e-6, atol=1e-7)
    avgp = orc.avg_jamming_power_dbm(d["jpos"], d["jb"], d["epos"], band, pj)
    np.testing.assert_allclose(avgp, float(d["avgp"]), rtol=1e-12)


# ---------------------------------------------------------------- edge vectors
def test_closed_form_lambda2():
    z = gu.load("edges.npz")
    for n in (2, 3, 4, 5, 6, 10, 50):      # K_n -> n  (reference tests/test_laplacian.py:261-284)
        A = ~np.eye(n, dtype=bool)
        lam = orc.lambda2_of_laplacian(orc.laplacian(A))
        assert lam == float(z[f"lam_K{n}"])
        assert abs(lam - n) < 1e-10
    for n in (5, 10, 50):                   # path / cycle (tests/test_laplacian.py:286-310)
        A = np.zeros((n, n), dtype=bool)
        for i in range(n - 1):
            A[i, i + 1] = A[i + 1, i] = True
        lam = orc.lambda2_of_laplacian(orc.laplacian(A))
        assert lam == float(z[f"lam_P{n}"])
        assert abs(lam - 2 * (1 - math.cos(math.pi / n))) < 1e-6
        A[0, n - 1] = A[n - 1, 0] = True
        lam = orc.lambda2_of_laplacian(orc.laplacian(A))
        assert lam == float(z[f"lam_C{n}"])
        assert abs(lam - 2 * (1 - math.cos(2 * math.pi / n))) < 1e-6
    assert abs(float(z["lam_two_paths"])) < 1e-10     # disconnected -> 0


def test_random_graph_lambda2():
    z = gu.load("edges.npz")
    for k in range(int(z["n_graphs"])):
        n = int(z[f"g{k}_n"])
        A = orc.unpack_rows(z[f"g{k}_adj"], n)
        assert orc.lambda2_of_laplacian(orc.laplacian(A)) == float(z[f"g{k}_lam"])


def test_unit_identities():
    # reference tests/test_fspl.py:40-58, 196-206
    assert abs(orc.db_to_watts(20.0) - 0.1) < 1e-15
    assert abs(orc.db_to_watts(-90.0) - 1e-12) < 1e-24
    assert abs(orc.watts_to_db(0.1) - 20.0) < 1e-12
    f = 2.4e9
    R = orc.compute_jam_range(0.1, 1e-12, f)
    assert abs(R - (3e8 / f) / (4 * np.pi) * math.sqrt(0.1 / 1e-12)) < 1e-10
    # power at the range equals the threshold; inverse-square law (tests/test_fspl.py:134-144)
    assert abs(orc.received_power_watts(0.1, R, f) / 1e-12 - 1) < 1e-12
    assert abs(orc.received_power_watts(0.1, 10.0, f) / orc.received_power_watts(0.1, 20.0, f) - 4) < 1e-12


def test_jam_mask_edges():
    z = gu.load("edges.npz")
    pj, pt = orc.db_to_watts(20.0), orc.db_to_watts(-40.0)
    epos = z["jam_epos"]
    for name, jp, jb in (("on_mid", [[50.0, 50.0]], [2]), ("wrong_band", [[50.0, 50.0]], [1]),
                         ("far", [[50.0, 2050.0]], [2])):
        J = orc.disrupted_links(np.array(jp), np.array(jb), epos, 2, pj, pt)
        assert np.array_equal(orc.pack_rows(J), z[f"jam_{name}"])
    J = orc.disrupted_links(np.array([[50.0, 50.0]]), np.array([2]), epos, 2, pj, pt)
    assert J[0, 1] and J[1, 0] and not J[0, 0]            # tests/test_jamming.py:174-206, 312-325
    # knife edge around the comm radius / jam radius
    tx, sens = orc.db_to_watts(20.0), orc.db_to_watts(-90.0)
    A = orc.adjacency_matrix(z["knife_pos"], tx, sens, orc.FREQUENCY_BANDS[3])
    assert np.array_equal(orc.pack_rows(A), z["knife_adj"])
    J = orc.disrupted_links(np.array([[0.0, 0.0]]), np.array([3]), z["knife_jam_epos"], 3, pj, pt)
    assert np.array_equal(orc.pack_rows(J), z["knife_jam"])


def test_dbscan_edges():
    z = gu.load("edges.npz")
    for name in z["db_names"]:
        pts = z[f"db_{name}_pts"]
        eps, ms = z[f"db_{name}_par"]
        cs = orc.ClusterState.empty(len(pts))
        orc.cluster_fit(cs, pts, float(eps), int(ms), 200.0)
        assert np.array_equal(cs.labels.astype(np.int32), z[f"db_{name}_labels"]), name
        nc = int(z[f"db_{name}_ncent"])
        assert cs.ncent == nc
        assert cs.cent_ids() == [int(x) for x in z[f"db_{name}_ids"][:nc]]
        assert np.array_equal(cs.cent_xy[:nc], z[f"db_{name}_xy"][:nc]), name
        assert np.array_equal(cs.cent_size[:nc], z[f"db_{name}_size"][:nc])


def test_dbscan_matches_sklearn_when_present():
    sk = pytest.importorskip("sklearn.cluster")
    rng = np.random.RandomState(0)
    for trial in range(60):
        n = int(rng.randint(2, 70))
        pts = rng.uniform(0, 100, size=(n, 2))
        eps = float(rng.uniform(4, 25))
        ms = int(rng.randint(1, 6))
        ref = sk.DBSCAN(eps=eps, min_samples=ms).fit_predict(pts)
        assert np.array_equal(orc.dbscan_labels(pts, eps, ms), ref)


def test_enemy_reflection_and_action_decode():
    z = gu.load("edges.npz")
    pos, vel = orc.enemy_step_random_walk(z["refl_pos0"].copy(), z["refl_vel0"].copy(),
                                          z["refl_noise"], 2.0, 0.5, 200.0)
    assert np.array_equal(pos, z["refl_pos1"]) and np.array_equal(vel, z["refl_vel1"])
    assert np.array_equal(orc.apply_action(z["act_pos0"], z["act_vel"], 0.5, 5.0, 200.0),
                          z["act_pos1_dt05"])
    assert np.array_equal(orc.apply_action(z["act_pos0"], z["act_vel"], 0.1, 5.0, 200.0),
                          z["act_pos1_dt01"])
    assert np.array_equal(orc.clip_bands(z["act_bands_in"]), z["act_bands_out"])
    # reference tests/test_action_space.py:116-146
    assert list(orc.clip_bands(np.array([-1, 0, 3, 5]))) == [0, 0, 3, 3]
    assert np.array_equal(orc.apply_action(np.array([[100.0, 100.0]]), np.array([[10.0, 0.0]]),
                                           1.0, 5.0, 200.0), [[105.0, 100.0]])
    assert np.array_equal(orc.apply_action(np.array([[199.0, 1.0]]), np.array([[5.0, -5.0]]),
                                           1.0, 5.0, 200.0), [[200.0, 0.0]])


def test_gae():
    z = gu.load("edges.npz")
    adv, ret = orc.gae(z["gae_rew"], z["gae_val"], z["gae_done"], float(z["gae_last_value"]),
                       False, 0.99, 0.95)
    assert np.array_equal(ret, z["gae_ret"])
    norm = (adv - adv.mean()) / (adv.std() + 1e-8)
    np.testing.assert_allclose(norm, z["gae_adv_norm"], rtol=1e-5, atol=1e-6)


def test_reward_and_obs_sentinels():
    # reference tests/test_reward.py:68-114,174-202,323-354 and tests/test_observation.py:122-149
    p = orc.EnvParams(M=2, N=4)
    cs = orc.ClusterState.empty(4)
    cs.cent_xy[0] = (50, 50); cs.cent_xy[1] = (150, 150); cs.ncent = 2; cs.n_clusters = 2
    cs.cent_size[:2] = 2
    jp = np.array([[50.0, 50.0], [150.0, 150.0]])
    none = np.full(2, orc.ASSIGN_NONE)
    v0 = np.zeros((2, 2))
    tot, c = orc.reward5(p, 5.0, 2.0, [2, 2], 2, jp, cs, v0, none, 10.0)
    assert c[0] == 0.0 and c[1] == 1.0 and abs(c[2] - 1.0) < 1e-12 and c[4] == 0.0
    tot, c = orc.reward5(p, 1.0, 0.0, [0, 1], 2, jp, cs, v0, none, 1000.0)
    assert c[0] == 0.0 and c[1] == 0.0 and c[4] == 1.0
    assert orc.reward5(orc.EnvParams(M=0, N=4), 1.0, 2.0, [], 2, np.zeros((0, 2)), cs,
                       np.zeros((0, 2)), none, 10.0)[0] == 0.0
    p1 = orc.EnvParams(M=1, N=4)
    obs = orc.build_observation(p1, jp[:1], np.array([2]), cs, 2, 10.0)
    assert obs.shape == (1, 5) and obs[0, 2] == 0.5 and obs[0, 4] == 1.0
    empty = orc.ClusterState.empty(4)
    obs = orc.build_observation(p1, np.array([[0.0, 0.0]]), np.array([1]), empty, 2, 10.0)
    assert abs(obs[0, 0] - math.hypot(100, 100) / 200) < 1e-6


def test_reward_per_agent():
    z = gu.load("per_agent.npz")
    for i in range(int(z["n_cases"])):
        g = lambda k: z[f"c{i}_{k}"]
        N, M = int(g("N")), int(g("M"))
        p = orc.EnvParams.fast_preset(N=N, M=M)
        cs = gu.cluster_state_from(N, g("cent_ids"), g("cent_xy"), g("cent_size"), g("ncent"))
        r = orc.reward_per_agent(p, float(g("lam")), float(g("lam0")), g("jb"), int(g("band")), g("jpos"), cs,
                                 g("vel"), g("assign"), float(g("R_jam")))
        np.testing.assert_allclose(r, g("rewards"), rtol=2e-6, atol=2e-7)


# ---------------------------------------------------------------- trajectories
@pytest.mark.parametrize("name", gu.TRAJ_NAMES + gu.CT_TRAJ_NAMES)
def test_trajectory(name):
    z = gu.load(f"traj_{name}.npz")
    p = gu.traj_params(z)
    resets = gu.traj_resets(z)
    st = orc.EnvState.empty(p)
    T = len(z["reward"])

    def do_reset(r):
        init = r["jpos"] if p.random_jammer_start else r["offsets"]
        obs, info = orc.env_reset(p, st, r["epos"], r["evel"], int(r["band"]), init, r["jb"], r.get("eomega"))
        assert np.array_equal(st.jammer_positions, r["jpos"])
        assert info["lambda2_initial"] == float(r["lam0"])
        assert st.R_jam == float(r["R_jam"])
        assert np.array_equal(st.clusters.labels.astype(np.int32), r["labels"])
        nc = int(r["ncent"])
        assert st.clusters.ncent == nc and st.clusters.n_clusters == int(r["n_clusters"])
        assert np.array_equal(st.clusters.cent_xy[:nc], r["cent_xy"][:nc])
        assert np.array_equal(st.assign.astype(np.int32), r["assign"])
        np.testing.assert_allclose(obs, r["obs"], rtol=1e-6, atol=1e-7)

    do_reset(resets[0])
    for t in range(T):
        obs, reward, term, trunc, info = orc.env_step(p, st, z["action"][t], z["noise"][t])
        assert info["step"] == int(z["step"][t])
        assert np.array_equal(st.enemy_pos, z["epos"][t]), t
        assert np.array_equal(st.enemy_vel, z["evel"][t]), t
        if p.enemy_mode == "coordinated_turn":
            assert np.array_equal(st.enemy_omega, z["eomega"][t]), t
        assert np.array_equal(st.jammer_positions, z["jpos"][t]), t
        assert np.array_equal(st.jammer_bands, z["jb"][t])
        assert np.array_equal(orc.pack_rows(info["adjacency"] | info["jammed"]), z["adj"][t])
        assert np.array_equal(orc.pack_rows(info["jammed"]), z["jam"][t])
        assert np.array_equal(st.clusters.labels.astype(np.int16), z["labels"][t])
        nc = int(z["ncent"][t])
        assert st.clusters.ncent == nc and info["n_clusters"] == int(z["n_clusters"][t])
        assert np.array_equal(st.clusters.cent_xy[:nc], z["cent_xy"][t][:nc])
        assert np.array_equal(st.clusters.cent_size[:nc], z["cent_size"][t][:nc])
        assert np.array_equal(st.assign.astype(np.int32), z["assign"][t]), t
        assert info["lambda2_current"] == float(z["lam"][t])
        np.testing.assert_allclose(reward, float(z["reward"][t]), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(info["reward_components"], z["comps"][t], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(obs, z["obs"][t], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(info["avg_jamming_power_dbm"], float(z["avgp"][t]), rtol=1e-12)
        assert trunc == bool(z["trunc"][t])
        if trunc:
            do_reset(resets[t + 1])


# ---------------------------------------------------------------- reset retry loop
def test_reset_retry_loop_matches_reference():
    """jammer_env.py:283-316 with attempts that fail the lambda2_initial test: the oracle's loop, fed
    the recorded draws of every attempt, ends in the reference's state (accepted attempt, cluster
    tables, previous centroids = those of the attempt before, jammers placed from the last fit)."""
    for p, seed, recs in gu.retry_cases():
        st = orc.EnvState.empty(p)
        for r in recs:
            n_att = int(r["n_attempts"])
            attempts = [(r["att_epos"][k], r["att_evel"][k], int(r["att_band"][k])) for k in range(n_att)]
            # more draws offered than the reference used: the loop must stop by itself
            attempts.append((r["att_epos"][0], r["att_evel"][0], int(r["att_band"][0])))
            init = r["jpos"] if p.random_jammer_start else r["offsets"]
            obs, info, used = orc.env_reset_with_retries(p, st, attempts, init, r["jb"])
            assert used == n_att
            assert info["lambda2_initial"] == float(r["lam0"])
            assert (info["lambda2_initial"] >= p.min_lambda2_initial) or used == p.max_reset_attempts
            assert np.array_equal(st.enemy_pos, r["epos"]) and st.enemy_band == int(r["band"])
            assert np.array_equal(st.clusters.labels.astype(np.int32), r["labels"])
            nc, pnc = int(r["ncent"]), int(r["prev_ncent"])
            assert st.clusters.ncent == nc and st.clusters.prev_ncent == pnc
            assert np.array_equal(st.clusters.cent_xy[:nc], r["cent_xy"][:nc])
            assert np.array_equal(st.clusters.prev_xy[:pnc], r["prev_xy"][:pnc])
            assert st.clusters.prev_ids() == [int(x) for x in r["prev_ids"][:pnc]]
            assert np.array_equal(st.assign.astype(np.int32), r["assign"])
            assert np.array_equal(st.jammer_positions, r["jpos"])
            np.testing.assert_allclose(obs, r["obs"], rtol=1e-6, atol=1e-7)


def test_lambda2_beyond_256_vertices_against_the_reference_branches():
    """tests/golden/lambda2_sparse.npz: the reference's dense and sparse branches of compute_lambda2
    (communication_graph.py:268-286) on graphs of 300 to 1 024 vertices.  The oracle follows the dense branch
    (the one the env takes); on connected graphs the sparse branch gives the same value."""
    z = gu.load("lambda2_sparse.npz")
    for k in range(int(z["n_graphs"])):
        n = int(z[f"g{k}_n"])
        A = orc.unpack_rows(z[f"g{k}_adj"], n)
        lam = orc.lambda2_of_laplacian(orc.laplacian(A))
        dense, sparse = float(z[f"g{k}_lam_dense"]), float(z[f"g{k}_lam_sparse"])
        assert abs(lam - dense) <= 1e-12 * max(1.0, abs(dense))
        if abs(dense) > 1e-9:
            assert abs(lam - sparse) <= 1e-9 * max(1.0, abs(sparse))
