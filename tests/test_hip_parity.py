"""Parity of the HIP path (through the C ABI) with the reference: golden vectors
produced by the reference itself, and the CPU oracle on seeded random inputs.

Bars: adjacency / jamming bits, DBSCAN labels, sizes, assignments, enemy and
jammer positions (float64 state): bit-exact.  lambda2: |d| <= 1e-5*max(|l2|,1)
(the reference returns +-1e-15 noise for disconnected graphs).  Rewards and
observations: 1e-5 relative (float32 outputs)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import golden_util as gu
from marl_amd import BatchedJammerEnv, EnvConfig, JammerEnv, _lib
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu

MAXN = 256   # N <= 64: one enemy per lane; 65..256: up to four per lane


def cfg_from(p: orc.EnvParams, **kw) -> EnvConfig:
    return EnvConfig(M=p.M, N=p.N, arena_size=p.arena_size, max_steps=p.max_steps, dt=p.dt, v_max=p.v_max,
                     v_enemy=p.v_enemy, P_tx_dbm=p.P_tx_dbm, P_sens_dbm=p.P_sens_dbm,
                     P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                     min_samples=p.min_samples, K_recompute=p.K_recompute,
                     random_jammer_start=p.random_jammer_start, enemy_mode=p.enemy_mode, omega_max=p.omega_max, **kw)


def dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def bits_np(t):
    return t.cpu().numpy().view(np.uint32)


def lam_ok(a, b):
    return abs(a - b) <= 1e-5 * max(abs(b), 1.0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def lambda2_bits(n, E, bits, rounds=0):
    """marl_lambda2_from_bits with its scratch buffer (needed for N > 64 only)."""
    lib = _lib.load()
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    nb = lib.marl_lambda2_workspace_bytes(n, E)
    ws = torch.zeros(max(nb, 8), dtype=torch.uint8, device="cuda")
    _lib.check(lib.marl_lambda2_from_bits(n, E, bits.data_ptr(), out.data_ptr(), rounds,
                                          ws.data_ptr() if nb else None, stream()))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------- stage ops vs reference vectors
def _stage_groups():
    groups = {}
    for p, d in gu.stage_cases():
        key = (p.N, p.M, p.arena_size, p.P_tx_dbm, p.P_sens_dbm, p.P_jammer_dbm, p.P_jam_thresh_dbm, p.eps,
               p.min_samples)
        groups.setdefault(key, (p, []))[1].append(d)
    return list(groups.values())


@pytest.mark.parametrize("gi", range(len(_stage_groups())))
def test_stage_ops_match_reference_vectors(gi):
    p, cases = _stage_groups()[gi]
    lib = _lib.load()
    P = cfg_from(p).to_params()
    E, N, M = len(cases), p.N, p.M
    W = (N + 31) // 32
    x = dev(np.stack([c["epos"][:, 0] for c in cases]), torch.float64)
    y = dev(np.stack([c["epos"][:, 1] for c in cases]), torch.float64)
    band = dev(np.array([int(c["band"]) for c in cases]), torch.int32)
    jx = dev(np.stack([c["jpos"][:, 0] for c in cases]), torch.float64)
    jy = dev(np.stack([c["jpos"][:, 1] for c in cases]), torch.float64)
    jb = dev(np.stack([c["jb"] for c in cases]), torch.int32)
    adj = torch.zeros(E, N, W, dtype=torch.int32, device="cuda")
    jam = torch.zeros(E, N, W, dtype=torch.int32, device="cuda")
    _lib.check(lib.marl_adjacency_bits(C.byref(P), E, x.data_ptr(), y.data_ptr(), band.data_ptr(),
                                       adj.data_ptr(), stream()))
    _lib.check(lib.marl_jam_mask_bits(C.byref(P), E, x.data_ptr(), y.data_ptr(), band.data_ptr(), jx.data_ptr(),
                                      jy.data_ptr(), jb.data_ptr(), jam.data_ptr(), stream()))
    eff = adj & ~jam
    lam = lambda2_bits(N, E, eff.contiguous())
    lam0 = lambda2_bits(N, E, adj)
    labels = torch.zeros(E, N, dtype=torch.int32, device="cuda")
    cx = torch.zeros(E, N + 1, dtype=torch.float64, device="cuda")
    cy = torch.zeros(E, N + 1, dtype=torch.float64, device="cuda")
    csz = torch.zeros(E, N + 1, dtype=torch.int32, device="cuda")
    ncl = torch.zeros(E, dtype=torch.int32, device="cuda")
    ncent = torch.zeros(E, dtype=torch.int32, device="cuda")
    _lib.check(lib.marl_dbscan(C.byref(P), E, x.data_ptr(), y.data_ptr(), labels.data_ptr(), cx.data_ptr(),
                               cy.data_ptr(), csz.data_ptr(), ncl.data_ptr(), ncent.data_ptr(), stream()))
    torch.cuda.synchronize()
    for e, c in enumerate(cases):
        assert np.array_equal(bits_np(adj[e]), c["adj"]), e
        assert np.array_equal(bits_np(jam[e]), c["jam"]), e
        assert lam_ok(float(lam[e]), float(c["lam"])), (e, float(lam[e]), float(c["lam"]))
        assert lam_ok(float(lam0[e]), float(c["lam0"]))
        assert np.array_equal(labels[e].cpu().numpy(), c["labels"]), e
        nc = int(c["ncent"])
        assert int(ncent[e]) == nc and int(ncl[e]) == int(c["n_clusters"])
        assert np.array_equal(cx[e, :nc].cpu().numpy(), c["cent_xy"][:nc, 0])
        assert np.array_equal(cy[e, :nc].cpu().numpy(), c["cent_xy"][:nc, 1])
        assert np.array_equal(csz[e, :nc].cpu().numpy(), c["cent_size"][:nc])


def test_lambda2_closed_forms_and_random_graphs():
    """K_n -> n, path, cycle, disconnected (reference tests/test_laplacian.py:251-310)
    and random graphs of every density against the reference's eigh values."""
    lib = _lib.load()
    z = gu.load("edges.npz")
    todo = []
    for n in (2, 3, 4, 5, 6, 10, 50):
        todo.append((~np.eye(n, dtype=bool), float(n)))
    for n in (5, 10, 50):
        A = np.zeros((n, n), dtype=bool)
        for i in range(n - 1):
            A[i, i + 1] = A[i + 1, i] = True
        todo.append((A.copy(), 2 * (1 - math.cos(math.pi / n))))
        A[0, n - 1] = A[n - 1, 0] = True
        todo.append((A.copy(), 2 * (1 - math.cos(2 * math.pi / n))))
    A = np.zeros((6, 6), dtype=bool)
    for i, j in ((0, 1), (1, 2), (3, 4), (4, 5)):
        A[i, j] = A[j, i] = True
    todo.append((A, 0.0))
    for k in range(int(z["n_graphs"])):
        n = int(z[f"g{k}_n"])
        if n <= MAXN:
            todo.append((orc.unpack_rows(z[f"g{k}_adj"], n), float(z[f"g{k}_lam"])))
    for A, want in todo:
        n = A.shape[0]
        bits = dev(orc.pack_rows(A).view(np.int32)[None], torch.int32)
        got = float(lambda2_bits(n, 1, bits)[0])
        assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (n, got, want)


def test_lambda2_accuracy_with_more_rounds():
    """With 9 multisection rounds the eigenvalue agrees with LAPACK to ~1e-11."""
    lib = _lib.load()
    rng = np.random.RandomState(3)
    n, E = 50, 64
    mats, want = [], []
    for e in range(E):
        A = np.triu(rng.rand(n, n) < rng.uniform(0.05, 0.95), 1)
        A = A | A.T
        mats.append(orc.pack_rows(A).view(np.int32))
        want.append(orc.lambda2_of_laplacian(orc.laplacian(A)))
    bits = dev(np.stack(mats), torch.int32)
    got = lambda2_bits(n, E, bits, rounds=9).cpu().numpy()
    assert np.max(np.abs(got - np.array(want))) < 1e-9


def test_lambda2_every_matrix_width():
    """Every matrix width of the register solver and every position of the last column inside a block of
    the unrolled column loops: n = 2..64, random graphs of several densities (most of them connected and
    with no vertex of full degree, so that the eigenproblem has n rows), against LAPACK."""
    lib = _lib.load()
    rng = np.random.RandomState(11)
    sizes = list(range(2, 65)) + [65, 66, 80, 96, 97, 112, 127, 128, 129, 130, 144, 160, 176, 192, 200, 207, 208,
                                  209, 224, 256]       # beyond 64: team solver (two widths) and workspace solver
    for n in sizes:
        mats, want = [], []
        for dens in (0.15, 0.3, 0.45, 0.6, 0.75, 0.9, 0.97):
            for _ in range(3 if n <= 64 else 1):
                A = np.triu(rng.rand(n, n) < dens, 1)
                A = A | A.T
                mats.append(orc.pack_rows(A).view(np.int32))
                want.append(orc.lambda2_of_laplacian(orc.laplacian(A)))
        bits = dev(np.stack(mats), torch.int32)
        got = lambda2_bits(n, len(mats), bits).cpu().numpy()
        want = np.array(want)
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 2e-9, (n, got, want)


def test_dbscan_edge_vectors():
    lib = _lib.load()
    z = gu.load("edges.npz")
    for name in z["db_names"]:
        pts = z[f"db_{name}_pts"]
        eps, ms = z[f"db_{name}_par"]
        n = len(pts)
        P = EnvConfig(N=n, M=2, eps=float(eps), min_samples=int(ms), arena_size=200.0).to_params()
        x, y = dev(pts[None, :, 0], torch.float64), dev(pts[None, :, 1], torch.float64)
        labels = torch.zeros(1, n, dtype=torch.int32, device="cuda")
        cx = torch.zeros(1, n + 1, dtype=torch.float64, device="cuda")
        cy = torch.zeros_like(cx)
        csz = torch.zeros(1, n + 1, dtype=torch.int32, device="cuda")
        ncl = torch.zeros(1, dtype=torch.int32, device="cuda")
        ncent = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lib.marl_dbscan(C.byref(P), 1, x.data_ptr(), y.data_ptr(), labels.data_ptr(), cx.data_ptr(),
                                   cy.data_ptr(), csz.data_ptr(), ncl.data_ptr(), ncent.data_ptr(), stream()))
        nc = int(z[f"db_{name}_ncent"])
        assert np.array_equal(labels[0].cpu().numpy(), z[f"db_{name}_labels"]), name
        assert int(ncent[0]) == nc, name
        assert np.array_equal(cx[0, :nc].cpu().numpy(), z[f"db_{name}_xy"][:nc, 0]), name
        assert np.array_equal(cy[0, :nc].cpu().numpy(), z[f"db_{name}_xy"][:nc, 1]), name
        assert np.array_equal(csz[0, :nc].cpu().numpy(), z[f"db_{name}_size"][:nc]), name


def test_knife_edge_thresholds():
    """Pairs within a few ulp of the comm radius / jam radius."""
    lib = _lib.load()
    z = gu.load("edges.npz")
    pos = z["knife_pos"]
    n = len(pos)
    P = EnvConfig(N=n, M=1, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0).to_params()
    x, y = dev(pos[None, :, 0], torch.float64), dev(pos[None, :, 1], torch.float64)
    band = dev([3], torch.int32)
    adj = torch.zeros(1, n, 1, dtype=torch.int32, device="cuda")
    _lib.check(lib.marl_adjacency_bits(C.byref(P), 1, x.data_ptr(), y.data_ptr(), band.data_ptr(),
                                       adj.data_ptr(), stream()))
    assert np.array_equal(bits_np(adj[0]), z["knife_adj"])
    ep = z["knife_jam_epos"]
    n = len(ep)
    P = EnvConfig(N=n, M=1, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0).to_params()
    x, y = dev(ep[None, :, 0], torch.float64), dev(ep[None, :, 1], torch.float64)
    jam = torch.zeros(1, n, 1, dtype=torch.int32, device="cuda")
    jx, jy, jb = dev([[0.0]], torch.float64), dev([[0.0]], torch.float64), dev([[3]], torch.int32)
    _lib.check(lib.marl_jam_mask_bits(C.byref(P), 1, x.data_ptr(), y.data_ptr(), band.data_ptr(), jx.data_ptr(),
                                      jy.data_ptr(), jb.data_ptr(), jam.data_ptr(), stream()))
    assert np.array_equal(bits_np(jam[0]), z["knife_jam"])
    # reference example: jammer on the midpoint, right / wrong band, far away (tests/test_jamming.py:174-231)
    ep = z["jam_epos"]
    P = EnvConfig(N=3, M=1, P_jammer_dbm=20.0, P_jam_thresh_dbm=-40.0).to_params()
    x, y = dev(ep[None, :, 0], torch.float64), dev(ep[None, :, 1], torch.float64)
    band = dev([2], torch.int32)
    for name, jp, b in (("on_mid", (50.0, 50.0), 2), ("wrong_band", (50.0, 50.0), 1), ("far", (50.0, 2050.0), 2)):
        jam = torch.zeros(1, 3, 1, dtype=torch.int32, device="cuda")
        jx, jy, jb = dev([[jp[0]]], torch.float64), dev([[jp[1]]], torch.float64), dev([[b]], torch.int32)
        _lib.check(lib.marl_jam_mask_bits(C.byref(P), 1, x.data_ptr(), y.data_ptr(), band.data_ptr(),
                                          jx.data_ptr(), jy.data_ptr(), jb.data_ptr(), jam.data_ptr(), stream()))
        assert np.array_equal(bits_np(jam[0]), z[f"jam_{name}"]), name


def test_gae_matches_reference_buffer():
    lib = _lib.load()
    z = gu.load("edges.npz")
    T, E = 64, 5
    rew = dev(np.repeat(z["gae_rew"][:, None], E, 1), torch.float32)
    val = dev(np.repeat(z["gae_val"][:, None], E, 1), torch.float32)
    don = dev(np.repeat(z["gae_done"][:, None], E, 1), torch.float32)
    lv = dev(np.full(E, float(z["gae_last_value"])), torch.float32)
    ld = dev(np.zeros(E), torch.float32)
    adv = torch.zeros(T, E, device="cuda")
    ret = torch.zeros(T, E, device="cuda")
    _lib.check(lib.marl_gae(T, E, rew.data_ptr(), val.data_ptr(), don.data_ptr(), lv.data_ptr(), ld.data_ptr(),
                            0.99, 0.95, adv.data_ptr(), ret.data_ptr(), stream()))
    for e in range(E):
        assert np.array_equal(ret[:, e].cpu().numpy(), z["gae_ret"])
        a = adv[:, e].cpu().numpy()
        np.testing.assert_allclose((a - a.mean()) / (a.std() + 1e-8), z["gae_adv_norm"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- whole trajectories of the reference env
@pytest.mark.parametrize("schedule", ["fused", "split"])
@pytest.mark.parametrize("name", gu.TRAJ_NAMES + gu.CT_TRAJ_NAMES)
def test_trajectory_matches_reference(name, schedule):
    """Random walk: the whole trajectory is bit-identical to the reference's.  Coordinated turn: the
    model calls sin / cos, which the device library and numpy round differently in the last bit, so
    positions, velocities, turn rates and centroids are compared to 1e-12 relative (+1e-10 absolute)
    there; the integer and boolean outputs stay exact."""
    z = gu.load(f"traj_{name}.npz")
    p = gu.traj_params(z)
    resets = gu.traj_resets(z)
    ct = p.enemy_mode == "coordinated_turn"

    def same(a, b):
        if ct:
            return np.allclose(a, b, rtol=1e-12, atol=1e-10)
        return np.array_equal(a, b)
    E = 3      # the same trajectory in every env of a small batch
    if p.N > 64 and schedule == "split":
        pytest.skip("the schedules only differ for N <= 64")
    env = BatchedJammerEnv(E, cfg_from(p, step_schedule=schedule), auto_reset=False, track_avg_power=True,
                           debug_bits=True)
    rep = lambda a: np.repeat(np.asarray(a)[None], E, 0)

    def do_reset(r):
        init = r["jpos"] if p.random_jammer_start else r["offsets"]
        obs, info = env.reset(enemy_pos=rep(r["epos"]), enemy_vel=rep(r["evel"]),
                              enemy_band=rep(np.int32(r["band"])), jam_init=rep(init), jam_band=rep(r["jb"]),
                              enemy_omega=rep(r["eomega"]) if ct else None)
        torch.cuda.synchronize()
        v = env.views
        for e in range(E):
            assert lam_ok(float(v["lambda2_initial"][e]), float(r["lam0"]))
            assert np.array_equal(v["jam_x"][e].cpu().numpy(), r["jpos"][:, 0])
            assert np.array_equal(v["jam_y"][e].cpu().numpy(), r["jpos"][:, 1])
            assert np.array_equal(v["labels"][e].cpu().numpy(), r["labels"])
            assert np.array_equal(v["assign"][e].cpu().numpy(), r["assign"])
            nc = int(r["ncent"])
            assert int(v["ncent"][e]) == nc and int(v["n_clusters"][e]) == int(r["n_clusters"])
            assert np.array_equal(v["cent_x"][e, :nc].cpu().numpy(), r["cent_xy"][:nc, 0])
            np.testing.assert_allclose(obs[e].cpu().numpy(), r["obs"], rtol=1e-5, atol=1e-6)

    do_reset(resets[0])
    T = len(z["reward"])
    omega = np.array(p.omega) * np.array([1, 1, 1, -1, -1])
    for t in range(T):
        obs, rew, term, trunc, info = env.step(dev(rep(z["action"][t]), torch.float32),
                                               dev(rep(z["noise"][t]), torch.float64))
        torch.cuda.synchronize()
        v = env.views
        for e in (0, E - 1):
            assert int(v["step_count"][e]) == int(z["step"][t])
            assert same(v["enemy_x"][e].cpu().numpy(), z["epos"][t][:, 0]), t
            assert same(v["enemy_y"][e].cpu().numpy(), z["epos"][t][:, 1]), t
            assert same(v["enemy_vx"][e].cpu().numpy(), z["evel"][t][:, 0]), t
            if ct:
                assert same(v["enemy_omega"][e].cpu().numpy(), z["eomega"][t]), t
            assert np.array_equal(v["jam_x"][e].cpu().numpy(), z["jpos"][t][:, 0]), t
            assert np.array_equal(v["jam_band"][e].cpu().numpy(), z["jb"][t])
            assert np.array_equal(bits_np(env.adj_bits[e]), z["adj"][t]), t
            assert np.array_equal(bits_np(env.jam_bits[e]), z["jam"][t]), t
            assert np.array_equal(v["labels"][e].cpu().numpy().astype(np.int16), z["labels"][t]), t
            nc = int(z["ncent"][t])
            assert int(v["ncent"][e]) == nc and int(v["n_clusters"][e]) == int(z["n_clusters"][t])
            assert same(v["cent_x"][e, :nc].cpu().numpy(), z["cent_xy"][t][:nc, 0]), t
            assert np.array_equal(v["cent_size"][e, :nc].cpu().numpy(), z["cent_size"][t][:nc])
            assert np.array_equal(v["assign"][e].cpu().numpy(), z["assign"][t]), t
            assert lam_ok(float(info["lambda2_current"][e]), float(z["lam"][t])), t
            assert abs(float(rew[e]) - float(z["reward"][t])) <= 1e-5 * max(1.0, abs(float(z["reward"][t]))), t
            np.testing.assert_allclose(info["reward_terms"][e].cpu().numpy().astype(np.float64) * omega,
                                       z["comps"][t], rtol=1e-5, atol=2e-6)
            np.testing.assert_allclose(obs[e].cpu().numpy(), z["obs"][t], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(float(info["avg_jamming_power_dbm"][e]), float(z["avgp"][t]), rtol=1e-9)
            assert bool(trunc[e]) == bool(z["trunc"][t])
        if bool(z["trunc"][t]):
            do_reset(resets[t + 1])


# ---------------------------------------------------------------- batched run vs the oracle on seeded inputs
@pytest.mark.parametrize("N,M,E,steps", [(10, 4, 48, 25), (50, 8, 24, 21), (64, 16, 6, 11), (3, 1, 8, 12),
                                         (65, 8, 4, 11), (130, 16, 3, 11), (200, 16, 3, 6), (256, 64, 2, 5)])
def test_batched_steps_match_oracle(N, M, E, steps):
    rng = np.random.RandomState(100 + N)
    p = orc.EnvParams.fast_preset(N=N, M=M)
    p.max_steps = 15
    env = BatchedJammerEnv(E, cfg_from(p), auto_reset=False, debug_bits=True)
    A = p.arena_size

    def draw_reset(mask):
        epos = rng.uniform(0.1 * A, 0.9 * A, size=(E, N, 2))
        ang = rng.uniform(0, 2 * np.pi, size=(E, N))
        evel = np.stack([2 * np.cos(ang), 2 * np.sin(ang)], axis=2)
        eband = rng.randint(0, 4, size=E).astype(np.int32)
        jinit = rng.uniform(0.1 * A, 0.9 * A, size=(E, M, 2))
        jb = rng.randint(0, 4, size=(E, M)).astype(np.int32)
        obs, _ = env.reset(mask=None if mask is None else torch.from_numpy(mask.astype(np.uint8)),
                           enemy_pos=epos, enemy_vel=evel, enemy_band=eband, jam_init=jinit, jam_band=jb)
        torch.cuda.synchronize()
        for e in range(E):
            if mask is None or mask[e]:
                o, _ = orc.env_reset(p, states[e], epos[e], evel[e], eband[e], jinit[e], jb[e])
                np.testing.assert_allclose(obs[e].cpu().numpy(), o, rtol=1e-5, atol=1e-6)

    states = [orc.EnvState.empty(p) for _ in range(E)]
    draw_reset(None)
    for t in range(steps):
        act = rng.uniform(-6, 6, size=(E, M, 3)).astype(np.float32)
        act[:, :, 2] = rng.uniform(-1, 5, size=(E, M))
        # half of the envs steer their jammers onto enemy link midpoints of the enemy band
        for e in range(0, E, 2):
            st = states[e]
            for j in range(M):
                i1, i2 = rng.randint(0, N, size=2)
                tgt = (st.enemy_pos[i1] + st.enemy_pos[i2]) / 2
                act[e, j, :2] = np.clip((tgt - st.jammer_positions[j]) / p.dt, -7, 7)
                act[e, j, 2] = st.enemy_band + 0.5
        noise = rng.normal(0, 0.1, size=(E, N, 2))
        obs, rew, _, trunc, info = env.step(torch.from_numpy(act).cuda(), torch.from_numpy(noise).cuda())
        torch.cuda.synchronize()
        v = env.views
        tr = trunc.cpu().numpy()
        for e in range(E):
            o, r, _, tru, inf = orc.env_step(p, states[e], act[e], noise[e], want_avg_power=False)
            st = states[e]
            assert np.array_equal(v["enemy_x"][e].cpu().numpy(), st.enemy_pos[:, 0])
            assert np.array_equal(v["enemy_vy"][e].cpu().numpy(), st.enemy_vel[:, 1])
            assert np.array_equal(v["jam_y"][e].cpu().numpy(), st.jammer_positions[:, 1])
            assert np.array_equal(bits_np(env.adj_bits[e]), orc.pack_rows(inf["adjacency"] | inf["jammed"]))
            assert np.array_equal(bits_np(env.jam_bits[e]), orc.pack_rows(inf["jammed"]))
            assert np.array_equal(v["labels"][e].cpu().numpy(), st.clusters.labels)
            assert np.array_equal(v["assign"][e].cpu().numpy(), st.assign)
            assert lam_ok(float(info["lambda2_current"][e]), inf["lambda2_current"])
            assert abs(float(rew[e]) - r) <= 1e-5 * max(1.0, abs(r))
            np.testing.assert_allclose(obs[e].cpu().numpy(), o, rtol=1e-5, atol=1e-6)
            assert bool(tr[e]) == tru
        if tr.any():
            draw_reset(tr)


def test_generator_mode_matches_noise_fed_mode():
    """noise=None draws the same enemy noise that marl_fill_noise writes out."""
    cfg = EnvConfig.fast_preset(N=50, M=8)
    a = BatchedJammerEnv(16, cfg, seed=5, auto_reset=False)
    b = BatchedJammerEnv(16, cfg, seed=5, auto_reset=False)
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(12):
        act = (torch.rand(16, 8, 3, device="cuda", generator=g) - 0.5) * 10
        noise = b.fill_noise()
        oa, ra, _, _, ia = a.step(act)
        ob, rb, _, _, ib = b.step(act, noise)
        assert torch.equal(oa, ob) and torch.equal(ra, rb)
        assert torch.equal(a.views["enemy_x"], b.views["enemy_x"])
    # statistics of the generator: N(0, 0.1)
    big = BatchedJammerEnv(2048, cfg, seed=9, auto_reset=False)
    big.reset()
    n = big.fill_noise()
    assert abs(float(n.mean())) < 2e-3 and abs(float(n.std()) - 0.1) < 2e-3


def test_on_device_reset_and_auto_reset_invariants():
    """Generator-mode reset: positions inside [0.1A, 0.9A], unit-speed headings,
    lambda2_initial = N for the complete graph, and auto-reset restarts step counts
    while reporting the terminal step's metrics."""
    cfg = EnvConfig.fast_preset(N=50, M=8, max_steps=7)
    E = 256
    env = BatchedJammerEnv(E, cfg, seed=3, auto_reset=True)
    obs, info = env.reset()
    v = env.views
    A = cfg.arena_size
    for k in ("enemy_x", "enemy_y", "jam_x", "jam_y"):
        assert float(v[k].min()) >= 0.1 * A and float(v[k].max()) <= 0.9 * A
    sp = torch.sqrt(v["enemy_vx"] ** 2 + v["enemy_vy"] ** 2)
    assert torch.allclose(sp, torch.full_like(sp, cfg.v_enemy), rtol=1e-12, atol=0)
    assert torch.all(v["lambda2_initial"] == 50.0)
    assert set(v["enemy_band"].unique().cpu().numpy().tolist()) == {0, 1, 2, 3}
    assert float(obs.min()) >= 0.0 and float(obs.max()) <= 1.0
    x0 = v["enemy_x"].clone()
    for t in range(1, 15):
        act = (torch.rand(E, 8, 3, device="cuda") - 0.5) * 10
        obs, rew, term, trunc, info = env.step(act)
        assert not bool(term.any())
        if t % 7 == 0:
            assert bool(trunc.all())
            assert torch.all(info["episode_step"] == 7)
            assert torch.all(v["step_count"] == 0)
            assert not torch.equal(v["enemy_x"], x0)      # fresh draw per episode
            x0 = v["enemy_x"].clone()
        else:
            assert not bool(trunc.any())
            assert torch.all(v["step_count"] == t % 7)
        assert float(rew.min()) >= -10 and float(rew.max()) <= 10
        assert float(obs.min()) >= 0.0 and float(obs.max()) <= 1.0


def test_batch_of_one_equals_batch_of_many():
    cfg = EnvConfig.fast_preset(N=50, M=8)
    big = BatchedJammerEnv(64, cfg, seed=11, auto_reset=False)
    big.reset()
    one = BatchedJammerEnv(1, cfg, seed=11, auto_reset=False, env_offset=37)
    one.reset()
    for k in ("enemy_x", "jam_x", "enemy_band"):
        assert torch.equal(one.views[k][0], big.views[k][37])
    for t in range(10):
        act = (torch.rand(64, 8, 3, device="cuda") - 0.5) * 10
        ob, rb, _, _, ib = big.step(act)
        oo, ro, _, _, io = one.step(act[37:38].contiguous())
        assert torch.equal(oo[0], ob[37]) and torch.equal(ro[0], rb[37])
        assert float(io["lambda2_current"][0]) == float(ib["lambda2_current"][37])


# ---------------------------------------------------------------- the reference-API facade
def test_facade_reproduces_reference_trajectory_from_seed():
    """JammerEnv(seed).reset(seed) + the recorded actions give the reference's
    observations and rewards: same RandomState consumption as the reference."""
    z = gu.load("traj_n10m4_random.npz")
    p = gu.traj_params(z)
    np.random.seed(42 + 1000)
    env = JammerEnv(M=p.M, N=p.N, seed=42, arena_size=p.arena_size, max_steps=p.max_steps,
                    P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                    random_jammer_start=True)
    resets = gu.traj_resets(z)
    obs, info = env.reset(seed=42)
    np.testing.assert_allclose(obs, resets[0]["obs"], rtol=1e-5, atol=1e-6)
    assert info["enemy_band"] == int(resets[0]["band"]) and info["step"] == 0
    assert lam_ok(info["lambda2_initial"], float(resets[0]["lam0"]))
    for t in range(260):
        obs, reward, term, trunc, info = env.step(z["action"][t])
        assert obs.dtype == np.float32 and obs.shape == (p.M, 5)
        np.testing.assert_allclose(obs, z["obs"][t], rtol=1e-5, atol=1e-6)
        assert abs(reward - float(z["reward"][t])) <= 1e-5 * max(1.0, abs(float(z["reward"][t])))
        assert np.array_equal(info["enemy_positions"], z["epos"][t])
        assert np.array_equal(info["jammer_positions"], z["jpos"][t])
        assert term is False and trunc == bool(z["trunc"][t])
        assert set(info) >= {"lambda2_initial", "lambda2_current", "lambda2_reduction", "lambda2_reduction_raw",
                             "enemy_band", "n_clusters", "step", "reward_components", "jammer_positions",
                             "enemy_positions", "avg_jamming_power_dbm"}
        if trunc:
            obs, info = env.reset()
            np.testing.assert_allclose(obs, resets[t + 1]["obs"], rtol=1e-5, atol=1e-6)


def test_facade_centroid_start_and_api_contract():
    """random_jammer_start=False path (global-RNG offsets) and the step contract the
    reference's tests/test_jammer_env.py:181-250,294-306 pin."""
    z = gu.load("traj_n20m4_centroid.npz")
    p = gu.traj_params(z)
    np.random.seed(5 + 1000)
    env = JammerEnv(M=p.M, N=p.N, seed=5, arena_size=p.arena_size, max_steps=p.max_steps,
                    P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                    random_jammer_start=False)
    resets = gu.traj_resets(z)
    obs, info = env.reset(seed=5)
    assert np.array_equal(env.jammer_positions, resets[0]["jpos"])
    np.testing.assert_allclose(obs, resets[0]["obs"], rtol=1e-5, atol=1e-6)
    for t in range(70):
        obs, reward, term, trunc, info = env.step(z["action"][t])
        np.testing.assert_allclose(obs, z["obs"][t], rtol=1e-5, atol=1e-6)
        if trunc:
            obs, info = env.reset()
            assert np.array_equal(env.jammer_positions, resets[t + 1]["jpos"])
    # pos += v*dt; bands from the action; dict actions; settable jammer_positions
    env = JammerEnv(M=2, N=5, seed=1, max_steps=5)
    env.reset(seed=1)
    env.jammer_positions = np.array([[100.0, 100.0], [50.0, 50.0]])
    obs, r, term, trunc, info = env.step({"velocity": np.array([[2.0, 0.0], [0.0, -2.0]], dtype=np.float32),
                                          "band": np.array([1, 3])})
    assert np.allclose(env.jammer_positions, [[101.0, 100.0], [50.0, 49.0]])
    assert list(env.jammer_bands) == [1, 3]
    assert obs.min() >= 0 and obs.max() <= 1
    for _ in range(4):
        obs, r, term, trunc, info = env.step(np.zeros((2, 3), dtype=np.float32))
    assert trunc and not term and info["step"] == 5


# ---------------------------------------------------------------- coordinated-turn enemy model
@pytest.mark.parametrize("N,M,E", [(12, 3, 16), (50, 8, 8), (130, 16, 3)])
def test_coordinated_turn_steps_match_oracle(N, M, E):
    """enemy_swarm.py:191-242 against the oracle from host-fed draws, with the corner cases planted:
    turn rates of exactly 0 and below 1e-6 (linear branch), at +-omega_max (clip), drones about to
    leave the arena on every side (np.mod wrap).  Tolerance as in the trajectory test: sin / cos."""
    rng = np.random.RandomState(77 + N)
    p = orc.EnvParams.fast_preset(N=N, M=M)
    p.enemy_mode, p.max_steps = "coordinated_turn", 1000
    A = p.arena_size
    env = BatchedJammerEnv(E, cfg_from(p), auto_reset=False, debug_bits=True)
    epos = rng.uniform(0.1 * A, 0.9 * A, size=(E, N, 2))
    ang = rng.uniform(0, 2 * np.pi, size=(E, N))
    evel = np.stack([2 * np.cos(ang), 2 * np.sin(ang)], axis=2)
    eom = rng.uniform(-0.25, 0.25, size=(E, N))
    eom[:, 0], eom[:, 1], eom[:, 2], eom[:, 3] = 0.0, 5e-7, 0.5, -0.5
    epos[:, 4] = (0.3, 100.0); evel[:, 4] = (-2.0, 0.0)          # leaves through x = 0
    epos[:, 5] = (A - 0.3, 50.0); evel[:, 5] = (2.0, 0.0)        # through x = A
    epos[:, 6] = (20.0, 0.2); evel[:, 6] = (0.0, -2.0)           # through y = 0
    epos[:, 7] = (30.0, A - 0.2); evel[:, 7] = (0.0, 2.0)        # through y = A
    eom[:, 4:8] = 0.0
    eband = rng.randint(0, 4, size=E).astype(np.int32)
    jinit = rng.uniform(0.1 * A, 0.9 * A, size=(E, M, 2))
    jb = rng.randint(0, 4, size=(E, M)).astype(np.int32)
    env.reset(enemy_pos=epos, enemy_vel=evel, enemy_band=eband, jam_init=jinit, jam_band=jb, enemy_omega=eom)
    states = [orc.EnvState.empty(p) for _ in range(E)]
    for e in range(E):
        orc.env_reset(p, states[e], epos[e], evel[e], eband[e], jinit[e], jb[e], eom[e])
    close = lambda a, b: np.allclose(a, b, rtol=1e-12, atol=1e-10)
    for t in range(12):
        act = rng.uniform(-6, 6, size=(E, M, 3)).astype(np.float32)
        act[:, :, 2] = rng.uniform(-1, 5, size=(E, M))
        noise = np.zeros((E, N, 2))
        noise[:, :, 0] = rng.normal(0, 0.05, size=(E, N))
        obs, rew, _, trunc, info = env.step(torch.from_numpy(act).cuda(), torch.from_numpy(noise).cuda())
        torch.cuda.synchronize()
        v = env.views
        for e in range(E):
            o, r, _, _, inf = orc.env_step(p, states[e], act[e], noise[e], want_avg_power=False)
            st = states[e]
            x = v["enemy_x"][e].cpu().numpy()
            assert close(x, st.enemy_pos[:, 0]) and close(v["enemy_y"][e].cpu().numpy(), st.enemy_pos[:, 1]), (t, e)
            assert close(v["enemy_vx"][e].cpu().numpy(), st.enemy_vel[:, 0])
            assert close(v["enemy_omega"][e].cpu().numpy(), st.enemy_omega)
            assert x.min() >= 0.0 and x.max() <= A
            assert np.array_equal(bits_np(env.adj_bits[e]), orc.pack_rows(inf["adjacency"] | inf["jammed"]))
            assert np.array_equal(bits_np(env.jam_bits[e]), orc.pack_rows(inf["jammed"]))
            assert np.array_equal(v["labels"][e].cpu().numpy(), st.clusters.labels)
            assert lam_ok(float(info["lambda2_current"][e]), inf["lambda2_current"])
            assert abs(float(rew[e]) - r) <= 1e-5 * max(1.0, abs(r))
            np.testing.assert_allclose(obs[e].cpu().numpy(), o, rtol=1e-5, atol=1e-6)
    # the wrapped drones really crossed
    assert float(v["enemy_x"][0, 4]) > A / 2 and float(v["enemy_x"][0, 5]) < A / 2


def test_coordinated_turn_generator_mode():
    """Device-drawn turn rates lie in [-omega_max/2, omega_max/2] (enemy_swarm.py:133-137); a step with
    the device generator equals a step fed with marl_fill_noise's output; the noise is N(0, 0.05)."""
    cfg = EnvConfig.fast_preset(N=50, M=8, enemy_mode="coordinated_turn")
    a = BatchedJammerEnv(64, cfg, seed=5, auto_reset=False)
    b = BatchedJammerEnv(64, cfg, seed=5, auto_reset=False)
    a.reset(); b.reset()
    w = a.views["enemy_omega"]
    assert float(w.min()) >= -0.25 and float(w.max()) <= 0.25 and float(w.std()) > 0.1
    assert torch.equal(w, b.views["enemy_omega"])
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(8):
        act = (torch.rand(64, 8, 3, device="cuda", generator=g) - 0.5) * 10
        noise = b.fill_noise()
        oa, ra, _, _, _ = a.step(act)
        ob, rb, _, _, _ = b.step(act, noise)
        assert torch.equal(oa, ob) and torch.equal(ra, rb)
        assert torch.equal(a.views["enemy_x"], b.views["enemy_x"])
        assert torch.equal(a.views["enemy_omega"], b.views["enemy_omega"])
    assert abs(float(noise[:, :, 0].std()) - 0.05) < 3e-3
    assert float(a.views["enemy_omega"].abs().max()) <= 0.5


def test_facade_coordinated_turn_reproduces_reference_from_seed():
    """JammerEnv(enemy_mode='coordinated_turn', seed=s): the facade consumes numpy's RandomState as the
    reference does (one extra uniform(N) for the turn rates at reset, N scalar normals per step)."""
    z = gu.load("traj_n10m4_ct.npz")
    p = gu.traj_params(z)
    np.random.seed(44 + 1000)
    env = JammerEnv(M=p.M, N=p.N, seed=44, arena_size=p.arena_size, max_steps=p.max_steps,
                    P_jammer_dbm=p.P_jammer_dbm, P_jam_thresh_dbm=p.P_jam_thresh_dbm, eps=p.eps,
                    random_jammer_start=p.random_jammer_start, enemy_mode="coordinated_turn")
    obs, info = env.reset(seed=44)
    r0 = gu.traj_resets(z)[0]
    assert lam_ok(info["lambda2_initial"], float(r0["lam0"]))
    np.testing.assert_allclose(obs, r0["obs"], rtol=1e-5, atol=1e-6)
    for t in range(40):
        obs, rew, term, trunc, info = env.step(z["action"][t])
        np.testing.assert_allclose(info["enemy_positions"], z["epos"][t], rtol=1e-12, atol=1e-10)
        assert lam_ok(info["lambda2_current"], float(z["lam"][t])), t
        assert abs(rew - float(z["reward"][t])) <= 1e-5 * max(1.0, abs(float(z["reward"][t])))


# ---------------------------------------------------------------- full benchmark size: properties
def _unpack_bits(bits, N):
    """[E,N,W] int32 -> bool [E,N,N] on the device."""
    w = bits.to(torch.int64) & 0xFFFFFFFF
    j = torch.arange(N, device=bits.device)
    return ((w[:, :, j // 32] >> (j % 32)) & 1).bool()


@pytest.mark.parametrize("mode,E,N,M", [("random_walk", 8192, 50, 8), ("coordinated_turn", 8192, 50, 8),
                                        ("random_walk", 1024, 200, 16), ("random_walk", 512, 100, 12)])
def test_full_size_step_properties(mode, E, N, M):
    """8 192 envs x (N=50, M=8), the benchmark's shard (and N = 200 / 100 batches for the two-launch
    path of the large swarms), device generator: size-independent properties of every step output, and
    lambda2 against a batched float64 eigvalsh of the Laplacians rebuilt from the kernel's own
    adjacency / mask bits."""
    cfg = EnvConfig.fast_preset(N=N, M=M, enemy_mode=mode, max_steps=6)
    env = BatchedJammerEnv(E, cfg, seed=11, auto_reset=False, debug_bits=True)
    twin = BatchedJammerEnv(E, cfg, seed=11, auto_reset=False)
    env.reset(); twin.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    eye = torch.eye(N, dtype=torch.bool, device="cuda")
    seen_complete = seen_disconnected = seen_solved = 0
    for t in range(6):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 12
        act[:, :, 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        obs, rew, term, trunc, info = env.step(act)
        obs2, rew2, _, _, info2 = twin.step(act)
        lam = info["lambda2_current"]
        # same seed, same inputs -> same outputs, whatever order the solves ran in
        assert torch.equal(obs, obs2) and torch.equal(rew, rew2) and torch.equal(lam, info2["lambda2_current"])
        adj, jam = _unpack_bits(env.adj_bits, N), _unpack_bits(env.jam_bits, N)
        assert torch.equal(adj, adj.transpose(1, 2)) and torch.equal(jam, jam.transpose(1, 2))
        assert not (adj & eye).any() and not (jam & eye).any()
        eff = (adj & ~jam).double()
        deg = eff.sum(2)
        L = torch.diag_embed(deg) - eff
        ref = torch.linalg.eigvalsh(L)[:, 1]
        assert float(((lam - ref).abs() / ref.abs().clamp(min=1.0)).max()) < 1e-9
        assert float(lam.min()) >= 0.0 and float(lam.max()) <= N
        assert bool((lam <= deg.min(1).values * N / (N - 1) + 1e-9).all())        # Fiedler's bound
        complete = deg.min(1).values == N - 1
        assert bool((lam[complete] == N).all())
        reach = (eff + eye.double())
        for _ in range(6):
            reach = (reach @ reach).clamp(max=1.0)
        connected = reach.min(2).values.min(1).values > 0
        assert bool((lam[~connected] == 0.0).all()) and bool((lam[connected] > 1e-9).all())
        seen_complete += int(complete.sum()); seen_disconnected += int((~connected).sum())
        seen_solved += int((connected & ~complete).sum())
        assert float(rew.min()) >= -10.0 and float(rew.max()) <= 10.0
        assert float(obs.min()) >= 0.0 and float(obs.max()) <= 1.0
        assert bool((trunc == (t == 5)).all()) and not term.any()
        x, y = env.views["enemy_x"], env.views["enemy_y"]
        assert float(x.min()) >= 0 and float(x.max()) <= 200 and float(y.min()) >= 0 and float(y.max()) <= 200
        sp = (env.views["enemy_vx"] ** 2 + env.views["enemy_vy"] ** 2).sqrt()
        assert float((sp - 2.0).abs().max()) < 1e-9                               # both models keep the speed
        if N <= 64:
            assert int(env.views["solve_hist"][:1280].abs().sum()) == 0           # hand-over histograms left clean
    assert seen_solved > 0                                                        # the eigen-solver ran
    if N <= 64:
        assert seen_complete > 0


def test_generated_noise_does_not_repeat_across_episodes():
    """The device generator is keyed by the env's step clock, which a reset does not rewind: step k of
    one episode and step k of the next draw different noise, and so do two resets of the same env even
    when the host passes the same episode number (a replayed HIP graph does exactly that)."""
    cfg = EnvConfig.fast_preset(N=20, M=4, max_steps=5)
    env = BatchedJammerEnv(32, cfg, seed=2, auto_reset=False)
    env.reset()
    first = env.fill_noise().clone()
    pos0 = env.views["enemy_x"].clone()
    act = torch.zeros(32, 4, 3, device="cuda")
    for _ in range(5):
        env.step(act)
    assert int(env.views["tick"].min()) == 5 and int(env.views["step_count"].max()) == 5
    env._episode = 0                               # same host-side episode number as the first reset
    env.reset()
    assert int(env.views["step_count"].max()) == 0 and int(env.views["tick"].min()) == 5
    assert not torch.equal(env.views["enemy_x"], pos0)
    again = env.fill_noise()
    assert float((again - first).abs().min()) > 0.0


def test_state_buffer_need_not_be_zeroed_before_the_first_full_reset():
    """A full reset initialises everything a step reads, hand-over histograms included: a state carved
    out of a buffer full of garbage gives the same trajectory as one carved out of zeros."""
    cfg = EnvConfig.fast_preset(N=50, M=8)
    a = BatchedJammerEnv(64, cfg, seed=4, auto_reset=False)
    b = BatchedJammerEnv(64, cfg, seed=4, auto_reset=False)
    b._buf.fill_(0x5A)
    b.views["tick"].zero_()                       # the step clock keys the generator: same start for both
    a.reset(); b.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(12):
        act = (torch.rand(64, 8, 3, device="cuda", generator=g) - 0.5) * 10
        oa, ra, _, _, ia = a.step(act)
        ob, rb, _, _, ib = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb)
        assert torch.equal(ia["lambda2_current"], ib["lambda2_current"])


def test_split_schedules_equal_fused_schedule_over_episodes():
    """The three-launch step (histogram, counting sort, longest-first solves, hand-over buffers) must give
    bit for bit what the one-launch step gives: 8 191 envs (on 256 CUs a batch for which the solve launch sets
    wave priorities), 130 steps across two auto-resets, with the optional outputs."""
    E, N, M = 8191, 50, 8
    # the split schedule twice: with the solves' issue priority following their remaining work (the default), and without
    envs = [BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M, max_steps=60, step_schedule=s, solve_priority=pr), seed=21,
                             track_avg_power=True, debug_bits=True)
            for s, pr in (("fused", True), ("split", True), ("split", False))]
    for env in envs:
        env.reset()
    g = torch.Generator(device="cuda").manual_seed(9)
    for t in range(130):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 11
        act[:, :, 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        outs = [env.step(act) for env in envs]
        of, rf, _, tf, inf_ = outs[0]
        for os_, rs, _, ts, ins in outs[1:]:
            assert torch.equal(of, os_) and torch.equal(rf, rs) and torch.equal(tf, ts), t
            assert torch.equal(inf_["lambda2_current"], ins["lambda2_current"]), t
            assert torch.equal(inf_["reward_terms"], ins["reward_terms"]), t
            assert torch.equal(inf_["avg_jamming_power_dbm"], ins["avg_jamming_power_dbm"]), t
        for env in envs[1:]:
            assert torch.equal(envs[0].adj_bits, env.adj_bits) and torch.equal(envs[0].jam_bits, env.jam_bits), t
    for env in envs[1:]:
        assert torch.equal(envs[0].views["enemy_x"], env.views["enemy_x"])
        assert torch.equal(envs[0].views["labels"], env.views["labels"])
        assert int(env.views["tick"].min()) == 130


def test_reward_per_agent_matches_reference_vectors():
    """marl_reward_per_agent against RewardCalculator.compute_per_agent outputs (tests/golden/per_agent.npz)."""
    import ctypes as C
    from marl_amd import _lib
    lib = _lib.load()
    z = gu.load("per_agent.npz")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(int(z["n_cases"])):
        g = lambda k: z[f"c{i}_{k}"]
        N, M = int(g("N")), int(g("M"))
        p = EnvConfig.fast_preset(N=N, M=M).to_params()
        assert abs(p.R_jam[int(g("band"))] - float(g("R_jam"))) < 1e-12
        nc = int(g("ncent"))
        cx = np.zeros((1, N + 1)); cy = np.zeros((1, N + 1))
        cx[0, :nc] = g("cent_xy")[:nc, 0]; cy[0, :nc] = g("cent_xy")[:nc, 1]
        t = dict(lam=dev(np.array([g("lam")]), torch.float64), lam0=dev(np.array([g("lam0")]), torch.float64),
                 band=dev(np.array([g("band")]), torch.int32), jx=dev(g("jpos")[None, :, 0], torch.float64),
                 jy=dev(g("jpos")[None, :, 1], torch.float64), jb=dev(g("jb")[None], torch.int32),
                 vx=dev(g("vel")[None, :, 0], torch.float32), vy=dev(g("vel")[None, :, 1], torch.float32),
                 cx=dev(cx, torch.float64), cy=dev(cy, torch.float64),
                 ncl=dev(np.array([g("n_clusters")]), torch.int32), nce=dev(np.array([nc]), torch.int32),
                 asg=dev(g("assign")[None], torch.int32))
        out = torch.zeros(1, M, dtype=torch.float32, device="cuda")
        _lib.check(lib.marl_reward_per_agent(C.byref(p), 1, t["lam"].data_ptr(), t["lam0"].data_ptr(),
                                             t["band"].data_ptr(), t["jx"].data_ptr(), t["jy"].data_ptr(),
                                             t["jb"].data_ptr(), t["vx"].data_ptr(), t["vy"].data_ptr(),
                                             t["cx"].data_ptr(), t["cy"].data_ptr(), t["ncl"].data_ptr(),
                                             t["nce"].data_ptr(), t["asg"].data_ptr(), out.data_ptr(), st))
        np.testing.assert_allclose(out[0].cpu().numpy(), g("rewards"), rtol=1e-5, atol=2e-6)
    # and from a live env: the mean over the agents relates to the shared reward's terms
    env = BatchedJammerEnv(16, EnvConfig.fast_preset(N=20, M=4), seed=1, auto_reset=False)
    env.reset()
    _, rew, _, _, info = env.step(torch.zeros(16, 4, 3, device="cuda"))
    pa = env.reward_per_agent()
    assert pa.shape == (16, 4)
    np.testing.assert_allclose(pa.mean(1).cpu().numpy(), rew.cpu().numpy(), rtol=1e-5, atol=2e-6)
