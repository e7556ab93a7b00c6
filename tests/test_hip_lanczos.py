"""The Lanczos path of the lambda2 solve (eigenproblems of 33 to 64 vertices; MARL_LAMBDA2_LANCZOS): the same
reference quantity as the direct solver, compute_lambda2 (physics/communication_graph.py:219-286), so the same
bars -- LAPACK on random graphs of every width and density, the closed forms of the reference's tests, the reference
trajectories step by step, and the direct solver on the benchmark's batch."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

import golden_util as gu
from marl_amd import BatchedJammerEnv, EnvConfig, _lib
from oracle import jammer_oracle as orc

pytestmark = pytest.mark.gpu


def _lambda2(adj, rounds):
    """marl_lambda2_from_bits on bool adjacency matrices [E, N, N]."""
    E, N, _ = adj.shape
    bits = torch.from_numpy(np.stack([orc.pack_rows(a) for a in adj]).view(np.int32)).cuda().contiguous()
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    lib = _lib.load()
    nb = lib.marl_lambda2_workspace_bytes(N, E)
    ws = torch.zeros(max(nb, 8), dtype=torch.uint8, device="cuda")
    _lib.check(lib.marl_lambda2_from_bits(N, E, bits.data_ptr(), out.data_ptr(), rounds, ws.data_ptr() if nb else None,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _random_graphs(rng, n, count):
    """Graphs whose complement (what the solver works on) ranges from a few edges to nearly complete, plus planted
    structures: two far-apart hubs in the complement, near-regular complements, disconnected complements."""
    out = []
    for k in range(count):
        dens = rng.choice([0.03, 0.1, 0.3, 0.5, 0.8, 0.95])
        up = np.triu(rng.rand(n, n) < dens, 1)
        comp = up | up.T                                   # complement graph: the jammed links
        kind = k % 4
        if kind == 1:                                      # two hubs that do not see each other
            comp[:] = False
            h1, h2 = 0, n - 1
            half = n // 2
            comp[h1, 1:half] = comp[1:half, h1] = True
            comp[h2, half:n - 1] = comp[half:n - 1, h2] = True
            extra = np.triu(rng.rand(n, n) < 0.05, 1)
            comp |= extra | extra.T
        elif kind == 2:                                    # ring + chords: many close eigenvalues at the top
            comp[:] = False
            i = np.arange(n)
            for s in (1, 2, 3):
                comp[i, (i + s) % n] = comp[(i + s) % n, i] = True
        np.fill_diagonal(comp, False)
        adj = ~comp
        np.fill_diagonal(adj, False)
        out.append(adj)
    return np.stack(out)


@pytest.mark.parametrize("n", [33, 34, 40, 41, 47, 48, 49, 50, 51, 56, 57, 63, 64])
def test_lanczos_lambda2_against_lapack(n):
    rng = np.random.RandomState(1000 + n)
    adj = _random_graphs(rng, n, 96)
    got = _lambda2(adj, _lib.LAMBDA2_LANCZOS)
    direct = _lambda2(adj, _lib.LAMBDA2_DIRECT)
    for e in range(adj.shape[0]):
        a = adj[e].astype(np.float64)
        want = np.linalg.eigvalsh(np.diag(a.sum(1)) - a)[1]
        assert abs(got[e] - want) <= 1e-8 * max(1.0, abs(want)), (n, e, got[e], want)
        assert abs(direct[e] - want) <= 2e-9 * max(1.0, abs(want)), (n, e, direct[e], want)


@pytest.mark.parametrize("n", [65, 66, 100, 127, 128, 129, 150, 192, 193, 200, 208, 209, 224, 256])
def test_lanczos_on_bit_rows_against_lapack(n):
    """Beyond 64 vertices the iteration works on the adjacency bits themselves (one wave per graph; graphs denser than
    one half through their complement); what does not meet its bound goes to the team solver.  Same bar as below 64."""
    rng = np.random.RandomState(2000 + n)
    adj = _random_graphs(rng, n, 36)
    got = _lambda2(adj, _lib.LAMBDA2_LANCZOS)
    direct = _lambda2(adj, _lib.LAMBDA2_DIRECT)
    for e in range(adj.shape[0]):
        a = adj[e].astype(np.float64)
        want = np.linalg.eigvalsh(np.diag(a.sum(1)) - a)[1]
        assert abs(got[e] - want) <= 1e-8 * max(1.0, abs(want)), (n, e, got[e], want)
        assert abs(direct[e] - want) <= 2e-9 * max(1.0, abs(want)), (n, e, direct[e], want)


def test_lanczos_failures_beyond_the_team_list_are_solved_in_place():
    """What the iteration does not settle goes on a list of 1 024 envs for the team solver; a failure that finds the list
    full is solved by its own wave through global memory.  1 200 copies of a graph whose complement is a path (the top
    of its spectrum, 2 - 2 cos(k pi / n), is a cluster with gaps of 1e-3), with the iteration limited to 24 steps
    (MARL_LAMBDA2_TEST_FEW_STEPS), fill the list and overflow it."""
    n, E = 130, 1200
    comp = np.zeros((n, n), dtype=bool)
    i = np.arange(n - 1)
    comp[i, i + 1] = comp[i + 1, i] = True
    adj = ~comp
    np.fill_diagonal(adj, False)
    a = adj.astype(np.float64)
    want = np.linalg.eigvalsh(np.diag(a.sum(1)) - a)[1]
    bits = torch.from_numpy(np.repeat(orc.pack_rows(adj)[None], E, 0).view(np.int32)).cuda().contiguous()
    lib = _lib.load()
    nb = lib.marl_lambda2_workspace_bytes(n, E)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    out = torch.zeros(E, dtype=torch.float64, device="cuda")
    _lib.check(lib.marl_lambda2_from_bits(n, E, bits.data_ptr(), out.data_ptr(), _lib.LAMBDA2_LANCZOS | _lib.LAMBDA2_TEST_FEW_STEPS,
                                          ws.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    ld = (n + 7) // 8 * 8
    listed = int(ws[E * ld * ld * 8 + (E + 63) // 64 * 64 * 4:][:4].view(torch.int32)[0])
    assert listed == E, listed                       # every copy failed its bound: 1 024 on the list, the rest in place
    got = out.cpu().numpy()
    assert np.all(np.abs(got - want) <= 2e-9 * max(1.0, abs(want))), (got.min(), got.max(), want)


def test_lanczos_closed_forms():
    """K_n minus a few edges, path and cycle (reference tests/test_laplacian.py:251-310) at 50 and 64 vertices."""
    for n in (50, 64):
        P = np.zeros((n, n), dtype=bool)
        i = np.arange(n - 1)
        P[i, i + 1] = P[i + 1, i] = True
        Cy = P.copy()
        Cy[0, n - 1] = Cy[n - 1, 0] = True
        K = ~np.eye(n, dtype=bool)
        Km = K.copy()                                      # complete graph minus a perfect matching: lambda2 = n - 2
        for j in range(0, n, 2):
            Km[j, j + 1] = Km[j + 1, j] = False
        got = _lambda2(np.stack([P, Cy, K, Km]), _lib.LAMBDA2_LANCZOS)
        assert abs(got[0] - 2 * (1 - math.cos(math.pi / n))) < 1e-8
        assert abs(got[1] - 2 * (1 - math.cos(2 * math.pi / n))) < 1e-8
        assert got[2] == n and abs(got[3] - (n - 2)) < 1e-8


def test_step_with_lanczos_equals_step_with_direct_solver():
    """2 048 envs x (N=50, M=8), split schedule, 40 steps across an auto-reset: state bit-equal, lambda2 to 1e-7
    max(1, lambda2), reward to 1e-6."""
    E, N, M = 2048, 50, 8
    envs = [BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M, max_steps=25, step_schedule="split", lambda2_solver=s),
                             seed=21, auto_reset=True) for s in ("direct", "lanczos")]
    for env in envs:
        env.reset()
    g = torch.Generator(device="cuda").manual_seed(8)
    for t in range(40):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        outs = []
        for env in envs:
            obs, rew, _, trunc, info = env.step(act)
            outs.append((obs.clone(), rew.clone(), info["lambda2_current"].clone(), trunc.clone()))
        (oa, ra, la, ta), (ob, rb, lb, tb) = outs
        assert torch.equal(oa, ob) and torch.equal(ta, tb), t
        assert torch.equal(envs[0].views["enemy_x"], envs[1].views["enemy_x"])
        assert float(((la - lb).abs() / la.abs().clamp(min=1.0)).max()) <= 1e-7, t
        assert float((ra - rb).abs().max()) <= 1e-6, t


@pytest.mark.parametrize("name", ["n50m8_random", "n50m8_chase"])
def test_reference_trajectory_with_lanczos(name):
    """The reference's own trajectories at N = 50 with the Lanczos path: lambda2 and reward of every step."""
    from test_hip_parity import cfg_from, dev, lam_ok
    z = gu.load(f"traj_{name}.npz")
    p = gu.traj_params(z)
    resets = gu.traj_resets(z)
    E = 3
    env = BatchedJammerEnv(E, cfg_from(p, step_schedule="split", lambda2_solver="lanczos"), auto_reset=False)
    rep = lambda a: np.repeat(np.asarray(a)[None], E, 0)

    def do_reset(r):
        env.reset(enemy_pos=rep(r["epos"]), enemy_vel=rep(r["evel"]), enemy_band=rep(np.int32(r["band"])),
                  jam_init=rep(r["jpos"] if p.random_jammer_start else r["offsets"]), jam_band=rep(r["jb"]))

    do_reset(resets[0])
    for t in range(len(z["reward"])):
        obs, rew, _, trunc, info = env.step(dev(rep(z["action"][t]), torch.float32), dev(rep(z["noise"][t]), torch.float64))
        torch.cuda.synchronize()
        assert lam_ok(float(info["lambda2_current"][1]), float(z["lam"][t])), t
        assert abs(float(rew[1]) - float(z["reward"][t])) <= 1e-5 * max(1.0, abs(float(z["reward"][t]))), t
        np.testing.assert_allclose(obs[1].cpu().numpy(), z["obs"][t], rtol=1e-5, atol=1e-6)
        if bool(z["trunc"][t]):
            do_reset(resets[t + 1])
