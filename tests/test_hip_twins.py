"""The eigenproblem is reduced before it is solved: twin vertices of the complement graph are merged (same neighbours,
adjacent to each other or not: they only contribute eigenvalues that cannot exceed the largest one) and components of
at most dmax + 1 vertices are dropped (their eigenvalues cannot either).  The quantity is compute_lambda2's
(physics/communication_graph.py:219-286), the bar LAPACK's dsyevr as for every other solver path; the graphs here are
built so that the reduction has work to do, up to the cases where nothing is left of the problem."""
import numpy as np
import pytest

from marl_amd import _lib
from test_hip_lanczos import _lambda2

pytestmark = pytest.mark.gpu


def _with_twins(rng, n, count):
    """Adjacency matrices [count, n, n] whose complements are random graphs on a few "class" vertices, each blown up into
    a class of open twins (an independent set) or of closed twins (a clique), plus untouched vertices."""
    out = []
    for k in range(count):
        idle = rng.randint(0, max(1, n // 4))                      # vertices adjacent to everyone: not in the problem
        m = n - idle
        classes = rng.randint(2, max(3, m // 2) + 1)
        owner = np.concatenate([np.arange(classes), rng.randint(0, classes, m - classes)]) if m > classes else np.arange(m)
        rng.shuffle(owner)
        dens = rng.choice([0.15, 0.4, 0.8])
        up = np.triu(rng.rand(classes, classes) < dens, 1)
        q = up | up.T
        for c in range(classes):                                   # no isolated class: every vertex of the problem misses a link
            if not q[c].any():
                d = (c + 1 + rng.randint(classes - 1)) % classes
                q[c, d] = q[d, c] = True
        closed = rng.rand(classes) < 0.4
        comp = np.zeros((n, n), dtype=bool)
        comp[:m, :m] = q[owner][:, owner] | ((owner[:, None] == owner[None, :]) & closed[owner][:, None])
        if k % 3 == 2:                                             # a few edges that break some of the classes again
            extra = np.triu(rng.rand(n, n) < 0.02, 1)
            comp |= extra | extra.T
        if k % 4 == 1 and m >= 12:                                 # small components beside the rest: a few pairs and a triangle
            t = m - 7
            comp[t:m, :] = comp[:, t:m] = False
            comp[t, t + 1] = comp[t + 2, t + 3] = comp[t + 4, t + 5] = comp[t + 5, t + 6] = comp[t + 4, t + 6] = True
            comp |= comp.T
        perm = rng.permutation(n)
        comp = comp[perm][:, perm]
        np.fill_diagonal(comp, False)
        adj = ~comp
        np.fill_diagonal(adj, False)
        out.append(adj)
    return np.stack(out)


def _lapack(adj):
    a = adj.astype(np.float64)
    return np.linalg.eigvalsh(np.diag(a.sum(1)) - a)[1]


@pytest.mark.parametrize("n", [12, 13, 16, 24, 31, 32, 33, 40, 48, 50, 57, 63, 64])
@pytest.mark.parametrize("solver", ["direct", "lanczos"])
def test_graphs_with_twin_classes_against_lapack(n, solver):
    rng = np.random.RandomState(7000 + n)
    adj = _with_twins(rng, n, 96)
    got = _lambda2(adj, _lib.LAMBDA2_DIRECT if solver == "direct" else _lib.LAMBDA2_LANCZOS)
    for e in range(adj.shape[0]):
        want = _lapack(adj[e])
        assert abs(got[e] - want) <= (2e-9 if solver == "direct" else 1e-8) * max(1.0, abs(want)), (n, e, got[e], want)


@pytest.mark.parametrize("n", [14, 32, 33, 50, 64])
def test_closed_forms_made_of_twins(n):
    """Complements that are nothing but twins: a complete graph on k vertices (one closed class: mu_max = k), a star
    (its leaves one open class: mu_max = k), complete bipartite and complete multipartite graphs (mu_max = k); and
    complements that are nothing but small components (disjoint edges, disjoint triangles)."""
    graphs, want = [], []
    for k in (n, n - 1, n - 5, 12):
        for kind in ("clique", "star", "bipartite", "tripartite", "matching", "triangles"):
            comp = np.zeros((n, n), dtype=bool)
            if kind == "clique":
                comp[:k, :k] = True
            elif kind == "star":
                comp[0, 1:k] = comp[1:k, 0] = True
            elif kind in ("matching", "triangles"):                # components of dmax + 1 vertices only: nothing to solve
                c = 2 if kind == "matching" else 3
                kk = k // c * c
                i = np.arange(kk)
                comp[:kk, :kk] = (i[:, None] // c) == (i[None, :] // c)
            else:
                parts = 2 if kind == "bipartite" else 3
                side = np.arange(k) % parts if kind == "tripartite" else (np.arange(k) < k // 3).astype(int)
                comp[:k, :k] = side[:, None] != side[None, :]
            np.fill_diagonal(comp, False)
            adj = ~comp
            np.fill_diagonal(adj, False)
            graphs.append(adj)
            # mu_max = k in the first four; disjoint edges: 2, disjoint triangles: 3
            want.append(float(n - {"matching": 2, "triangles": 3}.get(kind, k)))
    graphs = np.stack(graphs)
    for sel in (_lib.LAMBDA2_DIRECT, _lib.LAMBDA2_LANCZOS):
        got = _lambda2(graphs, sel)
        for e, w in enumerate(want):
            assert abs(got[e] - w) <= 1e-9 * max(1.0, w), (n, e, got[e], w)
            assert abs(got[e] - _lapack(graphs[e])) <= 1e-8, (n, e)


def test_step_graphs_are_merged_and_keep_their_lambda2():
    """On the graphs of a step (N = 50, M = 8, random actions) the front launch hands reduced problems over (solve_np
    carries the bracket of the graph before the reduction above bit 7) with fewer vertices than the complement had, and
    finishes others itself; lambda2 is LAPACK's on the adjacency bits of the same step."""
    import torch
    from marl_amd import BatchedJammerEnv, EnvConfig
    from oracle import jammer_oracle as orc
    E, N, M = 2048, 50, 8
    env = BatchedJammerEnv(E, EnvConfig.fast_preset(N=N, M=M, step_schedule="split"), seed=77, debug_bits=True)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(25):
        act = (torch.rand(E, M, 3, device="cuda", generator=g) - 0.5) * 10
        act[..., 2] = torch.randint(0, 4, (E, M), device="cuda", generator=g).float()
        env.step(act)
    word = env.views["solve_np"].cpu().numpy()
    np_now, np_before, dmax = word & 0xFF, (word >> 8) & 0xFF, word >> 16
    merged = np.nonzero(np_before > 0)[0]
    assert len(merged) > E // 10                                   # a common case, not a corner
    assert np.all(np_now[merged] < np_before[merged]) and np.all(np_now[merged] >= 2)
    assert np.all(dmax[merged] >= 1) and np.all(dmax[merged] < np_before[merged])
    eff = (env.adj_bits & ~env.jam_bits).cpu().numpy()
    lam = env.lambda2.cpu().numpy()
    for e in merged[:160]:
        a = orc.unpack_rows(eff[e].view(np.uint32), N)
        assert np_before[e] == int((a.sum(1) < N - 1).sum())
        want = _lapack(a)
        assert abs(lam[e] - want) <= 2e-9 * max(1.0, abs(want)), (e, lam[e], want)
    # every env of a slice, whatever became of its problem (left alone, reduced, or gone: solve_np = 0 with a complement)
    gone = 0
    for e in range(300):
        a = orc.unpack_rows(eff[e].view(np.uint32), N)
        want = _lapack(a)
        assert abs(lam[e] - want) <= 2e-9 * max(1.0, abs(want)), (e, lam[e], want)
        gone += int(word[e] == 0 and want > 1e-9 and (a.sum(1) < N - 1).any())
    assert gone > 10
