"""Loaders for the golden vectors in tests/golden/ (produced by make_golden.py
from the reference).  Shared by the oracle tests (CPU) and the HIP parity tests."""
import os

import numpy as np

from oracle import jammer_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJ_NAMES = ["n5m2_random", "n10m4_random", "n10m4_chase", "n50m8_random", "n50m8_chase",
              "n200m16_chase", "n20m4_centroid", "n30m6_bigarena"]
CT_TRAJ_NAMES = ["n10m4_ct", "n50m8_ct"]        # coordinated-turn enemy dynamics


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def stage_cases():
    z = load("stage.npz")
    meta = z["meta"]
    cases = []
    for i, m in enumerate(meta):
        N, M = int(m[0]), int(m[1])
        p = orc.EnvParams(M=M, N=N, arena_size=float(m[2]), P_tx_dbm=float(m[3]),
                          P_sens_dbm=float(m[4]), P_jammer_dbm=float(m[5]),
                          P_jam_thresh_dbm=float(m[6]), eps=float(m[7]), min_samples=int(m[8]))
        d = {k[len(f"c{i}_"):]: z[k] for k in z.files if k.startswith(f"c{i}_")}
        cases.append((p, d))
    return cases


def traj_params(z):
    g = lambda k: z[f"cfg_{k}"].item()
    ct = "cfg_coordinated_turn" in z.files and int(g("coordinated_turn")) == 1
    extra = dict(enemy_mode="coordinated_turn", omega_max=g("omega_max")) if ct else {}
    return orc.EnvParams(**extra, M=int(g("M")), N=int(g("N")), arena_size=g("arena_size"),
                         max_steps=int(g("max_steps")), dt=g("dt"), v_max=g("v_max"),
                         v_enemy=g("v_enemy"), P_tx_dbm=g("P_tx_dbm"), P_sens_dbm=g("P_sens_dbm"),
                         P_jammer_dbm=g("P_jammer_dbm"), P_jam_thresh_dbm=g("P_jam_thresh_dbm"),
                         eps=g("eps"), min_samples=int(g("min_samples")),
                         K_recompute=int(g("K_recompute")),
                         random_jammer_start=bool(int(g("random_jammer_start"))))


def traj_resets(z):
    """{at_step: dict of reset vectors}"""
    out = {}
    for i in range(int(z["n_resets"])):
        pre = f"reset{i}_"
        d = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        out[int(d["at_step"])] = d
    return out


def cluster_state_from(N, ids, xy, size, ncent):
    """Golden flattened dict -> oracle ClusterState (current part only)."""
    cs = orc.ClusterState.empty(N)
    ncent = int(ncent)
    cs.cent_xy[:ncent] = xy[:ncent]
    cs.cent_size[:ncent] = size[:ncent]
    cs.ncent = ncent
    cs.n_clusters = int(np.sum(ids[:ncent] >= 0))
    return cs


def lam_close(a, b, rtol=1e-5):
    """|a-b| <= rtol*max(|b|,1): the reference returns +-1e-15 noise for
    disconnected graphs, so the tolerance is relative to max(|lambda2|, 1)."""
    return abs(a - b) <= rtol * max(abs(b), 1.0)


def retry_cases():
    """Cases of reset_retry.npz: (params, seed, [reset dicts in call order])."""
    z = load("reset_retry.npz")
    out = []
    for c in range(int(z["n_cases"])):
        g = lambda k: z[f"c{c}_cfg_{k}"].item()
        p = orc.EnvParams(M=2, N=5, arena_size=g("arena_size"), max_steps=int(g("max_steps")),
                          P_jammer_dbm=g("P_jammer_dbm"), P_jam_thresh_dbm=g("P_jam_thresh_dbm"), eps=g("eps"),
                          random_jammer_start=bool(int(z[f"c{c}_rjs"])),
                          max_reset_attempts=int(z[f"c{c}_max_attempts"]))
        recs = []
        for r in range(int(z[f"c{c}_nresets"])):
            pre = f"c{c}_r{r}_"
            recs.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        out.append((p, int(z[f"c{c}_seed"]), recs))
    return out
