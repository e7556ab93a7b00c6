#!/usr/bin/env python3
"""Benchmark of the batched JammerEnv step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu E]

One "step" = one JammerEnv.step for every env of the batch (full step: action
decode, enemy dynamics, DBSCAN every 10th step, FSPL adjacency, jamming mask,
lambda2, reward, observation) plus the masked auto-reset of truncated envs.
Workload: BASELINE.json configs[2] -- N=50 enemies, M=8 jammers, fast-preset
physics, uniform random actions, 8,192 envs per GPU (65,536 over 8 GPUs, weak
scaling, no collective on the data path).  Inputs (state, action pool) are
resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (metric env-steps/s whole job) with a `roofline`
object for the step kernel and a `cpu_baseline` object (the numpy restatement of
the reference step, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENEMY, M_JAM = 50, 8
# Algorithmic cost model of one env-step (flops, bytes):
#   4/3 N^3 (tridiagonalise L) + 3 M N^2 (midpoint tests, symmetric half) + 3.3 N^2 (FSPL + DBSCAN/10)
#   + 60 N (dynamics) + 10 M^2 + 3 M N (reward, observation)
#   bytes: enemy pos+vel r/w 32N, jammer state 16M, actions 12M, obs 20M, centroid table 6N, scalars 64


def flops_per_env_step(N, M):
    return 4.0 / 3.0 * N ** 3 + 3.0 * M * N * N + 3.3 * N * N + 60.0 * N + 10.0 * M * M + 3.0 * M * N


def bytes_per_env_step(N, M):
    return 32.0 * N + 16.0 * M + 12.0 * M + 20.0 * M + 6.0 * N + 64.0


PEAK_FP32_VALU_TFLOPS = 157.3     # 256 CU x 128 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBPS = 8000.0


def cpu_worker(args):
    """Times the numpy oracle (restatement of the reference step) for `seconds` on one core."""
    seed, seconds = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import numpy as np
    from oracle import jammer_oracle as orc
    p = orc.EnvParams.fast_preset(N=N_ENEMY, M=M_JAM)
    rng = np.random.RandomState(seed)
    st = orc.EnvState.empty(p)

    def reset():
        A = p.arena_size
        epos = rng.uniform(0.1 * A, 0.9 * A, size=(p.N, 2))
        ang = rng.uniform(0, 2 * np.pi, size=p.N)
        evel = np.stack([p.v_enemy * np.cos(ang), p.v_enemy * np.sin(ang)], axis=1)
        orc.env_reset(p, st, epos, evel, rng.randint(0, 4), rng.uniform(0.1 * A, 0.9 * A, size=(p.M, 2)),
                      rng.randint(0, 4, size=p.M))

    reset()
    n = 0
    t0 = time.perf_counter()
    while True:
        act = rng.uniform(-5, 5, size=(p.M, 3)).astype(np.float32)
        act[:, 2] = rng.randint(0, 4, size=p.M)
        noise = rng.normal(0, p.noise_std, size=(p.N, 2))
        _, _, _, trunc, _ = orc.env_step(p, st, act, noise, want_avg_power=False)
        n += 1
        if trunc:
            reset()
        if (n & 31) == 0 and time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(seconds=8.0):
    import multiprocessing as mp
    try:
        import torch
        torch.set_num_threads(1)
    except Exception:
        pass
    n1, t1 = cpu_worker((1234, seconds))
    cores = int(os.environ.get("MARL_BENCH_CPU_CORES", min(16, os.cpu_count() or 1)))
    ctx = mp.get_context("fork")     # called before this process touches the GPU; no exec involved
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_worker, [(1234 + i, seconds) for i in range(cores)])
    tot = sum(n / t for n, t in res)
    return {"value": tot, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"numpy/scipy restatement of the reference step (oracle/jammer_oracle.py), N={N_ENEMY} M={M_JAM} "
                      f"fast-preset physics, random actions, full step incl. DBSCAN every 10th step and auto-reset; "
                      f"{cores} independent single-thread processes x {seconds:.0f} s; "
                      f"1 core alone: {n1 / t1:.1f} env-steps/s",
            "single_core_value": n1 / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CPU baseline first: its worker processes are forked before anything initialises the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)

    import torch
    from marl_amd import BatchedJammerEnv, EnvConfig
    dist = None
    if args.share_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    E = args.envs_per_gpu
    cfg = EnvConfig.fast_preset(N=N_ENEMY, M=M_JAM)
    env = BatchedJammerEnv(E, cfg, seed=1234, device=dev, auto_reset=True, env_offset=rank * E)
    env.reset()
    # action pool: v ~ U[-5,5]^2, band ~ U{0..3}; generated on the device before the timed region
    pool_n = 64
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    pool = torch.empty(pool_n, E, M_JAM, 3, device=dev, dtype=torch.float32)
    pool[..., :2] = (torch.rand(pool_n, E, M_JAM, 2, device=dev, generator=g) - 0.5) * 10.0
    pool[..., 2] = torch.randint(0, 4, (pool_n, E, M_JAM), device=dev, generator=g).float()

    for i in range(args.warmup):
        env.step(pool[i % pool_n])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every 4th step only: an event record is a stream marker of its own and costs about
    # a microsecond of device time each; torch records on the current stream, the one env.step launches on
    every = 4
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range((args.steps + every - 1) // every)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(len(ev0))]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # env.step() = marl_env_step (three launches: front, order, solve) + the masked reset launch on the
        # steps where an env can have reached max_steps (BatchedJammerEnv keeps host-side bounds on the
        # step counters; here: every 200th step, all envs at once)
        act = pool[(args.warmup + i) % pool_n]
        if i % every == 0:
            ev0[i // every].record()
        env.step(act)
        if i % every == 0:
            ev1[i // every].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / len(ev0)
    lam_mean = float(env.lambda2.mean())
    steps_per_s = world * E * args.steps / elapsed

    if rank == 0:
        F = flops_per_env_step(N_ENEMY, M_JAM)
        B = bytes_per_env_step(N_ENEMY, M_JAM)
        achieved_tflops = F * E / (kern_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node) at N=50,M=8",
            "value": steps_per_s, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: full JammerEnv step incl. DBSCAN + lambda2, N={N_ENEMY} "
                                   f"enemies, M={M_JAM} jammers, fast-preset physics, random actions, "
                                   f"{E} envs per GPU ({world * E} total), auto-reset every 200 steps",
                       "envs_per_gpu": E, "N": N_ENEMY, "M": M_JAM, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "valu", "achieved": achieved_tflops, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP32_VALU_TFLOPS, "traffic": traffic,
                         "kernel": "marl_env_step = env_step_front_kernel + solve_order_kernel + env_step_solve_kernel "
                                   "(back-to-back launches of one call; kernel_ms is their span, HIP events around every 4th step "
                                   "of the timed region)",
                         "kernel_ms": kern_ms,
                         "algorithmic_flops_per_env_step": F,
                         "hbm": {"algorithmic_bytes_per_env_step": B,
                                 "achieved_GBps": B * E / (kern_ms * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                                 "frac": B * E / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}},
            "lambda2_mean": lam_mean,
            "us_per_env_step_kernel": kern_ms * 1e3 / E,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_all_cores"] = steps_per_s / cpu["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
