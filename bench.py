#!/usr/bin/env python3
"""Benchmark of the batched JammerEnv step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--envs-per-gpu E]

--config picks one of BASELINE.json's configurations (default 3, the one the headline metric is quoted on):

  2  4,096 envs, N=10, M=4, "step + reward only": no re-clustering inside the step, no observation output
  3  8,192 envs per GPU (65,536 over 8 GPUs), N=50, M=8, full step incl. DBSCAN every 10th step + lambda2
  4  8,192 envs, N=200, M=16, full step (large swarm: the eigen-solve dominates)
  5  16,384 envs, N=50, M=8, PPO training: policy inference -> step -> device rollout buffer (T=64), then the
     PPO update; a "step" is one collection step, `value` counts the update time in (end to end), the
     collection rate alone is reported beside it

One "step" = one JammerEnv.step for every env of the batch plus the masked auto-reset of truncated envs.
Fast-preset physics, uniform random actions from a device-resident pool (configs 2-4), weak scaling over
GPUs with no collective on the data path (config 5: one flat gradient all-reduce per optimiser step).  Inputs
are resident in HBM before the timed region.  The envs are stepped, untimed, to 10 steps before the end of
their episode first, so that even a 20-step run contains an auto-reset of the whole batch.

Prints ONE JSON line on rank 0 with a `roofline` object for the step (HIP events on the launch stream around
the whole timed loop) and a `cpu_baseline` object (the numpy restatement of the reference step at the same
N, M, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    2: dict(E=4096, N=10, M=4, steps=2000, warmup=100, reward_only=True,
            name="BASELINE configs[1]: step + reward only (no re-clustering in the step, no observation output)"),
    3: dict(E=8192, N=50, M=8, steps=2000, warmup=200, reward_only=False,
            name="BASELINE configs[2]: full JammerEnv step incl. DBSCAN + lambda2"),
    4: dict(E=8192, N=200, M=16, steps=20, warmup=3, reward_only=False,
            name="BASELINE configs[3]: large swarm, full step (lambda2 eigen-solve and N^2 FSPL dominate)"),
    5: dict(E=16384, N=50, M=8, steps=128, warmup=64, reward_only=False,
            name="BASELINE configs[4]: PPO rollouts into the device rollout buffer + PPO update"),
}
ROLLOUT_T = 64            # config 5: steps per rollout
PEAK_FP32_VALU_TFLOPS = 157.3     # 256 CU x 128 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBPS = 8000.0


def flops_per_env_step(N, M):
    """Algorithmic cost of one env-step: 4/3 N^3 (tridiagonalise L) + 3 M N^2 (midpoint tests, symmetric half)
    + 3.3 N^2 (FSPL + DBSCAN/10) + 60 N (dynamics) + 10 M^2 + 3 M N (reward, observation)."""
    return 4.0 / 3.0 * N ** 3 + 3.0 * M * N * N + 3.3 * N * N + 60.0 * N + 10.0 * M * M + 3.0 * M * N


def bytes_per_env_step(N, M):
    """enemy pos+vel r/w 32N, jammer state 16M, actions 12M, obs 20M, centroid table 6N, scalars 64."""
    return 32.0 * N + 16.0 * M + 12.0 * M + 20.0 * M + 6.0 * N + 64.0


def cpu_worker(args):
    """Times the numpy oracle (restatement of the reference step) for `seconds` on one core."""
    seed, seconds, N, M = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import numpy as np
    from oracle import jammer_oracle as orc
    p = orc.EnvParams.fast_preset(N=N, M=M)
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
        if (n & 7) == 0 and time.perf_counter() - t0 >= seconds:
            break
    return n, time.perf_counter() - t0


def cpu_baseline(N, M, seconds=8.0):
    import multiprocessing as mp
    try:
        import torch
        torch.set_num_threads(1)
    except Exception:
        pass
    n1, t1 = cpu_worker((1234, seconds, N, M))
    cores = int(os.environ.get("MARL_BENCH_CPU_CORES", min(16, os.cpu_count() or 1)))
    ctx = mp.get_context("fork")     # called before this process touches the GPU; no exec involved
    with ctx.Pool(cores) as pool:
        res = pool.map(cpu_worker, [(1234 + i, seconds, N, M) for i in range(cores)])
    tot = sum(n / t for n, t in res)
    return {"value": tot, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"numpy/scipy restatement of the reference step (oracle/jammer_oracle.py), N={N} M={M} "
                      f"fast-preset physics, random actions, full step incl. DBSCAN every 10th step and auto-reset; "
                      f"{cores} independent single-thread processes x {seconds:.0f} s; "
                      f"1 core alone: {n1 / t1:.1f} env-steps/s",
            "single_core_value": n1 / t1}


def profile_summary(cfg_id):
    """Figures of the committed rocprofv3 runs of this same command (profiles/summary_c<n>.json): HBM bytes per step
    and VALU busy % from the counter passes, average kernel durations from the kernel trace, and the commit they were
    taken at.  They are NOT measured by this run and are reported under `profile` with their origin.  {} where no
    profile was taken."""
    path = os.path.join(ROOT, "profiles", f"summary_c{cfg_id}.json")
    if not os.path.exists(path):
        return {}
    try:
        return json.load(open(path))
    except Exception:
        return {}


def solve_launch_time(env, pool, first, reps=32):
    """The launch that holds the eigen-solves, live: `reps` more steps of `env` (one launch sequence) through
    marl_env_step_timed, which records a HIP event before and after that launch on the launch stream; microseconds
    per launch (mean, max)."""
    import torch
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, ev in enumerate(pairs):
        env.step(pool[(first + i) % pool.shape[0]], solve_events=ev)
    torch.cuda.synchronize()
    us = [a.elapsed_time(b) * 1e3 for a, b in pairs]
    return sum(us) / len(us), max(us)


def lambda2_stage_time(env, reps=10):
    """The eigen-solve stage alone, live: marl_lambda2_from_bits on the jammed graphs of the current state
    (bits from the stage ops), HIP events around `reps` calls; microseconds per call."""
    import ctypes as C
    import torch
    from marl_amd import _lib
    lib, v, E, N, M = env.lib, env.views, env.E, env.N, env.M
    W = (N + 31) // 32
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    adj = torch.zeros(E, N, W, dtype=torch.int32, device=env.device)
    jam = torch.zeros_like(adj)
    _lib.check(lib.marl_adjacency_bits(C.byref(env.params), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(),
                                       v["enemy_band"].data_ptr(), adj.data_ptr(), st))
    _lib.check(lib.marl_jam_mask_bits(C.byref(env.params), E, v["enemy_x"].data_ptr(), v["enemy_y"].data_ptr(),
                                      v["enemy_band"].data_ptr(), v["jam_x"].data_ptr(), v["jam_y"].data_ptr(),
                                      v["jam_band"].data_ptr(), jam.data_ptr(), st))
    eff = (adj & ~jam).contiguous()
    out = torch.zeros(E, dtype=torch.float64, device=env.device)
    nb = lib.marl_lambda2_workspace_bytes(N, E)
    ws = torch.zeros(max(nb, 8), dtype=torch.uint8, device=env.device)
    call = lambda: _lib.check(lib.marl_lambda2_from_bits(N, E, eff.data_ptr(), out.data_ptr(), 0,
                                                         ws.data_ptr() if nb else None, st))
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def launch_ranks(n, share_device):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script (fresh interpreters, one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the command line passed on) and wait for them.
    Rank 0 prints the JSON line.  This process never initialises the GPU (counting devices does not)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if not share_device and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return next((c for c in codes if c != 0), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--shards", type=int, default=None,
                    help="parts of the batch stepped on HIP streams of their own (default: 3 for config 3 and 2 for "
                         "config 2 when at least 400 steps are timed, else 1)")
    ap.add_argument("--schedule", default=os.environ.get("MARL_BENCH_SCHEDULE", "auto"),
                    choices=["auto", "fused", "split"], help="EnvConfig.step_schedule (testing: A/B runs)")
    ap.add_argument("--no-solve-priority", action="store_true",
                    help="testing: EnvConfig.solve_priority=False (the solve launch leaves the wave priority alone)")
    ap.add_argument("--solver", default="auto", choices=["auto", "direct", "lanczos"],
                    help="testing: EnvConfig.lambda2_solver (auto: the direct solver up to 64 enemies, the Lanczos iteration on bit rows beyond)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing only: initialise torch.distributed (and so RCCL's streams) even with one rank")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()
    cfgd = CONFIGS[args.config]
    N_ENEMY, M_JAM = cfgd["N"], cfgd["M"]
    steps = args.steps if args.steps is not None else cfgd["steps"]
    warmup = args.warmup if args.warmup is not None else cfgd["warmup"]
    E = args.envs_per_gpu if args.envs_per_gpu is not None else cfgd["E"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args.share_device))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CPU baseline first: its worker processes are forked before anything initialises the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(N_ENEMY, M_JAM, args.cpu_seconds)

    import torch
    from marl_amd import BatchedJammerEnv, EnvConfig
    dist = None
    if args.share_device:
        local_rank = 0
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cfg = EnvConfig.fast_preset(N=N_ENEMY, M=M_JAM, recluster=not cfgd["reward_only"],
                                step_schedule=args.schedule, solve_priority=not args.no_solve_priority,
                                lambda2_solver=args.solver)
    extra = {}
    if args.config == 5:
        from marl_amd.trainer import BatchedTrainer
        steps = max(ROLLOUT_T, (steps + ROLLOUT_T - 1) // ROLLOUT_T * ROLLOUT_T)      # whole rollouts
        warmup = (warmup + ROLLOUT_T - 1) // ROLLOUT_T * ROLLOUT_T
        tr = BatchedTrainer(cfg, num_envs=E, rollout_length=ROLLOUT_T, total_timesteps=10 ** 12, seed=1234, device=dev)
        env = tr.env
        for _ in range(max(2, warmup // ROLLOUT_T)):       # eager rollout, graph capture, first updates
            tr.collect_rollout()
            tr.update()
        warmup = max(2, warmup // ROLLOUT_T) * ROLLOUT_T
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (steps // ROLLOUT_T) + 1)]
        barrier()
        t0 = time.perf_counter()
        ev[0].record()
        for r in range(steps // ROLLOUT_T):
            tr.collect_rollout()
            ev[2 * r + 1].record()
            tr.update()
            ev[2 * r + 2].record()
        barrier()
        elapsed = time.perf_counter() - t0
        coll_ms = sum(ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(steps // ROLLOUT_T))
        upd_ms = sum(ev[2 * r + 1].elapsed_time(ev[2 * r + 2]) for r in range(steps // ROLLOUT_T))
        # the env step alone on this batch, for the roofline of the step kernels inside the loop
        pool = torch.empty(8, E, M_JAM, 3, device=dev).uniform_(-5, 5)
        pool[..., 2] = torch.randint(0, 4, (8, E, M_JAM), device=dev).float()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        solo = BatchedJammerEnv(E, cfg, seed=99, device=dev, auto_reset=True, env_offset=rank * E)
        solo.reset()
        for i in range(20):
            solo.step(pool[i % 8])
        e0.record()
        for i in range(50):
            solo.step(pool[i % 8])
        e1.record()
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / 50
        extra = {"collection_env_steps_per_s": world * E * steps / (coll_ms * 1e-3),
                 "collection_ms_per_step": coll_ms / steps,
                 "update_s_per_rollout": upd_ms * 1e-3 / (steps // ROLLOUT_T),
                 "rollout_length": ROLLOUT_T, "ppo_epochs": tr.agent.n_epochs,
                 "minibatch": tr.agent.batch_size,
                 "timed_region": "rollouts (policy inference + step + buffer writes, replayed HIP graph) and PPO updates "
                                 "(GAE kernel, 10 epochs x 32 minibatches unless the KL test stops earlier; a minibatch step = marl_ppo_rows + "
                                 "marl_ppo_products + gradient all-reduce + clip + Adam)",
                 "fused_update": bool(tr.agent.fused_update)}
        env_for_stage = solo
    else:
        # Parts on streams of their own pay over long runs; a timed region of a few milliseconds that holds an auto-reset
        # varies between 0.118 and 0.175 ms per step with three parts (tools/short_runs.sh: 20 steps, the same command
        # four times), 0.110 +- 0.001 with one launch sequence: short runs take the one stream.
        shards = args.shards if args.shards is not None else ({3: 3, 2: 2}.get(args.config, 1) if steps >= 400 else 1)
        env = BatchedJammerEnv(E, cfg, seed=1234, device=dev, auto_reset=True, env_offset=rank * E,
                               write_obs=not cfgd["reward_only"], num_shards=shards)
        env.reset()
        # action pool: v ~ U[-5,5]^2, band ~ U{0..3}; generated on the device before the timed region
        pool_n = 64 if E * M_JAM <= 8192 * 16 else 16
        g = torch.Generator(device=dev).manual_seed(4321 + rank)
        pool = torch.empty(pool_n, E, M_JAM, 3, device=dev, dtype=torch.float32)
        pool[..., :2] = (torch.rand(pool_n, E, M_JAM, 2, device=dev, generator=g) - 0.5) * 10.0
        pool[..., 2] = torch.randint(0, 4, (pool_n, E, M_JAM), device=dev, generator=g).float()
        # untimed: up to 10 steps (half the run if shorter) before the end of the episode, warm-up included
        to_end = min(10, max(1, steps // 2))
        preroll = (cfg.max_steps - warmup - to_end) % cfg.max_steps
        for i in range(preroll + warmup):
            env.step(pool[i % pool_n])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()            # torch records on the current stream, the one env.step launches on
        env.fork()             # (parts on streams of their own start after e0 ...
        for i in range(steps):
            env.step(pool[(preroll + warmup + i) % pool_n])
        env.join()             #  ... and e1 waits for them)
        e1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        kern_ms = e0.elapsed_time(e1) / steps
        extra = {"streams": shards,
                 "auto_resets_in_timed_region": (to_end <= steps) + max(0, steps - to_end) // cfg.max_steps,
                 "untimed_preroll_steps": preroll}
        if shards > 1 and rank == 0:
            # the same steps as one launch sequence on one stream (what the per-kernel profiles under profiles/ time)
            one = BatchedJammerEnv(E, cfg, seed=1234, device=dev, auto_reset=True, env_offset=rank * E,
                                   write_obs=not cfgd["reward_only"])
            one.reset()
            for i in range(preroll + warmup):
                one.step(pool[i % pool_n])
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for i in range(steps):
                one.step(pool[(preroll + warmup + i) % pool_n])
            f1.record()
            torch.cuda.synchronize()
            extra["one_stream_ms_per_step"] = f0.elapsed_time(f1) / steps
            env_for_stage = one
        else:
            env_for_stage = env
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [float(x.item()) / steps * 1e3 for x in every]
        elapsed = max(per_rank_ms) * steps * 1e-3
        extra["per_rank_ms_per_step"] = per_rank_ms
    lam_mean = float(env.lambda2.mean())
    steps_per_s = world * E * steps / elapsed

    if rank == 0:
        stage_us = lambda2_stage_time(env_for_stage)
        solve_us, solve_us_max = solve_launch_time(env_for_stage, pool, 0)
        F = flops_per_env_step(N_ENEMY, M_JAM)
        B = bytes_per_env_step(N_ENEMY, M_JAM)
        achieved_tflops = F * E / (kern_ms * 1e-3) / 1e12
        prof = profile_summary(args.config)
        sched = "one launch (solve inside the wave that stepped the env)" if (N_ENEMY <= 16 or E < 1024) else \
            ("env_step_front_kernel + solve_order_kernel + env_step_solve_kernel" if N_ENEMY <= 64
             else "env_step_big_kernel + env_step_team_kernel")
        out = {
            "metric": "env-steps/sec (whole node) at N=50,M=8" if args.config in (3, 5)
                      else f"env-steps/sec (whole node) at N={N_ENEMY},M={M_JAM}",
            "value": steps_per_s, "unit": "env-steps/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfgd['name']}, N={N_ENEMY} enemies, M={M_JAM} jammers, fast-preset physics, "
                                   f"{'policy actions' if args.config == 5 else 'random actions'}, "
                                   f"{E} envs per GPU ({world * E} total), auto-reset every {cfg.max_steps} steps",
                       "baseline_config_index": args.config - 1,
                       "envs_per_gpu": E, "N": N_ENEMY, "M": M_JAM, "parallelism": f"env-shard x{world}"},
            "roofline": {"bound": "valu", "achieved": achieved_tflops, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / PEAK_FP32_VALU_TFLOPS,
                         # HBM bytes per step: counter passes cannot run inside this process; the figure of the
                         # committed passes of this command, see `profile` for where it comes from
                         "traffic": prof.get("hbm_bytes_per_step"),
                         "kernel": f"marl_env_step = {sched} (+ the masked reset launch on the steps that need it); "
                                   + (f"the batch in {extra['streams']} parts on streams of their own, whose kernels overlap; "
                                      if extra.get("streams", 1) > 1 else "") +
                                   "kernel_ms = HIP-event span of the timed loop on the launch stream / steps"
                                   + (" (config 5: of a 50-step loop of the env step alone on the same batch)"
                                      if args.config == 5 else ""),
                         "kernel_ms": kern_ms,
                         "algorithmic_flops_per_env_step": F,
                         "hbm": {"algorithmic_bytes_per_env_step": B,
                                 "achieved_GBps": B * E / (kern_ms * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                                 "frac": B * E / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}},
            "lambda2_mean": lam_mean,
            "us_per_env_step_kernel": kern_ms * 1e3 / E,
            # Both measured by this run, after the timed region.  Solve launch: HIP events recorded by the library
            # around the launch of the step that holds the eigen-solves (longest first), 32 further steps.  Stage op:
            # marl_lambda2_from_bits on the jammed graphs of the final state (envs in index order; connectivity
            # test + complement compaction + solve in one wave).
            "lambda2_solve_us_per_env": solve_us / E,
            "lambda2_solve_launch_us": {"mean": solve_us, "max": solve_us_max, "launches": 32,
                                        "source": "live: marl_env_step_timed, HIP events on the launch stream"},
            "lambda2_stage_op_us_per_env": stage_us / E,
        }
        if prof:
            out["profile"] = {
                "source": f"profiles/summary_c{args.config}.json: rocprofv3 kernel trace and --pmc passes of "
                          f"`{prof.get('command', 'python3 bench.py --config %d --shards 1' % args.config)}`, taken at commit "
                          f"{prof.get('git_head', 'unknown')}; not measured by this run",
                "hbm_bytes_per_step": prof.get("hbm_bytes_per_step"),
                "valu_busy_pct": prof.get("valu_busy_pct"),
                "kernel_avg_us": prof.get("kernel_avg_us")}
        out.update(extra)
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_all_cores"] = steps_per_s / cpu["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
