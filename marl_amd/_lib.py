"""ctypes binding of libmarl_env.so (C ABI declared in include/marl_env.h).

The library is the product path: if it is missing this module raises, there is
no CPU fallback.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C marl_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARL_ENV_LIB") or os.path.join(_HERE, "libmarl_env.so")   # override: A/B builds

NUM_BANDS = 4
OBS_DIM = 5
ASSIGN_NONE = -2
MAX_N = 256
MAX_M = 64
ENEMY_RANDOM_WALK, ENEMY_COORDINATED_TURN = 0, 1
SCHEDULE_AUTO, SCHEDULE_FUSED, SCHEDULE_SPLIT = 0, 1, 2
STAGE_NO_RECLUSTER = 1
STAGE_NO_SOLVE_PRIORITY = 2
STAGE_GENERAL_RESET = 4
LAMBDA2_DIRECT, LAMBDA2_LANCZOS = 0x100, 0x200
LAMBDA2_TEST_FEW_STEPS = 0x400        # testing: the bit-row Lanczos iteration gives up after 24 steps

MARL_OK, MARL_EINVAL, MARL_EUNSUPPORTED, MARL_EHIP, MARL_ETOOBIG = 0, -1, -2, -3, -4
STATIC_SCRATCH_BYTES = 32 << 20

_pd, _pf, _pi, _pu8, _pu32 = (C.c_void_p,) * 5   # device pointers travel as integers


class MarlEnvParams(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("M", C.c_int32),
        ("max_steps", C.c_int32), ("K_recompute", C.c_int32), ("min_samples", C.c_int32),
        ("random_jammer_start", C.c_int32), ("max_reset_attempts", C.c_int32),
        ("lambda2_rounds", C.c_int32), ("enemy_mode", C.c_int32), ("step_schedule", C.c_int32),
        ("arena_size", C.c_double), ("dt", C.c_double), ("v_max", C.c_double), ("v_enemy", C.c_double),
        ("P_tx_w", C.c_double), ("P_sens_w", C.c_double), ("P_jam_w", C.c_double), ("P_thr_w", C.c_double),
        ("omega", C.c_double * 5),
        ("kappa", C.c_double), ("eps", C.c_double), ("drift_threshold", C.c_double),
        ("spread", C.c_double), ("noise_std", C.c_double), ("min_lambda2_initial", C.c_double),
        ("omega_max", C.c_double),
        ("comm_d2_max", C.c_double * NUM_BANDS), ("jam_d2_max", C.c_double * NUM_BANDS),
        ("R_jam", C.c_double * NUM_BANDS),
        ("dt_f32", C.c_float), ("v_max_f32", C.c_float), ("v_max_sq_f32", C.c_float),
        ("jam_lo_f32", C.c_float * NUM_BANDS), ("jam_hi_f32", C.c_float * NUM_BANDS), ("stage_flags", C.c_int32),
    ]


_STATE_ARRAYS = [
    "enemy_x", "enemy_y", "enemy_vx", "enemy_vy", "enemy_omega", "jam_x", "jam_y", "jam_band", "last_vx", "last_vy",
    "step_count", "enemy_band", "tick", "lambda2_initial", "labels", "cent_x", "cent_y", "cent_size",
    "n_clusters", "ncent", "prev_x", "prev_y", "prev_n_clusters", "prev_ncent", "assign", "workspace",
    "solve_rows", "solve_meta", "solve_np", "solve_slot", "solve_perm", "solve_hist", "solve_partial", "episode_return",
]


class MarlEnvState(C.Structure):
    _fields_ = [("E", C.c_int32), ("pad", C.c_int32)] + [(n, C.c_void_p) for n in _STATE_ARRAYS]


class MarlStepIO(C.Structure):
    _fields_ = [
        ("actions", _pf), ("noise", _pd), ("obs", _pf), ("reward", _pf), ("truncated", _pu8),
        ("lambda2", _pd), ("reward_terms", _pf), ("avg_power_dbm", _pd), ("adj_bits", _pu32),
        ("jam_bits", _pu32), ("episode_stats", _pd), ("seed", C.c_uint64), ("env_offset", C.c_uint64),
    ]


class MarlPolicyIO(C.Structure):
    _fields_ = [
        ("weights", _pf), ("obs", _pf), ("tick", _pi), ("actions", _pf), ("log_probs", _pf), ("values", _pf),
        ("seed", C.c_uint64), ("env_offset", C.c_uint64), ("call", C.c_uint32), ("deterministic", C.c_int32),
    ]


class MarlPpoIO(C.Structure):
    _fields_ = [
        ("weights", _pf), ("obs", _pf), ("actions", _pf), ("old_log_probs", _pf), ("advantages", _pf), ("returns", _pf),
        ("index", C.c_void_p), ("xe", _pf), ("y1e", _pf), ("y2e", _pf), ("dz1", _pf), ("dz2", _pf), ("actor_sums", _pf),
        ("dh", _pf), ("cy1e", _pf), ("cy2e", _pf), ("cdz1", _pf), ("cdz2", _pf), ("dv", _pf),
        ("stats", _pd),
        ("clip_eps", C.c_float), ("c1", C.c_float), ("c2", C.c_float), ("v_max", C.c_float),
        ("log_std_min", C.c_float), ("log_std_max", C.c_float),
    ]


class MarlPpoApply(C.Structure):
    _fields_ = [
        ("params", _pf), ("exp_avg", _pf), ("exp_avg_sq", _pf), ("steps", _pf), ("grads", _pf),
        ("lr_actor", _pf), ("lr_critic", _pf), ("gather", C.c_void_p), ("packed", _pf),
        ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("n_actor", C.c_int32), ("n_critic", C.c_int32), ("tensors_actor", C.c_int32), ("tensors_critic", C.c_int32),
        ("n_packed", C.c_int32), ("max_norm", C.c_float), ("grad_scale", C.c_float),
    ]


class MarlRewardObsIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "lambda2", "lambda2_initial", "enemy_band", "enemy_x", "enemy_y", "jam_x", "jam_y", "jam_band", "vx", "vy",
        "cent_x", "cent_y", "cent_size", "n_clusters", "ncent", "assign", "reward", "reward_terms", "obs", "avg_power_dbm")]


class MarlResetIO(C.Structure):
    _fields_ = [
        ("mask", _pu8), ("enemy_pos", _pd), ("enemy_vel", _pd), ("enemy_omega", _pd), ("enemy_band", _pi), ("jam_init", _pd),
        ("jam_band", _pi), ("obs", _pf), ("seed", C.c_uint64), ("episode", C.c_uint64),
        ("env_offset", C.c_uint64),
    ]


# every symbol include/marl_env.h declares, with its ctypes signature
SYMBOLS = {
    "marl_params_init": (C.c_int, [C.POINTER(MarlEnvParams)]),
    "marl_state_bytes": (C.c_size_t, [C.POINTER(MarlEnvParams), C.c_int32]),
    "marl_state_bind": (C.c_int, [C.POINTER(MarlEnvState), C.POINTER(MarlEnvParams), C.c_int32, C.c_void_p]),
    "marl_env_step_parts": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.POINTER(MarlStepIO), C.c_int32,
                                      C.POINTER(C.c_void_p), C.c_void_p]),
    "marl_env_reset_parts": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.POINTER(MarlResetIO), C.c_int32,
                                       C.POINTER(C.c_void_p), C.c_void_p]),
    "marl_shard_scratch_bytes": (C.c_size_t, []),
    "marl_state_shard": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.c_int32, C.c_int32, C.c_void_p,
                                   C.POINTER(MarlEnvState)]),
    "marl_strerror": (C.c_char_p, [C.c_int]),
    "marl_last_hip_error": (C.c_int, []),
    "marl_env_step": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.POINTER(MarlStepIO),
                                C.c_void_p]),
    "marl_env_step_timed": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.POINTER(MarlStepIO),
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "marl_env_reset": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.POINTER(MarlResetIO),
                                 C.c_void_p]),
    "marl_adjacency_bits": (C.c_int, [C.POINTER(MarlEnvParams), C.c_int32, _pd, _pd, _pi, _pu32, C.c_void_p]),
    "marl_jam_mask_bits": (C.c_int, [C.POINTER(MarlEnvParams), C.c_int32, _pd, _pd, _pi, _pd, _pd, _pi, _pu32,
                                     C.c_void_p]),
    "marl_lambda2_from_bits": (C.c_int, [C.c_int32, C.c_int32, _pu32, _pd, C.c_int32, C.c_void_p, C.c_void_p]),
    "marl_lambda2_from_bits_static": (C.c_int, [C.c_int32, C.c_int32, _pu32, _pd, C.c_int32, C.c_void_p]),
    "marl_lambda2_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "marl_dbscan": (C.c_int, [C.POINTER(MarlEnvParams), C.c_int32, _pd, _pd, _pi, _pd, _pd, _pi, _pi, _pi,
                              C.c_void_p]),
    "marl_gae": (C.c_int, [C.c_int32, C.c_int32, _pf, _pf, _pf, _pf, _pf, C.c_double, C.c_double, _pf, _pf,
                           C.c_void_p]),
    "marl_reward_per_agent": (C.c_int, [C.POINTER(MarlEnvParams), C.c_int32, _pd, _pd, _pi, _pd, _pd, _pi, _pf, _pf,
                                        _pd, _pd, _pi, _pi, _pi, _pf, C.c_void_p]),
    "marl_reward_obs": (C.c_int, [C.POINTER(MarlEnvParams), C.c_int32, C.POINTER(MarlRewardObsIO), C.c_void_p]),
    "marl_policy_weight_count": (C.c_size_t, [C.c_int32]),
    "marl_policy_act": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                  C.POINTER(MarlPolicyIO), C.c_void_p]),
    "marl_ppo_weight_count": (C.c_size_t, [C.c_int32]),
    "marl_ppo_sum_rows": (C.c_int32, []),
    "marl_ppo_partial_rows": (C.c_int32, []),
    "marl_ppo_partial_floats": (C.c_size_t, [C.c_int32]),
    "marl_ppo_products": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(MarlPpoIO), _pf, C.c_void_p]),
    "marl_ppo_reduce": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, _pf, _pf, _pf, C.c_void_p]),
    "marl_ppo_rows": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(MarlPpoIO), C.c_void_p]),
    "marl_ppo_apply": (C.c_int, [C.POINTER(MarlPpoApply), C.c_void_p]),
    "marl_fill_noise": (C.c_int, [C.POINTER(MarlEnvParams), C.POINTER(MarlEnvState), C.c_uint64, C.c_uint64,
                                  _pd, C.c_void_p]),
}

_lib = None


class MarlError(RuntimeError):
    pass


def load():
    """Load libmarl_env.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MarlError(
            f"{LIB_PATH} not found: the HIP library is the only implementation of the env step; "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=""):
    if code != MARL_OK:
        lib = load()
        msg = lib.marl_strerror(code).decode()
        if code == MARL_EHIP:
            msg += f" (hipError_t {lib.marl_last_hip_error()})"
        raise MarlError(f"{what}: {msg}" if what else msg)


def normalise_device(device):
    """torch.device with an explicit index: "cuda" becomes the current device, so that comparisons
    with tensor.device (which always carries the index) hold."""
    import torch
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d
