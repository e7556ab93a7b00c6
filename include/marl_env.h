/*
 * marl_env.h -- C ABI of libmarl_env.so: batched JammerEnv step on MI355X (gfx950).
 *
 * The reference (mrutyunjaykumarrao/MARL) is pure Python and has no FFI layer; the
 * interface each entry point replaces is therefore a Python method or function
 * of the reference, cited as file:line under /root/reference/src.  All device
 * pointers are plain HIP device pointers owned by the caller; every call is
 * asynchronous on the `stream` argument (a hipStream_t passed as void*), does
 * no host synchronisation, and returns 0 on success or a negative MARL_E* code
 * (marl_strerror gives the text).
 *
 * Layout of the batched state (E environments, N enemies, M jammers), all
 * row-major and contiguous, struct-of-arrays so that the lanes of one
 * wavefront read consecutive addresses:
 *
 *   enemy_x, enemy_y, enemy_vx, enemy_vy   double [E][N]    EnemySwarm.positions / .velocities
 *   jam_x, jam_y                           double [E][M]    JammerEnv.jammer_positions
 *   jam_band                               int32  [E][M]    JammerEnv.jammer_bands
 *   last_vx, last_vy                       float  [E][M]    JammerEnv.last_velocities (raw action)
 *   step_count, enemy_band                 int32  [E]
 *   lambda2_initial                        double [E]
 *   labels                                 int32  [E][N]    DBSCANClusterer.labels
 *   cent_x, cent_y, cent_size              [E][N+1]         centroids / cluster_sizes dicts in dict
 *                                                            order: clusters 0..n_clusters-1, then the
 *                                                            noise entry (id -1) when present
 *   n_clusters, ncent                      int32  [E]       clusters (ids >= 0) / dict entries
 *   prev_x, prev_y, prev_n_clusters, prev_ncent             DBSCANClusterer._prev_centroids
 *   assign                                 int32  [E][M]    jammer_assignments: cluster id per jammer,
 *                                                            -1 noise pseudo-cluster, -2 none
 *   workspace                              double [E][ld][ld] scratch of the lambda2 solver, N > 64 only
 *   solve_*                                scratch that carries an env from the front half of a step to its
 *                                          eigen-solve, N <= 64 only (see MarlEnvState)
 */
#ifndef MARL_ENV_H
#define MARL_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MARL_NUM_BANDS 4
#define MARL_OBS_DIM 5
#define MARL_ASSIGN_NONE (-2)
#define MARL_MAX_N 256          /* enemies per env supported by the kernels */
#define MARL_MAX_N_LAMBDA2 1024 /* vertices per graph supported by marl_lambda2_from_bits alone */
#define MARL_MAX_M 64           /* jammers per env */
#define MARL_ENEMY_RANDOM_WALK 0        /* EnemySwarm._step_random_walk, enemy_swarm.py:157-189 */
#define MARL_ENEMY_COORDINATED_TURN 1   /* EnemySwarm._step_coordinated_turn, enemy_swarm.py:191-242 */
#define MARL_SCHEDULE_AUTO 0            /* fused for N <= 16 or fewer than 1024 envs, split otherwise */
#define MARL_SCHEDULE_FUSED 1
#define MARL_SCHEDULE_SPLIT 2
/* Solver selectors, added to a `rounds` / lambda2_rounds value (at most one of them).  lambda2 = N - mu_max of the
 * complement graph's Laplacian; mu_max comes from a direct solver (Householder tridiagonalisation + Sturm
 * multisection) or, for eigenproblems of 33 to 64 vertices, from a Lanczos iteration that ends when the residual
 * bound on its largest Ritz value is below 1e-6 max(1, lambda2) (the direct solver takes over if it does not get
 * there in 64 steps).  With neither selector every entry point takes the direct solver up to 64 vertices (after the
 * reduction of the problems -- twin vertices merged, small components dropped -- it is the faster one at every batch
 * size measured); beyond, marl_env_step takes the iteration on the bit rows and marl_lambda2_from_bits the direct
 * (team) solver. */
#define MARL_LAMBDA2_DIRECT 0x100
#define MARL_LAMBDA2_LANCZOS 0x200
#define MARL_LAMBDA2_TEST_FEW_STEPS 0x400 /* testing, marl_lambda2_from_bits with MARL_LAMBDA2_LANCZOS at N > 64: the iteration
                                           gives up after 24 steps, so that what follows a failure (the team solver's list,
                                           the solve in place once that is full) can be exercised */
#define MARL_STAGE_NO_RECLUSTER 1        /* marl_env_step leaves the K_recompute re-clustering (jammer_env.py:394-402) out:
                                           the cluster table of the last reset stays; with MarlStepIO.obs = NULL this is the
                                           "step + reward only" form of the step */
#define MARL_STAGE_NO_SOLVE_PRIORITY 2    /* the split schedule's eigen-solves leave the wave priority alone (by default, for
                                           batches that the chip holds in about one round of solve waves, a solve's issue
                                           priority follows the work it has left, so that the solves of a SIMD finish together);
                                           marl_env_step_parts sets it for its parts */
#define MARL_STAGE_GENERAL_RESET 4        /* marl_env_reset takes its general kernel even where the un-jammed graph is known to be
                                           complete (generated spawns in an arena whose diagonal every band's communication
                                           range exceeds: lambda2_initial = N without adjacency test or eigen-solve); testing */

enum {
    MARL_OK = 0,
    MARL_EINVAL = -1,           /* bad argument (null pointer, size out of range) */
    MARL_EUNSUPPORTED = -2,     /* N or M beyond what the kernels support */
    MARL_EHIP = -3,             /* a HIP runtime call failed; see marl_last_hip_error */
    MARL_ETOOBIG = -4           /* marl_lambda2_from_bits_static: more scratch needed than the static buffer holds */
};

/* Constructor arguments of JammerEnv (environment/jammer_env.py:94-127) with the
 * powers already converted to Watts (physics/fspl.py:50-69), the constants the
 * reference fixes elsewhere (kappa reward.py:77, noise_std enemy_swarm.py:61,
 * drift_threshold jammer_env.py:398, spread jammer_env.py:335), and the values
 * marl_params_init derives from them. */
typedef struct MarlEnvParams {
    int32_t N, M;
    int32_t max_steps, K_recompute, min_samples;
    int32_t random_jammer_start;
    int32_t max_reset_attempts;
    int32_t lambda2_rounds;          /* multisection rounds of the eigenvalue search (0 = default), + a MARL_LAMBDA2_* selector */
    int32_t enemy_mode;              /* MARL_ENEMY_RANDOM_WALK or MARL_ENEMY_COORDINATED_TURN (jammer_env.py:103) */
    int32_t step_schedule;           /* N <= 64: MARL_SCHEDULE_AUTO, _FUSED (one launch, solve inside the wave that
                                        stepped the env) or _SPLIT (three launches, longest solve first) */
    double arena_size, dt, v_max, v_enemy;
    double P_tx_w, P_sens_w, P_jam_w, P_thr_w;
    double omega[5];
    double kappa, eps, drift_threshold, spread, noise_std, min_lambda2_initial;
    double omega_max;                /* turn-rate bound of the coordinated-turn model (enemy_swarm.py:60) */
    /* ---- derived by marl_params_init ---- */
    double comm_d2_max[MARL_NUM_BANDS];  /* largest squared distance with P_R >= P_sens, per enemy band */
    double jam_d2_max[MARL_NUM_BANDS];   /* largest squared jammer-midpoint distance with P_jam >= P_thr */
    double R_jam[MARL_NUM_BANDS];        /* compute_jam_range per enemy band */
    float dt_f32, v_max_f32, v_max_sq_f32;
    float jam_lo_f32[MARL_NUM_BANDS], jam_hi_f32[MARL_NUM_BANDS];   /* float32 bracket of 4 jam_d2_max for the step's
                                            first-stage test: below lo the exact test holds, above hi it fails */
    /* ---- input again (kept here so that the layout of the fields above does not move) ---- */
    int32_t stage_flags;                 /* MARL_STAGE_* bits; 0 = the full step */
} MarlEnvParams;

typedef struct MarlEnvState {
    int32_t E;
    int32_t pad;
    double *enemy_x, *enemy_y, *enemy_vx, *enemy_vy;
    double *enemy_omega;       /* [E][N] turn rate per drone (coordinated turn; unused by the random walk) */
    double *jam_x, *jam_y;
    int32_t *jam_band;
    float *last_vx, *last_vy;
    int32_t *step_count, *enemy_band;
    int32_t *tick;             /* [E] steps taken since the state was created (never reset): keys the on-device
                                  noise and reset draws, so that episodes of one env do not repeat each other */
    double *lambda2_initial;
    int32_t *labels;
    double *cent_x, *cent_y;
    int32_t *cent_size;
    int32_t *n_clusters, *ncent;
    double *prev_x, *prev_y;
    int32_t *prev_n_clusters, *prev_ncent;
    int32_t *assign;
    double *workspace;         /* N > 64 only (NULL otherwise): [E][ld][ld] lambda2 scratch, ld = N rounded up to 8,
                                  then int32 [E] sizes of the eigenproblems left to the second launch of a step
                                  (one 512-thread workgroup per env with 65..208 touched vertices) */
    /* N <= 64: a step runs as three launches.  The front kernel does everything but the eigen-solve and
     * leaves, for the envs whose graph is neither complete nor disconnected, the compacted complement
     * rows; the order kernel sorts those envs by matrix size, largest first; the solve kernel works
     * through them in that order (longest job first keeps the CUs evenly loaded) and finishes the reward.
     * A reset of every env (mask NULL) clears solve_hist; a step leaves it zero again. */
    uint64_t *solve_rows;      /* [E][64] complement adjacency row per compacted vertex */
    uint16_t *solve_meta;      /* [E][64] where twin vertices of the complement were merged (same neighbours: they only
                                  contribute eigenvalues that cannot be the largest): multiplicity | diagonal entry << 8
                                  of every vertex of the quotient, whose rows solve_rows then holds */
    int32_t *solve_np;         /* [E] vertices in the eigenproblem, 0 = lambda2 already final; above bit 7, for a
                                  quotient: n' | largest degree << 8 of the complement before merging */
    int32_t *solve_slot;       /* [E] rank of the env among those with the same solve_np */
    int32_t *solve_perm;       /* [E] env indices, largest eigenproblem first */
    int32_t *solve_hist;       /* [1344] 16 histograms of solve_np (80 words apart, envs spread by index to keep the
                                  atomics uncontended); [1280] = envs to solve in this step */
    double *solve_partial;     /* [E][4] reward terms 2..5 awaiting the lambda2 term (every N) */
    double *episode_return;    /* [E] sum of the rewards of the running episode (training/trainer.py:229-230) */
} MarlEnvState;

/* Inputs and outputs of one step.  Optional pointers may be NULL. */
typedef struct MarlStepIO {
    const float *actions;      /* [E][M][3]  (vx, vy, band) as the (M,3) action array, jammer_env.py:376-378 */
    const double *noise;       /* [E][N][2]  enemy velocity noise draw (enemy_swarm.py:165); coordinated turn:
                                  [e][i][0] = the turn-rate draw of drone i (enemy_swarm.py:228), [e][i][1] unused.
                                  NULL = counter-based generator keyed by (seed, env, step_count) */
    float *obs;                /* [E][M][5]  out; NULL = the observation is not written */
    float *reward;             /* [E]        out */
    uint8_t *truncated;        /* [E]        out: step_count >= max_steps */
    double *lambda2;           /* [E]        out, optional: lambda2_current */
    float *reward_terms;       /* [E][5]     out, optional: unweighted r1..r5 (reward.py:138-153) */
    double *avg_power_dbm;     /* [E]        out, optional: jammer_env.py:510-560 */
    uint32_t *adj_bits;        /* [E][N][ceil(N/32)] out, optional: FSPL adjacency before jamming */
    uint32_t *jam_bits;        /* [E][N][ceil(N/32)] out, optional: disrupted-links mask */
    double *episode_stats;     /* [4] in/out, optional: every episode that ends with this step adds (atomically)
                                  1, its return, its lambda2 reduction in percent and its length
                                  (training/trainer.py:248-268); the caller zeroes it when it starts a rollout */
    uint64_t seed;
    uint64_t env_offset;       /* global index of env 0 of this shard (multi-GPU: rank * E) */
} MarlStepIO;

/* Inputs and outputs of a (masked) reset.  Each init pointer may be NULL, in
 * which case the values are drawn on the device from the counter-based
 * generator keyed by (seed, env, episode) and the retry loop of
 * jammer_env.py:283-316 runs in the kernel. */
typedef struct MarlResetIO {
    const uint8_t *mask;       /* [E] reset env e iff mask[e] != 0; NULL = all */
    const double *enemy_pos;   /* [E][N][2] */
    const double *enemy_vel;   /* [E][N][2] */
    const double *enemy_omega; /* [E][N] coordinated turn only (enemy_swarm.py:133-137); NULL = drawn on the device */
    const int32_t *enemy_band; /* [E] */
    const double *jam_init;    /* [E][M][2] positions (random_jammer_start) or centroid offsets */
    const int32_t *jam_band;   /* [E][M] */
    float *obs;                /* [E][M][5] out (only rows of reset envs are written) */
    uint64_t seed;
    uint64_t episode;          /* distinguishes successive resets of one env in generator mode */
    uint64_t env_offset;
} MarlResetIO;

/* ---- host-side helpers ---------------------------------------------------- */

/* Fills the derived fields of *p.  comm_d2_max / jam_d2_max are found by
 * bisection over doubles of the reference's exact float64 chain
 * P * (lambda / (4*pi*max(sqrt(d2),1e-10)))**2 >= threshold
 * (physics/fspl.py:199-206), so `d2 <= d2_max` reproduces the reference's
 * comparison bit for bit. */
int marl_params_init(MarlEnvParams *p);

/* Bytes of device memory the state of E envs needs, and carving of one
 * caller-owned allocation `base` (256-byte aligned) into the arrays of *s. */
size_t marl_state_bytes(const MarlEnvParams *p, int32_t E);
int marl_state_bind(MarlEnvState *s, const MarlEnvParams *p, int32_t E, void *base);

/* Envs [first, first + count) of a bound state as a state of their own (*out: the same arrays, every per-env
 * pointer moved to row `first`, E = count), so that parts of one batch can be stepped by separate launches on
 * separate HIP streams: the front kernel of one part then runs beside the eigen-solves of another.  The envs of
 * the reference are independent (JammerEnv.step touches self only, environment/jammer_env.py:380-545), so the
 * results do not depend on the split; pass env_offset + first in the IO structs to keep the generator streams.
 * `scratch`: marl_shard_scratch_bytes() of device memory private to the part (the launch-ordering histogram),
 * zeroed once by the caller.  N <= 64 only (larger swarms keep per-batch tables in the workspace):
 * MARL_EUNSUPPORTED otherwise. */
size_t marl_shard_scratch_bytes(void);
int marl_state_shard(const MarlEnvParams *p, const MarlEnvState *whole, int32_t first, int32_t count,
                     void *scratch, MarlEnvState *out);

const char *marl_strerror(int code);
int marl_last_hip_error(void);

/* ---- the hot path --------------------------------------------------------- */

/* JammerEnv.step for E envs in lockstep (environment/jammer_env.py:356-457):
 * action decode, enemy dynamics (random walk or coordinated turn), DBSCAN every K_recompute steps with drift
 * test and proportional reassignment, FSPL adjacency, midpoint jamming mask,
 * Laplacian lambda2, 5-term reward, observation. */
int marl_env_step(const MarlEnvParams *p, const MarlEnvState *s, const MarlStepIO *io, void *stream);

/* Measurement aid (no counterpart in the reference): marl_env_step with two HIP events recorded on `stream`,
 * `solve_begin` before and `solve_end` after the launch that holds the Laplacian eigen-solves (the third launch
 * of the three-launch schedule, the wide-team launch for N > 64, the only launch of the one-launch schedule).
 * Both are hipEvent_t handles created by the caller with timing enabled; their elapsed time is the duration of
 * that launch on an otherwise idle stream. */
int marl_env_step_timed(const MarlEnvParams *p, const MarlEnvState *s, const MarlStepIO *io, void *stream,
                        void *solve_begin, void *solve_end);

/* JammerEnv.reset for the masked envs (environment/jammer_env.py:261-354). */
int marl_env_reset(const MarlEnvParams *p, const MarlEnvState *s, const MarlResetIO *io, void *stream);

/* The same step / reset issued as `parts` launches over consecutive runs of envs (part k: E / parts envs, the first
 * E % parts parts one more), part k on streams[k]: marl_state_shard and the row offsets of the IO arrays done
 * here, one call for the host.  `scratch`: parts * marl_shard_scratch_bytes() of zeroed device memory that stays
 * with this state.  The call orders nothing between the streams and the caller's other work; the caller does.
 * 1 <= parts <= MARL_MAX_PARTS, N <= 64. */
#define MARL_MAX_PARTS 16
int marl_env_step_parts(const MarlEnvParams *p, const MarlEnvState *s, const MarlStepIO *io, int32_t parts,
                        void *const *streams, void *scratch);
int marl_env_reset_parts(const MarlEnvParams *p, const MarlEnvState *s, const MarlResetIO *io, int32_t parts,
                         void *const *streams, void *scratch);

/* ---- separately testable stages -------------------------------------------- */

/* compute_adjacency_matrix (physics/communication_graph.py:41-105) on
 * x,y [E][N], band [E] -> bits [E][N][ceil(N/32)]. */
int marl_adjacency_bits(const MarlEnvParams *p, int32_t E, const double *x, const double *y,
                        const int32_t *band, uint32_t *bits, void *stream);

/* compute_disrupted_links (physics/jamming.py:151-247). */
int marl_jam_mask_bits(const MarlEnvParams *p, int32_t E, const double *x, const double *y,
                       const int32_t *band, const double *jam_x, const double *jam_y,
                       const int32_t *jam_band, uint32_t *bits, void *stream);

/* compute_laplacian + compute_lambda2 (physics/communication_graph.py:184-286)
 * from adjacency bits [E][N][ceil(N/32)] -> lambda2 [E].  N <= MARL_MAX_N_LAMBDA2: beyond MARL_MAX_N -- the range
 * of the reference's sparse branch (communication_graph.py:268-278, the same eigenvalue) -- a stand-alone solver,
 * one workgroup per graph (Householder + Sturm multisection, result within 1e-11 of LAPACK's; `rounds` unused). */
int marl_lambda2_from_bits(int32_t N, int32_t E, const uint32_t *bits, double *lambda2,
                           int32_t rounds, void *workspace, void *stream);

/* Bytes of device scratch marl_lambda2_from_bits needs (0 for N <= 64 and for N > MARL_MAX_N_LAMBDA2; the matrix
 * of the eigenproblem otherwise).  marl_state_bytes already includes the same amount for the step. */
size_t marl_lambda2_workspace_bytes(int32_t N, int32_t E);

/* marl_lambda2_from_bits for N <= MARL_MAX_N (larger graphs: MARL_EUNSUPPORTED) without a workspace argument and
 * without any allocation.  It has no
 * counterpart in the reference: it exists for callers that cannot call hipMalloc (code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation).  The scratch the solver needs for
 * N > 64 is a static device array of MARL_STATIC_SCRATCH_BYTES inside the library (one copy per GPU, held
 * until the process ends); a request with marl_lambda2_workspace_bytes(N, E) beyond that returns
 * MARL_ETOOBIG (32 MiB holds e.g. 104 envs at N = 200, 63 at N = 256 -- the pending list of the second launch
 * takes room beside the matrices --; any E at N <= 64, which needs none).
 * The first call on a device looks the array's address up (make that call outside graph capture).  All
 * calls on one GPU share the array: calls that overlap in time on one GPU (different streams) race. */
#define MARL_STATIC_SCRATCH_BYTES (32u << 20)
int marl_lambda2_from_bits_static(int32_t N, int32_t E, const uint32_t *bits, double *lambda2,
                                  int32_t rounds, void *stream);

/* DBSCANClusterer.fit + get_cluster_sizes (clustering/dbscan_clustering.py:78-144)
 * -> labels [E][N], cent_x/cent_y/cent_size [E][N+1], n_clusters/ncent [E]. */
int marl_dbscan(const MarlEnvParams *p, int32_t E, const double *x, const double *y,
                int32_t *labels, double *cent_x, double *cent_y, int32_t *cent_size,
                int32_t *n_clusters, int32_t *ncent, void *stream);

/* RolloutBuffer.compute_returns_and_advantages before normalisation
 * (agents/rollout_buffer.py:151-204), one lane per env:
 * rewards/values/dones [T][E] float -> advantages/returns [T][E];
 * gamma and gamma*lam are rounded to float once, as numpy does with Python floats. */
int marl_gae(int32_t T, int32_t E, const float *rewards, const float *values, const float *dones,
             const float *last_value, const float *last_done, double gamma, double lam,
             float *advantages, float *returns, void *stream);

/* RewardCalculator.compute_per_agent (environment/reward.py:308-360, with _compute_single_proximity
 * :362-396): per-jammer reward [E][M] float32, the lambda2 and overlap terms shared, band match,
 * proximity and energy the jammer's own; no clipping.  The env's step does not call it.  Tables in the
 * layout of MarlEnvState: cent_x / cent_y [E][N+1], vx / vy = last_vx / last_vy. */
int marl_reward_per_agent(const MarlEnvParams *p, int32_t E, const double *lambda2, const double *lambda2_initial,
                          const int32_t *enemy_band, const double *jam_x, const double *jam_y,
                          const int32_t *jam_band, const float *vx, const float *vy, const double *cent_x,
                          const double *cent_y, const int32_t *n_clusters, const int32_t *ncent,
                          const int32_t *assign, float *rewards, void *stream);

/* RewardCalculator.compute (environment/reward.py:106-176: the five terms and their clipped weighted sum) and
 * ObservationBuilder.build_vectorized (environment/observation.py:264-332) as a stage of their own, on tables in the
 * layout of MarlEnvState (cent_* [E][N+1] in dict order, assign [E][M], vx / vy = last_vx / last_vy); with enemy_x /
 * enemy_y and avg_power_dbm given, also JammerEnv._compute_avg_jamming_power (environment/jammer_env.py:510-560).
 * The step computes the same quantities with the same device functions; this entry point makes them testable
 * against the reference's stage outputs.  Output pointers may be NULL. */
typedef struct MarlRewardObsIO {
    const double *lambda2, *lambda2_initial;     /* [E] */
    const int32_t *enemy_band;                   /* [E] */
    const double *enemy_x, *enemy_y;             /* [E][N], only read for avg_power_dbm */
    const double *jam_x, *jam_y;                 /* [E][M] */
    const int32_t *jam_band;                     /* [E][M] */
    const float *vx, *vy;                        /* [E][M] raw action velocities (float32: jammer_env.py:377-380) */
    const double *cent_x, *cent_y;               /* [E][N+1] */
    const int32_t *cent_size;                    /* [E][N+1] */
    const int32_t *n_clusters, *ncent;           /* [E] */
    const int32_t *assign;                       /* [E][M] */
    float *reward;                               /* [E]       out */
    float *reward_terms;                         /* [E][5]    out: unweighted r1..r5 */
    float *obs;                                  /* [E][M][5] out */
    double *avg_power_dbm;                       /* [E]       out */
} MarlRewardObsIO;
int marl_reward_obs(const MarlEnvParams *p, int32_t E, const MarlRewardObsIO *io, void *stream);

/* ---- policy inference for the rollout loop ---------------------------------- */

/* Inputs and outputs of marl_policy_act.  Optional pointers may be NULL. */
typedef struct MarlPolicyIO {
    const float *weights;      /* packed actor + critic parameters, marl_policy_weight_count(hidden) floats (layout below) */
    const float *obs;          /* [E][M][5]  per-agent observations */
    const int32_t *tick;       /* [E] optional: MarlEnvState.tick, mixed into the draw counter so that a replayed launch
                                  (HIP graph: frozen arguments) still draws fresh actions after every env step */
    float *actions;            /* [E][M][3]  out: (vx, vy) clamped to +-v_max, band as float */
    float *log_probs;          /* [E][M]     out: log-probability of the unclamped draw */
    float *values;             /* [E]        out, optional: critic on the mean-pooled observation */
    uint64_t seed;
    uint64_t env_offset;       /* global index of env 0 of this shard */
    uint32_t call;             /* host-side call counter, part of the draw counter */
    int32_t deterministic;     /* != 0: velocity mean and arg-max band (evaluation) */
} MarlPolicyIO;

/* PPOAgent.get_action for E envs x M agents (agents/ppo_agent.py:193-225): Actor.sample (agents/actor.py:118-202)
 * and Critic.forward_pooled (agents/critic.py:124-134) in one launch, float32.  hidden = 64 or 128.
 * Packed weights (floats; torch layouts: Linear.weight is [out][in]).  Matrices are stored as the A-operand
 * fragments of v_mfma_f32_32x32x2 (output units on the rows of the product, 32 agents on its columns), so that the
 * accumulators of one layer are the B operand of the next without a shuffle.  With
 *     unit(s, h) = 32 (s / 16) + 8 ((s % 16) / 4) + 4 h + s % 4        (s = 0 .. H/2 - 1, h = 0, 1)
 * the hidden unit that a lane of wave half h holds in its s-th accumulator register, and lane l = 0 .. 63:
 *   first(W, b)  [H/32][3][64]      entry (bo, s, l) = [W | b][32 bo + l % 32][2 s + l / 32]   (bias = input 5)
 *   hidden(W)    [H/32][H/8][64][4] entry (bo, g, l, q) = W[32 bo + l % 32][unit(4 g + q, l / 32)]
 *   heads(W)     [H/8][64][4]       as hidden() with one block: rows 0..7 = (mu_head, log_std_head, band_head)
 *                                   stacked, rows 8..31 zero
 *   actor : first(trunk.0), trunk.1.weight [H], trunk.1.bias [H] (LayerNorm), hidden(trunk.3.weight), trunk.3.bias [H],
 *           trunk.4.weight [H], trunk.4.bias [H], heads(...), head biases [8]
 *   critic: first(network.0), hidden(network.2.weight), network.2.bias [H], network.4.weight [H],
 *           network.4.bias [1] + 3 floats of padding
 * marl_amd.agents.pack_policy_weights builds this buffer from the torch modules. */
size_t marl_policy_weight_count(int32_t hidden);
int marl_policy_act(int32_t E, int32_t M, int32_t hidden, float v_max, float log_std_min, float log_std_max,
                    const MarlPolicyIO *io, void *stream);

/* ---- PPO minibatch step ----------------------------------------------------------------------------------
 * PPOAgent.update (agents/ppo_agent.py:296-331) for one minibatch, up to the optimiser.
 *
 * marl_ppo_rows, the per-row pass (one launch per network): forward of actor (Actor.evaluate_actions,
 * agents/actor.py:204-247) and critic (agents/critic.py:124-134, applied to the row's observation as the reference's
 * update does), the losses
 *     actor  = mean(-min(ratio A, clip(ratio, 1 +- eps) A)) - c2 mean(entropy),   critic = c1 mean((V - R)^2),
 * and the backward pass down to every layer's pre-activation, one minibatch row per lane on the matrix cores (the
 * 1/B of the means is applied).  It writes the row matrices whose products are the weight gradients and sums the
 * per-unit gradients (LayerNorm parameters, biases) itself.
 *
 * Row matrices with H columns are stored TILE-MAJOR: 32 rows at a time, each tile as H / 4 chunks of
 * [32 rows][4 units]; element (row r, unit u) is float (r / 32) 32 H + (u / 4) 128 + (r % 32) 4 + u % 4.  All row arrays
 * hold whole tiles: ceil(B / 32) 32 rows (rows beyond B carry zero gradients).
 *
 * marl_ppo_products, the sums over rows, one pass over those arrays (same io; only its out pointers are read):
 * `partial` = [marl_ppo_partial_rows()][marl_ppo_partial_floats(hidden)], one slab per workgroup, to be summed over
 * the first dimension.  A slab:
 *   actor : dz2^T y1  [H][H]  = trunk.3.weight;   dz1^T xe [H][32]: columns 0..4 = trunk.0.weight, 5 = trunk.0.bias;
 *           dh^T y2   [32][H]: rows 0-1 mu_head, 2-3 log_std_head, 4-7 band_head;
 *           dz2^T xe  [H][32]: column 5 = trunk.3.bias;   dh^T xe [32][32]: column 5 of rows 0..7 = the head biases
 *   critic: cdz2^T cy1 [H][H] = network.2.weight; cdz1^T xe [H][32] = network.0 (weight | bias); dv^T cy2 [32][H]: row 0 =
 *           network.4.weight; cdz2^T xe [H][32]: column 5 = network.2.bias; dv^T xe [32][32]: row 0 column 5 = network.4.bias
 * The LayerNorm gradients are actor_sums summed over its first dimension.
 *
 * `stats` (float64 [5], added to): sums over the rows of the actor loss term, c1 (V - R)^2, the entropy, the
 * clip indicator |ratio - 1| > eps and the KL estimate (ratio - 1) - log ratio (ppo_agent.py:333-345).
 * hidden = 64 or 128; weights: marl_ppo_weight_count(hidden) floats, the layout of marl_policy_act's buffer with
 * the fragments of the transposed matrices added (marl_amd.agents.TrainPacker):
 *   actor : first(trunk.0), LN1 weight, bias, hidden(trunk.3.weight), trunk.3.bias, LN2 weight, bias, heads(W), head
 *           biases [8], headsT [H/32][4][64]: entry (b, s, l) = W_heads[2 s + l / 32][32 b + l % 32], hidden(trunk.3.weight^T)
 *   critic: first(network.0), hidden(network.2.weight), network.2.bias, network.4.weight [H], network.4.bias + 3 pad,
 *           hidden(network.2.weight^T) */
typedef struct MarlPpoIO {
    const float *weights;
    const float *obs;            /* [R][5]  flattened rollout buffer (rollout_buffer.py:231-269) */
    const float *actions;        /* [R][3] */
    const float *old_log_probs;  /* [R] */
    const float *advantages;     /* [R] */
    const float *returns;        /* [R] */
    const int64_t *index;        /* [B] rows of the minibatch; NULL = rows 0 .. B-1 */
    float *xe;                   /* [rows][8]  out, row-major: obs, 1, 0, 0 */
    float *y1e, *y2e;            /* [rows][H]  out, tile-major: actor activations after each LayerNorm + ReLU */
    float *dz1, *dz2;            /* [rows][H]  out, tile-major: actor dL/dz per layer */
    float *actor_sums;           /* [marl_ppo_sum_rows()][4 H] out, one row per wave: sums over its minibatch rows of
                                    dL/d(LayerNorm 1 output) (= trunk.1.bias), the same times the normalised z (trunk.1.weight),
                                    and both for LayerNorm 2 (trunk.4.bias, trunk.4.weight) */
    float *dh;                   /* [rows][8]  out, row-major: dL/d(head outputs) */
    float *cy1e, *cy2e;          /* [rows][H]  out, tile-major: critic activations */
    float *cdz1, *cdz2;          /* [rows][H]  out, tile-major: critic dL/dz per layer */
    float *dv;                   /* [rows]     out: dL/dV */
    double *stats;               /* [5] in/out, optional */
    float clip_eps, c1, c2, v_max, log_std_min, log_std_max;
} MarlPpoIO;
size_t marl_ppo_weight_count(int32_t hidden);
int32_t marl_ppo_sum_rows(void);
int marl_ppo_rows(int32_t B, int32_t hidden, const MarlPpoIO *io, void *stream);
int32_t marl_ppo_partial_rows(void);
size_t marl_ppo_partial_floats(int32_t hidden);
int marl_ppo_products(int32_t B, int32_t hidden, const MarlPpoIO *io, float *partial, void *stream);

/* The gradients in the caller's parameter order: out[t] = the sum over the slabs of partial[.][index[t]] when index[t] >= 0,
 * over the rows of actor_sums[.][-index[t] - 1] otherwise (index: device memory, built once from the slab layout above;
 * marl_amd.agents.PPOAgent points every parameter's .grad into `out`, so that the all-reduce, the clipping and Adam of
 * ppo_agent.py:324-331 follow without a copy).  Fixed order of addition: the same input gives the same bits. */
int marl_ppo_reduce(int32_t n_out, const int32_t *index, int32_t hidden, const float *partial, const float *actor_sums,
                    float *out, void *stream);

/* Gradient-norm clip and Adam step of both networks (ppo_agent.py:324-325, 330-331: clip_grad_norm_(max_grad_norm) per
 * network, then torch.optim.Adam with its default settings, betas / eps passed in) on flat float32 vectors of
 * n = n_actor + n_critic elements, the actor's first: the gradients (read only: multiplied by grad_scale -- 1 / world
 * size after a summing all-reduce -- and by the clip factor on the way in, not written back), the moments and the
 * parameters (updated in place).  The four vectors are 16-byte aligned and hold ceil(n / 4) 4 floats, the padding zero.
 * steps: one float32 counter per parameter tensor as torch.optim keeps them, tensors_actor then tensors_critic of
 * them, each incremented; steps[0] and steps[tensors_actor] give the two bias corrections.  lr_actor, lr_critic:
 * device scalars.  With n_packed > 0 the updated parameters are then laid out for the next marl_ppo_rows call:
 * packed[j] = 0 where gather[j] == 0, else params[gather[j] - 1].  n <= 65 536 (these networks: 36 233). */
typedef struct MarlPpoApply {
    float *params, *exp_avg, *exp_avg_sq, *steps;
    const float *grads;
    const float *lr_actor, *lr_critic;
    const int32_t *gather;       /* [n_packed] or NULL */
    float *packed;               /* [n_packed] or NULL */
    double beta1, beta2, eps;    /* as torch.optim.Adam holds them */
    int32_t n_actor, n_critic, tensors_actor, tensors_critic, n_packed;
    float max_norm, grad_scale;
} MarlPpoApply;
int marl_ppo_apply(const MarlPpoApply *a, void *stream);

/* The enemy-noise draw the step kernel uses when MarlStepIO.noise is NULL,
 * written out as [E][N][2] for step numbers step_count[e] + 1. */
int marl_fill_noise(const MarlEnvParams *p, const MarlEnvState *s, uint64_t seed,
                    uint64_t env_offset, double *noise, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MARL_ENV_H */
