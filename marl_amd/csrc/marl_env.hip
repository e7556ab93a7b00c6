// libmarl_env.so -- batched JammerEnv step for MI355X (gfx950, wave64).
//
// One wavefront advances one environment (N <= 64 enemies): lane i owns enemy i
// (position, velocity, one 64-bit row of every N x N bit matrix), lanes 0..M-1
// also own jammer j, lanes 0..ncent-1 a cluster centroid.  Workgroups are a
// single wave, so the only synchronisation inside a step is wave-level.
//
// Arithmetic that decides a boolean or integer output (adjacency, jamming mask,
// DBSCAN neighbourhoods, reflections, centroid sums) is float64 in the
// reference's operation order; the file is compiled with -ffp-contract=off so
// that no multiply-add is fused behind its back, and fma() is written out where
// fusing is wanted (the eigensolver).
//
// Reference files restated here (paths under /root/reference/src):
//   environment/jammer_env.py   step :356-457, reset :261-354, _compute_lambda2 :464-508,
//                               _compute_avg_jamming_power :510-560
//   environment/enemy_swarm.py  _step_random_walk :157-189, reset :244-283
//   environment/action_space.py apply_action :214-240, clip_bands :202-212
//   physics/fspl.py             received_power_watts :168-206, compute_jam_range :276-308
//   physics/jamming.py          compute_disrupted_links :151-247
//   physics/communication_graph.py  adjacency :41-105, laplacian :184-212, lambda2 :219-286
//   clustering/dbscan_clustering.py fit :78-128, sizes :130-144, drift :163-204,
//                               assign (proportional) :263-362, initial positions :365-406
//   environment/reward.py       compute :106-306
//   environment/observation.py  build_vectorized :264-332
//   agents/rollout_buffer.py    compute_returns_and_advantages :151-204

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>

#include "marl_env.h"
#include "marl_rng.h"

// Scratch of marl_lambda2_from_bits_static, the entry point for callers that cannot allocate: what
// marl_lambda2_from_bits takes as its `workspace` argument lives here instead.  Every GPU the library
// runs on holds its own copy of this array, at its own address, until the process ends; host code
// reaches it only through the per-device address that static_scratch() looks up once
// (hipGetSymbolAddress) -- the array's name itself is not a device address on the host side.
static __device__ __align__(256) unsigned char scratch[MARL_STATIC_SCRATCH_BYTES];

namespace {

constexpr int WAVE = 64;
constexpr double kSpeedOfLight = 3.0e8;                      // fspl.py:31
__host__ __device__ inline double band_hz(int b) {            // fspl.py:38-43
    return b == 0 ? 4.33e8 : (b == 1 ? 9.15e8 : (b == 2 ? 2.40e9 : 5.80e9));
}
constexpr double kFourPi = 12.566370614359172;               // 4 * np.pi as numpy evaluates it
constexpr double kTwoPi = 6.283185307179586;

thread_local int g_last_hip_error = 0;

// ------------------------------------------------------------------ wave helpers
#define WSYNC() __syncthreads()   /* workgroup == one wave */

// f(integral_constant<0>) ... f(integral_constant<E-1>): a loop whose index is a constant when parsed
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p ? 1 : 0); }

// Sums over the 64 lanes: rotate-and-add inside each row of 16 lanes with DPP moves (no LDS traffic,
// unlike a shuffle butterfly, whose six ds_bpermute round trips cost several hundred cycles), then the
// four row totals through SGPRs.
template <int CTRL>
__device__ __forceinline__ int dpp_mov32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_mov32<0x128>(v);      // row_ror:8
    v += dpp_mov32<0x124>(v);      // row_ror:4
    v += dpp_mov32<0x122>(v);      // row_ror:2
    v += dpp_mov32<0x121>(v);      // row_ror:1
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(dpp_mov32<0x128>(__float_as_int(v)));
    v += __int_as_float(dpp_mov32<0x124>(__float_as_int(v)));
    v += __int_as_float(dpp_mov32<0x122>(__float_as_int(v)));
    v += __int_as_float(dpp_mov32<0x121>(__float_as_int(v)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}
__device__ __forceinline__ double wave_sum(double v);      // below, next to the 64-bit DPP helper

using namespace marl_rng;

// Two independent N(0, sigma) draws for enemy `i` of env `genv` at step `step`.
__device__ __forceinline__ void noise_pair(uint64_t seed, uint64_t genv, uint32_t step, uint32_t i,
                                           double sigma, double& nx, double& ny) {
    // counter = (env, step's low 24 bits | drone), key = seed mixed with the stream id and the bits that do not fit
    uint32_t r[2];
    philox2x32((uint32_t)genv, (step << 8) | (i & 0xFFu),
               (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u) ^ ((uint32_t)(genv >> 32) * 0x85EBCA6Bu) ^
                   ((step >> 24) * 0xC2B2AE35u) ^ ((uint32_t)STREAM_NOISE << 28), r);
    // Box-Muller with the hardware's single-precision log2 / sin / cos (v_log_f32, v_sin_f32 and
    // v_cos_f32 take their argument in revolutions): these draws have no counterpart in the reference,
    // only their distribution matters, and seven good digits of it are plenty.
    const float u1 = u01f(r[0]), u2 = u01f(r[1]);                              // u1 >= 2^-25 > 0
    const double rad = (double)__builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1)) * sigma;   // -2 ln 2 * log2
    const double s = (double)__builtin_amdgcn_sinf(u2), c = (double)__builtin_amdgcn_cosf(u2);
    nx = rad * c; ny = rad * s;
}

// Counter word that tells one reset of an env from another in generator mode: the caller's episode
// number mixed with the env's step clock (steps taken since the state was created; never reset), so
// that a replayed launch sequence (HIP graph) still draws fresh values.
__device__ __forceinline__ uint32_t episode_key(const MarlResetIO& io, const MarlEnvState& S, int e) {
    return (uint32_t)io.episode ^ ((uint32_t)S.tick[e] * 0x9E3779B1u);
}

// np.mod(x, a) for a > 0 as numpy computes it for float64 (npy_divmod): C fmod, moved into [0, a).
__device__ __forceinline__ double numpy_mod(double x, double a) {
    double r = fmod(x, a);
    if (r < 0.0) r += a;
    else if (r == 0.0) r = 0.0;          // -0.0 -> +0.0
    return r;
}

// One drone's move given its state and its noise draw(s): the random walk (:157-189) or the
// coordinated-turn model (:191-242).  Pure arithmetic; `w` is the turn rate (coordinated turn only).
__device__ __forceinline__ void enemy_advance(const MarlEnvParams& P, double& x, double& y, double& vx, double& vy,
                                              double& w, double nx, double ny) {
    const double arena = P.arena_size;
    if (P.enemy_mode == MARL_ENEMY_COORDINATED_TURN) {
        const double dt = P.dt;
        double x1, y1, vx1, vy1;
        if (fabs(w) < 1e-6) {
            x1 = x + vx * dt; y1 = y + vy * dt; vx1 = vx; vy1 = vy;
        } else {
            double sn, cs;
            sincos(w * dt, &sn, &cs);
            x1 = x + (sn / w) * vx - ((1 - cs) / w) * vy;
            vx1 = cs * vx - sn * vy;
            y1 = y + ((1 - cs) / w) * vx + (sn / w) * vy;
            vy1 = sn * vx + cs * vy;
        }
        w = fmin(fmax(w + nx, -P.omega_max), P.omega_max);
        x = numpy_mod(x1, arena); y = numpy_mod(y1, arena); vx = vx1; vy = vy1;
    } else {
        vx = vx + nx; vy = vy + ny;
        double sp = sqrt(vx * vx + vy * vy);
        sp = fmax(sp, 1e-6);
        vx = vx / sp * P.v_enemy;
        vy = vy / sp * P.v_enemy;
        x = x + vx * P.dt; y = y + vy * P.dt;
        if (x < 0.0) { x = -x; vx = -vx; }
        if (x > arena) { x = 2 * arena - x; vx = -vx; }
        if (y < 0.0) { y = -y; vy = -vy; }
        if (y > arena) { y = 2 * arena - y; vy = -vy; }
        x = fmin(fmax(x, 0.0), arena);
        y = fmin(fmax(y, 0.0), arena);
    }
}

// EnemySwarm.step for drone i of env e (enemy_swarm.py:139-155): loads its state, moves it, stores it,
// returns the position.  `tick` keys the generated noise.
__device__ __forceinline__ void enemy_move(const MarlEnvParams& P, const MarlEnvState& S, const MarlStepIO& io,
                                           int e, int i, int tick, double& xo, double& yo) {
    const size_t o = (size_t)e * P.N + i;
    const bool ct = P.enemy_mode == MARL_ENEMY_COORDINATED_TURN;
    double nx, ny;
    if (io.noise) { nx = io.noise[o * 2]; ny = io.noise[o * 2 + 1]; }
    else noise_pair(io.seed, io.env_offset + e, (uint32_t)tick, (uint32_t)i, ct ? P.noise_std * 0.5 : P.noise_std, nx, ny);
    double x = S.enemy_x[o], y = S.enemy_y[o], vx = S.enemy_vx[o], vy = S.enemy_vy[o];
    double w = ct ? S.enemy_omega[o] : 0.0;
    enemy_advance(P, x, y, vx, vy, w, nx, ny);
    if (ct) S.enemy_omega[o] = w;
    S.enemy_x[o] = x; S.enemy_y[o] = y; S.enemy_vx[o] = vx; S.enemy_vy[o] = vy;
    xo = x; yo = y;
}

// EnemySwarm.reset for drone i of env e (enemy_swarm.py:244-283 and, for the coordinated turn,
// :119-137): host-fed values where given, else uniform position in [0.1A, 0.9A], uniform heading,
// uniform turn rate in [-omega_max/2, omega_max/2] from the counter-based generator.
__device__ __forceinline__ void enemy_spawn(const MarlEnvParams& P, const MarlEnvState& S, const MarlResetIO& io,
                                            int e, int i, int attempt, double& xo, double& yo) {
    const size_t o = (size_t)e * P.N + i;
    const double arena = P.arena_size;
    const bool ct = P.enemy_mode == MARL_ENEMY_COORDINATED_TURN;
    const bool host_init = io.enemy_pos != nullptr;
    const uint64_t genv = io.env_offset + e;
    const uint32_t k0 = (uint32_t)io.seed, k1 = (uint32_t)(io.seed >> 32);
    uint32_t r[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
    if (!host_init || (ct && io.enemy_omega == nullptr)) {
        philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)i << 8) ^ STREAM_RESET_ENEMY,
                   episode_key(io, S, e), (uint32_t)attempt, k0, k1, r);
        philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)i << 8) ^ STREAM_RESET_ENEMY,
                   episode_key(io, S, e), (uint32_t)attempt | 0x80000000u, k0, k1, r2);
    }
    double x, y, vx, vy;
    if (host_init) {
        x = io.enemy_pos[o * 2]; y = io.enemy_pos[o * 2 + 1];
        vx = io.enemy_vel[o * 2]; vy = io.enemy_vel[o * 2 + 1];
    } else {
        const double margin = arena * 0.1;
        x = margin + (arena - margin - margin) * u01(r[0], r[1]);
        y = margin + (arena - margin - margin) * u01(r[2], r[3]);
        double sn, cs;
        sincos(kTwoPi * u01(r2[0], r2[1]), &sn, &cs);
        vx = P.v_enemy * cs; vy = P.v_enemy * sn;
    }
    if (ct) {
        const double lo = -P.omega_max * 0.5, hi = P.omega_max * 0.5;
        S.enemy_omega[o] = io.enemy_omega ? io.enemy_omega[o] : lo + (hi - lo) * u01(r2[2], r2[3]);
    }
    S.enemy_x[o] = x; S.enemy_y[o] = y; S.enemy_vx[o] = vx; S.enemy_vy[o] = vy;
    xo = x; yo = y;
}

// ------------------------------------------------------------------ LDS layout of one env
// 5 120 bytes up to the solver's part: 32 envs per compute unit (160 KB), eight waves per SIMD.  A cluster table has
// at most 64 entries for 64 points (min_samples = 1: every point a cluster and none noise; otherwise clusters of two or
// more, and one entry for the noise).  The previous centroids, which only a re-clustering step looks at, lie in the
// storage of two arrays that are written after it.
struct WaveLds {
    double2 sxy[WAVE];                // enemy positions (x, y): one 16-byte read per enemy in the pair loops
    union {
        struct {
            float sxf[WAVE], syf[WAVE];   // the same in float32, x and y apart (pairs of neighbours are read as one packed
                                          // operand); entries beyond the swarm hold 1e30: first-stage jamming tests of the step
        };
        double py[WAVE];              // previous centroids (until the float32 copies are written)
    };
    double jx[WAVE], jy[WAVE];        // jammer positions
    union {
        unsigned long long xfer[WAVE];    // lane-to-lane hand-over of 64-bit rows
        double px[WAVE];                  // previous centroids (until the complement rows are compacted)
        double scale[WAVE];               // column scales of a quotient by twin classes (the eigen-solver, after the hand-over)
    };
    double cx[WAVE], cy[WAVE];        // centroids in dict order
    int csz[WAVE];
    int lab[WAVE];
    int asg[WAVE];
    int jb[WAVE];
    // eigen-solver only (kernels that do not solve allocate the struct up to here: front_lds_bytes)
    double2 de[WAVE];                 // tridiagonal: (diagonal, squared off-diagonal) per row
};

inline size_t front_lds_bytes() { return offsetof(WaveLds, de); }
inline size_t wave_lds_bytes(int) { return sizeof(WaveLds); }

// ------------------------------------------------------------------ pairwise bit rows
// Row `lane` of the FSPL adjacency (communication_graph.py:41-105): bit j set iff
// (xi-xj)^2 + (yi-yj)^2 <= d2max, diagonal clear.
__device__ __forceinline__ uint64_t adjacency_row(const WaveLds& L, int n, int lane, double d2max) {
    uint64_t row = 0;
    const double xi = L.sxy[lane & (WAVE - 1)].x, yi = L.sxy[lane & (WAVE - 1)].y;
    for (int j = 0; j < n; ++j) {
        const double dx = xi - L.sxy[j].x, dy = yi - L.sxy[j].y;
        const double d2 = dx * dx + dy * dy;
        row |= (uint64_t)(d2 <= d2max) << j;
    }
    row &= ~(1ull << lane);
    return lane < n ? row : 0ull;
}

// Row `lane` of the disrupted-links mask (jamming.py:151-247): bit j set iff some
// jammer k on the enemy band has |q_k - (p_i + p_j)/2|^2 <= d2max; diagonal clear.
__device__ __forceinline__ uint64_t jam_row(const WaveLds& L, int n, int m, int lane, int eband, double d2max) {
    uint64_t row = 0;
    const double xi = L.sxy[lane & (WAVE - 1)].x, yi = L.sxy[lane & (WAVE - 1)].y;
    for (int k = 0; k < m; ++k) {
        if (L.jb[k] != eband) continue;          // wave-uniform
        const double qx = L.jx[k], qy = L.jy[k];
        for (int j = 0; j < n; ++j) {
            const double mx = (xi + L.sxy[j].x) / 2, my = (yi + L.sxy[j].y) / 2;
            const double dx = qx - mx, dy = qy - my;
            const double d2 = dx * dx + dy * dy;
            row |= (uint64_t)(d2 <= d2max) << j;
        }
    }
    row &= ~(1ull << lane);
    return lane < n ? row : 0ull;
}

// Both rows in one pass over the enemies (step kernel; needs L.sxy).  The jamming test is evaluated
// in a scaled form that rounds exactly like the reference's: with s = fl(xi + xj),
//   fl(qx - s/2) = fl(2qx - s) / 2,   so   fl(dx^2 + dy^2) <= d2max  <=>  fl(DX^2 + DY^2) <= 4 d2max
// for DX = fl(2qx - s) (halving and doubling are exact), which shares s between the jammers and
// drops two multiplications per pair.  KJ = matching jammers handled in this pass (their doubled
// coordinates in q2x / q2y); ADJ = also build the adjacency row.
template <int KJ, bool ADJ>
__device__ __forceinline__ void pair_pass(const WaveLds& L, int n, int lane, double thr_comm, double thr_jam4,
                                          const double (&q2x)[4], const double (&q2y)[4], uint64_t& adj, uint64_t& jam) {
    const double2 pi = L.sxy[lane & (WAVE - 1)];
    uint32_t a_lo = 0, a_hi = 0, j_lo = 0, j_hi = 0;
    auto body = [&](int j, uint32_t bit, uint32_t& a, uint32_t& jm) {
        const double2 pj = L.sxy[j];
        if constexpr (ADJ) {
            const double dx = pi.x - pj.x, dy = pi.y - pj.y;
            a |= (dx * dx + dy * dy <= thr_comm) ? bit : 0u;
        }
        if constexpr (KJ > 0) {
            const double sx = pi.x + pj.x, sy = pi.y + pj.y;
            bool hit = false;
            static_for<0, KJ>([&](auto k) {
                const double DX = q2x[decltype(k)::value] - sx, DY = q2y[decltype(k)::value] - sy;
                hit |= (DX * DX + DY * DY <= thr_jam4);
            });
            jm |= hit ? bit : 0u;
        }
    };
    const int n_lo = n < 32 ? n : 32;
    for (int j = 0; j < n_lo; ++j) body(j, 1u << j, a_lo, j_lo);
    for (int j = 32; j < n; ++j) body(j, 1u << (j - 32), a_hi, j_hi);
    const uint64_t keep = (lane < n) ? ~(1ull << lane) : 0ull;
    if constexpr (ADJ) adj = (((uint64_t)a_hi << 32) | a_lo) & keep;
    jam |= (((uint64_t)j_hi << 32) | j_lo) & keep;
}

// First stage of the jamming test in float32 (twice the float64 rate on this GPU): per pair the squared
// doubled distance to up to KJ jammers against two thresholds that bracket the exact one by more than the
// float32 evaluation can be off (marl_params_init derives them from the arena size: positions inside a step
// lie in [0, A]^2).  `sure` = within the lower threshold (the exact test holds), `maybe` = within the upper one;
// a pair that is `maybe` for some jammer and `sure` for none is settled afterwards by the exact float64
// expression, so float32 never decides a bit.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KJ>
__device__ __forceinline__ void jam_pass_f32(const WaveLds& L, int n, int lane, float thr_lo, float thr_hi,
                                             const float (&q2x)[4], const float (&q2y)[4], uint64_t& sure, uint64_t& maybe) {
    // Two enemies j, j + 1 per iteration on the packed float32 pipe (v_pk_add / mul / fma: the same IEEE operations as
    // one at a time).  Per pair of points the smallest squared distance over the jammers decides; its tests are the
    // sign bits of d2 - t for t = the float after each threshold (d2 < t <=> d2 <= threshold), shifted into the
    // row words by one v_alignbit each - in reverse order, turned round at the end.
    const float xi = L.sxf[lane & (WAVE - 1)], yi = L.syf[lane & (WAVE - 1)];
    const v2f xi2 = {xi, xi}, yi2 = {yi, yi};
    const float tl = nextafterf(thr_lo, INFINITY), th = nextafterf(thr_hi, INFINITY);
    const v2f tl2 = {tl, tl}, th2 = {th, th};
    auto word = [&](int j0, int cnt, uint32_t& sb, uint32_t& mb) {      // cnt even, 2 .. 32
        uint32_t sw = 0, mw = 0;
        for (int j = j0; j < j0 + cnt; j += 2) {
            const v2f xj = *reinterpret_cast<const v2f*>(&L.sxf[j]), yj = *reinterpret_cast<const v2f*>(&L.syf[j]);
            const v2f sx = xi2 + xj, sy = yi2 + yj;
            v2f dmin = {0.0f, 0.0f};
            static_for<0, KJ>([&](auto k) {
                const v2f qx = {q2x[decltype(k)::value], q2x[decltype(k)::value]};
                const v2f qy = {q2y[decltype(k)::value], q2y[decltype(k)::value]};
                const v2f DX = qx - sx, DY = qy - sy;
                const v2f d2 = __builtin_elementwise_fma(DX, DX, DY * DY);
                if constexpr (decltype(k)::value == 0) dmin = d2;
                else dmin = v2f{fminf(dmin.x, d2.x), fminf(dmin.y, d2.y)};
            });
            const v2f a = dmin - tl2, b = dmin - th2;
            sw = __builtin_amdgcn_alignbit(sw, __float_as_uint(a.x), 31);
            sw = __builtin_amdgcn_alignbit(sw, __float_as_uint(a.y), 31);
            mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(b.x), 31);
            mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(b.y), 31);
        }
        sb |= __brev(sw) >> (32 - cnt);
        mb |= __brev(mw) >> (32 - cnt);
    };
    uint32_t s_lo = 0, s_hi = 0, m_lo = 0, m_hi = 0;
    const int n_lo = n < 32 ? ((n + 1) & ~1) : 32;
    word(0, n_lo, s_lo, m_lo);
    if (n > 32) word(32, (n - 32 + 1) & ~1, s_hi, m_hi);
    sure |= ((uint64_t)s_hi << 32) | s_lo;
    maybe |= ((uint64_t)m_hi << 32) | m_lo;
}

// Adjacency row (communication_graph.py:41-105) and disrupted-links row (jamming.py:151-247) of
// enemy `lane` inside a step.
//
// Every enemy lies in [0, A]^2 there (clip of the random walk, modulo of the coordinated turn), so when the
// communication range exceeds the arena's diagonal the adjacency is the complete graph and its test is
// skipped (the presets with A = 200 m all take that path: the range at 20 dBm / -90 dBm is 1.3 km at
// 5.8 GHz); otherwise it is one exact float64 pass.  The jamming mask goes through the float32 first stage
// above, four matching jammers per pass, and the exact float64 form of pair_pass for the pairs it leaves open.
__device__ __forceinline__ void pair_rows(const MarlEnvParams& P, const WaveLds& L, int n, int m, int lane, int eband,
                                          uint64_t& adj, uint64_t& jam) {
    const double arena = P.arena_size, d2_comm = P.comm_d2_max[eband], d2_jam = P.jam_d2_max[eband];
    const bool all_linked = d2_comm >= 2.000001 * arena * arena;
    const uint64_t active = (n >= 64) ? ~0ull : ((1ull << n) - 1);
    const uint64_t keep = (lane < n) ? (active & ~(1ull << lane)) : 0ull;
    const double no_q[4] = {1e300, 1e300, 1e300, 1e300};
    adj = keep;
    jam = 0;
    if (!all_linked) {
        uint64_t none = 0;
        pair_pass<0, true>(L, n, lane, d2_comm, 0.0, no_q, no_q, adj, none);
    }
    const uint64_t matching = ballot64(lane < m && L.jb[lane & (WAVE - 1)] == eband);
    if (matching == 0 || d2_jam < 0.0) return;
    const float thr_lo = P.jam_lo_f32[eband], thr_hi = P.jam_hi_f32[eband];
    uint64_t sure = 0, maybe = 0, mm = matching;
    while (mm) {
        float q2x[4], q2y[4];
        int cnt = 0;
        static_for<0, 4>([&](auto k) {
            float x = 1e30f, y = 1e30f;              // squares overflow to +inf: never within range
            if (mm) {
                const int j = __ffsll((unsigned long long)mm) - 1;
                mm &= mm - 1;
                x = (float)(2.0 * L.jx[j]); y = (float)(2.0 * L.jy[j]);
                ++cnt;
            }
            q2x[decltype(k)::value] = x; q2y[decltype(k)::value] = y;
        });
        switch (cnt) {
            case 1: jam_pass_f32<1>(L, n, lane, thr_lo, thr_hi, q2x, q2y, sure, maybe); break;
            case 2: jam_pass_f32<2>(L, n, lane, thr_lo, thr_hi, q2x, q2y, sure, maybe); break;
            case 3: jam_pass_f32<3>(L, n, lane, thr_lo, thr_hi, q2x, q2y, sure, maybe); break;
            default: jam_pass_f32<4>(L, n, lane, thr_lo, thr_hi, q2x, q2y, sure, maybe); break;
        }
    }
    jam = sure & keep;
    uint64_t open = maybe & ~sure & keep;
    if (ballot64(open != 0) == 0) return;
    // exact evaluation of the pairs left open (rare: a band of a few parts per million around the radius)
    const double thr4 = 4.0 * d2_jam;
    const double2 pi = L.sxy[lane & (WAVE - 1)];
    while (open) {
        const int j = __ffsll((unsigned long long)open) - 1;
        open &= open - 1;
        const double2 pj = L.sxy[j];
        const double sx = pi.x + pj.x, sy = pi.y + pj.y;
        bool hit = false;
        for (uint64_t mk = matching; mk; mk &= mk - 1) {
            const int k = __ffsll((unsigned long long)mk) - 1;
            const double DX = 2.0 * L.jx[k] - sx, DY = 2.0 * L.jy[k] - sy;
            hit |= (DX * DX + DY * DY <= thr4);
        }
        if (hit) jam |= 1ull << j;
    }
}

// ------------------------------------------------------------------ lambda2
// Sum over the 64 lanes without LDS traffic: rotate-and-add inside each row of 16
// lanes with DPP moves, then the four row totals through SGPRs.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    // every lane is written (full row and bank masks, rotations have no invalid source), so the
    // previous content of the destination does not matter and needs no initialisation
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int src) {      // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0x128>(v);      // row_ror:8
    v += dpp_mov<0x124>(v);      // row_ror:4
    v += dpp_mov<0x122>(v);      // row_ror:2
    v += dpp_mov<0x121>(v);      // row_ror:1
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }
// The same when only lanes 0..last hold non-zero terms (last wave-uniform): rows of 16 lanes
// beyond `last` are not visited.
__device__ __forceinline__ double wave_sum_dpp_upto(double v, int last) {
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    if (last < 16) return readlane_d(v, 0);
    double t = v + readlane_d(v, 16);            // meaningful in lanes 0..15
    if (last >= 32) {
        t += readlane_d(v, 32);
        t += readlane_d(v, 48);
    }
    return readlane_d(t, 0);
}

// 1/x and sqrt(x) (x > 0) from the hardware seeds (v_rcp_f64 / v_rsq_f64, about 23 good bits)
// and two Newton / Goldschmidt steps: accurate to an ulp or two, a third of the instructions of
// the correctly rounded library versions.  Used only inside the eigensolver, where an ulp is noise.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    double s = x * r, hh = 0.5 * r;
    double e = fma(-hh, s, 0.5);
    s = fma(s, e, s);
    hh = fma(hh, e, hh);
    e = fma(-hh, s, 0.5);
    return fma(s, e, s);
}
// One step only (about 45 bits): enough for the scale factors of a Householder reflector, whose
// relative error e moves the eigenvalues by about e |A| per step.
__device__ __forceinline__ double rcp_45(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double sqrt_45(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double s = x * r;
    return fma(s, fma(-0.5 * r, s, 0.5), s);
}

// Column loops of the register-resident Householder step.  Row j of the matrix lives in
// the registers a[0..NPAD) of lane j, so every index must be a constant already when the
// code is parsed (a loop variable, even of a fully unrollable loop, makes the compiler
// keep the array in scratch memory): loops are expanded by template recursion, in blocks
// of a few columns with a wave-uniform exit after the block that holds column l.  Columns
// l+1.. of the last block are no-ops because u and q are zero there.
//
// The broadcasts of u and q come from registers, not from LDS.  u (and q) are held
// replicated, U[k] of lane t = u[16 k + t % 16], so that column c = 16 k + i is lane i of every row of
// 16 lanes of U[k]: the DPP control row_newbcast:i of v_fmac_f64_dpp hands exactly that to all
// lanes, inside the multiply-add.  One VALU instruction per column and term, no LDS read.
// The DPP operand must not have been written by the VALU in the two preceding instructions: U and
// Q come from LDS reads, and the first instruction of each chain carries an s_nop.
template <int I, bool FIRST>
__device__ __forceinline__ void fmac_row_bcast(double& d, const double& rowsrc, const double& m) {
    if constexpr (FIRST)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(d) : "v"(rowsrc), "v"(m), "n"(I));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(d) : "v"(rowsrc), "v"(m), "n"(I));
}
// R[k] of lane t = v of lane 16 k + t % 16: the replicated form the row broadcasts read, by the LDS
// crossbar (ds_bpermute: no LDS memory, no barrier).
template <int KB>
__device__ __forceinline__ void replicate_rows(double v, int lane, double (&R)[KB]) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    static_for<0, KB>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int addr = (k * 16 + (lane & 15)) << 2;
        R[k] = __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, hi), __builtin_amdgcn_ds_bpermute(addr, lo));
    });
}
template <int NPAD, int B>
struct ColBlocks {
    static constexpr int BS = 4, C0 = B * BS, KB = (NPAD + 15) / 16;
    // p += A[:, C0..C0+BS-1] * u[C0..C0+BS-1]
    static __device__ __forceinline__ void matvec(const double (&a)[NPAD], const double (&U)[KB], int l,
                                                  double& p0, double& p1) {
        if constexpr (C0 < NPAD) {
            if constexpr (B > 0) {
                if (C0 > l) return;
            }
            static_for<0, BS>([&](auto jj) {
                constexpr int c = C0 + decltype(jj)::value;
                if constexpr (c < NPAD) fmac_row_bcast<c % 16, (c < 2)>((c & 1) ? p1 : p0, U[c / 16], a[c]);
            });
            ColBlocks<NPAD, B + 1>::matvec(a, U, l, p0, p1);
        }
    }
};

#include "householder_update.inc"

// a[i] for a wave-uniform runtime i in [LO, HI): binary search by uniform branches (team solver; the
// wave solver has the pivot column handed back by its update).  Publishing the pivot row through LDS
// instead costs a 16-byte store per column pair for the whole wave, whatever the execution mask.
template <int NPAD, int LO, int HI>
__device__ __forceinline__ double column_of(const double (&a)[NPAD], int i) {
    if constexpr (HI - LO == 1) {
        double v = a[LO];
        asm volatile("" : "+v"(v));      // opaque leaf: stops the compiler from rebuilding a dynamic a[i] (scratch array)
        return v;
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (i < MID) return column_of<NPAD, LO, MID>(a, i);
        return column_of<NPAD, MID, HI>(a, i);
    }
}

// Issue priority of the wave by the work it has left.  The SIMD's arbiter serves the oldest wave first, so of four
// solves that start together the last one is held back until the others are done and then runs alone, at the pace
// of its dependency chains (measured: a wave alone issues an instruction per 11 clocks, the SIMD has room for one
// per 4).  With the priority following the remaining work the waves of a SIMD finish together and the pipe stays
// full to the end.  Work left, in instructions: about 85 + 3 i per Householder step that reduces row i, then 4 per
// term and round of the eigenvalue search; the four levels of the hardware are spaced more finely towards the end,
// where the finishing order is decided.
#ifndef SOLVE_PRIO_T3            /* level thresholds, per cent of the largest solve's work */
#define SOLVE_PRIO_T3 30
#define SOLVE_PRIO_T2 15
#define SOLVE_PRIO_T1 6
#endif
template <bool ON>
struct SolvePriority {
    int left, level, scale;       // scale: the work of the largest solve, the measure of the levels
    bool boosted = false;
    static __device__ __forceinline__ void set(int lv) {
        if (lv == 3) __builtin_amdgcn_s_setprio(3);
        else if (lv == 2) __builtin_amdgcn_s_setprio(2);
        else if (lv == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
    // instructions of a solve of np vertices, roughly: 85 + 3 i per Householder step that reduces row i, then 4 per
    // term and round of the eigenvalue search
    static __device__ __forceinline__ int work(int np, int rounds) {
        const int i = np - 1;
        return np < 2 ? 0 : 85 * i + (3 * i * i) / 2 + 4 * rounds * np;
    }
    // `total`: the work the wave has before it, this solve included (a wave of the per-CU launch knows its list);
    // `top`: the work of the largest solve the launch can hold
    __device__ __forceinline__ void start(int total, int top) {
        if (boosted) return;
        if constexpr (ON) {
            left = total;
            scale = top > 16 ? top : 16;
            level = 0;
            spend(0);
        }
    }
    // Levels at 30 %, 15 % and 6 % of the top: closer together towards the end, where the finishing order is decided
    // (45 / 24 / 11, 35 / 20 / 10 and 20 / 9 / 3 are within 1 % of it).
    // (Alternating between neighbouring levels from step to step, for a priority that follows the work continuously,
    // slowed every wave down by half: priority changes are not free.)
    // A solve that turns out longer than its size promised (a Lanczos iteration that has not converged at its first
    // check) goes to the top level and stays there, whether or not the launch otherwise schedules by priority: what it
    // has left is the launch's critical path.
    __device__ __forceinline__ void boost() {
        if (!boosted) __builtin_amdgcn_s_setprio(3);
        boosted = true;
        level = 3;
    }
    __device__ __forceinline__ void spend(int cost) {
        if constexpr (!ON) return;
        if (boosted) return;
        left -= cost;
        const int x = left * 100;
        const int lv = x >= SOLVE_PRIO_T3 * scale ? 3 : x >= SOLVE_PRIO_T2 * scale ? 2 : x >= SOLVE_PRIO_T1 * scale ? 1 : 0;
        if (lv != level && (lv < level || cost == 0)) {      // rises only at the start
            level = lv;
            set(lv);
        }
    }
    __device__ __forceinline__ void finish() {
        if (boosted || (ON && level != 0)) __builtin_amdgcn_s_setprio(0);
        level = 0;
        boosted = false;
    }
};

// Number of eigenvalues below x of the symmetric tridiagonal matrix de[k] = (d_k, e_k^2), k < np (e_k couples rows
// k - 1 and k): the sign changes of the Sturm sequence p_0 = 1, p_k = (d_k - x) p_{k-1} - e_k^2 p_{k-2}.
// AHEAD: 16-byte LDS reads in flight, 8 or 4 (4 registers each; the Lanczos solver's checks run beside more live state
// than the direct solver's search and take 4).
// last / last_exp (optional): the last term p_np of the sequence, i.e. the characteristic polynomial at x, as
// last x 2^last_exp.
template <int AHEAD = 8>
__device__ __forceinline__ int sturm_count_below(const double2* de, int np, double x, double* last = nullptr, int* last_exp = nullptr) {
    // The signs of p_1, p_2, ... are shifted into a 32-bit register (one v_alignbit_b32 per term);
    // with `fresh` new bits at the bottom, the sign changes they add are the set bits
    // 0..fresh-1 of B ^ (B >> 1) (bit `fresh` is the sign before them; p_0 = 1 is a zero bit).
    // Flushed before more than 31 are fresh.
    double pm = 1.0, pc = de[0].x - x;
    uint32_t bits = (uint32_t)__double2hiint(pc) >> 31;
    int cnt = 0, fresh = 1, esum = 0;
    auto term = [&](int i) {
        const double2 t = de[i];
        const double pn = fma(t.x - x, pc, -(t.y * pm));
        bits = __builtin_amdgcn_alignbit(bits, (uint32_t)__double2hiint(pn), 31);
        pm = pc;
        pc = pn;
    };
    auto flush = [&]() {
        cnt += __popc((bits ^ (bits >> 1)) & ((1u << fresh) - 1u));
        fresh = 0;
    };
    int i = 1;
    for (; i + 8 <= np; i += 8) {
        if constexpr (AHEAD < 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) term(i + k);
            asm volatile("" ::: "memory");          // the second four reads are not issued before the first four terms are done
#pragma unroll
            for (int k = 4; k < 8; ++k) term(i + k);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) term(i + k);
        }
        fresh += 8;
        if (fresh >= 17) flush();
        // keep the recurrence inside the exponent range: a term grows or shrinks the pair by at most
        // |d - x| + e^2 < 2^18 (np <= 208), so every second block of 8 is often enough
        if (i & 8) {
            const int ex = __builtin_amdgcn_frexp_exp(fabs(pc) > fabs(pm) ? pc : pm);
            pc = ldexp(pc, -ex);
            pm = ldexp(pm, -ex);
            esum += ex;
        }
    }
    for (; i < np; ++i) { term(i); ++fresh; }
    flush();
    if (last) { *last = pc; *last_exp = esum; }
    return cnt;
}

// Where a straight line through the characteristic polynomial's values pa 2^ea at lo and pb 2^eb at hi (of opposite
// signs: one simple eigenvalue between them) crosses zero.
__device__ __forceinline__ double interpolated_root(double lo, double hi, double pa, int ea, double pb, int eb) {
    double ma = fabs(pa), mb = fabs(pb);
    const int d = max(-1000, min(1000, ea - eb));
    if (d >= 0) mb = ldexp(mb, -d); else ma = ldexp(ma, d);
    const double sum = ma + mb;
    const double rho = sum > 0.0 ? ma / sum : 0.5;
    return lo + (hi - lo) * rho;
}

// Points of a search round spaced geometrically around the estimate c inside [lo, hi]: HALF points on either side, at the
// distances delta ratio^k, k = 0 .. HALF - 1, the ratio such that the last one reaches the farther end of the bracket
// (clipped to the bracket).  k-th point below c in lane HALF - 1 - k of the group, k-th above in lane HALF + k: the
// points rise with the lane.  The bracket that comes out of the round is about as wide as the estimate was off.
template <int HALF>
__device__ __forceinline__ double geometric_point(int g, double lo, double hi, double c, double delta) {
    const float span = (float)fmax(hi - c, c - lo);
    const float lr = fmaxf(__log2f(span / (float)delta) * (1.0f / (float)(HALF - 1)), 0.05f);
    const int k = g < HALF ? HALF - 1 - g : g - HALF;
    const double d = delta * (double)exp2f(lr * (float)k);
    return g < HALF ? fmax(c - d, lo) : fmin(c + d, hi);
}
// A bracket narrower than this (after two uniform rounds, as a rule) is searched geometrically around the interpolated
// root when it holds one simple eigenvalue; a geometric round that gained less than a uniform one would have sends
// the next one back to uniform points (a second eigenvalue close below the bracket bends the polynomial).
#ifndef STURM_GEO_BELOW
#define STURM_GEO_BELOW 0.03
#endif
constexpr double STURM_GEOMETRIC_BELOW = STURM_GEO_BELOW;

// Sturm multisection for the largest eigenvalue of the symmetric tridiagonal matrix
// (L.de[k] = (d_k, e_k^2), k < np) inside [lo, hi]: lane t counts the eigenvalues below x_t.
// Every round shrinks the bracket 65-fold.
template <class LDS, bool PRIO = false>
__device__ __forceinline__ double sturm_largest_wave(const LDS& L, int np, int lane, double lo, double hi, int rounds,
                                                     SolvePriority<PRIO>* prio = nullptr) {
    bool geometric = false;
    double c = 0.0;
    for (int r = 0; r < rounds; ++r) {
        const double h = hi - lo;
        if (h <= 1e-11 * hi) break;
        if (prio) prio->spend(4 * np);
        const double x = geometric ? geometric_point<WAVE / 2>(lane, lo, hi, c, 0.4e-11 * hi)
                                   : lo + h * (1.0 / 65.0) * (double)(lane + 1);
        double pl;
        int pe;
        const int cnt = sturm_count_below(L.de, np, x, &pl, &pe);
        const uint64_t all = ballot64(cnt >= np);
        bool simple = false;                       // exactly one eigenvalue inside the new bracket, its ends in lanes t - 1, t
        int t = 0;
        if (all == 0) {
            lo = readlane_d(x, WAVE - 1);
        } else {
            t = __ffsll((unsigned long long)all) - 1;
            const double xh = readlane_d(x, t);
            const double xl = readlane_d(x, t > 0 ? t - 1 : 0);
            hi = xh;
            if (t > 0) {
                lo = xl;
                simple = np - __builtin_amdgcn_readlane(cnt, t - 1) == 1;
            }
        }
        const double hn = hi - lo;
        const bool gained = !geometric || hn * 65.0 <= h;
        geometric = simple && gained && hn <= STURM_GEOMETRIC_BELOW;
        if (geometric)
            c = interpolated_root(lo, hi, readlane_d(pl, t - 1), __builtin_amdgcn_readlane(pe, t - 1), readlane_d(pl, t),
                                  __builtin_amdgcn_readlane(pe, t));
    }
    return 0.5 * (lo + hi);
}

// Largest Laplacian eigenvalue of the graph on np <= NPAD <= 64 vertices whose adjacency
// rows are `crow` (lane j < np holds row j): Householder tridiagonalisation with row j in
// the registers of lane j, then Sturm multisection inside the Laplacian bracket
// [dmax + 1, min(np, 2 dmax)] (valid for a graph with at least one edge).
#ifdef MARL_EXP_TIMING
#define TSTAMP(k, pin)                                                                    \
    do {                                                                                  \
        uint64_t t_;                                                                      \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(pin)::"memory"); \
        tacc[k] += (double)(t_ - tlast);                                                  \
        tlast = t_;                                                                       \
    } while (0)
#else
#define TSTAMP(k, pin)
#endif
// An eigenproblem that reduce_problem has worked on: per lane (vertex of the quotient by twin classes) `meta` =
// multiplicity | diagonal entry << 8; per problem `brk` = n' | dmax << 8 of the graph before the reduction, which bracket
// the eigenvalue (0: nothing was done, the matrix is the plain Laplacian of the bit rows).
struct TwinInfo {
    int meta = 0, brk = 0;
};

// Row `lane` of the quotient of a Laplacian by its twin classes: a[c] = -s s_c where bit c of the row is set, s the
// square root of the lane's multiplicity and s_c the one of column c's vertex (sc[c], in LDS), `diag` on the diagonal.
// PAIR: two problems of at most 32 vertices in the two halves of the wave (sc points at the half's part).
template <int NPAD, bool PAIR = false>
__device__ __forceinline__ void quotient_row_registers(uint64_t crow, int diag, double s, const double* sc, double (&a)[NPAD]) {
    const double dg = (double)diag;
    double ns = -s;
    asm volatile("" : "+v"(ns) :: "memory");       // (a second build of the row starts from scratch, as in the builder below)
    const int dg_hi = __double2hiint(dg), dg_lo = __double2loint(dg);
    const uint32_t w_lo = (uint32_t)crow, w_hi = (uint32_t)(crow >> 32);
    static_for<0, NPAD>([&](auto k) {
        constexpr int c = decltype(k)::value;
        // (four columns' scales in flight at a time: all of them at once are a second matrix of registers)
        if constexpr (c % 4 == 0) asm volatile("" ::: "memory");
        const double v = ns * sc[c];
        const int on = __builtin_amdgcn_sbfe((int)(c < 32 ? w_lo : w_hi), c & 31, 1);       // 0 or -1
        int hi = __double2hiint(v) & on, lo = __double2loint(v) & on;
        const uint64_t dmask = PAIR ? (1ull << c) | (1ull << (c + 32)) : 1ull << c;
        asm volatile("v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %1, %1, %3, %4"
                     : "+v"(hi), "+v"(lo) : "v"(dg_hi), "v"(dg_lo), "s"(dmask));
        a[c] = __hiloint2double(hi, lo);
    });
}

// Row `lane` of the Laplacian of the graph whose adjacency row is `crow`, as NPAD registers: a[c] = -1 where bit c of
// the row is set, the degree on the diagonal (c == lane: the select mask is the constant 1 << c, spelled out because
// the compiler would build it with a compare per column).  PINNED: the statements are volatile, so that two builds of
// the same row in one kernel are not merged into one set of registers.
template <int NPAD, bool PINNED>
__device__ __forceinline__ void laplacian_row_registers(uint64_t crow, int cdeg, double (&a)[NPAD]) {
    const double dg = (double)cdeg;
    const int dg_hi = __double2hiint(dg), dg_lo = __double2loint(dg);
    uint32_t w_lo = (uint32_t)crow, w_hi = (uint32_t)(crow >> 32);
    // (opaque copies of the row words: the bit extractions of two builds are then not shared either - fifty values
    // kept from the first build for the second would be fifty registers live through whatever lies between)
    if constexpr (PINNED) asm volatile("" : "+v"(w_lo), "+v"(w_hi));
    static_for<0, NPAD>([&](auto k) {
        constexpr int c = decltype(k)::value;
        int hi = __builtin_amdgcn_sbfe((int)(c < 32 ? w_lo : w_hi), c & 31, 1) & (int)0xBFF00000, lo;
        const uint64_t diag = 1ull << c;
        if constexpr (PINNED)
            asm volatile("v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4"
                         : "+v"(hi), "=v"(lo) : "v"(dg_hi), "v"(dg_lo), "s"(diag));
        else
            asm("v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4"
                : "+v"(hi), "=v"(lo) : "v"(dg_hi), "v"(dg_lo), "s"(diag));
        a[c] = __hiloint2double(hi, lo);
    });
}

#include "lanczos_solve.inc"

// `rounds`: multisection rounds of the eigenvalue search.  LANCZOS: problems of at least LANCZOS_MIN_NP vertices go
// through the Lanczos iteration first (the direct solver below takes over if it does not meet its bound).
// nfull: vertices of the whole graph (lambda2 = nfull - mu_max scales the Lanczos tolerance).
// tw: the problem is a quotient by twin classes (sc: 64 doubles of LDS for the column scales).
template <int NPAD, class LDS, bool PRIO = false, bool LANCZOS = false>
__device__ __forceinline__ double mu_max_compact_wave(uint64_t crow, int np, int lane, LDS& L, int rounds, double nfull,
                                                      TwinInfo tw = TwinInfo(), double* sc = nullptr) {
    const int cdeg = __popcll(crow);
    int dmax = cdeg;
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x128, 0xF, 0xF, false));      // row_ror:8, 4, 2, 1
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x124, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x122, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x121, 0xF, 0xF, false));
    dmax = max(max(__builtin_amdgcn_readlane(dmax, 0), __builtin_amdgcn_readlane(dmax, 16)),
               max(__builtin_amdgcn_readlane(dmax, 32), __builtin_amdgcn_readlane(dmax, 48)));

    constexpr int KB = (NPAD + 15) / 16;
    // a quotient: the bracket is the one of the graph before merging, the matrix has weights
    const int brk = __builtin_amdgcn_readfirstlane(tw.brk);
    const bool weighted = brk && ballot64((tw.meta & 0xFF) > 1) != 0;       // (components dropped, no twins: plain rows)
    int npb = np, diag = cdeg;
    double s = 1.0;
    if (brk) {
        dmax = brk >> 8;
        npb = brk & 0xFF;
    }
    if (weighted) {
        diag = tw.meta >> 8;
        s = sqrt((double)(tw.meta & 0xFF));
        WSYNC();
        sc[lane] = s;
        WSYNC();
    }
    SolvePriority<PRIO> prio;
    if constexpr (LANCZOS) {
        if (np >= LANCZOS_MIN_NP) {
            prio.start(lanczos_work(np), lanczos_work(NPAD));
            double mu;
            // (the iteration works on a register matrix of its own, built a second time below if the direct solver has to
            // take over, which is rare: one set of registers live through both solvers costs the kernel a wave per SIMD)
            if (mu_max_lanczos_wave<NPAD, LDS, PRIO>(crow, diag, np, lane, dmax, npb, weighted ? sc : nullptr, s, nfull, L, rounds, prio, mu)) {
                prio.finish();
                return brk ? fmax(mu, (double)(dmax + 1)) : mu;
            }
#ifdef MARL_EXP_LANCZOS_STATS
            return nfull - 10000.0;
#endif
        }
        prio.start(SolvePriority<PRIO>::work(np, rounds), lanczos_work(NPAD));
    } else {
        prio.start(SolvePriority<PRIO>::work(np, rounds), SolvePriority<PRIO>::work(NPAD, rounds));
    }
    double a[NPAD];                                   // row `lane` of the matrix
    // (pinned: statements that may be hoisted out of the branch would keep both builds in registers at once)
    if (weighted) quotient_row_registers<NPAD>(crow, diag, s, sc, a);
    else laplacian_row_registers<NPAD, true>(crow, cdeg, a);

#ifdef MARL_EXP_TIMING
    double tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    // pivot column A[lane][i]: the first one from the bits, the later ones handed back by the update
    double col = (lane == np - 1) ? (double)diag : -(double)((crow >> (np - 1)) & 1ull);
    if (weighted && lane != np - 1) col *= s * sc[np - 1];
    for (int i = np - 1; i >= 2; --i) {
        const int l = i - 1;
        prio.spend(85 + 3 * i);
        TSTAMP(0, col);
        const double aii = col;                            // meaningful in lane i: A[i][i]
        double uj = (lane <= l) ? col : 0.0;
        double h = wave_sum_dpp_upto(uj * uj, l);
        TSTAMP(1, h);
        if (lane == i) L.de[i] = make_double2(aii, h);
        // h == 0 (row already reduced): u = 0 and 1/h := 0 turn the step into a no-op that still
        // hands the next pivot row over
        const double f = readlane_d(uj, l);
        const double sq = (h > 0.0) ? sqrt_45(h) : 0.0;
        const double g = (f >= 0.0) ? -sq : sq;
        const double h2 = h - f * g;
        if (lane == l) uj = f - g;
        TSTAMP(2, uj);
        // p = A u / h
        double p0 = 0.0, p1 = 0.0;
        double U[KB], Q[KB];
        replicate_rows<KB>(uj, lane, U);
        ColBlocks<NPAD, 0>::matvec(a, U, l, p0, p1);
        TSTAMP(3, p0);
        const double hinv = (h > 0.0) ? rcp_45(h2) : 0.0;
        double p = (lane <= l) ? (p0 + p1) * hinv : 0.0;
        TSTAMP(4, p);
        const double K = wave_sum_dpp_upto(uj * p, l) * (0.5 * hinv);
        double qj = fma(-K, uj, p);                         // lanes > l: u = p = 0
        // A -= u q^T + q u^T
        TSTAMP(5, qj);
        {
            replicate_rows<KB>(qj, lane, Q);
            const double nuj = -uj, nqj = -qj;
            HouseholderUpdate<NPAD>::run(a, nuj, nqj, l, U, Q, col);
        }
        TSTAMP(6, a[0]);
        TSTAMP(7, a[1]);
    }
#ifdef MARL_EXP_TIMING
    {
        double r_ = 0.0;
        static_for<0, 8>([&](auto k) { if ((blockIdx.x & 7) == decltype(k)::value) r_ = tacc[decltype(k)::value]; });
        return r_;
    }
#endif
    {
        const double a11 = readlane_d(a[1], 1), a10 = readlane_d(a[0], 1), a00 = readlane_d(a[0], 0);
        if (lane == 0) { L.de[1] = make_double2(a11, a10 * a10); L.de[0] = make_double2(a00, 0.0); }
    }
    WSYNC();
    // (a quotient whose eigenvalues all lie below dmax + 1 -- the largest one of the graph then belongs to a twin class and
    // equals dmax + 1 -- sends the search to the bracket's lower end, which is that value)
    const double lo = (double)(dmax + 1) * (1.0 - 1e-13) - 1e-13;
    const double hi = (double)min(npb, 2 * dmax) * (1.0 + 1e-13) + 1e-13;
    const double mu = sturm_largest_wave<LDS, PRIO>(L, np, lane, lo, hi, rounds, &prio);
    prio.finish();
    return brk ? fmax(mu, (double)(dmax + 1)) : mu;
}

// ---- two eigenproblems per wave (n' <= 32 each) ----------------------------------------------------------
// Half of the step's eigenproblems have at most 32 vertices, and more than half of a Householder step's
// instructions do not depend on the matrix size.  Two such problems share a wave: problem A in lanes 0..31,
// problem B in lanes 32..63, both reduced in lockstep over max(n'_A, n'_B) steps (the smaller matrix is padded
// with zero rows: its first steps find a zero column and do nothing).  The row broadcasts are local to rows of
// 16 lanes already; sums and pivot values become per-half quantities.
__device__ __forceinline__ double lane_xor16(double v) {           // the value of lane ^ 16 (ds_swizzle, bit mode)
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double half_sum(double v) {              // sum over the 32 lanes of the lane's half, in every lane
    v += dpp_mov<0x128>(v);
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    return v + lane_xor16(v);
}
__device__ __forceinline__ double half_lane(double v, int src, bool upper) {   // v of lane src (A) / 32 + src (B); src uniform
    const double a = readlane_d(v, src), b = readlane_d(v, 32 + src);
    return upper ? b : a;
}

// mu_max of both problems; `crow` = the lane's 32-bit complement row, np = max(n'_A, n'_B) <= 32, np_own = the
// size of the lane's own problem (0: no problem in this half).  The result is valid in every lane of a half.
// tw: the lane's problem is a quotient by twin classes (sc: 64 doubles of LDS for the column scales).
template <class LDS, bool PRIO = false>
__device__ __forceinline__ double mu_max_pair_wave(uint32_t crow, int np, int np_own, int lane, LDS& L, int rounds,
                                                   TwinInfo tw = TwinInfo(), double* sc = nullptr) {
    constexpr int NPAD = 32, KB = 2;
    const bool upper = lane >= 32;
    const int hl = lane & 31;
    const int cdeg = __popc(crow);
    int dmax = cdeg;
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x128, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x124, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x122, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_update_dpp(0, dmax, 0x121, 0xF, 0xF, false));
    dmax = max(dmax, __builtin_amdgcn_ds_swizzle(dmax, 0x401F));
    const bool weighted = ballot64((tw.meta & 0xFF) > 1) != 0;           // either problem has merged twins
    int npb = np_own, diag = cdeg;
    double s = 1.0;
    if (tw.brk) {
        dmax = tw.brk >> 8;
        npb = tw.brk & 0xFF;
    }
    if (weighted && tw.brk) {
        diag = tw.meta >> 8;
        s = sqrt((double)(tw.meta & 0xFF));
    }
    double a[NPAD];
    if (weighted) {
        WSYNC();
        sc[lane] = s;
        WSYNC();
        quotient_row_registers<NPAD, true>((uint64_t)crow, diag, s, sc + (lane & 32), a);
    } else {
        const double dg = (double)cdeg;
        const int dg_hi = __double2hiint(dg), dg_lo = __double2loint(dg);
        static_for<0, NPAD>([&](auto k) {
            constexpr int c = decltype(k)::value;
            int hi = __builtin_amdgcn_sbfe((int)crow, c, 1) & (int)0xBFF00000, lo;
            const uint64_t diag = (1ull << c) | (1ull << (c + 32));
            asm("v_cndmask_b32_e64 %0, %0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4"
                : "+v"(hi), "=v"(lo) : "v"(dg_hi), "v"(dg_lo), "s"(diag));
            a[c] = __hiloint2double(hi, lo);
        });
    }
    double col = (hl == np - 1) ? (double)diag : -(double)((crow >> (np - 1)) & 1u);
    if (weighted && hl != np - 1) col *= s * sc[(lane & 32) + np - 1];
    SolvePriority<PRIO> prio;
    prio.start(SolvePriority<PRIO>::work(np, rounds), SolvePriority<PRIO>::work(NPAD, rounds));
    for (int i = np - 1; i >= 2; --i) {
        const int l = i - 1;
        prio.spend(85 + 3 * i);
        const double aii = col;
        double uj = (hl <= l) ? col : 0.0;
        const double h = half_sum(uj * uj);
        if (hl == i) L.de[lane] = make_double2(aii, h);
        const double f = half_lane(uj, l, upper);
        const double sq = (h > 0.0) ? sqrt_45(h) : 0.0;
        const double g = (f >= 0.0) ? -sq : sq;
        const double h2 = h - f * g;
        if (hl == l) uj = f - g;
        double p0 = 0.0, p1 = 0.0;
        double U[KB], Q[KB];
        {
            const int ulo = __double2loint(uj), uhi = __double2hiint(uj);
            static_for<0, KB>([&](auto kk) {
                const int addr = ((lane & 32) + decltype(kk)::value * 16 + (lane & 15)) << 2;
                U[decltype(kk)::value] = __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, uhi), __builtin_amdgcn_ds_bpermute(addr, ulo));
            });
        }
        ColBlocks<NPAD, 0>::matvec(a, U, l, p0, p1);
        const double hinv = (h > 0.0) ? rcp_45(h2) : 0.0;
        const double pj = (hl <= l) ? (p0 + p1) * hinv : 0.0;
        const double K = half_sum(uj * pj) * (0.5 * hinv);
        const double qj = fma(-K, uj, pj);
        {
            const int qlo = __double2loint(qj), qhi = __double2hiint(qj);
            static_for<0, KB>([&](auto kk) {
                const int addr = ((lane & 32) + decltype(kk)::value * 16 + (lane & 15)) << 2;
                Q[decltype(kk)::value] = __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, qhi), __builtin_amdgcn_ds_bpermute(addr, qlo));
            });
            const double nuj = -uj, nqj = -qj;
            HouseholderUpdate<NPAD>::run(a, nuj, nqj, l, U, Q, col);
        }
    }
    {
        const double a11 = half_lane(a[1], 1, upper), a10 = half_lane(a[0], 1, upper), a00 = half_lane(a[0], 0, upper);
        if (hl == 0) { L.de[(lane & 32) + 1] = make_double2(a11, a10 * a10); L.de[lane & 32] = make_double2(a00, 0.0); }
    }
    WSYNC();
    // 32-way multisection per half on the Sturm sign changes (see sturm_largest_wave); the padded rows are zero
    // eigenvalues, below every x of the bracket, so "all np eigenvalues below x" is the test in both halves.
    double lo = (double)(dmax + 1) * (1.0 - 1e-13) - 1e-13;
    double hi = (double)min(npb, 2 * dmax) * (1.0 + 1e-13) + 1e-13;
    const double2* de = L.de + (lane & 32);
    // (Uniform rounds only.  The geometric rounds of sturm_largest_wave would make a half's result depend on its partner:
    // the recurrence runs over the larger problem's rows, the padding multiplies the polynomial's value by powers of x,
    // and the interpolated root moves in its last bits -- while lambda2 of an env must not depend on which env shares
    // its wave, or a batch and its shards would differ.  Measured gain of having them here: 0.1 % of a step.)
    for (int r = 0; r < rounds; ++r) {
        const bool active = (hi - lo) > 1e-11 * hi;
        if (ballot64(active) == 0) break;
        prio.spend(4 * np);
        const double w = (hi - lo) * (1.0 / 33.0);
        const double x = lo + w * (double)(hl + 1);
        double pm = 1.0, pc = de[0].x - x;
        uint32_t bits = (uint32_t)__double2hiint(pc) >> 31;
        int cnt = 0, fresh = 1;
        auto term = [&](int i) {
            const double2 t = de[i];
            const double pn = fma(t.x - x, pc, -(t.y * pm));
            bits = __builtin_amdgcn_alignbit(bits, (uint32_t)__double2hiint(pn), 31);
            pm = pc;
            pc = pn;
        };
        auto flush = [&]() {
            cnt += __popc((bits ^ (bits >> 1)) & ((1u << fresh) - 1u));
            fresh = 0;
        };
        int i = 1;
        for (; i + 8 <= np; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) term(i + k);
            fresh += 8;
            if (fresh >= 17) flush();
            if (i & 8) {
                const int ex = __builtin_amdgcn_frexp_exp(fabs(pc) > fabs(pm) ? pc : pm);
                pc = ldexp(pc, -ex);
                pm = ldexp(pm, -ex);
            }
        }
        for (; i < np; ++i) { term(i); ++fresh; }
        flush();
        const uint64_t all = ballot64(cnt >= np);
        const uint32_t mine = upper ? (uint32_t)(all >> 32) : (uint32_t)all;
        if (active) {
            if (mine == 0) {
                lo = lo + w * 32.0;                     // x of the half's last lane
            } else {
                const int t = __ffs((int)mine) - 1;
                const double xl = lo + w * (double)t;   // x of lane t - 1 (t > 0)
                hi = lo + w * (double)(t + 1);          // x of lane t, the same expression that lane evaluated
                if (t > 0) lo = xl;
            }
        }
    }
    prio.finish();
    const double mu = 0.5 * (lo + hi);
    return tw.brk ? fmax(mu, (double)(dmax + 1)) : mu;
}

// mu_max of one eigenproblem inside a wave, by the solver the step's solve launch would use for it: problems of
// at most 32 vertices go through the two-per-wave code (alone, the upper half idle), so that the result does not
// depend on the schedule; larger ones through the 64-lane solver.
template <int NPAD, class LDS, bool LANCZOS = false>
__device__ __forceinline__ double mu_max_in_wave(uint64_t crow, int np, int lane, LDS& L, int rounds, double nfull,
                                                 TwinInfo tw = TwinInfo(), double* sc = nullptr) {
    if (np <= 32) {
        const bool upper = lane >= 32;
        if (upper) tw = TwinInfo();
        return readlane_d(mu_max_pair_wave(upper ? 0u : (uint32_t)crow, np, upper ? 0 : np, lane, L, rounds + 2, tw, sc), 0);
    }
    if constexpr (NPAD > 32) return mu_max_compact_wave<NPAD, LDS, false, LANCZOS>(crow, np, lane, L, rounds, nfull, tw, sc);
    return 0.0;
}

// Second-smallest eigenvalue of L = D - A for the graph whose adjacency rows are
// `row` (communication_graph.py:184-286; the reference calls LAPACK dsyevr through
// scipy.linalg.eigh(subset_by_index=[0,1])).
//
//   * complete graph -> n, disconnected graph -> 0 (exact, one ballot per BFS level);
//   * otherwise lambda2(G) = n - mu_max(complement of G): L(G) + L(G') = n I - J, so on
//     the space orthogonal to the all-ones vector the spectra mirror each other.  Only
//     the n' vertices that miss at least one neighbour carry edges of the complement;
//     they are compacted to lanes 0..n'-1 (lambda2_prepare), the problem is reduced where its
//     structure allows (reduce_problem) and handed to mu_max_compact_wave / mu_max_pair_wave.
//
// Reductions that keep mu_max.  The largest Laplacian eigenvalue of a graph with an edge is at least dmax + 1 (dmax its
// largest degree), so whatever cannot exceed that need not be solved:
//   * Components.  mu_max is the largest over the components, and a component of c vertices stays at or below c:
//     those of at most dmax + 1 vertices are dropped.  If none is left, mu_max = dmax + 1 (the component of the vertex of
//     largest degree has exactly that many vertices, all of them its neighbours) and there is no eigenproblem at all.
//   * Twins.  Two vertices with the same neighbours (not adjacent to each other: "open" twins, N(a) = N(b)) or the
//     same neighbours and each other ("closed" twins, N[a] = N[b]) span an eigenvector e_a - e_b with the eigenvalue deg
//     or deg + 1.  Every other eigenvector is constant on a class of twins, so the rest of the spectrum is that of the
//     quotient: one vertex per class, Q_ii = the sum of the multiplicities m_j of the neighbouring classes,
//     Q_ij = -sqrt(m_i m_j) between neighbours.
// Together: mu_max(L) = max(mu_max(Q restricted to the large components), dmax + 1).  The graphs of a step have plenty
// of both (links missing towards the same far vertex, enemies jammed by the same jammer, small groups cut off
// together): 30 % of the tridiagonalisation's work, and a quarter of the eigenproblems vanish.
//
// What the solvers need to know about a reduced problem: per lane (vertex of the quotient) `meta` = m | Q_ii << 8; per
// problem `brk` = n' | dmax << 8 of the graph before the reduction, which bracket the eigenvalue (0: nothing was done and
// the matrix is the plain Laplacian of the bit rows).
// Smaller problems are left alone: the reduction costs what a few Householder steps do.  It pays from 6 vertices on
// where the problems of a launch are sorted and paired afterwards (the three-launch step), from 12 where the wave solves
// its own problem right away (4 096 envs x (10, 4) in one launch: 2.27 against 2.18 x 10^8 env-steps/s with 12 against
// 6).  The limit goes by the size of the swarm (up to 16 enemies a step is one launch whatever the batch), not by the
// schedule, so that an env's lambda2 is the same bits whichever schedule steps it.
__device__ __forceinline__ int reduce_min_np(int n) { return n <= 16 ? 12 : 6; }

// The bits of `row` at the set positions of `keep`, packed towards bit 0: one step per run of consecutive positions
// (`keep` is wave-uniform).
__device__ __forceinline__ uint64_t compress_bits(uint64_t row, uint64_t keep) {
    uint64_t out = 0;
    int k = 0;
    while (keep) {
        const int a = __ffsll((unsigned long long)keep) - 1;
        const uint64_t inv = ~(keep >> a);                   // zero bits mark the end of the run
        const int len = inv ? __ffsll((unsigned long long)inv) - 1 : 64;
        const uint64_t ones = (len >= 64) ? ~0ull : ((1ull << len) - 1);
        out |= ((row >> a) & ones) << k;
        k += len;
        keep = (a + len >= 64) ? 0ull : (keep & ~(ones << a));
    }
    return out;
}

// Reduces the problem on the np vertices whose rows are `crow` (lane j < np: row j; n: the whole graph's vertices) as described above; np = 0 on
// return: no eigenproblem is left, mu_max = dmax + 1 with dmax = tw.brk >> 8.  Twin classes are found through a hash
// table in LDS (a slot per 9-bit hash of the row, the last writer stays): every vertex compares its row with the one of
// its slot's owner, so a merge always rests on an exact comparison and a collision only leaves twins unmerged.
__device__ __forceinline__ void reduce_problem(uint64_t& crow, int& np, int n, int lane, WaveLds& L, TwinInfo& tw) {
    tw.meta = 0; tw.brk = 0;
    if (np < reduce_min_np(n)) return;
    const uint64_t self = 1ull << lane;
    const uint64_t all = (np >= 64) ? ~0ull : ((1ull << np) - 1);
    const int deg = __popcll(crow);
    int dmax = deg;                                        // (rows beyond np are zero)
    dmax = max(dmax, dpp_mov32<0x128>(dmax));              // row_ror:8, 4, 2, 1: the largest of every row of 16 lanes
    dmax = max(dmax, dpp_mov32<0x124>(dmax));
    dmax = max(dmax, dpp_mov32<0x122>(dmax));
    dmax = max(dmax, dpp_mov32<0x121>(dmax));
    dmax = max(max(__builtin_amdgcn_readlane(dmax, 0), __builtin_amdgcn_readlane(dmax, 16)),
               max(__builtin_amdgcn_readlane(dmax, 32), __builtin_amdgcn_readlane(dmax, 48)));
    // Components of more than dmax + 1 vertices.  The search starts at a vertex of the largest degree, whose component is
    // usually the only large one: once no more than dmax + 1 vertices are left unvisited, whatever components they form
    // are small and need not be told apart.
    uint64_t large = 0, unvisited = all;
    uint64_t comp = ballot64(deg == dmax);
    comp &= 0ull - comp;
    while (true) {
        uint64_t frontier = comp;
        while (frontier) {
            const uint64_t hit = ballot64((crow & frontier) != 0);
            const uint64_t nw = hit & ~comp;
            comp |= nw;
            frontier = nw;
        }
        if (__popcll(comp) > dmax + 1) large |= comp;
        unvisited &= ~comp;
        if (__popcll(unvisited) <= dmax + 1) break;
        comp = unvisited & (0ull - unvisited);
    }
    if (large == 0) {
        tw.brk = np | (dmax << 8);
        np = 0;
        crow = 0;
        return;
    }
    const bool valid = (large >> lane) & 1ull;
    unsigned char* table = reinterpret_cast<unsigned char*>(L.xfer);        // 512 slots
    int* count = reinterpret_cast<int*>(L.xfer);
    int rep = -1, add = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {             // open twins, then closed ones (a vertex cannot have both kinds)
        const uint64_t key = pass ? (crow | self) : crow;
        const uint32_t h = ((uint32_t)key * 0x9E3779B1u + (uint32_t)(key >> 32) * 0x85EBCA77u) >> 23;
        WSYNC();
        if (valid) table[h] = (unsigned char)lane;
        WSYNC();
        const int r = valid ? (int)table[h] : lane;
        const uint64_t key_r = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), r) << 32) | (uint32_t)__shfl((int)key, r);
        if (valid && r != lane && rep < 0 && key_r == key) { rep = r; add = pass ? 0x10001 : 1; }
    }
    const uint64_t merged = ballot64(rep >= 0);
    if (merged == 0 && large == all) return;               // nothing to drop, nothing to merge
    int mult = 1, qii = deg;
    if (merged) {
        WSYNC();
        count[lane] = 0;
        WSYNC();
        if (rep >= 0) atomicAdd(&count[rep], add);
        WSYNC();
        const int c = count[lane];
        mult = 1 + (c & 0xFFFF);
        qii = deg - ((c >> 16) ? mult - 1 : 0);            // the members of a closed class are neighbours of each other
    }
    const uint64_t keep = large & ~merged;
    const int pos = __popcll(keep & (self - 1));
    const uint64_t packed = compress_bits(crow & keep, keep);
    WSYNC();
    if ((keep >> lane) & 1ull) L.xfer[pos] = packed;
    WSYNC();
    tw.brk = np | (dmax << 8);
    np = __popcll(keep);
    crow = (lane < np) ? L.xfer[lane] : 0ull;
    WSYNC();
    if ((keep >> lane) & 1ull) count[pos] = mult | (qii << 8);
    WSYNC();
    tw.meta = (lane < np) ? count[lane] : 0;
}

// Everything before the eigen-solve: returns true when lambda2 is already known (in `lam`),
// false when the complement rows `crow` of the np compacted vertices have to be solved (reduce_problem may shrink the
// problem before that).
__device__ __forceinline__ bool lambda2_prepare(uint64_t row, int n, int lane, WaveLds& L, double& lam, uint64_t& crow, int& np) {
    crow = 0; np = 0; lam = 0.0;
    if (n <= 1) return true;
    const int deg = __popcll(row);
    const uint64_t active = (n >= 64) ? ~0ull : ((1ull << n) - 1);
    const uint64_t touched = ballot64(lane < n && deg < n - 1);
    if (touched == 0) { lam = (double)n; return true; }
    {   // connectivity: breadth-first search from vertex 0
        uint64_t comp = 1ull, frontier = 1ull;
        while (frontier) {
            const uint64_t hit = ballot64((row & frontier) != 0);
            const uint64_t nw = hit & ~comp;
            comp |= nw;
            frontier = nw;
        }
        if (comp != active) return true;
    }
    // complement rows restricted to the touched vertices, compacted
    np = __popcll(touched);
    crow = compress_bits(~row & active & ~(1ull << lane) & touched, touched);
    WSYNC();
    if ((touched >> lane) & 1ull) L.xfer[__popcll(touched & ((1ull << lane) - 1))] = crow;
    WSYNC();
    crow = (lane < np) ? L.xfer[lane] : 0ull;
    return false;
}

template <int NPAD, bool LANCZOS = false>
__device__ __forceinline__ double lambda2_wave(uint64_t row, int n, int lane, WaveLds& L, int rounds) {
    double lam;
    uint64_t crow;
    int np;
    TwinInfo tw;
    if (lambda2_prepare(row, n, lane, L, lam, crow, np)) return lam;
    reduce_problem(crow, np, n, lane, L, tw);
    if (np == 0) return (double)n - (double)((tw.brk >> 8) + 1);
    return (double)n - mu_max_in_wave<NPAD, WaveLds, LANCZOS>(crow, np, lane, L, rounds, (double)n, tw, L.scale);
}

// ------------------------------------------------------------------ DBSCAN + centroids
// Labels of sklearn DBSCAN(eps, min_samples) on the positions in L.sx/L.sy
// (dbscan_clustering.py:113), then centroids / sizes in dict order
// (dbscan_clustering.py:116-128, 130-144, 230-260) into L.cx/L.cy/L.csz.
// Returns the lane's label; n_clusters and ncent by reference (wave-uniform).
__device__ __forceinline__ int dbscan_wave(const MarlEnvParams& P, int n, int lane, WaveLds& L, int& n_clusters, int& ncent) {
    const double eps2 = P.eps * P.eps;
    uint64_t nb = 0;
    {
        const double xi = L.sxy[lane & (WAVE - 1)].x, yi = L.sxy[lane & (WAVE - 1)].y;
        for (int j = 0; j < n; ++j) {
            const double dx = xi - L.sxy[j].x, dy = yi - L.sxy[j].y;
            const double d2 = dx * dx + dy * dy;
            nb |= (uint64_t)(d2 <= eps2) << j;
        }
        if (lane >= n) nb = 0;
    }
    const uint64_t core = ballot64(lane < n && __popcll(nb) >= P.min_samples);
    int label = -1;
    int num = 0;
    uint64_t unvisited = core;
    while (unvisited) {
        const int s = __ffsll((unsigned long long)unvisited) - 1;
        uint64_t comp = 1ull << s, frontier = comp, reached = 0;
        while (frontier) {   // expand through core points only; remember every point touched
            const uint64_t hit = ballot64((nb & frontier) != 0);
            reached |= hit;
            const uint64_t nw = hit & core & ~comp;
            comp |= nw;
            frontier = nw;
        }
        const uint64_t members = comp | (reached & ~core);   // core component + its border points
        if (((members >> lane) & 1ull) && label == -1) label = num;
        unvisited &= ~comp;
        ++num;
    }
    L.lab[lane] = label;
    const bool has_noise = ballot64(lane < n && label == -1) != 0;
    const int entries = num + (has_noise ? 1 : 0);
    WSYNC();
    for (int c = lane; c < entries; c += WAVE) {
        const int id = (c < num) ? c : -1;
        double ax = 0.0, ay = 0.0;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {          // np.mean(axis=0): sequential sum in index order
            if (L.lab[i] == id) { ax = ax + L.sxy[i].x; ay = ay + L.sxy[i].y; ++cnt; }
        }
        L.cx[c] = ax / (double)cnt;
        L.cy[c] = ay / (double)cnt;
        L.csz[c] = cnt;
    }
    n_clusters = num;
    ncent = entries;
    if (num == 0) {                            // all noise: fallback centroid at the arena centre
        if (lane == 0) { L.cx[0] = P.arena_size / 2; L.cy[0] = P.arena_size / 2; L.csz[0] = n; }
        ncent = 1;
    }
    WSYNC();
    return label;
}

// should_reassign_jammers(drift_threshold) (dbscan_clustering.py:163-204) on the
// current (L.cx..) and previous (L.px..) centroid tables.
template <class LDS>
__device__ __forceinline__ bool should_reassign_wave(const MarlEnvParams& P, int lane, const LDS& L, int num, int ncent,
                                     int pnum, int pncent) {
    bool mine = false;
    for (int s = lane; s < ncent; s += WAVE) {
        const int id = (s < num) ? s : -1;
        const bool in_prev = (id >= 0) ? (id < pnum) : (pncent > pnum);
        if (!in_prev) { mine = true; continue; }              // new cluster id: drift = inf
        const int ps = (id >= 0) ? id : pnum;
        const double dx = L.cx[s] - L.px[ps], dy = L.cy[s] - L.py[ps];
        if (sqrt(dx * dx + dy * dy) > P.drift_threshold) mine = true;
    }
    return ballot64(mine) != 0 || ncent != pncent;
}

// assign_jammers_to_clusters(strategy="proportional") (dbscan_clustering.py:263-362) into L.asg.
template <class LDS>
__device__ __forceinline__ void assign_proportional_wave(int m, int lane, LDS& L, int num, int ncent) {
    if (lane == 0 && m > 0 && ncent > 0) {
        const int first = 0;
        const int nvalid = (num > 0) ? num : ncent;        // all-noise: the -1 entry is the one cluster
        int total = 0;
        for (int s = first; s < nvalid; ++s) total += L.csz[s];
        if (total == 0) total = nvalid;
        int j = 0, remaining = m;
        for (int s = 0; s < nvalid; ++s) {
            const int id = (num > 0) ? s : -1;
            int count;
            if (s == nvalid - 1) {
                count = remaining;
            } else {
                count = (int)((double)(m * L.csz[s]) / (double)total);
                if (count < 1) count = 1;
                if (count > remaining) count = remaining;
            }
            for (int t = 0; t < count; ++t) L.asg[j + t] = id;
            j += count;
            remaining -= count;
        }
    }
    WSYNC();
}

// ------------------------------------------------------------------ reward + observation
struct RewardOut { double total; double d2, d3, d5; float r0, r1, r2, r3, r4; float o0, o1, o2, o3, o4; };

// Term 1 (reward.py:178-199) and the clipped weighted sum (reward.py:282-306) from the four terms
// that do not depend on lambda2.
__device__ __forceinline__ double reward_total(const MarlEnvParams& P, double lam, double lam0, double r2, double r3,
                                               float r4, double r5, double& r1) {
    r1 = 0.0;
    if (lam0 > 0.0) {
        const double lc = lam > 0.0 ? lam : 0.0;
        r1 = 1.0 - (lc / lam0);
        r1 = fmin(fmax(r1, 0.0), 1.0);
    }
    const double raw = P.omega[0] * r1 + P.omega[1] * r2 + P.omega[2] * r3 - P.omega[3] * (double)r4 - P.omega[4] * r5;
    return fmin(fmax(raw, -10.0), 10.0);
}

// RewardCalculator.compute (reward.py:106-306) and ObservationBuilder.build_vectorized
// (observation.py:264-332).  Returns, in lane j < m, the five features of jammer j.
//
// The two loops (over the centroids and over the other jammers) are shared out over the lanes:
// with P = 64 / (m rounded up to a power of two) lanes per jammer, lane j*P + part takes the
// centroids / jammers part, part + P, ...; partial results meet by xor shuffles inside the group.
// For m = 8 that is 3 trips instead of 20 and 1 instead of 8.
template <class LDS>
__device__ __forceinline__ RewardOut reward_obs_wave(const MarlEnvParams& P, int n, int m, int lane, const LDS& L,
                                     int eband, int num, int ncent, double lam, double lam0,
                                     float raw_vx, float raw_vy) {
    RewardOut out;
    const bool act = lane < m;
    const double A = P.arena_size;
    const double Rjam = P.R_jam[eband];
    int lg_m = 0;
    while ((1 << lg_m) < m) ++lg_m;
    const int lg_p = 6 - lg_m, parts = 1 << lg_p;
    const int j = lane >> lg_p, part = lane & (parts - 1);
    const bool actj = j < m;
    const double xj = L.jx[j], yj = L.jy[j];
    // nearest centroid over every dict entry, first minimum wins = lexicographic minimum of
    // (squared distance, slot); one square root at the end.  (The outputs of this function are
    // float32 observations and a reward with a 1e-5 tolerance, so the square roots here are the
    // two-Newton-step ones.)
    double best2 = INFINITY;
    int bslot = 0x7fffffff;
    for (int s = part; s < ncent; s += parts) {
        const double dx = xj - L.cx[s], dy = yj - L.cy[s];
        const double d2 = dx * dx + dy * dy;
        if (d2 < best2) { best2 = d2; bslot = s; }
    }
    // pairwise jammer distances: overlap count (pairs k > j) and sum of the distances to the others
    const double thr = 2 * Rjam, thr2 = thr * thr;
    int ov = 0;
    double dsum = 0.0;
    for (int k = part; k < m; k += parts) {
        const double dx = xj - L.jx[k], dy = yj - L.jy[k];
        const double d2 = dx * dx + dy * dy;
        const double d = d2 > 0.0 ? fast_sqrt(d2) : 0.0;
        if (k != j) dsum = dsum + d;
        // overlap test of reward.py:300-304 on the exactly rounded distance when it is a near tie
        const bool close = (fabs(d2 - thr2) <= 1e-12 * thr2) ? (sqrt(d2) < thr) : (d2 < thr2);
        if (k > j && close) ++ov;
    }
    for (int o = 1; o < parts; o <<= 1) {
        const double od2 = __shfl_xor(best2, o);
        const int os = __shfl_xor(bslot, o);
        const double osum = __shfl_xor(dsum, o);
        if (od2 < best2 || (od2 == best2 && os < bslot)) { best2 = od2; bslot = os; }
        dsum = dsum + osum;
    }
    if (bslot == 0x7fffffff) bslot = 0;                  // no centroid at all
    const int ovl = wave_sum(actj ? ov : 0);
    const double r5 = (m > 1) ? (double)ovl / ((double)(m * (m - 1)) / 2) : 0.0;
    const int match = (act && L.jb[lane & (WAVE - 1)] == eband) ? 1 : 0;
    const int nmatch = wave_sum(match);
    const double r2 = (double)nmatch / (double)m;
    // term 3 (reward.py:218-260): assigned centroid if its id still exists, else nearest
    double prox = 0.0;
    if (actj && part == 0 && ncent > 0) {
        const int c = L.asg[j];
        int slot = bslot;
        if (c >= 0 && c < num) slot = c;
        else if (c == -1 && ncent > num) slot = num;
        const double dx = xj - L.cx[slot], dy = yj - L.cy[slot];
        const double d2 = dx * dx + dy * dy;
        prox = exp(-P.kappa * (d2 > 0.0 ? fast_sqrt(d2) : 0.0));
    }
    const double r3 = wave_sum(prox) / (double)m;
    // term 4 (reward.py:262-280), float32 like the reference's float32 action path
    float en = 0.0f;
    if (act) en = (raw_vx * raw_vx + raw_vy * raw_vy) / P.v_max_sq_f32;
    const float r4 = wave_sum(en) / (float)m;
    double r1;
    out.total = reward_total(P, lam, lam0, r2, r3, r4, r5, r1);
    out.d2 = r2; out.d3 = r3; out.d5 = r5;
    out.r0 = (float)r1; out.r1 = (float)r2; out.r2 = (float)r3; out.r3 = r4; out.r4 = (float)r5;
    // observation: jammer j's results move from lane j*P to lane j
    const int src = (lane << lg_p) & (WAVE - 1);
    const double best2_j = __shfl(best2, src), dsum_j = __shfl(dsum, src);
    const int bslot_j = __shfl(bslot, src);
    const double best = (best2_j > 0.0 && best2_j < INFINITY) ? fast_sqrt(best2_j) : best2_j;
    out.o0 = (float)fmin(fmax(best / A, 0.0), 1.0);
    out.o1 = (n > 0 && ncent > 0) ? (float)((double)L.csz[bslot_j] / (double)n) : 0.0f;
    out.o2 = (m > 1) ? (float)fmin(fmax(dsum_j / (double)(m - 1) / A, 0.0), 1.0) : 0.5f;
    out.o3 = (float)r5;
    out.o4 = match ? 1.0f : 0.0f;
    return out;
}

// _compute_avg_jamming_power (jammer_env.py:510-560): mean over links i<j of the
// strongest matching jammer's power at the link midpoint, in dBm; -80 when none.
//
// The strongest jammer is the nearest one, and P (wl / (4 pi d))^2 with d = max(sqrt(d2), 1e-10) is C / max(D2, 4e-20)
// for D2 = |2 q - (p_i + p_j)|^2 = 4 d2 and C = 4 P (wl / 4 pi)^2: no square root, one reciprocal (45 bits) per link.
// Enemy i takes the links to its (n - 1) / 2 successors on the circle 0..n-1 (even n: the first n / 2 also the one
// opposite), so every link is evaluated once and all lanes stay busy.  The order of the sum is not numpy's: the
// result agrees with the reference to about 1e-13 relative, not bit for bit (an info-only value).
struct LinkPower {
    double c4, psum = 0.0;
    int pcnt = 0;
    __device__ __forceinline__ LinkPower(const MarlEnvParams& P, int eband) {
        const double g = kSpeedOfLight / band_hz(eband) / kFourPi;
        c4 = 4.0 * P.P_jam_w * (g * g);
    }
    // D2: four times the squared distance of the nearest matching jammer to the midpoint
    __device__ __forceinline__ void add(double D2, bool counted) {
        const double pw = c4 * rcp_45(fmax(D2, 4e-20));
        if (counted && pw > 0.0) { psum += pw; ++pcnt; }
    }
    __device__ __forceinline__ double dbm() {
        const double tot = wave_sum(psum);
        const int cnt = wave_sum(pcnt);
        if (cnt == 0) return -80.0;
        return 10.0 * log10(tot / (double)cnt) + 30.0;
    }
};

__device__ __forceinline__ double avg_power_wave(const MarlEnvParams& P, int n, int m, int lane, const WaveLds& L, int eband) {
    const uint64_t matching = ballot64(lane < m && L.jb[lane & (WAVE - 1)] == eband);
    if (matching == 0) return -80.0;
    LinkPower acc(P, eband);
    const double2 pi = L.sxy[lane & (WAVE - 1)];
    const int half = (n - 1) >> 1, opposite = (n & 1) ? 0 : n >> 1;
    for (int k = 1; k <= half + (opposite ? 1 : 0); ++k) {
        int j = lane + k;
        j = j >= n ? j - n : j;
        const double2 pj = L.sxy[(lane < n) ? j : 0];
        const double sx = pi.x + pj.x, sy = pi.y + pj.y;
        double D2 = INFINITY;
        for (uint64_t mk = matching; mk; mk &= mk - 1) {
            const int q = __ffsll((unsigned long long)mk) - 1;
            const double dx = fma(2.0, L.jx[q], -sx), dy = fma(2.0, L.jy[q], -sy);
            D2 = fmin(D2, fma(dx, dx, dy * dy));
        }
        acc.add(D2, k <= half ? lane < n : lane < opposite);
    }
    return acc.dbm();
}

__device__ __forceinline__ void store_bits(uint32_t* dst, int e, int n, int lane, uint64_t row) {
    if (dst == nullptr || lane >= n) return;
    const int W = (n + 31) / 32;
    uint32_t* p = dst + ((size_t)e * n + lane) * W;
    p[0] = (uint32_t)row;
    if (W > 1) p[1] = (uint32_t)(row >> 32);
}

__device__ __forceinline__ WaveLds& carve(unsigned char* smem) { return *reinterpret_cast<WaveLds*>(smem); }

// Last act of a step for env e: the reward and lambda2 outputs, and the episode bookkeeping the rollout loop
// of the reference keeps on the host (training/trainer.py:229-268): the running return of the episode and,
// when the episode ends with this step, its contribution to the four sums {episodes, return, lambda2
// reduction in percent (100 (1 - lambda2_final / lambda2_initial), 0 when lambda2_initial <= 1e-10), length}.
__device__ __forceinline__ void finish_env(const MarlEnvParams& P, const MarlEnvState& S, const MarlStepIO& io, int e,
                                           int step, double total, double lam, double lam0) {
    const float r = (float)total;
    io.reward[e] = r;
    if (io.lambda2) io.lambda2[e] = lam;
    double ret = S.episode_return[e] + (double)r;
    if (step >= P.max_steps) {
        if (io.episode_stats) {
            atomicAdd(&io.episode_stats[0], 1.0);
            atomicAdd(&io.episode_stats[1], ret);
            atomicAdd(&io.episode_stats[2], lam0 > 1e-10 ? 100.0 * (1.0 - lam / lam0) : 0.0);
            atomicAdd(&io.episode_stats[3], (double)step);
        }
        ret = 0.0;
    }
    S.episode_return[e] = ret;
}

// ------------------------------------------------------------------ step kernels (N <= 64)
// A step is three launches.  The duration of the eigen-solve depends on the graph (nothing for a
// complete or disconnected graph, O(n'^3) otherwise), and a fused kernel, which can only take the
// envs in index order, leaves most SIMDs idle while the last long solves finish.  Split like this
// the cheap, uniform part runs at the occupancy its small register footprint allows, and the
// solves are started longest first.
// solve_hist: SOLVE_HIST_SUBS histograms over n' (the envs spread over them by index, which keeps the
// atomic traffic per counter low), SOLVE_HIST_STRIDE words apart, then the number of pending solves.
enum { SOLVE_HIST_SUBS = 16, SOLVE_HIST_STRIDE = 80, SOLVE_HIST_TOTAL = SOLVE_HIST_SUBS * SOLVE_HIST_STRIDE,
       SOLVE_HIST_WORDS = SOLVE_HIST_TOTAL + 64 };

// Front half: everything of JammerEnv.step but the eigen-solve.
// FUSED = 0: as described.  FUSED = NPAD > 0: the solve follows in the same wave (one launch per step): taken for
// small batches and small swarms, where the extra launches cost more than the scheduling gains.
template <int FUSED>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(FUSED > 0 ? 1 : 8, FUSED > 0 ? 4 : 8))) void env_step_front_kernel(MarlEnvParams P, MarlEnvState S, MarlStepIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int n = P.N, m = P.M, KS = n + 1;
    const int step = S.step_count[e] + 1;
    const int eband = S.enemy_band[e];
    const double arena = P.arena_size;
#ifdef MARL_EXP_TIMING
    float fsec[5] = {0, 0, 0, 0, 0};
    uint64_t ft0, ft1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ft0)::"memory");
#define FSTAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(ft1)::"memory"); fsec[k] = (float)(ft1 - ft0); ft0 = ft1; } while (0)
#else
#define FSTAMP(k)
#endif

    // Every load of the step is issued here, ahead of any arithmetic (indices clamped so that all lanes
    // load): with a handful of waves per SIMD a load that waits behind a dependent chain costs a
    // microsecond each time.
    const bool ct = P.enemy_mode == MARL_ENEMY_COORDINATED_TURN;
    const size_t oj = (size_t)e * m + (lane < m ? lane : m - 1);
    const size_t oi = (size_t)e * n + (lane < n ? lane : n - 1);
    const float* act_p = io.actions + oj * 3;
    const float a0 = act_p[0], a1 = act_p[1], a2 = act_p[2];
    const double jx0 = S.jam_x[oj], jy0 = S.jam_y[oj];
    const int asg0 = S.assign[oj];
    double ex = S.enemy_x[oi], ey = S.enemy_y[oi], evx = S.enemy_vx[oi], evy = S.enemy_vy[oi];
    double ew = ct ? S.enemy_omega[oi] : 0.0;
    double nx = 0.0, ny = 0.0;
    if (io.noise) { nx = io.noise[oi * 2]; ny = io.noise[oi * 2 + 1]; }
    const int tick = S.tick[e] + 1;
    int num = S.n_clusters[e], ncent = S.ncent[e];
    ncent = ncent < 0 ? 0 : (ncent > WAVE ? WAVE : ncent);          // never more than 64 entries (WaveLds)
    const double lam0 = S.lambda2_initial[e];
    // the cluster table is read up to its fill level only (the level arrives with the first loads; the
    // table is not needed before the enemy dynamics are done)
    double ccx = 0.0, ccy = 0.0;
    int ccs = 0;
    if (lane < ncent) {
        const size_t oc = (size_t)e * KS + lane;
        ccx = S.cent_x[oc]; ccy = S.cent_y[oc]; ccs = S.cent_size[oc];
    }

    // -- action decode (jammer_env.py:372-388, action_space.py:190-240)
    float raw_vx = 0.0f, raw_vy = 0.0f;
    if (lane < m) {
        raw_vx = a0; raw_vy = a1;
        int band = (int)a2;                                    // astype(int32): truncation
        band = band < 0 ? 0 : (band > MARL_NUM_BANDS - 1 ? MARL_NUM_BANDS - 1 : band);
        const float cvx = fminf(fmaxf(raw_vx, -P.v_max_f32), P.v_max_f32);
        const float cvy = fminf(fmaxf(raw_vy, -P.v_max_f32), P.v_max_f32);
        const float sx_ = cvx * P.dt_f32, sy_ = cvy * P.dt_f32;   // float32 product, as numpy does for f32 actions
        double x = jx0 + (double)sx_;
        double y = jy0 + (double)sy_;
        x = fmin(fmax(x, 0.0), arena);
        y = fmin(fmax(y, 0.0), arena);
        S.jam_x[oj] = x;
        S.jam_y[oj] = y;
        S.jam_band[oj] = band;
        S.last_vx[oj] = raw_vx;
        S.last_vy[oj] = raw_vy;
        L.jx[lane] = x; L.jy[lane] = y; L.jb[lane] = band;
        L.asg[lane] = asg0;
    }
    // -- enemy dynamics (enemy_swarm.py:139-242)
    if (lane < n) {
        if (!io.noise)
            noise_pair(io.seed, io.env_offset + e, (uint32_t)tick, (uint32_t)lane, ct ? P.noise_std * 0.5 : P.noise_std, nx, ny);
        enemy_advance(P, ex, ey, evx, evy, ew, nx, ny);
        if (ct) S.enemy_omega[oi] = ew;
        S.enemy_x[oi] = ex; S.enemy_y[oi] = ey; S.enemy_vx[oi] = evx; S.enemy_vy[oi] = evy;
        L.sxy[lane] = make_double2(ex, ey);
    }
    WSYNC();
    FSTAMP(0);
    // -- clusters: recompute every K steps (jammer_env.py:394-402), else load the table
    if (step % P.K_recompute == 0 && !(P.stage_flags & MARL_STAGE_NO_RECLUSTER)) {
        const int pnum = num, pncent = ncent;
        if (lane < pncent) { L.px[lane] = ccx; L.py[lane] = ccy; }    // centroids -> _prev_centroids
        for (int s = WAVE + lane; s < pncent; s += WAVE) {
            L.px[s] = S.cent_x[(size_t)e * KS + s];
            L.py[s] = S.cent_y[(size_t)e * KS + s];
        }
        const int label = dbscan_wave(P, n, lane, L, num, ncent);
        if (lane < n) S.labels[(size_t)e * n + lane] = label;
        for (int s = lane; s < pncent; s += WAVE) {
            S.prev_x[(size_t)e * KS + s] = L.px[s];
            S.prev_y[(size_t)e * KS + s] = L.py[s];
        }
        for (int s = lane; s < ncent; s += WAVE) {
            S.cent_x[(size_t)e * KS + s] = L.cx[s];
            S.cent_y[(size_t)e * KS + s] = L.cy[s];
            S.cent_size[(size_t)e * KS + s] = L.csz[s];
        }
        if (lane == 0) {
            S.prev_n_clusters[e] = pnum; S.prev_ncent[e] = pncent;
            S.n_clusters[e] = num; S.ncent[e] = ncent;
        }
        if (should_reassign_wave(P, lane, L, num, ncent, pnum, pncent)) {
            assign_proportional_wave(m, lane, L, num, ncent);
            if (lane < m) S.assign[(size_t)e * m + lane] = L.asg[lane];
        }
    } else {
        if (lane < ncent) { L.cx[lane] = ccx; L.cy[lane] = ccy; L.csz[lane] = ccs; }
        for (int s = WAVE + lane; s < ncent; s += WAVE) {      // N = 64 with a noise entry: 65 rows
            L.cx[s] = S.cent_x[(size_t)e * KS + s];
            L.cy[s] = S.cent_y[(size_t)e * KS + s];
            L.csz[s] = S.cent_size[(size_t)e * KS + s];
        }
    }
    // float32 copies for the first stage of the jamming test (their storage held the previous centroids until here)
    L.sxf[lane] = (lane < n) ? (float)ex : 1e30f;          // 1e30: squares overflow to +inf, never within range
    L.syf[lane] = (lane < n) ? (float)ey : 1e30f;
    WSYNC();
    FSTAMP(1);
    // -- lambda2 of the jammed graph (jammer_env.py:464-508)
    uint64_t adj, jam;
    pair_rows(P, L, n, m, lane, eband, adj, jam);
    store_bits(io.adj_bits, e, n, lane, adj);
    store_bits(io.jam_bits, e, n, lane, jam);
    FSTAMP(2);
    double lam;
    uint64_t crow;
    int np;
    TwinInfo tw;
    bool final_lam = lambda2_prepare(adj & ~jam, n, lane, L, lam, crow, np);
    if constexpr (FUSED > 0) {
        if (!final_lam) {
            reduce_problem(crow, np, n, lane, L, tw);
            lam = np == 0 ? (double)n - (double)((tw.brk >> 8) + 1)
                          : (double)n - mu_max_in_wave<FUSED>(crow, np, lane, L, P.lambda2_rounds, (double)n, tw, L.scale);
            final_lam = true;
            np = 0;
        }
    }
    FSTAMP(3);
    // -- reward, observation (jammer_env.py:414-438)
    const RewardOut rw = reward_obs_wave(P, n, m, lane, L, eband, num, ncent, lam, lam0, raw_vx, raw_vy);
    if (lane < m && io.obs) {
        float* o = io.obs + ((size_t)e * m + lane) * MARL_OBS_DIM;
        o[0] = rw.o0; o[1] = rw.o1; o[2] = rw.o2; o[3] = rw.o3; o[4] = rw.o4;
    }
    if (io.reward_terms && lane == 0) {
        float* r = io.reward_terms + (size_t)e * 5;
        r[0] = rw.r0; r[1] = rw.r1; r[2] = rw.r2; r[3] = rw.r3; r[4] = rw.r4;
    }
    double avgp = 0.0;
    if (io.avg_power_dbm) avgp = avg_power_wave(P, n, m, lane, L, eband);
    bool solved_late = false;
    if (FUSED == 0 && !final_lam) {
        // (here and not before the reward: the reduction has the registers of the reward's inputs to itself)
        reduce_problem(crow, np, n, lane, L, tw);
        if (np == 0) {                      // nothing left to solve: the reward is finished below, as the solve kernel would
            lam = (double)n - (double)((tw.brk >> 8) + 1);
            solved_late = true;
        }
        if (lane < np) {
            S.solve_rows[(size_t)e * WAVE + lane] = crow;
            if (tw.brk) S.solve_meta[(size_t)e * WAVE + lane] = (uint16_t)tw.meta;
        }
    }
    if (lane == 0) {
        S.step_count[e] = step;
        S.tick[e] = tick;
        io.truncated[e] = (uint8_t)(step >= P.max_steps);
        if (io.avg_power_dbm) io.avg_power_dbm[e] = avgp;
        if (FUSED == 0) S.solve_np[e] = np ? np | (tw.brk << 8) : 0;
        if (final_lam) {
            finish_env(P, S, io, e, step, rw.total, lam, lam0);
        } else if (solved_late) {
            double r1;
            const double total_r = reward_total(P, lam, lam0, rw.d2, rw.d3, rw.r3, rw.d5, r1);
            finish_env(P, S, io, e, step, total_r, lam, lam0);
#if !defined(MARL_EXP_TIMING) && !defined(MARL_EXP_BLOCKTIME)
            if (io.reward_terms) io.reward_terms[(size_t)e * 5] = (float)r1;
#endif
        } else {
            S.solve_slot[e] = atomicAdd(&S.solve_hist[(e & (SOLVE_HIST_SUBS - 1)) * SOLVE_HIST_STRIDE + np], 1);
            double* part = S.solve_partial + (size_t)e * 4;
            part[0] = rw.d2; part[1] = rw.d3; part[2] = (double)rw.r3; part[3] = rw.d5;
        }
    }
#ifdef MARL_EXP_TIMING
    FSTAMP(4);
    if (io.reward_terms && lane == 0) {
        float* r = io.reward_terms + (size_t)e * 5;
        r[0] = fsec[0]; r[1] = fsec[1]; r[2] = fsec[2]; r[3] = fsec[3]; r[4] = fsec[4];
    }
#endif
}

// Order: env indices of the pending solves into solve_perm, largest matrix first (a counting sort
// over the histogram the front kernel built).  Every block derives the bin offsets itself.
// prio_slots: wave slots of the chip for the solve kernel (0: the launch does not schedule by priority).  A launch of
// between SOLVE_PRIO_ITEMS_MIN and SOLVE_PRIO_ITEMS_MAX percent of them -- about one round of waves -- runs with every
// solve's issue priority following the work it has left (SolvePriority); the verdict goes to the solve blocks through
// solve_hist[SOLVE_HIST_TOTAL + 2] (1 or 0).
#ifndef SOLVE_PRIO_ITEMS_MIN
#define SOLVE_PRIO_ITEMS_MIN 99
#define SOLVE_PRIO_ITEMS_MAX 130
#endif
__global__ __launch_bounds__(1024) void solve_order_kernel(int E, MarlEnvState S, int prio_slots) {
    __shared__ int start[SOLVE_HIST_SUBS][WAVE + 1];
    if (threadIdx.x < WAVE) {                        // lane i stands for bin 64 - i: largest matrices first
        const int bin = WAVE - (int)threadIdx.x;
        int cnt[SOLVE_HIST_SUBS];
        int own = 0;
#pragma unroll
        for (int k = 0; k < SOLVE_HIST_SUBS; ++k) {
            cnt[k] = S.solve_hist[k * SOLVE_HIST_STRIDE + bin];
            own += cnt[k];
        }
        int incl = own;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += up;
        }
        int at = incl - own;
#pragma unroll
        for (int k = 0; k < SOLVE_HIST_SUBS; ++k) {
            start[k][bin] = at;
            at += cnt[k];
        }
        if (blockIdx.x == 0 && threadIdx.x == WAVE - 1) S.solve_hist[SOLVE_HIST_TOTAL] = incl;
        if (blockIdx.x == 0 && threadIdx.x == WAVE / 2 - 1) S.solve_hist[SOLVE_HIST_TOTAL + 1] = incl;     // problems of more than 32 vertices
        const int total = __shfl(incl, WAVE - 1), nbig = __shfl(incl, WAVE / 2 - 1);
        const int nitems = nbig + ((total - nbig + 1) >> 1);
        const bool by_priority = prio_slots > 0 && nitems * 100 >= SOLVE_PRIO_ITEMS_MIN * prio_slots &&
                                 nitems * 100 <= SOLVE_PRIO_ITEMS_MAX * prio_slots;
        // (the verdict: 0, or the number of SIMDs the solve blocks go round)
        if (blockIdx.x == 0 && threadIdx.x == 0) S.solve_hist[SOLVE_HIST_TOTAL + 2] = by_priority ? prio_slots / 4 : 0;
    }
    __syncthreads();
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        const int np = S.solve_np[e] & 0xFF;
        if (np > 0) {
            S.solve_perm[start[e & (SOLVE_HIST_SUBS - 1)][np] + S.solve_slot[e]] = e;
        }
    }
}

struct SolveLds {
    double2 de[WAVE];
    double scale[WAVE];        // column scales of a quotient by twin classes
};

#ifdef MARL_EXP_BLOCKTIME
// Timeline probe (tools/solve_timeline.py): in place of its env's reward terms a solve records the hardware id
// of its wave, the size of its problem and its start / end clock.
#define BT_BEGIN()                                                                                                \
    uint32_t bt_hw, bt_xcc;                                                                                       \
    uint64_t bt_t0, bt_t1;                                                                                        \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)" \
                 : "=s"(bt_hw), "=s"(bt_xcc), "=s"(bt_t0)::"memory")
#define BT_RECORD(env, npv, item)                                                                                 \
    do {                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bt_t1)::"memory"); \
        uint32_t* w_ = reinterpret_cast<uint32_t*>(io.reward_terms + (size_t)(env) * 5);                          \
        w_[0] = bt_hw; w_[1] = (bt_xcc & 0xF) | ((uint32_t)(npv) << 8) | 0xAB000000u;                             \
        w_[2] = (uint32_t)bt_t0; w_[3] = (uint32_t)bt_t1; w_[4] = (uint32_t)(item);                               \
    } while (0)
#else
#define BT_BEGIN()
#define BT_RECORD(env, npv, item)
#endif

// Work item `item` of a step's solve launch, by one wave: the items are the pending eigenproblems in the order of
// solve_perm (largest first).  The problems of more than 32 vertices are an item each (items 0..nbig-1), the others
// share one in pairs.  The wave then finishes the reward of the env (or the two).
template <int NPAD, bool PRIO, bool LANCZOS>
__device__ __forceinline__ void solve_item(const MarlEnvParams& P, const MarlEnvState& S, const MarlStepIO& io, int item,
                                           int nbig, int total, int lane, SolveLds& L) {
    BT_BEGIN();
    if (item >= nbig) {
        const int k = nbig + 2 * (item - nbig);
        const int eA = S.solve_perm[k], eB = (k + 1 < total) ? S.solve_perm[k + 1] : -1;
        const int wA = S.solve_np[eA], wB = (eB >= 0) ? S.solve_np[eB] : 0;        // n' | quotient bracket << 8
        const int npA = wA & 0xFF, npB = wB & 0xFF;                                 // sorted: npA >= npB
        const bool upper = lane >= 32;
        const int e = upper ? eB : eA, np_own = upper ? npB : npA, hl = lane & 31;
        const uint32_t crow = (e >= 0 && hl < np_own) ? (uint32_t)S.solve_rows[(size_t)e * WAVE + hl] : 0u;
        TwinInfo tw;
        tw.brk = (upper ? wB : wA) >> 8;
        if (tw.brk && hl < np_own) tw.meta = S.solve_meta[(size_t)e * WAVE + hl];
        const double mu = mu_max_pair_wave<SolveLds, PRIO>(crow, npA, np_own, lane, L, P.lambda2_rounds + 2, tw, L.scale);
        if (hl == 0 && e >= 0) {
            const double lam = (double)P.N - mu;
            const double* part = S.solve_partial + (size_t)e * 4;
            const double lam0 = S.lambda2_initial[e];
            double r1;
            const double total_r = reward_total(P, lam, lam0, part[0], part[1], (float)part[2], part[3], r1);
            finish_env(P, S, io, e, S.step_count[e], total_r, lam, lam0);
#if !defined(MARL_EXP_TIMING) && !defined(MARL_EXP_BLOCKTIME)
            if (io.reward_terms) io.reward_terms[(size_t)e * 5] = (float)r1;
#endif
            BT_RECORD(e, np_own, item);
        }
        return;
    }
    const int e = S.solve_perm[item];
    const int np = S.solve_np[e] & 0xFF;
    const uint64_t crow = (lane < np) ? S.solve_rows[(size_t)e * WAVE + lane] : 0ull;
    TwinInfo tw;
    tw.brk = S.solve_np[e] >> 8;
    if (tw.brk && lane < np) tw.meta = S.solve_meta[(size_t)e * WAVE + lane];
    const double lam = (double)P.N - mu_max_compact_wave<NPAD, SolveLds, PRIO, LANCZOS>(crow, np, lane, L, P.lambda2_rounds, (double)P.N, tw, L.scale);
    if (lane == 0) {
        const double* part = S.solve_partial + (size_t)e * 4;
        double r1;
        const double lam0 = S.lambda2_initial[e];
        const double total_r = reward_total(P, lam, lam0, part[0], part[1], (float)part[2], part[3], r1);
        finish_env(P, S, io, e, S.step_count[e], total_r, lam, lam0);
#if !defined(MARL_EXP_TIMING) && !defined(MARL_EXP_BLOCKTIME)
        if (io.reward_terms) io.reward_terms[(size_t)e * 5] = (float)r1;
#endif
        BT_RECORD(e, np, item);
    }
}

// Solve, one wave per workgroup: block b takes item b.
template <int NPAD, bool PRIO, bool LANCZOS = false>
__device__ __forceinline__ void solve_block(const MarlEnvParams& P, const MarlEnvState& S, const MarlStepIO& io) {
    __shared__ SolveLds L;
    const int lane = threadIdx.x;
    const int total = S.solve_hist[SOLVE_HIST_TOTAL];
    const int nbig = S.solve_hist[SOLVE_HIST_TOTAL + 1];
    const int nitems = nbig + ((total - nbig + 1) >> 1);
    if ((int)blockIdx.x >= nitems) {
        // the histogram has been consumed by the order kernel: the last block clears it for the next step
        // (bins only: the other blocks of this launch still read the total)
        if (blockIdx.x == gridDim.x - 1)
            for (int k = lane; k < SOLVE_HIST_TOTAL; k += WAVE) S.solve_hist[k] = 0;
        return;
    }
    if constexpr (PRIO) {
        // Whether this launch schedules by priority was decided by the order kernel, from the number of its items.  Two
        // copies of the solver, one taken by every block of a launch: keeping the books of the priority costs a solve
        // 2-3 % even when no level is ever changed.
        const int simds = S.solve_hist[SOLVE_HIST_TOTAL + 2];
        if (simds) {
#ifndef SOLVE_NO_SNAKE
            // One-wave workgroups go to the SIMDs in turn, block b to SIMD b mod simds: with the items in descending order
            // SIMD 0 would get the largest of every round.  Every second round runs backwards instead, so that the sums
            // per SIMD -- which is what a launch scheduled by priority takes -- are level.
            const int b = (int)blockIdx.x, q = b / simds, first = q * simds;
            const int len = min(simds, nitems - first);
            const int item = (q & 1) ? first + (len - 1 - (b - first)) : b;
#else
            const int item = (int)blockIdx.x;
#endif
            solve_item<NPAD, true, LANCZOS>(P, S, io, item, nbig, total, lane, L);
        } else solve_item<NPAD, false, LANCZOS>(P, S, io, (int)blockIdx.x, nbig, total, lane, L);
    } else {
        solve_item<NPAD, false, LANCZOS>(P, S, io, (int)blockIdx.x, nbig, total, lane, L);
    }
}
// In launches that the chip holds in about one round of waves (the order kernel's verdict) the issue priority of every
// solve follows the work it has left (SolvePriority), so that the solves of a SIMD finish together.
template <int NPAD>
__global__ __launch_bounds__(WAVE) void env_step_solve_kernel(MarlEnvParams P, MarlEnvState S, MarlStepIO io) {
    solve_block<NPAD, true>(P, S, io);
}
// The same with the problems of more than 32 vertices going through the Lanczos iteration first (lanczos_solve.inc).
// Three waves per SIMD: the iteration needs about 165 registers at 50 columns (the register matrix, the vectors in
// both forms, the reads of an eigenvalue search in flight), where the direct solver alone fits four waves -- which is
// why it is a kernel of its own and not a branch of the one above.
template <int NPAD>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(3, 8))) void env_step_solve_lanczos_kernel(MarlEnvParams P, MarlEnvState S, MarlStepIO io) {
    solve_block<NPAD, false, true>(P, S, io);
}

// ------------------------------------------------------------------ reset kernel (N <= 64)
// COMPLETE: the caller knows that the un-jammed graph is the complete one whatever is drawn (generated spawns lie in
// [0.1 A, 0.9 A]^2 and every band's communication range exceeds the arena's diagonal): lambda2 = N without adjacency
// test and without the eigensolver's registers, so that six to eight envs share a SIMD instead of one.
template <int NPAD, bool COMPLETE = false>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(COMPLETE ? 4 : 1, 8)))
void env_reset_wave_kernel(MarlEnvParams P, MarlEnvState S, MarlResetIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    // a reset of every env also (re)initialises the step's hand-over histograms, so that a state
    // buffer that was not zeroed by the caller becomes valid with its first full reset
    if (io.mask == nullptr && e == 0 && S.solve_hist != nullptr)
        for (int k = lane; k < SOLVE_HIST_WORDS; k += WAVE) S.solve_hist[k] = 0;
    if (io.mask && io.mask[e] == 0) return;
    const int n = P.N, m = P.M, KS = n + 1;
    const double arena = P.arena_size;
    const uint64_t genv = io.env_offset + e;
    const uint32_t k0 = (uint32_t)io.seed, k1 = (uint32_t)(io.seed >> 32);
    int num = S.n_clusters[e], ncent = S.ncent[e];
    ncent = ncent < 0 ? 0 : (ncent > KS ? KS : ncent);       // an unzeroed buffer must not send the copy loops astray
    ncent = ncent > WAVE ? WAVE : ncent;
    int eband = 2;
    double lam0 = 0.0;
    const bool host_init = io.enemy_pos != nullptr;
    const int attempts = host_init ? 1 : (P.max_reset_attempts > 0 ? P.max_reset_attempts : 1);
    for (int attempt = 0; attempt < attempts; ++attempt) {
        if (lane < n) {
            double x, y;
            enemy_spawn(P, S, io, e, lane, attempt, x, y);
            L.sxy[lane].x = x; L.sxy[lane].y = y;
        }
        if (io.enemy_band) {
            eband = io.enemy_band[e];
        } else {
            uint32_t r[4];
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ STREAM_RESET_ENV, episode_key(io, S, e),
                       (uint32_t)attempt, k0, k1, r);
            eband = (int)(r[0] & 3u);
        }
        eband = eband < 0 ? 0 : (eband > 3 ? 3 : eband);
        WSYNC();
        // _update_clusters (jammer_env.py:298): previous table <- current table, then fit
        const int pnum = num, pncent = ncent;
        for (int s = lane; s < pncent; s += WAVE) {
            S.prev_x[(size_t)e * KS + s] = S.cent_x[(size_t)e * KS + s];
            S.prev_y[(size_t)e * KS + s] = S.cent_y[(size_t)e * KS + s];
        }
        WSYNC();
        const int label = dbscan_wave(P, n, lane, L, num, ncent);
        if (lane < n) S.labels[(size_t)e * n + lane] = label;
        for (int s = lane; s < ncent; s += WAVE) {
            S.cent_x[(size_t)e * KS + s] = L.cx[s];
            S.cent_y[(size_t)e * KS + s] = L.cy[s];
            S.cent_size[(size_t)e * KS + s] = L.csz[s];
        }
        if (lane == 0) {
            S.prev_n_clusters[e] = pnum; S.prev_ncent[e] = pncent;
            S.n_clusters[e] = num; S.ncent[e] = ncent;
        }
        WSYNC();
        // lambda2 of the un-jammed graph (jammer_env.py:301)
        if constexpr (COMPLETE) {
            lam0 = n <= 1 ? 0.0 : (double)n;
        } else {
            const uint64_t adj = adjacency_row(L, n, lane, P.comm_d2_max[eband]);
            lam0 = lambda2_wave<NPAD>(adj, n, lane, L, P.lambda2_rounds);
        }
        if (lam0 >= P.min_lambda2_initial) break;
    }
    // jammers (jammer_env.py:321-339)
    if (P.random_jammer_start) {
        if (lane < m) L.asg[lane] = MARL_ASSIGN_NONE;
        WSYNC();
    } else {
        if (lane < m) L.asg[lane] = MARL_ASSIGN_NONE;
        WSYNC();
        assign_proportional_wave(m, lane, L, num, ncent);
    }
    if (lane < m) {
        const size_t o = (size_t)e * m + lane;
        double ix, iy;
        int band;
        uint32_t r[4], r2[4];
        if (io.jam_init == nullptr || io.jam_band == nullptr) {
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)lane << 8) ^ STREAM_RESET_JAMMER,
                       episode_key(io, S, e), 0u, k0, k1, r);
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)lane << 8) ^ STREAM_RESET_JAMMER,
                       episode_key(io, S, e), 1u, k0, k1, r2);
        }
        if (io.jam_init) { ix = io.jam_init[o * 2]; iy = io.jam_init[o * 2 + 1]; }
        else if (P.random_jammer_start) {
            ix = 0.1 * arena + (0.9 * arena - 0.1 * arena) * u01(r[0], r[1]);
            iy = 0.1 * arena + (0.9 * arena - 0.1 * arena) * u01(r[2], r[3]);
        } else {
            ix = -P.spread + (P.spread + P.spread) * u01(r[0], r[1]);
            iy = -P.spread + (P.spread + P.spread) * u01(r[2], r[3]);
        }
        band = io.jam_band ? io.jam_band[o] : (int)(r2[0] & 3u);
        double x = ix, y = iy;
        if (!P.random_jammer_start) {       // get_jammer_initial_positions: clip(centroid + offset)
            const int c = L.asg[lane];
            x = 0.0; y = 0.0;
            int slot = -1;
            if (c >= 0 && c < num) slot = c;
            else if (c == -1 && ncent > num) slot = num;
            if (slot >= 0) {
                x = fmin(fmax(L.cx[slot] + ix, 0.0), arena);
                y = fmin(fmax(L.cy[slot] + iy, 0.0), arena);
            }
        }
        S.jam_x[o] = x; S.jam_y[o] = y; S.jam_band[o] = band;
        S.last_vx[o] = 0.0f; S.last_vy[o] = 0.0f;
        S.assign[o] = L.asg[lane];
        L.jx[lane] = x; L.jy[lane] = y; L.jb[lane] = band;
    }
    WSYNC();
    const RewardOut rw = reward_obs_wave(P, n, m, lane, L, eband, num, ncent, lam0, lam0, 0.0f, 0.0f);
    if (lane < m && io.obs) {
        float* o = io.obs + ((size_t)e * m + lane) * MARL_OBS_DIM;
        o[0] = rw.o0; o[1] = rw.o1; o[2] = rw.o2; o[3] = rw.o3; o[4] = rw.o4;
    }
    if (lane == 0) {
        S.step_count[e] = 0;
        S.enemy_band[e] = eband;
        S.lambda2_initial[e] = lam0;
        S.episode_return[e] = 0.0;
    }
}

// ------------------------------------------------------------------ stage kernels (N <= 64)
__global__ __launch_bounds__(WAVE) void adjacency_bits_kernel(MarlEnvParams P, const double* x, const double* y,
                                                              const int32_t* band, uint32_t* bits) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N;
    if (lane < n) { L.sxy[lane].x = x[(size_t)e * n + lane]; L.sxy[lane].y = y[(size_t)e * n + lane]; }
    WSYNC();
    int b = band[e]; b = b < 0 ? 0 : (b > 3 ? 3 : b);
    store_bits(bits, e, n, lane, adjacency_row(L, n, lane, P.comm_d2_max[b]));
}

__global__ __launch_bounds__(WAVE) void jam_bits_kernel(MarlEnvParams P, const double* x, const double* y,
                                                        const int32_t* band, const double* jx, const double* jy,
                                                        const int32_t* jb, uint32_t* bits) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N, m = P.M;
    if (lane < n) { L.sxy[lane].x = x[(size_t)e * n + lane]; L.sxy[lane].y = y[(size_t)e * n + lane]; }
    if (lane < m) {
        L.jx[lane] = jx[(size_t)e * m + lane]; L.jy[lane] = jy[(size_t)e * m + lane];
        L.jb[lane] = jb[(size_t)e * m + lane];
    }
    WSYNC();
    int b = band[e]; b = b < 0 ? 0 : (b > 3 ? 3 : b);
    store_bits(bits, e, n, lane, jam_row(L, n, m, lane, b, P.jam_d2_max[b]));
}

template <int NPAD, bool LANCZOS>
__device__ __forceinline__ void lambda2_bits_body(int n, const uint32_t* bits, double* lambda2, int rounds) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int W = (n + 31) / 32;
    uint64_t row = 0;
    if (lane < n) {
        const uint32_t* p = bits + ((size_t)e * n + lane) * W;
        row = p[0];
        if (W > 1) row |= (uint64_t)p[1] << 32;
        row &= ~(1ull << lane);
        if (n < 64) row &= (1ull << n) - 1;
    }
    const double lam = lambda2_wave<NPAD, LANCZOS>(row, n, lane, L, rounds);
    if (lane == 0) lambda2[e] = lam;
}
template <int NPAD>
__global__ __launch_bounds__(WAVE) void lambda2_bits_kernel(int n, const uint32_t* bits, double* lambda2, int rounds) {
    lambda2_bits_body<NPAD, false>(n, bits, lambda2, rounds);
}
template <int NPAD>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(3, 8))) void lambda2_bits_lanczos_kernel(int n, const uint32_t* bits, double* lambda2, int rounds) {
    lambda2_bits_body<NPAD, true>(n, bits, lambda2, rounds);
}

__global__ __launch_bounds__(WAVE) void dbscan_kernel(MarlEnvParams P, const double* x, const double* y,
                                                      int32_t* labels, double* cx, double* cy, int32_t* csz,
                                                      int32_t* n_clusters, int32_t* ncent_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N, KS = n + 1;
    if (lane < n) { L.sxy[lane].x = x[(size_t)e * n + lane]; L.sxy[lane].y = y[(size_t)e * n + lane]; }
    WSYNC();
    int num, ncent;
    const int label = dbscan_wave(P, n, lane, L, num, ncent);
    if (lane < n) labels[(size_t)e * n + lane] = label;
    for (int s = lane; s < ncent; s += WAVE) {
        cx[(size_t)e * KS + s] = L.cx[s];
        cy[(size_t)e * KS + s] = L.cy[s];
        csz[(size_t)e * KS + s] = L.csz[s];
    }
    if (lane == 0) { n_clusters[e] = num; ncent_out[e] = ncent; }
}

// GAE (rollout_buffer.py:151-204): one lane per env, reverse scan over T in float32.
__global__ void gae_kernel(int T, int E, const float* rewards, const float* values, const float* dones,
                           const float* last_value, const float* last_done, float gamma, float gamma_lam,
                           float* adv, float* ret) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float ld = last_done[e];
    float next_v = last_value[e] * (1.0f - ld);
    float next_d = ld;
    float last = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t o = (size_t)t * E + e;
        const float nnt = 1.0f - next_d;
        const float v = values[o];
        const float delta = rewards[o] + gamma * next_v * nnt - v;
        last = delta + gamma_lam * nnt * last;
        adv[o] = last;
        ret[o] = last + v;
        next_v = v;
        next_d = dones[o];
    }
}

// RewardCalculator.compute_per_agent (reward.py:308-360): lane j < M owns jammer j; the tables are
// read from global memory (this is a stage op beside the step, which only needs the shared reward).
__global__ __launch_bounds__(WAVE) void reward_per_agent_kernel(MarlEnvParams P, const double* lambda2,
        const double* lambda2_initial, const int32_t* enemy_band, const double* jam_x, const double* jam_y,
        const int32_t* jam_band, const float* vx, const float* vy, const double* cent_x, const double* cent_y,
        const int32_t* n_clusters, const int32_t* ncent_arr, const int32_t* assign, float* rewards) {
    const int e = blockIdx.x, lane = threadIdx.x;
    const int n = P.N, m = P.M, KS = n + 1;
    const int eband = enemy_band[e], num = n_clusters[e], ncent = ncent_arr[e];
    const bool act = lane < m;
    const size_t o = (size_t)e * m + (act ? lane : 0);
    const double xj = jam_x[o], yj = jam_y[o];
    const double* cx = cent_x + (size_t)e * KS;
    const double* cy = cent_y + (size_t)e * KS;
    // shared terms
    const double lam = lambda2[e], lam0 = lambda2_initial[e];
    double r1 = 0.0;
    if (lam0 > 0.0) r1 = fmin(fmax(1.0 - ((lam > 0.0 ? lam : 0.0) / lam0), 0.0), 1.0);
    const double thr = 2 * P.R_jam[eband];
    int ov = 0;
    for (int k = lane + 1; k < m; ++k) {
        const double dx = xj - jam_x[(size_t)e * m + k], dy = yj - jam_y[(size_t)e * m + k];
        if (sqrt(dx * dx + dy * dy) < thr) ++ov;
    }
    const int ovl = wave_sum(act ? ov : 0);
    const double r5 = (m > 1) ? (double)ovl / ((double)(m * (m - 1)) / 2) : 0.0;
    if (!act) return;
    // own terms
    const double rb = (jam_band[o] == eband) ? 1.0 : 0.0;
    double rp = 0.0;
    if (ncent > 0) {
        const int c = assign[o];
        int slot = -1;
        if (c >= 0 && c < num) slot = c;
        else if (c == -1 && ncent > num) slot = num;
        if (slot < 0) {                                    // nearest, first minimum wins
            double best = INFINITY;
            for (int s2 = 0; s2 < ncent; ++s2) {
                const double dx = xj - cx[s2], dy = yj - cy[s2];
                const double d = sqrt(dx * dx + dy * dy);
                if (d < best) { best = d; slot = s2; }
            }
        }
        const double dx = xj - cx[slot], dy = yj - cy[slot];
        rp = exp(-P.kappa * sqrt(dx * dx + dy * dy));
    }
    const float re = (vx[o] * vx[o] + vy[o] * vy[o]) / P.v_max_sq_f32;
    rewards[o] = (float)(P.omega[0] * r1 + P.omega[1] * rb + P.omega[2] * rp - P.omega[3] * (double)re - P.omega[4] * r5);
}

// RewardCalculator.compute + ObservationBuilder.build_vectorized (+ the average jamming power) on caller tables:
// the device functions of the step, behind an entry point of their own.  LDS = WaveLds (N <= 64) or BigLds<S>.
template <class LDS>
__device__ __forceinline__ void reward_obs_stage(const MarlEnvParams& P, const MarlRewardObsIO& io, LDS& L, int e, int lane,
                                                 int num, int ncent, int eband) {
    const int n = P.N, m = P.M, KS = n + 1;
    if (lane < m) {
        const size_t o = (size_t)e * m + lane;
        L.jx[lane] = io.jam_x[o]; L.jy[lane] = io.jam_y[o]; L.jb[lane] = io.jam_band[o];
        L.asg[lane] = io.assign[o];
    }
    for (int s = lane; s < ncent; s += WAVE) {
        L.cx[s] = io.cent_x[(size_t)e * KS + s];
        L.cy[s] = io.cent_y[(size_t)e * KS + s];
        L.csz[s] = io.cent_size[(size_t)e * KS + s];
    }
    WSYNC();
    const size_t oj = (size_t)e * m + (lane < m ? lane : 0);
    const RewardOut rw = reward_obs_wave(P, n, m, lane, L, eband, num, ncent, io.lambda2[e], io.lambda2_initial[e],
                                         io.vx[oj], io.vy[oj]);
    if (lane < m && io.obs) {
        float* o = io.obs + ((size_t)e * m + lane) * MARL_OBS_DIM;
        o[0] = rw.o0; o[1] = rw.o1; o[2] = rw.o2; o[3] = rw.o3; o[4] = rw.o4;
    }
    if (lane == 0) {
        if (io.reward) io.reward[e] = (float)rw.total;
        if (io.reward_terms) {
            float* r = io.reward_terms + (size_t)e * 5;
            r[0] = rw.r0; r[1] = rw.r1; r[2] = rw.r2; r[3] = rw.r3; r[4] = rw.r4;
        }
    }
}

__global__ __launch_bounds__(WAVE) void reward_obs_kernel(MarlEnvParams P, MarlRewardObsIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    WaveLds& L = carve(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N;
    int eband = io.enemy_band[e];
    eband = eband < 0 ? 0 : (eband > 3 ? 3 : eband);
    int ncent = io.ncent[e];
    ncent = ncent < 0 ? 0 : (ncent > n + 1 ? n + 1 : ncent);
    const bool power = io.avg_power_dbm && io.enemy_x && io.enemy_y;
    if (power && lane < n) L.sxy[lane] = make_double2(io.enemy_x[(size_t)e * n + lane], io.enemy_y[(size_t)e * n + lane]);
    reward_obs_stage(P, io, L, e, lane, io.n_clusters[e], ncent, eband);
    if (power) {
        const double avgp = avg_power_wave(P, n, P.M, lane, L, eband);
        if (lane == 0) io.avg_power_dbm[e] = avgp;
    }
}

__global__ void fill_noise_kernel(MarlEnvParams P, MarlEnvState S, uint64_t seed, uint64_t env_offset, double* noise) {
    const int e = blockIdx.x, i = threadIdx.x;
    if (i >= P.N) return;
    double nx, ny;
    const double sigma = P.enemy_mode == MARL_ENEMY_COORDINATED_TURN ? P.noise_std * 0.5 : P.noise_std;
    noise_pair(seed, env_offset + e, (uint32_t)(S.tick[e] + 1), (uint32_t)i, sigma, nx, ny);
    noise[((size_t)e * P.N + i) * 2] = nx;
    noise[((size_t)e * P.N + i) * 2 + 1] = ny;
}

// ====================================================================== 64 < N <= 256
// One wavefront still advances one environment, each lane now owning S = ceil(N/64)
// enemies: enemy i = s*64 + lane lives in slot s of lane (i & 63), and an N-bit row is S
// 64-bit words, so that word s of a ballot over slot s is exactly word s of a vertex set.
// The stages are the ones above with a slot loop around them.  lambda2 uses the same
// complement compaction; with n' <= 64 touched vertices it runs the register-resident
// solver above, beyond that the n' x n' matrix lives in a caller-provided global workspace
// (column-major, lane j touches only the rows it owns, so no cross-lane visibility is
// needed) -- correct, but memory-bound.
template <int S>
// Three groups of arrays are never live together and share their storage: the previous centroids (a re-clustering
// step, before the link rows are built), the compacted complement rows (from the compaction to their hand-over: into
// registers, or to global memory for the launches that follow) and the solver's vectors (after that).  20 KB instead
// of 32 KB at S = 4: eight envs per compute unit instead of five.
struct BigLds {
    static constexpr int NN = WAVE * S;
    double sx[NN], sy[NN];
    double jx[WAVE], jy[WAVE];
    union {
        struct {
            double2 de[NN];
            __attribute__((aligned(16))) double uvec[NN], qvec[NN];
        };
        unsigned long long xfer[NN * S];
        struct { double px[NN + 2], py[NN + 2]; };
    };
    double cx[NN + 2], cy[NN + 2];
    int csz[NN + 2];
    int lab[NN];
    int asg[WAVE];
    int jb[WAVE];
};

__host__ __device__ inline int big_ld(int n) { return (n + 7) & ~7; }
constexpr int TEAM_MAX_NP = 208;        // largest eigenproblem of the team solver (104 columns per thread)
// The workspace of N > 64: E regions of ld x ld doubles (compacted rows for the team solver at their
// head, or the whole matrix for the global-memory solver), then the list of pending matrix sizes.
__host__ __device__ inline int* workspace_pending(double* ws, int N, int E) {
    const size_t ld = (size_t)big_ld(N);
    return reinterpret_cast<int*>(ws + (size_t)E * ld * ld);
}
// ... then the envs that the Lanczos launch leaves to the team solver: a count and up to TEAM_LIST_MAX env indices (the
// team launch that follows it has that many workgroups instead of one per env, which at 50 ns per workgroup that only
// looks and leaves would cost more than its solves; a failure beyond the list is solved by the wave that met it).
constexpr int TEAM_LIST_MAX = 1024;
// Set by marl_env_step in the parameter copy its kernels get (never by a caller): the Lanczos launch on bit rows follows
// the front launch, so eigenproblems beyond the team solver's size are left pending too.
constexpr int STAGE_ROWS_FOLLOW = 0x100;
__host__ __device__ inline int* workspace_team_list(double* ws, int N, int E) {
    return workspace_pending(ws, N, E) + (((size_t)E + 63) & ~(size_t)63);
}

template <int S>
__device__ __forceinline__ uint64_t word_mask(int n, int k) {      // valid vertex bits of word k
    const int cnt = n - k * WAVE;
    return cnt >= WAVE ? ~0ull : (cnt <= 0 ? 0ull : ((1ull << cnt) - 1));
}

// rows[s][k]: word k of the row of enemy s*64+lane.  `jam` = false: FSPL adjacency; true: jamming mask.
template <int S, bool JAM>
__device__ __forceinline__ void pair_rows_big(uint64_t (&rows)[S][S], const BigLds<S>& L, int n, int m, int lane,
                                              int eband, double d2max) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        const double xi = L.sx[i], yi = L.sy[i];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t word = 0;
            const int cnt = min(WAVE, n - k * WAVE);
            if constexpr (!JAM) {
                for (int b = 0; b < cnt; ++b) {
                    const double dx = xi - L.sx[k * WAVE + b], dy = yi - L.sy[k * WAVE + b];
                    const double d2 = dx * dx + dy * dy;
                    word |= (uint64_t)(d2 <= d2max) << b;
                }
            } else {
                for (int q = 0; q < m; ++q) {
                    if (L.jb[q] != eband) continue;
                    const double qx = L.jx[q], qy = L.jy[q];
                    for (int b = 0; b < cnt; ++b) {
                        const double mx = (xi + L.sx[k * WAVE + b]) / 2, my = (yi + L.sy[k * WAVE + b]) / 2;
                        const double dx = qx - mx, dy = qy - my;
                        const double d2 = dx * dx + dy * dy;
                        word |= (uint64_t)(d2 <= d2max) << b;
                    }
                }
            }
            if (k == s) word &= ~(1ull << lane);
            rows[s][k] = (i < n) ? word : 0ull;
        }
    }
}

// The step's version of the two passes above, fused: one loop over the enemies builds the adjacency and
// the jamming words of a 64-column chunk, with the jamming test in the scaled form of pair_pass (exact, see
// there) for up to four matching jammers at a time, and without the adjacency test when the communication
// range covers the arena (`arena` > 0 enables that shortcut; positions inside a step lie in [0, A]^2).
template <int KJ, bool ADJ, int S>
__device__ __forceinline__ void pair_chunk_big(const BigLds<S>& L, double xi, double yi, int base, int cnt,
                                               double thr_comm, double thr_jam4, const double (&q2x)[4],
                                               const double (&q2y)[4], uint64_t& aw, uint64_t& jw) {
    uint32_t a_lo = 0, a_hi = 0, j_lo = 0, j_hi = 0;
    auto body = [&](int b, uint32_t bit, uint32_t& a, uint32_t& jm) {
        const double xj = L.sx[base + b], yj = L.sy[base + b];
        if constexpr (ADJ) {
            const double dx = xi - xj, dy = yi - yj;
            a |= (dx * dx + dy * dy <= thr_comm) ? bit : 0u;
        }
        if constexpr (KJ > 0) {
            const double sx = xi + xj, sy = yi + yj;
            bool hit = false;
            static_for<0, KJ>([&](auto k) {
                const double DX = q2x[decltype(k)::value] - sx, DY = q2y[decltype(k)::value] - sy;
                hit |= (DX * DX + DY * DY <= thr_jam4);
            });
            jm |= hit ? bit : 0u;
        }
    };
    const int c_lo = cnt < 32 ? cnt : 32;
    for (int b = 0; b < c_lo; ++b) body(b, 1u << b, a_lo, j_lo);
    for (int b = 32; b < cnt; ++b) body(b, 1u << (b - 32), a_hi, j_hi);
    if constexpr (ADJ) aw = ((uint64_t)a_hi << 32) | a_lo;
    jw |= ((uint64_t)j_hi << 32) | j_lo;
}

template <int S>
__device__ __forceinline__ void pair_rows_big_fused(uint64_t (&adj)[S][S], uint64_t (&jam)[S][S], const BigLds<S>& L,
                                                    int n, int m, int lane, int eband, double d2_comm, double d2_jam,
                                                    double arena) {
    const bool all_linked = arena > 0.0 && d2_comm >= 2.000001 * arena * arena;
    const double thr4 = 4.0 * d2_jam;
    uint64_t mm = ballot64(lane < m && L.jb[lane] == eband);
    double q2x[4], q2y[4];
    auto next_four = [&]() {
        int taken = 0;
        static_for<0, 4>([&](auto k) {
            double x = 1e300, y = 1e300;           // squares overflow to +inf: never within range
            if (mm) {
                const int j = __ffsll((unsigned long long)mm) - 1;
                mm &= mm - 1;
                x = 2.0 * L.jx[j]; y = 2.0 * L.jy[j];
                ++taken;
            }
            q2x[decltype(k)::value] = x; q2y[decltype(k)::value] = y;
        });
        return taken;
    };
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < S; ++k) { adj[s][k] = 0; jam[s][k] = 0; }
    bool first = true;
    do {
        const int kj = next_four();
        const bool with_adj = first && !all_linked;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = s * WAVE + lane;
            const double xi = L.sx[i], yi = L.sy[i];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int cnt = min(WAVE, n - k * WAVE);
                if (cnt <= 0) continue;
                uint64_t aw = 0, jw = jam[s][k];
                if (with_adj) {
                    switch (kj) {
                        case 0: pair_chunk_big<0, true, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        case 1: pair_chunk_big<1, true, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        case 2: pair_chunk_big<2, true, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        case 3: pair_chunk_big<3, true, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        default: pair_chunk_big<4, true, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                    }
                    adj[s][k] = aw;
                } else if (kj > 0) {
                    switch (kj) {
                        case 1: pair_chunk_big<1, false, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        case 2: pair_chunk_big<2, false, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        case 3: pair_chunk_big<3, false, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                        default: pair_chunk_big<4, false, S>(L, xi, yi, k * WAVE, cnt, d2_comm, thr4, q2x, q2y, aw, jw); break;
                    }
                }
                jam[s][k] = jw;
            }
        }
        first = false;
    } while (mm);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t keep = (i < n) ? word_mask<S>(n, k) : 0ull;
            if (k == s) keep &= ~(1ull << lane);
            if (all_linked) adj[s][k] = keep;
            else adj[s][k] &= keep;
            jam[s][k] &= keep;
        }
    }
}

// The float32 first stage of the jamming test (jam_pass_f32: two thresholds around the exact one, packed pipe, sign
// bits) for rows of more than 64 columns: `cnt` (even, 2 .. 32) columns from j0 on, against KJ jammers.
template <int KJ>
__device__ __forceinline__ void jam_word_f32(const float* sxf, const float* syf, float xi, float yi, int j0, int cnt,
                                             float tl, float th, const float (&q2x)[4], const float (&q2y)[4],
                                             uint32_t& sb, uint32_t& mb) {
    const v2f xi2 = {xi, xi}, yi2 = {yi, yi}, tl2 = {tl, tl}, th2 = {th, th};
    uint32_t sw = 0, mw = 0;
    for (int j = j0; j < j0 + cnt; j += 2) {
        const v2f xj = *reinterpret_cast<const v2f*>(&sxf[j]), yj = *reinterpret_cast<const v2f*>(&syf[j]);
        const v2f sx = xi2 + xj, sy = yi2 + yj;
        v2f dmin = {0.0f, 0.0f};
        static_for<0, KJ>([&](auto k) {
            const v2f qx = {q2x[decltype(k)::value], q2x[decltype(k)::value]};
            const v2f qy = {q2y[decltype(k)::value], q2y[decltype(k)::value]};
            const v2f DX = qx - sx, DY = qy - sy;
            const v2f d2 = __builtin_elementwise_fma(DX, DX, DY * DY);
            if constexpr (decltype(k)::value == 0) dmin = d2;
            else dmin = v2f{fminf(dmin.x, d2.x), fminf(dmin.y, d2.y)};
        });
        const v2f a = dmin - tl2, b = dmin - th2;
        sw = __builtin_amdgcn_alignbit(sw, __float_as_uint(a.x), 31);
        sw = __builtin_amdgcn_alignbit(sw, __float_as_uint(a.y), 31);
        mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(b.x), 31);
        mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(b.y), 31);
    }
    sb |= __brev(sw) >> (32 - cnt);
    mb |= __brev(mw) >> (32 - cnt);
}

// Jamming rows of a step whose adjacency is the complete graph (communication range beyond the arena's diagonal): the
// float32 stage over every (row, column pair, matching jammer), then the exact float64 expression of pair_chunk_big
// for the pairs that fell between its two thresholds, which therefore still decides every bit.  The float32 copies of
// the positions lie in L.xfer, which nothing uses between the re-clustering and the compaction of the rows.
template <int S>
__device__ __forceinline__ void jam_rows_big_f32(uint64_t (&adj)[S][S], uint64_t (&jam)[S][S], BigLds<S>& L, int n, int m,
                                                 int lane, int eband, double d2_jam, float thr_lo, float thr_hi) {
    constexpr int NN = WAVE * S;
    float* sxf = reinterpret_cast<float*>(L.xfer);
    float* syf = sxf + NN;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        sxf[i] = (i < n) ? (float)L.sx[i] : 1e30f;          // 1e30: squares overflow to +inf, never within range
        syf[i] = (i < n) ? (float)L.sy[i] : 1e30f;
    }
    WSYNC();
    const uint64_t matching = ballot64(lane < m && L.jb[lane] == eband);
    uint64_t sure[S][S], maybe[S][S];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < S; ++k) { sure[s][k] = 0; maybe[s][k] = 0; }
    const float tl = nextafterf(thr_lo, INFINITY), th = nextafterf(thr_hi, INFINITY);
    uint64_t mm = (d2_jam < 0.0) ? 0ull : matching;
    while (mm) {
        float q2x[4], q2y[4];
        int kj = 0;
        static_for<0, 4>([&](auto k) {
            float x = 1e30f, y = 1e30f;
            if (mm) {
                const int j = __ffsll((unsigned long long)mm) - 1;
                mm &= mm - 1;
                x = (float)(2.0 * L.jx[j]); y = (float)(2.0 * L.jy[j]);
                ++kj;
            }
            q2x[decltype(k)::value] = x; q2y[decltype(k)::value] = y;
        });
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = s * WAVE + lane;
            const float xi = sxf[i], yi = syf[i];
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int left = n - k * WAVE;               // columns of this word: wave-uniform
                if (left <= 0) continue;
                uint32_t s_lo = 0, s_hi = 0, m_lo = 0, m_hi = 0;
                const int c_lo = left < 32 ? ((left + 1) & ~1) : 32;
                const int c_hi = left > 32 ? (left < 64 ? ((left - 32 + 1) & ~1) : 32) : 0;
                switch (kj) {
                    case 1: jam_word_f32<1>(sxf, syf, xi, yi, k * WAVE, c_lo, tl, th, q2x, q2y, s_lo, m_lo);
                            if (c_hi) jam_word_f32<1>(sxf, syf, xi, yi, k * WAVE + 32, c_hi, tl, th, q2x, q2y, s_hi, m_hi); break;
                    case 2: jam_word_f32<2>(sxf, syf, xi, yi, k * WAVE, c_lo, tl, th, q2x, q2y, s_lo, m_lo);
                            if (c_hi) jam_word_f32<2>(sxf, syf, xi, yi, k * WAVE + 32, c_hi, tl, th, q2x, q2y, s_hi, m_hi); break;
                    case 3: jam_word_f32<3>(sxf, syf, xi, yi, k * WAVE, c_lo, tl, th, q2x, q2y, s_lo, m_lo);
                            if (c_hi) jam_word_f32<3>(sxf, syf, xi, yi, k * WAVE + 32, c_hi, tl, th, q2x, q2y, s_hi, m_hi); break;
                    default: jam_word_f32<4>(sxf, syf, xi, yi, k * WAVE, c_lo, tl, th, q2x, q2y, s_lo, m_lo);
                            if (c_hi) jam_word_f32<4>(sxf, syf, xi, yi, k * WAVE + 32, c_hi, tl, th, q2x, q2y, s_hi, m_hi); break;
                }
                sure[s][k] |= ((uint64_t)s_hi << 32) | s_lo;
                maybe[s][k] |= ((uint64_t)m_hi << 32) | m_lo;
            }
        }
    }
    const double thr4 = 4.0 * d2_jam;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        const double xi = L.sx[i < n ? i : 0], yi = L.sy[i < n ? i : 0];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t keep = (i < n) ? word_mask<S>(n, k) : 0ull;
            if (k == s) keep &= ~(1ull << lane);
            adj[s][k] = keep;
            uint64_t jw = sure[s][k] & keep;
            uint64_t open = maybe[s][k] & ~sure[s][k] & keep;
            while (open) {                      // rare: a band of a few parts per million around the radius
                const int b = __ffsll((unsigned long long)open) - 1;
                open &= open - 1;
                const double sx = xi + L.sx[k * WAVE + b], sy = yi + L.sy[k * WAVE + b];
                bool hit = false;
                for (uint64_t mk = matching; mk; mk &= mk - 1) {
                    const int q = __ffsll((unsigned long long)mk) - 1;
                    const double DX = 2.0 * L.jx[q] - sx, DY = 2.0 * L.jy[q] - sy;
                    hit |= (DX * DX + DY * DY <= thr4);
                }
                if (hit) jw |= 1ull << b;
            }
            jam[s][k] = jw;
        }
    }
    WSYNC();                                    // L.xfer is free again
}

template <int S>
__device__ __forceinline__ void store_bits_big(uint32_t* dst, int e, int n, int lane, const uint64_t (&rows)[S][S]) {
    if (dst == nullptr) return;
    const int W = (n + 31) / 32;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        if (i >= n) continue;
        uint32_t* p = dst + ((size_t)e * n + i) * W;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            if (2 * k < W) p[2 * k] = (uint32_t)rows[s][k];
            if (2 * k + 1 < W) p[2 * k + 1] = (uint32_t)(rows[s][k] >> 32);
        }
    }
}

template <int S>
__device__ __forceinline__ bool any_words(const uint64_t (&w)[S]) {
    uint64_t o = 0;
#pragma unroll
    for (int k = 0; k < S; ++k) o |= w[k];
    return o != 0;
}

// hit[s] = lanes whose slot-s vertex has a neighbour in `frontier`
template <int S>
__device__ __forceinline__ void expand_big(const uint64_t (&rows)[S][S], const uint64_t (&frontier)[S], uint64_t (&hit)[S]) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint64_t o = 0;
#pragma unroll
        for (int k = 0; k < S; ++k) o |= rows[s][k] & frontier[k];
        hit[s] = ballot64(o != 0);
    }
}

// mu_max of the complement Laplacian of np <= S * 64 vertices by ONE wave, the matrix in global memory: element (r, c) at
// As[c * ld + r], lane t owns rows t, 64 + t, ...; w[s][q]: word q of the compacted row s * 64 + lane.  The slowest of
// the solvers (every Householder step reads and writes the matrix), for what fits no other: more vertices than the team
// solver holds, or a Lanczos failure that found the team's list full.  L: any LDS block with de[S * 64], uvec, qvec.
template <int S, class LDS>
__device__ __forceinline__ double mu_max_global_wave(const uint64_t (&w)[S][S], int np, int lane, LDS& L, int rounds,
                                                     double* As, int ld) {
    int dmax = 0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int r = s * WAVE + lane;
        if (r < np) {
            uint64_t cw[S];
            int cdeg = 0;
#pragma unroll
            for (int q = 0; q < S; ++q) { cw[q] = w[s][q]; cdeg += __popcll(cw[q]); }
            dmax = max(dmax, cdeg);
#pragma unroll
            for (int q = 0; q < S; ++q) {
                const int cnt = min(WAVE, np - q * WAVE);
                for (int b = 0; b < cnt; ++b) {
                    const int c = q * WAVE + b;
                    As[(size_t)c * ld + r] = (c == r) ? (double)cdeg : -(double)((cw[q] >> b) & 1ull);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
    for (int i = np - 1; i >= 2; --i) {
        const int l = i - 1;
        double u[S], p[S], own_i = 0.0, own_l = 0.0;
        double hh = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int r = s * WAVE + lane;
            const double v = (r <= i) ? As[(size_t)i * ld + r] : 0.0;      // A[r][i] == A[i][r]
            if (r == i) own_i = v;
            u[s] = (r <= l) ? v : 0.0;
            if (r == l) own_l = v;
            hh += u[s] * u[s];
        }
        const double h = wave_sum_dpp(hh);
        const double aii = readlane_d(own_i, i & (WAVE - 1));
        if (lane == 0) L.de[i] = make_double2(aii, h);
        const double f = readlane_d(own_l, l & (WAVE - 1));
        const double g = (f >= 0.0) ? -sqrt(h) : sqrt(h);
        const double h2 = h - f * g;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int r = s * WAVE + lane;
            if (r == l) u[s] = f - g;
            L.uvec[r] = u[s];
            p[s] = 0.0;
        }
        WSYNC();
        for (int c = 0; c <= l; ++c) {
            const double uc = L.uvec[c];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int r = s * WAVE + lane;
                if (s * WAVE <= l && r <= l) p[s] = fma(As[(size_t)c * ld + r], uc, p[s]);
            }
        }
        const double hinv = (h > 0.0) ? 1.0 / h2 : 0.0;
        double kk = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int r = s * WAVE + lane;
            p[s] = (r <= l) ? p[s] * hinv : 0.0;
            kk += u[s] * p[s];
        }
        const double K = wave_sum_dpp(kk) * (0.5 * hinv);
        double q[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int r = s * WAVE + lane;
            q[s] = (r <= l) ? fma(-K, u[s], p[s]) : 0.0;
            L.qvec[r] = q[s];
        }
        WSYNC();
        for (int c = 0; c <= l; ++c) {
            const double uc = L.uvec[c], qc = L.qvec[c];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int r = s * WAVE + lane;
                if (s * WAVE <= l && r <= l) {
                    double* pa = As + (size_t)c * ld + r;
                    *pa = fma(-u[s], qc, fma(-q[s], uc, *pa));
                }
            }
        }
        WSYNC();
    }
    {
        double a11 = 0.0, a10 = 0.0, a00 = 0.0;
        if (lane == 1) { a11 = As[(size_t)ld + 1]; a10 = As[1]; }
        if (lane == 0) a00 = As[0];
        a11 = readlane_d(a11, 1); a10 = readlane_d(a10, 1); a00 = readlane_d(a00, 0);
        if (lane == 0) { L.de[1] = make_double2(a11, a10 * a10); L.de[0] = make_double2(a00, 0.0); }
    }
    WSYNC();
    const double lo = (double)(dmax + 1) * (1.0 - 1e-13) - 1e-13;
    const double hi = (double)min(np, 2 * dmax) * (1.0 + 1e-13) + 1e-13;
    return sturm_largest_wave(L, np, lane, lo, hi, rounds + 1);
}

template <int S>
__device__ __forceinline__ double lambda2_big(const uint64_t (&rows)[S][S], int n, int lane, BigLds<S>& L, int rounds,
                                              double* As, int* pending = nullptr, bool rows_follow = false) {
    // `pending` non-null: an eigenproblem of 65..TEAM_MAX_NP vertices is not solved here; its compacted
    // complement rows (np x S words) go to the head of `As`, *pending = np, and the team kernel
    // (one workgroup of 512 threads per env, matrix in registers) takes over.  *pending = 0 otherwise.
    if (pending) *pending = 0;
    if (n <= 1) return 0.0;
    uint64_t touched[S];
    int deg[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < S; ++k) d += __popcll(rows[s][k]);
        deg[s] = d;
        touched[s] = ballot64(s * WAVE + lane < n && d < n - 1);
    }
    if (!any_words<S>(touched)) return (double)n;
    {   // connectivity from vertex 0
        uint64_t comp[S], frontier[S], hit[S];
#pragma unroll
        for (int k = 0; k < S; ++k) { comp[k] = 0; frontier[k] = 0; }
        comp[0] = 1ull; frontier[0] = 1ull;
        while (any_words<S>(frontier)) {
            expand_big<S>(rows, frontier, hit);
#pragma unroll
            for (int k = 0; k < S; ++k) { const uint64_t nw = hit[k] & ~comp[k]; comp[k] |= nw; frontier[k] = nw; }
        }
        bool all = true;
#pragma unroll
        for (int k = 0; k < S; ++k) all = all && (comp[k] == word_mask<S>(n, k));
        if (!all) return 0.0;
    }
    int np = 0;
    int before[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { before[s] = np; np += __popcll(touched[s]); }
    // compact the complement rows of the touched vertices: bit r of the new row = touched vertex of rank r
    WSYNC();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint64_t cw[S];
#pragma unroll
        for (int q = 0; q < S; ++q) cw[q] = 0;
        int r = 0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t cfull = ~rows[s][k] & word_mask<S>(n, k) & touched[k];
            if (k == s) cfull &= ~(1ull << lane);
            uint64_t t = touched[k];
            while (t) {                         // one step per run of consecutive touched vertices (t, r: wave-uniform)
                const int a = __ffsll((unsigned long long)t) - 1;
                const uint64_t inv = ~(t >> a);                      // zero bits mark the end of the run
                const int len = inv ? __ffsll((unsigned long long)inv) - 1 : 64;
                const uint64_t ones = (len >= 64) ? ~0ull : ((1ull << len) - 1);
                const uint64_t run = (cfull >> a) & ones;
                const int qd = r >> 6, sh = r & 63;
#pragma unroll
                for (int q = 0; q < S; ++q) {
                    if (q == qd) cw[q] |= run << sh;
                    if (q == qd + 1 && sh != 0) cw[q] |= run >> (64 - sh);      // the part of the run that spills over
                }
                r += len;
                t = (a + len >= 64) ? 0ull : (t & ~(ones << a));
            }
        }
        if ((touched[s] >> lane) & 1ull) {
            const int rank = before[s] + __popcll(touched[s] & ((1ull << lane) - 1));
#pragma unroll
            for (int q = 0; q < S; ++q) L.xfer[rank * S + q] = cw[q];
        }
    }
    WSYNC();
    if (np <= WAVE) {
        const uint64_t crow = (lane < np) ? L.xfer[lane * S] : 0ull;
        WSYNC();                                // the solver's arrays lie where the rows were
        return (double)n - mu_max_in_wave<WAVE>(crow, np, lane, L, rounds, (double)n);
    }
    // (rows_follow: the Lanczos launch on bit rows comes next, which takes any size up to S * 64 and, should it fail
    // on a problem beyond the team solver, falls back to the global-memory solver below by itself)
    if (pending != nullptr && (np <= TEAM_MAX_NP || rows_follow)) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(As);
        for (int k = lane; k < np * S; k += WAVE) out[k] = L.xfer[k];
        *pending = np;
        return 0.0;
    }
    // ---- n' beyond the team solver: one wave, matrix in the global workspace
    uint64_t w[S][S];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int q = 0; q < S; ++q) w[s][q] = (s * WAVE + lane < np) ? L.xfer[(s * WAVE + lane) * S + q] : 0ull;
    WSYNC();                                    // the solver's arrays lie where the rows were
    return (double)n - mu_max_global_wave<S>(w, np, lane, L, rounds, As, big_ld(n));
}

// DBSCAN + centroids for S slots per lane; labels by reference, tables in L.cx/L.cy/L.csz.
template <int S>
__device__ __forceinline__ void dbscan_big(const MarlEnvParams& P, int n, int lane, BigLds<S>& L, int (&label)[S],
                                           int& n_clusters, int& ncent) {
    const double eps2 = P.eps * P.eps;
    uint64_t nb[S][S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        const double xi = L.sx[i], yi = L.sy[i];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t word = 0;
            const int cnt = min(WAVE, n - k * WAVE);
            for (int b = 0; b < cnt; ++b) {
                const double dx = xi - L.sx[k * WAVE + b], dy = yi - L.sy[k * WAVE + b];
                const double d2 = dx * dx + dy * dy;
                word |= (uint64_t)(d2 <= eps2) << b;
            }
            nb[s][k] = (i < n) ? word : 0ull;
        }
    }
    uint64_t core[S], unvisited[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < S; ++k) cnt += __popcll(nb[s][k]);
        core[s] = ballot64(s * WAVE + lane < n && cnt >= P.min_samples);
        unvisited[s] = core[s];
        label[s] = -1;
    }
    int num = 0;
    while (any_words<S>(unvisited)) {
        uint64_t comp[S], frontier[S], reached[S], hit[S];
        bool seeded = false;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            comp[k] = 0; reached[k] = 0;
            if (!seeded && unvisited[k]) { comp[k] = unvisited[k] & (~unvisited[k] + 1); seeded = true; }
            frontier[k] = comp[k];
        }
        while (any_words<S>(frontier)) {
            expand_big<S>(nb, frontier, hit);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                reached[k] |= hit[k];
                const uint64_t nw = hit[k] & core[k] & ~comp[k];
                comp[k] |= nw;
                frontier[k] = nw;
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint64_t members = comp[s] | (reached[s] & ~core[s]);
            if (((members >> lane) & 1ull) && label[s] == -1) label[s] = num;
            unvisited[s] &= ~comp[s];
        }
        ++num;
    }
    bool noise_mine = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        L.lab[s * WAVE + lane] = label[s];
        noise_mine = noise_mine || (s * WAVE + lane < n && label[s] == -1);
    }
    const bool has_noise = ballot64(noise_mine) != 0;
    const int entries = num + (has_noise ? 1 : 0);
    WSYNC();
    for (int c = lane; c < entries; c += WAVE) {
        const int id = (c < num) ? c : -1;
        double ax = 0.0, ay = 0.0;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            if (L.lab[i] == id) { ax = ax + L.sx[i]; ay = ay + L.sy[i]; ++cnt; }
        }
        L.cx[c] = ax / (double)cnt;
        L.cy[c] = ay / (double)cnt;
        L.csz[c] = cnt;
    }
    n_clusters = num;
    ncent = entries;
    if (num == 0) {
        if (lane == 0) { L.cx[0] = P.arena_size / 2; L.cy[0] = P.arena_size / 2; L.csz[0] = n; }
        ncent = 1;
    }
    WSYNC();
}

template <int S>
__device__ __forceinline__ double avg_power_big(const MarlEnvParams& P, int n, int m, int lane, const BigLds<S>& L, int eband) {
    int nm = 0;
    for (int k = 0; k < m; ++k) nm += (L.jb[k] == eband);
    if (nm == 0) return -80.0;
    LinkPower acc(P, eband);
    const int half = (n - 1) >> 1, opposite = (n & 1) ? 0 : n >> 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        const double xi = L.sx[i < n ? i : 0], yi = L.sy[i < n ? i : 0];
        for (int k = 1; k <= half + (opposite ? 1 : 0); ++k) {
            int j = i + k;
            j = j >= n ? j - n : j;
            j = i < n ? j : 0;
            const double sx = xi + L.sx[j], sy = yi + L.sy[j];
            double D2 = INFINITY;
            for (int q = 0; q < m; ++q) {
                if (L.jb[q] != eband) continue;
                const double dx = fma(2.0, L.jx[q], -sx), dy = fma(2.0, L.jy[q], -sy);
                D2 = fmin(D2, fma(dx, dx, dy * dy));
            }
            acc.add(D2, k <= half ? i < n : i < opposite);
        }
    }
    return acc.dbm();
}

template <int S>
__device__ __forceinline__ void recluster_big(const MarlEnvParams& P, const MarlEnvState& St, int e, int n, int lane,
                                              BigLds<S>& L, int& num, int& ncent, int& pnum, int& pncent) {
    const int KS = n + 1;
    pnum = num; pncent = ncent;
    for (int s = lane; s < pncent; s += WAVE) {
        L.px[s] = St.cent_x[(size_t)e * KS + s];
        L.py[s] = St.cent_y[(size_t)e * KS + s];
    }
    int label[S];
    dbscan_big<S>(P, n, lane, L, label, num, ncent);
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (s * WAVE + lane < n) St.labels[(size_t)e * n + s * WAVE + lane] = label[s];
    for (int s = lane; s < pncent; s += WAVE) {
        St.prev_x[(size_t)e * KS + s] = L.px[s];
        St.prev_y[(size_t)e * KS + s] = L.py[s];
    }
    for (int s = lane; s < ncent; s += WAVE) {
        St.cent_x[(size_t)e * KS + s] = L.cx[s];
        St.cent_y[(size_t)e * KS + s] = L.cy[s];
        St.cent_size[(size_t)e * KS + s] = L.csz[s];
    }
    if (lane == 0) {
        St.prev_n_clusters[e] = pnum; St.prev_ncent[e] = pncent;
        St.n_clusters[e] = num; St.ncent[e] = ncent;
    }
}

template <int S>
__global__ __launch_bounds__(WAVE) void env_step_big_kernel(MarlEnvParams P, MarlEnvState St, MarlStepIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int n = P.N, m = P.M, KS = n + 1;
    const int step = St.step_count[e] + 1;
    const int eband = St.enemy_band[e];
    const double arena = P.arena_size;
    float raw_vx = 0.0f, raw_vy = 0.0f;
    if (lane < m) {
        const float* a = io.actions + ((size_t)e * m + lane) * 3;
        raw_vx = a[0]; raw_vy = a[1];
        int band = (int)a[2];
        band = band < 0 ? 0 : (band > MARL_NUM_BANDS - 1 ? MARL_NUM_BANDS - 1 : band);
        const float cvx = fminf(fmaxf(raw_vx, -P.v_max_f32), P.v_max_f32);
        const float cvy = fminf(fmaxf(raw_vy, -P.v_max_f32), P.v_max_f32);
        const float sx_ = cvx * P.dt_f32, sy_ = cvy * P.dt_f32;
        const size_t o = (size_t)e * m + lane;
        double x = St.jam_x[o] + (double)sx_, y = St.jam_y[o] + (double)sy_;
        x = fmin(fmax(x, 0.0), arena);
        y = fmin(fmax(y, 0.0), arena);
        St.jam_x[o] = x; St.jam_y[o] = y; St.jam_band[o] = band;
        St.last_vx[o] = raw_vx; St.last_vy[o] = raw_vy;
        L.jx[lane] = x; L.jy[lane] = y; L.jb[lane] = band;
        L.asg[lane] = St.assign[o];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        if (i < n) {
            double x, y;
            enemy_move(P, St, io, e, i, St.tick[e] + 1, x, y);
            L.sx[i] = x; L.sy[i] = y;
        }
    }
    int num = St.n_clusters[e], ncent = St.ncent[e];
    WSYNC();
    if (step % P.K_recompute == 0 && !(P.stage_flags & MARL_STAGE_NO_RECLUSTER)) {
        int pnum, pncent;
        recluster_big<S>(P, St, e, n, lane, L, num, ncent, pnum, pncent);
        if (should_reassign_wave(P, lane, L, num, ncent, pnum, pncent)) {
            assign_proportional_wave(m, lane, L, num, ncent);
            if (lane < m) St.assign[(size_t)e * m + lane] = L.asg[lane];
        }
    } else {
        for (int s = lane; s < ncent; s += WAVE) {
            L.cx[s] = St.cent_x[(size_t)e * KS + s];
            L.cy[s] = St.cent_y[(size_t)e * KS + s];
            L.csz[s] = St.cent_size[(size_t)e * KS + s];
        }
        WSYNC();
    }
    uint64_t adj[S][S], jam[S][S];
    if (arena > 0.0 && P.comm_d2_max[eband] >= 2.000001 * arena * arena)
        jam_rows_big_f32<S>(adj, jam, L, n, m, lane, eband, P.jam_d2_max[eband], P.jam_lo_f32[eband], P.jam_hi_f32[eband]);
    else
        pair_rows_big_fused<S>(adj, jam, L, n, m, lane, eband, P.comm_d2_max[eband], P.jam_d2_max[eband], arena);
    store_bits_big<S>(io.adj_bits, e, n, lane, adj);
    store_bits_big<S>(io.jam_bits, e, n, lane, jam);
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < S; ++k) adj[s][k] &= ~jam[s][k];
    // eigenproblems of 65..208 vertices are left to env_step_team_kernel, which also finishes the reward
    const int ld = big_ld(n);
    int pending;
    const double lam = lambda2_big<S>(adj, n, lane, L, P.lambda2_rounds, St.workspace + (size_t)e * ld * ld, &pending,
                                      (P.stage_flags & STAGE_ROWS_FOLLOW) != 0);
    const double lam0 = St.lambda2_initial[e];
    const RewardOut rw = reward_obs_wave(P, n, m, lane, L, eband, num, ncent, lam, lam0, raw_vx, raw_vy);
    if (lane < m && io.obs) {
        float* o = io.obs + ((size_t)e * m + lane) * MARL_OBS_DIM;
        o[0] = rw.o0; o[1] = rw.o1; o[2] = rw.o2; o[3] = rw.o3; o[4] = rw.o4;
    }
    if (io.reward_terms && lane == 0) {
        float* r = io.reward_terms + (size_t)e * 5;
        r[0] = rw.r0; r[1] = rw.r1; r[2] = rw.r2; r[3] = rw.r3; r[4] = rw.r4;
    }
    double avgp = 0.0;
    if (io.avg_power_dbm) avgp = avg_power_big<S>(P, n, m, lane, L, eband);
    if (lane == 0) {
        St.step_count[e] = step;
        St.tick[e] = St.tick[e] + 1;
        io.truncated[e] = (uint8_t)(step >= P.max_steps);
        if (io.avg_power_dbm) io.avg_power_dbm[e] = avgp;
        workspace_pending(St.workspace, n, St.E)[e] = pending;
        if (e == 0) workspace_team_list(St.workspace, n, St.E)[0] = 0;
        if (pending) {
            double* part = St.solve_partial + (size_t)e * 4;
            part[0] = rw.d2; part[1] = rw.d3; part[2] = (double)rw.r3; part[3] = rw.d5;
        } else {
            finish_env(P, St, io, e, step, rw.total, lam, lam0);
        }
    }
}


template <int S>
__global__ __launch_bounds__(WAVE) void env_reset_big_kernel(MarlEnvParams P, MarlEnvState St, MarlResetIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    if (io.mask && io.mask[e] == 0) return;
    const int n = P.N, m = P.M;
    const double arena = P.arena_size;
    const uint64_t genv = io.env_offset + e;
    const uint32_t k0 = (uint32_t)io.seed, k1 = (uint32_t)(io.seed >> 32);
    int num = St.n_clusters[e], ncent = St.ncent[e];
    ncent = ncent < 0 ? 0 : (ncent > P.N + 1 ? P.N + 1 : ncent);
    int eband = 2;
    double lam0 = 0.0;
    const bool host_init = io.enemy_pos != nullptr;
    const int attempts = host_init ? 1 : (P.max_reset_attempts > 0 ? P.max_reset_attempts : 1);
    const int ld = big_ld(n);
    for (int attempt = 0; attempt < attempts; ++attempt) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = s * WAVE + lane;
            if (i < n) {
                double x, y;
                enemy_spawn(P, St, io, e, i, attempt, x, y);
                L.sx[i] = x; L.sy[i] = y;
            }
        }
        if (io.enemy_band) {
            eband = io.enemy_band[e];
        } else {
            uint32_t r[4];
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ STREAM_RESET_ENV, episode_key(io, St, e),
                       (uint32_t)attempt, k0, k1, r);
            eband = (int)(r[0] & 3u);
        }
        eband = eband < 0 ? 0 : (eband > 3 ? 3 : eband);
        WSYNC();
        int pnum, pncent;
        recluster_big<S>(P, St, e, n, lane, L, num, ncent, pnum, pncent);
        WSYNC();
        uint64_t adj[S][S];
        pair_rows_big<S, false>(adj, L, n, m, lane, eband, P.comm_d2_max[eband]);
        lam0 = lambda2_big<S>(adj, n, lane, L, P.lambda2_rounds, St.workspace + (size_t)e * ld * ld);
        if (lam0 >= P.min_lambda2_initial) break;
    }
    if (lane < m) L.asg[lane] = MARL_ASSIGN_NONE;
    WSYNC();
    if (!P.random_jammer_start) assign_proportional_wave(m, lane, L, num, ncent);
    if (lane < m) {
        const size_t o = (size_t)e * m + lane;
        double ix, iy;
        int band;
        uint32_t r[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
        if (io.jam_init == nullptr || io.jam_band == nullptr) {
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)lane << 8) ^ STREAM_RESET_JAMMER,
                       episode_key(io, St, e), 0u, k0, k1, r);
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)lane << 8) ^ STREAM_RESET_JAMMER,
                       episode_key(io, St, e), 1u, k0, k1, r2);
        }
        if (io.jam_init) { ix = io.jam_init[o * 2]; iy = io.jam_init[o * 2 + 1]; }
        else if (P.random_jammer_start) {
            ix = 0.1 * arena + (0.9 * arena - 0.1 * arena) * u01(r[0], r[1]);
            iy = 0.1 * arena + (0.9 * arena - 0.1 * arena) * u01(r[2], r[3]);
        } else {
            ix = -P.spread + (P.spread + P.spread) * u01(r[0], r[1]);
            iy = -P.spread + (P.spread + P.spread) * u01(r[2], r[3]);
        }
        band = io.jam_band ? io.jam_band[o] : (int)(r2[0] & 3u);
        double x = ix, y = iy;
        if (!P.random_jammer_start) {
            const int c = L.asg[lane];
            x = 0.0; y = 0.0;
            int slot = -1;
            if (c >= 0 && c < num) slot = c;
            else if (c == -1 && ncent > num) slot = num;
            if (slot >= 0) {
                x = fmin(fmax(L.cx[slot] + ix, 0.0), arena);
                y = fmin(fmax(L.cy[slot] + iy, 0.0), arena);
            }
        }
        St.jam_x[o] = x; St.jam_y[o] = y; St.jam_band[o] = band;
        St.last_vx[o] = 0.0f; St.last_vy[o] = 0.0f;
        St.assign[o] = L.asg[lane];
        L.jx[lane] = x; L.jy[lane] = y; L.jb[lane] = band;
    }
    WSYNC();
    const RewardOut rw = reward_obs_wave(P, n, m, lane, L, eband, num, ncent, lam0, lam0, 0.0f, 0.0f);
    if (lane < m && io.obs) {
        float* o = io.obs + ((size_t)e * m + lane) * MARL_OBS_DIM;
        o[0] = rw.o0; o[1] = rw.o1; o[2] = rw.o2; o[3] = rw.o3; o[4] = rw.o4;
    }
    if (lane == 0) {
        St.step_count[e] = 0;
        St.enemy_band[e] = eband;
        St.lambda2_initial[e] = lam0;
        St.episode_return[e] = 0.0;
    }
}

template <int S>
__global__ __launch_bounds__(WAVE) void pair_bits_big_kernel(MarlEnvParams P, const double* x, const double* y,
                                                             const int32_t* band, const double* jx, const double* jy,
                                                             const int32_t* jb, uint32_t* bits, int jam) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N, m = P.M;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        if (i < n) { L.sx[i] = x[(size_t)e * n + i]; L.sy[i] = y[(size_t)e * n + i]; }
    }
    if (jam && lane < m) {
        L.jx[lane] = jx[(size_t)e * m + lane]; L.jy[lane] = jy[(size_t)e * m + lane];
        L.jb[lane] = jb[(size_t)e * m + lane];
    }
    WSYNC();
    int b = band[e]; b = b < 0 ? 0 : (b > 3 ? 3 : b);
    uint64_t rows[S][S];
    if (jam) pair_rows_big<S, true>(rows, L, n, m, lane, b, P.jam_d2_max[b]);
    else pair_rows_big<S, false>(rows, L, n, m, lane, b, P.comm_d2_max[b]);
    store_bits_big<S>(bits, e, n, lane, rows);
}

// ------------------------------------------------------------------ team solver (64 < n' <= 208)
// One workgroup of 512 threads per eigenproblem, the matrix in registers: row r belongs to two
// threads 16 lanes apart (a wave holds 32 rows: lanes 0..15 and 32..47 the columns of parity 0 of rows
// 0..15 and 16..31, lanes 16..31 and 48..63 those of parity 1), column c = 2k+h in a[k] of the thread of
// parity h.  The Householder step is the one of mu_max_compact_wave: u and p are published in LDS split
// by column parity and reach the multiply-adds by row_newbcast (TeamCols), the two half-row products
// meet by one lane swap, and the sums over the rows go through a wave sum plus one LDS word per wave.
// Column loops of the team solver: a[k] is the thread's k-th local column.  u (and p) of the thread's
// column parity sit in LDS; a chunk of 16 of them is read into one register (lane t: entry
// 16 q + t % 16) and handed to the 16 multiply-adds of the chunk by row_newbcast, the next chunk being
// read while this one is used.  Every lane of a row of 16 must want the same column: the two threads
// of a matrix row are therefore 16 lanes apart (see mu_max_team), not neighbours.
template <int CH, int B>
struct TeamCols {
    static constexpr int C0 = B * 4;
    // LDS holds the pivot column v; u differs from it in entry l only (u_l = v_l - g).  The lane whose chunk entry
    // is that one (fq == first column of the chunk) swaps it in after the read.
    static __device__ __forceinline__ double fixed(double chunk, int fq, double ul) { return (fq == C0) ? ul : chunk; }
    static __device__ __forceinline__ void matvec(const double (&a)[CH], const double* u16, int lh, int fq, double ul,
                                                  double& p0, double& p1, double cur, double nxt) {
        if constexpr (C0 < CH) {
            if constexpr (B > 0) {
                if (C0 > lh) return;
            }
            if constexpr (C0 % 16 == 0) {
                cur = fixed(B > 0 ? nxt : cur, fq, ul);
                if constexpr (C0 + 16 < CH) nxt = u16[C0 + 16];
            }
            static_for<0, 4>([&](auto jj) {
                constexpr int c = C0 + decltype(jj)::value;
                if constexpr (c < CH) fmac_row_bcast<c % 16, (c % 16 < 2)>((c & 1) ? p1 : p0, cur, a[c]);
            });
            TeamCols<CH, B + 1>::matvec(a, u16, lh, fq, ul, p0, p1, cur, nxt);
        }
    }
    // a[c] += nu * p_c + nw * u_c  (nu = -u_row, nw = -w_row); returns a[lh] (the thread's part of the next
    // pivot column) from the last block it visits.
    static __device__ __forceinline__ double update(double (&a)[CH], const double& nu, const double& nw, int lh, int fq,
                                                    double ul, const double* u16, const double* p16, double cu,
                                                    double nu16, double cp, double np16) {
        if constexpr (C0 % 16 == 0) {
            cu = fixed(B > 0 ? nu16 : cu, fq, ul);
            if constexpr (B > 0) cp = np16;
            if constexpr (C0 + 16 < CH) { nu16 = u16[C0 + 16]; np16 = p16[C0 + 16]; }
        }
        static_for<0, 4>([&](auto jj) {
            constexpr int c = C0 + decltype(jj)::value;
            if constexpr (c < CH) fmac_row_bcast<c % 16, (c % 16 < 2)>(a[c], cp, nu);
        });
        static_for<0, 4>([&](auto jj) {
            constexpr int c = C0 + decltype(jj)::value;
            if constexpr (c < CH) fmac_row_bcast<c % 16, false>(a[c], cu, nw);
        });
        if constexpr (C0 + 4 < CH) {
            if (lh >= C0 + 4) return TeamCols<CH, B + 1>::update(a, nu, nw, lh, fq, ul, u16, p16, cu, nu16, cp, np16);
        }
        const int w = lh - C0;                                   // wave-uniform
        double v = a[C0];
        static_for<1, 4>([&](auto jj) {
            constexpr int c = C0 + decltype(jj)::value;
            if constexpr (c < CH) v = (w == decltype(jj)::value) ? a[c] : v;
        });
        return v;
    }
};
// The value of the lane 16 away inside the same half wave (ds_swizzle, bit mode: lane ^ 16).
__device__ __forceinline__ double swap16(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
    return __hiloint2double(hi, lo);
}

constexpr int TEAM_THREADS = 512;
constexpr int TEAM_WAVES = TEAM_THREADS / WAVE;
constexpr int TEAM_ROWS = TEAM_THREADS / 2;

template <int CH>
struct TeamLds {
    __attribute__((aligned(16))) double vh[2][2][TEAM_ROWS / 2 + 8];   // pivot column by step parity and column parity
    __attribute__((aligned(16))) double qh[2][TEAM_ROWS / 2 + 8];      // p = A u / h by column parity
    double2 de[TEAM_ROWS];                                             // tridiagonal: (diagonal, squared off-diagonal)
    double red[2][TEAM_WAVES];
    double aii;
    double tail[3];
    int redi[TEAM_WAVES];
};

// mu_max of the complement Laplacian whose compacted rows (np x S words) are at `rows`; valid in every thread.
template <int CH, int S>
__device__ __forceinline__ double mu_max_team(const unsigned long long* rows, int np, TeamLds<CH>& L, int rounds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & (WAVE - 1);
    const int r = wave * 32 + ((lane >> 5) << 4) + (lane & 15), h = (lane >> 4) & 1;
    unsigned long long cw[S];
    int cdeg = 0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        cw[q] = (r < np) ? rows[(size_t)r * S + q] : 0ull;
        cdeg += __popcll(cw[q]);
    }
    int dmax = cdeg;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
    if (lane == 0) L.redi[wave] = dmax;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TEAM_WAVES; ++k) dmax = max(dmax, L.redi[k]);

    double a[CH];
    static_for<0, CH>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        const int c = 2 * k + h;
        const unsigned long long word = cw[(c >> 6) < S ? (c >> 6) : 0];
        const double off = ((c >> 6) < S) ? -(double)((word >> (c & 63)) & 1ull) : 0.0;
        a[k] = (r < np) ? ((c == r) ? (double)cdeg : off) : 0.0;
    });

    // Two workgroup barriers per step.  Barrier 1 publishes the pivot column v and the partial sums of |v|^2:
    // u = v except u_l = v_l - g, which every thread works out for itself and swaps into the chunks it reads
    // (TeamCols::fixed).  Barrier 2 publishes p = A u / H with the partial sums of K; the rank-2 update is
    // applied in the form  A -= u p^T + w u^T,  w = p - 2K u  (= u q^T + q u^T with q = p - K u), so that p
    // need not wait for K.  The update hands back the thread's part of the next pivot column.  v is double
    // buffered: a fast wave writes the next column while a slow one still reads this one.
#ifdef MARL_EXP_TIMING
    double tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    double mine = column_of<CH, 0, CH>(a, (np - 1) >> 1);
    for (int i = np - 1; i >= 2; --i) {
        const int l = i - 1, lh = l >> 1, buf = i & 1;
        TSTAMP(0, mine);
        const double other = swap16(mine);                          // the partner's value
        const double v = (h == (i & 1)) ? mine : other;             // A[r][i]
        const double vj = (r <= l) ? v : 0.0;
        const double hw = wave_sum_dpp(h == 0 ? vj * vj : 0.0);
        if (h == 0) {
            L.vh[buf][r & 1][r >> 1] = vj;
            if (r == i) L.aii = v;
        }
        if (lane == 0) L.red[0][wave] = hw;
        TSTAMP(1, mine);
        __syncthreads();
        TSTAMP(2, mine);
        const double f = L.vh[buf][l & 1][lh];
        double hsum = 0.0;
#pragma unroll
        for (int k = 0; k < TEAM_WAVES; ++k) hsum += L.red[0][k];
        if (tid == 0) L.de[i] = make_double2(L.aii, hsum);
        const double sq = (hsum > 0.0) ? fast_sqrt(hsum) : 0.0;
        const double g = (f >= 0.0) ? -sq : sq;
        const double h2 = hsum - f * g;
        const double hinv = (hsum > 0.0) ? fast_rcp(h2) : 0.0;
        const double ul = f - g;
        const double uj = (r == l) ? ul : vj;
        const int fq = (h == (l & 1)) ? lh - (lane & 15) : -1;      // == 16 q in the lane that holds u_l in chunk q
        double p0 = 0.0, p1 = 0.0;
        const double* u16 = L.vh[buf][h] + (lane & 15);
        const double u0 = u16[0];
        TSTAMP(3, p0);
        TeamCols<CH, 0>::matvec(a, u16, lh, fq, ul, p0, p1, u0, u0);
        TSTAMP(4, p0);
        const double ps = p0 + p1;
        const double pr = ps + swap16(ps);
        const double p = (r <= l) ? pr * hinv : 0.0;
        const double kw = wave_sum_dpp(h == 0 ? uj * p : 0.0);
        if (h == 0) L.qh[r & 1][r >> 1] = p;
        if (lane == 0) L.red[1][wave] = kw;
        __syncthreads();
        TSTAMP(5, mine);
        double ksum = 0.0;
#pragma unroll
        for (int k = 0; k < TEAM_WAVES; ++k) ksum += L.red[1][k];
        const double wj = fma(-(ksum * hinv), uj, p);                 // p - 2K u,  K = ksum hinv / 2
        {
            const double* p16 = L.qh[h] + (lane & 15);
            const double pq0 = p16[0];
            const double nuj = -uj, nwj = -wj;
            mine = TeamCols<CH, 0>::update(a, nuj, nwj, lh, fq, ul, u16, p16, u0, u0, pq0, pq0);
        }
        TSTAMP(6, mine);
    }
#ifdef MARL_EXP_TIMING
    {
        double r_ = 0.0;
        static_for<0, 8>([&](auto k) { if ((blockIdx.x & 7) == decltype(k)::value) r_ = tacc[decltype(k)::value]; });
        return r_;
    }
#endif
    // the remaining 2 x 2 block: A[1][1] is column 1 (parity 1, k = 0) of row 1, A[1][0] and A[0][0] are a[0]
    if (r == 1 && h == 1) L.tail[2] = a[0];
    if (r == 1 && h == 0) L.tail[1] = a[0];
    if (r == 0 && h == 0) L.tail[0] = a[0];
    __syncthreads();
    if (tid == 0) {
        L.de[1] = make_double2(L.tail[2], L.tail[1] * L.tail[1]);
        L.de[0] = make_double2(L.tail[0], 0.0);
    }
    __syncthreads();
    const double lo = (double)(dmax + 1) * (1.0 - 1e-13) - 1e-13;
    const double hi = (double)min(np, 2 * dmax) * (1.0 + 1e-13) + 1e-13;
    return sturm_largest_wave(L, np, lane, lo, hi, rounds + 1);     // every wave redundantly: no barrier inside
}

// Stage op: lambda2 for the envs the wave kernel left pending (pend[e] = np > 0).
template <int CH, int S>
__global__ __launch_bounds__(TEAM_THREADS) void lambda2_team_kernel(int n, const int* pend, const double* ws,
                                                                    size_t ws_stride, double* lambda2, int rounds,
                                                                    const int* team_list) {
    __shared__ TeamLds<CH> L;
    int e = blockIdx.x;
    if (team_list != nullptr) {                 // after the Lanczos launch: workgroup b takes entry b of what it left
        if (e >= min(team_list[0], TEAM_LIST_MAX)) return;
        e = team_list[1 + e];
    }
    const int np = pend[e];
    if (np <= 0) return;
    const double mu = mu_max_team<CH, S>(reinterpret_cast<const unsigned long long*>(ws + (size_t)e * ws_stride), np, L, rounds);
    if (threadIdx.x == 0) lambda2[e] = (double)n - mu;
}

// Second launch of a step for 64 < N <= 256: the pending eigenproblems, one workgroup each, then the
// reward of that env (term 1 and the clipped sum, as env_step_solve_kernel does for N <= 64).
template <int CH, int S>
__global__ __launch_bounds__(TEAM_THREADS) void env_step_team_kernel(MarlEnvParams P, MarlEnvState St, MarlStepIO io,
                                                                     int from_list) {
    __shared__ TeamLds<CH> L;
    int e = blockIdx.x;
    if (from_list) {                            // after the Lanczos launch: workgroup b takes entry b of what it left
        const int* team_list = workspace_team_list(St.workspace, P.N, St.E);
        if (e >= min(team_list[0], TEAM_LIST_MAX)) return;
        e = team_list[1 + e];
    }
    const int np = workspace_pending(St.workspace, P.N, St.E)[e];
    if (np <= 0) return;
    const size_t ld = (size_t)big_ld(P.N);
    const double mu = mu_max_team<CH, S>(reinterpret_cast<const unsigned long long*>(St.workspace + (size_t)e * ld * ld),
                                         np, L, P.lambda2_rounds);
    if (threadIdx.x == 0) {
        const double lam = (double)P.N - mu;
        const double* part = St.solve_partial + (size_t)e * 4;
        double r1;
        const double lam0 = St.lambda2_initial[e];
        const double total_r = reward_total(P, lam, lam0, part[0], part[1], (float)part[2], part[3], r1);
        finish_env(P, St, io, e, St.step_count[e], total_r, lam, lam0);
        if (io.reward_terms) io.reward_terms[(size_t)e * 5] = (float)r1;
    }
}

// Between the two launches of a step for 64 < N <= 256 (and of the stage op): the Lanczos iteration on the pending
// eigenproblems, one wave each, straight from the compacted bit rows.  A problem that meets its bound is finished here
// (pending count cleared: the team kernel's workgroup for it leaves at once); the others stay for the direct solver.
// LDS of the global-memory solver where the Lanczos kernels need it (they share one block of LDS)
template <int S>
struct GlobalSolveLds {
    double2 de[S * WAVE];
    double uvec[S * WAVE], qvec[S * WAVE];
};
template <int S>
constexpr size_t rows_lds_bytes() {
    return sizeof(LanczosRowsLds<S>) > sizeof(GlobalSolveLds<S>) ? sizeof(LanczosRowsLds<S>) : sizeof(GlobalSolveLds<S>);
}
// The Lanczos iteration on env e's pending eigenproblem, by the W waves of the workgroup.  True: mu holds mu_max.
// False: not within its bound, and the env is on the team solver's list now -- unless that was full, in which case
// the first wave has solved it by itself through global memory (true for that wave, the others leave).
template <int S, int W>
__device__ __forceinline__ bool rows_solve(double* region, int ld, int np, int tid, double nfull, unsigned char* raw, int rounds,
                                           int kmax, int e, int* team_list, double& mu) {
    const unsigned long long* rows = reinterpret_cast<const unsigned long long*>(region);
    LanczosRowsLds<S>& L = *reinterpret_cast<LanczosRowsLds<S>*>(raw);
    if (mu_max_lanczos_rows<S, W>(rows, np, tid, nfull, L, rounds, kmax, mu)) return true;
    const int lane = tid & (WAVE - 1);
    int at = TEAM_LIST_MAX;                  // (a problem beyond the team solver's size does not ask for a place)
    if (tid == 0 && np <= TEAM_MAX_NP) at = atomicAdd(&team_list[0], 1);
    if constexpr (W == 1) {
        at = __builtin_amdgcn_readfirstlane(at);
    } else {
        WSYNC();
        if (tid == 0) L.red[0][0] = (double)at;
        WSYNC();
        at = (int)L.red[0][0];
    }
    if (at < TEAM_LIST_MAX && np <= TEAM_MAX_NP) {
        if (tid == 0) team_list[1 + at] = e;
        return false;
    }
    if (tid >= WAVE) return false;           // the solve in place is one wave's
    uint64_t w[S][S];                        // the rows lie where the matrix goes: into registers first
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int q = 0; q < S; ++q) w[s][q] = (s * WAVE + lane < np) ? rows[(size_t)(s * WAVE + lane) * S + q] : 0ull;
    WSYNC();
    mu = mu_max_global_wave<S>(w, np, lane, *reinterpret_cast<GlobalSolveLds<S>*>(raw), rounds, region, ld);
    return true;
}

// Between the two launches of a step for 64 < N <= 256 (and of the stage op): the Lanczos iteration on the pending
// eigenproblems, straight from the compacted bit rows, W waves per problem.  A problem that meets its bound is finished
// here (pending count cleared); the others go on the list that the team launch works through.
//
// The launch is as long as its slowest solve started late, and the slow ones are the large ones (more vertices, more
// steps).  The grid therefore passes over the envs ROWS_PASSES times -- workgroup b looks at env b mod E and takes it
// only in the pass of its size class, largest first -- so that every large problem is running before the first small
// one starts; a workgroup that only looks costs a few nanoseconds.
constexpr int ROWS_PASSES = 3;
__device__ __forceinline__ int rows_pass_of(int np, int n) { return 4 * np >= 3 * n ? 0 : (2 * np >= n ? 1 : 2); }
template <int S, int W>
__global__ __launch_bounds__(WAVE * W) void env_step_lanczos_rows_kernel(MarlEnvParams P, MarlEnvState St, MarlStepIO io) {
    __shared__ __align__(16) unsigned char raw[rows_lds_bytes<S>()];
    const int e = blockIdx.x % St.E, tid = threadIdx.x;
    int* pend = workspace_pending(St.workspace, P.N, St.E);
    const int np = pend[e];
    if (np <= 0 || rows_pass_of(np, P.N) != (int)blockIdx.x / St.E) return;
    const int ld = big_ld(P.N);
    double mu;
    if (!rows_solve<S, W>(St.workspace + (size_t)e * ld * ld, ld, np, tid, (double)P.N, raw, P.lambda2_rounds, LANCZOS_ROWS_KMAX, e,
                          workspace_team_list(St.workspace, P.N, St.E), mu))
        return;
    if (tid == 0) {
        const double lam = (double)P.N - mu;
        const double* part = St.solve_partial + (size_t)e * 4;
        double r1;
        const double lam0 = St.lambda2_initial[e];
        const double total_r = reward_total(P, lam, lam0, part[0], part[1], (float)part[2], part[3], r1);
        finish_env(P, St, io, e, St.step_count[e], total_r, lam, lam0);
        if (io.reward_terms) io.reward_terms[(size_t)e * 5] = (float)r1;
        pend[e] = 0;
    }
}
template <int S, int W>
__global__ __launch_bounds__(WAVE * W) void lambda2_lanczos_rows_kernel(int n, int E, int* pend, double* ws, int ld, int* team_list,
                                                                        double* lambda2, int rounds, int kmax) {
    __shared__ __align__(16) unsigned char raw[rows_lds_bytes<S>()];
    const int e = blockIdx.x % E, tid = threadIdx.x;
    const int np = pend[e];
    if (np <= 0 || rows_pass_of(np, n) != (int)blockIdx.x / E) return;
    double mu;
    if (!rows_solve<S, W>(ws + (size_t)e * ld * ld, ld, np, tid, (double)n, raw, rounds, kmax, e, team_list, mu)) return;
    if (tid == 0) {
        lambda2[e] = (double)n - mu;
        pend[e] = 0;
    }
}

template <int S>
__global__ __launch_bounds__(WAVE) void lambda2_bits_big_kernel(int n, const uint32_t* bits, double* lambda2, int rounds,
                                                                double* ws, int* pend, int rows_follow) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x;
    const int W = (n + 31) / 32;
    uint64_t rows[S][S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            uint64_t w = 0;
            if (i < n) {
                const uint32_t* p = bits + ((size_t)e * n + i) * W;
                if (2 * k < W) w = p[2 * k];
                if (2 * k + 1 < W) w |= (uint64_t)p[2 * k + 1] << 32;
                w &= word_mask<S>(n, k);
                if (k == s) w &= ~(1ull << lane);
            }
            rows[s][k] = w;
        }
    }
    const int ld = big_ld(n);
    int pending;
    const double lam = lambda2_big<S>(rows, n, lane, L, rounds, ws + (size_t)e * ld * ld, &pending, rows_follow != 0);
    if (lane == 0) {
        pend[e] = pending;
        if (!pending) lambda2[e] = lam;
        if (e == 0) pend[((size_t)gridDim.x + 63) & ~(size_t)63] = 0;        // the team solver's list (workspace_team_list): empty
    }
}

template <int S>
__global__ __launch_bounds__(WAVE) void dbscan_big_kernel(MarlEnvParams P, const double* x, const double* y,
                                                          int32_t* labels, double* cx, double* cy, int32_t* csz,
                                                          int32_t* n_clusters, int32_t* ncent_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N, KS = n + 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int i = s * WAVE + lane;
        if (i < n) { L.sx[i] = x[(size_t)e * n + i]; L.sy[i] = y[(size_t)e * n + i]; }
    }
    WSYNC();
    int num, ncent, label[S];
    dbscan_big<S>(P, n, lane, L, label, num, ncent);
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (s * WAVE + lane < n) labels[(size_t)e * n + s * WAVE + lane] = label[s];
    for (int s = lane; s < ncent; s += WAVE) {
        cx[(size_t)e * KS + s] = L.cx[s];
        cy[(size_t)e * KS + s] = L.cy[s];
        csz[(size_t)e * KS + s] = L.csz[s];
    }
    if (lane == 0) { n_clusters[e] = num; ncent_out[e] = ncent; }
}

template <int S>
__global__ __launch_bounds__(WAVE) void reward_obs_big_kernel(MarlEnvParams P, MarlRewardObsIO io) {
    extern __shared__ __align__(16) unsigned char smem[];
    BigLds<S>& L = *reinterpret_cast<BigLds<S>*>(smem);
    const int e = blockIdx.x, lane = threadIdx.x, n = P.N;
    int eband = io.enemy_band[e];
    eband = eband < 0 ? 0 : (eband > 3 ? 3 : eband);
    int ncent = io.ncent[e];
    ncent = ncent < 0 ? 0 : (ncent > n + 1 ? n + 1 : ncent);
    const bool power = io.avg_power_dbm && io.enemy_x && io.enemy_y;
    if (power) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int i = s * WAVE + lane;
            if (i < n) { L.sx[i] = io.enemy_x[(size_t)e * n + i]; L.sy[i] = io.enemy_y[(size_t)e * n + i]; }
        }
    }
    reward_obs_stage(P, io, L, e, lane, io.n_clusters[e], ncent, eband);
    if (power) {
        const double avgp = avg_power_big<S>(P, n, P.M, lane, L, eband);
        if (lane == 0) io.avg_power_dbm[e] = avgp;
    }
}

// ------------------------------------------------------------------ host side
int hip_check(hipError_t err) {
    if (err == hipSuccess) return MARL_OK;
    g_last_hip_error = (int)err;
    return MARL_EHIP;
}

// Reference chain of received_power_watts (fspl.py:199-206) applied to a squared distance.
bool power_reaches(double d2, double P_w, double wavelength, double threshold_w) {
    double d = sqrt(d2);
    if (!(d > 1e-10)) d = 1e-10;                       // np.maximum(distance, eps)
    volatile double t = kFourPi * d;
    volatile double g = wavelength / t;
    volatile double pg = g * g;
    volatile double pr = P_w * pg;
    return pr >= threshold_w;
}

// Largest double d2 >= 0 with power_reaches(d2); -1 when even d2 = 0 fails.  The chain
// is monotone non-increasing in d2 (every operation is correctly rounded), so
// bisection over the ordered bit patterns of non-negative doubles finds it exactly.
double largest_reaching_d2(double P_w, double wavelength, double threshold_w) {
    if (!power_reaches(0.0, P_w, wavelength, threshold_w)) return -1.0;
    uint64_t lo = 0;                                    // reaches
    uint64_t hi = 0x7FEFFFFFFFFFFFFFull;                // DBL_MAX
    double dhi;
    memcpy(&dhi, &hi, 8);
    if (power_reaches(dhi, P_w, wavelength, threshold_w)) return dhi;
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        double dm;
        memcpy(&dm, &mid, 8);
        if (power_reaches(dm, P_w, wavelength, threshold_w)) lo = mid; else hi = mid;
    }
    double out;
    memcpy(&out, &lo, 8);
    return out;
}

// Device address of `scratch` on the current device, looked up on the first call per device.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {};
int static_scratch(void** out) {
    int dev = -1;
    int rc = hip_check(hipGetDevice(&dev));
    if (rc != MARL_OK) return rc;
    if (dev < 0 || dev >= kMaxDevices) return MARL_EUNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        rc = hip_check(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        if (rc != MARL_OK) return rc;
        if (ptr == nullptr) return MARL_EHIP;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MARL_OK;
}

// A `rounds` value: multisection rounds in the low byte (0 = the default, 8) plus at most one solver selector.
int check_rounds(int32_t* rounds) {
    const int32_t sel = *rounds & (MARL_LAMBDA2_DIRECT | MARL_LAMBDA2_LANCZOS);
    if (*rounds < 0 || (*rounds & ~(0xFF | MARL_LAMBDA2_DIRECT | MARL_LAMBDA2_LANCZOS | MARL_LAMBDA2_TEST_FEW_STEPS)) ||
        sel == (MARL_LAMBDA2_DIRECT | MARL_LAMBDA2_LANCZOS))
        return MARL_EINVAL;
    if ((*rounds & 0xFF) == 0) *rounds |= 8;
    return MARL_OK;
}

int check_sizes(const MarlEnvParams* p) {
    if (p == nullptr) return MARL_EINVAL;
    if (p->N < 1 || p->M < 1) return MARL_EINVAL;
    if (p->N > MARL_MAX_N || p->M > MARL_MAX_M) return MARL_EUNSUPPORTED;
    return MARL_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// Global scratch of the lambda2 solver for N > 64: one ld x ld float64 matrix per env.
size_t workspace_bytes(int N, int E) {
    if (N <= WAVE) return 0;
    const size_t ld = (size_t)big_ld(N);
    return (size_t)E * ld * ld * sizeof(double) + align256((size_t)E * sizeof(int)) +   // + pending list of the team solver
           align256((size_t)(TEAM_LIST_MAX + 1) * sizeof(int));                           // + what the Lanczos launch leaves to it
}


// Team solver launch: 104 columns per thread beyond 128 vertices, 64 up to there.
#if defined(MARL_DEV_FAST) && !defined(MARL_DEV_BIG)      /* development builds leave the kernels for N > 64 out (MARL_DEV_BIG: in) */
#define MARL_LAUNCH_TEAM(N, kern, E, stream, ...) do { } while (0)
#define MARL_DISPATCH_SLOTS(N, kern, grid, stream, ...) do { } while (0)
#define MARL_DISPATCH_ROWS(N, kern, grid, stream, ...) do { } while (0)
#else
#define MARL_LAUNCH_TEAM(N, kern, E, stream, ...)                                                                   \
    do {                                                                                                            \
        const int s_ = ((N) + WAVE - 1) / WAVE;                                                                     \
        if (s_ == 2) hipLaunchKernelGGL((kern<64, 2>), dim3(E), dim3(TEAM_THREADS), 0, (hipStream_t)(stream), __VA_ARGS__);      \
        else if (s_ == 3) hipLaunchKernelGGL((kern<104, 3>), dim3(E), dim3(TEAM_THREADS), 0, (hipStream_t)(stream), __VA_ARGS__); \
        else hipLaunchKernelGGL((kern<104, 4>), dim3(E), dim3(TEAM_THREADS), 0, (hipStream_t)(stream), __VA_ARGS__);             \
    } while (0)

#define MARL_DISPATCH_SLOTS(N, kern, grid, stream, ...)                                                            \
    switch (((N) + WAVE - 1) / WAVE) {                                                                             \
        case 2: hipLaunchKernelGGL((kern<2>), grid, dim3(WAVE), sizeof(BigLds<2>), (hipStream_t)(stream), __VA_ARGS__); break; \
        case 3: hipLaunchKernelGGL((kern<3>), grid, dim3(WAVE), sizeof(BigLds<3>), (hipStream_t)(stream), __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((kern<4>), grid, dim3(WAVE), sizeof(BigLds<4>), (hipStream_t)(stream), __VA_ARGS__); break; \
    }
// the same for kernels whose LDS is static (the Lanczos iteration on bit rows)
#ifndef ROWS_WAVES_4
#define ROWS_WAVES_2 1      /* waves per graph in the rows launch, by row slots (see mu_max_lanczos_rows) */
#define ROWS_WAVES_3 1
#define ROWS_WAVES_4 1
#endif
#define MARL_DISPATCH_ROWS(N, kern, grid, stream, ...)                                                             \
    switch (((N) + WAVE - 1) / WAVE) {                                                                             \
        case 2: hipLaunchKernelGGL((kern<2, ROWS_WAVES_2>), grid, dim3(WAVE * ROWS_WAVES_2), 0, (hipStream_t)(stream), __VA_ARGS__); break;  \
        case 3: hipLaunchKernelGGL((kern<3, ROWS_WAVES_3>), grid, dim3(WAVE * ROWS_WAVES_3), 0, (hipStream_t)(stream), __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL((kern<4, ROWS_WAVES_4>), grid, dim3(WAVE * ROWS_WAVES_4), 0, (hipStream_t)(stream), __VA_ARGS__); break; \
    }
#endif

// The wave kernels are instantiated per register-matrix width NPAD = N rounded up to 8, and for 50 (the
// reference's standard swarm: 12 registers fewer than 56 make room for a fourth wave per SIMD).
#define MARL_LAUNCH_NPAD(NP, kern, grid, lds, stream, ...) \
    hipLaunchKernelGGL((kern<NP>), grid, dim3(WAVE), lds, (hipStream_t)(stream), __VA_ARGS__)
#ifdef MARL_DEV_FAST      /* development builds: the 50-column kernels only (compiles in a fraction of the time) */
#define MARL_DISPATCH_NPAD(N, kern, grid, lds, stream, ...) MARL_LAUNCH_NPAD(50, kern, grid, lds, stream, __VA_ARGS__)
#else
#define MARL_DISPATCH_NPAD(N, kern, grid, lds, stream, ...)                                   \
    if ((N) > 48 && (N) <= 50) {                                                               \
        MARL_LAUNCH_NPAD(50, kern, grid, lds, stream, __VA_ARGS__);                            \
    } else switch (((N) + 7) / 8) {                                                            \
        case 1: MARL_LAUNCH_NPAD(8, kern, grid, lds, stream, __VA_ARGS__); break;              \
        case 2: MARL_LAUNCH_NPAD(16, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 3: MARL_LAUNCH_NPAD(24, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 4: MARL_LAUNCH_NPAD(32, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 5: MARL_LAUNCH_NPAD(40, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 6: MARL_LAUNCH_NPAD(48, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 7: MARL_LAUNCH_NPAD(56, kern, grid, lds, stream, __VA_ARGS__); break;             \
        default: MARL_LAUNCH_NPAD(64, kern, grid, lds, stream, __VA_ARGS__); break;            \
    }
#endif

// The kernels with the Lanczos iteration exist for swarms of more than 32 enemies only (smaller eigenproblems all go to
// the two-per-wave direct solver).
#ifdef MARL_DEV_FAST
#define MARL_DISPATCH_NPAD_BIG(N, kern, grid, lds, stream, ...) MARL_LAUNCH_NPAD(50, kern, grid, lds, stream, __VA_ARGS__)
#else
#define MARL_DISPATCH_NPAD_BIG(N, kern, grid, lds, stream, ...)                               \
    if ((N) > 48 && (N) <= 50) {                                                               \
        MARL_LAUNCH_NPAD(50, kern, grid, lds, stream, __VA_ARGS__);                            \
    } else switch (((N) + 7) / 8) {                                                            \
        case 5: MARL_LAUNCH_NPAD(40, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 6: MARL_LAUNCH_NPAD(48, kern, grid, lds, stream, __VA_ARGS__); break;             \
        case 7: MARL_LAUNCH_NPAD(56, kern, grid, lds, stream, __VA_ARGS__); break;             \
        default: MARL_LAUNCH_NPAD(64, kern, grid, lds, stream, __VA_ARGS__); break;            \
    }
#endif

// The Lanczos iteration halves the instructions of a large eigen-solve on average, at three waves per SIMD instead of four
// and with a spread of 1 : 2 between the fastest and the slowest solve of one size that the launch order cannot foresee.
// A launch that the chip holds in about one round of waves is as long as its slowest solve, and the direct solver's are all
// alike: 8 192 envs x (50, 8) on 256 CUs take 0.110 ms per step with it and 0.118 ms with the iteration (4 096 envs: 0.077
// against 0.090).  Batches of several rounds are bound by the sum instead, and there the iteration was ahead (16 384 envs
// 0.191 against 0.167 ms) until the problems were reduced before the solve (reduce_problem): what is left of them suits
// the direct solver better at every batch size measured (12 288 envs 9.7 against 8.7 x 10^7 env-steps/s, 16 384: 10.8
// against 9.7, 65 536: 13.1 against 12.9).  AUTO therefore takes the direct solver; the iteration stays selectable.

// Priorities in the solve launch pay when the launch is about one round of the chip's wave slots (16 per CU; about 0.6
// solve blocks per env): every slot then starts a solve at once and nothing follows, so what counts is that the four
// solves of a SIMD end together.  With fewer blocks the SIMDs are not contended, with more the blocks that end early make
// room for the next ones and holding them back costs -- which is the test the order kernel makes on the number of the
// launch's items (SOLVE_PRIO_ITEMS_MIN / MAX).  On 256 CUs, one launch sequence, us per step with / without:
// 4 096 envs 76.9 / 74.6, 6 144 envs 88.6 / 88.0, 7 168 envs 100.5 / 103.7, 8 192 envs 106.2 / 111.3, 9 216 envs
// 118.9 / 117.8, 10 240 envs 133.0 / 127.5 (tools/prio_sweep.sh).
enum { SOLVE_WAVES_PER_CU = 16 };          // the direct solve kernel: four waves on each of a compute unit's four SIMDs

// Compute units of the current device (cached per device; 0 on error).
int cu_count() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) return 0;
        __atomic_store_n(&cached[dev], c, __ATOMIC_RELAXED);
    }
    return c;
}

}  // namespace

// ====================================================================== C ABI
// for the other translation units of the library: what marl_last_hip_error reports
extern "C" __attribute__((visibility("hidden"))) void marl_note_hip_error(int err) { g_last_hip_error = err; }
// marl_lambda2_large.hip: the solver for graphs beyond MARL_MAX_N vertices (returns a hipError_t)
extern "C" __attribute__((visibility("hidden"))) int marl_lambda2_large_ld(int n);
extern "C" __attribute__((visibility("hidden"))) int marl_lambda2_large_launch(int n, int E, const uint32_t* bits, double* lambda2,
                                                                              double* workspace, void* stream);

extern "C" {

const char* marl_strerror(int code) {
    switch (code) {
        case MARL_OK: return "ok";
        case MARL_EINVAL: return "invalid argument";
        case MARL_EUNSUPPORTED: return "size not supported by the kernels (N <= 256, M <= 64)";
        case MARL_EHIP: return "HIP runtime error";
        case MARL_ETOOBIG: return "request needs more scratch than the static buffer holds (MARL_STATIC_SCRATCH_BYTES)";
        default: return "unknown error";
    }
}

int marl_last_hip_error(void) { return g_last_hip_error; }

int marl_params_init(MarlEnvParams* p) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (!(p->arena_size > 0) || !(p->dt > 0) || p->K_recompute < 1 || p->max_steps < 1) return MARL_EINVAL;
    if (p->enemy_mode != MARL_ENEMY_RANDOM_WALK && p->enemy_mode != MARL_ENEMY_COORDINATED_TURN) return MARL_EINVAL;
    if (!(p->omega_max > 0)) p->omega_max = 0.5;             // EnemySwarm default (enemy_swarm.py:60)
    if (p->step_schedule < MARL_SCHEDULE_AUTO || p->step_schedule > MARL_SCHEDULE_SPLIT) return MARL_EINVAL;
    for (int b = 0; b < MARL_NUM_BANDS; ++b) {
        const double wl = kSpeedOfLight / band_hz(b);
        p->comm_d2_max[b] = largest_reaching_d2(p->P_tx_w, wl, p->P_sens_w);
        p->jam_d2_max[b] = largest_reaching_d2(p->P_jam_w, wl, p->P_thr_w);
        volatile double ratio = p->P_jam_w / p->P_thr_w;     // compute_jam_range (fspl.py:306-308)
        volatile double a = wl / kFourPi;
        p->R_jam[b] = a * sqrt(ratio);
    }
    // Float32 thresholds of the step's first-stage jamming test, around 4 d2max (the doubled-distance form).
    // With coordinates in [0, A]: the float32 inputs are off by at most 2^-24 A each, the doubled difference
    // DX = 2q - (xi + xj) by delta = 8 2^-24 A, so d2 = DX^2 + DY^2 by at most
    // 2 sqrt(2 d2) delta + 2 delta^2 + 4 2^-24 d2; the band is twice that, and the thresholds are moved
    // outwards once more when they are rounded to float.
    for (int b = 0; b < MARL_NUM_BANDS; ++b) {
        const double t4 = 4.0 * p->jam_d2_max[b];
        if (!(p->jam_d2_max[b] >= 0.0)) { p->jam_lo_f32[b] = -1.0f; p->jam_hi_f32[b] = -1.0f; continue; }
        const double u = 5.9604644775390625e-08;         // 2^-24
        const double delta = 8.0 * u * p->arena_size;
        const double g = 2.0 * (2.0 * sqrt(2.0 * t4) * delta + 2.0 * delta * delta + 4.0 * u * t4);
        const double lo = t4 - g, hi = t4 + g;
        p->jam_lo_f32[b] = lo > 0.0 ? nextafterf((float)lo, -INFINITY) : -1.0f;
        p->jam_hi_f32[b] = hi < 3.0e38 ? nextafterf((float)hi, INFINITY) : 3.4e38f;
    }
    p->dt_f32 = (float)p->dt;
    p->v_max_f32 = (float)p->v_max;
    p->v_max_sq_f32 = (float)(p->v_max * p->v_max);
    if (p->stage_flags & ~(MARL_STAGE_NO_RECLUSTER | MARL_STAGE_NO_SOLVE_PRIORITY | MARL_STAGE_GENERAL_RESET)) return MARL_EINVAL;
    if (check_rounds(&p->lambda2_rounds) != MARL_OK) return MARL_EINVAL;
    return MARL_OK;
}

size_t marl_state_bytes(const MarlEnvParams* p, int32_t E) {
    if (check_sizes(p) != MARL_OK || E < 1) return 0;
    const size_t n = p->N, m = p->M, ks = n + 1, e = (size_t)E;
    size_t b = 0;
    b += 5 * align256(e * n * 8);        // enemy x, y, vx, vy, omega
    b += 2 * align256(e * m * 8);        // jam x, y
    b += align256(e * m * 4);            // jam_band
    b += 2 * align256(e * m * 4);        // last_vx, last_vy
    b += 3 * align256(e * 4);            // step_count, enemy_band, tick
    b += align256(e * 8);                // lambda2_initial
    b += align256(e * n * 4);            // labels
    b += 2 * align256(e * ks * 8);       // cent x, y
    b += align256(e * ks * 4);           // cent_size
    b += 2 * align256(e * 4);            // n_clusters, ncent
    b += 2 * align256(e * ks * 8);       // prev x, y
    b += 2 * align256(e * 4);            // prev_n_clusters, prev_ncent
    b += align256(e * m * 4);            // assign
    b += align256(workspace_bytes(p->N, E)); // lambda2 matrix workspace (N > 64 only)
    if (p->N <= WAVE) {                      // hand-over from the front kernel to the solve kernel
        b += align256(e * WAVE * 8);         // solve_rows
        b += align256(e * WAVE * 2);         // solve_meta
        b += 3 * align256(e * 4);            // solve_np, solve_slot, solve_perm
        b += align256(SOLVE_HIST_WORDS * 4); // solve_hist
    }
    b += align256(e * 4 * 8);                // solve_partial
    b += align256(e * 8);                    // episode_return
    return b;
}

int marl_state_bind(MarlEnvState* s, const MarlEnvParams* p, int32_t E, void* base) {
    if (s == nullptr || base == nullptr || E < 1) return MARL_EINVAL;
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (((uintptr_t)base & 255) != 0) return MARL_EINVAL;
    const size_t n = p->N, m = p->M, ks = n + 1, e = (size_t)E;
    unsigned char* c = (unsigned char*)base;
    auto take = [&](size_t bytes) { void* r = c; c += align256(bytes); return r; };
    s->E = E; s->pad = 0;
    s->enemy_x = (double*)take(e * n * 8); s->enemy_y = (double*)take(e * n * 8);
    s->enemy_vx = (double*)take(e * n * 8); s->enemy_vy = (double*)take(e * n * 8);
    s->enemy_omega = (double*)take(e * n * 8);
    s->jam_x = (double*)take(e * m * 8); s->jam_y = (double*)take(e * m * 8);
    s->jam_band = (int32_t*)take(e * m * 4);
    s->last_vx = (float*)take(e * m * 4); s->last_vy = (float*)take(e * m * 4);
    s->step_count = (int32_t*)take(e * 4); s->enemy_band = (int32_t*)take(e * 4);
    s->tick = (int32_t*)take(e * 4);
    s->lambda2_initial = (double*)take(e * 8);
    s->labels = (int32_t*)take(e * n * 4);
    s->cent_x = (double*)take(e * ks * 8); s->cent_y = (double*)take(e * ks * 8);
    s->cent_size = (int32_t*)take(e * ks * 4);
    s->n_clusters = (int32_t*)take(e * 4); s->ncent = (int32_t*)take(e * 4);
    s->prev_x = (double*)take(e * ks * 8); s->prev_y = (double*)take(e * ks * 8);
    s->prev_n_clusters = (int32_t*)take(e * 4); s->prev_ncent = (int32_t*)take(e * 4);
    s->assign = (int32_t*)take(e * m * 4);
    const size_t wsb = workspace_bytes(p->N, E);
    s->workspace = wsb ? (double*)take(wsb) : nullptr;
    const bool split = p->N <= WAVE;
    s->solve_rows = split ? (uint64_t*)take(e * WAVE * 8) : nullptr;
    s->solve_meta = split ? (uint16_t*)take(e * WAVE * 2) : nullptr;
    s->solve_np = split ? (int32_t*)take(e * 4) : nullptr;
    s->solve_slot = split ? (int32_t*)take(e * 4) : nullptr;
    s->solve_perm = split ? (int32_t*)take(e * 4) : nullptr;
    s->solve_hist = split ? (int32_t*)take(SOLVE_HIST_WORDS * 4) : nullptr;
    s->solve_partial = (double*)take(e * 4 * 8);
    s->episode_return = (double*)take(e * 8);
    return MARL_OK;
}

size_t marl_shard_scratch_bytes(void) { return align256(SOLVE_HIST_WORDS * 4); }

int marl_state_shard(const MarlEnvParams* p, const MarlEnvState* whole, int32_t first, int32_t count, void* scratch,
                     MarlEnvState* out) {
    if (p == nullptr || whole == nullptr || out == nullptr || scratch == nullptr) return MARL_EINVAL;
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (p->N > WAVE) return MARL_EUNSUPPORTED;
    if (first < 0 || count < 1 || (long long)first + count > whole->E || ((uintptr_t)scratch & 255) != 0) return MARL_EINVAL;
    const size_t n = p->N, m = p->M, ks = n + 1, f = (size_t)first;
    *out = *whole;
    out->E = count;
    auto move = [&](auto*& ptr, size_t row) { if (ptr != nullptr) ptr += f * row; };
    move(out->enemy_x, n); move(out->enemy_y, n); move(out->enemy_vx, n); move(out->enemy_vy, n); move(out->enemy_omega, n);
    move(out->jam_x, m); move(out->jam_y, m); move(out->jam_band, m); move(out->last_vx, m); move(out->last_vy, m);
    move(out->step_count, 1); move(out->enemy_band, 1); move(out->tick, 1); move(out->lambda2_initial, 1);
    move(out->labels, n); move(out->cent_x, ks); move(out->cent_y, ks); move(out->cent_size, ks);
    move(out->n_clusters, 1); move(out->ncent, 1); move(out->prev_x, ks); move(out->prev_y, ks);
    move(out->prev_n_clusters, 1); move(out->prev_ncent, 1); move(out->assign, m);
    move(out->solve_rows, WAVE); move(out->solve_meta, WAVE); move(out->solve_np, 1); move(out->solve_slot, 1); move(out->solve_perm, 1);
    move(out->solve_partial, 4); move(out->episode_return, 1);
    out->solve_hist = (int32_t*)scratch;
    return MARL_OK;
}

// The step's launches; with events, `before` and `after` are recorded on the stream around the launch that holds the
// eigen-solves (the only launch of the one-launch schedule).
static int step_launches(const MarlEnvParams* p, const MarlEnvState* s, const MarlStepIO* io, void* stream,
                         hipEvent_t before, hipEvent_t after) {
    if (p == nullptr || s == nullptr || io == nullptr) return MARL_EINVAL;
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (s->E < 1 || io->actions == nullptr || io->reward == nullptr ||
        io->truncated == nullptr || s->enemy_x == nullptr || s->episode_return == nullptr)
        return MARL_EINVAL;
    const int solver = p->lambda2_rounds & (MARL_LAMBDA2_DIRECT | MARL_LAMBDA2_LANCZOS);
    MarlEnvParams q = *p;
    q.lambda2_rounds &= 0xFF;                  // the kernels take the plain number of rounds
    p = &q;
    if (p->N > WAVE) {
        if (s->workspace == nullptr || s->solve_partial == nullptr) return MARL_EINVAL;
        if (solver != MARL_LAMBDA2_DIRECT) q.stage_flags |= STAGE_ROWS_FOLLOW;
        MARL_DISPATCH_SLOTS(p->N, env_step_big_kernel, dim3(s->E), stream, *p, *s, *io);
        if (before != nullptr && (rc = hip_check(hipEventRecord(before, (hipStream_t)stream))) != MARL_OK) return rc;
        // the pending eigenproblems (65 .. 208 vertices): Lanczos iteration on the bit rows, one wave each; the team
        // solver then takes what did not meet its bound (and everything, when the direct solver is asked for)
        if (solver != MARL_LAMBDA2_DIRECT) {
            MARL_DISPATCH_ROWS(p->N, env_step_lanczos_rows_kernel, dim3(s->E * ROWS_PASSES), stream, *p, *s, *io);
            MARL_LAUNCH_TEAM(p->N, env_step_team_kernel, (s->E < TEAM_LIST_MAX ? s->E : TEAM_LIST_MAX), stream, *p, *s, *io, 1);
        } else {
            MARL_LAUNCH_TEAM(p->N, env_step_team_kernel, s->E, stream, *p, *s, *io, 0);
        }
        if (after != nullptr && (rc = hip_check(hipEventRecord(after, (hipStream_t)stream))) != MARL_OK) return rc;
        return hip_check(hipGetLastError());
    }
    if (s->solve_rows == nullptr || s->solve_meta == nullptr || s->solve_np == nullptr || s->solve_slot == nullptr || s->solve_perm == nullptr ||
        s->solve_hist == nullptr || s->solve_partial == nullptr)
        return MARL_EINVAL;
    const bool fused = p->step_schedule == MARL_SCHEDULE_FUSED ||
                       (p->step_schedule == MARL_SCHEDULE_AUTO && (p->N <= 16 || s->E < 1024));
    if (fused) {
        if (before != nullptr && (rc = hip_check(hipEventRecord(before, (hipStream_t)stream))) != MARL_OK) return rc;
        MARL_DISPATCH_NPAD(p->N, env_step_front_kernel, dim3(s->E), wave_lds_bytes(p->N), stream, *p, *s, *io);
        if (after != nullptr && (rc = hip_check(hipEventRecord(after, (hipStream_t)stream))) != MARL_OK) return rc;
        return hip_check(hipGetLastError());
    }
    hipLaunchKernelGGL(env_step_front_kernel<0>, dim3(s->E), dim3(WAVE), front_lds_bytes(), (hipStream_t)stream, *p, *s, *io);
    const int all_cus = cu_count();
    const bool lanczos = p->N > 32 && solver == MARL_LAMBDA2_LANCZOS;
    const bool prio = !lanczos && all_cus > 0 && !(p->stage_flags & MARL_STAGE_NO_SOLVE_PRIORITY);
    hipLaunchKernelGGL(solve_order_kernel, dim3((s->E + 1023) / 1024), dim3(1024), 0, (hipStream_t)stream, (int)s->E, *s,
                       prio ? all_cus * SOLVE_WAVES_PER_CU : 0);
    if (before != nullptr && (rc = hip_check(hipEventRecord(before, (hipStream_t)stream))) != MARL_OK) return rc;
    // one block more than envs: the last one never has a solve and clears the histogram
    if (lanczos) {
        MARL_DISPATCH_NPAD_BIG(p->N, env_step_solve_lanczos_kernel, dim3(s->E + 1), 0, stream, *p, *s, *io);
    } else {
        MARL_DISPATCH_NPAD(p->N, env_step_solve_kernel, dim3(s->E + 1), 0, stream, *p, *s, *io);
    }
    if (after != nullptr && (rc = hip_check(hipEventRecord(after, (hipStream_t)stream))) != MARL_OK) return rc;
    return hip_check(hipGetLastError());
}

int marl_env_step(const MarlEnvParams* p, const MarlEnvState* s, const MarlStepIO* io, void* stream) {
    return step_launches(p, s, io, stream, nullptr, nullptr);
}

int marl_env_step_timed(const MarlEnvParams* p, const MarlEnvState* s, const MarlStepIO* io, void* stream,
                        void* solve_begin, void* solve_end) {
    if (solve_begin == nullptr || solve_end == nullptr) return MARL_EINVAL;
    return step_launches(p, s, io, stream, (hipEvent_t)solve_begin, (hipEvent_t)solve_end);
}

int marl_env_reset(const MarlEnvParams* p, const MarlEnvState* s, const MarlResetIO* io, void* stream) {
    if (p == nullptr || s == nullptr || io == nullptr) return MARL_EINVAL;
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (s->E < 1 || s->enemy_x == nullptr) return MARL_EINVAL;
    if ((io->enemy_pos == nullptr) != (io->enemy_vel == nullptr)) return MARL_EINVAL;
    MarlEnvParams q = *p;
    q.lambda2_rounds &= 0xFF;                  // the reset's lambda2 (of the unjammed graph) is the direct solver's
    p = &q;
    if (p->N > WAVE) {
        if (s->workspace == nullptr) return MARL_EINVAL;
        MARL_DISPATCH_SLOTS(p->N, env_reset_big_kernel, dim3(s->E), stream, *p, *s, *io);
        return hip_check(hipGetLastError());
    }
    bool complete = io->enemy_pos == nullptr && p->arena_size > 0.0 && !(p->stage_flags & MARL_STAGE_GENERAL_RESET);
    for (int b = 0; b < 4; ++b) complete = complete && p->comm_d2_max[b] >= 2.000001 * p->arena_size * p->arena_size;
    if (complete) {
        hipLaunchKernelGGL((env_reset_wave_kernel<0, true>), dim3(s->E), dim3(WAVE), front_lds_bytes(), (hipStream_t)stream, *p, *s, *io);
        return hip_check(hipGetLastError());
    }
    MARL_DISPATCH_NPAD(p->N, env_reset_wave_kernel, dim3(s->E), wave_lds_bytes(p->N), stream, *p, *s, *io);
    return hip_check(hipGetLastError());
}

static int part_bounds(const MarlEnvParams* p, const MarlEnvState* s, int32_t parts, void* const* streams, void* scratch) {
    if (p == nullptr || s == nullptr || streams == nullptr || scratch == nullptr) return MARL_EINVAL;
    if (parts < 1 || parts > MARL_MAX_PARTS || parts > s->E) return MARL_EINVAL;
    return MARL_OK;
}
#define rows_from(base, first, row) ((base) == nullptr ? nullptr : (base) + (first) * (row))

int marl_env_step_parts(const MarlEnvParams* p, const MarlEnvState* s, const MarlStepIO* io, int32_t parts,
                        void* const* streams, void* scratch) {
    int rc = part_bounds(p, s, parts, streams, scratch);
    if (rc != MARL_OK || io == nullptr) return rc != MARL_OK ? rc : MARL_EINVAL;
    const size_t n = p->N, m = p->M, w = (n + 31) / 32;
    // Solves of several parts share the SIMDs with each other and with the other parts' front waves: a part's small
    // solves had better leave early than wait for its large ones (measured: 104 against 107 us per step of 8 192 envs).
    MarlEnvParams pp = *p;
    if (parts > 1) pp.stage_flags |= MARL_STAGE_NO_SOLVE_PRIORITY;
    int32_t first = 0;
    for (int32_t k = 0; k < parts; ++k) {
        const int32_t count = s->E / parts + (k < s->E % parts ? 1 : 0);
        MarlEnvState part;
        rc = marl_state_shard(p, s, first, count, (unsigned char*)scratch + (size_t)k * marl_shard_scratch_bytes(), &part);
        if (rc != MARL_OK) return rc;
        MarlStepIO o = *io;
        const size_t f = (size_t)first;
        o.actions = rows_from(io->actions, f, m * 3); o.noise = rows_from(io->noise, f, n * 2);
        o.obs = rows_from(io->obs, f, m * MARL_OBS_DIM); o.reward = rows_from(io->reward, f, 1);
        o.truncated = rows_from(io->truncated, f, 1); o.lambda2 = rows_from(io->lambda2, f, 1);
        o.reward_terms = rows_from(io->reward_terms, f, 5); o.avg_power_dbm = rows_from(io->avg_power_dbm, f, 1);
        o.adj_bits = rows_from(io->adj_bits, f, n * w); o.jam_bits = rows_from(io->jam_bits, f, n * w);
        o.env_offset = io->env_offset + f;
        rc = marl_env_step(&pp, &part, &o, streams[k]);
        if (rc != MARL_OK) return rc;
        first += count;
    }
    return MARL_OK;
}

int marl_env_reset_parts(const MarlEnvParams* p, const MarlEnvState* s, const MarlResetIO* io, int32_t parts,
                         void* const* streams, void* scratch) {
    int rc = part_bounds(p, s, parts, streams, scratch);
    if (rc != MARL_OK || io == nullptr) return rc != MARL_OK ? rc : MARL_EINVAL;
    const size_t n = p->N, m = p->M;
    int32_t first = 0;
    for (int32_t k = 0; k < parts; ++k) {
        const int32_t count = s->E / parts + (k < s->E % parts ? 1 : 0);
        MarlEnvState part;
        rc = marl_state_shard(p, s, first, count, (unsigned char*)scratch + (size_t)k * marl_shard_scratch_bytes(), &part);
        if (rc != MARL_OK) return rc;
        MarlResetIO o = *io;
        const size_t f = (size_t)first;
        o.mask = rows_from(io->mask, f, 1); o.enemy_pos = rows_from(io->enemy_pos, f, n * 2);
        o.enemy_vel = rows_from(io->enemy_vel, f, n * 2); o.enemy_omega = rows_from(io->enemy_omega, f, n);
        o.enemy_band = rows_from(io->enemy_band, f, 1); o.jam_init = rows_from(io->jam_init, f, m * 2);
        o.jam_band = rows_from(io->jam_band, f, m); o.obs = rows_from(io->obs, f, m * MARL_OBS_DIM);
        o.env_offset = io->env_offset + f;
        rc = marl_env_reset(p, &part, &o, streams[k]);
        if (rc != MARL_OK) return rc;
        first += count;
    }
    return MARL_OK;
}
#undef rows_from

int marl_adjacency_bits(const MarlEnvParams* p, int32_t E, const double* x, const double* y, const int32_t* band,
                        uint32_t* bits, void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (E < 1 || !x || !y || !band || !bits) return MARL_EINVAL;
    if (p->N > WAVE) {
        MARL_DISPATCH_SLOTS(p->N, pair_bits_big_kernel, dim3(E), stream, *p, x, y, band, (const double*)nullptr,
                            (const double*)nullptr, (const int32_t*)nullptr, bits, 0);
        return hip_check(hipGetLastError());
    }
    hipLaunchKernelGGL(adjacency_bits_kernel, dim3(E), dim3(WAVE), wave_lds_bytes(p->N), (hipStream_t)stream, *p, x, y,
                       band, bits);
    return hip_check(hipGetLastError());
}

int marl_jam_mask_bits(const MarlEnvParams* p, int32_t E, const double* x, const double* y, const int32_t* band,
                       const double* jam_x, const double* jam_y, const int32_t* jam_band, uint32_t* bits,
                       void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (E < 1 || !x || !y || !band || !jam_x || !jam_y || !jam_band || !bits) return MARL_EINVAL;
    if (p->N > WAVE) {
        MARL_DISPATCH_SLOTS(p->N, pair_bits_big_kernel, dim3(E), stream, *p, x, y, band, jam_x, jam_y, jam_band, bits, 1);
        return hip_check(hipGetLastError());
    }
    hipLaunchKernelGGL(jam_bits_kernel, dim3(E), dim3(WAVE), wave_lds_bytes(p->N), (hipStream_t)stream, *p, x, y, band,
                       jam_x, jam_y, jam_band, bits);
    return hip_check(hipGetLastError());
}

size_t marl_lambda2_workspace_bytes(int32_t N, int32_t E) {
    if (N < 1 || N > MARL_MAX_N_LAMBDA2 || E < 1) return 0;
    if (N > MARL_MAX_N) {
        const size_t ld = (size_t)marl_lambda2_large_ld(N);
        return (size_t)E * ld * ld * sizeof(double);
    }
    return workspace_bytes(N, E);
}

int marl_lambda2_from_bits(int32_t N, int32_t E, const uint32_t* bits, double* lambda2, int32_t rounds,
                           void* workspace, void* stream) {
    if (N < 1 || E < 1 || !bits || !lambda2) return MARL_EINVAL;
    if (N > MARL_MAX_N_LAMBDA2) return MARL_EUNSUPPORTED;
    if (N > MARL_MAX_N) {                          // the stand-alone solver for large graphs (marl_lambda2_large.hip)
        if (workspace == nullptr) return MARL_EINVAL;
        return hip_check((hipError_t)marl_lambda2_large_launch(N, E, bits, lambda2, (double*)workspace, stream));
    }
    if (check_rounds(&rounds) != MARL_OK) return MARL_EINVAL;
    const bool lanczos = (rounds & MARL_LAMBDA2_LANCZOS) != 0;       // the stage op takes the direct solver unless asked
    const bool few_steps = (rounds & MARL_LAMBDA2_TEST_FEW_STEPS) != 0;
    rounds &= 0xFF;
    if (N > WAVE) {
        if (workspace == nullptr) return MARL_EINVAL;
        int* pend = workspace_pending((double*)workspace, N, E);
        MARL_DISPATCH_SLOTS(N, lambda2_bits_big_kernel, dim3(E), stream, (int)N, bits, lambda2, (int)rounds,
                            (double*)workspace, pend, (int)lanczos);
        const size_t ld = (size_t)big_ld(N);
        if (lanczos) {
            int* team_list = workspace_team_list((double*)workspace, N, E);
            MARL_DISPATCH_ROWS(N, lambda2_lanczos_rows_kernel, dim3(E * ROWS_PASSES), stream, (int)N, (int)E, pend, (double*)workspace, (int)ld, team_list,
                               lambda2, (int)rounds, few_steps ? 24 : LANCZOS_ROWS_KMAX);
            MARL_LAUNCH_TEAM(N, lambda2_team_kernel, (E < TEAM_LIST_MAX ? E : TEAM_LIST_MAX), stream, (int)N, (const int*)pend,
                             (const double*)workspace, ld * ld, lambda2, (int)rounds, (const int*)team_list);
        } else {
            MARL_LAUNCH_TEAM(N, lambda2_team_kernel, E, stream, (int)N, (const int*)pend, (const double*)workspace,
                             ld * ld, lambda2, (int)rounds, (const int*)nullptr);
        }
        return hip_check(hipGetLastError());
    }
    if (lanczos && N > 32) {
        MARL_DISPATCH_NPAD_BIG(N, lambda2_bits_lanczos_kernel, dim3(E), wave_lds_bytes(N), stream, (int)N, bits, lambda2, (int)rounds);
    } else {
        MARL_DISPATCH_NPAD(N, lambda2_bits_kernel, dim3(E), wave_lds_bytes(N), stream, (int)N, bits, lambda2, (int)rounds);
    }
    return hip_check(hipGetLastError());
}

int marl_lambda2_from_bits_static(int32_t N, int32_t E, const uint32_t* bits, double* lambda2, int32_t rounds,
                                  void* stream) {
    if (N < 1 || E < 1 || !bits || !lambda2) return MARL_EINVAL;
    if (N > MARL_MAX_N) return MARL_EUNSUPPORTED;
    const size_t need = workspace_bytes(N, E);
    if (need > (size_t)MARL_STATIC_SCRATCH_BYTES) return MARL_ETOOBIG;
    void* ws = nullptr;
    if (need > 0) {
        // Nothing in the scratch has to start at zero: the first launch writes the pending list entry of
        // every env and the rows / matrix of every env it leaves pending before the second launch reads them.
        const int rc = static_scratch(&ws);
        if (rc != MARL_OK) return rc;
    }
    return marl_lambda2_from_bits(N, E, bits, lambda2, rounds, ws, stream);
}

int marl_dbscan(const MarlEnvParams* p, int32_t E, const double* x, const double* y, int32_t* labels,
                double* cent_x, double* cent_y, int32_t* cent_size, int32_t* n_clusters, int32_t* ncent,
                void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (E < 1 || !x || !y || !labels || !cent_x || !cent_y || !cent_size || !n_clusters || !ncent)
        return MARL_EINVAL;
    if (p->N > WAVE) {
        MARL_DISPATCH_SLOTS(p->N, dbscan_big_kernel, dim3(E), stream, *p, x, y, labels, cent_x, cent_y, cent_size,
                            n_clusters, ncent);
        return hip_check(hipGetLastError());
    }
    hipLaunchKernelGGL(dbscan_kernel, dim3(E), dim3(WAVE), wave_lds_bytes(p->N), (hipStream_t)stream, *p, x, y, labels,
                       cent_x, cent_y, cent_size, n_clusters, ncent);
    return hip_check(hipGetLastError());
}

int marl_gae(int32_t T, int32_t E, const float* rewards, const float* values, const float* dones,
             const float* last_value, const float* last_done, double gamma, double lam, float* advantages,
             float* returns, void* stream) {
    if (T < 1 || E < 1 || !rewards || !values || !dones || !last_value || !last_done || !advantages || !returns)
        return MARL_EINVAL;
    const int block = 256;
    hipLaunchKernelGGL(gae_kernel, dim3((E + block - 1) / block), dim3(block), 0, (hipStream_t)stream, (int)T,
                       (int)E, rewards, values, dones, last_value, last_done, (float)gamma, (float)(gamma * lam), advantages,
                       returns);
    return hip_check(hipGetLastError());
}

int marl_reward_per_agent(const MarlEnvParams* p, int32_t E, const double* lambda2, const double* lambda2_initial,
                          const int32_t* enemy_band, const double* jam_x, const double* jam_y, const int32_t* jam_band,
                          const float* vx, const float* vy, const double* cent_x, const double* cent_y,
                          const int32_t* n_clusters, const int32_t* ncent, const int32_t* assign, float* rewards,
                          void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (E < 1 || !lambda2 || !lambda2_initial || !enemy_band || !jam_x || !jam_y || !jam_band || !vx || !vy ||
        !cent_x || !cent_y || !n_clusters || !ncent || !assign || !rewards)
        return MARL_EINVAL;
    hipLaunchKernelGGL(reward_per_agent_kernel, dim3(E), dim3(WAVE), 0, (hipStream_t)stream, *p, lambda2,
                       lambda2_initial, enemy_band, jam_x, jam_y, jam_band, vx, vy, cent_x, cent_y, n_clusters, ncent,
                       assign, rewards);
    return hip_check(hipGetLastError());
}

int marl_reward_obs(const MarlEnvParams* p, int32_t E, const MarlRewardObsIO* io, void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (E < 1 || io == nullptr || !io->lambda2 || !io->lambda2_initial || !io->enemy_band || !io->jam_x || !io->jam_y ||
        !io->jam_band || !io->vx || !io->vy || !io->cent_x || !io->cent_y || !io->cent_size || !io->n_clusters ||
        !io->ncent || !io->assign)
        return MARL_EINVAL;
    if (p->N > WAVE) {
        MARL_DISPATCH_SLOTS(p->N, reward_obs_big_kernel, dim3(E), stream, *p, *io);
        return hip_check(hipGetLastError());
    }
    hipLaunchKernelGGL(reward_obs_kernel, dim3(E), dim3(WAVE), wave_lds_bytes(p->N), (hipStream_t)stream, *p, *io);
    return hip_check(hipGetLastError());
}

int marl_fill_noise(const MarlEnvParams* p, const MarlEnvState* s, uint64_t seed, uint64_t env_offset,
                    double* noise, void* stream) {
    int rc = check_sizes(p);
    if (rc != MARL_OK) return rc;
    if (!s || s->E < 1 || !noise) return MARL_EINVAL;
    hipLaunchKernelGGL(fill_noise_kernel, dim3(s->E), dim3(MARL_MAX_N), 0, (hipStream_t)stream, *p, *s, seed,
                       env_offset, noise);
    return hip_check(hipGetLastError());
}

}  // extern "C"
