// Policy inference for the rollout loop: PPOAgent.get_action of the reference (agents/ppo_agent.py:193-225)
// for every agent of every env in one launch -- Actor.sample (agents/actor.py:118-202: trunk 5 -> H -> H with
// LayerNorm + ReLU, heads for the velocity mean, its clamped log-std and the band logits; Normal / Categorical
// draw, log-probability of the unclamped draw, velocity clamped to +-v_max for the env) and
// Critic.forward_pooled on the mean-pooled observation (agents/critic.py:71-134).  The results are written
// where the caller points: the action / log-prob / value slots of a device rollout buffer.
//
// The three layers of each network are dense contractions and run on the matrix cores in float32
// (v_mfma_f32_32x32x2): hidden units on the rows of the product, 32 agents (or envs) on its columns,
//     D[unit][agent] += W[unit][k] * x[k][agent].
// With that orientation a lane owns ONE agent (column lane % 32) and half of its hidden units
// (unit u is held by half hl = (u / 4) % 2 of the wave: the accumulator layout of the instruction), so
//   * LayerNorm and ReLU are lane-local plus one exchange with the lane 32 away,
//   * the accumulators of one layer are, register for register, the B operand of the next: the k-th step of the
//     next contraction takes the k-th unit a lane owns, and the weights are stored pre-permuted to match
//     (host side: pack_policy_weights), one 16-byte LDS read per four steps,
//   * the head outputs land in the first registers of the lanes (means and log-stds in the lower half of the wave,
//     band logits in the upper half).
// A workgroup (8 waves) keeps one network's permuted weights in LDS and walks over tiles of 32 agents; the
// grid is one workgroup per CU, split between actor and critic in proportion to their work.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "marl_env.h"
#include "marl_net.h"
#include "marl_rng.h"

namespace {

using namespace marl_rng;
using namespace marl_net;
constexpr int WAVE = 64;
constexpr int OBS = MARL_OBS_DIM;
constexpr int POLICY_THREADS = 512;
constexpr int POLICY_WAVES = POLICY_THREADS / WAVE;

// Offsets (in floats) of the packed weight buffer, see marl_policy_weight_count in marl_env.h.
template <int H>
struct Pack {
    static constexpr int NB = H / 32, NS = H / 2;
    static constexpr int a_f1 = 0;                          // [NB][3][64]     layer-1 fragments (bias as input 5)
    static constexpr int a_g1 = a_f1 + NB * 3 * 64;         // [H] LayerNorm 1 weight
    static constexpr int a_e1 = a_g1 + H;                   // [H] LayerNorm 1 bias
    static constexpr int a_f2 = a_e1 + H;                   // [NB][NS/4][64][4] layer-2 fragments
    static constexpr int a_b2 = a_f2 + H * H;               // [H]
    static constexpr int a_g2 = a_b2 + H;
    static constexpr int a_e2 = a_g2 + H;
    static constexpr int a_fh = a_e2 + H;                   // [NS/4][64][4]   head fragments (rows 8..31 zero)
    static constexpr int a_bh = a_fh + NS * 64;             // [8]
    static constexpr int a_end = a_bh + 8;
    static constexpr int c_f1 = a_end;                      // critic [NB][3][64]
    static constexpr int c_f2 = c_f1 + NB * 3 * 64;         // [NB][NS/4][64][4]
    static constexpr int c_b2 = c_f2 + H * H;               // [H]
    static constexpr int c_w3 = c_b2 + H;                   // [H]
    static constexpr int c_b3 = c_w3 + H;                   // [1] + 3 pad
    static constexpr int total = c_b3 + 4;
    static constexpr int lds_floats = a_end > total - c_f1 ? a_end : total - c_f1;
};

template <int H>
__global__ __launch_bounds__(POLICY_THREADS) void policy_act_kernel(const float* __restrict__ wts, const float* __restrict__ obs,
                                                                     const int32_t* __restrict__ tick, float* __restrict__ actions,
                                                                     float* __restrict__ log_probs, float* __restrict__ values,
                                                                     int E, int M, float v_max, float ls_min, float ls_max,
                                                                     uint64_t seed, uint64_t env_offset, uint32_t call,
                                                                     int deterministic, int actor_blocks) {
    using P = Pack<H>;
    constexpr int NB = P::NB, NS = P::NS;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int hl = lane >> 5, j = lane & 31;
    const bool actor = (int)blockIdx.x < actor_blocks;
    {   // this network's fragments -> LDS
        const float4* src = reinterpret_cast<const float4*>(wts + (actor ? 0 : P::c_f1));
        const int n4 = (actor ? P::a_end : P::total - P::c_f1) / 4;
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < n4; i += POLICY_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    if (actor) {
        const int rows = E * M;
        const int tiles = (rows + 31) / 32;
        for (int tile = blockIdx.x * POLICY_WAVES + wave; tile < tiles; tile += actor_blocks * POLICY_WAVES) {
            const int row = tile * 32 + j;
            const bool valid = row < rows;
            const int r = valid ? row : rows - 1;
            const float* o = obs + (size_t)r * OBS;
            const float x0 = o[hl], x1 = o[2 + hl], x2 = hl ? 1.0f : o[4];
            floatx16 y1[NB], y2[NB];
            first_layer<NB>(lds + P::a_f1, lane, x0, x1, x2, y1);
            layer_norm_relu<NB>(y1, lds + P::a_g1, lds + P::a_e1, hl);
            unit_vector<NB>(lds + P::a_b2, hl, y2);
            hidden_layer<NB>(lds + P::a_f2, lane, y1, y2);
            layer_norm_relu<NB>(y2, lds + P::a_g2, lds + P::a_e2, hl);
            // heads: two accumulators (even / odd steps) so that consecutive instructions do not depend on each other
            floatx16 h0 = {0}, h1 = {0};
            {
                const float4 bh = *reinterpret_cast<const float4*>(lds + P::a_bh + 4 * hl);
                h0[0] = bh.x; h0[1] = bh.y; h0[2] = bh.z; h0[3] = bh.w;
            }
            const float* fh = lds + P::a_fh + opaque(lane * 4);
            static_for<0, NS / 4>([&](auto ss) {
                constexpr int s4 = decltype(ss)::value;
                const float4 w = *reinterpret_cast<const float4*>(fh + s4 * 256);
                h0 = mfma(w.x, y2[(4 * s4) / 16][(4 * s4) % 16], h0);
                h1 = mfma(w.y, y2[(4 * s4 + 1) / 16][(4 * s4 + 1) % 16], h1);
                h0 = mfma(w.z, y2[(4 * s4 + 2) / 16][(4 * s4 + 2) % 16], h0);
                h1 = mfma(w.w, y2[(4 * s4 + 3) / 16][(4 * s4 + 3) % 16], h1);
                LDS_FENCE();
            });
            // this lane: heads 4 hl .. 4 hl + 3 of its agent; the partner (lane ^ 32) holds the other four
            float mine[4], other[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { mine[q] = h0[q] + h1[q]; other[q] = __shfl_xor(mine[q], 32); }
            const float mu0 = hl ? other[0] : mine[0], mu1 = hl ? other[1] : mine[1];
            const float rl0 = hl ? other[2] : mine[2], rl1 = hl ? other[3] : mine[3];
            float lg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) lg[q] = hl ? mine[q] : other[q];
            // distributions (actor.py:160-196); both halves compute, the lower half writes
            const float ls0 = fminf(fmaxf(rl0, ls_min), ls_max), ls1 = fminf(fmaxf(rl1, ls_min), ls_max);
            const float sd0 = expf(ls0), sd1 = expf(ls1);
            const float lmax = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
            float ex[4], esum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { ex[q] = expf(lg[q] - lmax); esum += ex[q]; }
            const float lse = lmax + logf(esum);
            float vx, vy;
            int band;
            if (deterministic) {
                vx = mu0; vy = mu1;
                band = 0;
#pragma unroll
                for (int q = 1; q < 4; ++q) if (lg[q] > lg[band]) band = q;          // first maximum, as argmax
            } else {
                const int e = r / M, a = r - e * M;
                const uint64_t genv = env_offset + e;
                uint32_t rn[4];
                philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)a << 8) ^ STREAM_POLICY,
                           tick ? (uint32_t)tick[e] : 0u, call, (uint32_t)seed, (uint32_t)(seed >> 32), rn);
                // Box-Muller for the two velocity draws, inverse CDF for the band
                const float u1 = ((float)(rn[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                const float u2 = ((float)(rn[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
                const float rad = sqrtf(-2.0f * logf(u1));
                float sn, cs;
                sincosf(6.2831853071795865f * u2, &sn, &cs);
                vx = fmaf(sd0, rad * cs, mu0);
                vy = fmaf(sd1, rad * sn, mu1);
                const float u3 = ((float)(rn[2] >> 8) + 0.5f) * (1.0f / 16777216.0f) * esum;
                band = 3;
                float cum = ex[0];
                if (u3 < cum) band = 0;
                else { cum += ex[1]; if (u3 < cum) band = 1; else { cum += ex[2]; if (u3 < cum) band = 2; } }
            }
            // Normal.log_prob of the unclamped draw, summed over the two components, + Categorical.log_prob
            const float z0 = (vx - mu0) / sd0, z1 = (vy - mu1) / sd1;
            const float lp = (-0.5f * z0 * z0 - ls0 - 0.91893853320467274f) + (-0.5f * z1 * z1 - ls1 - 0.91893853320467274f)
                             + (lg[band] - lse);
            if (valid && hl == 0) {
                float* out = actions + (size_t)row * 3;
                out[0] = fminf(fmaxf(vx, -v_max), v_max);
                out[1] = fminf(fmaxf(vy, -v_max), v_max);
                out[2] = (float)band;
                log_probs[row] = lp;
            }
        }
    } else {
        const int nblocks = gridDim.x - actor_blocks;
        const int tiles = (E + 31) / 32;
        // the critic's part of the buffer sits at the start of LDS: same offsets relative to c_f1
        const float* cf1 = lds;
        const float* cf2 = lds + (P::c_f2 - P::c_f1);
        const float* cb2 = lds + (P::c_b2 - P::c_f1);
        const float* cw3 = lds + (P::c_w3 - P::c_f1);
        const float cb3 = lds[P::c_b3 - P::c_f1];
        for (int tile = ((int)blockIdx.x - actor_blocks) * POLICY_WAVES + wave; tile < tiles; tile += nblocks * POLICY_WAVES) {
            const int env = tile * 32 + j;
            const bool valid = env < E;
            const int e = valid ? env : E - 1;
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f;                 // inputs hl, 2 + hl, 4 of the mean-pooled observation
            for (int m = 0; m < M; ++m) {
                const float* o = obs + ((size_t)e * M + m) * OBS;
                p0 += o[hl]; p1 += o[2 + hl]; p2 += o[4];
            }
            const float x0 = p0 / (float)M, x1 = p1 / (float)M, x2 = hl ? 1.0f : p2 / (float)M;
            floatx16 y1[NB], y2[NB];
            first_layer<NB>(cf1, lane, x0, x1, x2, y1);
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                y1[decltype(bb)::value][decltype(vv)::value] = fmaxf(y1[decltype(bb)::value][decltype(vv)::value], 0.0f); }); });
            unit_vector<NB>(cb2, hl, y2);
            hidden_layer<NB>(cf2, lane, y1, y2);
            float v = 0.0f;
            static_for<0, NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                static_for<0, 4>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    const float4 wq = *reinterpret_cast<const float4*>(cw3 + 32 * b + 8 * q + 4 * hl);
                    v = fmaf(wq.x, fmaxf(y2[b][4 * q], 0.0f), v);
                    v = fmaf(wq.y, fmaxf(y2[b][4 * q + 1], 0.0f), v);
                    v = fmaf(wq.z, fmaxf(y2[b][4 * q + 2], 0.0f), v);
                    v = fmaf(wq.w, fmaxf(y2[b][4 * q + 3], 0.0f), v);
                    LDS_FENCE();
                });
            });
            v += __shfl_xor(v, 32);
            if (valid && hl == 0) values[env] = v + cb3;
        }
    }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) void marl_note_hip_error(int err);     // marl_env.hip

extern "C" {

size_t marl_policy_weight_count(int32_t hidden) {
    switch (hidden) {
        case 64: return Pack<64>::total;
        case 128: return Pack<128>::total;
        default: return 0;
    }
}

int marl_policy_act(int32_t E, int32_t M, int32_t hidden, float v_max, float log_std_min, float log_std_max,
                    const MarlPolicyIO* io, void* stream) {
    if (io == nullptr || E < 1 || M < 1 || M > MARL_MAX_M) return MARL_EINVAL;
    if (io->weights == nullptr || io->obs == nullptr || io->actions == nullptr || io->log_probs == nullptr)
        return MARL_EINVAL;
    if ((long long)E * M > (1ll << 30)) return MARL_EUNSUPPORTED;
    if (hidden != 64 && hidden != 128) return MARL_EUNSUPPORTED;
    // one workgroup per CU; the critic gets its share of them (tiles x matrix instructions per tile)
    const long long ta = ((long long)E * M + 31) / 32, tc = io->values ? (E + 31) / 32 : 0;
    const double wa = (double)ta * (hidden / 32 * 3 + hidden / 32 * hidden / 2 + hidden / 2);
    const double wc = (double)tc * (hidden / 32 * 3 + hidden / 32 * hidden / 2);
    int grid = marl_net::device_cus(256);
    if (ta + tc < (long long)grid * POLICY_WAVES) grid = (int)((ta + tc + POLICY_WAVES - 1) / POLICY_WAVES) + (tc ? 1 : 0);
    int critic_blocks = tc ? (int)(grid * wc / (wa + wc) + 0.5) : 0;
    if (tc && critic_blocks < 1) critic_blocks = 1;
    if (critic_blocks >= grid) critic_blocks = grid - 1;
    if (grid < 1 + (tc ? 1 : 0)) { grid = 1 + (tc ? 1 : 0); critic_blocks = tc ? 1 : 0; }
    const int actor_blocks = grid - critic_blocks;
    hipError_t err = hipSuccess;
#define MARL_LAUNCH_POLICY(HH)                                                                                     \
    do {                                                                                                            \
        static bool attr_set[marl_net::kMaxDevices];                                                                \
        const size_t lds_bytes = (size_t)Pack<HH>::lds_floats * sizeof(float);                                      \
        err = marl_net::allow_dynamic_lds((const void*)policy_act_kernel<HH>, lds_bytes, attr_set);                 \
        if (err == hipSuccess)                                                                                      \
            hipLaunchKernelGGL(policy_act_kernel<HH>, dim3(grid), dim3(POLICY_THREADS), lds_bytes, (hipStream_t)stream, \
                               io->weights, io->obs, io->tick, io->actions, io->log_probs, io->values, (int)E, (int)M, \
                               v_max, log_std_min, log_std_max, io->seed, io->env_offset, io->call,                 \
                               (int)io->deterministic, actor_blocks);                                               \
    } while (0)
    if (hidden == 64) MARL_LAUNCH_POLICY(64); else MARL_LAUNCH_POLICY(128);
#undef MARL_LAUNCH_POLICY
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

}  // extern "C"
