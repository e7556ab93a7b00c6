// Policy inference for the rollout loop: PPOAgent.get_action of the reference (agents/ppo_agent.py:193-225)
// for every agent of every env in one launch -- Actor.sample (agents/actor.py:118-202: trunk 5 -> H -> H with
// LayerNorm + ReLU, heads for the velocity mean, its clamped log-std and the band logits; Normal / Categorical
// draw, log-probability of the unclamped draw, velocity clamped to +-v_max for the env) and
// Critic.forward_pooled on the mean-pooled observation (agents/critic.py:71-134).  The results are written
// where the caller points: the action / log-prob / value slots of a device rollout buffer.
//
// One lane = one agent (actor part) or one env (critic part), so LayerNorm and the heads are lane-local and
// every weight is wave-uniform: weights come through the scalar cache as SGPR operands of the multiply-adds,
// two output units per instruction, two independent accumulator chains (the hidden-layer matrix is stored
// interleaved in groups of four output units for that).  float32
// throughout, like the networks it mirrors.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "marl_env.h"
#include "marl_rng.h"

namespace {

using namespace marl_rng;
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int WAVE = 64;
constexpr int OBS = MARL_OBS_DIM;
constexpr int HEADS = 8;                 // mu_x, mu_y, log_std_x, log_std_y, 4 band logits

// Offsets (in floats) of the packed weight buffer, see marl_policy_weight_count in marl_env.h.
template <int H>
struct Pack {
    static constexpr int a_w1 = 0;                         // [H][5]
    static constexpr int a_b1 = a_w1 + H * OBS;            // [H]
    static constexpr int a_g1 = a_b1 + H;                  // LayerNorm 1 weight
    static constexpr int a_e1 = a_g1 + H;                  // LayerNorm 1 bias
    static constexpr int a_w2 = a_e1 + H;                  // [H/4][H][4]: (W2[4t][k], W2[4t+1][k], W2[4t+2][k], W2[4t+3][k])
    static constexpr int a_b2 = a_w2 + H * H;
    static constexpr int a_g2 = a_b2 + H;
    static constexpr int a_e2 = a_g2 + H;
    static constexpr int a_wh = a_e2 + H;                  // [H][8]: column j of the stacked head matrix
    static constexpr int a_bh = a_wh + H * HEADS;          // [8]
    static constexpr int c_w1 = a_bh + HEADS;              // critic [H][5]
    static constexpr int c_b1 = c_w1 + H * OBS;
    static constexpr int c_w2 = c_b1 + H;                  // [H/4][H][4]
    static constexpr int c_b2 = c_w2 + H * H;
    static constexpr int c_w3 = c_b2 + H;                  // [H]
    static constexpr int c_b3 = c_w3 + H;                  // [1]
    static constexpr int total = c_b3 + 1;
};

__device__ __forceinline__ float2v fma2(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }

// Four units of the hidden layer at once: (a0, a1) += W2[4t .. 4t+3][:] . h, with h held as H/2 register pairs
// whose halves are selected by the instruction (no copies), weights wave-uniform and contiguous.
template <int H>
__device__ __forceinline__ void four_units(const float* __restrict__ w4, float2v (&hp)[H / 2], float2v& a0, float2v& a1) {
    const float2v* __restrict__ w = reinterpret_cast<const float2v*>(w4);
#pragma unroll
    for (int kp = 0; kp < H / 2; ++kp) {
        // opaque redefinition inside the caller's loop: keeps the broadcasts below from being hoisted out of it as
        // H extra register pairs; as operands of the multiply-add they cost nothing (op_sel picks the half)
        asm volatile("" : "+v"(hp[kp]));
        const float2v lo = __builtin_shufflevector(hp[kp], hp[kp], 0, 0), hi = __builtin_shufflevector(hp[kp], hp[kp], 1, 1);
        a0 = fma2(w[4 * kp], lo, a0);
        a1 = fma2(w[4 * kp + 1], lo, a1);
        a0 = fma2(w[4 * kp + 2], hi, a0);
        a1 = fma2(w[4 * kp + 3], hi, a1);
    }
}

// h[j] = b[j] + W[j][:] . o   for the 5 observation features (weights wave-uniform)
template <int H>
__device__ __forceinline__ void first_layer(const float* __restrict__ w, const float* __restrict__ b,
                                            const float (&o)[OBS], float (&h)[H]) {
#pragma unroll
    for (int j = 0; j < H; ++j) {
        float x = b[j];
#pragma unroll
        for (int k = 0; k < OBS; ++k) x = fmaf(w[j * OBS + k], o[k], x);
        h[j] = x;
    }
}

template <int H>
__global__ __launch_bounds__(WAVE) void policy_act_kernel(const float* __restrict__ wts, const float* __restrict__ obs,
                                                          const int32_t* __restrict__ tick, float* __restrict__ actions,
                                                          float* __restrict__ log_probs, float* __restrict__ values,
                                                          int E, int M, float v_max, float ls_min, float ls_max,
                                                          uint64_t seed, uint64_t env_offset, uint32_t call,
                                                          int deterministic, int actor_blocks) {
    using P = Pack<H>;
    __shared__ float h2s[H * WAVE];                   // hidden layer 2 before its LayerNorm: [unit][lane]
    const int lane = threadIdx.x;
    if ((int)blockIdx.x < actor_blocks) {
        const int rows = E * M;
        const int row = blockIdx.x * WAVE + lane;
        const bool valid = row < rows;
        const int r = valid ? row : rows - 1;
        float o[OBS];
#pragma unroll
        for (int k = 0; k < OBS; ++k) o[k] = obs[(size_t)r * OBS + k];
        float h[H];
        first_layer<H>(wts + P::a_w1, wts + P::a_b1, o, h);
        {   // LayerNorm (eps 1e-5, biased variance) + ReLU
            float mean = 0.0f;
#pragma unroll
            for (int j = 0; j < H; ++j) mean += h[j];
            mean *= 1.0f / H;
            float var = 0.0f;
#pragma unroll
            for (int j = 0; j < H; ++j) { const float d = h[j] - mean; var = fmaf(d, d, var); }
            const float rstd = 1.0f / sqrtf(var * (1.0f / H) + 1e-5f);
#pragma unroll
            for (int j = 0; j < H; ++j)
                h[j] = fmaxf(fmaf((h[j] - mean) * rstd, wts[P::a_g1 + j], wts[P::a_e1 + j]), 0.0f);
        }
        float2v hp[H / 2];
#pragma unroll
        for (int j = 0; j < H / 2; ++j) {
            hp[j] = (float2v){h[2 * j], h[2 * j + 1]};
            asm volatile("" : "+v"(hp[j]));         // keep the pair as one 64-bit value: its halves are picked by op_sel
        }
        float sum2 = 0.0f;
        for (int t = 0; t < H / 4; ++t) {            // four hidden units per trip; the k loop is straight-line code
            float2v a0 = {wts[P::a_b2 + 4 * t], wts[P::a_b2 + 4 * t + 1]}, a1 = {wts[P::a_b2 + 4 * t + 2], wts[P::a_b2 + 4 * t + 3]};
            four_units<H>(wts + P::a_w2 + (size_t)t * H * 4, hp, a0, a1);
            h2s[(4 * t) * WAVE + lane] = a0.x;
            h2s[(4 * t + 1) * WAVE + lane] = a0.y;
            h2s[(4 * t + 2) * WAVE + lane] = a1.x;
            h2s[(4 * t + 3) * WAVE + lane] = a1.y;
            sum2 += (a0.x + a0.y) + (a1.x + a1.y);
        }
        const float mean2 = sum2 * (1.0f / H);
        float var2 = 0.0f;
        for (int j = 0; j < H; ++j) { const float d = h2s[j * WAVE + lane] - mean2; var2 = fmaf(d, d, var2); }
        const float rstd2 = 1.0f / sqrtf(var2 * (1.0f / H) + 1e-5f);
        float head[HEADS];
#pragma unroll
        for (int q = 0; q < HEADS; ++q) head[q] = wts[P::a_bh + q];
        for (int j = 0; j < H; ++j) {
            const float y = fmaxf(fmaf((h2s[j * WAVE + lane] - mean2) * rstd2, wts[P::a_g2 + j], wts[P::a_e2 + j]), 0.0f);
#pragma unroll
            for (int q = 0; q < HEADS; ++q) head[q] = fmaf(wts[P::a_wh + j * HEADS + q], y, head[q]);
        }
        // distributions (actor.py:160-196)
        const float ls0 = fminf(fmaxf(head[2], ls_min), ls_max), ls1 = fminf(fmaxf(head[3], ls_min), ls_max);
        const float sd0 = expf(ls0), sd1 = expf(ls1);
        float lmax = fmaxf(fmaxf(head[4], head[5]), fmaxf(head[6], head[7]));
        float ex[4], esum = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ex[q] = expf(head[4 + q] - lmax); esum += ex[q]; }
        const float lse = lmax + logf(esum);
        float vx, vy;
        int band;
        if (deterministic) {
            vx = head[0]; vy = head[1];
            band = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q) if (head[4 + q] > head[4 + band]) band = q;      // first maximum, as argmax
        } else {
            const int e = r / M, j = r - e * M;
            const uint64_t genv = env_offset + e;
            uint32_t rn[4];
            philox4x32((uint32_t)genv, (uint32_t)(genv >> 32) ^ ((uint32_t)j << 8) ^ STREAM_POLICY,
                       tick ? (uint32_t)tick[e] : 0u, call, (uint32_t)seed, (uint32_t)(seed >> 32), rn);
            // Box-Muller for the two velocity draws, inverse CDF for the band
            const float u1 = ((float)(rn[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((float)(rn[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.2831853071795865f * u2, &sn, &cs);
            vx = fmaf(sd0, rad * cs, head[0]);
            vy = fmaf(sd1, rad * sn, head[1]);
            const float u3 = ((float)(rn[2] >> 8) + 0.5f) * (1.0f / 16777216.0f) * esum;
            band = 3;
            float cum = ex[0];
            if (u3 < cum) band = 0;
            else { cum += ex[1]; if (u3 < cum) band = 1; else { cum += ex[2]; if (u3 < cum) band = 2; } }
        }
        // Normal.log_prob of the unclamped draw, summed over the two components, + Categorical.log_prob
        const float z0 = (vx - head[0]) / sd0, z1 = (vy - head[1]) / sd1;
        const float lp = (-0.5f * z0 * z0 - ls0 - 0.91893853320467274f) + (-0.5f * z1 * z1 - ls1 - 0.91893853320467274f)
                         + (head[4 + band] - lse);
        if (valid) {
            float* a = actions + (size_t)row * 3;
            a[0] = fminf(fmaxf(vx, -v_max), v_max);
            a[1] = fminf(fmaxf(vy, -v_max), v_max);
            a[2] = (float)band;
            log_probs[row] = lp;
        }
    } else if (values != nullptr) {
        const int env = ((int)blockIdx.x - actor_blocks) * WAVE + lane;
        const bool valid = env < E;
        const int e = valid ? env : E - 1;
        float o[OBS];
#pragma unroll
        for (int k = 0; k < OBS; ++k) o[k] = 0.0f;
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int k = 0; k < OBS; ++k) o[k] += obs[((size_t)e * M + m) * OBS + k];
#pragma unroll
        for (int k = 0; k < OBS; ++k) o[k] = o[k] / (float)M;              // obs.mean(dim=1)
        float h[H];
        first_layer<H>(wts + P::c_w1, wts + P::c_b1, o, h);
#pragma unroll
        for (int j = 0; j < H; ++j) h[j] = fmaxf(h[j], 0.0f);
        float2v hp[H / 2];
#pragma unroll
        for (int j = 0; j < H / 2; ++j) {
            hp[j] = (float2v){h[2 * j], h[2 * j + 1]};
            asm volatile("" : "+v"(hp[j]));         // keep the pair as one 64-bit value: its halves are picked by op_sel
        }
        float v = wts[P::c_b3];
        for (int t = 0; t < H / 4; ++t) {
            float2v a0 = {wts[P::c_b2 + 4 * t], wts[P::c_b2 + 4 * t + 1]}, a1 = {wts[P::c_b2 + 4 * t + 2], wts[P::c_b2 + 4 * t + 3]};
            four_units<H>(wts + P::c_w2 + (size_t)t * H * 4, hp, a0, a1);
            v = fmaf(wts[P::c_w3 + 4 * t], fmaxf(a0.x, 0.0f), v);
            v = fmaf(wts[P::c_w3 + 4 * t + 1], fmaxf(a0.y, 0.0f), v);
            v = fmaf(wts[P::c_w3 + 4 * t + 2], fmaxf(a1.x, 0.0f), v);
            v = fmaf(wts[P::c_w3 + 4 * t + 3], fmaxf(a1.y, 0.0f), v);
        }
        if (valid) values[env] = v;
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
    const int actor_blocks = (int)(((long long)E * M + WAVE - 1) / WAVE);
    const int critic_blocks = io->values ? (E + WAVE - 1) / WAVE : 0;
    const dim3 grid(actor_blocks + critic_blocks);
#define MARL_LAUNCH_POLICY(HH)                                                                                     \
    hipLaunchKernelGGL(policy_act_kernel<HH>, grid, dim3(WAVE), 0, (hipStream_t)stream, io->weights, io->obs,       \
                       io->tick, io->actions, io->log_probs, io->values, (int)E, (int)M, v_max, log_std_min,        \
                       log_std_max, io->seed, io->env_offset, io->call, (int)io->deterministic, actor_blocks)
    switch (hidden) {
        case 64: MARL_LAUNCH_POLICY(64); break;
        case 128: MARL_LAUNCH_POLICY(128); break;
        default: return MARL_EUNSUPPORTED;
    }
#undef MARL_LAUNCH_POLICY
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

}  // extern "C"
