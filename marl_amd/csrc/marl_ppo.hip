// Row pass of the PPO minibatch step (PPOAgent.update, agents/ppo_agent.py:296-331): for every row of a
// minibatch the forward pass of actor and critic (agents/actor.py:204-247 evaluate_actions,
// agents/critic.py:124-134), the clipped-surrogate / entropy / value losses, and the backward pass down to
// the pre-activations of every layer.  What is left of the gradient is a sum over rows per parameter --
// products of two of the matrices this kernel writes ([B][.] row-major: activations with a column of ones,
// pre-activation gradients, LayerNorm terms) -- which the caller takes with plain GEMMs.
//
// Layout as in the policy kernel (marl_net.h): hidden units on the rows of a matrix-core product, 32 minibatch
// rows on its columns, a lane owns one minibatch row and half of its hidden units.  The backward products
// W^T dz use the fragments of the transposed matrices (packed next to the forward ones); the accumulators of
// one product are the B operand of the next in both directions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "marl_env.h"
#include "marl_net.h"

namespace {

using namespace marl_net;
constexpr int WAVE = 64;
constexpr int OBS = MARL_OBS_DIM;
constexpr int PPO_THREADS = 512;
constexpr int PPO_WAVES = PPO_THREADS / WAVE;
#ifndef PPO_STORE_ROWS
#define PPO_STORE_ROWS true      // false (experiments only): the row pass without its stores
#endif

// Offsets (in floats) of the packed training weights, see marl_ppo_weight_count in marl_env.h.
template <int H>
struct TrainPack {
    static constexpr int NB = H / 32, NS = H / 2;
    static constexpr int a_f1 = 0;                          // [NB][3][64]
    static constexpr int a_g1 = a_f1 + NB * 3 * 64;
    static constexpr int a_e1 = a_g1 + H;
    static constexpr int a_f2 = a_e1 + H;                   // [NB][NS/4][64][4]
    static constexpr int a_b2 = a_f2 + H * H;
    static constexpr int a_g2 = a_b2 + H;
    static constexpr int a_e2 = a_g2 + H;
    static constexpr int a_fh = a_e2 + H;                   // [NS/4][64][4]
    static constexpr int a_bh = a_fh + NS * 64;             // [8]
    static constexpr int a_fhT = a_bh + 8;                  // [NB][4][64]  heads transposed: rows = units, k = head
    static constexpr int a_f2T = a_fhT + NB * 4 * 64;       // [NB][NS/4][64][4] fragments of W2^T
    static constexpr int a_end = a_f2T + H * H;
    static constexpr int c_f1 = a_end;
    static constexpr int c_f2 = c_f1 + NB * 3 * 64;
    static constexpr int c_b2 = c_f2 + H * H;
    static constexpr int c_w3 = c_b2 + H;
    static constexpr int c_b3 = c_w3 + H;                   // [1] + 3 pad
    static constexpr int c_f2T = c_b3 + 4;
    static constexpr int total = c_f2T + H * H;
    static constexpr int lds_floats = a_end > total - c_f1 ? a_end : total - c_f1;
};

// z -> (z - mean) * rstd over the H units of the lane's row, in place; returns rstd (eps 1e-5, biased variance).
template <int NB>
__device__ __forceinline__ float normalise(floatx16 (&x)[NB]) {
    constexpr int H = NB * 32;
    float sum = 0.0f;
    static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) { sum += x[decltype(bb)::value][decltype(vv)::value]; }); });
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / H);
    float var = 0.0f;
    static_for<0, NB>([&](auto bb) {
        static_for<0, 16>([&](auto vv) {
            const float d = x[decltype(bb)::value][decltype(vv)::value] - mean;
            var = fmaf(d, d, var);
        });
    });
    var += __shfl_xor(var, 32);
    const float rstd = 1.0f / sqrtf(var * (1.0f / H) + 1e-5f);
    static_for<0, NB>([&](auto bb) {
        static_for<0, 16>([&](auto vv) {
            x[decltype(bb)::value][decltype(vv)::value] = (x[decltype(bb)::value][decltype(vv)::value] - mean) * rstd;
        });
    });
    return rstd;
}

// y = relu(n * g + e), written to row `out` (the lane's units, 16 bytes at a time) when `valid`.
template <int NB>
__device__ __forceinline__ void affine_relu_store(const floatx16 (&n)[NB], const float* __restrict__ g, const float* __restrict__ e,
                                                  int hl, floatx16 (&y)[NB], float* __restrict__ out, bool valid) {
    const float* gb = g + opaque(4 * hl);
    const float* eb = e + opaque(4 * hl);
    float* ob = out + 4 * hl;
    asm volatile("" : "+v"(ob));
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            constexpr int u = 32 * b + 8 * q;
            const float4 gq = *reinterpret_cast<const float4*>(gb + u);
            const float4 eq = *reinterpret_cast<const float4*>(eb + u);
            float4 r;
            r.x = fmaxf(fmaf(n[b][4 * q], gq.x, eq.x), 0.0f);
            r.y = fmaxf(fmaf(n[b][4 * q + 1], gq.y, eq.y), 0.0f);
            r.z = fmaxf(fmaf(n[b][4 * q + 2], gq.z, eq.z), 0.0f);
            r.w = fmaxf(fmaf(n[b][4 * q + 3], gq.w, eq.w), 0.0f);
            y[b][4 * q] = r.x; y[b][4 * q + 1] = r.y; y[b][4 * q + 2] = r.z; y[b][4 * q + 3] = r.w;
            if (valid) *reinterpret_cast<float4*>(ob + u) = r;
            LDS_FENCE();
        });
    });
}

// The lane's units of a row, 16 bytes at a time.
template <int NB>
__device__ __forceinline__ void store_units(const floatx16 (&x)[NB], int hl, float* __restrict__ out, bool valid) {
    if (!valid) return;
    float* ob = out + 4 * hl;
    asm volatile("" : "+v"(ob));
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            *reinterpret_cast<float4*>(ob + 32 * b + 8 * q) =
                make_float4(x[b][4 * q], x[b][4 * q + 1], x[b][4 * q + 2], x[b][4 * q + 3]);
        });
    });
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int PATTERN>
__device__ __forceinline__ float swizzle_f(float x) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), PATTERN));
}
// Sums, over the 32 lanes of a half wave, of 16 values per lane -- as a reduce-scatter: each exchange halves the
// values a lane still carries (lane ^ 1, ^ 2, ^ 4, ^ 8), so 15 exchanges do what 16 separate reductions would need
// 64 for.  Lane j ends with the total of value 8 (j & 1) + 4 (j >> 1 & 1) + 2 (j >> 2 & 1) + (j >> 3 & 1); lanes j
// and j ^ 16 hold the same total.
__device__ __forceinline__ float reduce_scatter16(const float (&x)[16], int j) {
    const bool b0 = j & 1, b1 = j & 2, b2 = j & 4, b3 = j & 8;
    float y[8], z[4], w[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = (b0 ? x[i + 8] : x[i]) + dpp_f<0xB1>(b0 ? x[i] : x[i + 8]);          // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = (b1 ? y[i + 4] : y[i]) + dpp_f<0x4E>(b1 ? y[i] : y[i + 4]);          // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < 2; ++i) w[i] = (b2 ? z[i + 2] : z[i]) + swizzle_f<0x101F>(b2 ? z[i] : z[i + 2]);    // lane ^ 4
    float r = (b3 ? w[1] : w[0]) + dpp_f<0x128>(b3 ? w[0] : w[1]);                                           // row_ror:8 = lane ^ 8
    return r + swizzle_f<0x401F>(r);                                                                         // lane ^ 16
}
// The unit whose total reduce_scatter16 leaves in lane j of half hl, for accumulator block b.
__device__ __forceinline__ int scattered_unit(int b, int j, int hl) {
    const int v = 8 * (j & 1) + 4 * ((j >> 1) & 1) + 2 * ((j >> 2) & 1) + ((j >> 3) & 1);
    return unit_of(b, v, hl);
}

// Backward through y = relu(n g + e), n = LayerNorm-normalised z.  In: d = dL/dy; out: d = dL/dz.  The two
// per-row terms of the LayerNorm parameter gradients, E = dL/d(n g + e) (summed over the rows: the gradient of
// e) and G = E n (gradient of g), are summed over the 32 rows of the tile and added to accE / accG, one unit per
// lane and block (scattered_unit); rows beyond the minibatch carry d = 0.
template <int NB>
__device__ __forceinline__ void layer_norm_backward(floatx16 (&d)[NB], const floatx16 (&n)[NB], float rstd,
                                                    const float* __restrict__ g, const float* __restrict__ e, int hl, int j,
                                                    float (&accE)[NB], float (&accG)[NB]) {
    constexpr int H = NB * 32;
    float m1 = 0.0f, m2 = 0.0f;
    const float* gb = g + opaque(4 * hl);
    const float* eb = e + opaque(4 * hl);
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        float E[16], G[16];
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            constexpr int u = 32 * b + 8 * q;
            const float4 gq = *reinterpret_cast<const float4*>(gb + u);
            const float4 eq = *reinterpret_cast<const float4*>(eb + u);
            const float gv[4] = {gq.x, gq.y, gq.z, gq.w}, ev[4] = {eq.x, eq.y, eq.z, eq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float nv = n[b][4 * q + i];
                const float on = fmaf(nv, gv[i], ev[i]) > 0.0f ? d[b][4 * q + i] : 0.0f;
                E[4 * q + i] = on;
                G[4 * q + i] = on * nv;
                const float dx = on * gv[i];
                d[b][4 * q + i] = dx;
                m1 += dx;
                m2 = fmaf(dx, nv, m2);
            }
        });
        accE[b] += reduce_scatter16(E, j);
        accG[b] += reduce_scatter16(G, j);
        LDS_FENCE();
    });
    m1 += __shfl_xor(m1, 32);
    m2 += __shfl_xor(m2, 32);
    m1 *= 1.0f / H;
    m2 *= 1.0f / H;
    static_for<0, NB>([&](auto bb) {
        static_for<0, 16>([&](auto vv) {
            constexpr int b = decltype(bb)::value, v = decltype(vv)::value;
            d[b][v] = rstd * (d[b][v] - m1 - n[b][v] * m2);
        });
    });
}

__device__ __forceinline__ float wave_total(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ACTOR selects the network: one launch each (two bodies in one kernel are more code than the instruction cache
// of a CU pair holds, and the workgroups of the lighter network would idle at the end).
// The critic's tile needs 170 registers: three waves per SIMD (768 threads), where the actor's fits two.
template <bool ACTOR> constexpr int ROWS_THREADS = ACTOR ? PPO_THREADS : 768;
template <int H, bool ACTOR>
__global__ __launch_bounds__(ROWS_THREADS<ACTOR>) void ppo_rows_kernel(MarlPpoIO io, int B) {
    constexpr int THREADS = ROWS_THREADS<ACTOR>, WAVES = THREADS / WAVE;
    using P = TrainPack<H>;
    constexpr int NB = P::NB, NS = P::NS, LD = H + 4;
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int hl = lane >> 5, j = lane & 31;
    constexpr bool actor = ACTOR;
    const int actor_blocks = ACTOR ? (int)gridDim.x : 0;
    {
        const float4* src = reinterpret_cast<const float4*>(io.weights + (actor ? 0 : P::c_f1));
        const int n4 = (actor ? P::a_end : P::total - P::c_f1) / 4;
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < n4; i += THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const int tiles = (B + 31) / 32;
    const float inv_b = 1.0f / (float)B;
    if constexpr (ACTOR) {
        float s_loss = 0.0f, s_ent = 0.0f, s_clip = 0.0f, s_kl = 0.0f;
        float acc_e1[NB] = {}, acc_g1[NB] = {}, acc_e2[NB] = {}, acc_g2[NB] = {};      // LayerNorm parameter gradients
        // The row's inputs are loaded one tile ahead: loads issued before a tile's stores are older than them, so waiting
        // for the loads does not wait for the stores (one counter covers both, in order).
        struct RowIn { float o0, o1, o2, o3, o4, a0, a1, a2, old, adv; };
        auto load_row = [&](int tile) {
            const int p = tile * 32 + j;
            const int pp = p < B ? p : B - 1;
            const size_t r = io.index ? (size_t)io.index[pp] : (size_t)pp;
            const float* o = io.obs + r * OBS;
            const float* a = io.actions + r * 3;
            return RowIn{o[0], o[1], o[2], o[3], o[4], a[0], a[1], a[2], io.old_log_probs[r], io.advantages[r]};
        };
        const int tile0 = blockIdx.x * WAVES + wave, tstride = actor_blocks * WAVES;
        RowIn nxt = load_row(tile0 < tiles ? tile0 : 0);
        for (int tile = tile0; tile < tiles; tile += tstride) {
            const int p = tile * 32 + j;
            const bool valid = PPO_STORE_ROWS && p < B;
            const RowIn in = nxt;
            if (tile + tstride < tiles) nxt = load_row(tile + tstride);
            const float o0 = in.o0, o1 = in.o1, o2 = in.o2, o3 = in.o3, o4 = in.o4;
            const float x0 = hl ? o1 : o0, x1 = hl ? o3 : o2, x2 = hl ? 1.0f : o4;
            if (valid) {                                     // [obs | 1 | 0 0], shared with the critic's products
                float4* xe = reinterpret_cast<float4*>(io.xe + (size_t)p * 8);
                xe[hl] = hl ? make_float4(o4, 1.0f, 0.0f, 0.0f) : make_float4(o0, o1, o2, o3);
            }
            floatx16 n2[NB], y[NB];
            {   // layer 1; its normalised pre-activations are worked out again where the backward pass needs them
                // (12 matrix instructions) instead of being carried through the whole tile in 64 registers
                floatx16 n1[NB];
                first_layer<NB>(lds + P::a_f1, lane, x0, x1, x2, n1);
                normalise<NB>(n1);
                affine_relu_store<NB>(n1, lds + P::a_g1, lds + P::a_e1, hl, y, io.y1e + (size_t)p * LD, valid);
            }
            if (valid && hl == 0) *reinterpret_cast<float4*>(io.y1e + (size_t)p * LD + H) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            unit_vector<NB>(lds + P::a_b2, hl, n2);
            hidden_layer<NB, false>(lds + P::a_f2, lane, y, n2);
            const float rstd2 = normalise<NB>(n2);
            affine_relu_store<NB>(n2, lds + P::a_g2, lds + P::a_e2, hl, y, io.y2e + (size_t)p * LD, valid);
            if (valid && hl == 0) *reinterpret_cast<float4*>(io.y2e + (size_t)p * LD + H) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            // heads
            floatx16 h0 = {0}, h1 = {0};
            {
                const float4 bh = *reinterpret_cast<const float4*>(lds + P::a_bh + 4 * hl);
                h0[0] = bh.x; h0[1] = bh.y; h0[2] = bh.z; h0[3] = bh.w;
            }
            const float* fh = lds + P::a_fh + opaque(lane * 4);
            static_for<0, NS / 4>([&](auto ss) {
                constexpr int s4 = decltype(ss)::value;
                const float4 w = *reinterpret_cast<const float4*>(fh + s4 * 256);
                h0 = mfma(w.x, y[(4 * s4) / 16][(4 * s4) % 16], h0);
                h1 = mfma(w.y, y[(4 * s4 + 1) / 16][(4 * s4 + 1) % 16], h1);
                h0 = mfma(w.z, y[(4 * s4 + 2) / 16][(4 * s4 + 2) % 16], h0);
                h1 = mfma(w.w, y[(4 * s4 + 3) / 16][(4 * s4 + 3) % 16], h1);
                LDS_FENCE();
            });
            float mine[4], other[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { mine[q] = h0[q] + h1[q]; other[q] = __shfl_xor(mine[q], 32); }
            const float mu0 = hl ? other[0] : mine[0], mu1 = hl ? other[1] : mine[1];
            const float rl0 = hl ? other[2] : mine[2], rl1 = hl ? other[3] : mine[3];
            float lg[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) lg[q] = hl ? mine[q] : other[q];
            // evaluate_actions (actor.py:204-247) and the losses of ppo_agent.py:296-316; every lane of the row's pair
            // computes the same values
            const float vlim = 2.0f * io.v_max;
            const float a0 = fminf(fmaxf(in.a0, -vlim), vlim), a1 = fminf(fmaxf(in.a1, -vlim), vlim);
            int band = (int)in.a2;
            band = band < 0 ? 0 : (band > 3 ? 3 : band);
            const float ls0 = fminf(fmaxf(rl0, io.log_std_min), io.log_std_max), ls1 = fminf(fmaxf(rl1, io.log_std_min), io.log_std_max);
            const bool in0 = rl0 >= io.log_std_min && rl0 <= io.log_std_max, in1 = rl1 >= io.log_std_min && rl1 <= io.log_std_max;
            const float is0 = expf(-ls0), is1 = expf(-ls1);
            const float z0 = (a0 - mu0) * is0, z1 = (a1 - mu1) * is1;
            const float lmax = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
            float ex[4], esum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { ex[q] = expf(lg[q] - lmax); esum += ex[q]; }
            const float lse = lmax + logf(esum), ies = 1.0f / esum;
            float pr[4], lpk[4], hcat = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { pr[q] = ex[q] * ies; lpk[q] = lg[q] - lse; hcat -= pr[q] * lpk[q]; }
            const float HALF_LOG_2PI = 0.91893853320467274f;
            const float lp = (-0.5f * z0 * z0 - ls0 - HALF_LOG_2PI) + (-0.5f * z1 * z1 - ls1 - HALF_LOG_2PI) + lpk[band];
            const float ent = (0.5f + HALF_LOG_2PI + ls0) + (0.5f + HALF_LOG_2PI + ls1) + hcat;
            const float dlt = lp - in.old;
            const float ratio = expf(dlt), adv = in.adv;
            const float lo = 1.0f - io.clip_eps, hi = 1.0f + io.clip_eps;
            const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, lo), hi) * adv;
            const bool inside = ratio >= lo && ratio <= hi;
            // d(-min(s1, s2))/d ratio: the unclipped branch when it is the smaller one; on a tie both branches carry
            // half, and the clipped one passes it on only inside the clip range
            const float through = s1 < s2 ? 1.0f : (s1 == s2 && inside ? 1.0f : 0.0f);
            const float glp = valid ? -adv * through * ratio * inv_b : 0.0f;           // dL/d log_prob
            const float gent = valid ? -io.c2 * inv_b : 0.0f;                            // dL/d entropy
            float dh[8];
            dh[0] = glp * z0 * is0;
            dh[1] = glp * z1 * is1;
            dh[2] = in0 ? glp * (z0 * z0 - 1.0f) + gent : 0.0f;
            dh[3] = in1 ? glp * (z1 * z1 - 1.0f) + gent : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) dh[4 + q] = glp * ((q == band ? 1.0f : 0.0f) - pr[q]) - gent * pr[q] * (lpk[q] + hcat);
            if (valid) {
                float4* out = reinterpret_cast<float4*>(io.dh + (size_t)p * 8);
                out[hl] = hl ? make_float4(dh[4], dh[5], dh[6], dh[7]) : make_float4(dh[0], dh[1], dh[2], dh[3]);
                if (hl == 0) {
                    s_loss -= fminf(s1, s2);
                    s_ent += ent;
                    s_clip += fabsf(ratio - 1.0f) > io.clip_eps ? 1.0f : 0.0f;
                    s_kl += (ratio - 1.0f) - dlt;
                }
            }
            // dL/dy2 = heads^T dh: contraction over the 8 heads, input 2 s + hl in step s
            floatx16 d[NB];
            const float* fhT = lds + P::a_fhT + opaque(lane);
            static_for<0, NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                floatx16 c = {0};
                c = mfma(fhT[(b * 4 + 0) * 64], hl ? dh[1] : dh[0], c);
                c = mfma(fhT[(b * 4 + 1) * 64], hl ? dh[3] : dh[2], c);
                c = mfma(fhT[(b * 4 + 2) * 64], hl ? dh[5] : dh[4], c);
                c = mfma(fhT[(b * 4 + 3) * 64], hl ? dh[7] : dh[6], c);
                d[b] = c;
            });
            layer_norm_backward<NB>(d, n2, rstd2, lds + P::a_g2, lds + P::a_e2, hl, j, acc_e2, acc_g2);
            store_units<NB>(d, hl, io.dz2 + (size_t)p * H, valid);
            // dL/dy1 = W2^T dz2
            static_for<0, NB>([&](auto bb) { y[decltype(bb)::value] = floatx16{0}; });
            hidden_layer<NB, false>(lds + P::a_f2T, lane, d, y);
            first_layer<NB>(lds + P::a_f1, lane, x0, x1, x2, n2);          // n2 is free again: layer 1's pre-activations
            const float rstd1 = normalise<NB>(n2);
            layer_norm_backward<NB>(y, n2, rstd1, lds + P::a_g1, lds + P::a_e1, hl, j, acc_e1, acc_g1);
            store_units<NB>(y, hl, io.dz1 + (size_t)p * H, valid);
        }
        if ((j & 16) == 0) {                  // this wave's sums over its tiles: (bias 1, weight 1, bias 2, weight 2) x H
            float* out = io.ln_sums + (size_t)(blockIdx.x * WAVES + wave) * 4 * H;
            static_for<0, NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                const int u = scattered_unit(b, j, hl);
                out[u] = acc_e1[b]; out[H + u] = acc_g1[b]; out[2 * H + u] = acc_e2[b]; out[3 * H + u] = acc_g2[b];
            });
        }
        s_loss = wave_total(s_loss); s_ent = wave_total(s_ent); s_clip = wave_total(s_clip); s_kl = wave_total(s_kl);
        if (lane == 0 && io.stats) {
            atomicAdd(io.stats + 0, (double)s_loss);
            atomicAdd(io.stats + 2, (double)s_ent);
            atomicAdd(io.stats + 3, (double)s_clip);
            atomicAdd(io.stats + 4, (double)s_kl);
        }
    } else {
        const int nblocks = gridDim.x;
        const float* cf1 = lds;
        const float* cf2 = lds + (P::c_f2 - P::c_f1);
        const float* cb2 = lds + (P::c_b2 - P::c_f1);
        const float* cw3 = lds + (P::c_w3 - P::c_f1);
        const float cb3 = lds[P::c_b3 - P::c_f1];
        const float* cf2T = lds + (P::c_f2T - P::c_f1);
        float s_vloss = 0.0f;
        struct RowIn { float x0, x1, x2, ret; };                 // loaded one tile ahead, as in the actor's pass
        auto load_row = [&](int tile) {
            const int p = tile * 32 + j;
            const int pp = p < B ? p : B - 1;
            const size_t r = io.index ? (size_t)io.index[pp] : (size_t)pp;
            const float* o = io.obs + r * OBS;
            return RowIn{o[hl], o[2 + hl], hl ? 1.0f : o[4], io.returns[r]};
        };
        const int tile0 = (int)blockIdx.x * WAVES + wave, tstride = nblocks * WAVES;
        RowIn nxt = load_row(tile0 < tiles ? tile0 : 0);
        for (int tile = tile0; tile < tiles; tile += tstride) {
            const int p = tile * 32 + j;
            const bool valid = PPO_STORE_ROWS && p < B;
            const RowIn in = nxt;
            if (tile + tstride < tiles) nxt = load_row(tile + tstride);
            const float x0 = in.x0, x1 = in.x1, x2 = in.x2;
            floatx16 y1[NB], y2[NB];
            first_layer<NB>(cf1, lane, x0, x1, x2, y1);
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                y1[decltype(bb)::value][decltype(vv)::value] = fmaxf(y1[decltype(bb)::value][decltype(vv)::value], 0.0f); }); });
            store_units<NB>(y1, hl, io.cy1e + (size_t)p * LD, valid);
            if (valid && hl == 0) *reinterpret_cast<float4*>(io.cy1e + (size_t)p * LD + H) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            unit_vector<NB>(cb2, hl, y2);
            hidden_layer<NB>(cf2, lane, y1, y2);
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                y2[decltype(bb)::value][decltype(vv)::value] = fmaxf(y2[decltype(bb)::value][decltype(vv)::value], 0.0f); }); });
            store_units<NB>(y2, hl, io.cy2e + (size_t)p * LD, valid);
            if (valid && hl == 0) *reinterpret_cast<float4*>(io.cy2e + (size_t)p * LD + H) = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
            float v = 0.0f;
            const float* w3b = cw3 + opaque(4 * hl);
            static_for<0, NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                static_for<0, 4>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    const float4 wq = *reinterpret_cast<const float4*>(w3b + 32 * b + 8 * q);
                    v = fmaf(wq.x, y2[b][4 * q], v);
                    v = fmaf(wq.y, y2[b][4 * q + 1], v);
                    v = fmaf(wq.z, y2[b][4 * q + 2], v);
                    v = fmaf(wq.w, y2[b][4 * q + 3], v);
                    LDS_FENCE();
                });
            });
            v += __shfl_xor(v, 32);
            v += cb3;
            const float err = v - in.ret;
            const float dv = valid ? 2.0f * io.c1 * err * inv_b : 0.0f;                 // d(c1 mean (v - ret)^2)/dv
            if (valid && hl == 0) {
                io.dv[p] = dv;
                s_vloss = fmaf(io.c1 * err, err, s_vloss);
            }
            // dz2 = w3 dv where y2 > 0; dy1 = W2^T dz2; dz1 = dy1 where y1 > 0
            static_for<0, NB>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                static_for<0, 4>([&](auto qq) {
                    constexpr int q = decltype(qq)::value;
                    const float4 wq = *reinterpret_cast<const float4*>(w3b + 32 * b + 8 * q);
                    y2[b][4 * q] = y2[b][4 * q] > 0.0f ? wq.x * dv : 0.0f;
                    y2[b][4 * q + 1] = y2[b][4 * q + 1] > 0.0f ? wq.y * dv : 0.0f;
                    y2[b][4 * q + 2] = y2[b][4 * q + 2] > 0.0f ? wq.z * dv : 0.0f;
                    y2[b][4 * q + 3] = y2[b][4 * q + 3] > 0.0f ? wq.w * dv : 0.0f;
                    LDS_FENCE();
                });
            });
            store_units<NB>(y2, hl, io.cdz2 + (size_t)p * H, valid);
            floatx16 dy[NB];
            static_for<0, NB>([&](auto bb) { dy[decltype(bb)::value] = floatx16{0}; });
            hidden_layer<NB>(cf2T, lane, y2, dy);
            first_layer<NB>(cf1, lane, x0, x1, x2, y1);                     // layer 1's pre-activations again, for the ReLU mask
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                constexpr int b = decltype(bb)::value, q = decltype(vv)::value;
                dy[b][q] = y1[b][q] > 0.0f ? dy[b][q] : 0.0f; }); });
            store_units<NB>(dy, hl, io.cdz1 + (size_t)p * H, valid);
        }
        s_vloss = wave_total(s_vloss);
        if (lane == 0 && io.stats) atomicAdd(io.stats + 1, (double)s_vloss);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Sums over rows: the six products (row matrix)^T (row matrix) that turn what ppo_rows_kernel wrote into parameter
// gradients, in ONE pass over the arrays.  The arrays are row-major [B][.], which is the operand layout a
// contraction over rows wants: step s of a product takes rows 2 s and 2 s + 1, lane l the element of row
// 2 s + l / 32 in column 32 block + l % 32 -- two full 128-byte lines per load.  A workgroup owns a run of rows;
// waves 0..3 take the actor's products (wave w: row block w of dz2^T y1e and dz1^T xe, column block w of dh^T y2e),
// waves 4..7 the critic's; each workgroup leaves its partial 32 x 32 blocks in `partial`, summed by the caller.
template <int H>
struct ProductLayout {
    static constexpr int NBO = H / 32, NBJ = (H + 4 + 31) / 32, LDO = NBJ * 32;
    static constexpr int g2 = 0;                          // [H][LDO]   dz2^T y1e
    static constexpr int g1 = g2 + H * LDO;               // [H][32]    dz1^T xe
    static constexpr int gh = g1 + H * 32;                // [32][LDO]  dh^T y2e (rows 0..7)
    static constexpr int half = gh + 32 * LDO;            // the critic's three follow in the same order
    static constexpr int total = 2 * half;
};

__device__ __forceinline__ void store_block(const floatx16& c, float* __restrict__ out, int ldo, int lane) {
    const int j = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int v = 0; v < 16; ++v) out[(size_t)(8 * (v / 4) + 4 * hl + v % 4) * ldo + j] = c[v];
}

// One wave's share of the products, specialised on what the wave owns so that the loops carry no conditions:
// ROWS = row block w of the two hidden-layer products, FIRST / SECOND = column blocks w and w + 4 of the head product.
template <int H, bool ROWS, bool FIRST, bool SECOND>
__device__ __forceinline__ void products_body(const MarlPpoIO& io, int B, float* __restrict__ out, int w, bool critic, int lane,
                                              int s0, int s1) {
    using L = ProductLayout<H>;
    constexpr int NBJ = L::NBJ, LD = H + 4, U = 4;
    const int i = lane & 31, kh = lane >> 5;
    const float* dzb = critic ? io.cdz2 : io.dz2;          // [B][H]
    const float* dza = critic ? io.cdz1 : io.dz1;          // [B][H]
    const float* y1 = critic ? io.cy1e : io.y1e;           // [B][LD]
    const float* y2 = critic ? io.cy2e : io.y2e;           // [B][LD]
    const float* hd = critic ? io.dv : io.dh;              // critic: [B] (one head), actor: [B][8]
    const int heads = critic ? 1 : 8, hstride = critic ? 1 : 8;
    floatx16 acc2[NBJ], acc1 = {0}, acch0 = {0}, acch1 = {0};
    static_for<0, NBJ>([&](auto cc) { acc2[decltype(cc)::value] = floatx16{0}; });
    // Loads carry no branches: addresses are clamped into the arrays, and what lies beyond the minibatch or beyond a
    // row's columns is zeroed through the A operand or the column masks.
    const int hcol = i < heads ? i : 0, xcol = i < 8 ? i : 0;
    const float hmask = i < heads ? 1.0f : 0.0f, xmask = i < 8 ? 1.0f : 0.0f;
    int ycol[NBJ];
    float ymask[NBJ];
    static_for<0, NBJ>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        ycol[c] = 32 * c + i < LD ? 32 * c + i : LD - 1;
        ymask[c] = 32 * c + i < LD ? 1.0f : 0.0f;
    });
    const int hc0 = 32 * w + i, hc1 = 32 * (w + 4) + i;      // this wave's columns of the head product
    const int hcol0 = hc0 < LD ? hc0 : LD - 1, hcol1 = hc1 < LD ? hc1 : LD - 1;
    const float hm0 = hc0 < LD ? 1.0f : 0.0f, hm1 = hc1 < LD ? 1.0f : 0.0f;
    // Software pipeline: the loads of batch n + 1 (U steps) are in flight while the matrix instructions of batch n run
    // (two batches ahead does not fit the registers: 73 spills, 1.2 ms instead of 0.45).
    struct Batch { float ah[U], bh0[U], bh1[U], a2[U], a1[U], bx[U], by[U][NBJ]; };
    auto load = [&](int sb, Batch& t) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = 2 * (sb + u) + kh;
            const bool ok = sb + u < s1 && r < B;
            const size_t rr = ok ? (size_t)r : (size_t)(B - 1);
            const float live = ok ? 1.0f : 0.0f;
            if constexpr (FIRST || SECOND) t.ah[u] = hd[rr * hstride + hcol] * (hmask * live);
            if constexpr (FIRST) t.bh0[u] = y2[rr * LD + hcol0] * hm0;
            if constexpr (SECOND) t.bh1[u] = y2[rr * LD + hcol1] * hm1;
            if constexpr (ROWS) {
                t.a2[u] = dzb[rr * H + 32 * w + i] * live;
                t.a1[u] = dza[rr * H + 32 * w + i] * live;
                t.bx[u] = io.xe[rr * 8 + xcol] * xmask;
                static_for<0, NBJ>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    t.by[u][c] = y1[rr * LD + ycol[c]] * ymask[c];
                });
            }
        }
    };
    Batch cur, nxt;
    if (s0 < s1) load(s0, cur);
    for (int sb = s0; sb < s1; sb += U) {
        if (sb + U < s1) load(sb + U, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (FIRST) acch0 = mfma(cur.ah[u], cur.bh0[u], acch0);
            if constexpr (SECOND) acch1 = mfma(cur.ah[u], cur.bh1[u], acch1);
            if constexpr (ROWS) {
                acc1 = mfma(cur.a1[u], cur.bx[u], acc1);
                static_for<0, NBJ>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    acc2[c] = mfma(cur.a2[u], cur.by[u][c], acc2[c]);
                });
            }
        }
        cur = nxt;
    }
    if constexpr (ROWS) {
        static_for<0, NBJ>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            store_block(acc2[c], out + L::g2 + (size_t)32 * w * L::LDO + 32 * c, L::LDO, lane);
        });
        store_block(acc1, out + L::g1 + (size_t)32 * w * 32, 32, lane);
    }
    if constexpr (FIRST) store_block(acch0, out + L::gh + 32 * w, L::LDO, lane);
    if constexpr (SECOND) store_block(acch1, out + L::gh + 32 * (w + 4), L::LDO, lane);
}

template <int H>
__global__ __launch_bounds__(PPO_THREADS) void ppo_products_kernel(MarlPpoIO io, int B, float* __restrict__ partial) {
    using L = ProductLayout<H>;
    constexpr int NBO = L::NBO, NBJ = L::NBJ;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: scalar branches
    const bool critic = wave >= 4;
    const int w = wave & 3;
    const int steps = (B + 1) / 2;
    const int per = (steps + gridDim.x - 1) / gridDim.x;
    const int s0 = blockIdx.x * per, s1 = min(steps, s0 + per);
    float* out = partial + (size_t)blockIdx.x * L::total + (critic ? L::half : 0);
    const bool rows = w < NBO, first = w < NBJ, second = w + 4 < NBJ;
    if (rows && second) products_body<H, true, true, true>(io, B, out, w, critic, lane, s0, s1);
    else if (rows) products_body<H, true, true, false>(io, B, out, w, critic, lane, s0, s1);      // NBO <= NBJ: rows imply first
    else if (first) products_body<H, false, true, false>(io, B, out, w, critic, lane, s0, s1);
}

int g_ppo_cus = 0;

}  // namespace

extern "C" __attribute__((visibility("hidden"))) void marl_note_hip_error(int err);     // marl_env.hip

extern "C" {

size_t marl_ppo_weight_count(int32_t hidden) {
    switch (hidden) {
        case 64: return TrainPack<64>::total;
        case 128: return TrainPack<128>::total;
        default: return 0;
    }
}

static int ppo_cus() {
    if (g_ppo_cus == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2)
            cus = 256;
        g_ppo_cus = cus;
    }
    return g_ppo_cus;
}

int32_t marl_ppo_ln_rows(void) { return ppo_cus() * PPO_WAVES; }

int32_t marl_ppo_partial_rows(void) { return ppo_cus(); }

size_t marl_ppo_partial_floats(int32_t hidden) {
    switch (hidden) {
        case 64: return ProductLayout<64>::total;
        case 128: return ProductLayout<128>::total;
        default: return 0;
    }
}

int marl_ppo_products(int32_t B, int32_t hidden, const MarlPpoIO* io, float* partial, void* stream) {
    if (io == nullptr || partial == nullptr || B < 1) return MARL_EINVAL;
    if (hidden != 64 && hidden != 128) return MARL_EUNSUPPORTED;
    if (io->xe == nullptr || io->y1e == nullptr || io->y2e == nullptr || io->dz1 == nullptr || io->dz2 == nullptr ||
        io->dh == nullptr || io->cy1e == nullptr || io->cy2e == nullptr || io->cdz1 == nullptr || io->cdz2 == nullptr ||
        io->dv == nullptr)
        return MARL_EINVAL;
    const int grid = ppo_cus();                 // every workgroup writes its partial blocks (zeros if it has no rows)
    if (hidden == 64)
        hipLaunchKernelGGL(ppo_products_kernel<64>, dim3(grid), dim3(PPO_THREADS), 0, (hipStream_t)stream, *io, (int)B, partial);
    else
        hipLaunchKernelGGL(ppo_products_kernel<128>, dim3(grid), dim3(PPO_THREADS), 0, (hipStream_t)stream, *io, (int)B, partial);
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

int marl_ppo_rows(int32_t B, int32_t hidden, const MarlPpoIO* io, void* stream) {
    if (io == nullptr || B < 1) return MARL_EINVAL;
    if (hidden != 64 && hidden != 128) return MARL_EUNSUPPORTED;
    if (io->weights == nullptr || io->obs == nullptr || io->actions == nullptr || io->old_log_probs == nullptr ||
        io->advantages == nullptr || io->returns == nullptr || io->xe == nullptr || io->y1e == nullptr ||
        io->y2e == nullptr || io->dz1 == nullptr || io->dz2 == nullptr || io->ln_sums == nullptr ||
        io->dh == nullptr || io->cy1e == nullptr ||
        io->cy2e == nullptr || io->cdz1 == nullptr || io->cdz2 == nullptr || io->dv == nullptr)
        return MARL_EINVAL;
    const int tiles = (B + 31) / 32;
    const int cus = ppo_cus();                             // one workgroup per CU; fewer for a small minibatch
    // waves that do not run leave their rows of the partial sums at zero
    hipError_t err = hipMemsetAsync(io->ln_sums, 0, (size_t)marl_ppo_ln_rows() * 4 * hidden * sizeof(float), (hipStream_t)stream);
#define MARL_LAUNCH_PPO(HH, ACTOR, FLOATS)                                                                           \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        const size_t lds_bytes = (size_t)(FLOATS) * sizeof(float);                                                    \
        if (err == hipSuccess && !attr_set) {                                                                         \
            err = hipFuncSetAttribute((const void*)ppo_rows_kernel<HH, ACTOR>,                                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                    \
            attr_set = err == hipSuccess;                                                                             \
        }                                                                                                             \
        const int waves = ROWS_THREADS<ACTOR> / WAVE;                                                                 \
        const int grid = tiles < cus * waves ? (tiles + waves - 1) / waves : cus;                                     \
        if (err == hipSuccess)                                                                                        \
            hipLaunchKernelGGL((ppo_rows_kernel<HH, ACTOR>), dim3(grid), dim3(ROWS_THREADS<ACTOR>), lds_bytes,       \
                               (hipStream_t)stream, *io, (int)B);                                                     \
    } while (0)
    if (hidden == 64) {
        MARL_LAUNCH_PPO(64, true, TrainPack<64>::a_end);
        MARL_LAUNCH_PPO(64, false, TrainPack<64>::total - TrainPack<64>::c_f1);
    } else {
        MARL_LAUNCH_PPO(128, true, TrainPack<128>::a_end);
        MARL_LAUNCH_PPO(128, false, TrainPack<128>::total - TrainPack<128>::c_f1);
    }
#undef MARL_LAUNCH_PPO
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

}  // extern "C"
