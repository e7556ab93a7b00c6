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
#include <type_traits>

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

// Row matrices are written tile by tile (32 rows), each tile as H / 4 chunks of [32 rows][4 units]: element
// (row r, unit u) sits at float (r / 32) 32 H + (u / 4) 128 + (r % 32) 4 + u % 4.  A lane owns four consecutive units
// per register quad, so one store instruction of a wave covers two runs of 512 contiguous bytes (the row-major
// form scatters it over 32 rows, 16 bytes each, which is what bounded the pass); the products kernel transposes
// a tile through LDS.  `tile_out` = the tile's base + this lane's (4 j + 128 hl) floats.
// y = relu(n * g + e), kept in y and written to the tile.
template <int NB>
__device__ __forceinline__ void affine_relu_store(const floatx16 (&n)[NB], const float* __restrict__ g, const float* __restrict__ e,
                                                  int hl, floatx16 (&y)[NB], float* __restrict__ tile_out) {
    const float* gb = g + opaque(4 * hl);
    const float* eb = e + opaque(4 * hl);
    float* ob = tile_out;
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
            *reinterpret_cast<float4*>(ob + (8 * b + 2 * q) * 128) = r;
        });
        LDS_FENCE();                  // per block: eight reads in flight together (the pass has registers to spare)
    });
}

// The lane's units into the tile, 16 bytes at a time.
template <int NB>
__device__ __forceinline__ void store_units(const floatx16 (&x)[NB], float* __restrict__ tile_out) {
    float* ob = tile_out;
    asm volatile("" : "+v"(ob));
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            *reinterpret_cast<float4*>(ob + (8 * b + 2 * q) * 128) =
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
// The value (0..15) whose total reduce_scatter16 leaves in lane j, and the unit that is for accumulator block b of half hl.
__device__ __forceinline__ int scattered_index(int j) { return 8 * (j & 1) + 4 * ((j >> 1) & 1) + 2 * ((j >> 2) & 1) + ((j >> 3) & 1); }
__device__ __forceinline__ int scattered_unit(int b, int j, int hl) { return unit_of(b, scattered_index(j), hl); }

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
    });                               // (no fence between the blocks: their exchange chains interleave)
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
// Threads per workgroup.  The actor's tile at H = 128 wants more than the 256 registers a wave gets at two waves per
// SIMD (131 spilled; every reload then waits for the tile's stores, loads and stores sharing one in-order
// counter): with ONE wave per SIMD and 386 registers it is faster (0.31 against 0.35 ms), the second wave's overlap
// notwithstanding.  (Three waves per SIMD for the critic's pass, which would fit its registers, gain nothing.)
template <int H, bool ACTOR> constexpr int ROWS_THREADS = (ACTOR && H > 64) ? 256 : PPO_THREADS;
template <int H, bool ACTOR>
__global__ __launch_bounds__((ROWS_THREADS<H, ACTOR>)) void ppo_rows_kernel(MarlPpoIO io, int B) {
    constexpr int THREADS = ROWS_THREADS<H, ACTOR>, WAVES = THREADS / WAVE;
    using P = TrainPack<H>;
    constexpr int NB = P::NB, NS = P::NS;
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
            {   // [obs | 1 | 0 0], shared with the critic's products; every row of the tile is written (the arrays hold
                // whole tiles; rows beyond the minibatch carry zero gradients below)
                float4* xe = reinterpret_cast<float4*>(io.xe + (size_t)p * 8);
                xe[hl] = hl ? make_float4(o4, 1.0f, 0.0f, 0.0f) : make_float4(o0, o1, o2, o3);
            }
            const size_t tile_off = (size_t)tile * 32 * H + 4 * j + 128 * hl;         // this lane's place in a tile
            floatx16 n2[NB], y[NB];
            {   // layer 1; its normalised pre-activations are worked out again where the backward pass needs them
                // (12 matrix instructions) instead of being carried through the whole tile in 64 registers
                floatx16 n1[NB];
                first_layer<NB>(lds + P::a_f1, lane, x0, x1, x2, n1);
                normalise<NB>(n1);
                affine_relu_store<NB>(n1, lds + P::a_g1, lds + P::a_e1, hl, y, io.y1e + tile_off);
            }
            unit_vector<NB>(lds + P::a_b2, hl, n2);
            hidden_layer<NB>(lds + P::a_f2, lane, y, n2);
            const float rstd2 = normalise<NB>(n2);
            affine_relu_store<NB>(n2, lds + P::a_g2, lds + P::a_e2, hl, y, io.y2e + tile_off);
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
            {
                float4* out = reinterpret_cast<float4*>(io.dh + (size_t)p * 8);
                out[hl] = hl ? make_float4(dh[4], dh[5], dh[6], dh[7]) : make_float4(dh[0], dh[1], dh[2], dh[3]);
                if (valid && hl == 0) {
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
            store_units<NB>(d, io.dz2 + tile_off);
            // dL/dy1 = W2^T dz2
            static_for<0, NB>([&](auto bb) { y[decltype(bb)::value] = floatx16{0}; });
            hidden_layer<NB>(lds + P::a_f2T, lane, d, y);
            first_layer<NB>(lds + P::a_f1, lane, x0, x1, x2, n2);          // n2 is free again: layer 1's pre-activations
            const float rstd1 = normalise<NB>(n2);
            layer_norm_backward<NB>(y, n2, rstd1, lds + P::a_g1, lds + P::a_e1, hl, j, acc_e1, acc_g1);
            store_units<NB>(y, io.dz1 + tile_off);
        }
        if ((j & 16) == 0) {                  // this wave's sums over its tiles, see MarlPpoIO.actor_sums
            float* out = io.actor_sums + (size_t)(blockIdx.x * WAVES + wave) * 4 * H;
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
            const size_t tile_off = (size_t)tile * 32 * H + 4 * j + 128 * hl;
            store_units<NB>(y1, io.cy1e + tile_off);
            unit_vector<NB>(cb2, hl, y2);
            hidden_layer<NB>(cf2, lane, y1, y2);
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                y2[decltype(bb)::value][decltype(vv)::value] = fmaxf(y2[decltype(bb)::value][decltype(vv)::value], 0.0f); }); });
            store_units<NB>(y2, io.cy2e + tile_off);
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
            if (hl == 0) io.dv[p] = dv;
            if (valid && hl == 0) s_vloss = fmaf(io.c1 * err, err, s_vloss);

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
            store_units<NB>(y2, io.cdz2 + tile_off);

            floatx16 dy[NB];
            static_for<0, NB>([&](auto bb) { dy[decltype(bb)::value] = floatx16{0}; });
            hidden_layer<NB>(cf2T, lane, y2, dy);
            first_layer<NB>(cf1, lane, x0, x1, x2, y1);                     // layer 1's pre-activations again, for the ReLU mask
            static_for<0, NB>([&](auto bb) { static_for<0, 16>([&](auto vv) {
                constexpr int b = decltype(bb)::value, q = decltype(vv)::value;
                dy[b][q] = y1[b][q] > 0.0f ? dy[b][q] : 0.0f; }); });
            store_units<NB>(dy, io.cdz1 + tile_off);
        }
        s_vloss = wave_total(s_vloss);
        if (lane == 0 && io.stats) atomicAdd(io.stats + 1, (double)s_vloss);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Sums over rows: the six products (row matrix)^T (row matrix) that turn what the row pass wrote into parameter
// gradients, in ONE pass over the arrays.  A workgroup walks over tiles of 32 rows: all 512 threads bring a tile
// of the eight big arrays in with 16-byte loads (the tile-major layout makes them contiguous), one tile ahead in
// registers, and lay it down in LDS row by row; there the operand of a contraction over rows is a plain 4-byte
// read: step s takes rows 2 s and 2 s + 1, lane l the element of row 2 s + l / 32 in column 32 block + l % 32.
// One workgroup of four waves per network and CU (even workgroups: the actor's products, odd ones: the critic's;
// wave w: row block w of dz2^T y1 and of dz1^T xe, column block w of dh^T y2).  Two workgroups share a CU and run out
// of step with each other, so that one has its loads in flight while the other moves its tile into LDS.  Every pair
// leaves its partial 32 x 32 blocks in one slab of `partial`, summed by the caller (fixed order: deterministic).
template <int H>
struct ProductLayout {
    static constexpr int NBO = H / 32;
    static constexpr int g2 = 0;                          // [H][H]   dz2^T y1
    static constexpr int g1 = g2 + H * H;                 // [H][32]  dz1^T xe (columns 0..4 the weights, 5 the bias)
    static constexpr int gh = g1 + H * 32;                // [32][H]  dh^T y2 (rows 0..7; the critic's: row 0)
    static constexpr int gb = gh + 32 * H;                // [H][32]  dz2^T xe: column 5 (xe's ones) = the bias of layer 2
    static constexpr int ghb = gb + H * 32;               // [32][32] dh^T xe: column 5 = the head biases
    static constexpr int half = ghb + 32 * 32;            // the critic's five follow in the same order
    static constexpr int total = 2 * half;
    static constexpr int LD = H + 4;                      // LDS row pitch (floats)
    static constexpr int ARR = 32 * LD;
    static constexpr int lds_floats = 4 * ARR + 32 * 8 + 32 * 8;
};
constexpr int PRODUCT_THREADS = 256;

__device__ __forceinline__ void store_block(const floatx16& c, float* __restrict__ out, int ldo, int lane) {
    const int j = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int v = 0; v < 16; ++v) out[(size_t)(8 * (v / 4) + 4 * hl + v % 4) * ldo + j] = c[v];
}

template <int H>
__global__ __launch_bounds__(PRODUCT_THREADS) void ppo_products_kernel(MarlPpoIO io, int tiles, float* __restrict__ partial) {
    using L = ProductLayout<H>;
    constexpr int NBO = L::NBO, LD = L::LD, ARR = L::ARR, PER = H / 32;      // 16-byte pieces of a tile per thread and array
    extern __shared__ __align__(16) float lds[];
    float* lx = lds + 4 * ARR;
    float* ldh = lx + 32 * 8;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const bool critic = blockIdx.x & 1;
    const int pair = blockIdx.x >> 1, pairs = gridDim.x >> 1;
    const float* src[4] = {critic ? io.cdz2 : io.dz2, critic ? io.cy1e : io.y1e, critic ? io.cdz1 : io.dz1, critic ? io.cy2e : io.y2e};
    typedef float floatx4 __attribute__((ext_vector_type(4)));      // (HIP's float4 is a union type: arrays of it stay in scratch)
    floatx4 st[4][PER], sx = {0, 0, 0, 0};
    // (every index of st is a constant when the loops are expanded by static_for; a run-time loop index, even of a
    // fully unrollable loop, sends the array to scratch memory)
    auto fetch = [&](int tile) {
        static_for<0, 4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            const floatx4* p = reinterpret_cast<const floatx4*>(src[a] + (size_t)tile * 32 * H);
            static_for<0, PER>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                st[a][k] = p[tid + PRODUCT_THREADS * k];
            });
        });
        // the critic's only head gradient, dL/dV, takes the place of head 0 of eight (the others zero): one loop body
        // serves both networks
        if (tid < 64) sx = reinterpret_cast<const floatx4*>(io.xe + (size_t)tile * 32 * 8)[tid];
        else if (tid < 128) {
            if (!critic) sx = reinterpret_cast<const floatx4*>(io.dh + (size_t)tile * 32 * 8)[tid - 64];
            else if ((tid & 1) == 0) sx[0] = io.dv[(size_t)tile * 32 + ((tid - 64) >> 1)];
        }
    };
    auto lay_down = [&]() {
        static_for<0, 4>([&](auto aa) {
            constexpr int a = decltype(aa)::value;
            static_for<0, PER>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const int f = tid + PRODUCT_THREADS * k;         // piece f of the tile = chunk f / 32 (units 4 c ..), row f % 32
                *reinterpret_cast<floatx4*>(lds + a * ARR + (f & 31) * LD + 4 * (f >> 5)) = st[a][k];
            });
        });
        if (tid < 64) reinterpret_cast<floatx4*>(lx)[tid] = sx;
        else if (tid < 128) reinterpret_cast<floatx4*>(ldh)[tid - 64] = sx;
    };
    // The big product runs on the matrix cores; the narrow ones (6 columns of xe, 8 heads, the bias sums) are a few
    // multiply-adds per step on the vector pipe beside them: lane (unit i of block w, row parity) keeps its own
    // sums over the rows of its parity, the two parities meet at the end.
    floatx16 acc2[NBO];
    float s1[6] = {}, sh[8] = {}, sb = 0.0f, shb = 0.0f;
    static_for<0, NBO>([&](auto cc) { acc2[decltype(cc)::value] = floatx16{0}; });
    const float* A2 = lds;
    const float* Y1 = A2 + ARR;
    const float* A1 = A2 + 2 * ARR;
    const float* Y2 = A2 + 3 * ARR;
    const int col = 32 * w + i;
    const int i8 = i & 7;
    auto steps = [&]() {
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int row = 2 * s + kh;
            const float a2 = A2[row * LD + col], a1 = A1[row * LD + col], bh = Y2[row * LD + col];
            const float4 xa = *reinterpret_cast<const float4*>(lx + row * 8), xb = *reinterpret_cast<const float4*>(lx + row * 8 + 4);
            s1[0] = fmaf(a1, xa.x, s1[0]); s1[1] = fmaf(a1, xa.y, s1[1]); s1[2] = fmaf(a1, xa.z, s1[2]);
            s1[3] = fmaf(a1, xa.w, s1[3]); s1[4] = fmaf(a1, xb.x, s1[4]); s1[5] = fmaf(a1, xb.y, s1[5]);
            sb += a2;
            const float4 ha = *reinterpret_cast<const float4*>(ldh + row * 8), hb = *reinterpret_cast<const float4*>(ldh + row * 8 + 4);
            sh[0] = fmaf(ha.x, bh, sh[0]); sh[1] = fmaf(ha.y, bh, sh[1]); sh[2] = fmaf(ha.z, bh, sh[2]); sh[3] = fmaf(ha.w, bh, sh[3]);
            sh[4] = fmaf(hb.x, bh, sh[4]); sh[5] = fmaf(hb.y, bh, sh[5]); sh[6] = fmaf(hb.z, bh, sh[6]); sh[7] = fmaf(hb.w, bh, sh[7]);
            shb += ldh[row * 8 + i8];                        // lane i < 8: the sum of head i's gradients
            static_for<0, NBO>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                acc2[c] = mfma(a2, Y1[row * LD + 32 * c + i], acc2[c]);
            });
        }
    };
    int tile = pair;
    if (tile < tiles) fetch(tile);
    for (; tile < tiles; tile += pairs) {
        __syncthreads();                                   // the tile before has been consumed
        lay_down();
        __syncthreads();
#ifndef PRODUCT_SKIP_FETCH
        if (tile + pairs < tiles) fetch(tile + pairs);
#endif
#ifndef PRODUCT_SKIP_STEPS
        if (w < NBO) steps();
#endif
    }
    float* out = partial + (size_t)pair * L::total + (critic ? L::half : 0);
    if (w < NBO) {
        static_for<0, NBO>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            store_block(acc2[c], out + L::g2 + (size_t)32 * w * H + 32 * c, H, lane);
        });
        // the narrow products: both row parities together, then lane i (of the lower half) writes unit 32 w + i
#pragma unroll
        for (int k = 0; k < 6; ++k) s1[k] += __shfl_xor(s1[k], 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) sh[k] += __shfl_xor(sh[k], 32);
        sb += __shfl_xor(sb, 32);
        shb += __shfl_xor(shb, 32);
        if (kh == 0) {
            float* g1 = out + L::g1 + (size_t)col * 32;
#pragma unroll
            for (int k = 0; k < 6; ++k) g1[k] = s1[k];
#pragma unroll
            for (int k = 6; k < 32; ++k) g1[k] = 0.0f;
            float* gb = out + L::gb + (size_t)col * 32;
#pragma unroll
            for (int k = 0; k < 32; ++k) gb[k] = k == 5 ? sb : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) out[L::gh + (size_t)k * H + col] = sh[k];
            for (int k = 8; k < 32; ++k) out[L::gh + (size_t)k * H + col] = 0.0f;
            if (w == 0) {
                for (int k = 0; k < 32; ++k) out[L::ghb + i * 32 + k] = (k == 5 && i < (critic ? 1 : 8)) ? shb : 0.0f;
            }
        }
    }
}

// out[t] = the sum over the workgroup slabs of partial[.][index[t]] (index[t] >= 0) or over the wave rows of
// sums[.][-index[t] - 1]: the parameter gradients in the caller's order, in one launch and a fixed order of addition.
__global__ __launch_bounds__(1024) void ppo_reduce_kernel(int n_out, const int32_t* __restrict__ index,
                                                          const float* __restrict__ partial, int n_slabs, size_t slab_floats,
                                                          const float* __restrict__ sums, int n_rows, int row_floats,
                                                          float* __restrict__ out) {
    // 32 outputs per workgroup (neighbouring threads: neighbouring addresses), each summed by 32 threads that take
    // every 32nd slab / row, then one pass through LDS
    __shared__ float red[32][33];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int t = blockIdx.x * 32 + tx;
    float a0 = 0.0f, a1 = 0.0f;
    if (t < n_out) {
        const int k = index[t];
        const float* p = k >= 0 ? partial + k : sums + (-k - 1);
        const size_t stride = k >= 0 ? slab_floats : (size_t)row_floats;
        const int n = k >= 0 ? n_slabs : n_rows;
        int s = ty;
        for (; s + 32 < n; s += 64) { a0 += p[(size_t)s * stride]; a1 += p[(size_t)(s + 32) * stride]; }
        if (s < n) a0 += p[(size_t)s * stride];
    }
    red[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && t < n_out) {
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += red[r][tx];
        out[t] = v;
    }
}

// Gradient-norm clip and Adam for both networks on the flat vectors (36 k parameters: the work is a few
// microseconds; what it replaces is some thirty launches of a few microseconds each).  Element i < n_actor belongs
// to the actor.  Every workgroup sums the squares of ALL gradients for itself (no exchange between workgroups, and
// the same order of addition in each), then steps its own 4 096 elements.  Arithmetic of
// torch.nn.utils.clip_grad_norm_ and of torch.optim.Adam's default settings (no weight decay, no amsgrad).
constexpr int APPLY_THREADS = 1024;
constexpr int APPLY_MAX_SWEEPS = 16;         // 16 x 4 096 parameters at most

__global__ __launch_bounds__(APPLY_THREADS) void ppo_apply_kernel(MarlPpoApply a, int sweeps) {
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    __shared__ float part[2][APPLY_THREADS / WAVE];
    const int t = threadIdx.x;
    const int n = a.n_actor + a.n_critic, n4 = (n + 3) / 4;
    const floatx4* __restrict__ g4 = (const floatx4*)a.grads;
    float sa = 0.0f, sc = 0.0f;
    {
        floatx4 g[APPLY_MAX_SWEEPS];
#pragma unroll
        for (int k = 0; k < APPLY_MAX_SWEEPS; ++k)
            if (k < sweeps) { const int q = k * APPLY_THREADS + t; g[k] = g4[q < n4 ? q : n4 - 1]; }
#pragma unroll
        for (int k = 0; k < APPLY_MAX_SWEEPS; ++k)
            if (k < sweeps) {
                const int q = k * APPLY_THREADS + t;
                if (q < n4) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = g[k][c] * a.grad_scale;           // the padding behind element n - 1 holds zeros
                        if (4 * q + c < a.n_actor) sa += x * x; else sc += x * x;
                    }
                }
            }
    }
    sa = wave_total(sa);
    sc = wave_total(sc);
    if ((t & (WAVE - 1)) == 0) { part[0][t / WAVE] = sa; part[1][t / WAVE] = sc; }
    const double step_a = (double)a.steps[0] + 1.0, step_c = (double)a.steps[a.tensors_actor] + 1.0;
    const float lr_a = *a.lr_actor, lr_c = *a.lr_critic;
    __syncthreads();
    sa = 0.0f; sc = 0.0f;
#pragma unroll
    for (int w = 0; w < APPLY_THREADS / WAVE; ++w) { sa += part[0][w]; sc += part[1][w]; }
    const float clip_a = fminf(a.max_norm / (sqrtf(sa) + 1e-6f), 1.0f) * a.grad_scale;
    const float clip_c = fminf(a.max_norm / (sqrtf(sc) + 1e-6f), 1.0f) * a.grad_scale;
    // torch's kernel takes betas, eps and the learning rate as doubles and lets them promote every expression they
    // enter; the bias corrections and the square root are single precision there
    const float corr1_a = (float)(1.0 - pow(a.beta1, step_a)), corr1_c = (float)(1.0 - pow(a.beta1, step_c));
    const double size_a = (double)lr_a / (double)corr1_a, size_c = (double)lr_c / (double)corr1_c;
    const float root_a = sqrtf((float)(1.0 - pow(a.beta2, step_a))), root_c = sqrtf((float)(1.0 - pow(a.beta2, step_c)));
    const int q = blockIdx.x * APPLY_THREADS + t;
    if (q >= n4) return;
    floatx4* __restrict__ m4 = (floatx4*)a.exp_avg;
    floatx4* __restrict__ v4 = (floatx4*)a.exp_avg_sq;
    floatx4* __restrict__ p4 = (floatx4*)a.params;
    const floatx4 g = g4[q];
    floatx4 m = m4[q], v = v4[q], p = p4[q];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool actor = 4 * q + c < a.n_actor;
        const float x = g[c] * (actor ? clip_a : clip_c);
        m[c] = (float)((double)m[c] + (1.0 - a.beta1) * ((double)x - (double)m[c]));
        v[c] = (float)(a.beta2 * (double)v[c] + (1.0 - a.beta2) * (double)x * (double)x);
        const double denom = (double)(sqrtf(v[c]) / (actor ? root_a : root_c)) + a.eps;
        p[c] = (float)((double)p[c] - (actor ? size_a : size_c) * (double)m[c] / denom);
    }
    m4[q] = m;
    v4[q] = v;
    p4[q] = p;
}

// After the step: the counters (every workgroup of the step kernel reads them, so they are advanced here), and
// packed[j] = 0 where gather[j] == 0, else params[gather[j] - 1]: the weights in the order the row pass reads them.
__global__ __launch_bounds__(256) void ppo_pack_kernel(int n, const int32_t* __restrict__ gather, const float* __restrict__ params,
                                                       float* __restrict__ packed, float* __restrict__ steps, int n_steps) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n_steps) steps[j] += 1.0f;
    if (j < n) {
        const int k = gather[j];
        packed[j] = k > 0 ? params[k - 1] : 0.0f;
    }
}

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

// CUs of the current device: the caller sizes `partial` and `actor_sums` with the device current that it launches on
static int ppo_cus() { return marl_net::device_cus(256); }

int32_t marl_ppo_sum_rows(void) { return ppo_cus() * (PPO_THREADS / WAVE); }

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
    const int grid = 2 * ppo_cus();             // every workgroup writes its partial blocks (zeros if it has no tile)
    const int tiles = (B + 31) / 32;
    hipError_t err = hipSuccess;
#define MARL_LAUNCH_PRODUCTS(HH)                                                                                     \
    do {                                                                                                              \
        static bool attr_set[marl_net::kMaxDevices];                                                                  \
        const size_t lds_bytes = (size_t)ProductLayout<HH>::lds_floats * sizeof(float);                               \
        err = marl_net::allow_dynamic_lds((const void*)ppo_products_kernel<HH>, lds_bytes, attr_set);                 \
        if (err == hipSuccess)                                                                                        \
            hipLaunchKernelGGL(ppo_products_kernel<HH>, dim3(grid), dim3(PRODUCT_THREADS), lds_bytes, (hipStream_t)stream, \
                               *io, tiles, partial);                                                                  \
    } while (0)
    if (hidden == 64) MARL_LAUNCH_PRODUCTS(64); else MARL_LAUNCH_PRODUCTS(128);
#undef MARL_LAUNCH_PRODUCTS
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

int marl_ppo_reduce(int32_t n_out, const int32_t* index, int32_t hidden, const float* partial, const float* actor_sums,
                    float* out, void* stream) {
    if (n_out < 1 || index == nullptr || partial == nullptr || actor_sums == nullptr || out == nullptr) return MARL_EINVAL;
    const size_t slab = marl_ppo_partial_floats(hidden);
    if (slab == 0) return MARL_EUNSUPPORTED;
    hipLaunchKernelGGL(ppo_reduce_kernel, dim3((n_out + 31) / 32), dim3(32, 32), 0, (hipStream_t)stream, (int)n_out, index, partial,
                       (int)marl_ppo_partial_rows(), slab, actor_sums, (int)marl_ppo_sum_rows(), 4 * (int)hidden, out);
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess) return MARL_OK;
    marl_note_hip_error((int)err);
    return MARL_EHIP;
}

int marl_ppo_apply(const MarlPpoApply* a, void* stream) {
    if (a == nullptr || a->params == nullptr || a->grads == nullptr || a->exp_avg == nullptr || a->exp_avg_sq == nullptr ||
        a->steps == nullptr || a->lr_actor == nullptr || a->lr_critic == nullptr)
        return MARL_EINVAL;
    if (((uintptr_t)a->params | (uintptr_t)a->grads | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) & 15) return MARL_EINVAL;
    if (a->n_actor < 1 || a->n_critic < 1 || a->tensors_actor < 1 || a->tensors_critic < 1 ||
        a->tensors_actor + a->tensors_critic > APPLY_THREADS || a->n_packed < 0 ||
        (a->n_packed > 0 && (a->gather == nullptr || a->packed == nullptr)))
        return MARL_EINVAL;
    const int n4 = (a->n_actor + a->n_critic + 3) / 4;
    const int sweeps = (n4 + APPLY_THREADS - 1) / APPLY_THREADS;
    if (sweeps > APPLY_MAX_SWEEPS) return MARL_EUNSUPPORTED;
    hipLaunchKernelGGL(ppo_apply_kernel, dim3(sweeps), dim3(APPLY_THREADS), 0, (hipStream_t)stream, *a, sweeps);
    const int n_steps = a->tensors_actor + a->tensors_critic;
    const int most = a->n_packed > n_steps ? a->n_packed : n_steps;
    hipLaunchKernelGGL(ppo_pack_kernel, dim3((most + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)a->n_packed, a->gather,
                       (const float*)a->params, a->packed, a->steps, n_steps);
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
        io->y2e == nullptr || io->dz1 == nullptr || io->dz2 == nullptr || io->actor_sums == nullptr ||
        io->dh == nullptr || io->cy1e == nullptr ||
        io->cy2e == nullptr || io->cdz1 == nullptr || io->cdz2 == nullptr || io->dv == nullptr)
        return MARL_EINVAL;
    const int tiles = (B + 31) / 32;
    const int cus = ppo_cus();                             // one workgroup per CU; fewer for a small minibatch
    // waves that do not run leave their rows of the partial sums at zero
    hipError_t err = hipMemsetAsync(io->actor_sums, 0, (size_t)marl_ppo_sum_rows() * 4 * hidden * sizeof(float), (hipStream_t)stream);
#define MARL_LAUNCH_PPO(HH, ACTOR, FLOATS)                                                                           \
    do {                                                                                                              \
        static bool attr_set[marl_net::kMaxDevices];                                                                  \
        const size_t lds_bytes = (size_t)(FLOATS) * sizeof(float);                                                    \
        if (err == hipSuccess)                                                                                        \
            err = marl_net::allow_dynamic_lds((const void*)ppo_rows_kernel<HH, ACTOR>, lds_bytes, attr_set);          \
        const int waves = ROWS_THREADS<HH, ACTOR> / WAVE;                                                             \
        const int grid = tiles < cus * waves ? (tiles + waves - 1) / waves : cus;                                     \
        if (err == hipSuccess)                                                                                        \
            hipLaunchKernelGGL((ppo_rows_kernel<HH, ACTOR>), dim3(grid), dim3(ROWS_THREADS<HH, ACTOR>), lds_bytes,   \
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
