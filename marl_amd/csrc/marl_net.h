// Device helpers shared by the policy-inference and PPO-gradient kernels: dense layers of the 5 -> H -> H
// networks (agents/actor.py:56-116, agents/critic.py:53-122) on the float32 matrix cores
// (v_mfma_f32_32x32x2), hidden units on the rows of the product and 32 network inputs (agents, minibatch rows)
// on its columns.  A lane owns one column (lane % 32) and the units unit_of(b, v, lane / 32) of it: register v of
// accumulator block b.  Matrices are read as pre-permuted A-operand fragments (include/marl_env.h).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace marl_net {

// Host-side per-device caches.  A process may drive several GPUs: the CU count and the "large dynamic LDS"
// function attribute belong to the device that is current at the call, not to the process.
constexpr int kMaxDevices = 64;
inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}
// compute units of the current device (`fallback` when the query fails)
inline int device_cus(int fallback) {
    static int cached[kMaxDevices];
    const int dev = current_device_slot();
    if (dev < 0) return fallback;
    int c = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = fallback;
        __atomic_store_n(&cached[dev], c, __ATOMIC_RELAXED);
    }
    return c;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device: `done` is the kernel's own table
inline hipError_t allow_dynamic_lds(const void* kernel, size_t bytes, bool (&done)[kMaxDevices]) {
    const int dev = current_device_slot();
    if (dev >= 0 && __atomic_load_n(&done[dev], __ATOMIC_RELAXED)) return hipSuccess;
    const hipError_t err = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err == hipSuccess && dev >= 0) __atomic_store_n(&done[dev], true, __ATOMIC_RELAXED);
    return err;
}

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Keeps the compiler from hoisting every LDS read of an unrolled loop to its top (which costs more registers than
// a wave has): reads are not moved across it.
#define LDS_FENCE() asm volatile("" ::: "memory")

// An index the compiler must take as it is: the LDS reads that follow use it with constant offsets (one address
// register, immediate offsets) instead of one precomputed, loop-invariant -- and then spilled -- address each.
// (An index, not a pointer: a pointer that went through here would lose its address space.)
__device__ __forceinline__ int opaque(int i) {
    asm volatile("" : "+v"(i));
    return i;
}

// The unit a lane of half `hl` holds in register v of accumulator block b.
__device__ __forceinline__ constexpr int unit_of(int b, int v, int hl) { return 32 * b + 8 * (v / 4) + 4 * hl + (v % 4); }

// Per-unit vector (bias, LayerNorm weight ...) gathered into the lane's register order: 16-byte reads of the four
// consecutive units that registers 4q .. 4q+3 of a block hold.
template <int NB>
__device__ __forceinline__ void unit_vector(const float* __restrict__ lds_vec, int hl, floatx16 (&out)[NB]) {
    const float* base = lds_vec + opaque(4 * hl);
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 t = *reinterpret_cast<const float4*>(base + 32 * b + 8 * q);
            out[b][4 * q] = t.x; out[b][4 * q + 1] = t.y; out[b][4 * q + 2] = t.z; out[b][4 * q + 3] = t.w;
        });
    });
}

// x[unit] over the H units of the lane's agent -> relu(LayerNorm(x) * g + e) (eps 1e-5, biased variance), in place.
template <int NB>
__device__ __forceinline__ void layer_norm_relu(floatx16 (&x)[NB], const float* __restrict__ g, const float* __restrict__ e, int hl) {
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
    const float* gb = g + opaque(4 * hl);
    const float* eb = e + opaque(4 * hl);
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const float4 gq = *reinterpret_cast<const float4*>(gb + 32 * b + 8 * q);
            const float4 eq = *reinterpret_cast<const float4*>(eb + 32 * b + 8 * q);
            x[b][4 * q] = fmaxf(fmaf((x[b][4 * q] - mean) * rstd, gq.x, eq.x), 0.0f);
            x[b][4 * q + 1] = fmaxf(fmaf((x[b][4 * q + 1] - mean) * rstd, gq.y, eq.y), 0.0f);
            x[b][4 * q + 2] = fmaxf(fmaf((x[b][4 * q + 2] - mean) * rstd, gq.z, eq.z), 0.0f);
            x[b][4 * q + 3] = fmaxf(fmaf((x[b][4 * q + 3] - mean) * rstd, gq.w, eq.w), 0.0f);
            LDS_FENCE();
        });
    });
}

// Layer 1: D[unit][agent] = W1[unit][0..4] . obs + b1[unit] (the bias rides as input 5 = 1).  x0..x2 = the lane's
// B operands of the three steps (inputs 2s + hl).
template <int NB>
__device__ __forceinline__ void first_layer(const float* __restrict__ f1, int lane, float x0, float x1, float x2, floatx16 (&acc)[NB]) {
    const float* base = f1 + opaque(lane);
    static_for<0, NB>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        floatx16 c = {0};
        c = mfma(base[(b * 3 + 0) * 64], x0, c);
        c = mfma(base[(b * 3 + 1) * 64], x1, c);
        c = mfma(base[(b * 3 + 2) * 64], x2, c);
        acc[b] = c;
    });
}

// Layer 2: acc[bo] (initialised with the bias) += W2 . y, y = the lane's accumulators of the layer before.
// PREFETCH = false: a group's weights are read right before its matrix instructions (4 NB registers less; the
// other wave of the SIMD covers the LDS latency).
template <int NB, bool PREFETCH = true>
__device__ __forceinline__ void hidden_layer(const float* __restrict__ f2, int lane, const floatx16 (&y)[NB], floatx16 (&acc)[NB]) {
    constexpr int NS = NB * 16;
    const float* base = f2 + opaque(lane * 4);
    if constexpr (!PREFETCH) {
        static_for<0, NS / 4>([&](auto ss) {
            constexpr int s4 = decltype(ss)::value;
            float4 w[NB];
            static_for<0, NB>([&](auto bb) {
                constexpr int bo = decltype(bb)::value;
                w[bo] = *reinterpret_cast<const float4*>(base + (bo * (NS / 4) + s4) * 256);
            });
            static_for<0, 4>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                constexpr int s = 4 * s4 + q;
                const float yk = y[s / 16][s % 16];
                static_for<0, NB>([&](auto bb) {
                    constexpr int bo = decltype(bb)::value;
                    const float wk = q == 0 ? w[bo].x : (q == 1 ? w[bo].y : (q == 2 ? w[bo].z : w[bo].w));
                    acc[bo] = mfma(wk, yk, acc[bo]);
                });
            });
            LDS_FENCE();
        });
        return;
    }
    float4 w[NB], wn[NB];
    static_for<0, NB>([&](auto bb) {
        constexpr int bo = decltype(bb)::value;
        w[bo] = *reinterpret_cast<const float4*>(base + (bo * (NS / 4)) * 256);
    });
    static_for<0, NS / 4>([&](auto ss) {
        constexpr int s4 = decltype(ss)::value;
        if constexpr (s4 + 1 < NS / 4) {                         // the next group's weights travel while this group computes
            static_for<0, NB>([&](auto bb) {
                constexpr int bo = decltype(bb)::value;
                wn[bo] = *reinterpret_cast<const float4*>(base + (bo * (NS / 4) + s4 + 1) * 256);
            });
        }
        LDS_FENCE();
        static_for<0, 4>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            constexpr int s = 4 * s4 + q;
            const float yk = y[s / 16][s % 16];
            static_for<0, NB>([&](auto bb) {
                constexpr int bo = decltype(bb)::value;
                const float wk = q == 0 ? w[bo].x : (q == 1 ? w[bo].y : (q == 2 ? w[bo].z : w[bo].w));
                acc[bo] = mfma(wk, yk, acc[bo]);
            });
        });
        if constexpr (s4 + 1 < NS / 4) static_for<0, NB>([&](auto bb) { w[decltype(bb)::value] = wn[decltype(bb)::value]; });
    });
}


}  // namespace marl_net
