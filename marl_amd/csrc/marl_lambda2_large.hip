// lambda2 of graphs with 257 .. 1024 vertices: the size range in which the reference switches to its sparse
// solver (compute_lambda2, physics/communication_graph.py:219-286: eigsh(k=2, which='SM') for use_sparse, dense eigh
// otherwise -- the same number either way).  Not on the step's path (the env kernels stop at 256 enemies): a stage
// op, written for correctness.
//
// One workgroup of 1024 threads per graph.  Thread r owns ROW r of the Laplacian, kept in the caller's workspace
// column-major (element (r, c) at c ld + r: the threads of a wave touch neighbouring addresses, and no thread ever
// reads what another one wrote).  Householder tridiagonalisation from the last column down, then the second-smallest
// eigenvalue by 1024-way multisection on Sturm counts (the recurrence of LAPACK's dlaebz), five rounds: an interval
// of 2 N shrinks to 2 N / 1025^5 < 2e-12.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "marl_env.h"

namespace {

constexpr int LARGE_THREADS = 1024;
constexpr int LARGE_WAVES = LARGE_THREADS / 64;

struct LargeLds {
    double u[LARGE_THREADS];
    double q[LARGE_THREADS];
    double d[LARGE_THREADS];            // diagonal of the tridiagonal matrix
    double e[LARGE_THREADS];            // e[i] couples i - 1 and i; e[0] unused
    double red[LARGE_WAVES];
    double last[2];                      // x[m - 1] of the current step (slot m % 2: a thread that skips ahead on an
                                         // already tridiagonal column must not overwrite what a slower one still reads)
    int first;                           // multisection: first sample with two eigenvalues below it
};

// The sum of v over the workgroup, in every thread (same order of addition everywhere).  Every thread calls it.
__device__ __forceinline__ double block_sum(double v, LargeLds& L, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                     // the previous use of red has been read
    if ((tid & 63) == 0) L.red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < LARGE_WAVES; ++w) s += L.red[w];
    return s;
}

__device__ __forceinline__ double block_max(double v, LargeLds& L, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if ((tid & 63) == 0) L.red[tid >> 6] = v;
    __syncthreads();
    double s = L.red[0];
#pragma unroll
    for (int w = 1; w < LARGE_WAVES; ++w) s = fmax(s, L.red[w]);
    return s;
}

__global__ __launch_bounds__(LARGE_THREADS) void lambda2_large_kernel(int n, int ld, const uint32_t* __restrict__ bits,
                                                                      double* __restrict__ workspace, double* __restrict__ lambda2) {
    __shared__ LargeLds L;
    const int r = threadIdx.x;
    double* A = workspace + (size_t)blockIdx.x * ld * ld + r;            // row r: element c at A[c ld]
    const int W = (n + 31) / 32;
    // the Laplacian D - A of the graph (communication_graph.py:184-217), the diagonal bit ignored
    if (r < n) {
        const uint32_t* row = bits + ((size_t)blockIdx.x * n + r) * W;
        int deg = 0;
        for (int w = 0; w < W; ++w) {
            const uint32_t b = row[w];
            const int cols = n - 32 * w < 32 ? n - 32 * w : 32;
            for (int k = 0; k < cols; ++k) {
                const int c = 32 * w + k;
                const bool a = ((b >> k) & 1u) != 0 && c != r;
                A[(size_t)c * ld] = a ? -1.0 : 0.0;
                deg += a ? 1 : 0;
            }
        }
        A[(size_t)r * ld] = (double)deg;
    }
    // column m against rows 0 .. m-1: x; reflector u = x - alpha e_(m-1); p = A u / H; q = p - K u; A -= u q^T + q u^T
    for (int m = n - 1; m >= 1; --m) {
        const bool act = r < m;
        const double x = act ? A[(size_t)m * ld] : 0.0;
        if (r == m - 1) L.last[m & 1] = x;
        if (r == m) L.d[m] = A[(size_t)m * ld];
        const double off = block_sum(r < m - 1 ? x * x : 0.0, L, r);      // (its barriers publish L.last)
        const double xl = L.last[m & 1];
        if (off == 0.0) {                                                    // the column is already tridiagonal
            if (r == 0) L.e[m] = xl;
            continue;
        }
        const double sigma = off + xl * xl;
        const double alpha = xl >= 0.0 ? -sqrt(sigma) : sqrt(sigma);
        const double H = sigma - xl * alpha;
        const double ur = r == m - 1 ? xl - alpha : x;                      // zero for r >= m
        if (r == 0) L.e[m] = alpha;
        L.u[r] = act ? ur : 0.0;
        __syncthreads();
        double p = 0.0;
        if (act) {
#pragma unroll 8
            for (int c = 0; c < m; ++c) p = fma(A[(size_t)c * ld], L.u[c], p);
            p /= H;
        }
        const double K = block_sum(act ? ur * p : 0.0, L, r) / (2.0 * H);
        const double qr = p - K * ur;
        L.q[r] = act ? qr : 0.0;
        __syncthreads();
        if (act) {
#pragma unroll 8
            for (int c = 0; c < m; ++c) A[(size_t)c * ld] -= ur * L.q[c] + qr * L.u[c];
        }
        // (the next step's block_sum has a barrier before any thread overwrites u or q)
    }
    if (r == 0) { L.d[0] = A[0]; L.e[0] = 0.0; }
    __syncthreads();
    // bounds: every eigenvalue lies in [lo, hi) by Gershgorin; lambda2 is the smallest x with two eigenvalues <= x
    double g = -1.0, e2 = 0.0;
    if (r < n) {
        const double el = r > 0 ? fabs(L.e[r]) : 0.0, er = r + 1 < n ? fabs(L.e[r + 1]) : 0.0;
        g = L.d[r] + el + er;
        e2 = el * el;
    }
    double hi = block_max(g, L, r) + 1.0;
    double lo = -1.0;
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, block_max(e2, L, r));
    for (int round = 0; round < 5; ++round) {
        if (r == 0) L.first = LARGE_THREADS;
        __syncthreads();
        const double step = (hi - lo) / (double)(LARGE_THREADS + 1);
        const double x = lo + step * (double)(r + 1);
        double t = L.d[0] - x;
        if (fabs(t) < pivmin) t = -pivmin;
        int count = t <= 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            const double ei = L.e[i];
            t = L.d[i] - ei * ei / t - x;
            if (fabs(t) < pivmin) t = -pivmin;
            count += t <= 0.0 ? 1 : 0;
        }
        if (count >= 2) atomicMin(&L.first, r);
        __syncthreads();
        const int f = L.first;
        __syncthreads();                                                    // read before thread 0 resets it
        const double new_hi = f < LARGE_THREADS ? lo + step * (double)(f + 1) : hi;
        const double new_lo = f > 0 ? lo + step * (double)f : lo;
        hi = new_hi;
        lo = new_lo;
    }
    if (r == 0) lambda2[blockIdx.x] = 0.5 * (lo + hi);
}

}  // namespace

// Called by marl_lambda2_from_bits (marl_env.hip) for MARL_MAX_N < N <= MARL_MAX_N_LAMBDA2; workspace: E ld ld doubles.
extern "C" __attribute__((visibility("hidden"))) int marl_lambda2_large_ld(int n) { return (n + 63) & ~63; }

extern "C" __attribute__((visibility("hidden"))) int marl_lambda2_large_launch(int n, int E, const uint32_t* bits, double* lambda2,
                                                                              double* workspace, void* stream) {
    if (n <= MARL_MAX_N || n > MARL_MAX_N_LAMBDA2 || n > LARGE_THREADS || E < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lambda2_large_kernel, dim3(E), dim3(LARGE_THREADS), 0, (hipStream_t)stream, n, marl_lambda2_large_ld(n), bits,
                       workspace, lambda2);
    return (int)hipGetLastError();
}
