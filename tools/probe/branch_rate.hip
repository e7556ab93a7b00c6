// Cost of a never-taken wave-uniform exit test after every G multiply-adds in a straight line of 64
// v_fmac_f64_dpp (2 accumulators), two waves per SIMD.  Output buffer: 256 x 512 x 8 bytes = 1 MiB exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int G, int C>
struct Line {
    static __device__ __forceinline__ void go(double& p0, double& p1, const double& s, const double (&a)[64], int lim) {
        if constexpr (C < 64) {
            if constexpr (C > 0 && G > 0 && C % G == 0) {
                if (C > lim) return;
            }
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"((C & 1) ? p1 : p0) : "v"(s), "v"(a[C]));
            Line<G, C + 1>::go(p0, p1, s, a, lim);
        }
    }
};
template <int G>
__global__ __launch_bounds__(512) void rate(double* out, long long* cyc, int iters, int lim) {
    double a[64], s = out[(threadIdx.x + 1) & 63], p0 = 0, p1 = 0;
    for (int k = 0; k < 64; ++k) a[k] = out[(threadIdx.x + k) & 63] + k;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        Line<G, 0>::go(p0, p1, s, a, lim);
        asm volatile("" : "+v"(p0), "+v"(p1));
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p0 + p1;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int G>
void run(double* out, long long* cyc) {
    const int iters = 512;
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate<G>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1000);
        if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    }
    printf("exit test every %2d: %.3f ticks per multiply-add\n", G, (double)h / (iters * 64));
}
int main() {
    double* out; long long* cyc;
    if (hipMalloc(&out, 1 << 20) != hipSuccess || hipMemset(out, 0, 1 << 20) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    run<0>(out, cyc); run<16>(out, cyc); run<8>(out, cyc); run<4>(out, cyc); run<2>(out, cyc);
    return 0;
}
