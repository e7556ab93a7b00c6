// Issue rate of v_fmac_f64 with and without a DPP row broadcast as a function of the number of independent
// accumulators, one and two waves per SIMD.  Output buffer: 256 blocks x 512 threads x 8 bytes = 1 MiB exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DPP, int ACC>
__global__ void rate(double* out, long long* cyc, int iters) {
    double a[16], m = out[threadIdx.x & 63], s = out[(threadIdx.x + 1) & 63];
    for (int k = 0; k < 16; ++k) a[k] = k;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (DPP == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k % ACC]) : "v"(s), "v"(m));
            if (DPP == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[k % ACC]) : "v"(s), "v"(m));
        }
    }
    long long t1 = clock64();
    double r = 0;
    for (int k = 0; k < 16; ++k) r += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int DPP, int ACC>
void run(double* out, long long* cyc, int threads) {
    const int iters = 2048;
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate<DPP, ACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return; }
    }
    printf("%s acc %2d waves/SIMD %d: %.3f ticks/instr/wave\n", DPP ? "v_fmac_f64_dpp" : "v_fmac_f64    ", ACC, threads / 256,
           (double)h / (iters * 16));
}
int main() {
    double* out; long long* cyc;
    if (hipMalloc(&out, 1 << 20) != hipSuccess || hipMemset(out, 0, 1 << 20) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    for (int threads : {256, 512}) {
        run<0, 1>(out, cyc, threads); run<0, 2>(out, cyc, threads); run<0, 4>(out, cyc, threads); run<0, 8>(out, cyc, threads); run<0, 16>(out, cyc, threads);
        run<1, 1>(out, cyc, threads); run<1, 2>(out, cyc, threads); run<1, 4>(out, cyc, threads); run<1, 8>(out, cyc, threads); run<1, 16>(out, cyc, threads);
    }
    return 0;
}
