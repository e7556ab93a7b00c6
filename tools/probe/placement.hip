// Where do the one-wave workgroups of a launch land?  Every block records the hardware id of its wave
// (XCD, shader engine, CU, SIMD, wave slot) and its start time, then stays resident for a while so that the
// launch fills the chip the way the eigen-solve launch does (126 registers: four waves per SIMD).
// Prints, for the first blocks, the SIMD each one went to, and how many distinct SIMDs / how many blocks per
// SIMD the first 1024 / 4096 blocks cover.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(64) void where(unsigned* ids, unsigned long long* t_start, int spin_ticks) {
    unsigned hw, xcc;
    unsigned long long t0, t;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(hw), "=s"(xcc), "=s"(t0));
    asm volatile("v_mov_b32 v125, 0" ::: "v125");          // 126 registers: the solve launch's occupancy
    for (int k = 0; k < 100000; ++k) {                       // bounded: every wave leaves
        asm volatile("s_sleep 8\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
        if ((long long)(t - t0) >= spin_ticks) break;
    }
    if (threadIdx.x == 0) {
        ids[2 * blockIdx.x] = hw;
        ids[2 * blockIdx.x + 1] = xcc;
        t_start[blockIdx.x] = t0;
    }
}
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 6000, spin = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned* ids; unsigned long long* ts;
    if (hipMalloc(&ids, nb * 8) != hipSuccess || hipMalloc(&ts, nb * 8) != hipSuccess) return 1;
    std::vector<unsigned> h(2 * nb); std::vector<unsigned long long> ht(nb);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(where, dim3(nb), dim3(64), 1024, 0, ids, ts, spin);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    if (hipMemcpy(h.data(), ids, nb * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(ht.data(), ts, nb * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (+ more above); XCC_ID 3:0
    auto simd_key = [&](int b) { return ((h[2 * b + 1] & 0xF) << 16) | (h[2 * b] & 0xFF30); };
    unsigned long long tmin = *std::min_element(ht.begin(), ht.end());
    printf("block: xcc se sh cu simd wave  t_start\n");
    for (int b = 0; b < 48 && b < nb; ++b) {
        unsigned w = h[2 * b];
        printf("%5d: %2u %u %u %2u %u %2u  %llu\n", b, h[2 * b + 1] & 0xF, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 0xF, (w >> 4) & 3, w & 0xF, ht[b] - tmin);
    }
    for (int upto : {256, 1024, 2048, 4096, nb}) {
        if (upto > nb) continue;
        std::map<unsigned, int> per;
        for (int b = 0; b < upto; ++b) per[simd_key(b)]++;
        int mx = 0, mn = 1 << 30;
        for (auto& kv : per) { mx = std::max(mx, kv.second); mn = std::min(mn, kv.second); }
        printf("first %5d blocks: %4zu SIMDs, %d..%d blocks per SIMD\n", upto, per.size(), mn, mx);
    }
    // does block b share a SIMD with b+1 (depth first), with b+8, with b+1024 ... ?
    for (int d : {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048}) {
        int same = 0, cnt = 0;
        for (int b = 0; b + d < std::min(nb, 4096); ++b) { same += simd_key(b) == simd_key(b + d); ++cnt; }
        printf("stride %5d: %5.1f %% of the first 4096 blocks share a SIMD\n", d, 100.0 * same / cnt);
    }
    // start order: how many blocks had started when block b started, for a few b (is the launch in index order?)
    std::vector<unsigned long long> sorted(ht);
    std::sort(sorted.begin(), sorted.end());
    printf("start times: block 1023 at %llu, 4095 at %llu, 4096 at %llu, last at %llu ticks; 4096th start overall at %llu\n",
           ht[std::min(nb - 1, 1023)] - tmin, ht[std::min(nb - 1, 4095)] - tmin, ht[std::min(nb - 1, 4096)] - tmin, ht[nb - 1] - tmin,
           sorted[std::min(nb - 1, 4095)] - tmin);
    // full dump for offline study
    FILE* f = fopen("gpurun_out/placement.txt", "w");
    if (f) {
        for (int b = 0; b < nb; ++b) fprintf(f, "%d %u %u %llu\n", b, h[2 * b], h[2 * b + 1], ht[b] - tmin);
        fclose(f);
    }
    return 0;
}
