// Prints what the HIP runtime says about the residency of the step kernels (workgroups per CU).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Iinclude tools/occupancy.hip -o build_ab/occupancy
#include "../marl_amd/csrc/marl_env.hip"
#include <stdio.h>
int main() {
    int nb = 0;
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("CUs %d, LDS per CU %zu, regs per CU %d, max threads per CU %d\n", pr.multiProcessorCount,
           (size_t)pr.maxSharedMemoryPerMultiProcessor, pr.regsPerMultiprocessor, pr.maxThreadsPerMultiProcessor);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, env_step_front_kernel<0>, WAVE, front_lds_bytes());
    printf("front kernel: %d workgroups per CU (LDS %zu B)\n", nb, front_lds_bytes());
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, env_step_solve_kernel<56>, WAVE, 0);
    printf("solve kernel<56>: %d workgroups per CU\n", nb);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, env_step_front_kernel<56>, WAVE, sizeof(WaveLds));
    printf("fused kernel<56>: %d workgroups per CU (LDS %zu B)\n", nb, sizeof(WaveLds));
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, env_step_team_kernel<104, 4>, TEAM_THREADS, 0);
    printf("team kernel<104,4>: %d workgroups per CU\n", nb);
    return 0;
}
