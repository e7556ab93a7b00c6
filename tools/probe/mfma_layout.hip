// Prints the register layout of v_mfma_f32_32x32x2f32 as this GPU executes it: which D[i][j] lands in
// accumulator register v of lane l, for A[i][k] in lane i + 32 k and B[k][j] in lane j + 32 k.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    const float a = (l < 32) ? (float)(l % 32 + 1) : 0.0f;           // A[i][0] = i + 1, A[i][1] = 0
    const float b = (l < 32) ? (float)(l % 32 + 1000) : 0.0f;        // B[0][j] = j + 1000
    floatx16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[l * 16 + v] = c[v];
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    probe<<<1, 64>>>(d);
    float h[64 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 16; ++v) {
        const int val = (int)h[l * 16 + v];
        int i = -1, j = -1;
        for (int jj = 0; jj < 32 && i < 0; ++jj) if (val % (jj + 1000) == 0 && val / (jj + 1000) >= 1 && val / (jj + 1000) <= 32) { j = jj; i = val / (jj + 1000) - 1; }
        const int wi = 8 * (v / 4) + 4 * (l / 32) + (v % 4), wj = l % 32;
        if (i != wi || j != wj) { ok = 0; if (l % 16 == 0) printf("lane %d v %d: D[%d][%d], expected D[%d][%d]\n", l, v, i, j, wi, wj); }
    }
    printf(ok ? "layout as assumed: D[8(v/4) + 4(l/32) + v%%4][l%%32]\n" : "layout differs\n");
    return 0;
}
