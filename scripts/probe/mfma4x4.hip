// Probe: operand/result lane maps of v_mfma_f32_4x4x1_16b_f32 (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, int mode) {
    const int l = threadIdx.x;
    float a = mode == 0 ? (float)(l + 1) : 1.0f;
    float b = mode == 1 ? (float)(l + 1) : 1.0f;
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[(mode * 4 + r) * 64 + l] = c[r];
}
int main() {
    float *d; hipMalloc(&d, 2 * 4 * 64 * 4);
    probe<<<1, 64>>>(d, 0); probe<<<1, 64>>>(d, 1);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; mode++) {
        printf("mode %d (%s source lane + 1):\n", mode, mode ? "B" : "A");
        for (int r = 0; r < 4; r++) { printf(" reg %d:", r); for (int l = 0; l < 64; l++) printf(" %g", h[(mode * 4 + r) * 64 + l]); printf("\n"); }
    }
    return 0;
}
