// Lane / register layout probe of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction) with exact integer data:
// hypothesis: lane l = 4*block + x holds A[block][i = x] and B[block][j = x]; D register r of lane l is D[block][i = r][j = x].
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe && tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    const float a = (float)(l + 1), b = (float)(100 * (l + 1));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4;
            const float want = (float)(4 * blk + r + 1) * (float)(100 * (l + 1));   // A of lane 4*blk + r times B of lane l
            if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, r, h[l * 4 + r], want); ++bad; }
        }
    printf("4x4x1_16b layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    if (bad) { for (int l = 0; l < 8; ++l) printf("lane %d: %.0f %.0f %.0f %.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    return 0;
}
