// sin^2(u) for SnakeBeta (transformer.py:71-78): the polynomial of ev_kernels.h (ev_sin2: ~17 vector instructions) against a form on the
// hardware cosine — Cody-Waite reduction by pi in two fma steps, then sin^2 r = (1 - cos 2r) / 2 with v_cos_f32 (argument in revolutions):
// ~8 issue slots.  Prints the maximum / rms absolute error of both against fp64 over |u| <= 8, 40, 1000.
//   hipcc --offload-arch=gfx950 -O3 tools/sin2_probe.hip -o tools/sin2_probe && tools/sin2_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float sin2_poly(float u) {          // = ev_sin2
    const float k = rintf(u * 0.63661977236758134f);
    float r = fmaf(k, -1.57079637050628662109375f, u);
    r = fmaf(k, 4.371138828673793e-8f, r);
    r = fmaf(k, 1.7763568394002505e-15f, r);
    const float z = r * r;
    const float q = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    const float sn = fmaf(q * z, r, r);
    const float s2 = sn * sn;
    return ((int)k & 1) ? 1.f - s2 : s2;
}
__device__ __forceinline__ float sin2_hw(float u) {
    const float k = rintf(u * 0.318309886183790672f);          // u / pi
    float r = fmaf(k, -3.1415927410125732421875f, u);           // pi = hi + lo (fp32 pieces)
    r = fmaf(k, 8.742277657347586e-8f, r);
    // sin^2 has period pi: sin^2 u = sin^2 r, r in [-pi/2, pi/2];  (1 - cos 2r) / 2, v_cos takes revolutions: 2 r / (2 pi) = r / pi
    const float c = __builtin_amdgcn_cosf(r * 0.318309886183790672f);
    return fmaf(c, -0.5f, 0.5f);
}
__global__ void k(const float* u, float* a, float* b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { a[i] = sin2_poly(u[i]); b[i] = sin2_hw(u[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> u(n), a(n), b(n);
    float *du, *da, *db;
    CHK(hipMalloc(&du, n * 4)); CHK(hipMalloc(&da, n * 4)); CHK(hipMalloc(&db, n * 4));
    for (float range : {8.f, 40.f, 1000.f}) {
        unsigned x = 777u;
        for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; u[i] = ((float)(x >> 8) / 8388608.0f - 1.0f) * range; }
        CHK(hipMemcpy(du, u.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, du, da, db, n);
        CHK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost));
        double ma = 0, mb = 0, ra = 0, rb = 0;
        for (int i = 0; i < n; ++i) {
            const double s = sin((double)u[i]), r = s * s;
            const double ea = fabs(a[i] - r), eb = fabs(b[i] - r);
            ma = fmax(ma, ea); mb = fmax(mb, eb); ra += ea * ea; rb += eb * eb;
        }
        printf("|u| <= %6.0f: polynomial (ev_sin2) max %.3e rms %.3e   |   v_cos_f32 form max %.3e rms %.3e\n", range, ma, sqrt(ra / n), mb, sqrt(rb / n));
    }
    return 0;
}
