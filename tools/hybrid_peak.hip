// Can the VALU add fp32 FLOPs on top of a saturated fp32 MFMA pipe?  Every SIMD of the chip runs NM waves that stream
// v_mfma_f32_32x32x2_f32 and NV waves that stream independent v_pk_fma_f32 chains (16 accumulator pairs per lane), for
// ~50 ms so that power management settles; reports both rates and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/hybrid_peak.hip -o tools/hybrid_peak && tools/hybrid_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void hybrid(float* out, unsigned long long* clk, int nm_waves, int iters_m, int iters_v) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (wave < nm_waves) {
        f32x16 c[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) c[k][r] = (float)k;
        float a = 0.5f + lane * 1e-3f, b = 0.25f;
        for (int i = 0; i < iters_m; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
    } else {
        f32x2 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x2{lane * 1e-3f + k, 1.f};
        f32x2 m = {1.0001f, 0.9999f}, h = {0.5f, 0.25f};
        for (int i = 0; i < iters_v; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(m), "v"(h));
        }
        for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { clk[(blockIdx.x * 16 + wave) * 2] = t1 - t0; clk[(blockIdx.x * 16 + wave) * 2 + 1] = r1 - r0; }
}

int main() {
    const int blocks = 256;
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)blocks * 1024 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16 * 16);
    struct Cfg { int nm, nv; } cfgs[] = {{4, 0}, {0, 4}, {4, 4}, {8, 0}, {8, 8}, {4, 8}, {0, 8}};
    for (auto cf : cfgs) {
        const int threads = (cf.nm + cf.nv) * 64;
        // size both loops for ~40 ms alone: MFMA 16 per iter x 64 cycles; pk_fma 32 per iter x ~4 cycles (per wave; waves on one SIMD share it)
        const int per_simd_m = cf.nm / 4 > 0 ? cf.nm / 4 : 1, per_simd_v = cf.nv / 4 > 0 ? cf.nv / 4 : 1;
        const int iters_m = 90000 / per_simd_m, iters_v = 700000 / per_simd_v;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(hybrid, dim3(blocks), dim3(threads), 0, 0, out, clk, cf.nm, iters_m, iters_v);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 16 * 2);
            (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            double tm = 0, tv = 0, ghz_m = 0, ghz_v = 0;   // average wave durations (100 MHz ticks -> s) per class, block 0..blocks-1
            int cm = 0, cv = 0;
            for (int b = 0; b < blocks; ++b)
                for (int w = 0; w < cf.nm + cf.nv; ++w) {
                    const double cyc = (double)h[(b * 16 + w) * 2], ref = (double)h[(b * 16 + w) * 2 + 1];
                    if (w < cf.nm) { tm += ref * 1e-8; ghz_m += cyc / ref * 0.1; ++cm; } else { tv += ref * 1e-8; ghz_v += cyc / ref * 0.1; ++cv; }
                }
            const double fl_m = cm ? (double)cm * iters_m * 16.0 * 4096.0 : 0, fl_v = cv ? (double)cv * iters_v * 32.0 * 64 * 4 : 0;
            const double rate_m = cm ? fl_m / (tm / cm) / 1e12 : 0, rate_v = cv ? fl_v / (tv / cv) / 1e12 : 0;
            printf("MFMA waves/CU %d, VALU waves/CU %d: kernel %.1f ms | MFMA %.1f TFLOP/s over %.1f ms (clk %.2f GHz) | pk_fma %.1f TFLOP/s over %.1f ms (clk %.2f GHz)\n",
                   cf.nm, cf.nv, ms, rate_m, cm ? tm / cm * 1e3 : 0, cm ? ghz_m / cm : 0, rate_v, cv ? tv / cv * 1e3 : 0, cv ? ghz_v / cv : 0);
        }
    }
    return 0;
}
