// Achievable fp32-MFMA rate and in-kernel clock on this device.
//   variant 0: pure v_mfma_f32_32x32x2_f32 loop (4 independent accumulators per wave, operands in registers)
//   variant 1: + per 16 MFMAs two 1-KiB global loads (L2-resident) and two ds_read_b128 feeding the NEXT group (prefetch)
//   variant 2: variant 1 with the loads consumed by the SAME group's MFMAs (no prefetch distance)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* clk, const float* gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 1.f; c2[r] = 2.f; c3[r] = 3.f; }
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0.5f, 0.25f, 0.125f, 1.f}, a1 = a0, b0 = a0, b1 = a0, na0 = a0, na1 = a0, nb0 = a0, nb1 = a0;
    const float* gp = gsrc + (size_t)(blockIdx.x & 63) * 4096 + lane * 4;
    const float* lp = lds + lane * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters && V < 3; ++i) {
        if (V >= 1) {
            const int o = (i & 7) * 512;
            na0 = *(const f32x4*)(gp + o); na1 = *(const f32x4*)(gp + o + 256);
            nb0 = *(const f32x4*)(lp + o); nb1 = *(const f32x4*)(lp + o + 256);
            if (V == 2) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], c3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (V == 1) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, 64 * 4096 * 4 + 8192 * 4, 0x00020000);
    const int voff = ((blockIdx.x & 63) * 4096 + lane * 4) * 4;
    if (V >= 3) {
        // static double buffer (a0/b0 <-> na0/nb0), one load issued in each of the first four MFMA gaps of a group
        for (int i = 0; i < iters; i += 2) {
#define GROUP(A0, A1, B0, B1, NA0, NA1, NB0, NB1, O)                                        \
            if (V == 3 || V == 5) { NA0 = *(const f32x4*)(gp + (O)); NA1 = *(const f32x4*)(gp + (O) + 256); }          \
            if (V == 7) { i32x4 t0_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (O) * 4, 0); i32x4 t1_ = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((O) + 256) * 4, 0); \
                          NA0 = __builtin_bit_cast(f32x4, t0_); NA1 = __builtin_bit_cast(f32x4, t1_); }                 \
            if (V == 3 || V == 4) { NB0 = *(const f32x4*)(lp + (O)); NB1 = *(const f32x4*)(lp + (O) + 256); }          \
            if (V == 6) { NA0 = NA0 * 1.0001f; NA1 = NA1 * 0.9999f; NB0 = NB0 + 1e-6f; NB1 = NB1 - 1e-6f; }             \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                  \
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s], B0[s], c0, 0, 0, 0);        \
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[s], B1[s], c1, 0, 0, 0);        \
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], B0[s], c2, 0, 0, 0);        \
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[s], B1[s], c3, 0, 0, 0);        \
            }                                                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            const int o = (i & 6) * 512;
            GROUP(a0, a1, b0, b1, na0, na1, nb0, nb1, o)
            GROUP(na0, na1, nb0, nb1, a0, a1, b0, b1, o + 512)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
void run(int wps, const float* gsrc) {
    const int iters = 40000;
    const int blocks = 256 * wps;   // 256-thread blocks = 1 wave per SIMD each
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<V>, dim3(blocks), dim3(256), 0, 0, out, clk, gsrc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(2 * blocks);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double ghz = (double)h[0] / (double)h[1] * 0.1;
        double flops = (double)blocks * 4 * iters * 16.0 * 4096.0;
        if (rep == 1) printf("variant %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz\n", V, wps, ms, flops / ms / 1e9, ghz);
    }
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    float* gsrc; (void)hipMalloc(&gsrc, 64 * 4096 * 4 + 8192 * 4); (void)hipMemset(gsrc, 0, 64 * 4096 * 4 + 8192 * 4);
    for (int wps = 1; wps <= 3; ++wps) { run<0>(wps, gsrc); run<5>(wps, gsrc); run<7>(wps, gsrc); }
    return 0;
}
