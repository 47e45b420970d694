// Does the MFMA SHAPE matter under this chip's power limit?  MI355X_MICROARCH.md ("DVFS give-back", item 7) reports bare bf16 loops on random
// data delivering ~1.15x the FLOP/s with v_mfma_f32_16x16x32 than with 32x32x16 at equal cycles per FLOP (the chip holds a higher clock).
// The vocoder's fp16 builds (conv_h16_kernel, resblock_pair_h16_kernel) run 32x32x16 at the power limit; this probe prices the alternative
// before any kernel is rewritten:
//   pure : one wave tile of 64 x 64 outputs per wave, operands in registers (4 accumulators 32x32 / 16 accumulators 16x16)
//   fed  : the h16 K loop — per 32-deep slab the wave reads two fp16 pieces of a 64-row A fragment from L2 (buffer loads) and of a 64-column
//          B fragment from LDS, then issues the 3 products per (m, n) tile (12 MFMAs 32x32x16 x 2 k-steps, or 48 MFMAs 16x16x32)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe && tools/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned half_pair(unsigned h) {   // two fp16 values of unit-scale magnitude (exponent 10..17 of 31, random sign / mantissa)
    const unsigned a = ((h & 1u) << 15) | ((10u + ((h >> 1) & 7u)) << 10) | ((h >> 4) & 1023u);
    const unsigned b = (((h >> 14) & 1u) << 15) | ((10u + ((h >> 15) & 7u)) << 10) | ((h >> 18) & 1023u);
    return a | (b << 16);
}

template <int SHAPE, int FED>   // SHAPE 32: 32x32x16, 16: 16x16x32
__global__ __launch_bounds__(256) void loop_kernel(const unsigned* W, float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) { unsigned h = (unsigned)(i + 1) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lds[i] = half_pair(h); }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    unsigned woff = (unsigned)(blockIdx.x & 7) * 65536u + (unsigned)lane * 16u;
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        f16x8 a[2][2][2], b[2][2][2];                      // [k-step][piece][tile]
        for (int k = 0; k < 2; ++k) for (int p = 0; p < 2; ++p) for (int t = 0; t < 2; ++t) {
            a[k][p][t] = *(const f16x8*)(lds + ((k * 4 + p * 2 + t) * 256 + lane * 4)); b[k][p][t] = *(const f16x8*)(lds + 2048 + ((k * 4 + p * 2 + t) * 256 + lane * 4));
        }
        for (int it = 0; it < iters; ++it) {
            if (FED) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, (woff + (unsigned)((k * 4 + p * 2 + t) * 1024)) & 0xfffffu, 0, 0);
                            a[k][p][t] = *(f16x8*)&v;
                            b[k][p][t] = *(const f16x8*)(lds + ((it * 52 + (k * 4 + p * 2) * 8 + t * 13 * 32 + (lane & 31) * 17 + (lane >> 5) * 4) & 2047) * 4);
                        }
                woff += 8192u;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k][pr == 1][m], b[k][pr == 0][n], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    } else {
        f32x4 acc[4][4];
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
        f16x8 a[2][4], b[2][4];                            // [piece][tile], one 32-deep k-step
        for (int p = 0; p < 2; ++p) for (int t = 0; t < 4; ++t) {
            a[p][t] = *(const f16x8*)(lds + ((p * 4 + t) * 256 + lane * 4)); b[p][t] = *(const f16x8*)(lds + 2048 + ((p * 4 + t) * 256 + lane * 4));
        }
        for (int it = 0; it < iters; ++it) {
            if (FED) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, (woff + (unsigned)((p * 4 + t) * 1024)) & 0xfffffu, 0, 0);
                        a[p][t] = *(f16x8*)&v;
                        b[p][t] = *(const f16x8*)(lds + ((it * 52 + p * 16 + t * 13 * 16 + (lane & 15) * 17 + (lane >> 4) * 4) & 2047) * 4);
                    }
                woff += 8192u;
            }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pr == 1][m], b[pr == 0][n], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    }
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// layout check with exact data: lane l supplies A[row l % 16][k 8 (l / 16) .. + 7] and B[k 8 (l / 16) .. + 7][col l % 16]; D[row 4 (l / 16) + i][col l % 16] comes back in register i
__global__ void layout_kernel(const float* A /*16 x 32*/, const float* B /*32 x 16*/, float* D /*16 x 16*/) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[r * 32 + 8 * g + e]; b[e] = (_Float16)B[(8 * g + e) * 16 + r]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

template <int SHAPE, int FED> static int run(const char* name, const unsigned* W, float* out, int wpc) {
    const int iters = 20000, grid = 256 * wpc;
    static unsigned long long* clk = nullptr;
    if (!clk) CHK(hipHostMalloc((void**)&clk, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop_kernel<SHAPE, FED>), dim3(grid), dim3(256), 0, 0, W, out, iters, clk);
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((loop_kernel<SHAPE, FED>), dim3(grid), dim3(256), 0, 0, W, out, iters, clk);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double fl = (double)grid * 4 * iters * 24.0 * 32768.0;     // 24 MFMAs of 32x32x16 (= 48 of 16x16x32) per iteration and wave
    printf("  %-44s %d wave(s)/SIMD: %8.3f ms  executed %7.1f TFLOP/s = %6.1f TFLOP/s of fp32 work at 3 products   clock %.3f GHz\n", name, wpc, best, fl / best * 1e-9,
           fl / 3.0 / best * 1e-9, (double)clk[0] / (double)clk[1] / 10.0);
    return 0;
}

int main() {
    unsigned* W; float* out;
    CHK(hipMalloc(&W, 2 << 20)); CHK(hipMalloc(&out, 64));
    std::vector<unsigned> hw((2 << 20) / 4);
    unsigned x = 12345u;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; unsigned h = x; h ^= h >> 15; h *= 2246822519u;
                         v = (((h & 1u) << 15) | ((10u + ((h >> 1) & 7u)) << 10) | ((h >> 4) & 1023u)) | (((((h >> 14) & 1u) << 15) | ((10u + ((h >> 15) & 7u)) << 10) | ((h >> 18) & 1023u)) << 16); }
    CHK(hipMemcpy(W, hw.data(), 2 << 20, hipMemcpyHostToDevice));
    {
        std::vector<float> A(16 * 32), B(32 * 16), D(256);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 3 + k * 5) % 7 - 3);
        for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 2 + j * 3) % 5 - 2);
        float *dA, *dB, *dD; CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dD, D.size() * 4));
        CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CHK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float r = 0; for (int k = 0; k < 32; ++k) r += A[i * 32 + k] * B[k * 16 + j]; bad += (r != D[i * 16 + j]); }
        printf("v_mfma_f32_16x16x32_f16 layout hypothesis (A row l%%16, k 8(l/16)..; D row 4(l/16)+i, col l%%16): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    }
    printf("fp16 MFMA shapes on random unit-scale data, 256 CUs, 20000 iterations per wave, 64 x 64 outputs per wave, 3 products of two pieces:\n");
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) {
            if (run<32, 0>("v_mfma_f32_32x32x16_f16, operands in registers", W, out, w)) return 1;
            if (run<16, 0>("v_mfma_f32_16x16x32_f16, operands in registers", W, out, w)) return 1;
            if (run<32, 1>("v_mfma_f32_32x32x16_f16, LDS + L2 fed", W, out, w)) return 1;
            if (run<16, 1>("v_mfma_f32_16x16x32_f16, LDS + L2 fed", W, out, w)) return 1;
        }
    return 0;
}
