// bf16_split_probe: what the bf16 matrix pipe of gfx950 offers an fp32 contraction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bf16_split_probe tools/bf16_split_probe.hip && tools/bf16_split_probe
// 1. operand layout of v_mfma_f32_32x32x16_bf16, checked with exact small-integer data: lane l holds A[i = l % 32][k = 8 (l / 32) + 0..7]
//    and B[k = 8 (l / 32) + 0..7][j = l % 32] (8 bf16 = 16 bytes each); D as for the fp32 32 x 32 tiles (register 4 g + e of lane l is
//    D[i = 8 g + 4 (l / 32) + e][j = l % 32]).
// 2. accuracy of an fp32 dot product computed as 3, 6 or 9 bf16 x bf16 products per element pair (every fp32 value is the exact sum of three
//    bf16 pieces; the products are exact in the fp32 accumulator), against fp64, beside the fp32 MFMA and a plain fp32 FMA chain.
// 3. rates: the fp32 MFMA loop, the pure bf16 MFMA loop, and a loop fed like a conv K loop (per 16-deep slab of a 64 x 64 wave tile: six
//    16-byte LDS reads, six 16-byte buffer loads from an L2-resident array, 24 MFMAs = the 6-product split).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ---- 1 + 2: one 32 x 32 x K product per wave from row-major fp32 A[32][K], B[K][32]; terms = 1 (bf16 hi only), 3, 6, 9, or 0 = fp32 MFMA
__device__ __forceinline__ unsigned trunc_hi(float x) { return __float_as_uint(x) & 0xffff0000u; }
__device__ __forceinline__ void split3(float x, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = trunc_hi(x); const float r1 = x - __uint_as_float(p0);
    p1 = trunc_hi(r1); const float r2 = r1 - __uint_as_float(p1);
    p2 = trunc_hi(r2);
}
__global__ void prod_kernel(const float* A, const float* B, float* D, int K, int terms) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    f32x16 acc = {0};
    if (terms == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + lh], B[(k + lh) * 32 + li], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            union { bf16x8 v; unsigned short s[8]; } a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                unsigned p0, p1, p2;
                split3(A[li * K + k0 + 8 * lh + e], p0, p1, p2);
                a[0].s[e] = p0 >> 16; a[1].s[e] = p1 >> 16; a[2].s[e] = p2 >> 16;
                split3(B[(k0 + 8 * lh + e) * 32 + li], p0, p1, p2);
                b[0].s[e] = p0 >> 16; b[1].s[e] = p1 >> 16; b[2].s[e] = p2 >> 16;
            }
            // smallest terms first
            if (terms >= 9) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[2].v, acc, 0, 0, 0);
                              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[2].v, acc, 0, 0, 0);
                              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[1].v, acc, 0, 0, 0); }
            if (terms >= 6) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
                              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
                              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0); }
            if (terms >= 3) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
                              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
        }
    }
    for (int g = 0; g < 4; ++g) for (int e = 0; e < 4; ++e) D[(8 * g + 4 * lh + e) * 32 + li] = acc[4 * g + e];
}

// ---- 3: rates
template <int MODE>   // 0: fp32 MFMA, 1: pure bf16 MFMA, 2: fed bf16 loop (LDS + buffer loads), 3: fed loop with bigger wave tile (2 x 4)
__global__ __launch_bounds__(256) void rate_kernel(const float* W, float* out, int iters, int data, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // data 0: small constants; 1: what the split pieces of N(0,1) activations look like (two bf16 per word: random sign, exponent 120..127,
    // random 7-bit mantissa); 2: random bits
    for (int i = tid; i < 8192; i += 256) {
        unsigned hsh = (unsigned)i * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        unsigned lo = ((hsh & 1) << 15) | ((120 + ((hsh >> 1) & 7)) << 7) | ((hsh >> 4) & 127);
        unsigned hi = (((hsh >> 11) & 1) << 15) | ((120 + ((hsh >> 12) & 7)) << 7) | ((hsh >> 15) & 127);
        smem[i] = data == 0 ? (float)(i & 7) * 0.125f : data == 1 ? __uint_as_float(lo | (hi << 16)) : __uint_as_float(hsh & 0x7f7f7f7fu);
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (MODE == 0) {
        float a = data ? __uint_as_float(__float_as_uint(smem[lane]) << 16) : smem[lane], b = data ? __uint_as_float(__float_as_uint(smem[lane + 64]) & 0xffff0000u) : smem[lane + 64];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    } else if (MODE == 1) {
        bf16x8 a = *(const bf16x8*)(smem + 4 * lane), b = *(const bf16x8*)(smem + 4 * lane + 256);
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    } else {
        constexpr int TN = MODE == 2 ? 2 : 4;
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
        unsigned woff = (unsigned)(blockIdx.x & 7) * 65536u + (unsigned)lane * 16u;
        for (int it = 0; it < iters; ++it) {
            bf16x8 a[3][2], b[3][TN];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, (woff + (unsigned)(p * 2 + m) * 1024u) & 0xfffffu, 0, 0);
                    a[p][m] = *(bf16x8*)&v;
                }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int n = 0; n < TN; ++n) b[p][n] = *(const bf16x8*)(smem + ((it * 52 + p * 4 + n * 13 * 32 + (lane & 31) * 13 + (lane >> 5)) & 2047) * 4);
            woff += 6144u;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    f32x16& c = acc[m * TN + n];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[2][n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], b[0][n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[1][n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[1][n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[0][n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[0][n], c, 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int MODE> static void rate(const char* name, const float* W, float* out, int wgs_per_cu, double mfma_per_iter, double flop_per_mfma, double alg_div, int data) {
    const int iters = 20000, grid = 256 * wgs_per_cu;
    static unsigned long long* clk = nullptr;
    if (!clk) CHK(hipHostMalloc((void**)&clk, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 32768, 0, W, out, iters, data, clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 32768, 0, W, out, iters, data, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)grid * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("  %-46s data %d, %d wave(s)/SIMD: %7.3f ms  executed %7.1f TFLOP/s  = %6.1f TFLOP/s of fp32-equivalent work   (s_memtime / s_memrealtime = %.3f)\n", name, data, wgs_per_cu, ms,
           fl / ms * 1e-9, fl / alg_div / ms * 1e-9, (double)clk[0] / (double)clk[1]);
}

int main() {
    // ---- 1: layout with exact data
    {
        const int K = 16;
        std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) A[i * K + k] = (float)((i * 3 + k * 5) % 7 - 3);
        for (int k = 0; k < K; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 2 + j * 3) % 5 - 2);
        float *dA, *dB, *dD; CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dD, D.size() * 4));
        CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(prod_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1);
        CHK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float r = 0; for (int k = 0; k < K; ++k) r += A[i * K + k] * B[k * 32 + j]; bad += (r != D[i * 32 + j]); }
        printf("32x32x16_bf16 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    // ---- 2: accuracy
    {
        const int K = 1408;    // 128 channels x 11 taps
        std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
        srand(7);
        auto rnd = []() { float u = 0; for (int i = 0; i < 6; ++i) u += (float)rand() / RAND_MAX; return (u - 3.0f); };
        for (auto& v : A) v = rnd() * 0.05f;
        for (auto& v : B) v = rnd() * 1.3f;
        std::vector<double> ref(32 * 32);
        double scale = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double r = 0; for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * B[k * 32 + j]; ref[i * 32 + j] = r; scale += r * r; }
        scale = sqrt(scale / 1024);
        std::vector<float> chain(32 * 32);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float r = 0; for (int k = 0; k < K; ++k) r = fmaf(A[i * K + k], B[k * 32 + j], r); chain[i * 32 + j] = r; }
        float *dA, *dB, *dD; CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dD, D.size() * 4));
        CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        printf("dot products of length %d against fp64 (errors relative to the RMS of the results, %.3f):\n", K, scale);
        auto report = [&](const char* nm, const float* d) {
            double mx = 0, rms = 0;
            for (int i = 0; i < 1024; ++i) { double e = fabs(d[i] - ref[i]) / scale; mx = fmax(mx, e); rms += e * e; }
            printf("  %-28s max %.3e  rms %.3e\n", nm, mx, sqrt(rms / 1024));
        };
        report("fp32 FMA chain (CPU)", chain.data());
        const int tl[5] = {0, 1, 3, 6, 9};
        const char* nm[5] = {"fp32 MFMA 32x32x2", "bf16 x1 (plain bf16)", "bf16 x3", "bf16 x6", "bf16 x9"};
        for (int t = 0; t < 5; ++t) {
            hipLaunchKernelGGL(prod_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, tl[t]);
            CHK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
            report(nm[t], D.data());
        }
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    // ---- 3: rates
    {
        float *W, *out; CHK(hipMalloc(&W, 2 << 20)); CHK(hipMalloc(&out, 64));
        {   // weight stand-ins: bf16 pairs with a random sign, exponent 116..123 and random mantissa
            std::vector<unsigned> hw((2 << 20) / 4);
            unsigned x = 12345u;
            for (auto& v : hw) { x = x * 1664525u + 1013904223u; unsigned a = ((x >> 3) & 0x8000u) | ((116 + ((x >> 8) & 7)) << 7) | ((x >> 12) & 127); x = x * 1664525u + 1013904223u;
                                 unsigned b = ((x >> 3) & 0x8000u) | ((116 + ((x >> 8) & 7)) << 7) | ((x >> 12) & 127); v = a | (b << 16); }
            CHK(hipMemcpy(W, hw.data(), 2 << 20, hipMemcpyHostToDevice));
        }
        printf("rates (256 CUs, 2000 iterations per wave):\n");
        for (int data = 0; data < 3; ++data)
        for (int w = 1; w <= 2; ++w) {
            rate<0>("fp32 v_mfma_f32_32x32x2_f32", W, out, w, 8, 4096.0, 1.0, data);
            rate<1>("bf16 v_mfma_f32_32x32x16_bf16 (pure)", W, out, w, 8, 32768.0, 6.0, data);
            rate<2>("6-product split, 64x64 wave tile, LDS + L2 fed", W, out, w, 24, 32768.0, 6.0, data);
            rate<3>("6-product split, 64x128 wave tile, LDS + L2 fed", W, out, w, 48, 32768.0, 6.0, data);
        }
    }
    return 0;
}
