// mfma_chain_probe: what a lone wave per SIMD gets out of v_mfma_f32_32x32x16_bf16 in the K loop of ln_mlp_split_kernel's phase 1
// (per 16-deep slab: three 16-byte fragment loads from L2, six 16-byte LDS reads, twelve MFMAs on a 32 x 64 output tile).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain_probe tools/mfma_chain_probe.hip && tools/mfma_chain_probe
// Variants: accumulation chains (2: one per 32 x 32 tile, 4: two partial sums per tile), with / without the LDS reads, with / without
// the fragment loads, LDS row stride, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CHAINS, bool LDS, bool VMEM, int RS>
__global__ __launch_bounds__(256) void k(const float* W, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * RS / 4; i += 256) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned lo = ((h & 1) << 15) | ((120 + ((h >> 1) & 7)) << 7) | ((h >> 4) & 127), hi = (((h >> 11) & 1) << 15) | ((120 + ((h >> 12) & 7)) << 7) | ((h >> 15) & 127);
        smem[i] = __uint_as_float(lo | (hi << 16));
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const char* xrow = (const char*)smem + (lane & 31) * RS + 16 * (lane >> 5);
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 a[3], b[3][2], bn[3][2];
    unsigned woff = (unsigned)wave * 49152u + (unsigned)lane * 16u;
    for (int p = 0; p < 3; ++p) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, woff + p * 1024u, 0, 0); a[p] = __builtin_bit_cast(f32x4, v);
        for (int j = 0; j < 2; ++j) b[p][j] = *(const f32x4*)(xrow + j * 32 * RS + p * 512); }
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int sl = (it + 1) & 15;
        if (LDS) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) bn[p][j] = *(const f32x4*)(xrow + j * 32 * RS + p * 512 + sl * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16& c = acc[CHAINS == 4 ? j + 2 * (t & 1) : j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[t]]), __builtin_bit_cast(bf16x8, b[PB[t]][j]), c, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (VMEM) {
            woff = (woff + 3072u) & 0xfffffu;
#pragma unroll
            for (int p = 0; p < 3; ++p) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, woff + p * 1024u, 0, 0); a[p] = __builtin_bit_cast(f32x4, v); }
        }
        if (LDS) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[p][j] = bn[p][j];
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS, bool LDS, bool VMEM, int RS>
static void run(const char* name, const float* W, float* out, int wpc) {
    const int iters = 20000, grid = 256 * wpc;
    const size_t smem = 64 * RS;
    CHK(hipFuncSetAttribute((const void*)k<CHAINS, LDS, VMEM, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<CHAINS, LDS, VMEM, RS>), dim3(grid), dim3(256), smem, 0, W, out, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<CHAINS, LDS, VMEM, RS>), dim3(grid), dim3(256), smem, 0, W, out, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_per_slab = (double)ms * 1e6 / iters / wpc;
    printf("  %-58s %d wave(s)/SIMD: %6.1f ns per slab of 12 MFMAs per wave  (%5.1f ns per MFMA; 32 cycles = 13.3 ns at 2.4 GHz)\n", name, wpc, ns_per_slab, ns_per_slab / 12);
}

int main() {
    float *W, *out; CHK(hipMalloc(&W, 2 << 20)); CHK(hipMalloc(&out, 64));
    {
        std::vector<unsigned> hw((2 << 20) / 4);
        unsigned x = 12345u;
        for (auto& v : hw) { x = x * 1664525u + 1013904223u; unsigned a = ((x >> 3) & 0x8000u) | ((116 + ((x >> 8) & 7)) << 7) | ((x >> 12) & 127); x = x * 1664525u + 1013904223u;
                             unsigned b = ((x >> 3) & 0x8000u) | ((116 + ((x >> 8) & 7)) << 7) | ((x >> 12) & 127); v = a | (b << 16); }
        CHK(hipMemcpy(W, hw.data(), 2 << 20, hipMemcpyHostToDevice));
    }
    for (int w = 1; w <= 2; ++w) {
        run<2, true, true, 1552>("2 chains, LDS reads + fragment loads, row stride 1552", W, out, w);
        run<4, true, true, 1552>("4 chains, LDS reads + fragment loads, row stride 1552", W, out, w);
        run<4, false, true, 1552>("4 chains, fragment loads only", W, out, w);
        run<4, true, false, 1552>("4 chains, LDS reads only, row stride 1552", W, out, w);
        run<4, false, false, 1552>("4 chains, MFMAs only", W, out, w);
        run<2, false, false, 1552>("2 chains, MFMAs only", W, out, w);
        run<4, true, false, 400>("4 chains, LDS reads only, row stride 400", W, out, w);
        run<4, true, false, 784>("4 chains, LDS reads only, row stride 784", W, out, w);
    }
    return 0;
}
