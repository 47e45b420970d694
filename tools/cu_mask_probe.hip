// cu_mask_probe: the bf16 MFMA loop (operands that look like split activations) on CU-masked streams — under the power limit fewer CUs clock higher.
//   hipcc --offload-arch=gfx950 -O3 -o tools/cu_mask_probe tools/cu_mask_probe.hip && tools/cu_mask_probe
// (ends with _exit: a process that recorded events on CU-masked streams hung / crashed in the runtime's teardown on this image)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float smem[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += 256) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned lo = ((h & 1) << 15) | ((120 + ((h >> 1) & 7)) << 7) | ((h >> 4) & 127), hi = (((h >> 11) & 1) << 15) | ((120 + ((h >> 12) & 7)) << 7) | ((h >> 15) & 127);
        smem[i] = __uint_as_float(lo | (hi << 16));
    }
    __syncthreads();
    bf16x8 a = *(const bf16x8*)(smem + 4 * lane), b = *(const bf16x8*)(smem + 4 * lane + 256);
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    float s = 0; for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int frac : {256, 224, 192, 160, 128}) {
        std::vector<uint32_t> mask(8, 0);
        for (int i = 0; i < frac; ++i) mask[i / 32] |= 1u << (i % 32);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("create failed\n"); _exit(1); }
        hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, st, out, 10000); hipStreamSynchronize(st);
        hipEventRecord(e0, st); hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, st, out, 10000); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 4096.0 * 4 * 10000 * 8 * 32768.0;
        printf("bf16 MFMA loop (realistic operands) on the first %3d CUs: %7.2f ms  %7.1f TFLOP/s executed  (%.2f TFLOP/s per CU)\n", frac, ms, fl / ms * 1e-9, fl / ms * 1e-9 / frac);
        hipStreamSynchronize(st);
    }
    printf("done\n");
    _exit(0);
}
