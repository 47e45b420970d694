// How fast does a wave issue non-MFMA instructions while the other wave of its SIMD streams fp32 MFMAs?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run v_mfma_f32_32x32x2_f32 back to back (or idle, as
// the control), waves 4-7 run a "victim" loop and time it with s_memtime:
//   victims: dependent / independent v_fma_f32 chains, a dependent v_pk_fma_f32 chain, an s_add_u32 chain,
//   ds_read_b128 / buffer_load_dwordx4 streams (8 in flight, each result consumed by a v_add), ds_write_b128.
//   yield = what the MFMA wave does besides MFMAs: 0 nothing (pure stream), 1 s_sleep 1 per 16 MFMAs,
//   5 one ds_read + dependent use per 16 MFMAs (a real wait), 7 s_nop 7 after every MFMA.
// Findings (profiles/r01_valu_under_mfma.log): a wave that streams MFMAs back to back keeps its 64.0 cycles/MFMA and
// STARVES every dependent instruction chain of the other wave on its SIMD (no progress until the stream ends;
// s_setprio does not help).  (Short independent streams can look unaffected here because they finish before the MFMA
// waves have started; tools/hybrid_peak.hip is the chip-wide, long-running version of that case.)  The victim only
// advances in the gaps where the MFMA wave issues something else, and every such gap costs the MFMA wave pipe time.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_under_mfma.hip -o tools/valu_under_mfma && tools/valu_under_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int VICTIM, int NACC, int YIELD>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* clk, const float* gsrc, int mfma_on, int prio) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 1e-4f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave < 4) {
        f32x16 c[NACC];
        for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) c[k][r] = (float)k;
        float a = 0.5f + lane * 1e-3f, b = 0.25f;
        const unsigned long long m0 = __builtin_amdgcn_s_memtime();
        if (mfma_on) {
            for (int i = 0; i < 6000 * 4 / NACC; ++i) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int k = 0; k < NACC; ++k) {
                        c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[k], 0, 0, 0);
                        if (YIELD == 6) asm volatile("s_nop 0");
                        if (YIELD == 7) asm volatile("s_nop 7");
                        if (YIELD == 8 && k == NACC - 1) asm volatile("s_nop 15");
                        if (YIELD == 9) __builtin_amdgcn_s_sleep(1);
                    }
                // YIELD: what the MFMA wave does after each group of 4*NACC MFMAs
                if (YIELD == 1) __builtin_amdgcn_s_sleep(1);
                if (YIELD == 2) asm volatile("s_nop 15");
                if (YIELD == 3) { __builtin_amdgcn_s_setprio(0); }
                if (YIELD == 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (YIELD == 5) { float t = lds[(i & 63) * 64 + lane]; a += t * 1e-9f; }   // a ds_read + dependent use: a real wait
            }
        }
        const unsigned long long m1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += c[k][r];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) clk[1024 + blockIdx.x * 4 + wave] = m1 - m0;
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_sleep(64);    // let the MFMA waves get going
    float acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = lane * 0.001f + k;
    f32x4 v[8];
    for (int k = 0; k < 8; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, 65536, 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VICTIM == 0) {
        float x = acc[0];
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        }
        acc[0] = x;
    } else if (VICTIM == 1) {
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = __builtin_fmaf(acc[k], 1.0001f, 0.5f);
        }
    } else if (VICTIM == 5) {      // 8 independent chains, forced scalar-float v_fma_f32 (no packing)
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(acc[k]));
        }
    } else if (VICTIM == 6) {      // dependent chain of v_pk_fma_f32
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 x = {acc[0], acc[1]}, one = {1.0001f, 1.0001f}, half = {0.5f, 0.5f};
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(one), "v"(half));
        }
        acc[0] = x[0]; acc[1] = x[1];
    } else if (VICTIM == 7) {      // dependent chain of v_fma_f32, pinned
        float x = acc[0];
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(x));
        }
        acc[0] = x;
    } else if (VICTIM == 8) {      // scalar ALU chain
        int x = lane;
        x = __builtin_amdgcn_readfirstlane(x);
        for (int i = 0; i < 4096 / 16; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("s_add_u32 %0, %0, 3" : "+s"(x));
        }
        acc[0] += x;
    } else if (VICTIM == 2) {
        for (int i = 0; i < 1024 / 8; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += *(const f32x4*)(lds + ((i * 8 + k) & 63) * 256 + lane * 4);
        }
    } else if (VICTIM == 3) {
        for (int i = 0; i < 1024 / 8; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, ((i * 8 + k) & 63) * 1024, 0));
        }
    } else if (VICTIM == 4) {
        for (int i = 0; i < 1024 / 8; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) *(f32x4*)(lds + ((i * 8 + k) & 63) * 256 + lane * 4) = v[k];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k] + v[k][0] + v[k][1] + v[k][2] + v[k][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) clk[blockIdx.x * 4 + (wave - 4)] = t1 - t0;
}

template <int VICTIM, int NACC, int YIELD = 0>
void run(const float* gsrc, const char* name, int ninstr) {
    float* out; unsigned long long* clk;
    const int blocks = 256;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4); (void)hipMalloc(&clk, (size_t)(1024 + blocks * 4) * 8);
    for (int mode = 0; mode < 3; ++mode) {
        const int mfma_on = mode > 0, prio = mode == 2;
        hipLaunchKernelGGL((probe<VICTIM, NACC, YIELD>), dim3(blocks), dim3(512), 0, 0, out, clk, gsrc, mfma_on, prio);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(1024 + blocks * 4);
        (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0, msum = 0; for (int i = 0; i < blocks * 4; ++i) { sum += (double)h[i]; msum += (double)h[1024 + i]; }
        printf("%-28s nacc=%d yield=%d %-22s %8.1f ticks/instr   MFMA %6.2f ticks each\n", name, NACC, YIELD, mode == 0 ? "MFMA waves idle" : (mode == 1 ? "MFMA waves streaming" : "streaming, victim prio 3"),
               sum / (blocks * 4) / ninstr, msum / (blocks * 4) / (6000.0 * 16));
    }
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    float* gsrc; (void)hipMalloc(&gsrc, 65536 + 1024); (void)hipMemset(gsrc, 0, 65536 + 1024);
    run<7, 4>(gsrc, "dependent v_fma_f32 (asm)", 4096);
    run<7, 4, 1>(gsrc, "dependent v_fma_f32 (asm)", 4096);
    run<7, 4, 5>(gsrc, "dependent v_fma_f32 (asm)", 4096);
    run<7, 4, 7>(gsrc, "dependent v_fma_f32 (asm)", 4096);
    run<5, 4>(gsrc, "8 indep v_fma_f32 (asm)", 4096);
    run<6, 4>(gsrc, "dependent v_pk_fma_f32", 4096);
    run<8, 4>(gsrc, "s_add_u32 chain", 4096);
    run<2, 4>(gsrc, "ds_read_b128 + v_add", 1024);
    run<2, 4, 5>(gsrc, "ds_read_b128 + v_add", 1024);
    run<4, 4>(gsrc, "ds_write_b128", 1024);
    run<3, 4>(gsrc, "buffer_load_dwordx4 + v_add", 1024);
    run<3, 4, 5>(gsrc, "buffer_load_dwordx4 + v_add", 1024);
    return 0;
}
