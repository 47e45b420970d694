// Device kernels of the EmojiVoice hot path for gfx950 (MI355X, CDNA4).
//
// Data layout in HBM ("frame-major, padded"): every activation is a row-major
// (rows, C) fp32 matrix whose row index n runs over the flattened padded time
// axis of the batch: utterance b, frame t lives at row  n = b*S + P + t,  S = T + 2P.
// The P pad rows around each utterance are zero for the lifetime of the buffer
// (the workspace is zeroed when its shape changes and kernels only ever store
// rows with 0 <= t < T), so they ARE the conv zero padding: a K-tap dilated
// window is just K row offsets and tiles may straddle utterances.
//
// conv_gemm_kernel: implicit GEMM  Y[n, co] = sum_{tap, ci} W[tap][co][ci] * X[n + off[tap], ci]
// on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain,
// 157 TFLOP/s peak = the fp32 vector peak but 1 LDS dword per operand per 64 FLOP-lanes).
//   A operand = weights  (rows = output channels), B operand = activations (cols = frames).
// Both are staged to LDS as [row][32 k + 4 pad] so that one ds_read_b128 per lane
// feeds four consecutive MFMAs (k = kk + 4*half + s, s = 0..3) with no bank conflicts
// (row stride 36 dwords: 16 consecutive rows hit 16 distinct 4-bank slots).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define EV_MAX_TAPS 16
#define EV_BK 32          // k (input-channel) chunk per LDS stage
#define EV_LDK 36         // LDS row stride in floats (32 + 4 pad)
#define EV_WROWS 128      // weight rows per LDS stage (= taps_per_stage * BM)
// conv_gemm_sk_kernel: float4 of the X tile one staging round may hold (TW = 4: (64 + 64) rows x 8 chunks; TW = 1: 16 per thread = 32 rows x all 1024 input channels of the widest 1x1 layer)
#define EV_SK_MAXF4(TW) ((TW) == 4 ? (64 + 64) * 8 * 8 : 16 * 512)
#define EV_HALO 64        // max halo rows (lo + hi) an X tile may carry

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_SILU = 3, ACT_MISH = 4, ACT_SNAKE = 5 };

struct SkCtl {
    unsigned* ctrl;      // [0] epoch, [1] arrivals of the running launch, [2] waits that ran out (diagnostic); null = no hand-offs
    unsigned* flags;     // one word per workgroup: tag of the launch whose partial is in that workgroup's slot
    float* part;         // per workgroup: two tiles of `part_floats` (published partial | private spill of the fallback path)
    int part_floats;
    int q, r;            // units per workgroup: start(x) = x * q + min(x, r)
    int spin_limit;      // polls before an owner gives up waiting
    unsigned* claims;    // one word per workgroup (null: no stealing): 2 seq = "started its share of this launch", 2 seq + 1 = "its owner has taken the share over"
    unsigned seq;        // launch sequence number of the handle (host-side counter: monotonic, so the words never need a reset)
    int lds_word;        // conv_gemm_bal_kernel: byte offset of the sk_wait word in dynamic LDS
    // conv_gemm_bal_kernel, layers whose M tiles differ in cost (a 3-tap conv stacked over a 1x1 conv: half of the M tiles carry one
    // tap): the workgroups share WEIGHTED positions — a unit of M tile mt weighs wtab[mt] (its taps, 4 bits each, up to 16 M tiles) —
    // and a position is rounded down to the unit that contains it.  wsum = sum of the weights over the M tiles (0 = all units equal).
    int wsum; int mtiles; int nchunks; unsigned long long wtab;
};

struct ConvParams {
    const float* X; int ldx; int Cin;
    int isplit_log2, isstride;              // input column ci lives at (ci >> isplit_log2)*isstride + (ci & (2^isplit_log2 - 1))
    const float* W; int Mpad; int Kpad;     // fragment order [ntaps][Mpad/32][Kpad/8][64 lanes][4]
    const void* Wh; float wh_scale;         // conv_h16_kernel: the weights times wh_scale (a power of two) as two fp16 pieces, [ntaps][Mpad/32][Kpad/16][2][64 lanes][8]
    const void* Wq;                         // conv_h16_kernel<..., Q = 1>: the same pieces in v_mfma_f32_16x16x32_f16 fragment order, [ntaps][Mpad/16][Kpad/32][2][64 lanes][8] (null: not packed)
    const void* Wx;                         // conv_split_kernel: the same weights as three bf16 pieces, [ntaps][Mpad/32][Kpad/16][3][64 lanes][8] (null: not packed)
    const float* bias;                      // [Cout] or null
    float* Y; int ldy; int Cout;
    int osplit_log2, osstride;              // same column split for Y / R / Y2 (pair views of a strided slice)
    int mmul;                               // rowmask index = n*mmul + (co >> osplit_log2)
    int nrows;                              // rows of the flattened padded axis (B*S)
    int S, P, T;                            // store predicate: 0 <= (n % S) - P < T
    int ntaps; int off[EV_MAX_TAPS];
    int halo_lo, halo_hi;
    const int2* taplist; int tl_stride;     // per M tile: compact list of non-zero taps {tap index, row offset}; stride 0 = shared
    const int* nact_tab;                    // per M tile count of non-zero taps, or null (= ntaps)
    int mtiles, ntiles;
    float pro_slope; int pro_lrelu;         // prologue leaky-relu on X
    // epilogue (order: +bias, act, *mask1, *scale, +R, +Yold, /3, lrelu2, *mask2)
    int act; float act_slope; const float* act_a; const float* act_b;
    int mask1; float scale; const float* R; int ldr; int accum; int div3;
    int act2_lrelu; float act2_slope; int mask2;
    const float* rowmask;
    float* Y2; int ldy2;                    // optional second output = v * rowmask
    int ktaps_n, plane_bytes, koff0, kdoff; // dense layers (every tap in every tile) with evenly spaced taps: tap i = {i * plane_bytes,
                                            // koff0 + i * kdoff} — conv_gemm_sk_kernel derives the list from the kernel arguments instead of fetching it
    float* gn_part;                         // conv_sk32_kernel, one utterance: per (32-row tile, 32-channel group) {count, mean, M2} of the stored
                                            // tile — the GroupNorm statistics of the layer that follows (groupnorm_apply_kernel), or null
    int kstack_mt, kstack_tap;              // stacked layers (a k-tap conv over a 1x1 conv along Cout): 32-channel tiles >= kstack_mt carry the
                                            // single tap kstack_tap, the tiles below it all ntaps (0 = not such a layer); conv_sk32_kernel only
    int sk_kb;                              // conv_gemm_sk_kernel: k-chunks per staging round
    // Per-granule upper bounds of |tensor| ("amax slots": one word per 128 rows of a buffer's flattened padded axis, the fp32 bits of a bound on
    // every value stored there since the slots were last zeroed; 0x7f800000 = a non-finite value was stored).  Producers leave them from their
    // accumulators (ev_amax_emit), the fp16 builds take their tile scale from them instead of pre-scanning the X tile (conv_h16_kernel).
    unsigned* ymax; int ymax_mul;           // producer: slots of Y (null: none); a row n of this launch covers rows n * ymax_mul .. of the slots' axis
    const unsigned* rmax;                   // producer: slots of R (the residual's bound adds to the accumulators')
    const unsigned* yold;                   // producer of a running sum (accum): the slots that hold the bound of what Y held BEFORE this launch (never ymax itself:
                                            // a bound read from slots the same launch is updating would depend on timing — and the tile scales, hence the bits, with it)
    const unsigned* xmax; int xmax_n;       // consumer: slots of X and their count (null: pre-scan)
    int stagger_slots;                      // workgroups co-resident per CU (0 = no start stagger), see conv_gemm_kernel
    int mt_mul;                             // conv_h16_bal_kernel with 256-channel M tiles: tap list / tap count of M tile mt are those of the 128-channel tile mt_mul * mt (0 = 1)
    unsigned long long* stamps;             // dbg bit 16: per-workgroup {start, first stage done, K loop done, end} s_memtime stamps
    int dbg;                                // ablation bits for tools/conv_bench.py: 1 skip X loads, 2 skip A loads, 4 skip epilogue
    SkCtl sk;                               // conv_gemm_kernel<..., SK = true>: the balanced persistent grid (units = (tile, 32-channel k-chunk))
};


// Buffer (SRSRC) addressing for every global access of the conv kernels.  Measured on MI355X (tools/mfma_peak.hip):
// flat `global_load_dwordx4` with per-lane 64-bit addresses issued by one wave costs the MFMA stream of the OTHER
// waves on the same SIMD ~20 % (155 -> 120-125 TFLOP/s at 2-3 waves per SIMD); the same loads as `buffer_load_dwordx4`
// (128-bit descriptor in SGPRs + one 32-bit per-lane byte offset) cost nothing.  All tensors here are < 4 GiB.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ev_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ev_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ float ev_bload1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff_bytes, (int)soff_bytes, 0));
}
__device__ __forceinline__ void ev_bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff_bytes, 0, 0);
}
__device__ __forceinline__ void ev_bstore1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff_bytes, 0, 0);
}

// Tap-list entries are wave-uniform, but hipcc fetches them with a vector load and then treats everything derived from
// them (weight-fragment soffset, LDS row offset) as divergent: every buffer_load got a waterfall loop
// (v_readfirstlane / v_cmp / s_and_saveexec ...) and the offsets were recomputed with v_mul.  readfirstlane makes the
// uniformity provable, so the descriptor offsets live in SGPRs (guide T20).
__device__ __forceinline__ int2 ev_uniform(int2 v) {
    return make_int2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y));
}
// The whole tap list of an M tile (<= EV_MAX_TAPS entries) lives in ONE register pair, entry i on lane i, fetched once per
// workgroup; entry ti is then a v_readlane with a scalar index.  Fetching an entry per tap from memory — even one tap ahead —
// put an `s_waitcnt vmcnt(0)` at the top of every tap iteration (the entry's use must wait for a vector load, and the counter
// retires in order), which drained the weight-fragment prefetches there: the fragment pipeline was never deeper than the
// distance to the next tap boundary, whatever the code said.
__device__ __forceinline__ int2 ev_tap_at(int2 tlv, int i) {
    return make_int2(__builtin_amdgcn_readlane(tlv.x, i), __builtin_amdgcn_readlane(tlv.y, i));
}

// leaky-relu for slopes in [0, 1] (every use here): max(v, v*s) is two VALU instructions instead of compare / multiply / select
__device__ __forceinline__ float ev_lrelu(float v, float s) {
    float r;   // plain v_max_f32: fmaxf() makes hipcc quiet a possible sNaN first (one extra v_max v, v, v per element)
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(v * s));
    return r;
}
// mish(x) = x * tanh(softplus(x)) (decoder.py:41-43).  With w = e^x:  tanh(log(1 + w)) = (w^2 + 2w) / (w^2 + 2w + 2), so one
// exp and one division replace exp + log1p + tanh (the GroupNorm+Mish pass over 8.5 M elements per launch was VALU-bound
// on ocml's log1pf / tanhf).  torch's softplus threshold (x > 20 -> softplus = x, tanh = 1 in fp32) is kept.
__device__ __forceinline__ float ev_mish(float x) {
    const float w = expf(fminf(x, 20.f));
    const float n = w * (w + 2.f);
    return x > 20.f ? x : x * __fdividef(n, n + 2.f);
}
__device__ __forceinline__ float ev_silu(float x) { return x / (1.f + expf(-x)); }

// sin^2(u) in ~17 VALU instructions (the SnakeBeta feed-forward epilogue evaluates 34 M of these per launch; ocml's
// sinf with its Payne-Hanek tail made that epilogue cost more than the GEMM in front of it).  Cody-Waite reduction by
// pi/2 with three fma steps, the cephes single-precision sine polynomial on |r| <= pi/4, and sin^2 = 1 - sin^2 on odd
// quadrants.  Absolute error <= 2e-7 for |u| <= 1e4 (hidden activations are O(10)); degrades gracefully beyond that.
__device__ __forceinline__ float ev_sin2(float u) {
#ifdef EV_SIN2_HW
    // A/B build (-DEV_SIN2_HW): the hardware cosine.  sin^2 has period pi: Cody-Waite reduction by pi in two fma steps (r in [-pi/2, pi/2]), then
    // sin^2 r = (1 - cos 2r) / 2 with v_cos_f32, whose argument is in revolutions (2r / 2pi = r / pi).  ~8 issue slots instead of ~15; absolute
    // error against fp64 2.0e-7 max / 4.2e-8 rms over |u| <= 1000, against 1.1e-7 / 2.1e-8 for the polynomial (tools/sin2_probe.hip).
    const float kk = rintf(u * 0.318309886183790672f);
    float rr = fmaf(kk, -3.1415927410125732421875f, u);
    rr = fmaf(kk, 8.742277657347586e-8f, rr);
    return fmaf(__builtin_amdgcn_cosf(rr * 0.318309886183790672f), -0.5f, 0.5f);
#endif
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

__device__ __forceinline__ float ev_act(float v, int act, float slope, const float* a, const float* b, int co) {
    switch (act) {
        case ACT_LRELU: return ev_lrelu(v, slope);
        case ACT_TANH: return tanhf(v);
        case ACT_SILU: return ev_silu(v);
        case ACT_MISH: return ev_mish(v);
        case ACT_SNAKE: return fmaf(b[co], ev_sin2(v * a[co]), v);
        default: return v;
    }
}

// XCD-aware bijective remap of the linear block id: blocks that share an XCD
// (id % 8 equal) get a contiguous range of work items, so neighbouring tiles
// (same X rows, adjacent halos) hit the same per-XCD L2.
__device__ __forceinline__ int ev_xcd_remap(int id, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = id & 7, within = id >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence: hipcc emits s_waitcnt vmcnt(0) in front
// of it, i.e. it also waits for every global STORE this wave still has in flight — in the conv epilogue that made each
// slab's barrier wait for the previous slab's stores to be acknowledged by memory (30-60 us per workgroup, measured
// with the per-workgroup stamps in profiles/r01_conv_workgroup_timelines.log).
__device__ __forceinline__ void ev_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fused conv epilogue for one wave's (TM*32 channels) x (TN*32 frames) accumulator tile whose first channel / frame are
// mw0 / nw0.  Es = this wave's private LDS slab [32][TM*32 + 4].  Must be called by all waves of the workgroup
// (it contains workgroup barriers).
template <int TM, int TN, bool FULL_ACT = true>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[TM][TN], float* Es, int mw0, int nw0, int lane,
                                              int win_lo = -0x7fffffff, int win_hi = 0x7fffffff) {
    const int li = lane & 31, lh = lane >> 5;
    // ---- epilogue.  C/D layout of the 32x32 tile: col = lane&31 (frame), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // Each wave transposes one 32-frame slab of its tile through LDS into [frame][channel] so that the
    // fused epilogue runs on whole float4 channel groups and stores 16 lanes x 16 B contiguous per frame.
    constexpr int EC = TM * 32;          // channels of the wave's sub-tile
    constexpr int ELD = EC + 4;          // LDS row stride
    constexpr int C4 = EC / 4;           // float4 groups per frame
    constexpr int RPP = 64 / C4;         // frames per pass of the wave
    const bool vec_ok = ((p.ldy & 3) == 0) && ((p.Cout & 3) == 0) && ((p.osstride & 3) == 0) && (!p.R || (p.ldr & 3) == 0) &&
                        (!p.Y2 || (p.ldy2 & 3) == 0);
    const int omask = (p.osplit_log2 >= 31) ? 0x7fffffff : ((1 << p.osplit_log2) - 1);
    const int er = lane / C4;            // frame within a pass
    const int ec = (lane % C4) * 4;      // channel offset inside the sub-tile
    const int co = mw0 + ec;
    constexpr int NP = 32 / RPP;                 // passes per 32-frame slab
    constexpr int NB = NP > 4 ? 4 : NP;          // passes whose residual / accumulate loads are in flight together
    const bool co_ok = co < p.Cout;
    const int sp = (p.osplit_log2 >= 31) ? 0 : (co >> p.osplit_log2);
    const unsigned col = (unsigned)sp * p.osstride + (co & omask);   // column offset inside a row
    const bool full = vec_ok && (co + 3 < p.Cout);
    const __amdgpu_buffer_rsrc_t rY = ev_rsrc(p.Y), rR = ev_rsrc(p.R), rY2 = ev_rsrc(p.Y2);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && co_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[e] = p.bias[(co + e < p.Cout) ? co + e : p.Cout - 1];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int pb = 0; pb < NP; pb += NB) {
            // issue the global loads of this batch first: they fly while the slab is transposed through LDS
            f32x4 rr[NB], yo[NB];
            bool ok[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int n = nw0 + j * 32 + (pb + q) * RPP + er;
                const int t = (n % p.S) - p.P;
                ok[q] = co_ok && n < p.nrows && t >= 0 && t < p.T && n >= win_lo && n < win_hi;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rr[q] = z; yo[q] = z;
                if (ok[q]) {
                    if (full) {
                        if (p.R) rr[q] = ev_bload4(rR, ((unsigned)n * p.ldr + col) * 4u, 0);
                        if (p.accum) yo[q] = ev_bload4(rY, ((unsigned)n * p.ldy + col) * 4u, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (co + e < p.Cout) {
                                if (p.R) rr[q][e] = ev_bload1(rR, ((unsigned)n * p.ldr + col + e) * 4u, 0);
                                if (p.accum) yo[q][e] = ev_bload1(rY, ((unsigned)n * p.ldy + col + e) * 4u, 0);
                            }
                        }
                    }
                }
            }
            if (pb == 0) {
                ev_lds_barrier();            // LDS free (K loop done / previous slab consumed)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 q4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *(f32x4*)(Es + li * ELD + i * 32 + 8 * g + 4 * lh) = q4;
                    }
                ev_lds_barrier();
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                if (!ok[q]) continue;
                const int rl = (pb + q) * RPP + er;
                const int n = nw0 + j * 32 + rl;
                const f32x4 a = *(const f32x4*)(Es + rl * ELD + ec);
                float v[4] = {a[0], a[1], a[2], a[3]};
                const float rm = p.rowmask ? p.rowmask[(size_t)n * p.mmul + sp] : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (co + e < p.Cout) ? co + e : p.Cout - 1;
                    float x = v[e] + bs[e];
                    if constexpr (FULL_ACT) { if (p.act) x = ev_act(x, p.act, p.act_slope, p.act_a, p.act_b, c); }
                    else { if (p.act == ACT_LRELU) x = ev_lrelu(x, p.act_slope); }   // compact build: no transcendental code in the I-cache
                    if (p.mask1) x *= rm;
                    x *= p.scale;
                    if (p.R) x += rr[q][e];
                    if (p.accum) x += yo[q][e];
                    if (p.div3) x = x / 3.0f;
                    if (p.act2_lrelu) x = ev_lrelu(x, p.act2_slope);
                    if (p.mask2) x *= rm;
                    v[e] = x;
                }
                if (full) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    ev_bstore4(rY, ((unsigned)n * p.ldy + col) * 4u, o);
                    if (p.Y2) { f32x4 o2 = {v[0] * rm, v[1] * rm, v[2] * rm, v[3] * rm}; ev_bstore4(rY2, ((unsigned)n * p.ldy2 + col) * 4u, o2); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < p.Cout) {
                            ev_bstore1(rY, ((unsigned)n * p.ldy + col + e) * 4u, v[e]);
                            if (p.Y2) ev_bstore1(rY2, ((unsigned)n * p.ldy2 + col + e) * 4u, v[e] * rm);
                        }
                    }
                }
            }
        }
    }
}

#ifndef EV_CONV_MIN_WAVES
#define EV_CONV_MIN_WAVES 1
#endif
// ---------------------------------------------------------------------------
// Instruction-lean epilogue.  Per-workgroup stamps (profiles/r01_conv_workgroup_timelines.log) show the generic epilogue
// taking 30-60 us of a 80-210 us workgroup lifetime, and the longer the more the co-resident waves are MFMA-busy: every
// vector instruction of a memory-phase wave waits behind the 64-cycle fp32 MFMAs of its SIMD, so epilogue time is
// (number of vector instructions) x (MFMA issue slot).  This version serves the layers whose epilogue is
//     y = act(acc + bias) [+ R]          act in {none, leaky-relu},  Cout % 4 == 0, plain row-major Y / R
// with ~10x fewer vector instructions: the bias is preloaded into the accumulators by the caller, row bookkeeping is
// incremental (one modulo per lane), the per-wave LDS transposition needs no workgroup barrier after the first one,
// and the residual rows of slab j+1 are requested before slab j's stores (vmcnt retires in order).
// ---------------------------------------------------------------------------
// MODE 1: y = lrelu?(acc) [+ R]      MODE 2: y = snake(acc)      MODE 3: y = lrelu2?((acc + R + Yold) [/ 3])  (resblock mean)
__device__ __forceinline__ float ev_div3(float x) {   // correctly rounded x / 3 in three instructions
    const float q = x * 0.333333343267440796f;
    return fmaf(fmaf(-3.f, q, x), 0.333333343267440796f, q);
}
struct EvNoHook { __device__ __forceinline__ void operator()() const {} };
// `after_issue` runs between the request of the first slab's residual / mask rows and the barrier in front of the transposition
// (conv_gemm_sk_kernel sums its partial tiles there, in the shadow of those loads; it may still modify acc).
// `put_slab(j)` writes the wave's accumulators of the 32-frame slab j to Es as [frame][channel] (row stride TM * 32 + 4): the two accumulator
// layouts (32 x 32 tiles of v_mfma_f32_32x32x*, 16 x 16 tiles of v_mfma_f32_16x16x32_f16) differ only there.
template <int TM, int TN, int MODE, class Hook, class PutSlab>
__device__ __forceinline__ void conv_epilogue_lean_impl(const ConvParams& p, PutSlab put_slab, float* Es, int mw0, int nw0, int lane,
                                                        int win_lo, int win_hi, Hook after_issue) {
    constexpr int EC = TM * 32, ELD = EC + 4, C4 = EC / 4, RPP = 64 / C4, NP = 32 / RPP;
    // Everything per-lane below derives from `lane`.  Left visible, hipcc computes these offsets — loop invariants of the callers' tile /
    // chunk loops — at the top of the kernel and keeps them in registers through the K loop; in the persistent builds that cost 30-60
    // registers and put conv_h16_bal_kernel / ln_mlp_h16_kernel into scratch (round 3).  The empty asm makes them start HERE.
    asm volatile("" : "+v"(lane));
    const int er = lane / C4, ec = (lane % C4) * 4;
    const int co = mw0 + ec;
    const bool co_ok = co < p.Cout;
    const __amdgpu_buffer_rsrc_t rY = ev_rsrc(p.Y), rR = ev_rsrc(p.R);
    const bool has_r = p.R != nullptr;
    const bool do_act = p.act == ACT_LRELU;
    constexpr bool SNAKE = MODE == 2, ACC = MODE == 3;
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa;
    if constexpr (SNAKE) {
        if (co_ok) { sa = *(const f32x4*)(p.act_a + co); sb = *(const f32x4*)(p.act_b + co); }
    }
    // row bookkeeping for this lane: rows n0l, n0l + RPP, ... ; t = position inside the utterance
    const int n0l = nw0 + er;
    int t = n0l % p.S;
    if (t < 0) t += p.S;                              // (only the fused pair kernel starts before row 0)
    t -= p.P;
    // (output column split — the pair view a stride-2 transposed conv writes: logical channel co lives at column
    // (co >> osplit_log2) * osstride + (co & (2^osplit_log2 - 1)), and its mask row is n * mmul + (co >> osplit_log2); a lane's four
    // channels never straddle a split)
    const int sp = (p.osplit_log2 >= 31) ? 0 : (co >> p.osplit_log2);
    const unsigned col = (p.osplit_log2 >= 31) ? (unsigned)co : (unsigned)sp * (unsigned)p.osstride + (unsigned)(co & ((1 << p.osplit_log2) - 1));
    unsigned yoff = ((unsigned)n0l * p.ldy + col) * 4u, roff = ((unsigned)n0l * p.ldr + col) * 4u;
    const unsigned ystep = (unsigned)(RPP * p.ldy) * 4u, rstep = (unsigned)(RPP * p.ldr) * 4u;
    // ONE register set for the residual / running-sum / mask rows of a slab: row q of slab j+1 is requested right after row q
    // of slab j has been consumed, so a whole slab of loads is still in flight while the current slab is transposed and stored
    // (vmcnt retires in order).  A second set (double buffering) cost 40-80 VGPRs in the 128 x 128 build and with them one
    // workgroup per CU: 172 / 236 registers admit two waves per SIMD, <= 168 admit three.
    f32x4 rr[NP], ra[ACC ? NP : 1];
    float rmk[NP];                                    // frame-mask values (mask1: before + R, mask2: after), one per row
    const bool mk = p.mask1 || p.mask2;
    const __amdgpu_buffer_rsrc_t rM = ev_rsrc(p.rowmask);
    // Every load of the walk is UNCONDITIONAL (a load under a branch makes hipcc stop counting vmcnt and drain it at the next
    // use): a launch without residual / mask / running sum reads — and ignores — the first bytes of Y instead, and rows outside
    // the tensor read offset 0.  Rows that are stored always get their real operands.  (Measured as an in-run A/B of two builds
    // on one box: 0.6 % on the config-2 vocoder; box-to-box spread is larger than that.)
    const __amdgpu_buffer_rsrc_t rRe = has_r ? rR : rY, rMe = mk ? rM : rY;
    const bool use_acc = ACC && p.accum;
    int nn = n0l;
    unsigned ro = roff, ao = yoff;
    auto issue_one = [&](int q) {                      // next row of the walk (nn / ro / ao advance with it) into slot q
        const bool inr = co_ok && nn >= 0 && nn < p.nrows;
        rr[q] = ev_bload4(rRe, (has_r && inr) ? ro : 0u, 0);
        rmk[q] = ev_bload1(rMe, (mk && inr) ? ((unsigned)nn * (unsigned)p.mmul + (unsigned)sp) * 4u : 0u, 0);
        if constexpr (ACC) {
            ra[q] = ev_bload4(rY, (use_acc && inr) ? ao : 0u, 0);
            ao += ystep;
        }
        nn += RPP;
        ro += rstep;
    };
    const int lim = p.S - p.P;                        // t walks in [-P, S - P); S >= 4 (host: lean_ok), so two conditional wraps cover RPP <= 8
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_one(q);
    after_issue();
    ev_lds_barrier();                                 // every wave is done reading the X tile: LDS can be reused
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // per-wave transposition through this wave's private LDS slab (ordered by the wave's own program order)
        put_slab(j);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int rl = q * RPP + er;
            const int n = n0l + (j * NP + q) * RPP;
            f32x4 v = *(const f32x4*)(Es + rl * ELD + ec);
            const bool ok = co_ok && t >= 0 && t < p.T && n < p.nrows && n >= win_lo && n < win_hi;
            if constexpr (SNAKE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(sb[e], ev_sin2(v[e] * sa[e]), v[e]);
            } else if (do_act) {
                v[0] = fmaxf(v[0], v[0] * p.act_slope); v[1] = fmaxf(v[1], v[1] * p.act_slope);
                v[2] = fmaxf(v[2], v[2] * p.act_slope); v[3] = fmaxf(v[3], v[3] * p.act_slope);
            }
            if (p.mask1) v *= rmk[q];
            if (has_r) v += rr[q];
            if (p.mask2) v *= rmk[q];
            if constexpr (ACC) {
                if (use_acc) v += ra[q];
                if (p.div3) { v[0] = ev_div3(v[0]); v[1] = ev_div3(v[1]); v[2] = ev_div3(v[2]); v[3] = ev_div3(v[3]); }
                if (p.act2_lrelu) {
                    v[0] = fmaxf(v[0], v[0] * p.act2_slope); v[1] = fmaxf(v[1], v[1] * p.act2_slope);
                    v[2] = fmaxf(v[2], v[2] * p.act2_slope); v[3] = fmaxf(v[3], v[3] * p.act2_slope);
                }
            }
            if (ok) ev_bstore4(rY, yoff, v);
            if (j + 1 < TN) issue_one(q);             // slot q is free: request the same row of the next slab
            t += RPP; t -= (t >= lim) ? p.S : 0; t -= (t >= lim) ? p.S : 0;
            yoff += ystep;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab reads done before the next slab overwrites Es
    }
}
template <int TM, int TN, int MODE = 1, class Hook = EvNoHook>
__device__ __forceinline__ void conv_epilogue_lean(const ConvParams& p, f32x16 (&acc)[TM][TN], float* Es, int mw0, int nw0, int lane,
                                                   int win_lo = -0x7fffffff, int win_hi = 0x7fffffff, Hook after_issue = Hook()) {
    // C/D layout of a 32x32 tile: col = lane & 31 (frame), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
    auto put = [&](int j) {
        constexpr int ELD = TM * 32 + 4;
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 q4 = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *(f32x4*)(Es + li * ELD + i * 32 + 8 * g + 4 * lh) = q4;
            }
    };
    conv_epilogue_lean_impl<TM, TN, MODE>(p, put, Es, mw0, nw0, lane, win_lo, win_hi, after_issue);
}
// ... for accumulators in 16 x 16 tiles (v_mfma_f32_16x16x32_f16): acc[a][b] = channels 16 a + 4 (lane >> 4) + 0..3 of frame 16 b + (lane & 15);
// TM / TN count 32-wide units as above (the wave tile is 2 TM x 2 TN tiles of 16 x 16)
template <int TM, int TN, int MODE = 1>
__device__ __forceinline__ void conv_epilogue_lean_q(const ConvParams& p, f32x4 (&acc)[2 * TM][2 * TN], float* Es, int mw0, int nw0, int lane,
                                                     int win_lo = -0x7fffffff, int win_hi = 0x7fffffff) {
    auto put = [&](int j) {
        constexpr int ELD = TM * 32 + 4;
        const int f = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int a = 0; a < 2 * TM; ++a) *(f32x4*)(Es + (b2 * 16 + f) * ELD + a * 16 + 4 * kg) = acc[a][2 * j + b2];
    };
    conv_epilogue_lean_impl<TM, TN, MODE>(p, put, Es, mw0, nw0, lane, win_lo, win_hi, EvNoHook());
}

// ---------------------------------------------------------------------------
// Balanced work assignment inside ONE launch ("stream-K").  The U-Net launches of a batch-64 decode are about one round of
// workgroups, so a launch lasts as long as its busiest CU: 520 row tiles of a feed-forward on 256 CUs leave eight CUs with three
// tiles while the others hold two (measured with tools/shape_profile.py: 62 utterances — 504 tiles — decode 14 % faster than
// 64 for 3 % less work).  Here a launch is G persistent workgroups (G = CUs x workgroups that fit a CU), the work is cut into
// UNITS finer than a tile (a feed-forward row tile = 8 hidden-width chunks), and workgroup g takes the contiguous unit range
// [start(g), start(g + 1)), start(x) = x * q + min(x, r) with q = U / G, r = U % G.  A tile whose units straddle two (or three)
// workgroups is finished by the OWNER — the workgroup holding its first unit, which reaches it LAST in its own range — the others
// reach their share of it FIRST and publish an un-biased partial accumulator tile, which the owner adds in ascending workgroup
// order (deterministic).  Hand-off = MI355X guide, Guideline 16 R1: write-through (sc1) payload stores, every storing wave drains
// vmcnt, workgroup barrier, ONE lane stores the flag with an agent-scope atomic; the consumer polls that word relaxed, ONE agent
// acquire, barrier, then sc1 loads.  Flags carry a launch tag (device-side epoch word + 1, bumped by the last workgroup to
// arrive at the end of the launch), so nothing has to be zeroed per launch and a replayed hipGraph stays correct.
// No protocol step depends on dispatch order or residency: a publisher never waits before it publishes, and an owner whose wait
// runs out (a contributor that is not resident yet) recomputes the missing share itself AS A SEPARATE partial sum, i.e. with
// the bits the contributor would have delivered — slower, never different, never a hang.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned ev_gu32;
__device__ __forceinline__ int sk_start(const SkCtl& c, int x) {
    const int pos = x * c.q + (x < c.r ? x : c.r);
    if (c.wsum == 0) return pos;
    // weighted: position -> unit.  Units are ordered (n tile, M tile, chunk); one n tile spans nchunks * wsum positions
    const int per_nt = c.nchunks * c.wsum;
    const int nt = pos / per_nt;
    int rem = pos - nt * per_nt, mt = 0;
    for (; mt < c.mtiles - 1; ++mt) {
        const int span = c.nchunks * (int)((c.wtab >> (4 * mt)) & 15ull);
        if (rem < span) break;
        rem -= span;
    }
    const int w = (int)((c.wtab >> (4 * mt)) & 15ull);
    int ch = rem / w;
    if (ch > c.nchunks) ch = c.nchunks;                    // (pos == total: the end sentinel lands on the first unit of the next n tile)
    return (nt * c.mtiles + mt) * c.nchunks + ch;
}
__device__ __forceinline__ unsigned sk_tag(const SkCtl& c) {
    return c.ctrl ? (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((ev_gu32*)c.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u : 0u;
}
__device__ __forceinline__ f32x4 ev_bload4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, 0, 16));    // aux 16 = sc1: served by L2, never a stale L1 line
}
__device__ __forceinline__ void ev_bstore4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff_bytes, 0, 16);       // write-through
}
// ... with the wave-uniform part of the address in the scalar offset: the per-lane part stays ONE register for every slot of a partial tile
// (as `base + lane part + k KiB` in the vector offset hipcc hoisted sixteen loop-invariant per-lane addresses out of the unit loop and
// spilled them to scratch — the conv_h16_bal_kernel / ln_mlp_h16_kernel spills of round 3)
__device__ __forceinline__ f32x4 ev_bload4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff_bytes, (int)soff_bytes, 16));
}
__device__ __forceinline__ void ev_bstore4_sc1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (int)voff_bytes, (int)soff_bytes, 16);
}
// after every wave's payload stores: drain, meet, ONE lane raises the flag
__device__ __forceinline__ void sk_publish(const SkCtl& c, int g, unsigned tag, int tid) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("" : "+s"(tag));                      // (the tag lives in a scalar register; its vector copy for the store is made here, not kept — or spilled — across the unit loop)
    if (tid == 0) __hip_atomic_store((ev_gu32*)(c.flags + g), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true when workgroup gi's partial of this launch is readable by every wave of the caller (all threads must call; `word` = one LDS int)
__device__ __forceinline__ bool sk_wait(const SkCtl& c, int gi, unsigned tag, int tid, int* word) {
    if (tid == 0) {
        int ok = 0;
        for (int spins = 0; spins < c.spin_limit; ++spins) {
            if (__hip_atomic_load((ev_gu32*)(c.flags + gi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok) atomicAdd(c.ctrl + 2, 1u);
        *word = ok;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // (readfirstlane: the result steers wave-uniform control flow — chunk ranges, hence descriptor offsets of the fragment loads;
    // read as a plain LDS value it made hipcc treat all of that as divergent and wrap every buffer load of the K loops in a waterfall loop)
    const bool ok = __builtin_amdgcn_readfirstlane(*word) != 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // (the word may be rewritten by the next wait)
    return ok;
}
// Work stealing (round 4).  An owner used to wait — up to the spin limit, ~4.5 ms — for a contributor that was not even RESIDENT yet: with two
// pipelines in flight the vocoder's workgroups hold CU slots and about one balanced launch in ten had such a wait run out (then the owner
// recomputed the share: duplicated work after a long stall).  Now every workgroup announces itself (`sk_claim_start`: one atomic maximum of
// 2 seq at kernel start) and an owner that finds a contributor's word below 2 seq takes the share over at once (compare-and-swap to 2 seq + 1)
// and computes it itself, as a separate partial sum — the bits the contributor would have delivered; the contributor, when it finally starts,
// sees 2 seq + 1 and skips that share.  Every race resolves to either "wait for a contributor that has started" or "compute it here": never a
// different result, never a hang.  (A replayed hipGraph repeats seq: stale marks then only cost efficiency — both sides still agree.)
__device__ __forceinline__ unsigned sk_claim_start(const SkCtl& c, int g, int tid) {    // -> previous value of the word (lane 0 of wave 0 only)
    unsigned prev = 0;
    if (c.claims && tid == 0) prev = __hip_atomic_fetch_max((ev_gu32*)(c.claims + g), 2u * c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prev;
}
// contributor side: has this workgroup's first share (the part of a tile that belongs to another owner) been taken over?  All threads call.
__device__ __forceinline__ bool sk_claim_taken(const SkCtl& c, unsigned prev, int tid, int* word) {
    if (!c.claims) return false;
    if (tid == 0) *word = (prev == 2u * c.seq + 1u) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool taken = __builtin_amdgcn_readfirstlane(*word) != 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // (the word may be rewritten by a later wait)
    return taken;
}
// The same for the n (<= 64) consecutive workgroups gi .. gi + n - 1: lane k of wave 0 polls flag gi + k, so the flags are read in parallel
// and ONE acquire covers them all.  Returns how many LEADING flags were up when the polling ended (n; fewer when the first missing contributor
// has been taken over — no wait — or after the spin limit).
__device__ __forceinline__ int sk_wait_many(const SkCtl& c, int gi, int n, unsigned tag, int tid, int* word) {
    if (tid < 64) {                                    // wave 0 (a whole wave: scalar branch)
        unsigned long long up = 0, taken = 0;
        const unsigned long long want = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        if (c.claims) {                                // which contributors have not started: take their shares over
            bool mine = false;
            if (tid < n) {
                ev_gu32* w = (ev_gu32*)(c.claims + gi + tid);
                const unsigned started = 2u * c.seq, stolen = started + 1u;
                unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == stolen) mine = true;          // (taken over by an earlier call of this owner)
                else if (v != started) {
                    unsigned expect = v;
                    mine = __hip_atomic_compare_exchange_strong(w, &expect, stolen, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || expect == stolen;
                }
            }
            taken = __ballot(mine);
        }
        for (int spins = 0; spins < c.spin_limit; ++spins) {
            const unsigned v = tid < n ? __hip_atomic_load((ev_gu32*)(c.flags + gi + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            up = __ballot(v == tag) & ~taken;          // (a taken-over contributor never counts as delivered: its share is computed here)
            const unsigned long long miss = ~up & want;
            if (!miss || ((taken >> __builtin_ctzll(miss)) & 1ull)) break;     // all there, or the first missing one is ours to compute
            __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned long long miss = ~up & want;
        const int ready = miss ? __builtin_ctzll(miss) : n;
        if (tid == 0) {
            if (ready < n) atomicAdd(c.ctrl + (((taken >> ready) & 1ull) ? 3 : 2), 1u);    // [3] shares taken over, [2] waits that ran out
            *word = ready;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int ready = __builtin_amdgcn_readfirstlane(*word);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return ready;
}
// end of the launch: the last workgroup to arrive advances the epoch (next launch: tag + 1) and re-arms the counter
__device__ __forceinline__ void sk_arrive(const SkCtl& c, unsigned tag, int tid) {
    if (!c.ctrl || tid != 0) return;
    const unsigned old = __hip_atomic_fetch_add((ev_gu32*)(c.ctrl + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gridDim.x) {
        __hip_atomic_store((ev_gu32*)(c.ctrl + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((ev_gu32*)c.ctrl, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ bool evh_is_finite(float m);
// ---------------------------------------------------------------------------
// Tile maxima from the producer.  The fp16 builds need max |x| over the rows a tile stages before they can split
// them; conv_h16_kernel used to find it with a pre-scan — a second read of every X tile (198 GB of HBM traffic per config-2 step against
// 156 GB before the fp16 builds, 5-10 % of a 3-tap layer's time).  Every producer on the vocoder's chain already holds the values in its
// accumulators: it now leaves an UPPER BOUND of |y| per 128-row granule of its output (the maximum of |acc| over the wave tile's frames on
// either side of the granule boundary it may straddle, plus the residual's and the running sum's own bounds — triangle inequality; leaky-relu,
// masks and the / 3 of the resblock mean only shrink values), merged into the slot with an atomic maximum on the fp32 bits.  A bound, not the
// maximum: the tile scale only needs max * scale < 65504, and the two-piece form keeps fp32-grade precision while the bound is within 2^17 of
// the truth (error bound) — here it is within a small factor.  Slots are zeroed once per ev_hifigan call; a buffer rewritten
// within the call keeps the larger of its generations' bounds.  An Inf in the accumulators marks the slot non-finite: the consumer then
// repeats the search over the finite values of its tile (the pre-scan, kept as that slow path); NaNs never reach a maximum.
// ---------------------------------------------------------------------------
// The residual's and the running sum's bounds are requested at the START of the kernel (wave-uniform addresses: four loads whose latency
// the K loop hides), the per-lane maxima are taken from the accumulators in front of the epilogue, and the wave's reduction + atomics
// come AFTER the epilogue, where nothing waits for them.  (First version: loads, reduction and atomics together in front of the
// epilogue — the narrow fused pairs, whose whole launch is a few microseconds per workgroup, ran 5-18 % slower: profiles/r04_amax_ab.txt.)
struct EvAmax { float rlo, rhi, ylo, yhi, mlo, mhi; };
__device__ __forceinline__ EvAmax ev_amax_begin(const ConvParams& p, int nw0, int nframes) {
    EvAmax a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!p.ymax || p.ymax_mul != 1) return a;
    const int first = nw0 > 0 ? nw0 : 0;
    if (first >= p.nrows || nw0 + nframes <= 0) return a;
    const int g0 = first >> 7;
    if (p.rmax) { a.rlo = __uint_as_float(p.rmax[g0]); a.rhi = __uint_as_float(p.rmax[g0 + 1]); }
    if (p.yold) { a.ylo = __uint_as_float(p.yold[g0]); a.yhi = __uint_as_float(p.yold[g0 + 1]); }
    return a;
}
__device__ __forceinline__ void ev_amax_emit(const ConvParams& p, const EvAmax& a, int nw0, int nframes, int lane) {
    if (!p.ymax) return;
    // a.mlo / a.mhi: this lane's maxima of |acc| over the wave tile's frames below / from the granule boundary bnd = ((max(nw0, 0) >> 7) + 1) << 7
    float mlo = a.mlo, mhi = a.mhi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mlo = fmaxf(mlo, __shfl_xor(mlo, o, 64)); mhi = fmaxf(mhi, __shfl_xor(mhi, o, 64)); }
    if (lane != 0) return;
    const int first = nw0 > 0 ? nw0 : 0, last = nw0 + nframes - 1;
    if (last < 0 || first >= p.nrows) return;
    auto bits = [](float b) -> unsigned { return evh_is_finite(b) ? __float_as_uint(b * 1.000001f) : 0x7f800000u; };   // (rounded up: the sums below round to nearest)
    if (p.ymax_mul == 1) {
        const int g0 = first >> 7;
        atomicMax(p.ymax + g0, bits(mlo + a.rlo + a.ylo));
        if ((last >> 7) > g0) atomicMax(p.ymax + g0 + 1, bits(mhi + a.rhi + a.yhi));
    } else {                                            // a polyphase transposed conv: row n holds ymax_mul output frames — one bound for all of them
        const unsigned b = bits(fmaxf(mlo, mhi));
        const int ga = (first * p.ymax_mul) >> 7, gb = ((last + 1) * p.ymax_mul - 1) >> 7;
        for (int g = ga; g <= gb; ++g) atomicMax(p.ymax + g, b);
    }
}
// this lane's share of the maxima for the two accumulator layouts (frames = columns of the C/D tiles)
template <int TM, int TN>
__device__ __forceinline__ void ev_amax_from_acc(const ConvParams& p, EvAmax& a, const f32x16 (&acc)[TM][TN], int nw0, int lane) {
    if (!p.ymax) return;
    const int bnd = (((nw0 > 0 ? nw0 : 0) >> 7) + 1) << 7;
    float mlo = 0.f, mhi = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) m = fmaxf(m, fmaxf(fabsf(acc[i][j][r]), fabsf(acc[i][j][r + 1])));
        const bool lo = nw0 + j * 32 + (lane & 31) < bnd;
        mlo = fmaxf(mlo, lo ? m : 0.f); mhi = fmaxf(mhi, lo ? 0.f : m);
    }
    a.mlo = mlo; a.mhi = mhi;
}
template <int QM, int QN>
__device__ __forceinline__ void ev_amax_from_acc_q(const ConvParams& p, EvAmax& a, const f32x4 (&acc)[QM][QN], int nw0, int lane) {
    if (!p.ymax) return;
    const int bnd = (((nw0 > 0 ? nw0 : 0) >> 7) + 1) << 7;
    float mlo = 0.f, mhi = 0.f;
#pragma unroll
    for (int b = 0; b < QN; ++b) {
        float m = 0.f;
#pragma unroll
        for (int aa = 0; aa < QM; ++aa) m = fmaxf(m, fmaxf(fmaxf(fabsf(acc[aa][b][0]), fabsf(acc[aa][b][1])), fmaxf(fabsf(acc[aa][b][2]), fabsf(acc[aa][b][3]))));
        const bool lo = nw0 + b * 16 + (lane & 15) < bnd;
        mlo = fmaxf(mlo, lo ? m : 0.f); mhi = fmaxf(mhi, lo ? 0.f : m);
    }
    a.mlo = mlo; a.mhi = mhi;
}
// consumer: the bound of everything a tile stages (rows [row_lo, row_hi] of X), or a negative value when the slots cannot be used
__device__ __forceinline__ float ev_amax_read(const ConvParams& p, int row_lo, int row_hi, int lane) {
    int ga = (row_lo > 0 ? row_lo : 0) >> 7, gb = row_hi >> 7;
    gb = gb < p.xmax_n - 1 ? gb : p.xmax_n - 1;
    float m = 0.f;
    for (int g = ga + (lane & 7); g <= gb; g += 8) m = fmaxf(m, __uint_as_float(p.xmax[g]));    // (at most three granules for a 128-row tile: one pass)
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
}

// KB = 32-channel k-chunks staged per barrier pair (LDS row = 32*KB + 4 floats).  KB = 2 halves the number of
// stage / barrier episodes — what the 1x1 and k=3 layers need (a 1x1 conv has only 4 k-groups = 32 MFMAs per wave
// between two barrier pairs at KB = 1); wide-halo layers (k = 11, d = 5) already run 352 MFMAs per chunk and keep KB = 1
// for its smaller LDS tile (more workgroups per CU).  The accumulation order (chunk, tap, k-group) is the same for both.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool PF = false, bool FULL_ACT = true, int LEAN = 0, int KB = 1>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int LDK = 32 * KB + 4;                   // LDS row stride (floats): 36 / 68 are both conflict-free for ds_read_b128
    static_assert(!PF || KB == 1, "the register-prefetch build stages one 32-channel chunk at a time");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                  // [(BN + EV_HALO)][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: keeps fragment offsets in SGPRs
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.mtiles * p.ntiles;
    const int work = ev_xcd_remap(blockIdx.x, nwg);
    const int mt = work % p.mtiles;
    const int nt = work / p.mtiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    // tiles that contain no storable row (pure padding) do nothing
    {
        int t_first = (n0 % p.S) - p.P;  // may be negative
        // first valid row at or after n0: if t_first in [0,T) -> valid; else next utterance start
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - (n0 % p.S) + p.P;
        if (dist >= BN || n0 + dist >= p.nrows) return;
    }

    // active (non-zero) taps of this M tile: a host-built compact list read with scalar loads
    const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
    const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mt] : p.ntaps);
    if ((p.dbg & 16) && threadIdx.x == 0) p.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();

    // Start stagger.  All tiles of a launch cost the same, so the workgroups that share a CU (dispatched together at
    // t = 0) would otherwise stay in lockstep for the whole launch: every CU runs its MFMA phases at the same time and
    // then its memory-bound epilogues at the same time.  Delaying the k-th co-resident workgroup of the FIRST wave of
    // workgroups by k/slots of one tile time puts (and keeps) them out of phase: one streams its epilogue while the
    // others feed the matrix pipe.
    if (p.stagger_slots > 1 && (int)blockIdx.x < 256 * p.stagger_slots) {
        const int slot = blockIdx.x >> 8;
        const long tile_cycles = (long)(p.Kpad / EV_BK) * nact * (16L * TM * TN * 64) * p.stagger_slots;
        const long wait = tile_cycles * slot / p.stagger_slots;
        const long long t0 = __builtin_amdgcn_s_memtime();
        while ((long)(__builtin_amdgcn_s_memtime() - t0) < wait) __builtin_amdgcn_s_sleep(32);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bq[g] = z;
            if constexpr (LEAN != 0) {   // bias preloaded into the accumulators: C/D row of register 4g+e is 8g + 4*half + e
                const int c0 = m0 + wm * (TM * 32) + a * 32 + 8 * g + 4 * lh;
                if (p.bias && c0 < p.Cout) bq[g] = *(const f32x4*)(p.bias + c0);
            }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
    }

    const int xrows = BN + p.halo_lo + p.halo_hi;
    constexpr int TPR = 8 * KB, RPS = 256 / TPR;   // staging: threads per row, rows per pass
    const int srow = tid / TPR;         // staging row within a pass
    const int sc4 = (tid % TPR) * 4;    // staging column (floats)
    const int nchunks = p.Kpad / EV_BK;

    // A operand (weights) never touches LDS: the host packs them in MFMA-fragment order
    //   Wf[tap][row-tile of 32][k-group of 8][lane 64][4]   with lane = (row & 31) + 32*half, element s <-> k = 8*kg + 4*half + s,
    // so one global_load_dwordx4 per wave reads a contiguous 1 KiB (L2-resident) fragment feeding four MFMAs per
    // 32-row tile.  Fragments are prefetched one k-group ahead, across taps and k-chunks, so the waves of a
    // workgroup only meet at the two barriers around each X-tile (activation) stage.
    constexpr int XPASS = (BN + EV_HALO) / RPS;
    constexpr int XG = XPASS > 8 ? 8 : XPASS;      // passes per load batch (bounds the registers in flight)
    const int mt32 = (m0 + wm * (TM * 32)) >> 5;          // first 32-row tile of this wave
    const int MT32 = p.Mpad >> 5, KG8 = p.Kpad >> 3;
    const __amdgpu_buffer_rsrc_t rW = ev_rsrc(p.W), rX = ev_rsrc(p.X);
    const unsigned wlane = (unsigned)lane * 16u;           // per-lane byte offset inside a 1 KiB fragment
    const unsigned wbase = (unsigned)(mt32 * KG8) * 1024u;
    (void)MT32;
    auto a_off = [&](int tap_bytes, int kg8) -> unsigned {  // byte offset (wave-uniform) of row-tile mt32's fragment at k-group kg8;
        return (unsigned)tap_bytes + wbase + (unsigned)kg8 * 1024u;   // the tap list carries tap * (Mpad/32) * (Kpad/8) * 1 KiB
    };
    // Two statically named fragment sets (A0/B0 for even k-groups, A1/B1 for odd ones): the loads for k-group g+1
    // are issued before the MFMAs of k-group g and first touched one k-group later — no register copies, so the
    // compiler's s_waitcnt lands at the first use and L2 / LDS latency hides under 16 MFMAs.
    f32x4 A0[TM], A1[TM], B0[TN], B1[TN];
    // (no conditional loads anywhere in the K loop: a load under a branch — even a wave-uniform one — makes hipcc's waitcnt
    // insertion give up counting and wait vmcnt(0) at the next use, which drains the whole fragment pipeline)
    auto ldAp = [&](f32x4 (&dst)[TM], unsigned aoff) {
#ifdef EV_ABLATE_A_LOADS
        if (p.dbg & 256) return;                         // tools/conv_bench.py ablation build: no weight-fragment loads (timing only)
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG8 * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[TN], const float* brow, int kg) {
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[j] = *(const f32x4*)(brow + j * 32 * LDK + kg * 8);
    };
    auto mma = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s4], b[j][s4], acc[i][j], 0, 0, 0);
    };
    const float* bbase = Xs + (wn * (TN * 32) + li + p.halo_lo) * LDK + 4 * lh;
    const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
    const int2 tv_first = ev_tap_at(tlv, 0);
    // Weight fragments run one whole tap (four k-groups) ahead of the MFMAs in four statically named sets: measured with
    // tools/conv_bench.py, a one-k-group-deep pipeline left 6-12 % on the table on every deep layer (and halving the L2 stream
    // with a 4 x 1 wave layout recovered 1 % of it: fragment LATENCY under load, not bandwidth).
    f32x4 A2[TM], A3[TM];
    if (nact > 0) {
        const unsigned a0 = a_off(tv_first.x, 0);
        ldAp(A0, a0); ldAp(A1, a0 + 1024u); ldAp(A2, a0 + 2048u); ldAp(A3, a0 + 3072u);
    }
    // ---- X staging.  Per-pass row offsets are fixed for the tile (xoff), the chunk's column base is a scalar (soffset), rows
    // outside the tensor read the all-zero pad row 0 and passes beyond the tile's rows re-read that row: every load is
    // UNCONDITIONAL (see the note at ldAp).  PF: the loads of chunk c+1 are issued right after chunk c's tile has been
    // published, fly under chunk c's MFMAs and are written to LDS at the next stage, so the stage between the two barriers is
    // LDS traffic only.  vmcnt retires in order, so the first fragment wait that also has to wait for these loads is the one
    // for fragments requested after them — a whole tap (>= 64 MFMAs) later, thanks to the fragment pipeline above.
    unsigned xoff[XPASS];
    const int npass = (xrows + RPS - 1) / RPS;          // passes that carry rows of this tile (uniform)
#pragma unroll
    for (int q = 0; q < XPASS; ++q) {
        const int gr = n0 - p.halo_lo + q * RPS + srow;
        xoff[q] = ((q < npass && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
    }
    auto x_soff = [&](int ch) -> unsigned {
        const int c0 = ch * EV_BK;
        return ((unsigned)(c0 >> p.isplit_log2) * p.isstride + (unsigned)(c0 & ((1 << p.isplit_log2) - 1))) * 4u;
    };
    f32x4 xv[PF ? XPASS : 1];
    auto x_issue = [&](int ch) {                        // PF only: all passes of one chunk into registers
        const unsigned soff = x_soff(ch);
#pragma unroll
        for (int q = 0; q < XPASS; ++q) xv[PF ? q : 0] = ev_bload4(rX, xoff[q], soff);
    };
    auto x_put = [&](int q, f32x4 v, int c0, bool ctail) {
        const int r = q * RPS + srow;
        if (ctail && c0 + sc4 >= p.Cin) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
        if (p.pro_lrelu) {
            v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
            v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
        }
        if (r < xrows) *(f32x4*)(Xs + r * LDK + sc4) = v;
    };
    if constexpr (PF) x_issue(0);
    unsigned long long ts0 = 0, acc_st = 0, acc_b1 = 0;   // dbg 2 (+16): accumulated stage / first-barrier time of this wave
    for (int ch = 0; ch < nchunks; ++ch) {
        // Memory phases (X staging, epilogue) issue few instructions but were measured to stretch 2-3x when the other
        // workgroups of the CU are in their MFMA loops (issue arbitration favours the older, MFMA-issuing waves): run
        // them at raised priority so a workgroup gets back to feeding the matrix pipe sooner.
        const int sub = ch & (KB - 1);
        if (sub == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (p.dbg & 2) ts0 = __builtin_amdgcn_s_memrealtime();
        if (!(p.dbg & 8)) ev_lds_barrier();  // previous chunk's MFMAs are done with Xs (dbg 8: timing-only ablation without barriers)
        if (p.dbg & 2) { const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(); acc_b1 += t1 - ts0; }
        {
            const int c0 = ch * EV_BK;
            const bool ctail = (c0 + 32 * KB > p.Cin);       // uniform: only the last chunk of a Cin that is not a multiple of 32
            if constexpr (PF) {
#pragma unroll
                for (int q = 0; q < XPASS; ++q) x_put(q, xv[PF ? q : 0], c0, ctail);
            } else {
                const unsigned soff = x_soff(ch);
#pragma unroll
                for (int q0 = 0; q0 < XPASS; q0 += XG) {
                    if (q0 * RPS >= xrows) continue;
                    f32x4 xg[XG];
#pragma unroll
                    for (int q = 0; q < XG; ++q) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if ((q0 + q) * RPS < xrows && !(p.dbg & 1)) v = ev_bload4(rX, xoff[q0 + q], soff);
                        xg[q] = v;
                    }
#pragma unroll
                    for (int q = 0; q < XG; ++q) x_put(q0 + q, xg[q], c0, ctail);
                }
            }
        }
        if (!(p.dbg & 8)) ev_lds_barrier();
        __builtin_amdgcn_s_setprio(0);
        if (p.dbg & 2) acc_st += __builtin_amdgcn_s_memrealtime() - ts0;
        if constexpr (PF) x_issue(ch + 1 < nchunks ? ch + 1 : ch);   // (after the last chunk: a harmless re-read)
        }
        const float* bsub = bbase + sub * 32;
        if ((p.dbg & 18) == 16 && ch == 0 && threadIdx.x == 0) p.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        int tap = tv_first.x;
        const float* brow = bsub + tv_first.y * LDK;
        if (sub == 0) ldB(B0, brow, 0);                     // (later sub-chunks: prefetched by the last tap of the previous one)
        for (int ti = 0; ti < nact; ++ti) {
            const bool last_tap = (ti + 1 == nact);
            const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
            const int ntap = ntv.x;
            const float* nbrow = bsub + ntv.y * LDK;
            const bool have_next = !(last_tap && ch + 1 == nchunks);
            // fragments of the next tap (k-groups +1 KiB each); after the very last tap: a harmless re-read of the first fragments
            const unsigned nap = have_next ? a_off(ntap, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, 0);
            // sched_barrier pins "loads, then the 16 MFMAs of a k-group": hipcc otherwise sinks the prefetches into the MFMA
            // block and waits for them a few MFMAs later.  Set g is refilled with k-group g of the NEXT tap right after its MFMAs.
            ldB(B1, brow, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(A0, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A0, nap);
            ldB(B0, brow, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(A1, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A1, nap + 1024u);
            ldB(B1, brow, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(A2, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A2, nap + 2048u);
            // first B fragments of the next tap (after the last tap of a chunk: a harmless read of the tile being retired,
            // or — KB > 1 — the next sub-chunk, which is already in LDS)
            ldB(B0, (KB > 1 && last_tap && sub + 1 < KB && ch + 1 < nchunks) ? nbrow + 32 : nbrow, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(A3, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A3, nap + 3072u);
            (void)tap;
            tap = ntap; brow = nbrow;
        }
    }
    if ((p.dbg & 18) == 18 && threadIdx.x == 0) {   // stamps 1 / 2 become start + accumulated stage time / first-barrier wait
        p.stamps[4 * blockIdx.x + 1] = p.stamps[4 * blockIdx.x] + acc_st;
        p.stamps[4 * blockIdx.x + 2] = p.stamps[4 * blockIdx.x] + acc_b1;
    }

    __builtin_amdgcn_s_setprio(3);   // epilogue: see the note on memory phases above
    if ((p.dbg & 18) == 16 && threadIdx.x == 0) p.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    if (p.dbg & 4) { if (acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }   // tools/conv_bench.py ablation: no epilogue
    if constexpr (LEAN == 1 || LEAN == 3) {                // (host: ymax only with these epilogues)
        EvAmax am = ev_amax_begin(p, n0 + wn * (TN * 32), TN * 32);
        ev_amax_from_acc<TM, TN>(p, am, acc, n0 + wn * (TN * 32), lane);
        conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
        ev_amax_emit(p, am, n0 + wn * (TN * 32), TN * 32, lane);
    } else if constexpr (LEAN != 0) conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
    else conv_epilogue<TM, TN, FULL_ACT>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
    if ((p.dbg & 16) && threadIdx.x == 0) p.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}

// ---------------------------------------------------------------------------
// conv_split_kernel: the same implicit GEMM on the bf16 matrix pipe, fp32 in and out.
//
// The fp32 MFMA of gfx950 runs at the vector rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 runs sixteen times faster and
// accumulates in fp32.  Every fp32 value is the EXACT sum of three bf16 values (8 + 8 + 8 significand bits: x0 = the upper half
// of the word, x1 = the upper half of x - x0, x2 = x - x0 - x1), and a bf16 x bf16 product is exact in the fp32 accumulator, so
//     a . b  =  sum over pieces  a_p . b_q
// with the six products of weight p + q <= 2 (a0b0, a0b1, a1b0, a0b2, a1b1, a2b0) leaves out only terms below 2^-24 |a b| — the
// size of ONE fp32 rounding.  Measured (tools/bf16_split_probe.hip, dot products of length 1408 against fp64): fp32 MFMA max
// 3.1e-6 / rms 6.9e-7 of the result's scale, six products 3.4e-6 / 5.9e-7, nine products the same, three products 8.9e-5: six it
// is.  Cost: 6 MFMAs of 32 cycles per 16-deep slab against 8 of 64 — 2.67x the rate of the fp32 pipe (probe: 365 TFLOP/s of
// fp32-equivalent work in a loop fed from LDS and L2, against 153).
//   weights:     split once by the loader, fragment order [tap][32-row tile][16-deep slab][piece][lane][8 bf16]
//                (lane = (row & 31) + 32 * half, element e <-> k = 16 slab + 8 half + e): one 16-byte load per lane and piece
//   activations: split while they are staged: LDS row = three bf16 planes of the 64-channel chunk (+ 16 bytes: an odd multiple of
//                16 keeps the 16-byte fragment reads conflict-free); a chunk is four slabs, so the fragment ring of the K loop
//                (four statically named sets, one tap ahead) carries over unchanged from conv_gemm_kernel
//   epilogue:    the D layout of the 32 x 32 tile is that of the fp32 instruction: conv_epilogue_lean as is
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define EVX_KC 64                           // input channels per LDS stage
#define EVX_RSB (3 * EVX_KC * 2 + 16)       // LDS row stride in bytes (400 = 25 x 16)
// byte offset of the 64-channel chunk `ch` inside an input row: channel c lives at (c >> isplit_log2) * isstride + (c & (2^isplit_log2 - 1))
// (the pair view of a strided slice: the stride-2 down conv reads rows of 2 C channels whose halves lie isstride apart; no split: 2^31)
__device__ __forceinline__ unsigned evx_chunk_off(const ConvParams& p, int ch) {
    const int c0 = ch * EVX_KC;
    return ((unsigned)(c0 >> p.isplit_log2) * (unsigned)p.isstride + (unsigned)(c0 & ((1 << p.isplit_log2) - 1))) * 4u;
}
// three bf16 pieces of four fp32 values, packed as the four channels' bf16 in channel order (8 bytes per piece)
__device__ __forceinline__ void evx_split4(const f32x4 v, uint2& q0, uint2& q1, uint2& q2) {
    unsigned u[4], w[4];
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { u[e] = __float_as_uint(v[e]); r[e] = v[e] - __uint_as_float(u[e] & 0xffff0000u); w[e] = __float_as_uint(r[e]); }
    q0.x = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u); q0.y = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
    q1.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u); q1.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = __float_as_uint(r[e] - __uint_as_float(w[e] & 0xffff0000u));
    q2.x = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u); q2.y = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
}
// test hook (ev_op_split_pieces): the three pieces of every element, as fp32 values, so that a test can check p0 + p1 + p2 == x bit for bit
// and that each piece is a bf16 value
__global__ void evx_split_check_kernel(const float* x, float* pieces /*[3][n]*/, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = {x[i], i + 1 < n ? x[i + 1] : 0.f, i + 2 < n ? x[i + 2] : 0.f, i + 3 < n ? x[i + 3] : 0.f};
    uint2 q0, q1, q2;
    evx_split4(v, q0, q1, q2);
    const unsigned w[3][2] = {{q0.x, q0.y}, {q1.x, q1.y}, {q2.x, q2.y}};
    for (int pc = 0; pc < 3; ++pc)
        for (int e = 0; e < 4; ++e)
            if (i + e < n) pieces[(size_t)pc * n + i + e] = __uint_as_float(((w[pc][e >> 1] >> (16 * (e & 1))) & 0xffffu) << 16);
}

template <int TM, int TN, int TERMS>
__device__ __forceinline__ void evx_mma(f32x16 (&acc)[TM][TN], const f32x4 (&a)[3][TM], const f32x4 (&b)[3][TN]) {
    // smallest products first; the accumulators of the TM x TN tiles alternate, so consecutive MFMAs are independent
    constexpr int PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 9 - TERMS; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[t]][i]), __builtin_bit_cast(bf16x8, b[PB[t]][j]), acc[i][j], 0, 0, 0);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int LEAN, int TERMS = 6>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
    static_assert(TERMS == 3 || TERMS == 6 || TERMS == 9, "products per element pair");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                            // [(BN + halo)][EVX_RSB bytes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.mtiles * p.ntiles;
    const int work = ev_xcd_remap(blockIdx.x, nwg);
    const int mt = work % p.mtiles;
    const int nt = work / p.mtiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    {   // tiles that contain no storable row (pure padding) do nothing
        const int t_first = (n0 % p.S) - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - (n0 % p.S) + p.P;
        if (dist >= BN || n0 + dist >= p.nrows) return;
    }
    const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
    const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mt] : p.ntaps);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bq[g] = z;
            const int c0 = m0 + wm * (TM * 32) + a * 32 + 8 * g + 4 * lh;
            if (p.bias && c0 < p.Cout) bq[g] = *(const f32x4*)(p.bias + c0);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
    }

    const int xrows = BN + p.halo_lo + p.halo_hi;
    constexpr int TPR = EVX_KC / 4, RPS = 256 / TPR;   // staging: 16 threads per row, 16 rows per pass
    const int srow = tid / TPR;
    const int sc4 = (tid % TPR) * 4;
    const int nchunks = p.Kpad / EVX_KC;
    constexpr int XPASS = (BN + EV_HALO) / RPS;
    constexpr int XG = 6;
    static_assert(XPASS % XG == 0, "staging batches");
    const int mt32 = (m0 + wm * (TM * 32)) >> 5;
    const int KG16 = p.Kpad >> 4;
    const __amdgpu_buffer_rsrc_t rW = ev_rsrc(p.Wx), rX = ev_rsrc(p.X);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(mt32 * KG16) * 3072u;
    // the tap list carries byte offsets of the fp32 planes ((Mpad/32) (Kpad/8) KiB per tap); a split plane is 1.5 x that
    auto a_off = [&](int tap_bytes, int kg16) -> unsigned { return (unsigned)tap_bytes + ((unsigned)tap_bytes >> 1) + wbase + (unsigned)kg16 * 3072u; };
    f32x4 A0[3][TM], A1[3][TM], A2[3][TM], A3[3][TM], B0[3][TN], B1[3][TN];
    auto ldAp = [&](f32x4 (&dst)[3][TM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int i = 0; i < TM; ++i) dst[pc][i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG16 * 3072 + pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[3][TN], const char* brow, int slab) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * EVX_RSB + pc * (EVX_KC * 2) + slab * 32);
    };
    const char* bbase = Xb + (wn * (TN * 32) + li + p.halo_lo) * EVX_RSB + 16 * lh;
    const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
    const int2 tv_first = ev_tap_at(tlv, 0);
    if (nact > 0) {
        const unsigned a0 = a_off(tv_first.x, 0);
        ldAp(A0, a0); ldAp(A1, a0 + 3072u); ldAp(A2, a0 + 6144u); ldAp(A3, a0 + 9216u);
    }
    unsigned xoff[XPASS];
#pragma unroll
    for (int q = 0; q < XPASS; ++q) {
        const int gr = n0 - p.halo_lo + q * RPS + srow;
        xoff[q] = ((q * RPS < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
    }
    for (int ch = 0; ch < nchunks; ++ch) {
        __builtin_amdgcn_s_setprio(3);
        if (!(p.dbg & 8)) ev_lds_barrier();             // the previous chunk's MFMAs are done with the tile
        if (!(p.dbg & 64) || ch == 0) {                 // (dbg 64: stage the first chunk only — timing ablation)
            const unsigned soff = evx_chunk_off(p, ch);
#pragma unroll
            for (int q0 = 0; q0 < XPASS; q0 += XG) {
                if (q0 * RPS >= xrows) continue;
                f32x4 xg[XG];
#pragma unroll
                for (int q = 0; q < XG; ++q) xg[q] = ev_bload4(rX, xoff[q0 + q], soff);      // (passes beyond the tile re-read row 0)
#pragma unroll
                for (int q = 0; q < XG; ++q) {
                    const int r = (q0 + q) * RPS + srow;
                    f32x4 v = xg[q];
                    if (p.pro_lrelu) {
                        v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                        v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                    }
                    uint2 q0v, q1v, q2v;
                    if (p.dbg & 32) { q0v.x = __float_as_uint(v[0]); q0v.y = __float_as_uint(v[1]); q1v = q0v; q2v = q0v; }   // tools/conv_bench.py ablation: no split arithmetic (timing only)
                    else evx_split4(v, q0v, q1v, q2v);
                    if (r < xrows) {
                        char* dst = Xb + r * EVX_RSB + sc4 * 2;
                        *(uint2*)(dst) = q0v; *(uint2*)(dst + EVX_KC * 2) = q1v; *(uint2*)(dst + EVX_KC * 4) = q2v;
                    }
                }
            }
        }
        if (!(p.dbg & 8)) ev_lds_barrier();
        __builtin_amdgcn_s_setprio(0);
        const char* brow = bbase + tv_first.y * EVX_RSB;
        ldB(B0, brow, 0);
        for (int ti = 0; ti < nact; ++ti) {
            const bool last_tap = (ti + 1 == nact);
            const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
            const char* nbrow = bbase + ntv.y * EVX_RSB;
            const bool have_next = !(last_tap && ch + 1 == nchunks);
            const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, 0);   // unconditional loads
            ldB(B1, brow, 1);
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, A0, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A0, nap);
            ldB(B0, brow, 2);
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, A1, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A1, nap + 3072u);
            ldB(B1, brow, 3);
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, A2, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A2, nap + 6144u);
            ldB(B0, nbrow, 0);                          // (after a chunk's last tap: a harmless read of the tile being retired)
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, A3, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A3, nap + 9216u);
            brow = nbrow;
        }
    }
    __builtin_amdgcn_s_setprio(3);
    if (p.dbg & 4) { if (acc[0][0][0] == 12345.678f) p.Y[0] = 1.f; return; }   // tools/conv_bench.py ablation: no epilogue
    conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
}

// ---------------------------------------------------------------------------
// conv_gemm_bal_kernel: conv_gemm_kernel for launches of about ONE ROUND of workgroups (the U-Net convs of a batch-64 decode), as
// a balanced persistent grid (SkCtl above): G workgroups, a unit = (tile, 32-channel k-chunk), workgroup g takes the contiguous
// unit range [start(g), start(g + 1)) in (tile-major, chunk-minor) order.  A tile whose chunks straddle workgroups is finished by
// the one that holds chunk 0 (it reaches that tile LAST; bias in its accumulators), the others reach their share of it FIRST and
// hand over an accumulator tile without bias; the owner adds them in ascending workgroup order.  Same K loop as conv_gemm_kernel
// (KB = 1, no register-prefetched staging), lean epilogues only; the summation order over k-chunks of a split tile differs from
// the one-tile-per-workgroup build (partial sums), deterministically for a given device.
// Measured motive (tools/shape_profile.py, 62 vs 64 utterances): 1040 tiles of 64 x 64 on 256 CUs leave the matrix pipes of three
// quarters of the chip idle for the last fifth of the launch (4.06 tiles per CU = five rounds on some CUs).
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int LEAN>
__global__ __launch_bounds__(256, 3) void conv_gemm_bal_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int LDK = 36;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1 && LEAN != 0, "balanced build: 4 waves, lean epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                  // [(BN + halo)][LDK]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.Kpad / EV_BK;
    const int g = blockIdx.x;
    const unsigned tag = sk_tag(p.sk);
    int u = sk_start(p.sk, g);
    const int ue = sk_start(p.sk, g + 1);
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(p.sk.part), rW = ev_rsrc(p.W), rX = ev_rsrc(p.X);
    const unsigned pslot = (unsigned)p.sk.part_floats * 8u;                  // bytes per workgroup: [published partial | private spill]
    constexpr int PN = TM * TN * 4;                                          // float4 of the accumulator tile per lane
    const unsigned pelem = (unsigned)(wave * PN) * 1024u + (unsigned)lane * 16u;
    int* skw = (int*)((char*)smem + p.sk.lds_word);                          // one word for sk_wait, behind everything else in LDS
    bool pend_pub = false;
    const unsigned wlane = (unsigned)lane * 16u;
    const int KG8 = p.Kpad >> 3;
    constexpr int TPR = 8, RPS = 256 / TPR;
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;
    constexpr int XPASS = (BN + EV_HALO) / RPS;
    constexpr int XG = XPASS > 8 ? 8 : XPASS;
    const int xrows = BN + p.halo_lo + p.halo_hi;
    const int npass = (xrows + RPS - 1) / RPS;

    while (u < ue) {
        const int t = u / nchunks, c0 = u - t * nchunks;
        const int c1 = (ue - u < nchunks - c0) ? c0 + (ue - u) : nchunks;
        u += c1 - c0;
        const int mt = t % p.mtiles, nt = t / p.mtiles;
        const int m0 = mt * BM, n0 = nt * BN;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= BN || n0 + dist >= p.nrows) continue;
        }
        if (pend_pub) { sk_publish(p.sk, g, tag, tid); pend_pub = false; }   // the previous segment's partial tile: drain, then raise its flag
        const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
        const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mt] : p.ntaps);
        const int mt32 = (m0 + wm * (TM * 32)) >> 5;
        const unsigned wbase = (unsigned)(mt32 * KG8) * 1024u;
        auto a_off = [&](int tap_bytes, int kg8) -> unsigned { return (unsigned)tap_bytes + wbase + (unsigned)kg8 * 1024u; };
        f32x4 A0[TM], A1[TM], A2[TM], A3[TM], B0[TN], B1[TN];
        auto ldAp = [&](f32x4 (&dst)[TM], unsigned aoff) {
#pragma unroll
            for (int i = 0; i < TM; ++i) dst[i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG8 * 1024));
        };
        auto ldB = [&](f32x4 (&dst)[TN], const float* brow, int kg) {
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[j] = *(const f32x4*)(brow + j * 32 * LDK + kg * 8);
        };
        const float* bbase = Xs + (wn * (TN * 32) + li + p.halo_lo) * LDK + 4 * lh;
        const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
        const int2 tv_first = ev_tap_at(tlv, 0);
        unsigned xoff[XPASS];
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int gr = n0 - p.halo_lo + q * RPS + srow;
            xoff[q] = ((q < npass && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
        }
        auto x_soff = [&](int ch) -> unsigned {
            const int cc = ch * EV_BK;
            return ((unsigned)(cc >> p.isplit_log2) * p.isstride + (unsigned)(cc & ((1 << p.isplit_log2) - 1))) * 4u;
        };
        auto x_put = [&](int q, f32x4 v, int cc, bool ctail) {
            const int r = q * RPS + srow;
            if (ctail && cc + sc4 >= p.Cin) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
            if (p.pro_lrelu) {
                v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
            }
            if (r < xrows) *(f32x4*)(Xs + r * LDK + sc4) = v;
        };

        // ---- passes over chunk ranges of this tile (see ln_mlp_kernel): [c0, c1) first; an owner whose wait for a contributor ran
        // out adds one pass over that contributor's range with fresh accumulators (the contributor's bits)
        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;
        const int tile_end = (t + 1) * nchunks;
        for (;;) {
            f32x16 acc[TM][TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    bq[q] = z;
                    const int cc = m0 + wm * (TM * 32) + a * 32 + 8 * q + 4 * lh;   // bias preloaded (lean epilogue), by the owner's first pass only
                    if (p.bias && cA == 0 && cc < p.Cout) bq[q] = *(const f32x4*)(p.bias + cc);
                }
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
            }
            auto mma = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s4], b[j][s4], acc[i][j], 0, 0, 0);
            };
            if (nact > 0) {
                const unsigned a0 = a_off(tv_first.x, cA * 4);
                ldAp(A0, a0); ldAp(A1, a0 + 1024u); ldAp(A2, a0 + 2048u); ldAp(A3, a0 + 3072u);
            }
            for (int ch = cA; ch < cB; ++ch) {
                __builtin_amdgcn_s_setprio(3);
                ev_lds_barrier();                          // the previous chunk's MFMAs (or the previous segment's epilogue) are done with LDS
                {
                    const int cc = ch * EV_BK;
                    const bool ctail = (cc + 32 > p.Cin);
                    const unsigned soff = x_soff(ch);
#pragma unroll
                    for (int q0 = 0; q0 < XPASS; q0 += XG) {
                        if (q0 * RPS >= xrows) continue;
                        f32x4 xg[XG];
#pragma unroll
                        for (int q = 0; q < XG; ++q) {
                            f32x4 v = {0.f, 0.f, 0.f, 0.f};
                            if ((q0 + q) * RPS < xrows) v = ev_bload4(rX, xoff[q0 + q], soff);
                            xg[q] = v;
                        }
#pragma unroll
                        for (int q = 0; q < XG; ++q) x_put(q0 + q, xg[q], cc, ctail);
                    }
                }
                ev_lds_barrier();
                __builtin_amdgcn_s_setprio(0);
                const float* brow = bbase + tv_first.y * LDK;
                ldB(B0, brow, 0);
                for (int ti = 0; ti < nact; ++ti) {
                    const bool last_tap = (ti + 1 == nact);
                    const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
                    const float* nbrow = bbase + ntv.y * LDK;
                    const bool have_next = !(last_tap && ch + 1 == cB);
                    const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, cA * 4);
                    ldB(B1, brow, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(A0, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A0, nap);
                    ldB(B0, brow, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(A1, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A1, nap + 1024u);
                    ldB(B1, brow, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(A2, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A2, nap + 2048u);
                    ldB(B0, nbrow, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(A3, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A3, nap + 3072u);
                    brow = nbrow;
                }
            }
            auto acc_io = [&](unsigned base, int mode) {    // mode 0: store (write-through), 1: add from memory
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned soff = base + (unsigned)((a * TN + b) * 4 + q) * 1024u;     // (wave-uniform: scalar offset)
                            if (mode == 0) {
                                const f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                                ev_bstore4_sc1(rPart, pelem, soff, v);
                            } else {
                                const f32x4 v = ev_bload4_sc1(rPart, pelem, soff);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[a][b][4 * q + e] += v[e];
                            }
                        }
            };
            if (spilled) { acc_io((unsigned)g * pslot + pslot / 2, 1); spilled = false; }   // running sum (spilled) + this contributor's share
            if (c0 != 0) {                                 // not the owner: hand the partial tile over (flag raised at the next segment)
                acc_io((unsigned)g * pslot, 0);
                pend_pub = true;
                break;
            }
            bool again = false;
            while (c1 < nchunks) {                         // owner: add the contributors' partial tiles in ascending workgroup order
                int n = 0;
                while (n < 64 && gi + n < (int)gridDim.x && sk_start(p.sk, gi + n) < tile_end) ++n;
                if (n == 0) break;
                const int ready = sk_wait_many(p.sk, gi, n, tag, tid, skw);
                for (int k = 0; k < ready; ++k) acc_io((unsigned)(gi + k) * pslot, 1);
                gi += ready;
                if (ready < n) {                           // gi is not there in time: spill the running sum, compute its share here
                    const int sgi = sk_start(p.sk, gi);
                    int egi = sk_start(p.sk, gi + 1);
                    egi = egi < tile_end ? egi : tile_end;
                    acc_io((unsigned)g * pslot + pslot / 2, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cA = sgi - t * nchunks; cB = egi - t * nchunks;
                    spilled = true; again = true;
                    ++gi;
                    break;
                }
            }
            if (again) continue;
            __builtin_amdgcn_s_setprio(3);
            conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
            __builtin_amdgcn_s_setprio(0);
            break;
        }
    }
    if (pend_pub) sk_publish(p.sk, g, tag, tid);
    sk_arrive(p.sk, tag, tid);
}

// ---------------------------------------------------------------------------
// conv_h16_kernel: conv_split_kernel with fp16 pieces.  An fp16 value carries 11 significand bits, so TWO pieces (x = h0 + h1, h0 = the
// nearest fp16, h1 = the nearest fp16 of the rest) represent an fp32 operand to 2^-22..2^-23, and the three products h0 g0 + h0 g1 + h1 g0
// (what is left out is below 2^-22 |a b|) give dot products whose error against fp64 is that of the fp32 FMA chain — measured (in-run probe,
// length 1408): max 2.4e-6 / rms 4.3e-7 against 3.1e-6 / 6.9e-7 — for HALF the MFMAs of the bf16 form.  What fp16 lacks is RANGE (6e-8 ..
// 65504), so both operands are block-scaled by exact powers of two:
//   weights:     one scale per layer, chosen by the loader so that the largest weight lands in [8192, 16384)
//   activations: one scale per workgroup tile, from a pre-scan of the rows it is about to stage (max |x| after the prologue): the largest
//                staged value lands in [8192, 16384); values many octaves below the tile's maximum lose relative precision, which is
//                irrelevant in a sum dominated by the large ones; an all-zero tile takes scale 1
// The accumulators run in scaled units (bias preloaded times both scales) and are brought back by one exact multiplication per
// register before the epilogue.  LDS row = two fp16 planes of the 64-channel chunk + 16 bytes (272 = 17 x 16).
// ---------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define EVH_RSB (2 * EVX_KC * 2 + 16)
// two fp16 pieces of four fp32 values (already scaled), packed in channel order (8 bytes per piece)
__device__ __forceinline__ void evh_split4(const f32x4 v, uint2& q0, uint2& q1) {
    _Float16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (_Float16)v[e]; l[e] = (_Float16)(v[e] - (float)h[e]); }
    const f16x2 a = {h[0], h[1]}, b = {h[2], h[3]}, c = {l[0], l[1]}, d = {l[2], l[3]};
    q0.x = __builtin_bit_cast(unsigned, a); q0.y = __builtin_bit_cast(unsigned, b);
    q1.x = __builtin_bit_cast(unsigned, c); q1.y = __builtin_bit_cast(unsigned, d);
}
// power of two s with max * s in [8192, 16384), clamped to [2^-40, 2^40]  (max = 0 or not finite: 1).
// The clamp bounds everything derived from a scale: the product with a weight scale (also within 2^+-40) stays within 2^+-80, and the factor
// xn / xs by which conv_h16_bal_kernel / ln_mlp_h16_kernel move their running sums (bounded by 2^40 in their own unit) from one chunk's scale
// to the next is at most 2^80: no step can overflow fp32.  A tile whose maximum is below 2^-27 is ~0 beside the bias; one whose finite maximum
// exceeds 2^55 (3.6e16) leaves fp16's range and comes out as Inf where the fp32 builds would still be finite — documented in DESIGN 3.
__device__ __forceinline__ float evh_scale_for(float mx) {
    const int ex = (int)((__float_as_uint(mx) >> 23) & 255u);          // biased exponent of max
    if (ex == 0 || ex == 255) return 1.f;
    int e2 = 267 - ex;                                                  // biased exponent of 2^(13 - floor(log2 max))
    e2 = e2 > 167 ? 167 : (e2 < 87 ? 87 : e2);
    return __uint_as_float((unsigned)e2 << 23);
}
// A tile maximum that is not finite (an Inf among the staged values; NaNs never reach a maximum: v_max ignores them) must not set the
// tile's scale: with scale 1 every finite value above 65504 of the SAME tile would overflow too and the small ones would lose their second
// piece — rows of another utterance that merely share the tile with the bad one.  The kernels therefore repeat their search over the FINITE
// values only (workgroup-uniform slow path, never taken on clean data); the non-finite elements themselves become fp16 Inf / NaN and spoil
// exactly the outputs whose taps touch them, as in the fp32 builds.
__device__ __forceinline__ bool evh_is_finite(float m) { return (__float_as_uint(m) & 0x7f800000u) != 0x7f800000u; }
__device__ __forceinline__ float evh_absmax4(const f32x4 v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }
__device__ __forceinline__ float evh_absmax4_finite(const f32x4 v) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float a = fabsf(v[e]); m = fmaxf(m, evh_is_finite(a) ? a : 0.f); }
    return m;
}
template <int TM, int TN>
__device__ __forceinline__ void evh_mma(f32x16 (&acc)[TM][TN], const f32x4 (&a)[2][TM], const f32x4 (&b)[2][TN]) {
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA[t]][i]), __builtin_bit_cast(f16x8, b[PB[t]][j]), acc[i][j], 0, 0, 0);
}

// Q = 1: the K loop on v_mfma_f32_16x16x32_f16 (16 x 16 output tiles, 32-deep steps) instead of v_mfma_f32_32x32x16_f16.  Same FLOP per
// cycle, same operand bytes per FLOP at the same 64 x 64 wave tile — but under this chip's power limit the 16 x 16 shape holds a higher clock:
// tools/mfma_shape_probe.hip (LDS + L2 fed loop of this kernel's shape, random data, two waves per SIMD): 1254 vs 1444 TFLOP/s executed,
// 1.52 vs 1.72 GHz (profiles/r04_mfma_shape_probe.txt; MI355X_MICROARCH.md, DVFS give-back item 7).  Weights: p.Wq, the same two fp16 pieces in
// the 16 x 16 x 32 fragment order [tap][Mpad/16][Kpad/32][piece][64 lanes][8]: lane = (row & 15) + 16 kgroup, element e = k 32 kg32 + 8 kgroup + e.
template <int BM, int BN, int WAVES_M, int WAVES_N, int LEAN, int Q = 0>
__global__ __launch_bounds__(256, 2) void conv_h16_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                            // [(BN + halo)][EVH_RSB bytes]
    float* red = smem + ((BN + EV_HALO) * EVH_RSB) / 4;  // 4 floats behind the tile: the waves' maxima

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.mtiles * p.ntiles;
    const int work = ev_xcd_remap(blockIdx.x, nwg);
    const int mt = work % p.mtiles;
    const int nt = work / p.mtiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    {   // tiles that contain no storable row (pure padding) do nothing
        const int t_first = (n0 % p.S) - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - (n0 % p.S) + p.P;
        if (dist >= BN || n0 + dist >= p.nrows) return;
    }
    EvAmax am = ev_amax_begin(p, n0 + wn * (TN * 32), TN * 32);   // (the residual's / running sum's bounds: requested now, used behind the epilogue)
    const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
    const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mt] : p.ntaps);

    const int xrows = BN + p.halo_lo + p.halo_hi;
    constexpr int TPR = EVX_KC / 4, RPS = 256 / TPR;   // staging: 16 threads per row, 16 rows per pass
    const int srow = tid / TPR;
    const int sc4 = (tid % TPR) * 4;
    const int nchunks = p.Kpad / EVX_KC;
    constexpr int XPASS = (BN + EV_HALO) / RPS;
    constexpr int XG = XPASS % 6 == 0 ? 6 : 4;
    static_assert(XPASS % XG == 0, "staging batches");
    const int mt32 = (m0 + wm * (TM * 32)) >> 5;
    const int KG16 = p.Kpad >> 4;
    const __amdgpu_buffer_rsrc_t rW = ev_rsrc(p.Wh), rX = ev_rsrc(p.X);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(mt32 * KG16) * 2048u;
    // (a tap's fp16 plane — (Mpad/32) (Kpad/16) x 2 KiB — is as large as its fp32 plane: the tap list's byte offsets apply as they are)
    auto a_off = [&](int tap_bytes, int kg16) -> unsigned { return (unsigned)tap_bytes + wbase + (unsigned)kg16 * 2048u; };
    f32x4 A0[2][TM], A1[2][TM], A2[2][TM], A3[2][TM], B0[2][TN], B1[2][TN];
    auto ldAp = [&](f32x4 (&dst)[2][TM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int i = 0; i < TM; ++i) dst[pc][i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG16 * 2048 + pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[2][TN], const char* brow, int slab) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * EVH_RSB + pc * (EVX_KC * 2) + slab * 32);
    };
    const char* bbase = Xb + (wn * (TN * 32) + li + p.halo_lo) * EVH_RSB + 16 * lh;
    const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
    const int2 tv_first = ev_tap_at(tlv, 0);
    // the 16 x 16 x 32 form's fragments (Q = 1; unused and eliminated otherwise)
    constexpr int QM = 2 * TM, QN = 2 * TN;            // 16 x 16 tiles of the wave tile
    const int f16i = lane & 15, kg = lane >> 4;        // frame (column) inside a tile / 8-deep k group of the 32-deep step
    const __amdgpu_buffer_rsrc_t rQ = ev_rsrc(p.Wq);
    const int KG32 = p.Kpad >> 5;
    const unsigned qbase = (unsigned)(((m0 + wm * (TM * 32)) >> 4) * KG32) * 2048u;
    // fragment (tile a, piece pc) of the 32-deep step kg32 of a tap: tap bytes + ((m16 KG32 + kg32) 2 + pc) KiB
    auto q_off = [&](int tap_bytes, int kg32) -> unsigned { return (unsigned)tap_bytes + qbase + (unsigned)kg32 * 2048u; };
    f32x4 QA0[2][QM], QA1[2][QM], QB0[2][QN], QB1[2][QN];
    auto ldQA = [&](f32x4 (&dst)[2][QM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int a = 0; a < QM; ++a) dst[pc][a] = ev_bload4(rQ, wlane, aoff + (unsigned)(a * KG32 * 2048 + pc * 1024));
    };
    if (nact > 0) {                                     // the first tap's fragments fly under the pre-scan
        if constexpr (Q == 0) {
            const unsigned a0 = a_off(tv_first.x, 0);
            ldAp(A0, a0); ldAp(A1, a0 + 2048u); ldAp(A2, a0 + 4096u); ldAp(A3, a0 + 6144u);
        } else {
            const unsigned a0 = q_off(tv_first.x, 0);
            ldQA(QA0, a0); ldQA(QA1, a0 + 2048u);
        }
    }
    unsigned xoff[XPASS];
#pragma unroll
    for (int q = 0; q < XPASS; ++q) {
        const int gr = n0 - p.halo_lo + q * RPS + srow;
        xoff[q] = ((q * RPS < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
    }
    // ---- pre-scan: max |x| (after the prologue: |lrelu(x)| <= |x|, so the raw maximum bounds it) over everything this tile stages
    float xs;
    {
        auto scan = [&](auto finite_only) -> float {
            constexpr bool FIN = decltype(finite_only)::value;
            float mx = 0.f;
            for (int ch = 0; ch < ((p.dbg & 2048) ? 0 : nchunks); ++ch) {      // (dbg 2048: tools/conv_bench.py ablation — no pre-scan, a fixed scale of 1024)
                const unsigned soff = evx_chunk_off(p, ch);
#pragma unroll
                for (int q0 = 0; q0 < XPASS; q0 += XG) {
                    if (q0 * RPS >= xrows) continue;
                    f32x4 xg[XG];
#pragma unroll
                    for (int q = 0; q < XG; ++q) xg[q] = ev_bload4(rX, xoff[q0 + q], soff);
#pragma unroll
                    for (int q = 0; q < XG; ++q) mx = fmaxf(mx, FIN ? evh_absmax4_finite(xg[q]) : evh_absmax4(xg[q]));
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) red[wave] = mx;
            ev_lds_barrier();
            mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            return mx;
        };
        // the producer's bound for the granules this tile stages (ev_amax_emit), when the input has one: no pre-scan.  A non-finite mark
        // sends the tile to the finite-only search — the pre-scan, kept as that slow path and for inputs without slots.
        float mx = p.xmax ? ev_amax_read(p, n0 - p.halo_lo, n0 + BN + p.halo_hi - 1, lane) : -1.f;
        if (mx < 0.f) mx = scan(std::false_type{});
        if (!evh_is_finite(mx)) {                           // an Inf in the tile (workgroup-uniform): the finite maximum sets the scale
            if (!p.xmax) ev_lds_barrier();                  // (every wave has read red[])
            mx = scan(std::true_type{});
        }
        xs = (p.dbg & 2048) ? 1024.f : evh_scale_for(mx);
    }
    const float acc_in = p.wh_scale * xs;              // bias in accumulator units
    const float acc_out = 1.0f / acc_in;               // (both powers of two: exact)

    auto stage = [&](int ch) {                          // chunk ch of the X tile: prologue, scale, split, LDS
        {
            const unsigned soff = evx_chunk_off(p, ch);
#pragma unroll
            for (int q0 = 0; q0 < XPASS; q0 += XG) {
                if (q0 * RPS >= xrows) continue;
                f32x4 xg[XG];
#pragma unroll
                for (int q = 0; q < XG; ++q) xg[q] = ev_bload4(rX, xoff[q0 + q], soff);      // (passes beyond the tile re-read row 0)
#pragma unroll
                for (int q = 0; q < XG; ++q) {
                    const int r = (q0 + q) * RPS + srow;
                    f32x4 v = xg[q];
                    if (p.pro_lrelu) {
                        v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                        v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                    }
                    uint2 q0v, q1v;
                    evh_split4(v * xs, q0v, q1v);
                    if (r < xrows) {
                        char* dst = Xb + r * EVH_RSB + sc4 * 2;
                        *(uint2*)(dst) = q0v; *(uint2*)(dst + EVX_KC * 2) = q1v;
                    }
                }
            }
        }
    };
    if constexpr (Q == 0) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bq[g] = z;
            const int c0 = m0 + wm * (TM * 32) + a * 32 + 8 * g + 4 * lh;
            if (p.bias && c0 < p.Cout) bq[g] = *(const f32x4*)(p.bias + c0) * acc_in;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
    }

    for (int ch = 0; ch < nchunks; ++ch) {
        __builtin_amdgcn_s_setprio(3);
        ev_lds_barrier();                               // the previous chunk's MFMAs are done with the tile
        stage(ch);
        ev_lds_barrier();
        __builtin_amdgcn_s_setprio(0);
        const char* brow = bbase + tv_first.y * EVH_RSB;
        ldB(B0, brow, 0);
        for (int ti = 0; ti < nact; ++ti) {
            const bool last_tap = (ti + 1 == nact);
            const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
            const char* nbrow = bbase + ntv.y * EVH_RSB;
            const bool have_next = !(last_tap && ch + 1 == nchunks);
            const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, 0);   // unconditional loads
            ldB(B1, brow, 1);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, A0, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A0, nap);
            ldB(B0, brow, 2);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, A1, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A1, nap + 2048u);
            ldB(B1, brow, 3);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, A2, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A2, nap + 4096u);
            ldB(B0, nbrow, 0);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, A3, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(A3, nap + 6144u);
            brow = nbrow;
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] *= acc_out;
    __builtin_amdgcn_s_setprio(3);
    ev_amax_from_acc<TM, TN>(p, am, acc, n0 + wn * (TN * 32), lane);
    conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
    ev_amax_emit(p, am, n0 + wn * (TN * 32), TN * 32, lane);
    } else {
        // ---------------- the 16 x 16 x 32 form ----------------
        f32x4 acc[QM][QN];
#pragma unroll
        for (int a = 0; a < QM; ++a) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            const int c0 = m0 + wm * (TM * 32) + a * 16 + 4 * kg;   // C/D layout: rows 4 (lane >> 4) + 0..3 of tile a
            if (p.bias && c0 < p.Cout) bq = *(const f32x4*)(p.bias + c0) * acc_in;
#pragma unroll
            for (int b = 0; b < QN; ++b) acc[a][b] = bq;
        }
        // B fragment of step ks (32 channels of the 64-channel chunk): lane = frame (lane & 15) of tile b, 8 channels 8 (lane >> 4) ..
        auto ldQB = [&](f32x4 (&dst)[2][QN], const char* brow, int ks) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int b = 0; b < QN; ++b) dst[pc][b] = *(const f32x4*)(brow + b * 16 * EVH_RSB + pc * (EVX_KC * 2) + ks * 64);
        };
        auto mmaQ = [&](const f32x4 (&a)[2][QM], const f32x4 (&b)[2][QN]) {
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < QM; ++i)
#pragma unroll
                    for (int jj = 0; jj < QN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[PA[t]][i]), __builtin_bit_cast(f16x8, b[PB[t]][jj]), acc[i][jj], 0, 0, 0);
        };
        const char* qbb = Xb + (wn * (TN * 32) + f16i + p.halo_lo) * EVH_RSB + 16 * kg;
        for (int ch = 0; ch < nchunks; ++ch) {
            __builtin_amdgcn_s_setprio(3);
            ev_lds_barrier();                           // the previous chunk's MFMAs are done with the tile
            stage(ch);
            ev_lds_barrier();
            __builtin_amdgcn_s_setprio(0);
            const char* brow = qbb + tv_first.y * EVH_RSB;
            ldQB(QB0, brow, 0);
            for (int ti = 0; ti < nact; ++ti) {
                const bool last_tap = (ti + 1 == nact);
                const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
                const char* nbrow = qbb + ntv.y * EVH_RSB;
                const bool have_next = !(last_tap && ch + 1 == nchunks);
                const unsigned nap = have_next ? q_off(ntv.x, last_tap ? ch * 2 + 2 : ch * 2) : q_off(tv_first.x, 0);   // unconditional loads
                ldQB(QB1, brow, 1);
                __builtin_amdgcn_sched_barrier(0);
                mmaQ(QA0, QB0);
                __builtin_amdgcn_sched_barrier(0);
                ldQA(QA0, nap);
                ldQB(QB0, nbrow, 0);
                __builtin_amdgcn_sched_barrier(0);
                mmaQ(QA1, QB1);
                __builtin_amdgcn_sched_barrier(0);
                ldQA(QA1, nap + 2048u);
                brow = nbrow;
            }
        }
#pragma unroll
        for (int a = 0; a < QM; ++a)
#pragma unroll
            for (int b = 0; b < QN; ++b) acc[a][b] *= acc_out;
        __builtin_amdgcn_s_setprio(3);
        ev_amax_from_acc_q<QM, QN>(p, am, acc, n0 + wn * (TN * 32), lane);
        conv_epilogue_lean_q<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
        ev_amax_emit(p, am, n0 + wn * (TN * 32), TN * 32, lane);
    }
}

// ---------------------------------------------------------------------------
// conv_split_bal_kernel: the balanced persistent grid of conv_gemm_bal_kernel with the K loop of conv_split_kernel (bf16 pipe, six
// exact products per element pair).  A unit = (tile, 64-channel chunk); hand-off, tags, bounded waits and the recompute fallback
// are those of SkCtl.  Layers whose M tiles carry different tap counts (a 3-tap conv stacked over a 1x1 conv) take it too, with
// equal unit weights: the 128-channel tiles fall on either side of the stacking boundary.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int LEAN, int TERMS = 6>
__global__ __launch_bounds__(256, 2) void conv_split_bal_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1 && LEAN != 0, "balanced build: 4 waves, lean epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.Kpad / EVX_KC;
    const int g = blockIdx.x;
    const unsigned tag = sk_tag(p.sk);
    int u = sk_start(p.sk, g);
    const int ue = sk_start(p.sk, g + 1);
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(p.sk.part), rW = ev_rsrc(p.Wx), rX = ev_rsrc(p.X);
    const unsigned pslot = (unsigned)p.sk.part_floats * 8u;
    constexpr int PN = TM * TN * 4;
    const unsigned pelem = (unsigned)(wave * PN) * 1024u + (unsigned)lane * 16u;
    int* skw = (int*)((char*)smem + p.sk.lds_word);
    bool pend_pub = false;
    const unsigned wlane = (unsigned)lane * 16u;
    const int KG16 = p.Kpad >> 4;
    constexpr int TPR = EVX_KC / 4, RPS = 256 / TPR;
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;
    constexpr int XPASS = (BN + EV_HALO) / RPS;
    constexpr int XG = 4;
    static_assert(XPASS % XG == 0, "staging batches");
    const int xrows = BN + p.halo_lo + p.halo_hi;

    while (u < ue) {
        const int t = u / nchunks, c0 = u - t * nchunks;
        const int c1 = (ue - u < nchunks - c0) ? c0 + (ue - u) : nchunks;
        u += c1 - c0;
        const int mt = t % p.mtiles, nt = t / p.mtiles;
        const int m0 = mt * BM, n0 = nt * BN;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= BN || n0 + dist >= p.nrows) continue;
        }
        if (pend_pub) { sk_publish(p.sk, g, tag, tid); pend_pub = false; }
        const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
        const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mt] : p.ntaps);
        const int mt32 = (m0 + wm * (TM * 32)) >> 5;
        const unsigned wbase = (unsigned)(mt32 * KG16) * 3072u;
        auto a_off = [&](int tap_bytes, int kg16) -> unsigned { return (unsigned)tap_bytes + ((unsigned)tap_bytes >> 1) + wbase + (unsigned)kg16 * 3072u; };
        f32x4 A0[3][TM], A1[3][TM], A2[3][TM], A3[3][TM], B0[3][TN], B1[3][TN];
        auto ldAp = [&](f32x4 (&dst)[3][TM], unsigned aoff) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int i = 0; i < TM; ++i) dst[pc][i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG16 * 3072 + pc * 1024));
        };
        auto ldB = [&](f32x4 (&dst)[3][TN], const char* brow, int slab) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * EVX_RSB + pc * (EVX_KC * 2) + slab * 32);
        };
        const char* bbase = Xb + (wn * (TN * 32) + li + p.halo_lo) * EVX_RSB + 16 * lh;
        const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
        const int2 tv_first = ev_tap_at(tlv, 0);
        int gr0 = n0 - p.halo_lo + srow;                   // first staging row of this thread (rows outside the tensor / the tile read pad row 0)
        auto xoff = [&](int q) -> unsigned {
            const int gr = gr0 + q * RPS;
            return ((q * RPS < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
        };
        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;
        const int tile_end = (t + 1) * nchunks;
        for (;;) {
            f32x16 acc[TM][TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    bq[q] = z;
                    const int cc = m0 + wm * (TM * 32) + a * 32 + 8 * q + 4 * lh;   // bias preloaded (lean epilogue), by the owner's first pass only
                    if (p.bias && cA == 0 && cc < p.Cout) bq[q] = *(const f32x4*)(p.bias + cc);
                }
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
            }
            if (nact > 0) {
                const unsigned a0 = a_off(tv_first.x, cA * 4);
                ldAp(A0, a0); ldAp(A1, a0 + 3072u); ldAp(A2, a0 + 6144u); ldAp(A3, a0 + 9216u);
            }
            for (int ch = cA; ch < cB; ++ch) {
                __builtin_amdgcn_s_setprio(3);
                ev_lds_barrier();                          // the previous chunk's MFMAs (or the previous segment's epilogue) are done with LDS
                {
                    const unsigned soff = evx_chunk_off(p, ch);
                    asm volatile("" : "+v"(gr0));          // (row offsets recomputed per chunk, not kept — and spilled — as loop invariants: conv_h16_bal_kernel)
#pragma unroll
                    for (int q0 = 0; q0 < XPASS; q0 += XG) {
                        if (q0 * RPS >= xrows) continue;
                        f32x4 xg[XG];
#pragma unroll
                        for (int q = 0; q < XG; ++q) xg[q] = ev_bload4(rX, xoff(q0 + q), soff);
#pragma unroll
                        for (int q = 0; q < XG; ++q) {
                            const int r = (q0 + q) * RPS + srow;
                            f32x4 v = xg[q];
                            if (p.pro_lrelu) {
                                v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                                v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                            }
                            uint2 q0v, q1v, q2v;
                            evx_split4(v, q0v, q1v, q2v);
                            if (r < xrows) {
                                char* dst = Xb + r * EVX_RSB + sc4 * 2;
                                *(uint2*)(dst) = q0v; *(uint2*)(dst + EVX_KC * 2) = q1v; *(uint2*)(dst + EVX_KC * 4) = q2v;
                            }
                        }
                    }
                }
                ev_lds_barrier();
                __builtin_amdgcn_s_setprio(0);
                const char* brow = bbase + tv_first.y * EVX_RSB;
                ldB(B0, brow, 0);
                for (int ti = 0; ti < nact; ++ti) {
                    const bool last_tap = (ti + 1 == nact);
                    const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
                    const char* nbrow = bbase + ntv.y * EVX_RSB;
                    const bool have_next = !(last_tap && ch + 1 == cB);
                    const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, cA * 4);
                    ldB(B1, brow, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    evx_mma<TM, TN, TERMS>(acc, A0, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A0, nap);
                    ldB(B0, brow, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    evx_mma<TM, TN, TERMS>(acc, A1, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A1, nap + 3072u);
                    ldB(B1, brow, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    evx_mma<TM, TN, TERMS>(acc, A2, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A2, nap + 6144u);
                    ldB(B0, nbrow, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    evx_mma<TM, TN, TERMS>(acc, A3, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A3, nap + 9216u);
                    brow = nbrow;
                }
            }
            auto acc_io = [&](unsigned base, int mode) {    // mode 0: store (write-through), 1: add from memory
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned soff = base + (unsigned)((a * TN + b) * 4 + q) * 1024u;     // (wave-uniform: scalar offset)
                            if (mode == 0) {
                                const f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                                ev_bstore4_sc1(rPart, pelem, soff, v);
                            } else {
                                const f32x4 v = ev_bload4_sc1(rPart, pelem, soff);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[a][b][4 * q + e] += v[e];
                            }
                        }
            };
            if (spilled) { acc_io((unsigned)g * pslot + pslot / 2, 1); spilled = false; }
            if (c0 != 0) {                                 // not the owner: hand the partial tile over (flag raised at the next segment)
                acc_io((unsigned)g * pslot, 0);
                pend_pub = true;
                break;
            }
            bool again = false;
            while (c1 < nchunks) {                         // owner: add the contributors' partial tiles in ascending workgroup order
                int n = 0;
                while (n < 64 && gi + n < (int)gridDim.x && sk_start(p.sk, gi + n) < tile_end) ++n;
                if (n == 0) break;
                const int ready = sk_wait_many(p.sk, gi, n, tag, tid, skw);
                for (int k = 0; k < ready; ++k) acc_io((unsigned)(gi + k) * pslot, 1);
                gi += ready;
                if (ready < n) {                           // gi is not there in time: spill the running sum, compute its share here
                    const int sgi = sk_start(p.sk, gi);
                    int egi = sk_start(p.sk, gi + 1);
                    egi = egi < tile_end ? egi : tile_end;
                    acc_io((unsigned)g * pslot + pslot / 2, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cA = sgi - t * nchunks; cB = egi - t * nchunks;
                    spilled = true; again = true;
                    ++gi;
                    break;
                }
            }
            if (again) continue;
            __builtin_amdgcn_s_setprio(3);
            conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
            __builtin_amdgcn_s_setprio(0);
            break;
        }
    }
    if (pend_pub) sk_publish(p.sk, g, tag, tid);
    sk_arrive(p.sk, tag, tid);
}

// ---------------------------------------------------------------------------
// conv_h16_bal_kernel: conv_split_bal_kernel with the fp16 form of conv_h16_kernel.  Every staged chunk carries its own activation scale (from
// its rows, in registers, before they are split) and every pass brings its accumulators back to true units before they are handed over,
// spilled or stored, so workgroups that share a tile may run different scales.
// ---------------------------------------------------------------------------
// XP = staging passes of 16 rows a chunk is read in: 12 covers the widest halo, 9 the 3-tap layers of the U-Net (host-checked: BN + halo <= 16 XP) —
// twelve registers fewer held beside the accumulators and the weight ring.
// EIGHT waves (<256, 128, 4, 2, ...>: one workgroup per CU, all 256 output channels of a 128-frame tile; passes of 32 rows): the two workgroups of
// a CU in the four-wave build run the same program in lockstep and stage the SAME X rows when the layer has two M tiles — merged, the tile is
// staged once (half the bytes of the burst every chunk begins with), the MFMA phase is the same two waves per SIMD.
template <int BM, int BN, int WAVES_M, int WAVES_N, int LEAN, int XP = (BN + EV_HALO) / (4 * WAVES_M * WAVES_N), int XK = (XP > 9 ? 8 : XP)>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, WAVES_M * WAVES_N == 4 ? 2 : 1) void conv_h16_bal_kernel(const ConvParams p) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    static_assert((NW == 4 || NW == 8) && TM >= 1 && TN >= 1 && LEAN != 0, "balanced build: 4 waves (two workgroups per CU) or 8 (one), lean epilogue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int nchunks = p.Kpad / EVX_KC;
    const int g = blockIdx.x;
    // (the epoch is requested here and consumed at its first use — a hand-off, long after — so that its round trip does not stand in front
    // of the first tile's loads)
    const unsigned tag_v = p.sk.ctrl ? (unsigned)__hip_atomic_load((ev_gu32*)p.sk.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    auto get_tag = [&]() -> unsigned { return p.sk.ctrl ? (unsigned)__builtin_amdgcn_readfirstlane((int)tag_v) + 1u : 0u; };
    int u = sk_start(p.sk, g);
    const int ue = sk_start(p.sk, g + 1);
    const unsigned claim_prev = sk_claim_start(p.sk, g, threadIdx.x);     // (announce this workgroup: an owner that finds it absent takes its first share over)
    bool first_seg = true;
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(p.sk.part), rW = ev_rsrc(p.Wh), rX = ev_rsrc(p.X);
    const unsigned pslot = (unsigned)p.sk.part_floats * 8u;
    constexpr int PN = TM * TN * 4;
    const unsigned pelem = (unsigned)(wave * PN) * 1024u + (unsigned)lane * 16u;
    int* skw = (int*)((char*)smem + p.sk.lds_word);
    float* hred = (float*)((char*)smem + p.sk.lds_word + 16);                 // the waves' maxima of a pass's pre-scan
    bool pend_pub = false;
    const unsigned wlane = (unsigned)lane * 16u;
    const int KG16 = p.Kpad >> 4;
    constexpr int TPR = EVX_KC / 4, RPS = NTHR / TPR;   // staging: 16 threads per row, 16 (32) rows per pass
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;
    constexpr int XPASS = XP;
    const int xrows = BN + p.halo_lo + p.halo_hi;
    int nst = 0;                                       // diagnostic (EV_BAL_STAMPS): up to 16 s_memrealtime stamps per workgroup
    auto stamp = [&]() { if (p.stamps && tid == 0 && nst < 16) p.stamps[16 * g + nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
    stamp();
    // Start stagger (EV_BAL_STAGGER=<us>, A/B): the two workgroups of a CU run the same program on the same amount of work, so they stage (HBM-bound,
    // all workgroups of the launch at once) and issue MFMAs (both on the same SIMDs) IN PHASE — per-workgroup stamps: 3.5 us staged + 4 us
    // MFMAs per chunk where one workgroup alone on the matrix pipes needs 2.2.  The second half of the grid (the workgroups that share a CU
    // with the first half under the dispatcher's round-robin; a placement assumption for speed only) starts late by about half a chunk period.
    if (p.stagger_slots > 0 && g >= (int)(gridDim.x >> 1))
        for (int i_ = 0; i_ < p.stagger_slots; ++i_) __builtin_amdgcn_s_sleep(32);      // 32 x 64 cycles ~ 1 us per step

    while (u < ue) {
        const int t = u / nchunks, c0 = u - t * nchunks;
        const int c1 = (ue - u < nchunks - c0) ? c0 + (ue - u) : nchunks;
        u += c1 - c0;
        const int mt = t % p.mtiles, nt = t / p.mtiles;
        const int m0 = mt * BM, n0 = nt * BN;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= BN || n0 + dist >= p.nrows) continue;
        }
        if (first_seg) {                                   // (only a workgroup's first segment can be a contributor's share)
            first_seg = false;
            if (c0 != 0 && sk_claim_taken(p.sk, claim_prev, tid, skw)) continue;     // its owner computes it: skip
        }
        if (pend_pub) { sk_publish(p.sk, g, get_tag(), tid); pend_pub = false; }
        const int mtl = p.mt_mul > 1 ? mt * p.mt_mul : mt;   // (the tap tables are per 128-channel tile)
        const int2* tl = p.taplist + (size_t)mtl * p.tl_stride;
        const int nact = __builtin_amdgcn_readfirstlane(p.nact_tab ? p.nact_tab[mtl] : p.ntaps);
        const int mt32 = (m0 + wm * (TM * 32)) >> 5;
        const unsigned wbase = (unsigned)(mt32 * KG16) * 2048u;
        auto a_off = [&](int tap_bytes, int kg16) -> unsigned { return (unsigned)tap_bytes + wbase + (unsigned)kg16 * 2048u; };
        f32x4 A0[2][TM], A1[2][TM], A2[2][TM], A3[2][TM], B0[2][TN], B1[2][TN];
        auto ldAp = [&](f32x4 (&dst)[2][TM], unsigned aoff) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int i = 0; i < TM; ++i) dst[pc][i] = ev_bload4(rW, wlane, aoff + (unsigned)(i * KG16 * 2048 + pc * 1024));
        };
        auto ldB = [&](f32x4 (&dst)[2][TN], const char* brow, int slab) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * EVH_RSB + pc * (EVX_KC * 2) + slab * 32);
        };
        const char* bbase = Xb + (wn * (TN * 32) + li + p.halo_lo) * EVH_RSB + 16 * lh;
        const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
        const int2 tv_first = ev_tap_at(tlv, 0);
        int gr0 = n0 - p.halo_lo + srow;                   // first staging row of this thread (rows outside the tensor / the tile read pad row 0)
        auto xoff = [&](int q) -> unsigned {
            const int gr = gr0 + q * RPS;
            return ((q * RPS < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u;
        };
        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;
        const int tile_end = (t + 1) * nchunks;
        for (;;) {
            // One activation scale per staged CHUNK, taken from the chunk's own rows while they wait in registers (no pre-scan pass: the
            // launches of the U-Net are one round of workgroups in lockstep, and a pre-scan that re-read every tile — 70-100 MB per
            // launch, all workgroups at once — took 13-27 us of their 62-90 us: per-workgroup stamps, EV_BAL_STAMPS,
            // profiles/r03_conv_h16_bal_stamps.txt).  The accumulators run in units of wh_scale * (scale of the chunk in LDS) and are
            // rescaled by an exact power of two when that changes (ln_mlp_h16_kernel does the same per hidden chunk).
            float xs = 1.0f;                               // scale of the chunk whose planes are in LDS (none yet: unit 1)
            f32x16 acc[TM][TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    bq[q] = z;
                    const int cc = m0 + wm * (TM * 32) + a * 32 + 8 * q + 4 * lh;   // bias preloaded (lean epilogue), by the owner's first pass only
                    if (p.bias && cA == 0 && cc < p.Cout) bq[q] = *(const f32x4*)(p.bias + cc) * p.wh_scale;
                }
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = bq[r >> 2][r & 3];
            }
            if (nact > 0) {
                const unsigned a0 = a_off(tv_first.x, cA * 4);
                ldAp(A0, a0); ldAp(A1, a0 + 2048u); ldAp(A2, a0 + 4096u); ldAp(A3, a0 + 6144u);
            }
            for (int ch = cA; ch < cB; ++ch) {
                __builtin_amdgcn_s_setprio(3);
                {
                    // the chunk's rows into registers (requested before the barrier: the round trip overlaps the other waves' last MFMAs),
                    // prologue applied, and this thread's maximum over the rows the tile really holds
                    const unsigned soff = evx_chunk_off(p, ch);
                    // XK of the XPASS staging passes stay in registers from the maximum search to the split; the passes beyond (wide-halo
                    // layers only: XPASS = 12) are read TWICE — once for the maximum, once more (L2-hot, 16 KB per workgroup) for the split —
                    // which keeps the build at 36 staging registers instead of 48 and out of scratch.
                    f32x4 xg[XK];
                    // (the row offsets are recomputed per chunk — four vector instructions per pass: as loop invariants hipcc kept all of them,
                    // range tests included, in registers across the MFMA phase and spilled them to scratch; the empty asm hides the invariance)
                    asm volatile("" : "+v"(gr0));
                    auto pro = [&](f32x4 v) -> f32x4 {
                        if (p.pro_lrelu) {
                            v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                            v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                        }
                        return v;
                    };
                    float mx = 0.f, mxf = 0.f;             // maximum / maximum over the finite values of the re-read passes
                    if constexpr (XPASS > XK) {
                        f32x4 xt[XPASS - XK];
#pragma unroll
                        for (int q = XK; q < XPASS; ++q) xt[q - XK] = ev_bload4(rX, xoff(q), soff);
#pragma unroll
                        for (int q = 0; q < XK; ++q) xg[q] = ev_bload4(rX, xoff(q), soff);
#pragma unroll
                        for (int q = XK; q < XPASS; ++q) {
                            const f32x4 v = pro(xt[q - XK]);
                            const bool in = q * RPS + srow < xrows;
                            mx = in ? fmaxf(mx, evh_absmax4(v)) : mx;
                            mxf = in ? fmaxf(mxf, evh_absmax4_finite(v)) : mxf;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < XK; ++q) xg[q] = ev_bload4(rX, xoff(q), soff);
                    }
#pragma unroll
                    for (int q = 0; q < XK; ++q) {
                        const f32x4 v = pro(xg[q]);
                        xg[q] = v;
                        mx = (q * RPS + srow < xrows) ? fmaxf(mx, evh_absmax4(v)) : mx;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                    float* hr = hred + NW * ((ch - cA) & 1);   // (two sets: the set of chunk ch - 2 was read before that chunk's second barrier)
                    if (lane == 0) hr[wave] = mx;
                    ev_lds_barrier();                      // maxima published; the previous chunk's MFMAs (or the previous segment's epilogue) are done with LDS
                    float cmx = fmaxf(fmaxf(hr[0], hr[1]), fmaxf(hr[2], hr[3]));
                    if constexpr (NW == 8) cmx = fmaxf(cmx, fmaxf(fmaxf(hr[4], hr[5]), fmaxf(hr[6], hr[7])));
                    if (!evh_is_finite(cmx)) {             // an Inf among the chunk's rows (workgroup-uniform, never on clean data): the finite maximum sets the scale
                        mx = mxf;
#pragma unroll
                        for (int q = 0; q < XK; ++q) mx = (q * RPS + srow < xrows) ? fmaxf(mx, evh_absmax4_finite(xg[q])) : mx;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                        float* h3 = hred + 2 * NW;              // (a third set: only this path writes it, always behind the barrier above)
                        if (lane == 0) h3[wave] = mx;
                        ev_lds_barrier();
                        cmx = fmaxf(fmaxf(h3[0], h3[1]), fmaxf(h3[2], h3[3]));
                        if constexpr (NW == 8) cmx = fmaxf(cmx, fmaxf(fmaxf(h3[4], h3[5]), fmaxf(h3[6], h3[7])));
                    }
                    const float xn = evh_scale_for(cmx);
                    if (xn != xs) {                        // (workgroup-uniform)
                        const float f = xn / xs;
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int b = 0; b < TN; ++b)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[a][b][r] *= f;
                        xs = xn;
                    }
                    auto put = [&](int q, const f32x4 v) {
                        const int r = q * RPS + srow;
                        uint2 q0v, q1v;
                        evh_split4(v * xs, q0v, q1v);
                        if (r < xrows) {
                            char* dst = Xb + r * EVH_RSB + sc4 * 2;
                            *(uint2*)(dst) = q0v; *(uint2*)(dst + EVX_KC * 2) = q1v;
                        }
                    };
                    if constexpr (XPASS > XK) {            // second read of the passes that were not kept
                        f32x4 xt[XPASS - XK];
#pragma unroll
                        for (int q = XK; q < XPASS; ++q) xt[q - XK] = ev_bload4(rX, xoff(q), soff);
#pragma unroll
                        for (int q = 0; q < XK; ++q) put(q, xg[q]);
#pragma unroll
                        for (int q = XK; q < XPASS; ++q) put(q, pro(xt[q - XK]));
                    } else {
#pragma unroll
                        for (int q = 0; q < XK; ++q) put(q, xg[q]);
                    }
                }
                ev_lds_barrier();
                __builtin_amdgcn_s_setprio(0);
                stamp();                                   // chunk staged
                const char* brow = bbase + tv_first.y * EVH_RSB;
                ldB(B0, brow, 0);
                for (int ti = 0; ti < nact; ++ti) {
                    const bool last_tap = (ti + 1 == nact);
                    const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
                    const char* nbrow = bbase + ntv.y * EVH_RSB;
                    const bool have_next = !(last_tap && ch + 1 == cB);
                    const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, cA * 4);
                    ldB(B1, brow, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    evh_mma<TM, TN>(acc, A0, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A0, nap);
                    ldB(B0, brow, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    evh_mma<TM, TN>(acc, A1, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A1, nap + 2048u);
                    ldB(B1, brow, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    evh_mma<TM, TN>(acc, A2, B0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A2, nap + 4096u);
                    ldB(B0, nbrow, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    evh_mma<TM, TN>(acc, A3, B1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldAp(A3, nap + 6144u);
                    brow = nbrow;
                }
                stamp();                                   // chunk's MFMAs issued
            }
            const float acc_out = 1.0f / (p.wh_scale * xs);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] *= acc_out;      // back to true units: partial tiles are handed over in them
            auto acc_io = [&](unsigned base, int mode) {    // mode 0: store (write-through), 1: add from memory
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned soff = base + (unsigned)((a * TN + b) * 4 + q) * 1024u;     // (wave-uniform: scalar offset)
                            if (mode == 0) {
                                const f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                                ev_bstore4_sc1(rPart, pelem, soff, v);
                            } else {
                                const f32x4 v = ev_bload4_sc1(rPart, pelem, soff);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[a][b][4 * q + e] += v[e];
                            }
                        }
            };
            if (spilled) { acc_io((unsigned)g * pslot + pslot / 2, 1); spilled = false; }
            if (c0 != 0) {                                 // not the owner: hand the partial tile over (flag raised at the next segment)
                acc_io((unsigned)g * pslot, 0);
                pend_pub = true;
                stamp();                                   // partial stored
                break;
            }
            bool again = false;
            while (c1 < nchunks) {                         // owner: add the contributors' partial tiles in ascending workgroup order
                int n = 0;
                while (n < 64 && gi + n < (int)gridDim.x && sk_start(p.sk, gi + n) < tile_end) ++n;
                if (n == 0) break;
                const int ready = sk_wait_many(p.sk, gi, n, get_tag(), tid, skw);
                for (int k = 0; k < ready; ++k) acc_io((unsigned)(gi + k) * pslot, 1);
                gi += ready;
                if (ready < n) {                           // gi is not there in time: spill the running sum, compute its share here
                    const int sgi = sk_start(p.sk, gi);
                    int egi = sk_start(p.sk, gi + 1);
                    egi = egi < tile_end ? egi : tile_end;
                    acc_io((unsigned)g * pslot + pslot / 2, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cA = sgi - t * nchunks; cB = egi - t * nchunks;
                    spilled = true; again = true;
                    ++gi;
                    break;
                }
            }
            if (again) continue;
            stamp();                                       // contributors' partials added
            __builtin_amdgcn_s_setprio(3);
            conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane);
            __builtin_amdgcn_s_setprio(0);
            stamp();                                       // epilogue done
            break;
        }
    }
    stamp();
    if (pend_pub) sk_publish(p.sk, g, get_tag(), tid);
    sk_arrive(p.sk, get_tag(), tid);
}

// (A loader-wave build of this kernel — two extra waves per workgroup that only stage X tiles, so that the MFMA waves
// never wait for HBM behind their in-order vmcnt — was built and measured 5-30 % SLOWER on every shape: a wave that
// streams MFMAs back to back starves every dependent instruction chain of the other waves on its SIMD
// (tools/valu_under_mfma.hip, profiles/r01_valu_under_mfma.log), so the loaders only ran in the MFMA waves' gaps.)

// ---------------------------------------------------------------------------
// conv_gemm_sk_kernel: the small-launch build.  A batch-1 decode issues ~500 convs of 20-150 tiles each: far fewer
// workgroups than CUs, every one a serial chain of (HBM latency per 32-channel stage) x (K/32 stages) + K/2 MFMAs of one
// wave.  Here a workgroup is 4*KS waves: the 64 x 64 tile is still computed by 4 waves of 32 x 32, but KS such groups
// split the K loop (k-chunk c goes to group c mod KS), all 256*KS threads stage up to eight k-chunks of the X tile at
// once (one latency episode instead of eight), and the KS partial tiles are summed through LDS in a fixed order
// (group 0 + 1 + 2 + ...: deterministic) before group 0 runs the usual epilogue.  Same operands and products as
// conv_gemm_kernel; only the summation order over k-chunks differs.
// ---------------------------------------------------------------------------
// TW = 4: a 64 x 64 tile (2 x 2 waves of 32 x 32) x KS wave groups.  TW = 1: a 32 x 32 tile whose K loop is split over KS single
// waves — the tile's MFMA work is bound to ONE CU's four matrix pipes however many waves share it (a 64 x 64 x 768 tile is 384
// MFMAs = 10 us per SIMD), so a launch of a few dozen 64 x 64 tiles on 256 CUs is sped up by spreading it over four times as
// many CUs, not by more waves per tile.
// Latency order of a workgroup (a launch is ~one workgroup per CU, so its duration IS this chain): dense layers carry
// their tap list in the kernel arguments (tap i = {i * plane_bytes, koff0 + i * kdoff}), so each wave issues the weight fragments of its first NPRE (chunk,
// tap) steps straight from the kernarg scalars — before, and in the shadow of, the X staging; TW = 1 stages up to 32 k-chunks at
// once (the whole K of every U-Net layer: one staging episode, one barrier).  Round 1 fetched the tap list through LDS, started
// the first weight load after the staging barrier and kept one step in flight: three to five exposed memory latencies per launch.
template <int KS, bool FULL_ACT, int LEAN, int TW = 4>
__global__ __launch_bounds__(64 * TW * KS) void conv_gemm_sk_kernel(const ConvParams pk) {
    static_assert(TW == 4 || TW == 1, "tile waves");
    constexpr int BM = TW == 4 ? 64 : 32, BN = BM, NTHR = 64 * TW * KS;
    constexpr int MAXPASS = EV_SK_MAXF4(TW) / NTHR;       // float4 of the X tile a thread stages per round
    constexpr int NPRE = TW == 1 ? 4 : 1;                 // (chunk, tap) steps whose weight fragments are issued up front
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int2* taps = (int2*)smem;                             // the (<= 16) tap-list entries of this M tile, then the X tile
    float* Xs = smem + 2 * EV_MAX_TAPS;
    // Every kernel argument this kernel touches, fetched as ONE batch of scalar loads into a local copy whose fields are opaque to
    // the compiler from here on ("+s": they cannot be re-fetched).  hipcc otherwise loads each field right in front of its first
    // use and, short of SGPRs, AGAIN in front of later uses: ~20 dependent s_load / s_waitcnt round trips of ~0.3 us each in a
    // row — most of the 4.2 us between the start of a workgroup and its staging barrier (per-workgroup stamps,
    // tools/conv_bench.py 16:-1 / 1040:-1 / 2064:-1; profiles/r02_sk32_workgroup_timeline.log).
    ConvParams p = pk;
    asm volatile("" : "+s"(p.X), "+s"(p.ldx), "+s"(p.Cin), "+s"(p.isplit_log2), "+s"(p.isstride), "+s"(p.W), "+s"(p.Kpad), "+s"(p.bias),
                      "+s"(p.Y), "+s"(p.ldy), "+s"(p.Cout), "+s"(p.nrows), "+s"(p.S), "+s"(p.P), "+s"(p.T));
    asm volatile("" : "+s"(p.ntaps), "+s"(p.halo_lo), "+s"(p.halo_hi), "+s"(p.taplist), "+s"(p.tl_stride), "+s"(p.nact_tab), "+s"(p.mtiles),
                      "+s"(p.pro_slope), "+s"(p.pro_lrelu), "+s"(p.act), "+s"(p.act_slope), "+s"(p.act_a), "+s"(p.act_b), "+s"(p.mask1), "+s"(p.R));
    asm volatile("" : "+s"(p.ldr), "+s"(p.accum), "+s"(p.div3), "+s"(p.act2_lrelu), "+s"(p.act2_slope), "+s"(p.mask2), "+s"(p.rowmask),
                      "+s"(p.ktaps_n), "+s"(p.plane_bytes), "+s"(p.koff0), "+s"(p.kdoff), "+s"(p.sk_kb), "+s"(p.stamps), "+s"(p.dbg));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = wave % TW, ks = wave / TW;
    const int wm = TW == 4 ? (tw >> 1) : 0, wn = TW == 4 ? (tw & 1) : 0;
    const int li = lane & 31, lh = lane >> 5;
    const int work = blockIdx.x;
    const int mt = work % p.mtiles, nt = work / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;
    {   // tiles that contain no storable row (pure padding) do nothing
        int t_first = (n0 % p.S) - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - (n0 % p.S) + p.P;
        if (dist >= BN || n0 + dist >= p.nrows) return;
    }
    if ((p.dbg & 16) && tid == 0) p.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
#ifdef EV_NOPSLED
    asm volatile(".rept 1024\n s_nop 0\n .endr" ::: "memory");   // probe: 4 KB of straight-line code on the path
#endif
    const bool ktp = p.ktaps_n > 0;                       // tap list in the kernel arguments
    const int2* tl = p.taplist + (size_t)mt * p.tl_stride;
    int nact = p.ntaps;
    if (!ktp && p.nact_tab) nact = __builtin_amdgcn_readfirstlane(p.nact_tab[mt]);
    if (!ktp && tid < nact) taps[tid] = tl[tid];          // (published by the first staging barrier)
    const int nchunks = p.Kpad / EV_BK;
    const int KBs = p.sk_kb;                              // k-chunks per staging round (host: fits LDS and MAXPASS)
    const int LDKs = 32 * KBs + 4;                        // row stride: 4 (mod 32) floats, conflict-free for ds_read_b128
    const int xrows = BN + p.halo_lo + p.halo_hi;

    const int mt32 = (m0 + wm * 32) >> 5;
    const int KG8 = p.Kpad >> 3;
    const __amdgpu_buffer_rsrc_t rW = ev_rsrc(p.W), rX = ev_rsrc(p.X);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(mt32 * KG8) * 1024u;
    const int f4row = KBs * 8;                            // float4 per staged row
    const int nf4 = xrows * f4row;

    // ---- weight fragments of this wave's first steps of round 0 (step s: tap s % nact of chunk ks + (s / nact) * KS)
    f32x4 A[NPRE][4];
    int2 tvs[NPRE];
    int cs[NPRE];
    int ti0 = 0, cn0 = ks;
    const int kb0 = nchunks < KBs ? nchunks : KBs;
    const int npre = __builtin_amdgcn_readfirstlane(ktp && kb0 > ks ? (((kb0 - ks + KS - 1) / KS) * nact < NPRE ? ((kb0 - ks + KS - 1) / KS) * nact : NPRE) : 0);
#pragma unroll
    for (int s = 0; s < NPRE; ++s) {
        tvs[s] = make_int2(0, 0); cs[s] = 0;
        if (s < npre) {
            const int2 tv = make_int2(ti0 * p.plane_bytes, p.koff0 + ti0 * p.kdoff);
            tvs[s] = tv; cs[s] = cn0;
            const unsigned ap = (unsigned)tv.x + wbase + (unsigned)(cn0 * 4) * 1024u;
            A[s][0] = ev_bload4(rW, wlane, ap); A[s][1] = ev_bload4(rW, wlane, ap + 1024u);
            A[s][2] = ev_bload4(rW, wlane, ap + 2048u); A[s][3] = ev_bload4(rW, wlane, ap + 3072u);
            if (++ti0 == nact) { ti0 = 0; cn0 += KS; }
        }
    }

    f32x16 acc;
    f32x4 bq[4];                                          // (consumed after the staging barrier: no wait for it in front of the X loads)
    {   // bias through a descriptor bounded by Cout (a null bias has zero records): out-of-range channels read 0, no branches
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (LEAN != 0 && ks == 0 && p.bias) ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = ev_bload4(rB, (unsigned)(m0 + wm * 32 + 8 * g + 4 * lh) * 4u, 0);
    }
    auto mfma16 = [&](const f32x4 (&a)[4], int2 tv, int c) {
        const float* brow = Xs + (wn * 32 + li + p.halo_lo + tv.y) * LDKs + 4 * lh + c * 32;
        const f32x4 b0 = *(const f32x4*)(brow), b1 = *(const f32x4*)(brow + 8);
        const f32x4 b2 = *(const f32x4*)(brow + 16), b3 = *(const f32x4*)(brow + 24);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][s4], b0[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][s4], b1[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][s4], b2[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3][s4], b3[s4], acc, 0, 0, 0);
    };

    for (int c0 = 0; c0 < nchunks; c0 += KBs) {
        const int kb = (nchunks - c0 < KBs) ? nchunks - c0 : KBs;
        ev_lds_barrier();                                 // the previous round's MFMAs are done with Xs
        if ((p.dbg & 16) && tid == 0 && c0 == 0 && ((p.dbg >> 10) & 3) == 2) p.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        {   // ---- all threads stage kb chunks of the X tile: loads first, then (prologue +) LDS writes.  Element i = q * NTHR + tid is
            // (row i / f4row, float4 column i % f4row): one division per thread, then a carry walk per pass — two waves share a SIMD
            // here, and a division per pass and per phase made this staging a 4 us VALU episode of an 11 us launch
            f32x4 xv[MAXPASS];
            const int adv_r = NTHR / f4row, adv_c = NTHR % f4row;
            const int r0 = tid / f4row, cc0 = tid - r0 * f4row;
            int r = r0, cc = cc0;
            asm volatile("" : "+v"(r));                   // (the per-pass row offsets are invariants of the round loop: visible, hipcc keeps their products in
                                                          // registers across the MFMA phase — four of them went to scratch at the 128-register cap of this build)
#pragma unroll
            for (int q = 0; q < MAXPASS; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q * NTHR < nf4) {
                    const int c = (c0 * 8 + cc) * 4;
                    const int gr = n0 - p.halo_lo + r;
                    if (r < xrows && c < p.Cin && cc < kb * 8 && gr >= 0 && gr < p.nrows) {
                        const unsigned xcol = (unsigned)(c >> p.isplit_log2) * p.isstride + (c & ((1 << p.isplit_log2) - 1));
                        v = ev_bload4(rX, ((unsigned)gr * p.ldx + xcol) * 4u, 0);
                    }
                }
                xv[q] = v;
                cc += adv_c; r += adv_r;
                if (cc >= f4row) { cc -= f4row; ++r; }
            }
            if ((p.dbg & 16) && tid == 0 && c0 == 0) {
                const int set = (p.dbg >> 10) & 3;
                if (set == 1) p.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            }
            r = r0; cc = cc0;
            asm volatile("" : "+v"(r));                   // (as above: the LDS write addresses)
#pragma unroll
            for (int q = 0; q < MAXPASS; ++q) {
                if (q * NTHR < nf4 && r < xrows) {
                    f32x4 v = xv[q];
                    if (p.pro_lrelu) {
                        v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                        v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                    }
                    *(f32x4*)(Xs + r * LDKs + cc * 4) = v;
                }
                cc += adv_c; r += adv_r;
                if (cc >= f4row) { cc -= f4row; ++r; }
            }
        }
        if ((p.dbg & 16) && tid == 0 && c0 == 0 && ((p.dbg >> 10) & 3) == 3) p.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        ev_lds_barrier();
        if ((p.dbg & 16) && tid == 0 && c0 == 0 && ((p.dbg >> 10) & 3) == 0) p.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        if (c0 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bq[r >> 2][r & 3];
        }
        // ---- this wave group's chunks of the round (c = ks, ks + KS, ...) x taps
        const int nmy = (kb > ks) ? ((kb - ks + KS - 1) / KS) * nact : 0;
        int it0 = 0, ti = 0, cn = ks;
        if (c0 == 0 && npre > 0) {                        // the steps whose fragments are already in registers
#pragma unroll
            for (int s = 0; s < NPRE; ++s)
                if (s < npre) mfma16(A[s], tvs[s], cs[s]);
            it0 = npre; ti = ti0; cn = cn0;
        }
        // (sparse tap lists, rounds after the first, steps beyond NPRE) weight fragments one step ahead
        auto tap_at = [&](int i) -> int2 { return ktp ? make_int2(i * p.plane_bytes, p.koff0 + i * p.kdoff) : ev_uniform(taps[i]); };
        f32x4 an[4];
        int2 tvn = make_int2(0, 0);
        if (it0 < nmy) {
            tvn = tap_at(ti);
            const unsigned ap = (unsigned)tvn.x + wbase + (unsigned)((c0 + cn) * 4) * 1024u;
            an[0] = ev_bload4(rW, wlane, ap); an[1] = ev_bload4(rW, wlane, ap + 1024u);
            an[2] = ev_bload4(rW, wlane, ap + 2048u); an[3] = ev_bload4(rW, wlane, ap + 3072u);
        }
        for (int it = it0; it < nmy; ++it) {
            const int2 tv = tvn;
            const int c = cn;
            const f32x4 a[4] = {an[0], an[1], an[2], an[3]};
            if (++ti == nact) { ti = 0; cn += KS; }
            if (it + 1 < nmy) {
                tvn = tap_at(ti);
                const unsigned ap = (unsigned)tvn.x + wbase + (unsigned)((c0 + cn) * 4) * 1024u;
                an[0] = ev_bload4(rW, wlane, ap); an[1] = ev_bload4(rW, wlane, ap + 1024u);
                an[2] = ev_bload4(rW, wlane, ap + 2048u); an[3] = ev_bload4(rW, wlane, ap + 3072u);
            }
            mfma16(a, tv, c);
        }
    }
    // ---- sum the KS partial tiles through LDS (aliases the X tile), fixed order
    if ((p.dbg & 16) && tid == 0) { if (acc[0] == 12345.678f) p.Y[0] = 1.f; p.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); }
    float* red = smem;                                    // [KS-1][TW tiles][16 regs][64 lanes]
    if (ks > 0) {
        ev_lds_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((ks - 1) * TW + tw) * 16 + r) * 64 + lane] = acc[r];
        ev_lds_barrier();
        return;                                           // s_barrier only counts surviving waves from here on
    }
    f32x16 accs[1][1];
    accs[0][0] = acc;
    auto reduce = [&]() {                                 // (the same two barriers as the ks > 0 waves)
        ev_lds_barrier();
        ev_lds_barrier();
#pragma unroll
        for (int k = 1; k < KS; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[0][0][r] += red[(((k - 1) * TW + tw) * 16 + r) * 64 + lane];
    };
    if constexpr (LEAN != 0)   // the residual / mask rows of the epilogue are requested before the reduction, not after it
        conv_epilogue_lean<1, 1, LEAN>(p, accs, smem + tw * (32 * 36), m0 + wm * 32, n0 + wn * 32, lane, -0x7fffffff, 0x7fffffff, reduce);
    else {
        reduce();
        conv_epilogue<1, 1, FULL_ACT>(p, accs, smem + tw * (32 * 36), m0 + wm * 32, n0 + wn * 32, lane);
    }
    if ((p.dbg & 16) && tid == 0) p.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}

// ---------------------------------------------------------------------------
// conv_sk32_kernel: conv_gemm_sk_kernel<8, ., LEAN, 1> for the layers a batch-1 decode is made of, written for INSTRUCTION COUNT.
// A launch of ~140 workgroups on 256 CUs is one serial chain per workgroup, and a wave issues at most one instruction every four
// cycles (a 1024 x s_nop probe in the prologue of the general build cost 1.7 us: profiles/r02_sk32_workgroup_timeline.log), so
// the ~1500 instruction-slots the general build spends between its first instruction and its staging barrier (index arithmetic
// of a 16-pass unrolled staging with every layout option, tile-index divisions, 150 SGPR spills, kernel arguments fetched field
// by field) ARE its 3.5 us prologue — more than its MFMAs.  Host-checked preconditions make that code disappear:
//   dense layer with evenly spaced taps (tap list = arithmetic; or the stacked [k-tap conv | 1x1 conv] pattern: kstack_mt), Cin = Kpad in {128, 256, 512, 1024} (one staging round, a thread's
//   float4 column is fixed and its rows advance by a constant), plain row-major X without prologue activation, a lean epilogue,
//   2-D grid (no tile-index division).
// Same operands, products and summation order as conv_gemm_sk_kernel<8, false, LEAN, 1>.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v);   // (defined with the normalisation kernels below)
template <int LEAN>
__global__ __launch_bounds__(512, 4) void conv_sk32_kernel(const ConvParams pk) {
    constexpr int KS = 8, NTHR = 512, MAXPASS = 16, NPRE = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ConvParams p = pk;          // one batch of scalar loads, opaque afterwards (see conv_gemm_sk_kernel)
    asm volatile("" : "+s"(p.X), "+s"(p.ldx), "+s"(p.W), "+s"(p.Kpad), "+s"(p.bias), "+s"(p.Y), "+s"(p.ldy), "+s"(p.Cout), "+s"(p.nrows), "+s"(p.S),
                      "+s"(p.P), "+s"(p.T), "+s"(p.ntaps), "+s"(p.halo_lo), "+s"(p.halo_hi));
    asm volatile("" : "+s"(p.act), "+s"(p.act_slope), "+s"(p.act_a), "+s"(p.act_b), "+s"(p.mask1), "+s"(p.R), "+s"(p.ldr), "+s"(p.accum), "+s"(p.div3),
                      "+s"(p.act2_lrelu), "+s"(p.act2_slope), "+s"(p.mask2), "+s"(p.rowmask), "+s"(p.plane_bytes), "+s"(p.koff0));
    asm volatile("" : "+s"(p.kdoff), "+s"(p.stamps), "+s"(p.dbg), "+s"(p.kstack_mt), "+s"(p.kstack_tap), "+s"(p.gn_part));
    const int tid = threadIdx.x, lane = tid & 63;
    const int ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if ((p.dbg & 16) && tid == 0) p.stamps[4 * wg + 0] = __builtin_amdgcn_s_memrealtime();
    const int LDK = p.Kpad + 4;
    const int f4_log2 = 31 - __builtin_clz(p.Kpad) - 2;   // float4 per row = Kpad / 4 = 32, 64, 128 or 256
    const int adv_r = NTHR >> f4_log2;                    // rows a pass of all threads covers: 16, 8, 4 or 2
    const int xrows = 32 + p.halo_lo + p.halo_hi;
    int nact = p.ntaps, t0 = 0;                           // this M tile's taps: t0, t0 + 1, ..., t0 + nact - 1
    if (p.kstack_mt > 0 && (int)blockIdx.x >= p.kstack_mt) { nact = 1; t0 = p.kstack_tap; }
    const int nmy = (((p.Kpad >> 5) + 7 - ks) >> 3) * nact;   // this wave's (chunk, tap) steps: chunks ks, ks + 8, ... < Kpad / 32
    // (X through a descriptor bounded by the tensor — host: nrows * ldx * 4 < 2 GB — so that offset 0x80000000 reads zeros)
    const __amdgpu_buffer_rsrc_t rW = ev_rsrc(p.W);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.nrows * p.ldx * 4, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(blockIdx.x * (p.Kpad >> 3)) * 1024u;

    // ---- weight fragments of the first NPRE steps, bias: in flight across the whole staging
    f32x4 A[NPRE][4];
    int offs[NPRE], cs[NPRE];
    int ti = 0, cn = ks;
#pragma unroll
    for (int s = 0; s < NPRE; ++s) {
        offs[s] = 0; cs[s] = 0;
        if (s < nmy) {
            offs[s] = p.koff0 + (t0 + ti) * p.kdoff; cs[s] = cn;
            const unsigned ap = (unsigned)((t0 + ti) * p.plane_bytes) + wbase + (unsigned)(cn * 4) * 1024u;
            A[s][0] = ev_bload4(rW, wlane, ap); A[s][1] = ev_bload4(rW, wlane, ap + 1024u);
            A[s][2] = ev_bload4(rW, wlane, ap + 2048u); A[s][3] = ev_bload4(rW, wlane, ap + 3072u);
            if (++ti == nact) { ti = 0; cn += KS; }
        }
    }
    // ---- X tile: thread = (row tid >> f4_log2 [+ adv_r per pass], float4 column tid & (f4 - 1)); rows outside the tensor read
    // an out-of-range offset (the descriptor returns 0): no branches around the loads
    {
        f32x4 xv[MAXPASS];
        const int r0 = tid >> f4_log2, c4 = tid & ((1 << f4_log2) - 1);
        const int npass = (xrows - r0 + adv_r - 1) >> (9 - f4_log2);   // (per thread)
        int gr = n0 - p.halo_lo + r0;
        unsigned xo = ((unsigned)gr * (unsigned)p.ldx + (unsigned)c4 * 4u) * 4u;
        const unsigned xstep = (unsigned)(adv_r * p.ldx) * 4u;
#pragma unroll
        for (int q = 0; q < MAXPASS; ++q) {
            if (q * adv_r < xrows) {
                xv[q] = ev_bload4(rX, (q < npass && (unsigned)gr < (unsigned)p.nrows) ? xo : 0x80000000u, 0);
                gr += adv_r; xo += xstep;
            }
        }
        float* dst = smem + r0 * LDK + c4 * 4;
        const int dstep = adv_r * LDK;
#pragma unroll
        for (int q = 0; q < MAXPASS; ++q) {
            if (q * adv_r < xrows) {
                if (q < npass) *(f32x4*)dst = xv[q];
                dst += dstep;
            }
        }
    }
    // bias (requested once the staging registers are free: the kernel stays within 128 VGPRs = two workgroups per CU, so a launch
    // of 257-512 tiles is still one round)
    f32x4 bq[4];
    {
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (ks == 0 && p.bias) ? p.Cout * 4 : 0, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = ev_bload4(rB, (unsigned)(m0 + 8 * g + 4 * lh) * 4u, 0);
    }
    ev_lds_barrier();
    if ((p.dbg & 16) && tid == 0) p.stamps[4 * wg + 1] = __builtin_amdgcn_s_memrealtime();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bq[r >> 2][r & 3];
    auto mfma16 = [&](const f32x4 (&a)[4], int off, int c) {
        const float* brow = smem + (li + p.halo_lo + off) * LDK + 4 * lh + c * 32;
        const f32x4 b0 = *(const f32x4*)(brow), b1 = *(const f32x4*)(brow + 8);
        const f32x4 b2 = *(const f32x4*)(brow + 16), b3 = *(const f32x4*)(brow + 24);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][s4], b0[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][s4], b1[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2][s4], b2[s4], acc, 0, 0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3][s4], b3[s4], acc, 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NPRE; ++s)
        if (s < nmy) mfma16(A[s], offs[s], cs[s]);
    if (nmy > NPRE) {                                     // (Cin = 512 with three taps: steps 4 and 5) one step ahead
        f32x4 an[4];
        int offn = p.koff0 + (t0 + ti) * p.kdoff;
        {
            const unsigned ap = (unsigned)((t0 + ti) * p.plane_bytes) + wbase + (unsigned)(cn * 4) * 1024u;
            an[0] = ev_bload4(rW, wlane, ap); an[1] = ev_bload4(rW, wlane, ap + 1024u);
            an[2] = ev_bload4(rW, wlane, ap + 2048u); an[3] = ev_bload4(rW, wlane, ap + 3072u);
        }
#pragma unroll 1
        for (int it = NPRE; it < nmy; ++it) {
            const int off = offn, c = cn;
            const f32x4 a[4] = {an[0], an[1], an[2], an[3]};
            if (++ti == nact) { ti = 0; cn += KS; }
            if (it + 1 < nmy) {
                offn = p.koff0 + (t0 + ti) * p.kdoff;
                const unsigned ap = (unsigned)((t0 + ti) * p.plane_bytes) + wbase + (unsigned)(cn * 4) * 1024u;
                an[0] = ev_bload4(rW, wlane, ap); an[1] = ev_bload4(rW, wlane, ap + 1024u);
                an[2] = ev_bload4(rW, wlane, ap + 2048u); an[3] = ev_bload4(rW, wlane, ap + 3072u);
            }
            mfma16(a, off, c);
        }
    }
    if ((p.dbg & 16) && tid == 0) { if (acc[0] == 12345.678f) p.Y[0] = 1.f; p.stamps[4 * wg + 2] = __builtin_amdgcn_s_memrealtime(); }
    // ---- sum the eight partial tiles through LDS (aliases the X tile), fixed order 0 + 1 + ... + 7
    float* red = smem;                                    // [7][16 regs][64 lanes]
    if (ks > 0) {
        ev_lds_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((ks - 1) * 16 + r) * 64 + lane] = acc[r];
        ev_lds_barrier();
        return;
    }
    f32x16 accs[1][1];
    accs[0][0] = acc;
    auto reduce = [&]() {
        ev_lds_barrier();
        ev_lds_barrier();
#pragma unroll 1
        for (int k = 0; k < KS - 1; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) accs[0][0][r] += red[(k * 16 + r) * 64 + lane];
        if (p.gn_part && (p.kstack_mt == 0 || (int)blockIdx.x < p.kstack_mt)) {
            // (host: one utterance, plain epilogue — the tile is stored as it stands) statistics of this (32 frames x one group)
            // tile over its frames inside [0, T): lane = frame li, 16 of the group's channels per lane half
            const int t = n0 + li - p.P;
            const bool valid = t >= 0 && t < p.T;
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += accs[0][0][r];
            const float cnt = wave_sum(valid ? 16.f : 0.f);
            const float mean = cnt > 0.f ? wave_sum(valid ? sm : 0.f) / cnt : 0.f;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = accs[0][0][r] - mean; q += d * d; }
            q = wave_sum(valid ? q : 0.f);
            if (lane == 0) {
                f32x4 o = {cnt, mean, q, 0.f};
                *(f32x4*)(p.gn_part + ((size_t)blockIdx.y * 8 + blockIdx.x) * 4) = o;
            }
        }
    };
    conv_epilogue_lean<1, 1, LEAN>(p, accs, smem, m0, n0, lane, -0x7fffffff, 0x7fffffff, reduce);
    if ((p.dbg & 16) && tid == 0) p.stamps[4 * wg + 3] = __builtin_amdgcn_s_memrealtime();
}

// ---------------------------------------------------------------------------
// resblock_pair_kernel: one (c1, c2) pair of HiFi-GAN ResBlock1 (hifigan/models.py:90-97) in ONE launch:
//     x' = c2( lrelu( c1( lrelu(x) ) + b1 ) ) + b2 + x          c1: k taps, dilation d;  c2: k taps, dilation 1
// for the narrow stages (C = 32 / 64), where a single conv is HBM-coupled (16-37 FLOP/B: one activation read,
// one residual read and one write per 2*C*C*k FLOP).  The intermediate lrelu(c1(.)) tile never leaves the CU: it
// is written to LDS in the [frame][channel] operand layout and consumed by the second MFMA loop, so a pair costs
// one read of x (+ an L2-hot re-read of the tile for the residual) and one write of x' — 2.5x less HBM traffic, half
// the launches and twice the matrix work per memory phase.
//   compute rows r in [0, NT): global row g0 + r, g0 = n0 - h2;  c1 is evaluated on all NT rows, c2's outputs are
//   valid for r in [h2, NT - h2) = global [n0, n0 + NT - 2*h2) (tiles overlap by the c2 halo).  y1 rows outside the
//   utterance are zeroed: they are c2's zero padding.
// ---------------------------------------------------------------------------
struct PairParams {
    ConvParams c2;                 // c2 + epilogue view: X = x (input), W/bias/taplist = c2's, R = x, Y = output, flags
    const float* W1; const float* b1; const int2* taplist1; int ntaps1;
    const void* W1h; float w1h_scale;   // resblock_pair_h16_kernel: c1's weights times w1h_scale as two fp16 pieces (c2's: c2.Wh, c2.wh_scale)
    const void* W1q;               // resblock_pair_h16q_kernel: the same pieces in the 16 x 16 x 32 fragment order (c2's: c2.Wq)
    const void* W1x;               // resblock_pair_split_kernel: c1's weights as three bf16 pieces (c2's: c2.Wx)
    int h1, h2;                    // halos of c1 (dilated) and c2
    float mid_slope;               // leaky-relu slope between the convs
    int out_rows;                  // NT - 2*h2
};

// (LDS admits 3 / 5 / 3 workgroups per CU for C = 32 / 64 / 128: keep the register allocation from going below that)
template <int WAVES_M, int WAVES_N, int LEAN = 0>
__global__ __launch_bounds__(256, WAVES_M == 2 ? 4 : 3) void resblock_pair_kernel(const PairParams pp) {
    constexpr int TM = 1, TN = 2;
    constexpr int NT = WAVES_N * TN * 32;               // compute rows per workgroup
    constexpr int NCH = WAVES_M;                        // 32-channel chunks (C = 32 * WAVES_M)
    constexpr int XROWS = NT + EV_HALO;
    constexpr int YROWS = NT + 16;                      // + 2*h2 (<= 10) rounded up
    constexpr int XPASS = XROWS / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    const ConvParams& p = pp.c2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                   // phase 1: [XROWS][EV_LDK]
    float* Ys = smem;                                   // phase 2 (aliases Xs): [NCH][YROWS][EV_LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;

    const int nt = ev_xcd_remap(blockIdx.x, p.ntiles);
    const int n0 = nt * pp.out_rows;                    // first output row of this tile
    const int g0 = n0 - pp.h2;                          // global row of compute row 0
    {   // tiles whose output window holds no storable row do nothing
        const int s0 = n0 % p.S, t_first = s0 - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - s0 + p.P;
        if (dist >= pp.out_rows || n0 + dist >= p.nrows) return;
    }

    const int KG8 = p.Kpad >> 3;
    const unsigned wlane = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(p.X);
    f32x16 acc[TM][TN];
    f32x4 A0[TM], A1[TM], A2[TM], A3[TM], B0[TN], B1[TN];

    // one conv phase: K loop over NCH chunks x taps with the A/B fragment pipeline of conv_gemm_kernel
    // (the accumulators start from the layer's bias when `binit` is given: C/D register 4g+e is channel 8g + 4*half + e)
    auto phase = [&](const float* Wf, const int2* tl, int nact, const float* binit, auto&& chunk_base) {
        const __amdgpu_buffer_rsrc_t rW = ev_rsrc(Wf);
        auto a_off = [&](int tap_bytes, int kg8) -> unsigned { return (unsigned)tap_bytes + (unsigned)(wm * KG8 + kg8) * 1024u; };
        auto ldAp = [&](f32x4 (&dst)[TM], unsigned aoff) { dst[0] = ev_bload4(rW, wlane, aoff); };
        auto ldB = [&](f32x4 (&dst)[TN], const float* brow, int kg) {
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[j] = *(const f32x4*)(brow + j * 32 * EV_LDK + kg * 8);
        };
        auto mma = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][s4], b[j][s4], acc[0][j], 0, 0, 0);
        };
        {
            f32x4 bq[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                bq[g] = binit ? *(const f32x4*)(binit + wm * 32 + 8 * g + 4 * lh) : z;
            }
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][b][r] = bq[r >> 2][r & 3];
        }
        const int2 tlv = (lane < nact) ? tl[lane] : make_int2(0, 0);
        const int2 tv_first = ev_tap_at(tlv, 0);
        {   // weight fragments one tap (four k-groups) ahead, as in conv_gemm_kernel
            const unsigned a0 = a_off(tv_first.x, 0);
            ldAp(A0, a0); ldAp(A1, a0 + 1024u); ldAp(A2, a0 + 2048u); ldAp(A3, a0 + 3072u);
        }
        for (int ch = 0; ch < NCH; ++ch) {
            const float* bbase = chunk_base(ch);        // LDS row 0 of this chunk for this lane (stages + barriers inside)
            int tap = tv_first.x;
            const float* brow = bbase + tv_first.y * EV_LDK;
            ldB(B0, brow, 0);
            for (int ti = 0; ti < nact; ++ti) {
                const bool last_tap = (ti + 1 == nact);
                const int2 ntv = last_tap ? tv_first : ev_tap_at(tlv, ti + 1);
                const float* nbrow = bbase + ntv.y * EV_LDK;
                const bool have_next = !(last_tap && ch + 1 == NCH);
                const unsigned nap = have_next ? a_off(ntv.x, last_tap ? ch * 4 + 4 : ch * 4) : a_off(tv_first.x, 0);   // unconditional loads
                ldB(B1, brow, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(A0, B0);
                __builtin_amdgcn_sched_barrier(0);
                ldAp(A0, nap);
                ldB(B0, brow, 2);
                __builtin_amdgcn_sched_barrier(0);
                mma(A1, B1);
                __builtin_amdgcn_sched_barrier(0);
                ldAp(A1, nap + 1024u);
                ldB(B1, brow, 3);
                __builtin_amdgcn_sched_barrier(0);
                mma(A2, B0);
                __builtin_amdgcn_sched_barrier(0);
                ldAp(A2, nap + 2048u);
                ldB(B0, nbrow, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(A3, B1);
                __builtin_amdgcn_sched_barrier(0);
                ldAp(A3, nap + 3072u);
                (void)tap;
                tap = ntv.x; brow = nbrow;
            }
        }
    };

    // ---------------- phase 1: c1 over lrelu(x), X tile rows [g0 - h1, g0 + NT + h1) staged per 32-channel chunk
    const int xrows = NT + 2 * pp.h1;
    phase(pp.W1, pp.taplist1, pp.ntaps1, pp.b1, [&](int ch) -> const float* {
        ev_lds_barrier();
        const int c = ch * EV_BK + sc4;
        f32x4 xv[XPASS];
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * 32 + srow;
            const int gr = g0 - pp.h1 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < xrows && gr >= 0 && gr < p.nrows) v = ev_bload4(rX, ((unsigned)gr * p.ldx + c) * 4u, 0);
            xv[q] = v;
        }
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * 32 + srow;
            f32x4 v = xv[q];
            v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
            v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
            if (r < xrows) *(f32x4*)(Xs + r * EV_LDK + sc4) = v;
        }
        ev_lds_barrier();
        return Xs + (wn * (TN * 32) + li + pp.h1) * EV_LDK + 4 * lh;
    });

    // ---------------- y1 = lrelu(c1 + b1), zero outside the utterance, into LDS [chunk = wm][row + h2][channel]
    ev_lds_barrier();                                    // every wave is done reading Xs
    {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + li;
            const int n = g0 + r;
            const int t = (n >= 0 && n < p.nrows) ? (n % p.S) - p.P : -1;
            const float inside = (t >= 0 && t < p.T) ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ev_lrelu(acc[0][j][4 * g + e], pp.mid_slope) * inside;
                *(f32x4*)(Ys + (wm * YROWS + r + pp.h2) * EV_LDK + 8 * g + 4 * lh) = v;
            }
        }
        // the 2*h2 border rows only feed outputs outside the stored window, but must be finite
        if (tid < 2 * pp.h2 * 8 * NCH) {
            const int chn = tid / (2 * pp.h2 * 8), rem = tid % (2 * pp.h2 * 8);
            const int br = rem / 8, c4 = (rem % 8) * 4;
            const int row = br < pp.h2 ? br : NT + br;   // rows [0,h2) and [NT+h2, NT+2*h2)
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(Ys + (chn * YROWS + row) * EV_LDK + c4) = z;
        }
    }
    ev_lds_barrier();

    // ---------------- phase 2: c2 over the LDS-resident y1 (tap i reads rows r + i)
    phase(p.W, p.taplist, p.ntaps, LEAN ? p.bias : nullptr, [&](int ch) -> const float* {
        return Ys + (ch * YROWS + wn * (TN * 32) + li + pp.h2) * EV_LDK + 4 * lh;
    });

    // ---------------- epilogue: + b2 + x (residual re-read, L2-hot), optional running resblock mean, window [n0, n0 + out_rows)
    if constexpr (LEAN != 0) conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + pp.out_rows);
    else conv_epilogue<TM, TN, false>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + pp.out_rows);
}

// ---------------------------------------------------------------------------
// resblock_pair_split_kernel: resblock_pair_kernel on the bf16 matrix pipe (see conv_split_kernel: six exact bf16 products per
// element pair, fp32 accumulation).  Same tile geometry (compute rows NT, waves = channel tiles x row tiles, wave tile 32 channels
// x 64 rows).  LDS rows hold ALL C channels as three bf16 planes (row stride 6 C + 16 bytes: 208 / 400 / 784, odd multiples of 16),
// so x is staged — and split — once, and the intermediate y1 = lrelu(c1 + b1) is split as it leaves the accumulators (register
// 4 g + e of a lane = four consecutive channels of one row: one 8-byte store per piece).  K loop: the (tap, 16-deep slab) sequence
// in groups of two slabs, weight fragments two groups ahead in four named sets, activation fragments double-buffered by slab.
// ---------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int LEAN, int TERMS = 6>
__global__ __launch_bounds__(256, 2) void resblock_pair_split_kernel(const PairParams pp) {
    constexpr int TM = 1, TN = 2;
    constexpr int C = 32 * WAVES_M;
    constexpr int NT = WAVES_N * TN * 32;
    constexpr int RSB = 6 * C + 16;                     // LDS row stride in bytes
    constexpr int NS = C / 16, H = NS / 2;              // slabs per tap; two-slab groups per tap
    constexpr int TPR = C / 4, RPS = 256 / TPR;         // staging: threads per row, rows per pass
    constexpr int XPASS = (NT + EV_HALO) / RPS;
    constexpr int XG = XPASS > 8 ? XPASS / 2 : XPASS;
    static_assert(WAVES_M * WAVES_N == 4 && XPASS % XG == 0, "4 waves per workgroup");
    const ConvParams& p = pp.c2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // phase 1: [NT + 2 h1][RSB];  phase 2 (aliased): y1 [NT + 2 h2][RSB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;

    const int nt = ev_xcd_remap(blockIdx.x, p.ntiles);
    const int n0 = nt * pp.out_rows;
    const int g0 = n0 - pp.h2;
    {   // tiles whose output window holds no storable row do nothing
        const int s0 = n0 % p.S, t_first = s0 - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - s0 + p.P;
        if (dist >= pp.out_rows || n0 + dist >= p.nrows) return;
    }
    const int KG16 = p.Kpad >> 4;
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(wm * KG16) * 3072u;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(p.X);
    f32x16 acc[TM][TN];
    f32x4 A0[3][TM], A1[3][TM], A2[3][TM], A3[3][TM], B0[3][TN], B1[3][TN];

    auto ldAp = [&](const __amdgpu_buffer_rsrc_t& rW, f32x4 (&dst)[3][TM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) dst[pc][0] = ev_bload4(rW, wlane, aoff + (unsigned)(pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[3][TN], const char* brow, int slab) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * RSB + pc * (2 * C) + slab * 32);
    };
    // group g of a phase = slabs 2 (g % H), 2 (g % H) + 1 of tap g / H; byte offset of its first weight fragment (tap-list entries
    // carry the fp32 plane offset: a split plane is 1.5 x that)
    auto g_off = [&](int2 tlv, int g, int ngroups) -> unsigned {
        const int gg = g < ngroups ? g : 0;             // (beyond the phase: a harmless re-read)
        const unsigned tb = (unsigned)__builtin_amdgcn_readlane(tlv.x, gg / H);
        return tb + (tb >> 1) + wbase + (unsigned)(2 * (gg % H)) * 3072u;
    };
    auto g_row = [&](int2 tlv, int g, int ngroups) -> int {
        const int gg = g < ngroups ? g : 0;
        return __builtin_amdgcn_readlane(tlv.y, gg / H);
    };
    auto acc_init = [&](const float* binit) {
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4*)(binit + wm * 32 + 8 * g + 4 * lh);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][b][r] = bq[r >> 2][r & 3];
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups) {
        const unsigned o0 = g_off(tlv, 0, ngroups), o1 = g_off(tlv, 1, ngroups);
        ldAp(rW, A0, o0); ldAp(rW, A1, o0 + 3072u); ldAp(rW, A2, o1); ldAp(rW, A3, o1 + 3072u);
    };
    // the K loop of one conv: `bbase` = this lane's LDS row 0 (halo already added), fragments of groups 0 and 1 already requested
    auto kloop = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups, const char* bbase) {
        auto group = [&](f32x4 (&Aa)[3][TM], f32x4 (&Ab)[3][TM], int g) {
            const char* brow = bbase + g_row(tlv, g, ngroups) * RSB;
            const char* nbrow = bbase + g_row(tlv, g + 1, ngroups) * RSB;
            const int s0 = 2 * (g % H), ns0 = 2 * ((g + 1) % H);
            const unsigned nap = g_off(tlv, g + 2, ngroups);
            ldB(B1, brow, s0 + 1);
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, Aa, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Aa, nap);
            ldB(B0, nbrow, ns0);
            __builtin_amdgcn_sched_barrier(0);
            evx_mma<TM, TN, TERMS>(acc, Ab, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Ab, nap + 3072u);
        };
        ldB(B0, bbase + g_row(tlv, 0, ngroups) * RSB, 0);
        int g = 0;
        for (; g + 1 < ngroups; g += 2) { group(A0, A1, g); group(A2, A3, g + 1); }
        if (g < ngroups) group(A0, A1, g);
    };

    // ---------------- phase 1: c1 over lrelu(x); X tile rows [g0 - h1, g0 + NT + h1), all channels, split into three planes
    const __amdgpu_buffer_rsrc_t rW1 = ev_rsrc(pp.W1x), rW2 = ev_rsrc(p.Wx);
    const int ng1 = pp.ntaps1 * H, ng2 = p.ntaps * H;
    const int2 tlv1 = (lane < pp.ntaps1) ? pp.taplist1[lane] : make_int2(0, 0);
    const int2 tlv2 = (lane < p.ntaps) ? p.taplist[lane] : make_int2(0, 0);
    ring_fill(rW1, tlv1, ng1);
    acc_init(pp.b1);
    {
        const int xrows = NT + 2 * pp.h1;
#pragma unroll
        for (int q0 = 0; q0 < XPASS; q0 += XG) {
            if (q0 * RPS >= xrows) continue;
            f32x4 xg[XG];
#pragma unroll
            for (int q = 0; q < XG; ++q) {
                const int r = (q0 + q) * RPS + srow;
                const int gr = g0 - pp.h1 + r;
                xg[q] = ev_bload4(rX, ((r < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u, 0);   // (row 0 is a zero pad row)
            }
#pragma unroll
            for (int q = 0; q < XG; ++q) {
                const int r = (q0 + q) * RPS + srow;
                f32x4 v = xg[q];
                v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
                v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
                uint2 q0v, q1v, q2v;
                evx_split4(v, q0v, q1v, q2v);
                if (r < xrows) {
                    char* dst = Xb + r * RSB + sc4 * 2;
                    *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v; *(uint2*)(dst + 4 * C) = q2v;
                }
            }
        }
    }
    ev_lds_barrier();
    kloop(rW1, tlv1, ng1, Xb + (wn * (TN * 32) + li + pp.h1) * RSB + 16 * lh);
    ring_fill(rW2, tlv2, ng2);                           // c2's first fragments fly under the hand-over below

    // ---------------- y1 = lrelu(c1 + b1), zero outside the utterance, split, into LDS rows r + h2
    ev_lds_barrier();                                    // every wave is done reading the X tile
    {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + li;
            const int n = g0 + r;
            const int t = (n >= 0 && n < p.nrows) ? (n % p.S) - p.P : -1;
            const float inside = (t >= 0 && t < p.T) ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ev_lrelu(acc[0][j][4 * g + e], pp.mid_slope) * inside;
                uint2 q0v, q1v, q2v;
                evx_split4(v, q0v, q1v, q2v);
                char* dst = Xb + (r + pp.h2) * RSB + (wm * 32 + 8 * g + 4 * lh) * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v; *(uint2*)(dst + 4 * C) = q2v;
            }
        }
        // the 2 h2 border rows only feed outputs outside the stored window, but must be finite: zero all three planes
        for (int i = tid; i < 2 * pp.h2 * (6 * C / 16); i += 256) {
            const int br = i / (6 * C / 16), c16 = i % (6 * C / 16);
            const int row = br < pp.h2 ? br : NT + br;
            uint4 z = {0u, 0u, 0u, 0u};
            *(uint4*)(Xb + row * RSB + c16 * 16) = z;
        }
    }
    acc_init(LEAN ? p.bias : pp.b1);
    ev_lds_barrier();

    // ---------------- phase 2: c2 over the LDS-resident y1 (tap offset t reads rows r + h2 + t)
    kloop(rW2, tlv2, ng2, Xb + (wn * (TN * 32) + li + pp.h2) * RSB + 16 * lh);

    conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + pp.out_rows);
}

// ---------------------------------------------------------------------------
// resblock_pair_h16_kernel: the fused ResBlock pair with fp16 pieces (conv_h16_kernel: two block-scaled fp16 pieces per operand, three
// products per fp32 product).  Scales, all exact powers of two: the layers' weight scales from the loader; sx from the maximum of the X
// tile, which is loaded into registers ONCE (maximum, then leaky-relu, scaling, split, LDS); sy from the maximum of the intermediate
// lrelu(c1 + b1) over the workgroup's tile, exchanged through LDS at the barrier the hand-over has anyway.  LDS row = two fp16 planes of
// all C channels + 16 bytes (144 / 272 / 528 = 9, 17, 33 x 16).
// ---------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int LEAN>
__global__ __launch_bounds__(256, 2) void resblock_pair_h16_kernel(const PairParams pp) {
    constexpr int TM = 1, TN = 2;
    constexpr int C = 32 * WAVES_M;
    constexpr int NT = WAVES_N * TN * 32;
    constexpr int RSB = 4 * C + 16;                     // LDS row stride in bytes
    constexpr int NS = C / 16, H = NS / 2;              // slabs per tap; two-slab groups per tap
    constexpr int TPR = C / 4, RPS = 256 / TPR;         // staging: threads per row, rows per pass
    constexpr int XPASS = (NT + EV_HALO) / RPS;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    const ConvParams& p = pp.c2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // phase 1: [NT + 2 h1][RSB];  phase 2 (aliased): y1 [NT + 2 h2][RSB]
    float* red = smem + ((NT + EV_HALO) * RSB) / 4;     // 16 floats behind the tiles: the waves' maxima (x: 0..3, y1: 4..7; their finite-only repeats: 8..11, 12..15)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;

    const int nt = ev_xcd_remap(blockIdx.x, p.ntiles);
    const int n0 = nt * pp.out_rows;
    const int g0 = n0 - pp.h2;
    {   // tiles whose output window holds no storable row do nothing
        const int s0 = n0 % p.S, t_first = s0 - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - s0 + p.P;
        if (dist >= pp.out_rows || n0 + dist >= p.nrows) return;
    }
    const int KG16 = p.Kpad >> 4;
    const unsigned wlane = (unsigned)lane * 16u;
    EvAmax am = ev_amax_begin(p, g0 + wn * (TN * 32), TN * 32);   // (the residual's / running sum's bounds: requested now, used behind the epilogue)
    const unsigned wbase = (unsigned)(wm * KG16) * 2048u;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(p.X);
    f32x16 acc[TM][TN];
    f32x4 A0[2][TM], A1[2][TM], A2[2][TM], A3[2][TM], B0[2][TN], B1[2][TN];

    auto ldAp = [&](const __amdgpu_buffer_rsrc_t& rW, f32x4 (&dst)[2][TM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dst[pc][0] = ev_bload4(rW, wlane, aoff + (unsigned)(pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[2][TN], const char* brow, int slab) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * RSB + pc * (2 * C) + slab * 32);
    };
    // group g of a phase = slabs 2 (g % H), 2 (g % H) + 1 of tap g / H (a tap's fp16 plane is as large as its fp32 plane: the tap list's
    // byte offsets apply as they are)
    auto g_off = [&](int2 tlv, int g, int ngroups) -> unsigned {
        const int gg = g < ngroups ? g : 0;             // (beyond the phase: a harmless re-read)
        return (unsigned)__builtin_amdgcn_readlane(tlv.x, gg / H) + wbase + (unsigned)(2 * (gg % H)) * 2048u;
    };
    auto g_row = [&](int2 tlv, int g, int ngroups) -> int {
        const int gg = g < ngroups ? g : 0;
        return __builtin_amdgcn_readlane(tlv.y, gg / H);
    };
    auto acc_init = [&](const float* binit, float unit) {   // bias in accumulator units
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4*)(binit + wm * 32 + 8 * g + 4 * lh) * unit;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][b][r] = bq[r >> 2][r & 3];
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups) {
        const unsigned o0 = g_off(tlv, 0, ngroups), o1 = g_off(tlv, 1, ngroups);
        ldAp(rW, A0, o0); ldAp(rW, A1, o0 + 2048u); ldAp(rW, A2, o1); ldAp(rW, A3, o1 + 2048u);
    };
    auto kloop = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups, const char* bbase) {
        auto group = [&](f32x4 (&Aa)[2][TM], f32x4 (&Ab)[2][TM], int g) {
            const char* brow = bbase + g_row(tlv, g, ngroups) * RSB;
            const char* nbrow = bbase + g_row(tlv, g + 1, ngroups) * RSB;
            const int s0 = 2 * (g % H), ns0 = 2 * ((g + 1) % H);
            const unsigned nap = g_off(tlv, g + 2, ngroups);
            ldB(B1, brow, s0 + 1);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, Aa, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Aa, nap);
            ldB(B0, nbrow, ns0);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, Ab, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Ab, nap + 2048u);
        };
        ldB(B0, bbase + g_row(tlv, 0, ngroups) * RSB, 0);
        int g = 0;
        for (; g + 1 < ngroups; g += 2) { group(A0, A1, g); group(A2, A3, g + 1); }
        if (g < ngroups) group(A0, A1, g);
    };
    auto wg_max = [&](float mx, int slot) -> float {        // workgroup maximum through LDS (one barrier)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[slot + wave] = mx;
        ev_lds_barrier();
        return fmaxf(fmaxf(red[slot], red[slot + 1]), fmaxf(red[slot + 2], red[slot + 3]));
    };

    // ---------------- phase 1: c1 over lrelu(x); X tile rows [g0 - h1, g0 + NT + h1), all channels, loaded once
    const __amdgpu_buffer_rsrc_t rW1 = ev_rsrc(pp.W1h), rW2 = ev_rsrc(p.Wh);
    const int ng1 = pp.ntaps1 * H, ng2 = p.ntaps * H;
    const int2 tlv1 = (lane < pp.ntaps1) ? pp.taplist1[lane] : make_int2(0, 0);
    const int2 tlv2 = (lane < p.ntaps) ? p.taplist[lane] : make_int2(0, 0);
    ring_fill(rW1, tlv1, ng1);
    float sx;
    {
        const int xrows = NT + 2 * pp.h1;
        f32x4 xg[XPASS];
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * RPS + srow;
            const int gr = g0 - pp.h1 + r;
            xg[q] = ev_bload4(rX, ((r < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u, 0);   // (row 0 is a zero pad row)
        }
#pragma unroll
        for (int q = 0; q < XPASS; ++q) mx = fmaxf(mx, evh_absmax4(xg[q]));
        float tmx = wg_max(mx, 0);                      // (|lrelu(x)| <= |x|)
        if (!evh_is_finite(tmx)) {                      // an Inf in the tile (workgroup-uniform): the finite maximum sets the scale
            mx = 0.f;
#pragma unroll
            for (int q = 0; q < XPASS; ++q) mx = fmaxf(mx, evh_absmax4_finite(xg[q]));
            tmx = wg_max(mx, 8);
        }
        sx = evh_scale_for(tmx);
        am.rlo = am.rhi = tmx;                          // the residual IS this tile's input: its bound for the amax slots is the tile maximum just found (no slots of X needed)
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * RPS + srow;
            f32x4 v = xg[q];
            v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
            v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
            uint2 q0v, q1v;
            evh_split4(v * sx, q0v, q1v);
            if (r < xrows) {
                char* dst = Xb + r * RSB + sc4 * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
    }
    const float u1 = pp.w1h_scale * sx;
    acc_init(pp.b1, u1);
    ev_lds_barrier();
    kloop(rW1, tlv1, ng1, Xb + (wn * (TN * 32) + li + pp.h1) * RSB + 16 * lh);
    ring_fill(rW2, tlv2, ng2);                           // c2's first fragments fly under the hand-over below

    // ---------------- y1 = lrelu(c1 + b1), zero outside the utterance; its maximum over the workgroup -> sy; split into LDS rows r + h2
    float sy;
    {
        const float inv1 = 1.0f / u1;
        float my = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + li;
            const int n = g0 + r;
            const int t = (n >= 0 && n < p.nrows) ? (n % p.S) - p.P : -1;
            const float inside = (t >= 0 && t < p.T) ? inv1 : 0.f;
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
                const float v = ev_lrelu(acc[0][j][r16] * inside, pp.mid_slope);    // (back to true units; 0 outside the utterance)
                acc[0][j][r16] = v;
                my = fmaxf(my, fabsf(v));
            }
        }
        float tmy = wg_max(my, 4);                      // (the barrier inside: every wave is done reading the X tile)
        if (!evh_is_finite(tmy)) {
            my = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) { const float a = fabsf(acc[0][j][r16]); my = fmaxf(my, evh_is_finite(a) ? a : 0.f); }
            tmy = wg_max(my, 12);
        }
        sy = evh_scale_for(tmy);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[0][j][4 * g] * sy, acc[0][j][4 * g + 1] * sy, acc[0][j][4 * g + 2] * sy, acc[0][j][4 * g + 3] * sy};
                uint2 q0v, q1v;
                evh_split4(v, q0v, q1v);
                char* dst = Xb + (r + pp.h2) * RSB + (wm * 32 + 8 * g + 4 * lh) * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
        // the 2 h2 border rows only feed outputs outside the stored window, but must be finite: zero both planes
        for (int i = tid; i < 2 * pp.h2 * (4 * C / 16); i += 256) {
            const int br = i / (4 * C / 16), c16 = i % (4 * C / 16);
            const int row = br < pp.h2 ? br : NT + br;
            uint4 z = {0u, 0u, 0u, 0u};
            *(uint4*)(Xb + row * RSB + c16 * 16) = z;
        }
    }
    const float u2 = p.wh_scale * sy;
    acc_init(LEAN ? p.bias : pp.b1, u2);
    ev_lds_barrier();

    // ---------------- phase 2: c2 over the LDS-resident y1 (tap offset t reads rows r + h2 + t)
    kloop(rW2, tlv2, ng2, Xb + (wn * (TN * 32) + li + pp.h2) * RSB + 16 * lh);
    {
        const float inv2 = 1.0f / u2;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) acc[0][j][r16] *= inv2;
    }
    ev_amax_from_acc<TM, TN>(p, am, acc, g0 + wn * (TN * 32), lane);
    conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + pp.out_rows);
    ev_amax_emit(p, am, g0 + wn * (TN * 32), TN * 32, lane);
}

// ---------------------------------------------------------------------------
// resblock_chain_h16_kernel: a WHOLE ResBlock1 (hifigan/models.py:90-97: three times x = x + c2(lrelu(c1_d(lrelu(x)))), dilations 1, 3, 5) in one
// launch, for the narrow levels (C = 32 / 64) and the kernel size whose halos allow it (k = 3: 12 rows a side).  The pairs of such a chain are
// half HBM-bound: each reads its 1.1 GB input and writes 1.1 GB back (levels 3 / 4 at batch 64) for 2 x 3 taps of arithmetic.  Here the running x
// never leaves the CU: it lives in the accumulator layout in registers (fp32, 32 per lane), its leaky-relu goes to LDS as two fp16 planes
// for c1, the intermediate likewise for c2 (the hand-over of resblock_pair_h16_kernel, twice per pair), and c2's result is added to the registers.
// One tile of NT frames is read once and its middle NT - 2 halo frames are stored once: a third of the pairs' traffic, a third of their
// staging / epilogue episodes, for halo / NT more arithmetic (rows within `halo` of the tile's edges see zeros where their neighbours' frames
// would be and are not stored).  Scales as in the pair kernel: one power of two per staged tile from the workgroup's maximum, found in registers.
// ---------------------------------------------------------------------------
struct ChainParams {
    ConvParams c2;                 // geometry + epilogue view: X = the ResBlock's input, Y = its output (or the running mean it joins), flags; R unused
    const void* W1h[3]; const void* W2h[3]; float w1_scale[3], w2_scale[3];
    const float* b1[3]; const float* b2[3];
    const int2* tl1[3]; const int2* tl2[3];
    int ntaps1, ntaps2;            // taps of the dilated convs / of the plain ones
    int hb;                        // LDS rows in front of the tile (>= every conv's halo): zero, like the rows behind it
    int halo;                      // sum over the pairs of (c1's halo + c2's halo): frames a side that are computed but not stored
    int out_rows;                  // NT - 2 halo
    float mid_slope;
};
template <int WAVES_M, int WAVES_N, int LEAN>
__global__ __launch_bounds__(256, 2) void resblock_chain_h16_kernel(const ChainParams cp) {
    constexpr int TM = 1, TN = 2;
    constexpr int C = 32 * WAVES_M;
    constexpr int NT = WAVES_N * TN * 32;
    constexpr int RSB = 4 * C + 16;                     // LDS row stride in bytes: two fp16 planes of all C channels + 16
    constexpr int NS = C / 16, H = NS / 2;
    static_assert(WAVES_M * WAVES_N == 4 && (C == 32 || C == 64), "4 waves per workgroup; C = 32 / 64");
    const ConvParams& p = cp.c2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // [hb | NT | hb][RSB]
    float* red = smem + ((NT + EV_HALO) * RSB) / 4;     // 16 floats behind the tile: the waves' maxima (two alternating sets + their finite-only repeats)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;
    const int nt = ev_xcd_remap(blockIdx.x, p.ntiles);
    const int n0 = nt * cp.out_rows;
    const int g0 = n0 - cp.halo;
    {   // tiles whose output window holds no storable row do nothing
        const int s0 = n0 % p.S, t_first = s0 - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - s0 + p.P;
        if (dist >= cp.out_rows || n0 + dist >= p.nrows) return;
    }
    const int HB = cp.hb;
    const int KG16 = p.Kpad >> 4;
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)(wm * KG16) * 2048u;
    EvAmax am = ev_amax_begin(p, g0 + wn * (TN * 32), TN * 32);
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(p.X);
    // ---- the running x of this wave's 32 channels x 64 frames, accumulator layout (lane = frame, registers = channels 8 q + 4 lh + e)
    f32x16 xr[TN];
    float inside[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = g0 + wn * (TN * 32) + j * 32 + li;
        const bool in = n >= 0 && n < p.nrows;
        const int t = in ? (n % p.S) - p.P : -1;
        inside[j] = (t >= 0 && t < p.T) ? 1.f : 0.f;
        const unsigned ro = (in ? (unsigned)n * (unsigned)p.ldx : 0u) * 4u + (unsigned)(wm * 32 + 4 * lh) * 4u;     // (row 0 is a zero pad row)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = ev_bload4(rX, ro + (unsigned)(8 * q) * 4u, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[j][4 * q + e] = v[e];
        }
    }
    // the rows in front of and behind the tile: zero for the whole chain (nothing writes them)
    for (int i = tid; i < 2 * HB * (RSB / 16); i += 256) {
        const int br = i / (RSB / 16), c16 = i % (RSB / 16);
        const int row = br < HB ? br : NT + br;
        uint4 z = {0u, 0u, 0u, 0u};
        *(uint4*)(Xb + row * RSB + c16 * 16) = z;
    }
    f32x16 acc[TM][TN];
    f32x4 A0[2][TM], A1[2][TM], A2[2][TM], A3[2][TM], B0[2][TN], B1[2][TN];
    auto ldAp = [&](const __amdgpu_buffer_rsrc_t& rW, f32x4 (&dst)[2][TM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dst[pc][0] = ev_bload4(rW, wlane, aoff + (unsigned)(pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[2][TN], const char* brow, int slab) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int j = 0; j < TN; ++j) dst[pc][j] = *(const f32x4*)(brow + j * 32 * RSB + pc * (2 * C) + slab * 32);
    };
    auto g_off = [&](int2 tlv, int g, int ngroups) -> unsigned {
        const int gg = g < ngroups ? g : 0;             // (beyond the phase: a harmless re-read)
        return (unsigned)__builtin_amdgcn_readlane(tlv.x, gg / H) + wbase + (unsigned)(2 * (gg % H)) * 2048u;
    };
    auto g_row = [&](int2 tlv, int g, int ngroups) -> int {
        const int gg = g < ngroups ? g : 0;
        return __builtin_amdgcn_readlane(tlv.y, gg / H);
    };
    auto acc_init = [&](const float* binit, float unit) {   // bias in accumulator units
        f32x4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4*)(binit + wm * 32 + 8 * g + 4 * lh) * unit;
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][b][r] = bq[r >> 2][r & 3];
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups) {
        const unsigned o0 = g_off(tlv, 0, ngroups), o1 = g_off(tlv, 1, ngroups);
        ldAp(rW, A0, o0); ldAp(rW, A1, o0 + 2048u); ldAp(rW, A2, o1); ldAp(rW, A3, o1 + 2048u);
    };
    auto kloop = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int ngroups, const char* bbase) {
        auto group = [&](f32x4 (&Aa)[2][TM], f32x4 (&Ab)[2][TM], int g) {
            const char* brow = bbase + g_row(tlv, g, ngroups) * RSB;
            const char* nbrow = bbase + g_row(tlv, g + 1, ngroups) * RSB;
            const int s0 = 2 * (g % H), ns0 = 2 * ((g + 1) % H);
            const unsigned nap = g_off(tlv, g + 2, ngroups);
            ldB(B1, brow, s0 + 1);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, Aa, B0);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Aa, nap);
            ldB(B0, nbrow, ns0);
            __builtin_amdgcn_sched_barrier(0);
            evh_mma<TM, TN>(acc, Ab, B1);
            __builtin_amdgcn_sched_barrier(0);
            ldAp(rW, Ab, nap + 2048u);
        };
        ldB(B0, bbase + g_row(tlv, 0, ngroups) * RSB, 0);
        int g = 0;
        for (; g + 1 < ngroups; g += 2) { group(A0, A1, g); group(A2, A3, g + 1); }
        if (g < ngroups) group(A0, A1, g);
    };
    auto wg_max = [&](float mx, int slot) -> float {        // workgroup maximum through LDS (one barrier)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[slot + wave] = mx;
        ev_lds_barrier();
        return fmaxf(fmaxf(red[slot], red[slot + 1]), fmaxf(red[slot + 2], red[slot + 3]));
    };
    // this lane's values (true units, any prologue applied) -> maximum over the workgroup -> scale -> two fp16 planes at rows r + HB.
    // The barrier inside wg_max is also what lets the planes be overwritten: every wave has left the K loop that read them.
    auto hand_over = [&](f32x16 (&v)[TN], int slot) -> float {
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(fabsf(v[j][r]), fabsf(v[j][r + 1])));
        float tmx = wg_max(mx, slot);
        if (!evh_is_finite(tmx)) {                      // an Inf in the tile (workgroup-uniform slow path): the finite maximum sets the scale
            mx = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float a = fabsf(v[j][r]); mx = fmaxf(mx, evh_is_finite(a) ? a : 0.f); }
            tmx = wg_max(mx, 8 + slot);
        }
        const float sc = evh_scale_for(tmx);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = {v[j][4 * g] * sc, v[j][4 * g + 1] * sc, v[j][4 * g + 2] * sc, v[j][4 * g + 3] * sc};
                uint2 q0v, q1v;
                evh_split4(w, q0v, q1v);
                char* dst = Xb + (r + HB) * RSB + (wm * 32 + 8 * g + 4 * lh) * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
        return sc;
    };
    const char* bbase = Xb + (wn * (TN * 32) + li + HB) * RSB + 16 * lh;
    const int ng1 = cp.ntaps1 * H, ng2 = cp.ntaps2 * H;
#pragma unroll 1
    for (int m = 0; m < 3; ++m) {
        const __amdgpu_buffer_rsrc_t rW1 = ev_rsrc(cp.W1h[m]), rW2 = ev_rsrc(cp.W2h[m]);
        const int2 tlv1 = (lane < cp.ntaps1) ? cp.tl1[m][lane] : make_int2(0, 0);
        const int2 tlv2 = (lane < cp.ntaps2) ? cp.tl2[m][lane] : make_int2(0, 0);
        ring_fill(rW1, tlv1, ng1);                          // c1's first fragments fly under the hand-over
        // ---- lrelu(x) -> planes; c1
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = ev_lrelu(xr[j][r], p.pro_slope);
        const float sx = hand_over(acc[0], 0);
        const float u1 = cp.w1_scale[m] * sx;
        acc_init(cp.b1[m], u1);
        ev_lds_barrier();
        kloop(rW1, tlv1, ng1, bbase);
        ring_fill(rW2, tlv2, ng2);
        // ---- y1 = lrelu(c1 + b1), zero outside the utterance -> planes; c2
        {
            const float inv1 = 1.0f / u1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float f = inv1 * inside[j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] = ev_lrelu(acc[0][j][r] * f, cp.mid_slope);
            }
        }
        const float sy = hand_over(acc[0], 4);
        const float u2 = cp.w2_scale[m] * sy;
        acc_init(cp.b2[m], u2);
        ev_lds_barrier();
        kloop(rW2, tlv2, ng2, bbase);
        // ---- x += c2 + b2 (zero outside the utterance: the next conv must see the padding the stored tensor would have)
        {
            const float inv2 = 1.0f / u2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[j][r] = fmaf(acc[0][j][r], inv2, xr[j][r]) * inside[j];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = xr[j];
    ev_amax_from_acc<TM, TN>(p, am, acc, g0 + wn * (TN * 32), lane);
    conv_epilogue_lean<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + cp.out_rows);
    ev_amax_emit(p, am, g0 + wn * (TN * 32), TN * 32, lane);
}

// ---------------------------------------------------------------------------
// resblock_pair_h16q_kernel: resblock_pair_h16_kernel with both K loops on v_mfma_f32_16x16x32_f16 (see conv_h16_kernel<..., Q = 1>: the 16 x 16
// shape holds a higher clock under the chip's power limit).  A "group" of the K loops — two 16-deep slabs of one tap — is exactly one 32-deep
// step here; a wave's 32 channels x 64 frames are 2 x 4 tiles of 16 x 16.  Weights: W1q / c2.Wq (16 x 16 x 32 fragment order), same pieces,
// same scales, same accumulation order over (tap, k) as the 32 x 32 form; LDS layout, staging, scale search and epilogue are unchanged.
// ---------------------------------------------------------------------------
template <int WAVES_M, int WAVES_N, int LEAN>
__global__ __launch_bounds__(256, 2) void resblock_pair_h16q_kernel(const PairParams pp) {
    constexpr int TM = 1, TN = 2, QM = 2, QN = 4;
    constexpr int C = 32 * WAVES_M;
    constexpr int NT = WAVES_N * TN * 32;
    constexpr int RSB = 4 * C + 16;                     // LDS row stride in bytes
    constexpr int H = C / 32;                           // 32-deep steps per tap
    constexpr int TPR = C / 4, RPS = 256 / TPR;         // staging: threads per row, rows per pass
    constexpr int XPASS = (NT + EV_HALO) / RPS;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    const ConvParams& p = pp.c2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // phase 1: [NT + 2 h1][RSB];  phase 2 (aliased): y1 [NT + 2 h2][RSB]
    float* red = smem + ((NT + EV_HALO) * RSB) / 4;     // 16 floats behind the tiles: the waves' maxima (x: 0..3, y1: 4..7; their finite-only repeats: 8..11, 12..15)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int f16i = lane & 15, kg = lane >> 4;         // frame inside a 16-wide tile / 8-deep k group (= channel group 4 kg .. of a C/D tile)
    const int srow = tid / TPR, sc4 = (tid % TPR) * 4;

    const int nt = ev_xcd_remap(blockIdx.x, p.ntiles);
    const int n0 = nt * pp.out_rows;
    const int g0 = n0 - pp.h2;
    {   // tiles whose output window holds no storable row do nothing
        const int s0 = n0 % p.S, t_first = s0 - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - s0 + p.P;
        if (dist >= pp.out_rows || n0 + dist >= p.nrows) return;
    }
    const unsigned wlane = (unsigned)lane * 16u;
    EvAmax am = ev_amax_begin(p, g0 + wn * (TN * 32), TN * 32);   // (the residual's / running sum's bounds: requested now, used behind the epilogue)
    const unsigned wbase = (unsigned)(wm * 2 * H) * 2048u;   // this wave's two 16-channel row tiles: m16 = 2 wm + a, H steps of 2 KiB each
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(p.X);
    f32x4 acc[QM][QN];
    f32x4 A0[2][QM], A1[2][QM], B0[2][QN], B1[2][QN];

    auto ldA = [&](const __amdgpu_buffer_rsrc_t& rW, f32x4 (&dst)[2][QM], unsigned aoff) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int a = 0; a < QM; ++a) dst[pc][a] = ev_bload4(rW, wlane, aoff + (unsigned)(a * H * 2048 + pc * 1024));
    };
    auto ldB = [&](f32x4 (&dst)[2][QN], const char* brow, int ks) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int b = 0; b < QN; ++b) dst[pc][b] = *(const f32x4*)(brow + b * 16 * RSB + pc * (2 * C) + ks * 64);
    };
    // step g of a phase = the 32-deep step g % H of tap g / H (a tap's plane is as large as its fp32 plane: the tap list's byte offsets apply)
    auto g_off = [&](int2 tlv, int g, int nsteps) -> unsigned {
        const int gg = g < nsteps ? g : 0;              // (beyond the phase: a harmless re-read)
        return (unsigned)__builtin_amdgcn_readlane(tlv.x, gg / H) + wbase + (unsigned)(gg % H) * 2048u;
    };
    auto g_row = [&](int2 tlv, int g, int nsteps) -> int {
        const int gg = g < nsteps ? g : 0;
        return __builtin_amdgcn_readlane(tlv.y, gg / H);
    };
    auto acc_init = [&](const float* binit, float unit) {   // bias in accumulator units; C/D rows 4 kg + 0..3 of tile a
#pragma unroll
        for (int a = 0; a < QM; ++a) {
            const f32x4 bq = *(const f32x4*)(binit + wm * 32 + a * 16 + 4 * kg) * unit;
#pragma unroll
            for (int b = 0; b < QN; ++b) acc[a][b] = bq;
        }
    };
    auto mma = [&](const f32x4 (&a)[2][QM], const f32x4 (&b)[2][QN]) {
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < QM; ++i)
#pragma unroll
                for (int jj = 0; jj < QN; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[PA[t]][i]), __builtin_bit_cast(f16x8, b[PB[t]][jj]), acc[i][jj], 0, 0, 0);
    };
    auto ring_fill = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int nsteps) {
        ldA(rW, A0, g_off(tlv, 0, nsteps)); ldA(rW, A1, g_off(tlv, 1, nsteps));
    };
    auto kloop = [&](const __amdgpu_buffer_rsrc_t& rW, int2 tlv, int nsteps, const char* bbase) {
        auto step = [&](f32x4 (&Aa)[2][QM], f32x4 (&Bc)[2][QN], f32x4 (&Bn)[2][QN], int g) {
            const char* nbrow = bbase + g_row(tlv, g + 1, nsteps) * RSB;
            ldB(Bn, nbrow, (g + 1) % H);
            __builtin_amdgcn_sched_barrier(0);
            mma(Aa, Bc);
            __builtin_amdgcn_sched_barrier(0);
            ldA(rW, Aa, g_off(tlv, g + 2, nsteps));
        };
        ldB(B0, bbase + g_row(tlv, 0, nsteps) * RSB, 0);
        int g = 0;
        for (; g + 1 < nsteps; g += 2) { step(A0, B0, B1, g); step(A1, B1, B0, g + 1); }
        if (g < nsteps) step(A0, B0, B1, g);
    };
    auto wg_max = [&](float mx, int slot) -> float {        // workgroup maximum through LDS (one barrier)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[slot + wave] = mx;
        ev_lds_barrier();
        return fmaxf(fmaxf(red[slot], red[slot + 1]), fmaxf(red[slot + 2], red[slot + 3]));
    };

    // ---------------- phase 1: c1 over lrelu(x); X tile rows [g0 - h1, g0 + NT + h1), all channels, loaded once
    const __amdgpu_buffer_rsrc_t rW1 = ev_rsrc(pp.W1q), rW2 = ev_rsrc(p.Wq);
    const int ng1 = pp.ntaps1 * H, ng2 = p.ntaps * H;
    const int2 tlv1 = (lane < pp.ntaps1) ? pp.taplist1[lane] : make_int2(0, 0);
    const int2 tlv2 = (lane < p.ntaps) ? p.taplist[lane] : make_int2(0, 0);
    ring_fill(rW1, tlv1, ng1);
    float sx;
    {
        const int xrows = NT + 2 * pp.h1;
        f32x4 xg[XPASS];
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * RPS + srow;
            const int gr = g0 - pp.h1 + r;
            xg[q] = ev_bload4(rX, ((r < xrows && gr >= 0 && gr < p.nrows) ? (unsigned)gr * (unsigned)p.ldx : 0u) * 4u + (unsigned)sc4 * 4u, 0);   // (row 0 is a zero pad row)
        }
#pragma unroll
        for (int q = 0; q < XPASS; ++q) mx = fmaxf(mx, evh_absmax4(xg[q]));
        float tmx = wg_max(mx, 0);                      // (|lrelu(x)| <= |x|)
        if (!evh_is_finite(tmx)) {                      // an Inf in the tile (workgroup-uniform): the finite maximum sets the scale
            mx = 0.f;
#pragma unroll
            for (int q = 0; q < XPASS; ++q) mx = fmaxf(mx, evh_absmax4_finite(xg[q]));
            tmx = wg_max(mx, 8);
        }
        sx = evh_scale_for(tmx);
        am.rlo = am.rhi = tmx;                          // the residual IS this tile's input: its bound for the amax slots is the tile maximum just found (no slots of X needed)
#pragma unroll
        for (int q = 0; q < XPASS; ++q) {
            const int r = q * RPS + srow;
            f32x4 v = xg[q];
            v[0] = ev_lrelu(v[0], p.pro_slope); v[1] = ev_lrelu(v[1], p.pro_slope);
            v[2] = ev_lrelu(v[2], p.pro_slope); v[3] = ev_lrelu(v[3], p.pro_slope);
            uint2 q0v, q1v;
            evh_split4(v * sx, q0v, q1v);
            if (r < xrows) {
                char* dst = Xb + r * RSB + sc4 * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
    }
    const float u1 = pp.w1h_scale * sx;
    acc_init(pp.b1, u1);
    ev_lds_barrier();
    kloop(rW1, tlv1, ng1, Xb + (wn * (TN * 32) + f16i + pp.h1) * RSB + 16 * kg);
    ring_fill(rW2, tlv2, ng2);                           // c2's first fragments fly under the hand-over below

    // ---------------- y1 = lrelu(c1 + b1), zero outside the utterance; its maximum over the workgroup -> sy; split into LDS rows r + h2
    float sy;
    {
        const float inv1 = 1.0f / u1;
        float my = 0.f;
#pragma unroll
        for (int b = 0; b < QN; ++b) {
            const int r = wn * (TN * 32) + b * 16 + f16i;
            const int n = g0 + r;
            const int t = (n >= 0 && n < p.nrows) ? (n % p.S) - p.P : -1;
            const float inside = (t >= 0 && t < p.T) ? inv1 : 0.f;
#pragma unroll
            for (int a = 0; a < QM; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ev_lrelu(acc[a][b][e] * inside, pp.mid_slope);    // (back to true units; 0 outside the utterance)
                    acc[a][b][e] = v;
                    my = fmaxf(my, fabsf(v));
                }
        }
        float tmy = wg_max(my, 4);                      // (the barrier inside: every wave is done reading the X tile)
        if (!evh_is_finite(tmy)) {
            my = 0.f;
#pragma unroll
            for (int a = 0; a < QM; ++a)
#pragma unroll
                for (int b = 0; b < QN; ++b) my = fmaxf(my, evh_absmax4_finite(acc[a][b]));
            tmy = wg_max(my, 12);
        }
        sy = evh_scale_for(tmy);
#pragma unroll
        for (int b = 0; b < QN; ++b) {
            const int r = wn * (TN * 32) + b * 16 + f16i;
#pragma unroll
            for (int a = 0; a < QM; ++a) {
                uint2 q0v, q1v;
                evh_split4(acc[a][b] * sy, q0v, q1v);
                char* dst = Xb + (r + pp.h2) * RSB + (wm * 32 + a * 16 + 4 * kg) * 2;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
        // the 2 h2 border rows only feed outputs outside the stored window, but must be finite: zero both planes
        for (int i = tid; i < 2 * pp.h2 * (4 * C / 16); i += 256) {
            const int br = i / (4 * C / 16), c16 = i % (4 * C / 16);
            const int row = br < pp.h2 ? br : NT + br;
            uint4 z = {0u, 0u, 0u, 0u};
            *(uint4*)(Xb + row * RSB + c16 * 16) = z;
        }
    }
    const float u2 = p.wh_scale * sy;
    acc_init(p.bias, u2);
    ev_lds_barrier();

    // ---------------- phase 2: c2 over the LDS-resident y1 (tap offset t reads rows r + h2 + t)
    kloop(rW2, tlv2, ng2, Xb + (wn * (TN * 32) + f16i + pp.h2) * RSB + 16 * kg);
    {
        const float inv2 = 1.0f / u2;
#pragma unroll
        for (int a = 0; a < QM; ++a)
#pragma unroll
            for (int b = 0; b < QN; ++b) acc[a][b] *= inv2;
    }
    ev_amax_from_acc_q<QM, QN>(p, am, acc, g0 + wn * (TN * 32), lane);
    conv_epilogue_lean_q<TM, TN, LEAN>(p, acc, smem + wave * (32 * (TM * 32 + 4)), wm * 32, g0 + wn * (TN * 32), lane, n0, n0 + pp.out_rows);
    ev_amax_emit(p, am, g0 + wn * (TN * 32), TN * 32, lane);
}

// ---------------------------------------------------------------------------
// Wave / block reductions (64-wide wavefronts)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NTHR>
__device__ __forceinline__ float block_sum(float v, float* red /*[NTHR/64]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NTHR / 64; ++i) t += red[i];
    return t;
}

// ---------------------------------------------------------------------------
// GroupNorm(groups of CG channels over all T frames of one utterance, padded
// frames included — decoder.py:41-43) + Mish + mask, then one of
//   mode 0:  y = mish(gn(x)) * m
//   mode 1:  y = (mish(gn(x)) * m + temb[c]) * m        (ResnetBlock1D block1 + time MLP, decoder.py:56-57)
//   mode 2:  y = mish(gn(x)) * m + R[n][c]              (block2 + res_conv, decoder.py:58-60)
// One workgroup per (utterance, group); wavefront shuffles for the reductions.
// ---------------------------------------------------------------------------
struct GNParams {
    const float* X; int ldx; float* Y; int ldy;
    const float* gamma; const float* beta;
    const float* rowmask; const float* temb; const float* R; int ldr;
    int S, P, T, CG; int mode; float eps;
};

template <int NTHR, bool PRE = false>
__global__ __launch_bounds__(NTHR) void groupnorm_mish_kernel(const GNParams p) {
    __shared__ float red[NTHR / 64];
    const int b = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x;
    const int c4n = p.CG / 4;            // float4 columns per row
    const int rpp = NTHR / c4n;          // rows per pass
    const int r0 = tid / c4n, c4 = (tid % c4n) * 4;
    const size_t rowbase = (size_t)b * p.S + p.P;
    const int cbase = g * p.CG + c4;
    const float cnt = (float)p.T * (float)p.CG;
    // The (T x 32-channel) slab of one (utterance, group) is read from HBM/L2 ONCE and kept in registers when it fits
    // (T <= GN_REG_PASSES * rows-per-pass: 768 frames with 256 threads, 1024 with the 1024-thread build used for small batches); longer utterances fall
    // back to three passes over the (L2-resident) slab.
    constexpr int GN_REG_PASSES = NTHR == 256 ? 24 : (NTHR == 512 ? 12 : 8);
    const bool in_regs = p.T <= GN_REG_PASSES * rpp;
    f32x4 keep[GN_REG_PASSES];

    // The residual rows (mode 2) and the frame mask do not depend on the statistics: on the register-resident path they are requested
    // together with the slab, so that the apply phase after the two reductions is arithmetic and stores only (one memory round trip
    // less on every workgroup's critical chain).  An A/B build (EV_GN_PRE=1): measured slower at batch 64 — its 141 registers halve the
    // workgroups per CU — and not the default.
    // (PRE: only with >= 512 threads — the 256-thread build keeps 24 slab passes and has no registers left for it)
    f32x4 rkeep[PRE ? GN_REG_PASSES : 1];
    float mkeep[PRE ? GN_REG_PASSES : 1];
    float s = 0.f;
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < GN_REG_PASSES; ++q) {
            const int t = r0 + q * rpp;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (t < p.T) v = *(const f32x4*)(p.X + (rowbase + t) * p.ldx + cbase);
            keep[q] = v;
            if constexpr (PRE) {
                f32x4 r = {0.f, 0.f, 0.f, 0.f};
                float m = 0.f;
                if (t < p.T) {
                    m = p.rowmask[rowbase + t];
                    if (p.mode == 2) r = *(const f32x4*)(p.R + (rowbase + t) * p.ldr + cbase);
                }
                rkeep[q] = r; mkeep[q] = m;
            }
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int t = r0; t < p.T; t += rpp) {
            f32x4 v = *(const f32x4*)(p.X + (rowbase + t) * p.ldx + cbase);
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    const float mean = block_sum<NTHR>(s, red) / cnt;
    float q2 = 0.f;
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < GN_REG_PASSES; ++q) {
            if (r0 + q * rpp < p.T) {
                const f32x4 v = keep[q];
                float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                q2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
    } else {
        for (int t = r0; t < p.T; t += rpp) {
            f32x4 v = *(const f32x4*)(p.X + (rowbase + t) * p.ldx + cbase);
            float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            q2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float var = block_sum<NTHR>(q2, red) / cnt;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    float ga[4], be[4], te[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ga[e] = p.gamma[cbase + e]; be[e] = p.beta[cbase + e];
        te[e] = (p.mode == 1) ? p.temb[cbase + e] : 0.f;
    }
    auto apply = [&](int t, f32x4 v) {
        const size_t n = rowbase + t;
        const float m = p.rowmask[n];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = (v[e] - mean) * rstd * ga[e] + be[e];
            y = ev_mish(y) * m;
            if (p.mode == 1) y = (y + te[e]) * m;
            o[e] = y;
        }
        if (p.mode == 2) {
            f32x4 r = *(const f32x4*)(p.R + n * p.ldr + cbase);
            o[0] += r[0]; o[1] += r[1]; o[2] += r[2]; o[3] += r[3];
        }
        *(f32x4*)(p.Y + n * p.ldy + cbase) = o;
    };
    if (in_regs) {
#pragma unroll
        for (int q = 0; q < GN_REG_PASSES; ++q) {
            const int t = r0 + q * rpp;
            if constexpr (PRE) {
                if (t < p.T) {
                    const float m = mkeep[q];
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = (keep[q][e] - mean) * rstd * ga[e] + be[e];
                        y = ev_mish(y) * m;
                        if (p.mode == 1) y = (y + te[e]) * m;
                        o[e] = y;
                    }
                    if (p.mode == 2) { o[0] += rkeep[q][0]; o[1] += rkeep[q][1]; o[2] += rkeep[q][2]; o[3] += rkeep[q][3]; }
                    *(f32x4*)(p.Y + (rowbase + t) * p.ldy + cbase) = o;
                }
            } else {
                if (t < p.T) apply(t, keep[q]);
            }
        }
    } else {
        for (int t = r0; t < p.T; t += rpp) apply(t, *(const f32x4*)(p.X + (rowbase + t) * p.ldx + cbase));
    }
}

// groupnorm_apply_kernel: GroupNorm + Mish of ONE utterance whose per-tile statistics the producing conv_sk32_kernel left in
// `part` ([row tiles of 32][8 groups]{count, mean, M2}).  groupnorm_mish_kernel is one workgroup per (utterance, group): at batch 1
// eight CUs evaluate all T x 256 Mish (7.7 us, 130 launches per decode).  Here a workgroup is one (32-frame tile, group): every
// thread merges the tiles' statistics in ascending tile order (Chan et al.'s pairwise update — deterministic, and at least as
// accurate as the two-pass sums), then normalises its four channels of one frame.  Same epilogue modes as groupnorm_mish_kernel.
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const GNParams p, const float* part, int ntiles) {
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int t = blockIdx.x * 32 + (tid >> 3) - p.P, c4 = (tid & 7) * 4;
    const int cbase = g * 32 + c4;
    const bool live = t >= 0 && t < p.T;
    const size_t n = (size_t)p.P + (live ? t : 0);
    // every operand is requested before the first use: one memory round trip for the whole workgroup
    const f32x4 v = *(const f32x4*)(p.X + n * p.ldx + cbase);
    const f32x4 ga = *(const f32x4*)(p.gamma + cbase), be = *(const f32x4*)(p.beta + cbase);
    f32x4 te = {0.f, 0.f, 0.f, 0.f}, rr = te;
    if (p.mode == 1) te = *(const f32x4*)(p.temb + cbase);
    if (p.mode == 2) rr = *(const f32x4*)(p.R + n * p.ldr + cbase);
    const float m = p.rowmask[n];
    // tile j's {count, mean, M2} on lane j (64 tiles per round), merged in ascending order through v_readlane
    float na = 0.f, mean = 0.f, m2 = 0.f;
    for (int j0 = 0; j0 < ntiles; j0 += 64) {
        int pn = 0, pm = 0, pq = 0;
        if (j0 + lane < ntiles) {
            const int* pp = (const int*)(part + ((size_t)(j0 + lane) * 8 + g) * 4);
            pn = pp[0]; pm = pp[1]; pq = pp[2];
        }
        const int nj = ntiles - j0 < 64 ? ntiles - j0 : 64;
        for (int j = 0; j < nj; ++j) {
            const float nb = __int_as_float(__builtin_amdgcn_readlane(pn, j));
            const float mb = __int_as_float(__builtin_amdgcn_readlane(pm, j));
            const float qb = __int_as_float(__builtin_amdgcn_readlane(pq, j));
            if (nb > 0.f) {                      // (wave-uniform)
                const float nn = na + nb, d = mb - mean, w = nb / nn;
                mean += d * w;
                m2 += qb + d * d * na * w;
                na = nn;
            }
        }
    }
    if (!live) return;
    const float rstd = 1.0f / sqrtf(m2 / na + p.eps);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float y = (v[e] - mean) * rstd * ga[e] + be[e];
        y = ev_mish(y) * m;
        if (p.mode == 1) y = (y + te[e]) * m;
        o[e] = y;
    }
    if (p.mode == 2) { o[0] += rr[0]; o[1] += rr[1]; o[2] += rr[2]; o[3] += rr[3]; }
    *(f32x4*)(p.Y + n * p.ldy + cbase) = o;
}

// ---------------------------------------------------------------------------
// LayerNorm over C = 256 channels of each valid frame (transformer.py norm1/norm3,
// eps 1e-5, affine).  One wavefront per frame: 4 channels per lane, shuffle reduce.
// ---------------------------------------------------------------------------
struct LNParams {
    const float* X; int ldx; float* Y; int ldy; const float* gamma; const float* beta;
    int nrows, S, P, T; float eps;
};

// one 256-channel row held as 4 channels per lane of a wavefront (two-pass mean / biased variance, like torch's LayerNorm)
__device__ __forceinline__ f32x4 ev_ln256_row(f32x4 v, f32x4 g, f32x4 be, float eps) {
    float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
    float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
    float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 256.0f);
    float rstd = 1.0f / sqrtf(var + eps);
    f32x4 o = {d0 * rstd * g[0] + be[0], d1 * rstd * g[1] + be[1], d2 * rstd * g[2] + be[2], d3 * rstd * g[3] + be[3]};
    return o;
}

__global__ __launch_bounds__(256) void layernorm256_kernel(const LNParams p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.nrows) return;
    const int t = (n % p.S) - p.P;
    if (t < 0 || t >= p.T) return;
    f32x4 v = *(const f32x4*)(p.X + (size_t)n * p.ldx + lane * 4);
    f32x4 g = *(const f32x4*)(p.gamma + lane * 4);
    f32x4 be = *(const f32x4*)(p.beta + lane * 4);
    *(f32x4*)(p.Y + (size_t)n * p.ldy + lane * 4) = ev_ln256_row(v, g, be, p.eps);
}

// ---------------------------------------------------------------------------
// ln_mlp_kernel: the feed-forward half (MODE 0) or the QKV projection (MODE 1) of a BasicTransformerBlock
// (transformer.py:243-316) in ONE launch per 32-frame row tile:
//   MODE 0:  y = x + W2 . SnakeBeta(W1 . LN(x) + b1) + b2,  * mask        (norm3 -> ff -> residual, transformer.py:300-316)
//   MODE 1:  y = Wqkv . LN(x)                                             (norm1 -> to_q | to_k | to_v, :262-271)
// The unfused path runs LayerNorm, Linear 256 -> 1024 (+SnakeBeta) and Linear 1024 -> 256 as three launches: two stage /
// epilogue heads and tails per launch on a grid of about one round of workgroups, one HBM round trip of the normalised rows
// and one of the 1024-wide hidden activations (2 x 136 MB per block at batch 64).  Here a workgroup stages its 32 x 256 input
// rows ONCE (one latency episode), normalises them in LDS (the layernorm256_kernel arithmetic), and then runs over the hidden
// width in chunks of 128: phase 1 computes the chunk h = SnakeBeta(W1_c . LN(x) + b1_c) into LDS (the [frame][channel] operand
// layout), phase 2 accumulates y += W2[:, c] . h into registers that live across all chunks.  Neither intermediate touches
// HBM.  Accumulation order over k is the unfused kernels' (bias first, k ascending), so results are bit-identical to them.
// Weights stream from L2 in MFMA-fragment order as in conv_gemm_kernel, four k-groups ahead.
// ---------------------------------------------------------------------------
struct MlpParams {
    ConvParams ep;                 // epilogue view of the LAST linear: Y / ldy / Cout / bias, R / ldr, rowmask / mask2, nrows / S / P / T
    const float* X; int ldx;       // (rows, 256) input of the LayerNorm
    const float* ln_g; const float* ln_b; float ln_eps;
    const float* W1; const float* b1; int M1;        // fragment order [M1/32][256/8][64][4]; M1 a multiple of 128
    const float* alpha; const float* binv;           // SnakeBeta vectors over the hidden width (MODE 0)
    const float* W2;                                 // fragment order [256/32][M1/8][64][4] (MODE 0)
    const void* W1h; const void* W2h; float w1_scale, w2_scale;   // ln_mlp_h16_kernel: both linears times their power-of-two scale as two fp16 pieces
    const void* W1x; const void* W2x;                // ln_mlp_split_kernel: both linears as three bf16 pieces (conv_split_kernel's fragment order)
    int ntiles;                    // 32-row tiles of the launch; the grid is either ntiles workgroups (one tile each: sk.q = M1 / 128,
    SkCtl sk;                      // sk.r = 0, no hand-offs) or the balanced persistent grid described at SkCtl
    int qkv_pack; float qkv_scale[3];   // ln_qkv_h16_kernel: write q / k / v as fp16 piece pairs (see attn_out_h16_kernel), times these powers of two
};

// unit = (32-row tile, 128-wide chunk of the hidden [MODE 1: output] width); a workgroup walks its unit range tile segment by tile
// segment: stage + LayerNorm the tile's rows once per segment, then the chunks [c0, c1) of it.
// WPC = workgroups per CU the build is meant for: 3 for the one-tile-per-workgroup grid (LDS admits three: the register allocation
// stays at three waves per SIMD), 2 for the balanced persistent grid (measured with tools/shape_profile.py, EV_SK_WGS: two
// workgroups of ~8 units each beat three of ~5.4 — fewer split tiles, fewer staging episodes — and 256 registers leave room to
// fetch a contributor's partial tile under the last chunk's MFMAs).
template <int MODE, int WPC>
__global__ __launch_bounds__(256, WPC) void ln_mlp_kernel(const MlpParams mp) {
    constexpr int NT = 32, C = 256, XLD = C + 4, HC = 128, HLD = HC + 4;
    const ConvParams& p = mp.ep;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                  // [NT][XLD] normalised input rows (B operand of phase 1)
    float* Hs = smem + NT * XLD;                       // [NT][HLD] hidden chunk (B operand of phase 2); MODE 1: epilogue slabs
    int* skw = (int*)(smem + NT * XLD + 4 * 32 * 36);   // one word for sk_wait (plain LDS accesses: the asm barriers around it order them)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(mp.X), rW1 = ev_rsrc(mp.W1), rW2 = ev_rsrc(MODE == 0 ? mp.W2 : mp.W1);
    const unsigned wlane = (unsigned)lane * 16u;
    const int nchunk = mp.M1 / HC;
    const int KG2 = mp.M1 / 8;                          // k-groups per row tile of W2
    const int g = blockIdx.x;
    const unsigned tag = MODE == 0 ? sk_tag(mp.sk) : 0u;
    int u = sk_start(mp.sk, g);
    const int ue = sk_start(mp.sk, g + 1);
    // ONE set of eight fragment registers serves both phases: phase 1 (one 32-row tile of W1, 4 MFMAs per fragment) runs eight
    // k-groups ahead, phase 2 (two row tiles of W2, 8 MFMAs per fragment pair) four — 32 MFMAs of cover either way.  In the last
    // eight (four) k-groups of a phase every register, once consumed, is refilled with the first fragments of the NEXT phase, so
    // the hand-over between the phases (SnakeBeta, LDS write, barriers) never waits for L2.  All loads are unconditional.
    f32x4 F0, F1, F2, F3, F4, F5, F6, F7;
    auto ldP = [&](int ht, int kg) { return ev_bload4(rW1, wlane, (unsigned)(ht * (C / 8) + kg) * 1024u); };
    auto ldQ = [&](int hc, int kg, int a) { return ev_bload4(rW2, wlane, (unsigned)((wave * 2 + a) * KG2 + hc * (HC / 8) + kg) * 1024u); };
    // Per-channel vectors of the hidden width (bias, SnakeBeta alpha / 1/beta) are requested one phase before their use, as
    // part of the same counted load stream: fetched at the point of use they put one exposed L2 round trip (and a vmcnt(0) that
    // drained the fragment ring) into every chunk — twice.  mp.b1 always points at readable memory (zeros when the layer has no bias).
    const __amdgpu_buffer_rsrc_t rB1 = ev_rsrc(mp.b1), rSa = ev_rsrc(MODE == 0 ? mp.alpha : mp.b1), rSb = ev_rsrc(MODE == 0 ? mp.binv : mp.b1);
    const unsigned coff = (unsigned)(4 * lh) * 4u;     // C/D register 4g+e of a 32-channel tile is channel 8g + 4*half + e
    const float* xrow = Xs + li * XLD + 4 * lh;
    const float* hrow = Hs + li * HLD + 4 * lh;
    // this workgroup's slot of the hand-off area: [published partial | private spill]; element (wave, a, q) of a tile = 64 lanes x 16 B
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(mp.sk.part);
    const unsigned pslot = (unsigned)mp.sk.part_floats * 8u;            // bytes per workgroup (two tiles)
    const unsigned pelem = (unsigned)(wave * 8) * 1024u + wlane;
    bool pend_pub = false;                             // partial stored, flag not raised yet (raised behind the next staging's loads)

    while (u < ue) {
        const int t = u / nchunk, c0 = u - t * nchunk;
        const int c1 = (ue - u < nchunk - c0) ? c0 + (ue - u) : nchunk;
        u += c1 - c0;
        const int n0 = t * NT;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= NT || n0 + dist >= p.nrows) continue;
        }
        F0 = ldP(c0 * 4 + wave, 0); F1 = ldP(c0 * 4 + wave, 1); F2 = ldP(c0 * 4 + wave, 2); F3 = ldP(c0 * 4 + wave, 3);
        F4 = ldP(c0 * 4 + wave, 4); F5 = ldP(c0 * 4 + wave, 5); F6 = ldP(c0 * 4 + wave, 6); F7 = ldP(c0 * 4 + wave, 7);
        ev_lds_barrier();                              // the previous segment's epilogue is done with its LDS slabs
        {   // ---- stage the 32 x 256 input rows: all loads first (one latency episode)
            f32x4 xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = q * 256 + tid, r = i >> 6, c4 = (i & 63) * 4, gr = n0 + r;
                xv[q] = ev_bload4(rX, ((unsigned)(gr < p.nrows ? gr : 0) * mp.ldx + c4) * 4u, 0);   // (beyond the tensor: pad row 0)
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = q * 256 + tid, r = i >> 6, c4 = (i & 63) * 4;
                *(f32x4*)(Xs + r * XLD + c4) = xv[q];
            }
        }
        if (pend_pub) {   // every wave has consumed staging loads issued AFTER its partial stores (vmcnt retires in order): the partial
            sk_publish(mp.sk, g, tag, tid);   // tile is written, raise the flag — the drain in sk_publish finds nothing left to wait for
            pend_pub = false;
        } else ev_lds_barrier();
        {   // ---- LayerNorm in place: wave w owns rows 8w .. 8w+7, one row = 4 channels per lane
            const f32x4 gm = *(const f32x4*)(mp.ln_g + lane * 4), be = *(const f32x4*)(mp.ln_b + lane * 4);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float* row = Xs + (wave * 8 + r) * XLD + lane * 4;
                *(f32x4*)row = ev_ln256_row(*(const f32x4*)row, gm, be, mp.ln_eps);
            }
        }
        ev_lds_barrier();

        // ---- passes over chunk ranges of this tile: [c0, c1) first; an owner whose wait for a contributor ran out adds one pass
        // over that contributor's range (fresh accumulators: the contributor's bits)
        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;                                // next contributor of this tile (owner only)
        const int tile_end = (t + 1) * nchunk;
        for (;;) {
            f32x16 acc2[2][1];                         // MODE 0: this wave's 64 output channels x 32 frames, alive across all chunks of the pass
            if constexpr (MODE == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias && cA == 0) bq = *(const f32x4*)(p.bias + wave * 64 + a * 32 + 8 * q + 4 * lh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[a][0][4 * q + e] = bq[e];
                    }
                }
            }
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)((cA * 4 + wave) * 32 + 8 * q) * 4u, 0);

            // Owner of a split tile, balanced build: the first contributor's partial tile is fetched UNDER the MFMAs of this pass's last
            // chunk — flag word requested at the start of that chunk, published to the workgroup through the chunk's own two barriers,
            // acquire by one lane between them, then eight sc1 loads per lane that land during phase 2.  (These loads sit under a
            // wave-uniform branch, so hipcc drains vmcnt at the next use — once per split tile, in its last chunk only; issued
            // unconditionally in every chunk they cost every chunk an L2 miss in front of its in-order fragment stream.)
            const bool pf_ok = MODE == 0 && WPC == 2 && mp.sk.ctrl != nullptr && c0 == 0 && c1 < nchunk && cA == c0 && g + 1 < (int)gridDim.x;
            bool pf_hit = false;
            f32x4 PF[8];
            for (int hc = cA; hc < cB; ++hc) {
                const int ht = hc * 4 + wave;                  // this wave's 32 hidden (MODE 1: output) channels of the chunk
                const bool pf_try = pf_ok && hc == cB - 1;
                unsigned pf_flag = 0;
                if constexpr (MODE == 0 && WPC == 2) {
                    const __amdgpu_buffer_rsrc_t rFl = ev_rsrc(mp.sk.flags ? (const void*)mp.sk.flags : (const void*)mp.b1);
                    if (pf_try) pf_flag = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rFl, 0, (g + 1) * 4, 16);   // (scalar offset: one request per wave)
                }
                const int htn = hc + 1 < cB ? ht + 4 : cA * 4 + wave;   // next chunk's tile (after the last chunk: a harmless re-read)
                // ================= phase 1: acc1 = W1[ht] . LN(x) + b1, K = 256 = 32 k-groups =================
                f32x16 acc1;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc1[4 * q + e] = bq[q][e];
                f32x4 sa[4], sb[4];
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sa[q] = ev_bload4(rSa, coff + (unsigned)(ht * 32 + 8 * q) * 4u, 0);
                        sb[q] = ev_bload4(rSb, coff + (unsigned)(ht * 32 + 8 * q) * 4u, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)(htn * 32 + 8 * q) * 4u, 0);   // next chunk's bias
                }
                {
                    f32x4 B0 = *(const f32x4*)(xrow), B1;
                    auto mma1 = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], b[s4], acc1, 0, 0, 0);
                    };
                    // one k-group: B fragment of the next k-group, 4 MFMAs, refill of the consumed register
#define EV_P1_STEP(F, BC, BN_, I, REFILL)                                                        \
                    BN_ = *(const f32x4*)(xrow + ((kb + (I) + 1) & 31) * 8);                      \
                    __builtin_amdgcn_sched_barrier(0);                                           \
                    mma1(F, BC);                                                                 \
                    __builtin_amdgcn_sched_barrier(0);                                           \
                    F = REFILL;
#pragma unroll 1
                    for (int kb = 0; kb < 24; kb += 8) {
                        EV_P1_STEP(F0, B0, B1, 0, ldP(ht, kb + 8))  EV_P1_STEP(F1, B1, B0, 1, ldP(ht, kb + 9))
                        EV_P1_STEP(F2, B0, B1, 2, ldP(ht, kb + 10)) EV_P1_STEP(F3, B1, B0, 3, ldP(ht, kb + 11))
                        EV_P1_STEP(F4, B0, B1, 4, ldP(ht, kb + 12)) EV_P1_STEP(F5, B1, B0, 5, ldP(ht, kb + 13))
                        EV_P1_STEP(F6, B0, B1, 6, ldP(ht, kb + 14)) EV_P1_STEP(F7, B1, B0, 7, ldP(ht, kb + 15))
                    }
                    {   // last eight k-groups: the registers go over to the next phase
                        constexpr int kb = 24;
                        if constexpr (MODE == 0) {
                            EV_P1_STEP(F0, B0, B1, 0, ldQ(hc, 0, 0)) EV_P1_STEP(F1, B1, B0, 1, ldQ(hc, 0, 1))
                            EV_P1_STEP(F2, B0, B1, 2, ldQ(hc, 1, 0)) EV_P1_STEP(F3, B1, B0, 3, ldQ(hc, 1, 1))
                            EV_P1_STEP(F4, B0, B1, 4, ldQ(hc, 2, 0)) EV_P1_STEP(F5, B1, B0, 5, ldQ(hc, 2, 1))
                            EV_P1_STEP(F6, B0, B1, 6, ldQ(hc, 3, 0)) EV_P1_STEP(F7, B1, B0, 7, ldQ(hc, 3, 1))
                        } else {
                            EV_P1_STEP(F0, B0, B1, 0, ldP(htn, 0)) EV_P1_STEP(F1, B1, B0, 1, ldP(htn, 1))
                            EV_P1_STEP(F2, B0, B1, 2, ldP(htn, 2)) EV_P1_STEP(F3, B1, B0, 3, ldP(htn, 3))
                            EV_P1_STEP(F4, B0, B1, 4, ldP(htn, 4)) EV_P1_STEP(F5, B1, B0, 5, ldP(htn, 5))
                            EV_P1_STEP(F6, B0, B1, 6, ldP(htn, 6)) EV_P1_STEP(F7, B1, B0, 7, ldP(htn, 7))
                        }
                    }
#undef EV_P1_STEP
                }
                if constexpr (MODE == 1) {
                    // ---- projection only: store this wave's 32 channels x 32 frames (per-wave slab in the Hs region)
                    f32x16 accs[1][1];
                    accs[0][0] = acc1;
                    conv_epilogue_lean<1, 1, 1>(p, accs, Hs + wave * (32 * 36), ht * 32, n0, lane);
                    continue;
                } else {
                    // ---- h = SnakeBeta(acc1) into Hs[frame][hidden channel]; C/D register 4g+e is channel 8g + 4*half + e
                    f32x4 hv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = acc1[4 * q + e]; hv[q][e] = fmaf(sb[q][e], ev_sin2(v * sa[q][e]), v); }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)(htn * 32 + 8 * q) * 4u, 0);   // next chunk's bias, under phase 2
                    if constexpr (WPC == 2) { if (pf_try && tid == 0) *skw = (pf_flag == tag) ? 1 : 0; }
                    ev_lds_barrier();        // every wave is done reading the previous chunk's Hs (its phase 2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(f32x4*)(Hs + li * HLD + wave * 32 + 8 * q + 4 * lh) = hv[q];
                    if constexpr (WPC == 2) {
                        if (pf_try) {
                            pf_hit = __builtin_amdgcn_readfirstlane(*skw) != 0;
                            // (wave 0, a SCALAR branch: under `tid == 0` — a divergent branch inside the chunk loop — hipcc's uniformity analysis gave up on
                            // the loop's offsets and wrapped every fragment load of both K loops in a waterfall loop)
                            if (pf_hit && wave == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                        }
                    }
                    ev_lds_barrier();
                    if constexpr (WPC == 2) {
                        if (pf_hit) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) PF[k] = ev_bload4_sc1(rPart, (unsigned)(g + 1) * pslot + pelem + (unsigned)k * 1024u);
                        }
                    }
                    // ================= phase 2: acc2 += W2[64 channels of this wave][chunk] . h, K = 128 = 16 k-groups =================
                    f32x4 B0 = *(const f32x4*)(hrow), B1;
                    auto mma2 = [&](const f32x4& a0, const f32x4& a1, const f32x4& b) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], b[s4], acc2[0][0], 0, 0, 0);
                            acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], b[s4], acc2[1][0], 0, 0, 0);
                        }
                    };
#define EV_P2_STEP(FA, FB, BC, BN_, I, RA, RB)                                                   \
                    BN_ = *(const f32x4*)(hrow + ((kb + (I) + 1) & 15) * 8);                      \
                    __builtin_amdgcn_sched_barrier(0);                                           \
                    mma2(FA, FB, BC);                                                            \
                    __builtin_amdgcn_sched_barrier(0);                                           \
                    FA = RA; FB = RB;
#pragma unroll 1
                    for (int kb = 0; kb < 12; kb += 4) {
                        EV_P2_STEP(F0, F1, B0, B1, 0, ldQ(hc, kb + 4, 0), ldQ(hc, kb + 4, 1)) EV_P2_STEP(F2, F3, B1, B0, 1, ldQ(hc, kb + 5, 0), ldQ(hc, kb + 5, 1))
                        EV_P2_STEP(F4, F5, B0, B1, 2, ldQ(hc, kb + 6, 0), ldQ(hc, kb + 6, 1)) EV_P2_STEP(F6, F7, B1, B0, 3, ldQ(hc, kb + 7, 0), ldQ(hc, kb + 7, 1))
                    }
                    {   // last four k-groups: the registers go back to phase 1 of the next chunk
                        constexpr int kb = 12;
                        EV_P2_STEP(F0, F1, B0, B1, 0, ldP(htn, 0), ldP(htn, 1)) EV_P2_STEP(F2, F3, B1, B0, 1, ldP(htn, 2), ldP(htn, 3))
                        EV_P2_STEP(F4, F5, B0, B1, 2, ldP(htn, 4), ldP(htn, 5)) EV_P2_STEP(F6, F7, B1, B0, 3, ldP(htn, 6), ldP(htn, 7))
                    }
#undef EV_P2_STEP
                }
            }
            if constexpr (MODE == 1) break;
            else {
                if (spilled) {                         // fallback pass done: running sum (spilled) + this contributor's share
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = ev_bload4_sc1(rPart, (unsigned)g * pslot + pslot / 2 + pelem + (unsigned)(a * 4 + q) * 1024u);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc2[a][0][4 * q + e] += v[e];
                        }
                    spilled = false;
                }
                if (c0 != 0) {                         // not the owner: publish the partial tile (no bias) and move on
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = {acc2[a][0][4 * q], acc2[a][0][4 * q + 1], acc2[a][0][4 * q + 2], acc2[a][0][4 * q + 3]};
                            ev_bstore4_sc1(rPart, (unsigned)g * pslot + pelem + (unsigned)(a * 4 + q) * 1024u, v);
                        }
                    pend_pub = true;
                    break;
                }
                // owner: add the contributors' partials in ascending workgroup order
                bool again = false;
                if constexpr (WPC == 2) {
                    if (pf_hit && gi == g + 1) {           // the first contributor's tile is already in registers
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int e2 = 0; e2 < 4; ++e2) acc2[a][0][4 * q + e2] += PF[a * 4 + q][e2];
                        ++gi;
                    }
                }
                while (c1 < nchunk) {
                    // the workgroups gi .. gi + n - 1 hold the rest of this tile (every workgroup holds at least one unit)
                    int n = 0;
                    while (n < 64 && gi + n < (int)gridDim.x && sk_start(mp.sk, gi + n) < tile_end) ++n;
                    if (n == 0) break;
                    const int ready = sk_wait_many(mp.sk, gi, n, tag, tid, skw);
                    for (int k = 0; k < ready; ++k) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = ev_bload4_sc1(rPart, (unsigned)(gi + k) * pslot + pelem + (unsigned)(a * 4 + q) * 1024u);
#pragma unroll
                                for (int e2 = 0; e2 < 4; ++e2) acc2[a][0][4 * q + e2] += v[e2];
                            }
                    }
                    gi += ready;
                    if (ready < n) {                       // gi is not there in time: spill the running sum, compute its share here
                        const int s = sk_start(mp.sk, gi);
                        int e = sk_start(mp.sk, gi + 1);
                        e = e < tile_end ? e : tile_end;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = {acc2[a][0][4 * q], acc2[a][0][4 * q + 1], acc2[a][0][4 * q + 2], acc2[a][0][4 * q + 3]};
                                ev_bstore4_sc1(rPart, (unsigned)g * pslot + pslot / 2 + pelem + (unsigned)(a * 4 + q) * 1024u, v);
                            }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        cA = s - t * nchunk; cB = e - t * nchunk;
                        spilled = true; again = true;
                        ++gi;
                        break;
                    }
                }
                if (again) {
                    F0 = ldP(cA * 4 + wave, 0); F1 = ldP(cA * 4 + wave, 1); F2 = ldP(cA * 4 + wave, 2); F3 = ldP(cA * 4 + wave, 3);
                    F4 = ldP(cA * 4 + wave, 4); F5 = ldP(cA * 4 + wave, 5); F6 = ldP(cA * 4 + wave, 6); F7 = ldP(cA * 4 + wave, 7);
                    continue;
                }
                // ---- epilogue: + residual, * mask, store (the lean epilogue's first barrier also retires the last phase 2's LDS reads)
                conv_epilogue_lean<2, 1, 1>(p, acc2, smem + wave * (32 * 68), wave * 64, n0, lane);
                break;
            }
        }
    }
    if constexpr (MODE == 0) {
        if (pend_pub) sk_publish(mp.sk, g, tag, tid);   // (the partial was this workgroup's last segment)
        sk_arrive(mp.sk, tag, tid);
    }
}

// ---------------------------------------------------------------------------
// ln_mlp_split_kernel: the feed-forward of ln_mlp_kernel<0> (LayerNorm -> 256 x M1 linear + bias -> SnakeBeta -> M1 x 256 linear + bias,
// + residual, * mask) on the bf16 matrix pipe (conv_split_kernel: six exact bf16 products per element pair, fp32 accumulation).
// The bf16 pipe is sixteen times faster than the fp32 one, so the weight stream per row tile has to be paid by more rows: a
// workgroup takes 64 rows (the 32-row tiles of the fp32 build would pull 3 GB of fragments per launch through L2), ONE workgroup
// per CU holds the normalised rows of its tile (three bf16 planes x 256 channels = 97 KB) and one 128-wide hidden chunk (49 KB).
//   phase 1: wave w computes hidden channels 32 w .. 32 w + 31 of the chunk for all 64 rows (K = 256 = 16 slabs, 12 MFMAs each)
//   hand-over: SnakeBeta on the accumulators, split into three pieces, 8-byte stores into the chunk's LDS planes
//   phase 2: wave w accumulates output channels 64 w .. 64 w + 63 for all 64 rows (K = 128 = 8 slabs, 24 MFMAs each)
// One ring of eight fragment sets serves both phases (eight slabs ahead in phase 1, four in phase 2; the tail of each phase
// refills it with the head of the next).  Units, hand-offs and the recompute fallback are those of SkCtl (unit = (64-row tile,
// 128-wide hidden chunk); a partial tile is 64 x 256 floats = one hand-off slot).
// ---------------------------------------------------------------------------
template <int TERMS = 6>
__global__ __launch_bounds__(256, 1) void ln_mlp_split_kernel(const MlpParams mp) {
    constexpr int NT = 64, C = 256, HC = 128;
    constexpr int XRS = 6 * C + 16, HRS = 6 * HC + 16;   // LDS row strides in bytes (97 x 16, 49 x 16)
    const ConvParams& p = mp.ep;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // [NT][XRS]: LN(x) as three bf16 planes
    char* Hb = Xb + NT * XRS;                           // [NT][HRS]: SnakeBeta(hidden chunk) as three bf16 planes
    int* skw = (int*)(Hb + NT * HRS);
    float* Sv = (float*)(Hb + NT * HRS + 16);          // SnakeBeta vectors over the hidden width: [alpha (M1) | 1 / beta (M1)], M1 <= 1024
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(mp.X), rW1 = ev_rsrc(mp.W1x), rW2 = ev_rsrc(mp.W2x);
    const unsigned wlane = (unsigned)lane * 16u;
    const int nchunk = mp.M1 / HC;
    const int KS2 = mp.M1 / 16;                         // slabs per 32-row tile of W2
    const int g = blockIdx.x;
    const unsigned tag = sk_tag(mp.sk);
    int u = sk_start(mp.sk, g);
    const int ue = sk_start(mp.sk, g + 1);
    f32x4 R0[3], R1[3], R2[3], R3[3], R4[3], R5[3], R6[3], R7[3];
    auto ldP = [&](f32x4 (&dst)[3], int ht, int sl) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) dst[pc] = ev_bload4(rW1, wlane, (unsigned)(((ht * (C / 16) + sl) * 3 + pc) * 1024));
    };
    auto ldQ = [&](f32x4 (&dst)[3], int hc, int sl, int a) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) dst[pc] = ev_bload4(rW2, wlane, (unsigned)((((wave * 2 + a) * KS2 + hc * (HC / 16) + sl) * 3 + pc) * 1024));
    };
    const __amdgpu_buffer_rsrc_t rB1 = ev_rsrc(mp.b1);
    // ring set i of the phase that comes next, whichever it is (no load under a branch): phase 1 of hidden tile hx -> slab i of W1's
    // row tile hx; phase 2 of chunk hx -> slab i / 2 of W2's row tile 2 wave + (i & 1)
    auto ldN = [&](f32x4 (&dst)[3], bool q2, int hx, int i) {
        const __amdgpu_buffer_rsrc_t rs = q2 ? rW2 : rW1;
        const unsigned off = q2 ? (unsigned)((((wave * 2 + (i & 1)) * KS2 + hx * (HC / 16) + (i >> 1)) * 3) * 1024) : (unsigned)(((hx * (C / 16) + i) * 3) * 1024);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) dst[pc] = ev_bload4(rs, wlane, off + (unsigned)(pc * 1024));
    };
    {   // (all loads first: one latency episode; published by the first staging barrier)
        float va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = q * 256 + tid; va[q] = i < mp.M1 ? mp.alpha[i] : 0.f; vb[q] = i < mp.M1 ? mp.binv[i] : 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = q * 256 + tid; if (i < mp.M1) { Sv[i] = va[q]; Sv[mp.M1 + i] = vb[q]; } }
    }
    const unsigned coff = (unsigned)(4 * lh) * 4u;
    const char* xrow = Xb + li * XRS + 16 * lh;
    const char* hrow = Hb + li * HRS + 16 * lh;
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(mp.sk.part);
    const unsigned pslot = (unsigned)mp.sk.part_floats * 8u;
    const unsigned pelem = (unsigned)(wave * 16) * 1024u + wlane;      // element (wave, a, j, q) of a partial tile = 64 lanes x 16 B
    bool pend_pub = false;
    int nst = 0;                                       // diagnostic (EV_MLP_STAMPS): up to 32 s_memrealtime stamps per workgroup
    auto stamp = [&]() { if (p.stamps && tid == 0 && nst < 32) p.stamps[32 * g + nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
    stamp();
    constexpr int PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};

    while (u < ue) {
        const int t = u / nchunk, c0 = u - t * nchunk;
        const int c1 = (ue - u < nchunk - c0) ? c0 + (ue - u) : nchunk;
        u += c1 - c0;
        const int n0 = t * NT;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= NT || n0 + dist >= p.nrows) continue;
        }
        {
            const int ht0 = c0 * 4 + wave;
            ldP(R0, ht0, 0); ldP(R1, ht0, 1); ldP(R2, ht0, 2); ldP(R3, ht0, 3); ldP(R4, ht0, 4); ldP(R5, ht0, 5); ldP(R6, ht0, 6); ldP(R7, ht0, 7);
        }
        ev_lds_barrier();                              // the previous segment's epilogue is done with its LDS slabs
        {   // ---- stage + LayerNorm + split: wave w owns rows 16 w .. 16 w + 15, a row = 4 channels per lane
            const f32x4 gm = *(const f32x4*)(mp.ln_g + lane * 4), be = *(const f32x4*)(mp.ln_b + lane * 4);
            f32x4 xv[16];                               // all sixteen rows requested at once: one latency episode
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = n0 + wave * 16 + r;
                xv[r] = ev_bload4(rX, ((unsigned)(gr < p.nrows ? gr : 0) * mp.ldx + lane * 4) * 4u, 0);   // (beyond the tensor: pad row 0)
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 o = ev_ln256_row(xv[r], gm, be, mp.ln_eps);
                uint2 q0v, q1v, q2v;
                evx_split4(o, q0v, q1v, q2v);
                char* dst = Xb + (wave * 16 + r) * XRS + lane * 8;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v; *(uint2*)(dst + 4 * C) = q2v;
            }
        }
        if (pend_pub) { sk_publish(mp.sk, g, tag, tid); pend_pub = false; }   // (drain + barrier + flag: the barrier also publishes the staged rows)
        else ev_lds_barrier();
        stamp();

        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;
        const int tile_end = (t + 1) * nchunk;
        for (;;) {
            f32x16 acc2[2][2];                         // this wave's 64 output channels x 64 rows, alive across all chunks of the pass
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && cA == 0) bq = *(const f32x4*)(p.bias + wave * 64 + a * 32 + 8 * q + 4 * lh);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[a][j][4 * q + e] = bq[e];
                }
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)((cA * 4 + wave) * 32 + 8 * q) * 4u, 0);
            // Software pipeline over the chunks: phase 1 of chunk hc, then phase 2 of chunk hc - 1 with the SnakeBeta + split arithmetic of
            // chunk hc in the shadow of its MFMAs (one wave per SIMD: nothing else would fill the matrix pipe during ~800 vector
            // instructions), then the pieces go to LDS between two barriers.  One more phase 2 after the loop.
            uint2 hq[8][3];                            // chunk hc's hidden values as bf16 pieces, (j, q) -> 4 channels x 3 pieces
            f32x4 B0[3][2], B1[3][2];
            auto ldBh = [&](f32x4 (&dst)[3][2], int sl) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int j = 0; j < 2; ++j) dst[pc][j] = *(const f32x4*)(hrow + j * 32 * HRS + pc * (2 * HC) + sl * 32);
            };
            auto mma2 = [&](const f32x4 (&a0)[3], const f32x4 (&a1)[3], const f32x4 (&b)[3][2]) {
#pragma unroll
                for (int tt = 9 - TERMS; tt < 9; ++tt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[PA[tt]]), __builtin_bit_cast(bf16x8, b[PB[tt]][j]), acc2[0][j], 0, 0, 0);
                        acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[PA[tt]]), __builtin_bit_cast(bf16x8, b[PB[tt]][j]), acc2[1][j], 0, 0, 0);
                    }
            };
            for (int hc = cA; hc <= cB; ++hc) {
                const bool p1 = hc < cB, p2 = hc > cA;             // phase 1 of chunk hc / phase 2 of chunk hc - 1 in this iteration
                const int ht = (p1 ? hc : cA) * 4 + wave;          // (past the last chunk: harmless re-reads)
                const int htn = hc + 1 < cB ? ht + 4 : cA * 4 + wave;
                // (two partial accumulators per row tile = four accumulation chains for a wave that is alone on its SIMD; tools/mfma_chain_probe.hip
                // prices two against four chains at 344 vs 315 ns per slab of this loop)
                f32x16 acc1[2], acc1b[2];
                if (p1) {
                    // ================= phase 1: acc1 = W1[ht] . LN(x) + b1, K = 256 = 16 slabs; the ring holds slabs 0..7 of ht =================
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { acc1[j][4 * q + e] = bq[q][e]; acc1b[j][4 * q + e] = 0.f; }
                    auto ldB = [&](f32x4 (&dst)[3][2], int sl) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                            for (int j = 0; j < 2; ++j) dst[pc][j] = *(const f32x4*)(xrow + j * 32 * XRS + pc * (2 * C) + sl * 32);
                    };
                    auto mma1 = [&](const f32x4 (&a)[3], const f32x4 (&b)[3][2]) {
#pragma unroll
                        for (int tt = 9 - TERMS; tt < 9; ++tt)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                if (tt & 1) acc1b[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[tt]]), __builtin_bit_cast(bf16x8, b[PB[tt]][j]), acc1b[j], 0, 0, 0);
                                else acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[tt]]), __builtin_bit_cast(bf16x8, b[PB[tt]][j]), acc1[j], 0, 0, 0);
                            }
                    };
                    const int hq2 = p2 ? hc - 1 : hc;              // the phase 2 that follows: the previous chunk's (first iteration: none — re-reads)
                    // One wave per SIMD: nothing hides a block of loads between two blocks of MFMAs.  The loads of a step — the B fragments of
                    // the next slab and the refill of the ring set consumed one step EARLIER (its registers are free) — are interleaved with
                    // the step's MFMAs by an explicit schedule (sched_group_barrier: 2 MFMAs, 1 LDS read, [1 fragment load]).  The set
                    // consumed by a phase's last step is refilled by the first step of whatever phase comes next (sets 6 and 7: a re-load of
                    // what is already there is harmless).  Per-phase stamps (EV_MLP_STAMPS, profiles/r03_ln_mlp_split_stamps.txt): 3.9 us per
                    // phase 1 = 245 ns per slab, where twelve MFMAs alone take 219 ns at the clock the chip holds under bf16 MFMA load
                    // (tools/mfma_chain_probe.hip: 18.3 ns each, i.e. 32 cycles at ~1.75 GHz) and the probe's loop with the same loads 315 ns.
                    const bool nq1 = p2 || hc + 1 >= cB;                       // what follows this phase 1: a phase 2 (of chunk nx1) or the next chunk's phase 1
                    const int nx1 = p2 ? hq2 : (hc + 1 < cB ? ht + 4 : hc);
                    ldB(B0, 0);
#define EVX_P1(RR, BC, BN_, SL, REFILL, NV)                                                       \
                    ldB(BN_, ((SL) + 1) & 15);                                                    \
                    REFILL;                                                                       \
                    mma1(RR, BC);                                                                \
                    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                            \
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                        \
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
                        if (i_ < (NV)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             \
                    }                                                                             \
                    __builtin_amdgcn_sched_barrier(0);
                    EVX_P1(R0, B0, B1, 0, ldP(R6, ht, 6); ldP(R7, ht, 7), 6)
                    EVX_P1(R1, B1, B0, 1, ldP(R0, ht, 8), 3)   EVX_P1(R2, B0, B1, 2, ldP(R1, ht, 9), 3)   EVX_P1(R3, B1, B0, 3, ldP(R2, ht, 10), 3)
                    EVX_P1(R4, B0, B1, 4, ldP(R3, ht, 11), 3)  EVX_P1(R5, B1, B0, 5, ldP(R4, ht, 12), 3)  EVX_P1(R6, B0, B1, 6, ldP(R5, ht, 13), 3)
                    EVX_P1(R7, B1, B0, 7, ldP(R6, ht, 14), 3)  EVX_P1(R0, B0, B1, 8, ldP(R7, ht, 15), 3)
                    // from here on the ring goes over to what comes next (set i of it)
                    EVX_P1(R1, B1, B0, 9, ldN(R0, nq1, nx1, 0), 3)   EVX_P1(R2, B0, B1, 10, ldN(R1, nq1, nx1, 1), 3)  EVX_P1(R3, B1, B0, 11, ldN(R2, nq1, nx1, 2), 3)
                    EVX_P1(R4, B0, B1, 12, ldN(R3, nq1, nx1, 3), 3)  EVX_P1(R5, B1, B0, 13, ldN(R4, nq1, nx1, 4), 3)  EVX_P1(R6, B0, B1, 14, ldN(R5, nq1, nx1, 5), 3)
                    EVX_P1(R7, B1, B0, 15, ldN(R6, nq1, nx1, 6), 3)
#undef EVX_P1
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[j] += acc1b[j];
                    stamp();
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)(htn * 32 + 8 * q) * 4u, 0);   // next chunk's bias
                }
                // SnakeBeta + split of (row tile j, register group q) of chunk hc: C/D register 4 q + e is hidden channel 8 q + 4 half + e
                auto snake = [&](int jq) {
                    const int j = jq >> 2, q = jq & 3;
                    const f32x4 saq = *(const f32x4*)(Sv + ht * 32 + 8 * q + 4 * lh), sbq = *(const f32x4*)(Sv + mp.M1 + ht * 32 + 8 * q + 4 * lh);
                    f32x4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = acc1[j][4 * q + e]; hv[e] = fmaf(sbq[e], ev_sin2(v * saq[e]), v); }
                    evx_split4(hv, hq[jq][0], hq[jq][1], hq[jq][2]);
                };
                auto phase2 = [&](auto with_snake) {
                    constexpr bool WS = decltype(with_snake)::value;     // (a template parameter, not a branch: the vector arithmetic must sit in the MFMAs' basic block)
                    // ================= phase 2 of chunk hc - 1: acc2 += W2[64 channels of this wave][chunk] . h, K = 128 = 8 slabs;
                    // one (j, q) group of chunk hc's SnakeBeta per slab in the MFMAs' shadow =================
                    const int hp = hc - 1;
                    // what comes next: phase 1 of chunk hc + 1; after the last phase 1 of the pass the phase 2 of chunk hc; after the
                    // last phase 2 nothing (a harmless re-read)
                    const bool nq = p1 && hc + 1 >= cB;
                    const int nx = hc + 1 < cB ? ht + 4 : (p1 ? hc : cA * 4 + wave);
                    ldBh(B0, 0);
#define EVX_P2(RA, RB, BC, BN_, SL, REFILL)                                                       \
                    ldBh(BN_, ((SL) + 1) & 7);                                                    \
                    REFILL;                                                                       \
                    mma2(RA, RB, BC);                                                            \
                    if constexpr (WS) {                                                           \
                        snake(SL);                                                                \
                        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                        \
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);                   \
                        }                                                                         \
                    } else {                                                                      \
                        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                        \
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
                        }                                                                         \
                    }                                                                             \
                    __builtin_amdgcn_sched_barrier(0);
                    EVX_P2(R0, R1, B0, B1, 0, ldN(R6, true, hp, 6); ldN(R7, true, hp, 7))
                    EVX_P2(R2, R3, B1, B0, 1, ldQ(R0, hp, 4, 0); ldQ(R1, hp, 4, 1))  EVX_P2(R4, R5, B0, B1, 2, ldQ(R2, hp, 5, 0); ldQ(R3, hp, 5, 1))
                    EVX_P2(R6, R7, B1, B0, 3, ldQ(R4, hp, 6, 0); ldQ(R5, hp, 6, 1))  EVX_P2(R0, R1, B0, B1, 4, ldQ(R6, hp, 7, 0); ldQ(R7, hp, 7, 1))
                    // from here on the ring goes over to what comes next (sets 6 and 7: by that phase's first step)
                    EVX_P2(R2, R3, B1, B0, 5, ldN(R0, nq, nx, 0); ldN(R1, nq, nx, 1))  EVX_P2(R4, R5, B0, B1, 6, ldN(R2, nq, nx, 2); ldN(R3, nq, nx, 3))
                    EVX_P2(R6, R7, B1, B0, 7, ldN(R4, nq, nx, 4); ldN(R5, nq, nx, 5))
#undef EVX_P2
                };
                if (p2) { if (p1) phase2(std::true_type{}); else phase2(std::false_type{}); }
                else if (p1) {
#pragma unroll
                    for (int jq = 0; jq < 8; ++jq) snake(jq);
                }
                stamp();
                if (p1) {
                    ev_lds_barrier();        // every wave is done reading the previous chunk's planes (its phase 2)
#pragma unroll
                    for (int jq = 0; jq < 8; ++jq) {
                        char* dst = Hb + ((jq >> 2) * 32 + li) * HRS + (wave * 32 + 8 * (jq & 3) + 4 * lh) * 2;
                        *(uint2*)(dst) = hq[jq][0]; *(uint2*)(dst + 2 * HC) = hq[jq][1]; *(uint2*)(dst + 4 * HC) = hq[jq][2];
                    }
                    ev_lds_barrier();
                    stamp();
                }
            }
            auto acc_io = [&](unsigned base, int mode) {    // mode 0: store (write-through), 1: add from memory
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned soff = base + (unsigned)((a * 2 + j) * 4 + q) * 1024u;      // (wave-uniform: scalar offset)
                            if (mode == 0) {
                                const f32x4 v = {acc2[a][j][4 * q], acc2[a][j][4 * q + 1], acc2[a][j][4 * q + 2], acc2[a][j][4 * q + 3]};
                                ev_bstore4_sc1(rPart, pelem, soff, v);
                            } else {
                                const f32x4 v = ev_bload4_sc1(rPart, pelem, soff);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc2[a][j][4 * q + e] += v[e];
                            }
                        }
            };
            if (spilled) { acc_io((unsigned)g * pslot + pslot / 2, 1); spilled = false; }
            if (c0 != 0) {                             // not the owner: publish the partial tile (no bias) and move on
                acc_io((unsigned)g * pslot, 0);
                pend_pub = true;
                break;
            }
            bool again = false;
            while (c1 < nchunk) {                      // owner: add the contributors' partials in ascending workgroup order
                int n = 0;
                while (n < 64 && gi + n < (int)gridDim.x && sk_start(mp.sk, gi + n) < tile_end) ++n;
                if (n == 0) break;
                const int ready = sk_wait_many(mp.sk, gi, n, tag, tid, skw);
                for (int k = 0; k < ready; ++k) acc_io((unsigned)(gi + k) * pslot, 1);
                gi += ready;
                if (ready < n) {                       // gi is not there in time: spill the running sum, compute its share here
                    const int sgi = sk_start(mp.sk, gi);
                    int egi = sk_start(mp.sk, gi + 1);
                    egi = egi < tile_end ? egi : tile_end;
                    acc_io((unsigned)g * pslot + pslot / 2, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cA = sgi - t * nchunk; cB = egi - t * nchunk;
                    spilled = true; again = true;
                    ++gi;
                    break;
                }
            }
            if (again) {
                const int ht0 = cA * 4 + wave;
                ldP(R0, ht0, 0); ldP(R1, ht0, 1); ldP(R2, ht0, 2); ldP(R3, ht0, 3); ldP(R4, ht0, 4); ldP(R5, ht0, 5); ldP(R6, ht0, 6); ldP(R7, ht0, 7);
                continue;
            }
            // ---- epilogue: + residual, * mask, store (its first barrier also retires the last phase 2's LDS reads)
            conv_epilogue_lean<2, 2, 1>(p, acc2, smem + wave * (32 * 68), wave * 64, n0, lane);
            break;
        }
    }
    if (pend_pub) sk_publish(mp.sk, g, tag, tid);
    sk_arrive(mp.sk, tag, tid);
}

// ---------------------------------------------------------------------------
// ln_mlp_h16_kernel: ln_mlp_split_kernel in the fp16 form of conv_h16_kernel (two block-scaled fp16 pieces per operand, three products).
// Scales: the two linears' weight scales from the loader; sx from the maximum of the normalised 64 x 256 tile (one more LDS exchange in the
// staging); one scale per hidden chunk from the maximum of SnakeBeta over the workgroup's 64 x 128 values, exchanged at the barrier the
// plane write has anyway.  The hidden values wait in registers in true units until that scale is known; the phase-2 accumulators are
// rescaled (exact power of two) when the chunk scale changes and return to true units at the end of a pass.
// ---------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ __launch_bounds__(256, 1) void ln_mlp_h16_kernel(const MlpParams mp) {
    constexpr int NT = 64, C = 256, HC = 128;
    constexpr int XRS = 4 * C + 16, HRS = 4 * HC + 16;   // LDS row strides in bytes: two fp16 planes (65 x 16, 33 x 16)
    const ConvParams& p = mp.ep;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // [NT][XRS]: LN(x) times the tile scale as two fp16 planes
    char* Hb = Xb + NT * XRS;                           // [NT][HRS]: SnakeBeta(hidden chunk) times the chunk scale as two fp16 planes
    int* skw = (int*)(Hb + NT * HRS);
    float* Sv = (float*)(Hb + NT * HRS + 16);          // SnakeBeta vectors over the hidden width: [alpha (M1) | 1 / beta (M1)], M1 <= 1024
    float* red = Sv + 2 * mp.M1;                       // the waves' maxima: [0..3] of the normalised rows, [4..7] of a hidden chunk; [8..11], [12..15]: their finite-only repeats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(mp.X), rW1 = ev_rsrc(mp.W1h), rW2 = ev_rsrc(mp.W2h);
    const unsigned wlane = (unsigned)lane * 16u;
    const int nchunk = mp.M1 / HC;
    const int KS2 = mp.M1 / 16;                         // slabs per 32-row tile of W2
    const int g = blockIdx.x;
    const unsigned tag = sk_tag(mp.sk);
    int u = sk_start(mp.sk, g);
    const int ue = sk_start(mp.sk, g + 1);
    const unsigned claim_prev = sk_claim_start(mp.sk, g, threadIdx.x);    // (announce this workgroup: an owner that finds it absent takes its first share over)
    bool first_seg = true;
    f32x4 R0[2], R1[2], R2[2], R3[2], R4[2], R5[2], R6[2], R7[2];
    auto ldP = [&](f32x4 (&dst)[2], int ht, int sl) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dst[pc] = ev_bload4(rW1, wlane, (unsigned)(((ht * (C / 16) + sl) * 2 + pc) * 1024));
    };
    auto ldQ = [&](f32x4 (&dst)[2], int hc, int sl, int a) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dst[pc] = ev_bload4(rW2, wlane, (unsigned)((((wave * 2 + a) * KS2 + hc * (HC / 16) + sl) * 2 + pc) * 1024));
    };
    const __amdgpu_buffer_rsrc_t rB1 = ev_rsrc(mp.b1);
    // ring set i of the phase that comes next, whichever it is (no load under a branch): phase 1 of hidden tile hx -> slab i of W1's
    // row tile hx; phase 2 of chunk hx -> slab i / 2 of W2's row tile 2 wave + (i & 1)
    auto ldN = [&](f32x4 (&dst)[2], bool q2, int hx, int i) {
        const __amdgpu_buffer_rsrc_t rs = q2 ? rW2 : rW1;
        const unsigned off = q2 ? (unsigned)((((wave * 2 + (i & 1)) * KS2 + hx * (HC / 16) + (i >> 1)) * 2) * 1024) : (unsigned)(((hx * (C / 16) + i) * 2) * 1024);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) dst[pc] = ev_bload4(rs, wlane, off + (unsigned)(pc * 1024));
    };
    {   // (all loads first: one latency episode; published by the first staging barrier)
        float va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = q * 256 + tid; va[q] = i < mp.M1 ? mp.alpha[i] : 0.f; vb[q] = i < mp.M1 ? mp.binv[i] : 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int i = q * 256 + tid; if (i < mp.M1) { Sv[i] = va[q]; Sv[mp.M1 + i] = vb[q]; } }
    }
    const unsigned coff = (unsigned)(4 * lh) * 4u;
    const char* xrow = Xb + li * XRS + 16 * lh;
    const char* hrow = Hb + li * HRS + 16 * lh;
    const __amdgpu_buffer_rsrc_t rPart = ev_rsrc(mp.sk.part);
    const unsigned pslot = (unsigned)mp.sk.part_floats * 8u;
    const unsigned pelem = (unsigned)(wave * 16) * 1024u + wlane;      // element (wave, a, j, q) of a partial tile = 64 lanes x 16 B
    bool pend_pub = false;
    int nst = 0;                                       // diagnostic (EV_MLP_STAMPS): up to 32 s_memrealtime stamps per workgroup
    auto stamp = [&]() { if (p.stamps && tid == 0 && nst < 32) p.stamps[32 * g + nst] = __builtin_amdgcn_s_memrealtime(); ++nst; };
    stamp();
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};

    while (u < ue) {
        const int t = u / nchunk, c0 = u - t * nchunk;
        const int c1 = (ue - u < nchunk - c0) ? c0 + (ue - u) : nchunk;
        u += c1 - c0;
        const int n0 = t * NT;
        {   // tiles that contain no storable row (pure padding) do nothing — owner and contributors agree, the test only reads t
            int t_first = (n0 % p.S) - p.P;
            int dist;
            if (t_first >= 0 && t_first < p.T) dist = 0;
            else if (t_first < 0) dist = -t_first;
            else dist = p.S - (n0 % p.S) + p.P;
            if (dist >= NT || n0 + dist >= p.nrows) continue;
        }
        if (first_seg) {                               // (only a workgroup's first segment can be a contributor's share)
            first_seg = false;
            if (c0 != 0 && sk_claim_taken(mp.sk, claim_prev, tid, skw)) continue;    // its owner computes it: skip
        }
        {
            const int ht0 = c0 * 4 + wave;
            ldP(R0, ht0, 0); ldP(R1, ht0, 1); ldP(R2, ht0, 2); ldP(R3, ht0, 3); ldP(R4, ht0, 4); ldP(R5, ht0, 5); ldP(R6, ht0, 6); ldP(R7, ht0, 7);
        }
        ev_lds_barrier();                              // the previous segment's epilogue is done with its LDS slabs
        float sx;
        {   // ---- stage + LayerNorm + split: wave w owns rows 16 w .. 16 w + 15, a row = 4 channels per lane
            const f32x4 gm = *(const f32x4*)(mp.ln_g + lane * 4), be = *(const f32x4*)(mp.ln_b + lane * 4);
            f32x4 xv[16];                               // all sixteen rows requested at once: one latency episode
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = n0 + wave * 16 + r;
                xv[r] = ev_bload4(rX, ((unsigned)(gr < p.nrows ? gr : 0) * mp.ldx + lane * 4) * 4u, 0);   // (beyond the tensor: pad row 0)
            }
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xv[r] = ev_ln256_row(xv[r], gm, be, mp.ln_eps);
                mx = fmaxf(mx, evh_absmax4(xv[r]));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) red[wave] = mx;
            ev_lds_barrier();
            float tmx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (!evh_is_finite(tmx)) {                  // an Inf among the normalised rows (workgroup-uniform): the finite maximum sets the scale
                mx = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, evh_absmax4_finite(xv[r]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                if (lane == 0) red[8 + wave] = mx;
                ev_lds_barrier();
                tmx = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));
            }
            sx = evh_scale_for(tmx);                    // the tile's activation scale (conv_h16_kernel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                uint2 q0v, q1v;
                evh_split4(xv[r] * sx, q0v, q1v);
                char* dst = Xb + (wave * 16 + r) * XRS + lane * 8;
                *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
            }
        }
        if (pend_pub) { sk_publish(mp.sk, g, tag, tid); pend_pub = false; }   // (drain + barrier + flag: the barrier also publishes the staged rows)
        else ev_lds_barrier();
        stamp();

        int cA = c0, cB = c1;
        bool spilled = false;
        int gi = g + 1;
        const int tile_end = (t + 1) * nchunk;
        for (;;) {
            // accumulator units (all powers of two): phase 1 runs in w1_scale * sx; phase 2 in w2_scale * (scale of the hidden chunk whose
            // planes are in LDS) — acc2 is rescaled when that changes and brought back to true units at the end of the pass
            const float u1 = mp.w1_scale * sx, inv1 = 1.0f / u1;
            float cur_unit = mp.w2_scale, sh_lds = 1.f;
            f32x16 acc2[2][2];                         // this wave's 64 output channels x 64 rows, alive across all chunks of the pass
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias && cA == 0) bq = *(const f32x4*)(p.bias + wave * 64 + a * 32 + 8 * q + 4 * lh) * cur_unit;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[a][j][4 * q + e] = bq[e];
                }
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)((cA * 4 + wave) * 32 + 8 * q) * 4u, 0);
            // Software pipeline over the chunks: phase 1 of chunk hc, then phase 2 of chunk hc - 1 with the SnakeBeta + split arithmetic of
            // chunk hc in the shadow of its MFMAs (one wave per SIMD: nothing else would fill the matrix pipe during ~800 vector
            // instructions), then the pieces go to LDS between two barriers.  One more phase 2 after the loop.
            f32x4 hvs[8];                              // chunk hc's hidden values in true units, (j, q) -> 4 channels
            float hmax = 0.f;
            f32x4 B0[2][2], B1[2][2];
            auto ldBh = [&](f32x4 (&dst)[2][2], int sl) {
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                    for (int j = 0; j < 2; ++j) dst[pc][j] = *(const f32x4*)(hrow + j * 32 * HRS + pc * (2 * HC) + sl * 32);
            };
            auto mma2 = [&](const f32x4 (&a0)[2], const f32x4 (&a1)[2], const f32x4 (&b)[2][2]) {
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[PA[tt]]), __builtin_bit_cast(f16x8, b[PB[tt]][j]), acc2[0][j], 0, 0, 0);
                        acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[PA[tt]]), __builtin_bit_cast(f16x8, b[PB[tt]][j]), acc2[1][j], 0, 0, 0);
                    }
            };
            __builtin_assume(cA < cB);                 // (a pass always holds at least one chunk: no zero-trip path for the register allocator to keep the ring alive across)
            for (int hc = cA; hc <= cB; ++hc) {
                const bool p1 = hc < cB, p2 = hc > cA;             // phase 1 of chunk hc / phase 2 of chunk hc - 1 in this iteration
                const int ht = (p1 ? hc : cA) * 4 + wave;          // (past the last chunk: harmless re-reads)
                const int htn = hc + 1 < cB ? ht + 4 : cA * 4 + wave;
                // (two partial accumulators per row tile = four accumulation chains for a wave that is alone on its SIMD; tools/mfma_chain_probe.hip
                // prices two against four chains at 344 vs 315 ns per slab of this loop)
                f32x16 acc1[2], acc1b[2];
                if (p1) {
                    // ================= phase 1: acc1 = W1[ht] . LN(x) + b1, K = 256 = 16 slabs; the ring holds slabs 0..7 of ht =================
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { acc1[j][4 * q + e] = bq[q][e] * u1; acc1b[j][4 * q + e] = 0.f; }
                    auto ldB = [&](f32x4 (&dst)[2][2], int sl) {
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                            for (int j = 0; j < 2; ++j) dst[pc][j] = *(const f32x4*)(xrow + j * 32 * XRS + pc * (2 * C) + sl * 32);
                    };
                    auto mma1 = [&](const f32x4 (&a)[2], const f32x4 (&b)[2][2]) {
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                if (tt & 1) acc1b[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA[tt]]), __builtin_bit_cast(f16x8, b[PB[tt]][j]), acc1b[j], 0, 0, 0);
                                else acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA[tt]]), __builtin_bit_cast(f16x8, b[PB[tt]][j]), acc1[j], 0, 0, 0);
                            }
                    };
                    const int hq2 = p2 ? hc - 1 : hc;              // the phase 2 that follows: the previous chunk's (first iteration: none — re-reads)
                    // One wave per SIMD: nothing hides a block of loads between two blocks of MFMAs.  The loads of a step — the B fragments of
                    // the next slab and the refill of the ring set consumed one step EARLIER (its registers are free) — are interleaved with
                    // the step's MFMAs by an explicit schedule (sched_group_barrier: 2 MFMAs, 1 LDS read, [1 fragment load]).  The set
                    // consumed by a phase's last step is refilled by the first step of whatever phase comes next (sets 6 and 7: a re-load of
                    // what is already there is harmless).  Per-phase stamps (EV_MLP_STAMPS, profiles/r03_ln_mlp_split_stamps.txt): 3.9 us per
                    // phase 1 = 245 ns per slab, where twelve MFMAs alone take 219 ns at the clock the chip holds under bf16 MFMA load
                    // (tools/mfma_chain_probe.hip: 18.3 ns each, i.e. 32 cycles at ~1.75 GHz) and the probe's loop with the same loads 315 ns.
                    const bool nq1 = p2 || hc + 1 >= cB;                       // what follows this phase 1: a phase 2 (of chunk nx1) or the next chunk's phase 1
                    const int nx1 = p2 ? hq2 : (hc + 1 < cB ? ht + 4 : hc);
                    ldB(B0, 0);
#define EVX_P1(RR, BC, BN_, SL, REFILL, NV)                                                       \
                    ldB(BN_, ((SL) + 1) & 15);                                                    \
                    REFILL;                                                                       \
                    mma1(RR, BC);                                                                \
                    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                            \
                        if (i_ < 3) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);            \
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        \
                        if (i_ < (NV)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         \
                    }                                                                             \
                    __builtin_amdgcn_sched_barrier(0);
                    EVX_P1(R0, B0, B1, 0, ldP(R6, ht, 6); ldP(R7, ht, 7), 4)
                    EVX_P1(R1, B1, B0, 1, ldP(R0, ht, 8), 2)   EVX_P1(R2, B0, B1, 2, ldP(R1, ht, 9), 2)   EVX_P1(R3, B1, B0, 3, ldP(R2, ht, 10), 2)
                    EVX_P1(R4, B0, B1, 4, ldP(R3, ht, 11), 2)  EVX_P1(R5, B1, B0, 5, ldP(R4, ht, 12), 2)  EVX_P1(R6, B0, B1, 6, ldP(R5, ht, 13), 2)
                    EVX_P1(R7, B1, B0, 7, ldP(R6, ht, 14), 2)  EVX_P1(R0, B0, B1, 8, ldP(R7, ht, 15), 2)
                    // from here on the ring goes over to what comes next (set i of it)
                    EVX_P1(R1, B1, B0, 9, ldN(R0, nq1, nx1, 0), 2)   EVX_P1(R2, B0, B1, 10, ldN(R1, nq1, nx1, 1), 2)  EVX_P1(R3, B1, B0, 11, ldN(R2, nq1, nx1, 2), 2)
                    EVX_P1(R4, B0, B1, 12, ldN(R3, nq1, nx1, 3), 2)  EVX_P1(R5, B1, B0, 13, ldN(R4, nq1, nx1, 4), 2)  EVX_P1(R6, B0, B1, 14, ldN(R5, nq1, nx1, 5), 2)
                    EVX_P1(R7, B1, B0, 15, ldN(R6, nq1, nx1, 6), 2)
#undef EVX_P1
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc1[j] = (acc1[j] + acc1b[j]) * inv1;      // back to true units
                    stamp();
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = ev_bload4(rB1, coff + (unsigned)(htn * 32 + 8 * q) * 4u, 0);   // next chunk's bias
                }
                // SnakeBeta + split of (row tile j, register group q) of chunk hc: C/D register 4 q + e is hidden channel 8 q + 4 half + e
                auto snake = [&](int jq) {
                    const int j = jq >> 2, q = jq & 3;
                    const f32x4 saq = *(const f32x4*)(Sv + ht * 32 + 8 * q + 4 * lh), sbq = *(const f32x4*)(Sv + mp.M1 + ht * 32 + 8 * q + 4 * lh);
                    f32x4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = acc1[j][4 * q + e]; hv[e] = fmaf(sbq[e], ev_sin2(v * saq[e]), v); }
                    hvs[jq] = hv;
                    hmax = fmaxf(fmaxf(hmax, fmaxf(fabsf(hv[0]), fabsf(hv[1]))), fmaxf(fabsf(hv[2]), fabsf(hv[3])));
                };
                auto phase2 = [&](auto with_snake) {
                    constexpr bool WS = decltype(with_snake)::value;     // (a template parameter, not a branch: the vector arithmetic must sit in the MFMAs' basic block)
                    // ================= phase 2 of chunk hc - 1: acc2 += W2[64 channels of this wave][chunk] . h, K = 128 = 8 slabs;
                    // one (j, q) group of chunk hc's SnakeBeta per slab in the MFMAs' shadow =================
                    const int hp = hc - 1;
                    // what comes next: phase 1 of chunk hc + 1; after the last phase 1 of the pass the phase 2 of chunk hc; after the
                    // last phase 2 nothing (a harmless re-read)
                    const bool nq = p1 && hc + 1 >= cB;
                    const int nx = hc + 1 < cB ? ht + 4 : (p1 ? hc : cA * 4 + wave);
                    {   // the planes in LDS carry the scale sh_lds: bring the running sums to that unit
                        const float nu = mp.w2_scale * sh_lds, f = nu / cur_unit;
                        cur_unit = nu;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc2[a][j][r] *= f;
                    }
                    ldBh(B0, 0);
#define EVX_P2(RA, RB, BC, BN_, SL, REFILL)                                                       \
                    ldBh(BN_, ((SL) + 1) & 7);                                                    \
                    REFILL;                                                                       \
                    mma2(RA, RB, BC);                                                            \
                    if constexpr (WS) {                                                           \
                        snake(SL);                                                                \
                        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                        \
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x002, 30, 0);                   \
                        }                                                                         \
                    } else {                                                                      \
                        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                        \
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
                        }                                                                         \
                    }                                                                             \
                    __builtin_amdgcn_sched_barrier(0);
                    EVX_P2(R0, R1, B0, B1, 0, ldN(R6, true, hp, 6); ldN(R7, true, hp, 7))
                    EVX_P2(R2, R3, B1, B0, 1, ldQ(R0, hp, 4, 0); ldQ(R1, hp, 4, 1))  EVX_P2(R4, R5, B0, B1, 2, ldQ(R2, hp, 5, 0); ldQ(R3, hp, 5, 1))
                    EVX_P2(R6, R7, B1, B0, 3, ldQ(R4, hp, 6, 0); ldQ(R5, hp, 6, 1))  EVX_P2(R0, R1, B0, B1, 4, ldQ(R6, hp, 7, 0); ldQ(R7, hp, 7, 1))
                    // from here on the ring goes over to what comes next (sets 6 and 7: by that phase's first step)
                    EVX_P2(R2, R3, B1, B0, 5, ldN(R0, nq, nx, 0); ldN(R1, nq, nx, 1))  EVX_P2(R4, R5, B0, B1, 6, ldN(R2, nq, nx, 2); ldN(R3, nq, nx, 3))
                    EVX_P2(R6, R7, B1, B0, 7, ldN(R4, nq, nx, 4); ldN(R5, nq, nx, 5))
#undef EVX_P2
                };
                if (p2) { if (p1) phase2(std::true_type{}); else phase2(std::false_type{}); }
                else if (p1) {
#pragma unroll
                    for (int jq = 0; jq < 8; ++jq) snake(jq);
                }
                stamp();
                if (p1) {
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) hmax = fmaxf(hmax, __shfl_xor(hmax, o, 64));
                    if (lane == 0) red[4 + wave] = hmax;
                    hmax = 0.f;
                    ev_lds_barrier();        // every wave is done reading the previous chunk's planes (its phase 2); the waves' maxima are published
                    float hm = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
                    if (!evh_is_finite(hm)) {           // (workgroup-uniform slow path, see evh_is_finite)
                        float m2 = 0.f;
#pragma unroll
                        for (int jq = 0; jq < 8; ++jq) m2 = fmaxf(m2, evh_absmax4_finite(hvs[jq]));
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
                        if (lane == 0) red[12 + wave] = m2;
                        ev_lds_barrier();
                        hm = fmaxf(fmaxf(red[12], red[13]), fmaxf(red[14], red[15]));
                    }
                    sh_lds = evh_scale_for(hm);         // this chunk's scale
#pragma unroll
                    for (int jq = 0; jq < 8; ++jq) {
                        uint2 q0v, q1v;
                        evh_split4(hvs[jq] * sh_lds, q0v, q1v);
                        char* dst = Hb + ((jq >> 2) * 32 + li) * HRS + (wave * 32 + 8 * (jq & 3) + 4 * lh) * 2;
                        *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * HC) = q1v;
                    }
                    ev_lds_barrier();
                    stamp();
                }
            }
            {   // back to true units: partial tiles are handed over, spilled and stored in them
                const float f = 1.0f / cur_unit;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[a][j][r] *= f;
            }
            auto acc_io = [&](unsigned base, int mode) {    // mode 0: store (write-through), 1: add from memory
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned soff = base + (unsigned)((a * 2 + j) * 4 + q) * 1024u;      // (wave-uniform: scalar offset)
                            if (mode == 0) {
                                const f32x4 v = {acc2[a][j][4 * q], acc2[a][j][4 * q + 1], acc2[a][j][4 * q + 2], acc2[a][j][4 * q + 3]};
                                ev_bstore4_sc1(rPart, pelem, soff, v);
                            } else {
                                const f32x4 v = ev_bload4_sc1(rPart, pelem, soff);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc2[a][j][4 * q + e] += v[e];
                            }
                        }
            };
            if (spilled) { acc_io((unsigned)g * pslot + pslot / 2, 1); spilled = false; }
            if (c0 != 0) {                             // not the owner: publish the partial tile (no bias) and move on
                acc_io((unsigned)g * pslot, 0);
                pend_pub = true;
                break;
            }
            bool again = false;
            while (c1 < nchunk) {                      // owner: add the contributors' partials in ascending workgroup order
                int n = 0;
                while (n < 64 && gi + n < (int)gridDim.x && sk_start(mp.sk, gi + n) < tile_end) ++n;
                if (n == 0) break;
                const int ready = sk_wait_many(mp.sk, gi, n, tag, tid, skw);
                for (int k = 0; k < ready; ++k) acc_io((unsigned)(gi + k) * pslot, 1);
                gi += ready;
                if (ready < n) {                       // gi is not there in time: spill the running sum, compute its share here
                    const int sgi = sk_start(mp.sk, gi);
                    int egi = sk_start(mp.sk, gi + 1);
                    egi = egi < tile_end ? egi : tile_end;
                    acc_io((unsigned)g * pslot + pslot / 2, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    cA = sgi - t * nchunk; cB = egi - t * nchunk;
                    spilled = true; again = true;
                    ++gi;
                    break;
                }
            }
            if (again) {
                const int ht0 = cA * 4 + wave;
                ldP(R0, ht0, 0); ldP(R1, ht0, 1); ldP(R2, ht0, 2); ldP(R3, ht0, 3); ldP(R4, ht0, 4); ldP(R5, ht0, 5); ldP(R6, ht0, 6); ldP(R7, ht0, 7);
                continue;
            }
            // ---- epilogue: + residual, * mask, store (its first barrier also retires the last phase 2's LDS reads)
            conv_epilogue_lean<2, 2, 1>(p, acc2, smem + wave * (32 * 68), wave * 64, n0, lane);
            break;
        }
    }
    if (pend_pub) sk_publish(mp.sk, g, tag, tid);
    sk_arrive(mp.sk, tag, tid);
}

// ---------------------------------------------------------------------------
// ln_qkv_h16_kernel: MODE 1 of ln_mlp_kernel (y = Wqkv . LN(x) [+ b1], transformer.py:262-271) in the fp16 form of conv_h16_kernel, for
// the projection width 384 = 2 heads x 64 x {q, k, v}.  One workgroup per 64-row tile: rows staged once, normalised and split in
// registers (the staging of ln_mlp_h16_kernel, tile scale from the maximum of the normalised rows), then each wave runs its 96 output
// channels (three 32-row weight tiles) x 64 rows over K = 256 = 16 slabs, three products per slab pair, weight fragments straight from
// L2 three slabs ahead.  The accumulators run in units of w1_scale * sx and return to true units before the lean epilogue (two calls:
// its lane map needs a channel count that divides 256).
// ---------------------------------------------------------------------------
template <int DBG = 0>   // diagnostic build (EV_QKV_DBG=1): no weight-fragment loads inside the K loop
__global__ __launch_bounds__(256, 2) void ln_qkv_h16_kernel(const MlpParams mp) {
    constexpr int NT = 64, C = 256, TM = 3;
    constexpr int XRS = 4 * C + 16;
    const ConvParams& p = mp.ep;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* Xb = (char*)smem;                             // [NT][XRS]: LN(x) times the tile scale as two fp16 planes; later the epilogue slabs
    float* red = (float*)(Xb + NT * XRS);               // the waves' maxima
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rX = ev_rsrc(mp.X), rW1 = ev_rsrc(mp.W1h), rB1 = ev_rsrc(mp.b1);
    const unsigned wlane = (unsigned)lane * 16u;
    const int n0 = blockIdx.x * NT;
    auto stamp = [&](int i) { if (p.stamps && tid == 0) p.stamps[8 * blockIdx.x + i] = __builtin_amdgcn_s_memrealtime(); };   // diagnostic (EV_QKV_STAMPS)
    stamp(0);
    {   // tiles that contain no storable row (pure padding) do nothing
        int t_first = (n0 % p.S) - p.P;
        int dist;
        if (t_first >= 0 && t_first < p.T) dist = 0;
        else if (t_first < 0) dist = -t_first;
        else dist = p.S - (n0 % p.S) + p.P;
        if (dist >= NT || n0 + dist >= p.nrows) return;
    }
    // weight fragments: set s of the ring = slab (k 16 s .. 16 s + 15) of this wave's three row tiles, two pieces each
    f32x4 A[4][TM][2];
    auto ldA = [&](f32x4 (&dst)[TM][2], int sl) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) dst[i][pc] = ev_bload4(rW1, wlane, (unsigned)((((wave * TM + i) * (C / 16) + sl) * 2 + pc) * 1024));
    };
    ldA(A[0], 0); ldA(A[1], 1); ldA(A[2], 2);
    f32x4 bq[TM][4];
    const unsigned coff = (unsigned)(4 * lh) * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[i][q] = ev_bload4(rB1, coff + (unsigned)((wave * TM + i) * 32 + 8 * q) * 4u, 0);
    float sx;
    {   // ---- stage + LayerNorm + split: wave w owns rows 16 w .. 16 w + 15, a row = 4 channels per lane
        const f32x4 gm = *(const f32x4*)(mp.ln_g + lane * 4), be = *(const f32x4*)(mp.ln_b + lane * 4);
        f32x4 xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = n0 + wave * 16 + r;
            xv[r] = ev_bload4(rX, ((unsigned)(gr < p.nrows ? gr : 0) * mp.ldx + lane * 4) * 4u, 0);   // (beyond the tensor: pad row 0)
        }
        float mx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            xv[r] = ev_ln256_row(xv[r], gm, be, mp.ln_eps);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(xv[r][0]), fabsf(xv[r][1]))), fmaxf(fabsf(xv[r][2]), fabsf(xv[r][3])));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) red[wave] = mx;
        stamp(1);
        ev_lds_barrier();
        float tmx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (!evh_is_finite(tmx)) {                      // (workgroup-uniform slow path, see evh_is_finite)
            mx = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, evh_absmax4_finite(xv[r]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (lane == 0) red[4 + wave] = mx;
            ev_lds_barrier();
            tmx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        }
        sx = evh_scale_for(tmx);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            uint2 q0v, q1v;
            evh_split4(xv[r] * sx, q0v, q1v);
            char* dst = Xb + (wave * 16 + r) * XRS + lane * 8;
            *(uint2*)(dst) = q0v; *(uint2*)(dst + 2 * C) = q1v;
        }
    }
    ev_lds_barrier();
    stamp(2);
    const float u1 = mp.w1_scale * sx, inv1 = 1.0f / u1;
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = bq[i][q][e] * u1;
    const char* xrow = Xb + li * XRS + 16 * lh;
    f32x4 B[2][2][2];                                   // [ring][piece][row tile j]
    auto ldB = [&](f32x4 (&dst)[2][2], int sl) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[pc][j] = *(const f32x4*)(xrow + j * 32 * XRS + pc * (2 * C) + sl * 32);
    };
    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
    ldB(B[0], 0);
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) {
        if (sl + 1 < 16) ldB(B[(sl + 1) & 1], sl + 1);
        if (DBG != 1 && sl + 3 < 16) ldA(A[(sl + 3) & 3], sl + 3);
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[sl & 3][i][PA[tt]]), __builtin_bit_cast(f16x8, B[sl & 1][PB[tt]][j]), acc[i][j], 0, 0, 0);
        // (without this fence hipcc sinks the ring's loads behind the MFMAs of the slab they are meant to run under and drains vmcnt to 0
        // in every slab: the K loop then takes 10.5 us instead of 5, profiles/r03_ln_qkv_h16_stamps.txt)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!mp.qkv_pack) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv1;
    } else {
        // The packed form attn_out_h16_kernel consumes: every value times its tensor's power-of-two scale (q, k, v: channel tiles 0-3, 4-7,
        // 8-11) as two fp16 pieces, IN PLACE of the fp32 words: the lane holds channels c, c + 1 (c even) in registers 2u, 2u + 1, and leaves
        // word c = (hi c, hi c+1), word c + 1 = (lo c, lo c+1).  Same row layout and bytes as the fp32 tensor; the lean epilogue only moves words.
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float f = inv1 * mp.qkv_scale[(wave * TM + i) >> 2];      // (both powers of two)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float c0 = acc[i][j][2 * u] * f, c1 = acc[i][j][2 * u + 1] * f;
                    const _Float16 h0 = (_Float16)c0, h1 = (_Float16)c1;
                    const _Float16 l0 = (_Float16)(c0 - (float)h0), l1 = (_Float16)(c1 - (float)h1);
                    const f16x2 hh = {h0, h1}, ll = {l0, l1};
                    acc[i][j][2 * u] = __builtin_bit_cast(float, hh);
                    acc[i][j][2 * u + 1] = __builtin_bit_cast(float, ll);
                }
        }
    }
    stamp(3);
    // ---- store (the epilogue's first barrier retires the K loop's LDS reads; its slabs lie in the dead X planes)
    {
        f32x16 a2[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) a2[i][j] = acc[i][j];
        conv_epilogue_lean<2, 2, 1>(p, a2, smem + wave * (32 * 68), wave * 96, n0, lane);
    }
    stamp(4);
    {
        f32x16 a1[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) a1[0][j] = acc[2][j];
        conv_epilogue_lean<1, 2, 1>(p, a1, smem + wave * (32 * 68), wave * 96 + 64, n0, lane);
    }
    stamp(5);
}

// ---------------------------------------------------------------------------
// Self-attention over mel frames, 64-dim heads, fp32 matrix cores, flash-style
// online softmax.  Semantics of diffusers 0.25 Attention + SDPA with a FLOAT mask
// (transformer.py:266-271): scores = q.k/8 + m[key], m = 1.0 for frames inside the
// utterance length, 0.0 for padded frames (< Tp) — padded frames stay live keys.
//
// Per wave: 32 queries.  S^T = K.Q^T is computed with keys on the MFMA rows so the
// score tile lands with the query on the lane and the 32 keys in the 16 registers
// (x2 lane halves); each register is then directly the B operand of the P.V product
// (O^T += V^T.P^T), no LDS round trip for P.
// ---------------------------------------------------------------------------
struct AttnParams {
    const float* QKV; int ld;   // rows: [q(H*64) | k(H*64) | v(H*64)]
    float* O; int ldo;          // rows: H*64
    const float* rowmask;
    int S, P, T, H; float scale;
};

#define ATT_LDK 68
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[32 * ATT_LDK];
    __shared__ __attribute__((aligned(16))) float Vs[32 * ATT_LDK];
    __shared__ float Ms[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const size_t rowbase = (size_t)b * p.S + p.P;
    const int HD = p.H * 64;
    const float* Qp = p.QKV + hd * 64;
    const float* Kp = p.QKV + HD + hd * 64;
    const float* Vp = p.QKV + 2 * HD + hd * 64;

    // Q fragments: lane (query li, half lh) holds Q[q][8g + 4lh + s], pre-scaled
    f32x4 qf[8];
    {
        const int tq = q0 + li;
        const bool ok = tq < p.T;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(Qp + (rowbase + tq) * p.ld + 8 * g + 4 * lh);
            qf[g] = v * (p.scale * 1.44269504088896340736f);   // scores in the log2 domain: exp(x) = exp2(x * log2 e), one v_exp_f32 each
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -1e30f, lrun = 0.f;
    const bool wave_active = __builtin_amdgcn_readfirstlane(q0) < p.T;

    const int nkt = (p.T + 31) / 32;
    // K/V tiles are staged through registers one tile ahead: the global loads of tile kt+1 fly while tile kt is being
    // multiplied, so the per-tile critical path is LDS + MFMA only (matters most at small batch, where a workgroup's
    // 9-17 key tiles are a serial chain).
    const int sr = tid >> 3;             // key row within the tile, 0..31
    const int sc = (tid & 7) * 8;        // first of 8 dims, 0..56
    f32x4 k0, k1, v0, v1;
    float mk;
    auto kv_load = [&](int kt) {
        const int tk = kt * 32 + sr;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        k0 = z; k1 = z; v0 = z; v1 = z;
        if (tk < p.T) {
            const float* kr = Kp + (rowbase + tk) * p.ld + sc;
            const float* vr = Vp + (rowbase + tk) * p.ld + sc;
            k0 = *(const f32x4*)kr; k1 = *(const f32x4*)(kr + 4);
            v0 = *(const f32x4*)vr; v1 = *(const f32x4*)(vr + 4);
        }
        const int tm = kt * 32 + (tid & 31);
        mk = (tm < p.T) ? p.rowmask[rowbase + tm] : -1e30f;
    };
    kv_load(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        // publish the prefetched 32 keys x 64 dims of K and V (rows past T are zero, masked below)
        *(f32x4*)(Ks + sr * ATT_LDK + sc) = k0; *(f32x4*)(Ks + sr * ATT_LDK + sc + 4) = k1;
        *(f32x4*)(Vs + sr * ATT_LDK + sc) = v0; *(f32x4*)(Vs + sr * ATT_LDK + sc + 4) = v1;
        if (tid < 32) Ms[tid] = mk * 1.44269504088896340736f;
        __syncthreads();
        if (kt + 1 < nkt) kv_load(kt + 1);
        // a wave whose 32 queries all lie beyond the utterance (T = 516 = 4 x 128 + 4: three of the four waves of every fifth
        // workgroup; 15-25 % of all waves at the benchmark shapes) only helps staging the K / V tiles
        if (!wave_active) continue;
        // S^T[key][q] = sum_d K[key][d] * Q[q][d]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 a = *(const f32x4*)(Ks + li * ATT_LDK + 8 * g + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[g][e], s, 0, 0, 0);
        }
        // + mask[key]; key of register r on this lane half: (r&3) + 8*(r>>2) + 4*lh
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] += Ms[(r & 3) + 8 * (r >> 2) + 4 * lh];
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mnew); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        lrun = lrun * alpha + ps;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // O^T[d][q] += sum_key V[key][d] * P^T[key][q]; register r pairs keys (kr, kr+4) across the lane halves
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float a0 = Vs[key * ATT_LDK + li];
            const float a1 = Vs[key * ATT_LDK + 32 + li];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
        }
    }
    const int tq = q0 + li;
    if (tq < p.T) {
        const float inv = 1.0f / lrun;
        float* orow = p.O + (rowbase + tq) * p.ldo + hd * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            f32x4 c = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
            *(f32x4*)(orow + 8 * g + 4 * lh) = a;
            *(f32x4*)(orow + 32 + 8 * g + 4 * lh) = c;
        }
    }
}

// ---------------------------------------------------------------------------
// attn_out_kernel: self-attention of BOTH heads + the output projection + the residual, one launch (round 3):
//     H[q] += Wout . concat_h(softmax_k(q_h . k_h / 8 + m[k]) v_h) + bout          (transformer.py:262-271, diffusers Attention.to_out[0])
// A workgroup is ONE 32-query tile of one utterance; its four waves are (head h, key half kh): wave (h, kh) runs the flash-style
// loop of attention_kernel for head h over its half of the key tiles — on K / V tiles staged into LDS that is PRIVATE to the wave
// (register-prefetched one tile ahead), so the key loop has no workgroup barrier at all — then the two key halves of a head are
// merged through LDS (ascending order, deterministic), the normalised 32 x 128 output rows become the B operand of the
// 128 -> 256 projection (each wave 64 output channels, weight fragments straight from L2 as in conv_gemm_kernel), and the lean
// conv epilogue adds the residual rows and stores.  The additive mask rides in the QK product as one more MFMA (k-slot 0 = mask of
// the key, times 1), so the softmax has no per-register LDS read.
// Why this shape: at batch 64 the launches are about one round of workgroups; 128-query x one-head workgroups gave 640 / 384
// workgroups of very unequal cost (T = 516 = 4 x 128 + 4: every fifth one held 4 queries) — 29-42 % MFMA-busy — followed by a
// 128 -> 256 GEMM at 46-60 TFLOP/s.  32-query tiles are 1088 / 576 equal workgroups that the dispatcher deals out dynamically
// (two resident per CU), and the projection runs on rows that never leave the CU.
// ---------------------------------------------------------------------------
struct AttnOutParams {
    const float* QKV; int ld;   // rows: [q(2*64) | k(2*64) | v(2*64)]
    const float* rowmask;
    const float* Wout;          // fragment order [256/32][128/8][64 lanes][4]
    ConvParams ep;              // epilogue view: Y = R = the hidden rows (in place), ldy = ldr, Cout = 256, bias, nrows / S / P / T
    int S, P, T, B, nq;         // nq = query tiles per utterance that the 32 x 32 workgroups take (all of them, or all but a short last one)
    int ntail;                  // > 0: the last tile holds only ntail (<= 4) queries and is computed by one small workgroup per utterance on
                                // 4 x 4 MFMA blocks (blocks 0 .. B-1 of the grid, dispatched first): see attn_tail_path
    float scale; int xcd_map;   // xcd_map: B % 8 == 0 -> the tiles of an utterance share an XCD (its K / V stay in that XCD's L2)
    unsigned long long* stamps; // diagnostic (EV_ATTN_STAMPS): six s_memrealtime stamps per workgroup, or null
    // attn_out_h16_kernel: QKV holds fp16 piece pairs times the powers of two sq / sk / sv (ln_qkv_h16_kernel, qkv_pack)
    float inv_sq, inv_sk, inv_sv;
    float mask_a, mask_b;       // powers of two, both fp16 numbers, mask_a mask_b = 8 sq sk (the frame mask in accumulator units)
    const void* Wouth; float wo_scale;   // the projection as two fp16 pieces in conv_h16_kernel's fragment order, times wo_scale
};
// one packed word pair -> two fp32 values: words (hi c, hi c+1), (lo c, lo c+1)
__device__ __forceinline__ void ev_unpack_pair(float w_hi, float w_lo, float inv, float& a, float& b) {
    const f16x2 hh = __builtin_bit_cast(f16x2, w_hi), ll = __builtin_bit_cast(f16x2, w_lo);
    a = ((float)hh[0] + (float)ll[0]) * inv; b = ((float)hh[1] + (float)ll[1]) * inv;
}
__device__ __forceinline__ f32x4 ev_unpack4(const f32x4 w, float inv) {
    float a, b, c, d;
    ev_unpack_pair(w[0], w[1], inv, a, b); ev_unpack_pair(w[2], w[3], inv, c, d);
    const f32x4 r = {a, b, c, d};
    return r;
}

#define AO_LDK 68
#define AO_OLD 132
#define AO_TAILQ 4             // most queries a short last tile may hold to take the 4 x 4 path
// The short last query tile of an utterance (T = 516 = 16 x 32 + 4 queries, T / 2 = 258 = 8 x 32 + 2) as a workgroup of its own costs a
// full one — 32-column MFMA tiles for 2-4 queries — and pushed every launch into one more round of workgroups: per-workgroup stamps
// (profiles/r03_attn_workgroup_stamps.txt) show 64 of the 576 / 1088 workgroups starting when all others have finished, 40 % of the
// T / 2 launch.  Here those queries run on v_mfma_f32_4x4x1_16b_f32 — sixteen independent 4 x 4 outer products per instruction at the
// full fp32 matrix rate, so four query columns waste nothing — in one workgroup per utterance that is dispatched first and gone after
// a few microseconds.  (The same on the VECTOR ALUs was tried first: beside co-resident waves that stream MFMAs its dependent VALU
// chains got almost no issue slots and the 64 workgroups finished last.)  Layout of the instruction, confirmed with exact integer
// data (tools/mfma4x4_probe.hip): lane l = 4 block + x holds A[block][i = x] and B[block][j = x]; D register r of lane l is
// D[block][i = r][j = x].
//   scores:  block = (key group kg of 4 keys, dim half dh);  A = K[4 kg + x][32 dh + s], B = Q[query x][32 dh + s], s = 0..31  ->
//            D reg i = half score of key 4 kg + i for query x; the two dim halves are lane ^ 4
//   P . V:   block = 4 dims;  A = P[query x][key] (every block the same: gathered with ds_bpermute from the score layout),
//            B = V[key][lane]  ->  D reg i = O[query i][dim lane]
// Same (head, key half) waves, wave-private K / V tiles, merge of the key halves and projection (one output channel per thread here).
template <bool PK = false>   // PK: QKV in the packed fp16-pair form (decoded where it is staged: this path stays on the fp32 4 x 4 blocks)
__device__ __forceinline__ void attn_tail_path(const AttnOutParams& p, int b, float* smem, int tid, int lane, int wave) {
    const int x = lane & 3, dh = (lane >> 2) & 1, kg = lane >> 3, li = lane & 31, h = wave & 1, kh = wave >> 1;
    const int NT = p.ntail;
    const int q0 = p.nq * 32;
    const unsigned rowbase = (unsigned)b * p.S + p.P;
    float* Ks = smem + wave * (2 * 32 * AO_LDK);
    float* Vs = Ks + 32 * AO_LDK;
    const __amdgpu_buffer_rsrc_t rQ = ev_rsrc(p.QKV), rM = ev_rsrc(p.rowmask), rW = ev_rsrc(p.Wout);
    const float L2E = 1.44269504088896340736f;
    const unsigned ldb = (unsigned)p.ld * 4u;
    // this lane's query (x) and dim half (dh), pre-scaled into the log2 domain; queries beyond the tile's are zero
    f32x4 qv[8];
    {
        const bool ok = x < NT;
        const unsigned off = (rowbase + (unsigned)(ok ? q0 + x : 0)) * ldb + (unsigned)(h * 64 + 32 * dh) * 4u;
        const float qs = ok ? p.scale * L2E : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 w = ev_bload4(rQ, off + (unsigned)(4 * i) * 4u, 0);
            qv[i] = (PK ? ev_unpack4(w, p.inv_sq) : w) * qs;
        }
    }
    const int nkt = (p.T + 31) / 32, nh0 = (nkt + 1) / 2;
    const int kt0 = kh ? nh0 : 0, kt1 = kh ? nkt : nh0;
    f32x4 kr[8], vr[8];
    f32x4 mk4 = {0.f, 0.f, 0.f, 0.f};                  // masks of this lane's four keys (log2 domain)
    const int srow = lane >> 4, sc4 = (lane & 15) * 4;
    const unsigned kcol = (unsigned)(128 + h * 64 + sc4) * 4u, vcol = (unsigned)(256 + h * 64 + sc4) * 4u;
    auto kv_issue = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tk = kt * 32 + 4 * j + srow;
            const unsigned ro = (rowbase + (unsigned)(tk < p.T ? tk : p.T - 1)) * ldb;
            kr[j] = ev_bload4(rQ, ro + kcol, 0);
            vr[j] = ev_bload4(rQ, ro + vcol, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tm = kt * 32 + 4 * kg + i;
            const float m = ev_bload1(rM, (rowbase + (unsigned)(tm < p.T ? tm : p.T - 1)) * 4u, 0);
            mk4[i] = tm < p.T ? m * L2E : -1e30f;
        }
    };
    f32x4 o = {0.f, 0.f, 0.f, 0.f}, o2 = o;            // o[i] (+ o2[i]) = O[query i][dim lane], un-normalised
    float mrun = -1e30f, lrun = 0.f;                   // running max / sum of THIS lane's query x
    if (kt0 < kt1) kv_issue(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(f32x4*)(Ks + (4 * j + srow) * AO_LDK + sc4) = PK ? ev_unpack4(kr[j], p.inv_sk) : kr[j];
            *(f32x4*)(Vs + (4 * j + srow) * AO_LDK + sc4) = PK ? ev_unpack4(vr[j], p.inv_sv) : vr[j];
        }
        const f32x4 mkey = mk4;
        kv_issue(kt + 1 < kt1 ? kt + 1 : kt);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sc2 = sc;
        f32x4 k4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) k4[i] = *(const f32x4*)(Ks + (4 * kg + x) * AO_LDK + 32 * dh + 4 * i);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc = __builtin_amdgcn_mfma_f32_4x4x1f32(k4[i][0], qv[i][0], sc, 0, 0, 0);
            sc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(k4[i][1], qv[i][1], sc2, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_4x4x1f32(k4[i][2], qv[i][2], sc, 0, 0, 0);
            sc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(k4[i][3], qv[i][3], sc2, 0, 0, 0);
        }
        sc += sc2;
        float mx = -1e30f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = sc[i] + __shfl_xor(sc[i], 4, 64) + mkey[i]; mx = fmaxf(mx, sc[i]); }
        mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[i] = __builtin_amdgcn_exp2f(sc[i] - mnew); ps += sc[i]; }
        ps += __shfl_xor(ps, 8, 64); ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
        lrun = lrun * alpha + ps;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float ai = __shfl(alpha, i, 64); o[i] *= ai; o2[i] *= ai; }   // lane i holds query i's alpha (kg = 0, dh = 0, x = i)
        float pa[32], vb[32];                          // all gathers and reads in flight together, then the MFMAs on two accumulators
#pragma unroll
        for (int key = 0; key < 32; ++key) {
            pa[key] = __shfl(sc[key & 3], ((key >> 2) << 3) | x, 64);          // P[query x][key] from the lane that holds key group key / 4
            vb[key] = Vs[key * AO_LDK + lane];
        }
#pragma unroll
        for (int key = 0; key < 32; key += 2) {
            o = __builtin_amdgcn_mfma_f32_4x4x1f32(pa[key], vb[key], o, 0, 0, 0);
            o2 = __builtin_amdgcn_mfma_f32_4x4x1f32(pa[key + 1], vb[key + 1], o2, 0, 0, 0);
        }
    }
    o += o2;
    // ---- every lane gets all four queries' (m, l); merge the key halves, normalise, rows into Ot[query][128] (wave 0's dead region)
    float mq[4], lq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mq[i] = __shfl(mrun, i, 64); lq[i] = __shfl(lrun, i, 64); }
    float* Ot = smem;
    ev_lds_barrier();
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Ks[i * 64 + lane] = o[i];
        if (lane < 4) { Ks[256 + lane] = mq[lane & 3]; Ks[260 + lane] = lq[lane & 3]; }
    }
    ev_lds_barrier();
    if (kh == 0) {
        const float* Ps = smem + (wave + 2) * (2 * 32 * AO_LDK);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float m1 = Ps[256 + i], l1 = Ps[260 + i], o1 = Ps[i * 64 + lane];
            const float m = fmaxf(mq[i], m1);
            const float a0 = __builtin_amdgcn_exp2f(mq[i] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
            const float den = lq[i] * a0 + l1 * a1;
            o[i] = den > 0.f ? (o[i] * a0 + o1 * a1) / den : 0.f;
        }
    }
    ev_lds_barrier();                                  // (partner regions read; Ot lies in wave 0's tiles, dead since the first barrier)
    if (kh == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Ot[i * 128 + h * 64 + lane] = o[i];
    }
    ev_lds_barrier();
    // ---- projection + bias + residual: thread c owns output channel c of the tail rows (weights in MFMA-fragment order: element
    // (row c, k) sits at (((c / 32) * 16 + k / 8) * 64 + (c % 32) + 32 * ((k % 8) / 4)) * 4 + k % 4)
    (void)li;
    const int c = tid;
    float acc[4];
    const float bc = p.ep.bias ? p.ep.bias[c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bc;
#pragma unroll 4
    for (int kgp = 0; kgp < 16; ++kgp) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const f32x4 w4 = ev_bload4(rW, (unsigned)((((c >> 5) * 16 + kgp) * 64 + (c & 31) + 32 * hf) * 16), 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 x4 = *(const f32x4*)(Ot + j * 128 + kgp * 8 + hf * 4);
                acc[j] = fmaf(w4[0], x4[0], acc[j]); acc[j] = fmaf(w4[1], x4[1], acc[j]);
                acc[j] = fmaf(w4[2], x4[2], acc[j]); acc[j] = fmaf(w4[3], x4[3], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tq = q0 + j;
        if (j < NT && tq < p.T) {
            const size_t off = (size_t)(rowbase + tq) * p.ep.ldy + c;
            p.ep.Y[off] = acc[j] + p.ep.R[(size_t)(rowbase + tq) * p.ep.ldr + c];
        }
    }
}

__global__ __launch_bounds__(256, 2) void attn_out_kernel(const AttnOutParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int h = wave & 1, kh = wave >> 1;
    int b, qt;
    {
        int id = blockIdx.x;
        const int ntw = p.ntail > 0 ? p.B : 0;         // the short last tiles first: small 4 x 4-block workgroups (attn_tail_path)
        if (id < ntw) {
            if (p.stamps && tid == 0) p.stamps[6 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
            attn_tail_path(p, p.xcd_map ? (id & 7) + 8 * (id >> 3) : id, smem, tid, lane, wave);
            if (p.stamps && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); for (int k = 1; k < 6; ++k) p.stamps[6 * blockIdx.x + k] = t; }
            return;
        }
        id -= ntw;                                     // (B % 8 == 0 under xcd_map: the XCD of a block id is unchanged)
        if (p.xcd_map) { const int within = id >> 3; qt = within % p.nq; b = (id & 7) + 8 * (within / p.nq); }
        else { b = id / p.nq; qt = id - b * p.nq; }
    }
    const int q0 = qt * 32;
    const unsigned rowbase = (unsigned)b * p.S + p.P;
    auto stamp = [&](int k) { if (p.stamps && tid == 0) p.stamps[6 * blockIdx.x + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    float* Ks = smem + wave * (2 * 32 * AO_LDK);       // this wave's private K tile [32 keys][64 + 4]
    float* Vs = Ks + 32 * AO_LDK;                      // ... and V tile
    const __amdgpu_buffer_rsrc_t rQ = ev_rsrc(p.QKV), rM = ev_rsrc(p.rowmask), rW = ev_rsrc(p.Wout);
    const float L2E = 1.44269504088896340736f;
    const unsigned ldb = (unsigned)p.ld * 4u;

    // Q fragments: lane (query li, half lh) holds Q[q][8g + 4lh + s], pre-scaled into the log2 domain (exp(x) = exp2(x log2 e))
    f32x4 qf[8];
    {
        const int tq = q0 + li;
        const float qs = tq < p.T ? p.scale * L2E : 0.f;
        const unsigned off = (rowbase + (unsigned)(tq < p.T ? tq : 0)) * ldb + (unsigned)(h * 64 + 4 * lh) * 4u;
#pragma unroll
        for (int g = 0; g < 8; ++g) qf[g] = ev_bload4(rQ, off + (unsigned)(8 * g) * 4u, 0) * qs;
    }
    const int nkt = (p.T + 31) / 32, nh0 = (nkt + 1) / 2;
    const int kt0 = kh ? nh0 : 0, kt1 = kh ? nkt : nh0;
    // staging: float4 j of this lane = (key row 4j + lane/16, dims 4 (lane % 16) ..): a load instruction covers 4 rows x 256 B
    f32x4 kr[8], vr[8];
    float mk = 0.f;
    const int srow = lane >> 4, sc4 = (lane & 15) * 4;
    const unsigned kcol = (unsigned)(128 + h * 64 + sc4) * 4u, vcol = (unsigned)(256 + h * 64 + sc4) * 4u;
    auto kv_issue = [&](int kt) {      // every load unconditional: keys beyond the utterance re-read its last frame and are masked out
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tk = kt * 32 + 4 * j + srow;
            const unsigned ro = (rowbase + (unsigned)(tk < p.T ? tk : p.T - 1)) * ldb;
            kr[j] = ev_bload4(rQ, ro + kcol, 0);
            vr[j] = ev_bload4(rQ, ro + vcol, 0);
        }
        const int tm = kt * 32 + li;
        const float m = ev_bload1(rM, (rowbase + (unsigned)(tm < p.T ? tm : p.T - 1)) * 4u, 0);
        mk = tm < p.T ? m * L2E : -1e30f;
    };
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -1e30f, lrun = 0.f;
    if (kt0 < kt1) kv_issue(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt == kt0 + 1) stamp(1);                   // (first key tile done)
        // publish the prefetched tile to this wave's LDS (the wave's LDS operations execute in order: no barrier, no other reader)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(f32x4*)(Ks + (4 * j + srow) * AO_LDK + sc4) = kr[j];
            *(f32x4*)(Vs + (4 * j + srow) * AO_LDK + sc4) = vr[j];
        }
        const float amask = lh == 0 ? mk : 0.f;        // A operand of the mask product: k-slot 0 = mask of key li, k-slot 1 = 0
        kv_issue(kt + 1 < kt1 ? kt + 1 : kt);          // next tile (after the last: a harmless re-read)
        // S^T[key][q] = sum_d K[key][d] Q[q][d] + mask[key]
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 a = *(const f32x4*)(Ks + li * AO_LDK + 8 * g + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[g][e], s, 0, 0, 0);
        }
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(amask, 1.0f, s, 0, 0, 0);
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mnew); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        lrun = lrun * alpha + ps;
        mrun = mnew;
        if (!__all(alpha == 1.0f)) {                   // (wave-uniform: the running maximum of most tiles after the first few is old)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        // O^T[d][q] += sum_key V[key][d] P^T[key][q]; register r pairs keys (kr, kr + 4) across the lane halves
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float a0 = Vs[key * AO_LDK + li];
            const float a1 = Vs[key * AO_LDK + 32 + li];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
        }
    }
    stamp(2);
    // ---- first weight fragments of the projection (this wave: output channels 64 wave .. +63 = row tiles 2 wave, 2 wave + 1; 16
    // k-groups), into the staging registers: they land while the key halves are merged
    const unsigned wlane = (unsigned)lane * 16u;
    auto ldW = [&](int a, int kg) { return ev_bload4(rW, wlane, (unsigned)((wave * 2 + a) * 16 + kg) * 1024u); };
#pragma unroll
    for (int j = 0; j < 8; ++j) { kr[j] = ldW(0, j); vr[j] = ldW(1, j); }
    // ---- merge the two key halves of each head.  Wave (h, 1) leaves its state in its OWN (now dead) K / V region:
    // [8 float4 slots of o0 | 8 of o1] x 64 lanes, then m and l; wave (h, 0) adds it to its own state (half 0 + half 1) and writes
    // the normalised rows into Os[query][128] — inside wave 0's dead region (17.4 KB >= 16.9 KB)
    float* Os = smem;                                  // [32][AO_OLD]
    ev_lds_barrier();                                  // every wave has left its key loop
    if (kh == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = {o0[4 * j], o0[4 * j + 1], o0[4 * j + 2], o0[4 * j + 3]};
            const f32x4 c = {o1[4 * j], o1[4 * j + 1], o1[4 * j + 2], o1[4 * j + 3]};
            *(f32x4*)(Ks + (j * 64 + lane) * 4) = a;
            *(f32x4*)(Ks + ((4 + j) * 64 + lane) * 4) = c;
        }
        Ks[8 * 256 + lane] = mrun;
        Ks[8 * 256 + 64 + lane] = lrun;
    }
    ev_lds_barrier();
    if (kh == 0) {
        const float* Ps = smem + (wave + 2) * (2 * 32 * AO_LDK);       // region of wave (h, 1)
        const float m1 = Ps[8 * 256 + lane], l1 = Ps[8 * 256 + 64 + lane];
        const float m = fmaxf(mrun, m1);
        const float a0 = __builtin_amdgcn_exp2f(mrun - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
        const float inv = 1.0f / (lrun * a0 + l1 * a1);
        const float w0 = a0 * inv, w1 = a1 * inv;
        f32x4 oa[4], oc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 pa = *(const f32x4*)(Ps + (j * 64 + lane) * 4), pc = *(const f32x4*)(Ps + ((4 + j) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { oa[j][e] = o0[4 * j + e] * w0 + pa[e] * w1; oc[j][e] = o1[4 * j + e] * w0 + pc[e] * w1; }
        }
        // C/D register 4j + e of o0 is dim 8j + 4lh + e of query li (o1: + 32).  Os lies inside wave 0's region, dead since the
        // barrier after the key loops, and the partner states are read from the regions of waves 2 / 3: no further barrier here
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)(Os + li * AO_OLD + h * 64 + 8 * j + 4 * lh) = oa[j];
            *(f32x4*)(Os + li * AO_OLD + h * 64 + 32 + 8 * j + 4 * lh) = oc[j];
        }
    }
    ev_lds_barrier();
    stamp(3);
    // ---- projection: Y^T[256][32] = Wout . O^T, K = 128 = 16 k-groups; bias preloaded into the accumulators (lean epilogue convention)
    f32x16 acc[2][1];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.ep.bias) bq = *(const f32x4*)(p.ep.bias + wave * 64 + a * 32 + 8 * q + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][0][4 * q + e] = bq[e];
        }
    const float* orow = Os + li * AO_OLD + 4 * lh;
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
        const f32x4 bfr = *(const f32x4*)(orow + kg * 8);
        const f32x4 fa = kr[kg & 7], fb = vr[kg & 7];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], bfr[e], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[e], bfr[e], acc[1][0], 0, 0, 0);
        }
        if (kg < 8) { kr[kg] = ldW(0, kg + 8); vr[kg] = ldW(1, kg + 8); }
    }
    stamp(4);
    // ---- + residual rows, store (rows of THIS utterance only: a tile's tail rows may belong to the next one)
    conv_epilogue_lean<2, 1, 1>(p.ep, acc, smem + wave * (32 * 68), wave * 64, (int)rowbase + q0, lane, (int)rowbase, (int)rowbase + p.T);
    stamp(5);
}

// ---------------------------------------------------------------------------
// attn_out_h16_kernel: attn_out_kernel on the fp16 matrix pipe (round 4).  Same workgroup (one 32-query tile, waves = (head, key half),
// wave-private K / V tiles, no barrier in the key loop), same merge, projection and epilogue; the two products of the key loop run on
// v_mfma_f32_32x32x16_f16 over operands that are fp16 PIECE PAIRS written by the producer (ln_qkv_h16_kernel, qkv_pack): every q / k / v
// value x is stored, in place of its fp32 word pair, as x s = hi + lo (hi = nearest fp16, lo = nearest fp16 of the rest: 22 significand bits)
// with a power-of-two s per tensor that the LOADER derives from a bound no input can exceed (|LayerNorm(x)_c| <= sqrt(C - 1)|gamma_c| + |beta_c|,
// so |k_j| <= sum_c |W_jc| (...)): s maps the bound to (16384, 32768], nothing overflows, and values far below the bound keep an ABSOLUTE
// precision of 2^-25 / s = 2^-40 of the bound.  A row of the packed tensor reads, per four fp16 columns, (hi c, hi c+1, lo c, lo c+1), c even.
//   scores: S^T[key][q] = sum_d (hK + lK)(hQ + lQ).  A = 16 bytes of the key's row as they lie = (hK0 hK1 lK0 lK1 hK2 hK3 lK2 lK3): four dims, both
//           pieces; B1 = (hQ0 hQ1 hQ0 hQ1 hQ2 hQ3 hQ2 hQ3), B2 = the same of lQ: two MFMAs per four dims x two lane halves = ALL FOUR piece products
//           (16 MFMAs per 32-key tile for 64 dims; the three-product form would need 12 plus a re-arrangement of K in the vector ALUs).
//   P . V:  O^T[column][q] += sum_key V[key][column] P[q][key] over the 128 fp16 COLUMNS of V's row (both pieces of 64 dims: four M tiles); the A
//           operand is the transposed read ds_read_b64_tr_b16 of the V tile as it lies (row stride 320 B: four keys' 64-byte blocks on disjoint
//           banks), the B operands are hP and lP straight from the score registers — P' = 2^8 exp2(s - mref) <= 2^15 as two pieces — whose key
//           order (register r of lane half lh = key (r & 3) + 8 (r >> 2) + 4 lh) the transposed reads follow.  O[dim] = the hi column's row + the
//           lo column's row, both in one lane (registers 4a + e and 4a + 2 + e): added once after the key loop.
// Neither tile passes through the vector ALUs: global -> registers -> LDS as bytes, row addresses in the loads' scalar offset.  The frame mask is
// a 17th MFMA of the scores; the softmax reference follows the running maximum lazily (see the loop), so a tile costs ~130 vector instructions
// beside its 33 MFMAs (the first version, with per-lane row arithmetic, an FMA-added mask and eager rescaling: ~230).  The launch time did not
// move with that (82 us at T = 516): it is set by per-workgroup latencies and the third, nearly empty round of workgroups — profiles/r04_attn_h16_ablation.txt.
// Arithmetic: all piece products of 22-bit operands, fp32 accumulation — the class of the convolutions' setting 16.
// ---------------------------------------------------------------------------
typedef short ev_v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define AOH_KRS 272
#define AOH_VRS 320
#define AOH_WB (32 * AOH_KRS + 32 * AOH_VRS + 256)     // a wave's K tile and V tile (19200 bytes; four waves, two workgroups per CU: 150 KB)
__global__ __launch_bounds__(256, 2) void attn_out_h16_kernel(const AttnOutParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int h = wave & 1, kh = wave >> 1;
    int b, qt;
    {
        int id = blockIdx.x;
        const int ntw = p.ntail > 0 ? p.B : 0;         // the short last tiles first, on the fp32 4 x 4 blocks (attn_tail_path decodes the pairs)
        if (id < ntw) {
            if (p.stamps && tid == 0) p.stamps[6 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
            attn_tail_path<true>(p, p.xcd_map ? (id & 7) + 8 * (id >> 3) : id, smem, tid, lane, wave);
            if (p.stamps && tid == 0) { const unsigned long long t = __builtin_amdgcn_s_memrealtime(); for (int k = 1; k < 6; ++k) p.stamps[6 * blockIdx.x + k] = t; }
            return;
        }
        id -= ntw;
        if (p.xcd_map) { const int within = id >> 3; qt = within % p.nq; b = (id & 7) + 8 * (within / p.nq); }
        else { b = id / p.nq; qt = id - b * p.nq; }
    }
    const int q0 = qt * 32;
    const unsigned rowbase = (unsigned)b * p.S + p.P;
    auto stamp = [&](int k) { if (p.stamps && tid == 0) p.stamps[6 * blockIdx.x + k] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    char* Kt = (char*)smem + wave * AOH_WB;            // [32 keys][AOH_KRS]: the key rows of head h as they lie in the packed tensor
    char* Vt = Kt + 32 * AOH_KRS;                      // [32 keys][AOH_VRS]
    const __amdgpu_buffer_rsrc_t rQ = ev_rsrc(p.QKV), rM = ev_rsrc(p.rowmask), rW = ev_rsrc(p.Wouth);
    const float L2E = 1.44269504088896340736f;
    const unsigned ldb = (unsigned)p.ld * 4u;
    const int nkt = (p.T + 31) / 32, nh0 = (nkt + 1) / 2;
    const int kt0 = kh ? nh0 : 0, kt1 = kh ? nkt : nh0;
    // staging: float4 j of this lane = (key row 4j + lane/16, words 4 (lane % 16) ..): a load instruction covers 4 rows x 256 B.  The row of a
    // load is wave-uniform up to lane/16: it rides in the instruction's SCALAR offset, the lane's part is one register for the whole loop
    // (as per-lane row arithmetic this cost ~50 vector instructions per tile, a quarter of the loop's).
    f32x4 kr[8], vr[8];
    float mk = 0.f;
    const int srow = lane >> 4, sc4 = (lane & 15) * 4;
    const unsigned kcol = (unsigned)(128 + h * 64 + sc4) * 4u, vcol = (unsigned)(256 + h * 64 + sc4) * 4u;
    const unsigned kvo = (unsigned)srow * ldb + kcol, vvo = (unsigned)srow * ldb + vcol;
    auto kv_issue = [&](int kt) {
        if (kt * 32 + 32 <= p.T) {                      // (wave-uniform)
            const unsigned so = (rowbase + (unsigned)(kt * 32)) * ldb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                kr[j] = ev_bload4(rQ, kvo, so + (unsigned)(4 * j) * ldb);
                vr[j] = ev_bload4(rQ, vvo, so + (unsigned)(4 * j) * ldb);
            }
            mk = ev_bload1(rM, (unsigned)li * 4u, (rowbase + (unsigned)(kt * 32)) * 4u);
        } else {                                        // the utterance's last tile: keys beyond it re-read its last frame (and get no weight below)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tk = kt * 32 + 4 * j + srow;
                const unsigned ro = (rowbase + (unsigned)(tk < p.T ? tk : p.T - 1)) * ldb;
                kr[j] = ev_bload4(rQ, ro + kcol, 0);
                vr[j] = ev_bload4(rQ, ro + vcol, 0);
            }
            const int tm = kt * 32 + li;
            mk = ev_bload1(rM, (rowbase + (unsigned)(tm < p.T ? tm : p.T - 1)) * 4u, 0);
        }
    };
    if (kt0 < kt1) kv_issue(kt0);
    // Q operands: lane (query li, half lh), step g = dims 8g + 4lh .. + 3 = words (h01, l01, h23, l23) of the packed row
    f32x4 qb1[8], qb2[8];
    {
        const int tq = q0 + li;
        const unsigned off = (rowbase + (unsigned)(tq < p.T ? tq : 0)) * ldb + (unsigned)(h * 64 + 4 * lh) * 4u;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 w = ev_bload4(rQ, off + (unsigned)(8 * g) * 4u, 0);
            const f32x4 b1 = {w[0], w[0], w[2], w[2]}, b2 = {w[1], w[1], w[3], w[3]};
            qb1[g] = b1; qb2[g] = b2;
        }
    }
    // The additive frame mask (1 inside the utterance's length, 0 on its padded frames: a FLOAT mask, transformer.py:266-271) joins the scores as
    // one more MFMA: in accumulator units it is m 8 sq sk — an exact power of two, split by the host into mask_a mask_b, both fp16 numbers —
    // in k slot 0 of lane half 0: A = m[key] mask_a, B = mask_b.
    const float q_unit = p.scale * L2E * p.inv_sq * p.inv_sk;       // accumulator units -> log2-domain scores
    f32x4 mb4 = {0.f, 0.f, 0.f, 0.f};
    {
        const f16x2 mbp = {(_Float16)(lh == 0 ? p.mask_b : 0.f), (_Float16)0.f};
        mb4[0] = __builtin_bit_cast(float, mbp);
    }
    f32x16 o[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[m][r] = 0.f;
    // Online softmax with a LAZY reference: probabilities are taken against mref, which follows the running maximum only when that has grown by
    // more than 2^7 (a wave-uniform, rare branch: after the first tile of most rows never) — P' = 2^8 exp2(s - mref) <= 2^15 stays an fp16 number,
    // and the usual per-tile rescaling of O (64 registers) and of the row sums disappears from the loop.
    float mref = -1e30f, lrun = 0.f;
    // transposed reads: lane 4q + pp of the 16-lane group g16 supplies (key 4 (g16 >> 1) + q [+ 16 step + 8 e4], columns 16 (g16 & 1) + 4 pp .. [+ 32 mt])
    const __attribute__((address_space(3))) char* vtr = (const __attribute__((address_space(3))) char*)(Vt + (4 * (lane >> 5) + ((lane >> 2) & 3)) * AOH_VRS +
                                                                                                       2 * (16 * ((lane >> 4) & 1) + 4 * (lane & 3)));
    const char* krow = Kt + li * AOH_KRS + 16 * lh;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt == kt0 + 1) stamp(1);
        // publish the prefetched tile to this wave's LDS (the wave's LDS operations execute in order: no barrier, no other reader)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(f32x4*)(Kt + (4 * j + srow) * AOH_KRS + sc4 * 4) = kr[j];
            *(f32x4*)(Vt + (4 * j + srow) * AOH_VRS + sc4 * 4) = vr[j];
        }
        f32x4 ma4 = {0.f, 0.f, 0.f, 0.f};
        {
            const f16x2 map = {(_Float16)(lh == 0 ? mk * p.mask_a : 0.f), (_Float16)0.f};
            ma4[0] = __builtin_bit_cast(float, map);
        }
        kv_issue(kt + 1 < kt1 ? kt + 1 : kt);          // next tile (after the last: a harmless re-read)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ma4), __builtin_bit_cast(f16x8, mb4), s, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f16x8 a = __builtin_bit_cast(f16x8, *(const f32x4*)(krow + 32 * g));
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, qb1[g]), s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, qb2[g]), s, 0, 0, 0);
        }
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * q_unit;
        if (__any(mx > mref + 7.f)) {                  // (wave-uniform)
            const float mnew = mx > mref + 7.f ? mx : mref;
            const float alpha = __builtin_amdgcn_exp2f(mref - mnew);
            lrun *= alpha;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[m][r] *= alpha;
            mref = mnew;
        }
        const float nb = 8.f - mref;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], q_unit, nb));
        if (kt * 32 + 32 > p.T) {                       // (wave-uniform: the last tile) keys beyond the utterance get no weight
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh < p.T) ? s[r] : 0.f;
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) ps += s[r];
        lrun += ps;                                     // (this lane half's keys; the halves are added behind the loop)
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const _Float16 hh = (_Float16)s[r];
            ph[r >> 3][r & 7] = hh;
            pl[r >> 3][r & 7] = (_Float16)(s[r] - (float)hh);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const ev_v4s t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ev_v4s*)(vtr + (16 * st) * AOH_VRS + 64 * m));
                const ev_v4s t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ev_v4s*)(vtr + (16 * st + 8) * AOH_VRS + 64 * m));
                const s16x8 av = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                const f16x8 a = __builtin_bit_cast(f16x8, av);
                o[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ph[st], o[m], 0, 0, 0);
                o[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pl[st], o[m], 0, 0, 0);
            }
    }
    lrun += __shfl_xor(lrun, 32, 64);
    const float mrun = mref;
    stamp(2);
    // hi column row + lo column row: od[m][a] = dims 16 m + 4 a + 2 lh + (0, 1) of query li, un-normalised, in units of 2^13 sv
    float od[4][4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int a = 0; a < 4; ++a) { od[m][a][0] = o[m][4 * a] + o[m][4 * a + 2]; od[m][a][1] = o[m][4 * a + 1] + o[m][4 * a + 3]; }
    // Merge, projection and the epilogue's operand loads overlap: the lean epilogue requests its residual rows first and runs `after_issue` while
    // they fly (the hook conv_gemm_sk_kernel sums its partial tiles in) — here the merge of the key halves and the projection, ~4 us.
    f32x16 acc[2][1];
    auto merge_project = [&]() {
    // ---- first weight fragments of the projection (two fp16 pieces, conv_h16_kernel's fragment order: row tile 2 wave + a, 16-deep step ks, piece pc
    // at (((2 wave + a) 8 + ks) 2 + pc) KiB), into the staging registers: they land while the key halves are merged.  Ring slot of (a, ks, pc) =
    // kr / vr [4 (ks & 1) + 2 a + pc] for even / odd ks >> 1 ... i.e. steps 0-3 now, steps 4-7 as their slots come free.
    const unsigned wlane = (unsigned)lane * 16u;
    auto ldW = [&](int a, int ks, int pc) { return ev_bload4(rW, wlane, (unsigned)((((wave * 2 + a) * 8 + ks) * 2 + pc) * 1024)); };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) { kr[4 * ks + 2 * a + pc] = ldW(a, ks, pc); vr[4 * ks + 2 * a + pc] = ldW(a, ks + 2, pc); }
    // ---- merge the two key halves of each head through the (now dead) region of wave (h, 1); normalised rows — in units of sv: |O| <= max |v|,
    // so v's scale fits — as two fp16 planes into Os[query][hi 128 | lo 128] (row stride 528 B = AO_OLD words) inside wave 0's region
    char* Os = (char*)smem;
    float* Pw = (float*)Kt;                            // wave (h, 1): [8 float4 slots][64 lanes], then m and l
    ev_lds_barrier();                                  // every wave has left its key loop
    if (kh == 1) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                const f32x4 v = {od[m][2 * a2][0], od[m][2 * a2][1], od[m][2 * a2 + 1][0], od[m][2 * a2 + 1][1]};
                *(f32x4*)(Pw + ((m * 2 + a2) * 64 + lane) * 4) = v;
            }
        Pw[8 * 256 + lane] = mrun;
        Pw[8 * 256 + 64 + lane] = lrun;
    }
    ev_lds_barrier();
    if (kh == 0) {
        const float* Ps = (const float*)((const char*)smem + (wave + 2) * AOH_WB);       // region of wave (h, 1)
        const float m1 = Ps[8 * 256 + lane], l1 = Ps[8 * 256 + 64 + lane];
        const float mm = fmaxf(mrun, m1);
        const float a0 = __builtin_amdgcn_exp2f(mrun - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
        const float inv = 1.0f / (lrun * a0 + l1 * a1);           // (the 2^8 of P' cancels between O and l; sv stays: see above)
        const float w0 = a0 * inv, w1 = a1 * inv;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                const f32x4 pv = *(const f32x4*)(Ps + ((m * 2 + a2) * 64 + lane) * 4);
                const f32x4 x = {od[m][2 * a2][0] * w0 + pv[0] * w1, od[m][2 * a2][1] * w0 + pv[1] * w1,
                                 od[m][2 * a2 + 1][0] * w0 + pv[2] * w1, od[m][2 * a2 + 1][1] * w0 + pv[3] * w1};
                uint2 q0v, q1v;                         // (hi x0 x1 | hi x2 x3), (lo ...): dims d, d + 1 and d + 4, d + 5, d = 64 h + 16 m + 8 a2 + 2 lh
                evh_split4(x, q0v, q1v);
                char* dst = Os + li * (AO_OLD * 4) + 2 * (h * 64 + 16 * m + 8 * a2 + 2 * lh);
                *(unsigned*)(dst) = q0v.x; *(unsigned*)(dst + 8) = q0v.y;
                *(unsigned*)(dst + 256) = q1v.x; *(unsigned*)(dst + 256 + 8) = q1v.y;
            }
    }
    ev_lds_barrier();
    stamp(3);
    // ---- projection: Y^T[256][32] = Wout . O^T, K = 128 = 8 steps of 16, three piece products; accumulators in units of wo_scale sv,
    // bias preloaded (lean epilogue convention)
    const float u = p.wo_scale / p.inv_sv, inv_u = 1.0f / u;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (p.ep.bias) bq = *(const f32x4*)(p.ep.bias + wave * 64 + a * 32 + 8 * q + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][0][4 * q + e] = bq[e] * u;
        }
    const char* orow = Os + li * (AO_OLD * 4) + 16 * lh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f16x8 bh = __builtin_bit_cast(f16x8, *(const f32x4*)(orow + 32 * ks)), bl = __builtin_bit_cast(f16x8, *(const f32x4*)(orow + 256 + 32 * ks));
        f32x4 (&ring)[8] = (ks & 2) ? vr : kr;
        const int s4 = 4 * (ks & 1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const f16x8 ah = __builtin_bit_cast(f16x8, ring[s4 + 2 * a]), al = __builtin_bit_cast(f16x8, ring[s4 + 2 * a + 1]);
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[a][0], 0, 0, 0);
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[a][0], 0, 0, 0);
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[a][0], 0, 0, 0);
        }
        if (ks < 4) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) ring[s4 + 2 * a + pc] = ldW(a, ks + 4, pc);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][0][r] *= inv_u;
    stamp(4);
    };
    // ---- + residual rows, store (rows of THIS utterance only: a tile's tail rows may belong to the next one)
    conv_epilogue_lean<2, 1, 1>(p.ep, acc, smem + wave * (32 * 68), wave * 64, (int)rowbase + q0, lane, (int)rowbase, (int)rowbase + p.T, merge_project);
    stamp(5);
}

// attention_part_kernel + attention_merge_kernel: the small-launch build of attention_kernel.  A batch-1 decode has 6-10
// workgroups per attention launch, and each of their waves is a serial chain over all 9-17 key tiles: 64 MFMAs = 1.7 us of its
// SIMD's matrix pipe per tile, i.e. the launch is bound by the pipes of the handful of CUs it runs on (27-48 us, 60 launches
// per decode; splitting the key tiles over more waves of the SAME workgroup was measured to gain nothing).  Here the key
// tiles are split over KS workgroups (blockIdx.z = utterance * KS + part; part p takes tiles p, p + KS, ...), each writing its
// un-normalised online-softmax state (m, l, O) per query; a second small kernel merges the KS states in a fixed order
// (deterministic) and normalises.  Same products as attention_kernel; only the order in which the key tiles' contributions
// are combined differs.
// ---------------------------------------------------------------------------
struct AttnPartParams {
    AttnParams a;
    float* PO;      // [KS][rows][H*64] un-normalised partial outputs (row index as in O)
    float* PML;     // [KS][rows][H][2] running max (log2 domain) and sum
    int KS; int rows;
};

__global__ __launch_bounds__(256) void attention_part_kernel(const AttnPartParams pp) {
    const AttnParams& p = pp.a;
    __shared__ __attribute__((aligned(16))) float Ks[32 * ATT_LDK];
    __shared__ __attribute__((aligned(16))) float Vs[32 * ATT_LDK];
    __shared__ float Ms[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z / pp.KS, part = blockIdx.z % pp.KS, hd = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const size_t rowbase = (size_t)b * p.S + p.P;
    const int HD = p.H * 64;
    const float* Qp = p.QKV + hd * 64;
    const float* Kp = p.QKV + HD + hd * 64;
    const float* Vp = p.QKV + 2 * HD + hd * 64;
    f32x4 qf[8];
    {
        const int tq = q0 + li;
        const bool ok = tq < p.T;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4*)(Qp + (rowbase + tq) * p.ld + 8 * g + 4 * lh);
            qf[g] = v * (p.scale * 1.44269504088896340736f);
        }
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -1e30f, lrun = 0.f;
    const bool wave_active = __builtin_amdgcn_readfirstlane(q0) < p.T;
    const int nkt = (p.T + 31) / 32;
    const int sr = tid >> 3, sc = (tid & 7) * 8;
    f32x4 k0, k1, v0, v1;
    float mk;
    auto kv_load = [&](int kt) {
        const int tk = kt * 32 + sr;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        k0 = z; k1 = z; v0 = z; v1 = z;
        if (tk < p.T) {
            const float* kr = Kp + (rowbase + tk) * p.ld + sc;
            const float* vr = Vp + (rowbase + tk) * p.ld + sc;
            k0 = *(const f32x4*)kr; k1 = *(const f32x4*)(kr + 4);
            v0 = *(const f32x4*)vr; v1 = *(const f32x4*)(vr + 4);
        }
        const int tm = kt * 32 + (tid & 31);
        mk = (tm < p.T) ? p.rowmask[rowbase + tm] : -1e30f;
    };
    kv_load(part);
    for (int kt = part; kt < nkt; kt += pp.KS) {
        __syncthreads();
        *(f32x4*)(Ks + sr * ATT_LDK + sc) = k0; *(f32x4*)(Ks + sr * ATT_LDK + sc + 4) = k1;
        *(f32x4*)(Vs + sr * ATT_LDK + sc) = v0; *(f32x4*)(Vs + sr * ATT_LDK + sc + 4) = v1;
        if (tid < 32) Ms[tid] = mk * 1.44269504088896340736f;
        __syncthreads();
        if (kt + pp.KS < nkt) kv_load(kt + pp.KS);
        if (!wave_active) continue;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 a = *(const f32x4*)(Ks + li * ATT_LDK + 8 * g + 4 * lh);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[g][e], s, 0, 0, 0);
        }
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] += Ms[(r & 3) + 8 * (r >> 2) + 4 * lh];
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mnew); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        lrun = lrun * alpha + ps;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float a0 = Vs[key * ATT_LDK + li];
            const float a1 = Vs[key * ATT_LDK + 32 + li];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
        }
    }
    const int tq = q0 + li;
    if (tq < p.T) {   // un-normalised state of this part (a part without key tiles leaves m = -1e30, l = 0, O = 0)
        const size_t row = (size_t)part * pp.rows + rowbase + tq;
        float* orow = pp.PO + row * HD + hd * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a = {o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
            f32x4 c = {o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
            *(f32x4*)(orow + 8 * g + 4 * lh) = a;
            *(f32x4*)(orow + 32 + 8 * g + 4 * lh) = c;
        }
        if (lh == 0) { float* ml = pp.PML + (row * p.H + hd) * 2; ml[0] = mrun; ml[1] = lrun; }
    }
}

// one thread per (frame, head, 4 dims): O = sum_p w_p O_p / sum_p w_p l_p,  w_p = 2^(m_p - max_p m_p), parts in ascending order
__global__ __launch_bounds__(256) void attention_merge_kernel(const AttnPartParams pp) {
    const AttnParams& p = pp.a;
    const int HD = p.H * 64, per_row = HD / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = idx / per_row;
    const int c4 = (int)(idx % per_row) * 4, hd = c4 >> 6;
    if (n >= pp.rows) return;
    const int t = (int)(n % p.S) - p.P;
    if (t < 0 || t >= p.T) return;
    float mmax = -1e30f;
    for (int k = 0; k < pp.KS; ++k) mmax = fmaxf(mmax, pp.PML[(((size_t)k * pp.rows + n) * p.H + hd) * 2]);
    float l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < pp.KS; ++k) {
        const float* ml = pp.PML + (((size_t)k * pp.rows + n) * p.H + hd) * 2;
        const float w = __builtin_amdgcn_exp2f(ml[0] - mmax);
        l += ml[1] * w;
        const f32x4 v = *(const f32x4*)(pp.PO + ((size_t)k * pp.rows + n) * HD + c4);
        o += v * w;
    }
    const float inv = 1.0f / l;
    *(f32x4*)(p.O + (size_t)n * p.ldo + c4) = o * inv;
}

// ---------------------------------------------------------------------------
// Layout / elementwise helpers
// ---------------------------------------------------------------------------
// rowmask[n] = 1 if 0 <= t < ceil(len[b] / sub) (sub = 1: level-0 mask; sub = 2: mask[:, :, ::2]), else 0
// fp32 (rows, 384) q | k | v -> the packed fp16-pair form of ln_qkv_h16_kernel (qkv_pack), for callers that hold an fp32 tensor (ev_op_attn_out)
__global__ void qkv_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t npairs, float s0, float s1, float s2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const int c = (int)(i % 192) * 2;
    const float sc = c < 128 ? s0 : (c < 256 ? s1 : s2);
    const float c0 = src[2 * i] * sc, c1 = src[2 * i + 1] * sc;
    const _Float16 h0 = (_Float16)c0, h1 = (_Float16)c1;
    const f16x2 hh = {h0, h1}, ll = {(_Float16)(c0 - (float)h0), (_Float16)(c1 - (float)h1)};
    dst[2 * i] = __builtin_bit_cast(float, hh);
    dst[2 * i + 1] = __builtin_bit_cast(float, ll);
}
__global__ void rowmask_kernel(float* m, const int32_t* lengths, int nrows, int S, int P, int T, int sub) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nrows) return;
    int b = n / S, t = n % S - P;
    int len = lengths ? lengths[b] : (T * sub);
    int lim = (len + sub - 1) / sub;
    m[n] = (t >= 0 && t < T && t < lim) ? 1.f : 0.f;
}

// (B, C, T) channel-major -> frame-major rows (b*S + P + t), columns [c0, c0+C), optional row mask and scale
__global__ void cm_to_fm_kernel(const float* __restrict__ src, float* dst, int ld, int c0, int C, int T, int S, int P,
                                const float* rowmask, float scale) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, cb = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int c = cb + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int t = t0 + i, c = cb + tx;
        if (t < T && c < C) {
            size_t n = (size_t)b * S + P + t;
            float v = tile[tx][i] * scale;
            if (rowmask) v *= rowmask[n];
            dst[n * ld + c0 + c] = v;
        }
    }
}

// frame-major -> (B, C, T) channel-major with out = v*scale + shift
__global__ void fm_to_cm_kernel(const float* src, int ld, int c0, float* __restrict__ dst, int C, int T, int S, int P,
                                float scale, float shift) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, cb = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        int t = t0 + i, c = cb + tx;
        tile[i][tx] = (t < T && c < C) ? src[((size_t)b * S + P + t) * ld + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = cb + i, t = t0 + tx;
        if (c < C && t < T) dst[((size_t)b * C + c) * T + t] = tile[tx][i] * scale + shift;
    }
}

// broadcast a per-utterance vector (B, C) over all valid frames: dst[n][c0 + c] = v[b][c] * rowmask[n]
__global__ void bcast_rows_kernel(const float* v, float* dst, int ld, int c0, int C, int nrows, int S, int P, int T,
                                  const float* rowmask) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    int n = idx / C, c = idx % C;
    if (n >= nrows) return;
    int b = n / S, t = n % S - P;
    if (t < 0 || t >= T) return;
    dst[(size_t)n * ld + c0 + c] = v[b * C + c] * (rowmask ? rowmask[n] : 1.f);
}

// ---------------------------------------------------------------------------
// Text encoder (text_encoder.py) support kernels.  The encoder is <1 % of the path's FLOPs (north_star keeps it a
// once-per-utterance host stage); these kernels exist so that a streaming caller does not pay ~60 framework launches for
// it.  All of its convolutions run on conv_gemm_kernel; what is left is elementwise / per-frame work.
// ---------------------------------------------------------------------------
// token embedding * sqrt(C) (text_encoder.py:395) into frame-major rows: E = emb[id]*s (unmasked), Xm = E * mask
// An id outside [0, nvocab) raises IndexError in the reference's nn.Embedding; here the row is computed from a clamped id (no
// out-of-bounds read) and the event is reported through `bad` (a host-mapped word read by ev_text_encoder_status).
__global__ void enc_embed_kernel(const int64_t* ids, const int32_t* lengths, const float* emb, int nvocab, float scale, float* E, float* Xm,
                                 int C, int B, int Tx, int S, int P, int* bad) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C / 4;
    const int row = idx / c4n, c4 = (idx % c4n) * 4;
    if (row >= B * Tx) return;
    const int b = row / Tx, t = row % Tx;
    long id = ids[row];
    if (id < 0 || id >= nvocab) {
        if (bad && c4 == 0 && t < lengths[b]) *(volatile int*)bad = 1;
        id = id < 0 ? 0 : nvocab - 1;
    }
    f32x4 v = *(const f32x4*)(emb + (size_t)id * C + c4);
    v *= scale;
    const size_t n = (size_t)b * S + P + t;
    *(f32x4*)(E + n * C + c4) = v;
    if (t >= lengths[b]) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
    *(f32x4*)(Xm + n * C + c4) = v;
}

// channel-wise LayerNorm of text_encoder.py:15-33 (mean / biased variance over the channels of one frame, eps inside the
// rsqrt), optionally on x + R, optionally followed by ReLU, always multiplied by the frame mask (see ev_engine.hip: the
// encoder keeps every activation masked, which valid frames cannot observe).  One wavefront per frame, C = 64 * NPL.
struct CLNParams { const float* X; int ldx; const float* R; int ldr; const float* gamma; const float* beta; const float* rowmask;
                   float* Y; int ldy; int nrows, S, P, T; int relu; float eps; };
template <int NPL>
__global__ __launch_bounds__(256) void chan_layernorm_kernel(const CLNParams p) {
    constexpr int C = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.nrows) return;
    const int t = (n % p.S) - p.P;
    if (t < 0 || t >= p.T) return;
    float v[NPL], s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        v[i] = p.X[(size_t)n * p.ldx + lane + 64 * i];
        if (p.R) v[i] += p.R[(size_t)n * p.ldr + lane + 64 * i];
        s += v[i];
    }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] -= mean; q += v[i] * v[i]; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + p.eps);
    const float m = p.rowmask ? p.rowmask[n] : 1.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        float o = v[i] * rstd * p.gamma[lane + 64 * i] + p.beta[lane + 64 * i];
        if (p.relu) o = fmaxf(o, 0.f);
        p.Y[(size_t)n * p.ldy + lane + 64 * i] = o * m;
    }
}

// rotary embedding (text_encoder.py:97-172) on the first D features of every head of the q and k blocks of a
// [q | k | v] row, in place: pairs (j, j + D/2), angle = t * theta[j]  (theta = the reference's table, computed by the loader)
__global__ void enc_rope_kernel(float* QKV, int ld, const float* theta, int heads, int kc, int D, int nrows, int S, int P, int T) {
    const int half = D / 2;
    const int per_row = 2 * heads * half;          // (q|k) x heads x pairs
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = idx / per_row, r = idx % per_row;
    if (n >= nrows) return;
    const int t = (n % S) - P;
    if (t < 0 || t >= T) return;
    const int blk = r / (heads * half), hh = (r / half) % heads, j = r % half;
    float* x = QKV + (size_t)n * ld + blk * heads * kc + hh * kc;
    const float ang = (float)t * theta[j];
    const float c = cosf(ang), sn = sinf(ang);
    const float a = x[j], b = x[j + half];
    x[j] = a * c + (-b) * sn;
    x[j + half] = b * c + a * sn;
}

// MultiHeadAttention core of text_encoder.py:216-246, one query frame per wavefront, four per workgroup (kc <= 128 features per
// head, kc % 4 == 0).  As the reference orders it: scores = q.k / sqrt(kc) for all keys, softmax, then p.v.  Keys beyond the
// utterance length are masked_fill(-1e4) in the reference, i.e. they contribute exp(-1e4 - max) = 0 exactly to a valid query — they
// are skipped here; queries beyond the length give 0 (their output is masked by every consumer).
//   scores: the workgroup stages 64 keys x kc at a time in LDS and LANE j owns key j (a kc-long dot product per lane, q broadcast
//   from LDS) — no cross-lane reduction per key, which made the one-key-at-a-time form of round 1 a 160 us launch at batch 1;
//   p.v: lane d owns output features d and d + 64, the probabilities are broadcast from LDS, v rows are read coalesced from L2.
// dynamic LDS: (64 * (kc | 4) + 4 * kc + 4 * ceil64(Tx)) floats.
__global__ __launch_bounds__(256) void enc_attention_kernel(const float* QKV, int ld, const int32_t* lengths, float* O, int ldo,
                                                            int heads, int kc, int B, int Tx, int S, int P, float score_div) {
    extern __shared__ __attribute__((aligned(16))) float enc_sm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int LDK = kc | 4, Txp = (Tx + 63) & ~63;           // key row stride: 4 (mod 8) floats, conflict-free for ds_read_b128
    float* Ks = enc_sm;
    float* Qs = Ks + 64 * LDK + wave * kc;
    float* Pw = enc_sm + 64 * LDK + 4 * kc + wave * Txp;
    const int hh = blockIdx.y, b = blockIdx.z;
    const int tq = blockIdx.x * 4 + wave;
    const int len = lengths[b] < Tx ? lengths[b] : Tx;
    const size_t row0 = (size_t)b * S + P;
    const bool qok = tq < len;                               // (wave-uniform)
    for (int d = lane; d < kc; d += 64) Qs[d] = qok ? QKV[(row0 + tq) * ld + hh * kc + d] : 0.f;
    const float* Kb = QKV + row0 * ld + heads * kc + hh * kc;
    const float* Vb = QKV + row0 * ld + 2 * heads * kc + hh * kc;
    const int nkt = (len + 63) >> 6, f4 = kc >> 2;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                     // the previous tile is consumed (first pass: publishes Qs)
        for (int i = tid; i < 64 * f4; i += 256) {
            const int r = i / f4, c = (i % f4) * 4, j = kt * 64 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < len) v = *(const f32x4*)(Kb + (size_t)j * ld + c);
            *(f32x4*)(Ks + r * LDK + c) = v;
        }
        __syncthreads();
        if (qok) {
            const float* kr = Ks + lane * LDK;
            float s = 0.f;
            for (int d = 0; d < kc; d += 4) {
                const f32x4 k4 = *(const f32x4*)(kr + d), q4 = *(const f32x4*)(Qs + d);
                s += q4[0] * k4[0]; s += q4[1] * k4[1]; s += q4[2] * k4[2]; s += q4[3] * k4[3];
            }
            const int j = kt * 64 + lane;
            Pw[j] = j < len ? s / score_div : -INFINITY;
        }
    }
    float l = 1.f;
    if (qok) {                                               // softmax over the wave's own score row (lane j touches j, j + 64, ...)
        float m = -INFINITY;
        for (int j = lane; j < nkt * 64; j += 64) m = fmaxf(m, Pw[j]);
        m = wave_max(m);
        float ls = 0.f;
        for (int j = lane; j < nkt * 64; j += 64) { const float e = expf(Pw[j] - m); Pw[j] = e; ls += e; }
        l = wave_sum(ls);
    }
    __syncthreads();                                         // every lane's probabilities are visible to the whole wave
    if (tq >= Tx) return;
    float o0 = 0.f, o1 = 0.f;
    if (qok) {
        const bool a0 = lane < kc, a1 = lane + 64 < kc;
        const float* v0 = Vb + (a0 ? lane : 0);
        const float* v1 = Vb + (a1 ? lane + 64 : 0);
        int j = 0;
        for (; j + 4 <= len; j += 4) {
            const f32x4 pj = *(const f32x4*)(Pw + j);
            const float x0 = v0[(size_t)j * ld], x1 = v0[(size_t)(j + 1) * ld], x2 = v0[(size_t)(j + 2) * ld], x3 = v0[(size_t)(j + 3) * ld];
            const float y0 = v1[(size_t)j * ld], y1 = v1[(size_t)(j + 1) * ld], y2 = v1[(size_t)(j + 2) * ld], y3 = v1[(size_t)(j + 3) * ld];
            o0 += (pj[0] / l) * x0; o0 += (pj[1] / l) * x1; o0 += (pj[2] / l) * x2; o0 += (pj[3] / l) * x3;
            o1 += (pj[0] / l) * y0; o1 += (pj[1] / l) * y1; o1 += (pj[2] / l) * y2; o1 += (pj[3] / l) * y3;
        }
        for (; j < len; ++j) { const float pj = Pw[j] / l; o0 += pj * v0[(size_t)j * ld]; o1 += pj * v1[(size_t)j * ld]; }
    }
    float* orow = O + (row0 + tq) * ldo + hh * kc;
    if (lane < kc) orow[lane] = o0;
    if (lane + 64 < kc) orow[lane + 64] = o1;
}

// Hard monotonic alignment + expansion (utils/model.py:29-41 generate_path, matcha_tts.py:131-135): one workgroup per
// utterance.  cum = cumsum(w_ceil) is accumulated sequentially (in double, rounded per prefix: torch's CPU cumsum); path[i][ty] = [ty < cum[i]] -
// [ty < cum[i-1]] (float compare against arange, as sequence_mask does), masked by x_mask[i] * y_mask[ty]; mu_y[:, ty] =
// sum_i path[i][ty] * mu_x[:, i] — at most one non-zero term per output frame, so the matmul of the reference is a gather.
__global__ __launch_bounds__(256) void enc_align_kernel(const float* wceil, const float* mu_x, const int32_t* xlen, const int64_t* ylen,
                                                        float* mu_y, float* attn, int C, int Tx, int Tp) {
    extern __shared__ float cum[];                  // [Tx]
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        double acc = 0.0;                          // torch's CPU cumsum accumulates float32 in double and rounds each prefix
        for (int i = 0; i < Tx; ++i) { acc += (double)wceil[(size_t)b * Tx + i]; cum[i] = (float)acc; }
    }
    __syncthreads();
    const int lx = xlen[b];
    const long ly = ylen[b];
    for (int ty = tid; ty < Tp; ty += blockDim.x) {
        const float fy = (float)ty;
        int src = -1;                                // the token whose [cum[i-1], cum[i]) interval holds ty
        if (ty < ly) {
            int lo = 0, hi = Tx;                     // first i with fy < cum[i]
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (fy < cum[mid]) hi = mid; else lo = mid + 1; }
            if (lo < Tx && lo < lx) src = lo;
        }
        if (attn) for (int i = 0; i < Tx; ++i) attn[((size_t)b * Tx + i) * Tp + ty] = (i == src) ? 1.f : 0.f;
        for (int c = 0; c < C; ++c) mu_y[((size_t)b * C + c) * Tp + ty] = src >= 0 ? mu_x[((size_t)b * C + c) * Tx + src] : 0.f;
    }
}

// zero_pads_kernel: the pad rows (the convolutions' zero padding: rows [0, P) and [P + T, S) of every utterance) of up to 64
// frame-major tensors in one launch.  No kernel ever stores to a pad row, and every other row is written before it is read, so
// this — not a memset of the whole arena (0.3-0.7 ms per streaming utterance, each of which has a new length) — is all a change
// of geometry needs.
struct ZeroPadParams {
    float* p[64]; int C[64]; int lvl[64];
    int S[5], P[5], T[5];
    int B, n;
};
__global__ __launch_bounds__(256) void zero_pads_kernel(const ZeroPadParams zp) {
    const int k = blockIdx.y;
    if (k >= zp.n) return;
    const int l = zp.lvl[k], C4 = zp.C[k] >> 2, S = zp.S[l], P = zp.P[l], T = zp.T[l];
    const int padrows = S - T;                                  // P in front, S - P - T behind
    const long total = (long)zp.B * padrows * C4;
    f32x4* base = (f32x4*)zp.p[k];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const long r = i / C4;
        const int b = (int)(r / padrows), pr = (int)(r % padrows);
        const int row = pr < P ? pr : pr + T;
        base[((long)b * S + row) * C4 + c] = z;
    }
}

// zero fill as a KERNEL (a captured ev_cfm_decode re-zeroes its plan with this instead of a memset node: a captured call then consists of kernel
// nodes only)
__global__ __launch_bounds__(256) void zero_fill_kernel(f32x4* __restrict__ p, size_t n16) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}

// amax slots of a tensor that no producer bounded (operator-level tests, ev_op_conv1d): one workgroup per 128-row granule, the exact
// maximum of |x| over its rows (non-finite values mark the slot as the producers do)
__global__ __launch_bounds__(256) void amax_rows_kernel(const float* __restrict__ X, int ld, int C, int nrows, unsigned* slots) {
    __shared__ float red[4];
    const int g = blockIdx.x, r0 = g * 128;
    float m = 0.f;
    bool bad = false;
    const int c4n = C / 4;
    for (int i = threadIdx.x; i < 128 * c4n; i += 256) {
        const int r = r0 + i / c4n, c = (i % c4n) * 4;
        if (r >= nrows) break;
        const f32x4 v = *(const f32x4*)(X + (size_t)r * ld + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = fabsf(v[e]); if ((__float_as_uint(a) & 0x7f800000u) == 0x7f800000u) bad = bad || (a == a); else m = fmaxf(m, a); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const bool anybad = __syncthreads_or(bad);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) slots[g] = anybad ? 0x7f800000u : __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// ---------------------------------------------------------------------------
// Denoiser (hifigan/denoiser.py) support kernels.  The STFT / inverse STFT (n_fft 1024, hop 256, periodic Hann window,
// centred with reflect padding — torch.stft / torch.istft defaults as the reference calls them) are two convolutions on
// conv_gemm_kernel over the padded signal viewed as rows of 256 samples: frame f = rows f .. f+3 (4 taps, 256 "channels")
// against the windowed DFT basis; the inverse is the transposed basis with taps 0, -1, -2, -3 (overlap-add).
// ---------------------------------------------------------------------------
// (B, L) audio -> rows of 256 samples of the reflect-padded signal (pad 512 each side), frame-major padded layout
__global__ void dn_pad_reflect_kernel(const float* audio, float* rows, int B, int L, int S, int P) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Lp = L + 1024;
    if (idx >= (size_t)B * Lp) return;
    const int b = (int)(idx / Lp), m = (int)(idx % Lp);
    int i = m - 512;
    if (i < 0) i = -i;
    else if (i >= L) i = 2 * (L - 1) - i;
    rows[((size_t)b * S + P) * 256 + m] = audio[(size_t)b * L + i];
}

// spectral gain of denoiser.py:58-64 on [re(516) | im(516)] rows: mag' = clamp(|X| - bias*strength, 0), X' = mag' * exp(i*angle(X))
// = X * mag'/|X| (and mag' for |X| = 0: angle 0).  Rows that are not frames (f >= F, pads) are zeroed.  mag_out (optional): |X|.
__global__ void dn_gain_kernel(float* spec, const float* bias, float strength, float* mag_out, int nrows, int S, int P, int F) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nrows * 516) return;
    const int n = (int)(idx / 516), k = (int)(idx % 516);
    const int f = (n % S) - P;
    float* row = spec + (size_t)n * 1032;
    if (f < 0 || f >= F || k >= 513) { row[k] = 0.f; row[516 + k] = 0.f; return; }
    const float re = row[k], im = row[516 + k];
    const float mag = sqrtf(re * re + im * im);
    if (mag_out) mag_out[((size_t)(n / S) * 513 + k) * F + f] = mag;
    if (!bias) return;
    const float m2 = fmaxf(mag - bias[k] * strength, 0.f);
    if (mag > 0.f) { const float g = m2 / mag; row[k] = re * g; row[516 + k] = im * g; }
    else { row[k] = m2; row[516 + k] = 0.f; }
}

// overlap-added rows -> (B, L): drop the 512-sample centre padding and divide by the window envelope sum_f w^2[m - 256 f]
// over the frames f in [0, F) that cover padded sample m (torch.istft's normalisation)
__global__ void dn_crop_norm_kernel(const float* rows, const float* win2, float* out, int B, int L, int S, int P, int F) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * L) return;
    const int b = (int)(idx / L), i = (int)(idx % L);
    const int m = i + 512, r = m >> 8, c = m & 255;
    float env = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int f = r - j; if (f >= 0 && f < F) env += win2[256 * j + c]; }
    const float v = rows[((size_t)b * S + P + r) * 256 + c];
    out[idx] = env > 1e-11f ? v / env : v;
}

// conv_post (hifigan/models.py:195-196: Conv1d(C, 1, K, padding = K/2) + tanh) straight into the (B, T) waveform.  One
// output channel makes this a sliding dot product, not a GEMM: a 32-row MFMA tile would waste 31/32 of the matrix
// pipe and the launch was latency-bound at ~0.65 TB/s.  Here a workgroup stages 256 + K - 1 frames x C channels in LDS
// (coalesced 16-byte loads, row stride C + 4: conflict-free ds_read_b128 with one frame per lane) and each lane
// accumulates its frame's K*C products; weights are wave-uniform scalar loads.  HBM-bound: one read of the C-channel
// activations, one write of the waveform, pads stripped on the way out (no separate strip pass).
template <int C, int K>
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ X, const float* __restrict__ W /*[K][C]*/, float bias,
                                                        float* __restrict__ wav, int T, int S, int P) {
    constexpr int LD = C + 4, ROWS = 256 + K - 1, C4 = C / 4;
    __shared__ __attribute__((aligned(16))) float xs[ROWS * LD];
    const int tid = threadIdx.x;
    const int tiles = (T + 255) / 256;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * 256;
    const size_t row0 = (size_t)b * S + P + t0 - K / 2;          // P >= K/2: the pad rows are the conv's zero padding
    for (int i = tid; i < ROWS * C4; i += 256) {
        const int r = i / C4, c4 = (i % C4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t0 - K / 2 + r < T + P) v = *(const f32x4*)(X + (row0 + r) * C + c4);
        *(f32x4*)(xs + r * LD + c4) = v;
    }
    __syncthreads();
    float acc = bias;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const f32x4 v = *(const f32x4*)(xs + (tid + k) * LD + c4 * 4);
            const float* w = W + k * C + c4 * 4;
            acc = fmaf(v[0], w[0], acc); acc = fmaf(v[1], w[1], acc); acc = fmaf(v[2], w[2], acc); acc = fmaf(v[3], w[3], acc);
        }
    if (t0 + tid < T) wav[(size_t)b * T + t0 + tid] = tanhf(acc);
}

// frame-major (rows, C) -> flat waveform (B, T): C == 1 with pads stripped
__global__ void strip_pad_kernel(const float* src, float* dst, int T, int S, int P, size_t total) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    size_t b = idx / T, t = idx % T;
    dst[idx] = src[b * S + P + t];
}
