// Host engine + C ABI of the EmojiVoice hot path (see include/emojivoice.h).
// Weight re-layout, workspace management and the launch sequence of the CFM
// U-Net estimator (reference decoder.py:363-443) and the HiFi-GAN V1 generator
// (reference hifigan/models.py:181-197) on the kernels of ev_kernels.h.
#include "ev_kernels.h"
#include "../../include/emojivoice.h"

#include <math.h>
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct HostTensor {
    const float* p = nullptr;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};
typedef std::unordered_map<std::string, HostTensor> TensorMap;

struct Geom { int nrows, S, P, T; };  // flattened padded time axis of one resolution level

// One convolution / linear layer packed for conv_gemm_kernel.
struct ConvLayer {
    float* W = nullptr;       // [ntaps][Mpad][Kpad]
    unsigned short* Wh = nullptr; float wh_scale = 1.f;   // the weights times wh_scale (a power of two: the largest lands in [8192, 16384)) as two fp16
                                    // pieces in conv_h16_kernel's fragment order (Kpad % 16 == 0)
    unsigned short* Wx = nullptr;   // the same weights as three bf16 pieces in conv_split_kernel's fragment order (Kpad % 64 == 0 only)
    unsigned short* Wq = nullptr;   // the two fp16 pieces of Wh in v_mfma_f32_16x16x32_f16 fragment order (conv_h16_kernel<..., Q = 1>; Kpad % 64 == 0 only)
    float* bias = nullptr;    // [Cout] (stacked / phase-replicated as needed) or null
    int2* taplist[3] = {nullptr, nullptr, nullptr};   // for BM = 128, 64, 32: [mtiles][EV_MAX_TAPS] {tap, row offset} (one shared row when dense)
    int* nact[3] = {nullptr, nullptr, nullptr};       // per-tile active tap count (null when dense)
    unsigned char nact64[16] = {0};                   // host copy of the BM = 64 counts of the first 16 M tiles (unit weights of the balanced build)
    bool pair256_uniform = false;                     // Mpad % 256 == 0 and the two 128-channel tiles of every 256-channel tile carry the same tap list (conv_h16_bal_kernel<256, 128>)
    bool sparse_taps = false;
    bool tile128_exact = true;              // every 128-channel M tile carries exactly the taps of both its 64-channel halves (no union waste)
    int kstack_mt = 0, kstack_tap = 0;      // sparse_taps of the stacked [k-tap conv | 1x1 conv] kind: 32-channel tiles >= kstack_mt carry only tap kstack_tap
    int ntaps = 0, off[EV_MAX_TAPS] = {0};
    int halo_lo = 0, halo_hi = 0;
    int Cin = 0, Cout = 0, Mpad = 0, Kpad = 0;
    double macs_per_row = 0;  // algorithmic MACs per output view-row (reference arithmetic, no padding)
};

struct Epi {
    int act = ACT_NONE; float act_slope = 0.f; const float* act_a = nullptr; const float* act_b = nullptr;
    int mask1 = 0; float scale = 1.f; const float* R = nullptr; int ldr = 0; int accum = 0; int div3 = 0;
    int act2_lrelu = 0; float act2_slope = 0.f; int mask2 = 0; const float* rowmask = nullptr; int mmul = 1;
    float* Y2 = nullptr; int ldy2 = 0;
    float pro_slope = -1.f;                 // >= 0: prologue leaky-relu on the input
    int dbg = 0; int force_cfg = -1; int stagger = -1; unsigned long long* stamps = nullptr;
    float* gn_part = nullptr;               // ask for the per-tile GroupNorm statistics of the output (conv_sk32_kernel, one utterance); whether
                                            // they were produced is left in ev_handle::gn_stats_tiles (0 = no)
    int isplit_log2 = 31, isstride = 0;     // input column split (pair view of a strided slice)
    // amax slots (ConvParams::ymax / rmax / xmax, ev_kernels.h "Tile maxima from the producer"): ev_hifigan passes the slots of the tensors a launch
    // writes / adds / reads; a launch whose build does not leave bounds reports so in ev_handle::amax_emitted
    unsigned* ymax = nullptr; int ymax_mul = 1; const unsigned* rmax = nullptr; const unsigned* yold = nullptr; const unsigned* xmax = nullptr; int xmax_n = 0;
    int osplit_log2 = 31, osstride = 0;     // output column split
};

// c1r = block1's conv (k = 3) and res_conv (1x1, centre tap only) of one ResnetBlock1D stacked along the output axis:
// both read the same input, so one launch stages it once; the res half's M tiles carry a one-entry tap list
struct ResnetW { ConvLayer c1r, c2; float *g1, *b1, *g2, *b2; };
struct TransW {
    ConvLayer qkv, out, ff1, ff2; float *ln1g, *ln1b, *ln3g, *ln3b, *alpha, *binv;
    // powers of two for the packed q / k / v of attn_out_h16_kernel, from bounds no input can exceed (qkv_pack_scales); 0 = not available
    float qkv_scale[3] = {0.f, 0.f, 0.f};
};

struct EstimatorW {
    bool loaded = false;
    int in_ch = 0;  // 224 (or 160 single-speaker)
    ConvLayer t1, t2, tmlp;  // time MLP: linear_1 (+SiLU), linear_2 (+Mish), 6 stacked resnet mlps
    ResnetW rn[6];           // down0, down1, mid0, mid1, up0, up1
    // rn[0] consumes [x | mu | spk] (decoder.py:405-407): mu and spk do not change over the Euler steps, so their share of
    // block1's conv and of res_conv (incl. the biases) is computed once per decode and added as a residual to the x share
    ConvLayer rn0_c1r_x, rn0_c1r_ms;
    TransW tr[6];
    ConvLayer down0, down1, up0, up1, fin_conv, fin_proj;
    float *fin_g, *fin_b;
};

struct VocoderW {
    bool loaded = false;
    ConvLayer pre, post, ups[4];
    ConvLayer c1[12][3], c2[12][3];
    int ch[5];  // channels per level: 512, 256, 128, 64, 32
    float* post_w = nullptr; float post_b = 0.f; int post_k = 0;   // conv_post as [K][C] for conv_post_kernel
};

// Text encoder + duration predictor (text_encoder.py:328-410); hidden width C = n_channels + spk_emb_dim
struct EncLayerW { ConvLayer qkv, out, ff1, ff2; float *g1, *b1, *g2, *b2; };
struct TextEncW {
    bool loaded = false;
    int nvocab = 0, nch = 0, C = 0, heads = 0, nlayers = 0, ffc = 0;
    float* emb = nullptr; float* theta = nullptr;
    ConvLayer pre[3], pre_proj; float *pre_g[3], *pre_b[3];
    std::vector<EncLayerW> layers;
    ConvLayer proj_m, dp1, dp2, dp_proj; float *dp_g1, *dp_b1, *dp_g2, *dp_b2;
};

// Denoiser STFT pair (hifigan/denoiser.py): windowed DFT bases as two 4-tap convolutions over rows of 256 samples
struct DenoiserW { bool ready = false; ConvLayer fwd, inv; float* win2 = nullptr; };

}  // namespace

struct ev_handle {
    int device = 0;
    ev_model_dims dims;
    std::string err;
    std::vector<void*> owned;   // device allocations of the weights
    EstimatorW est;
    VocoderW voc;
    TextEncW enc;
    DenoiserW dn;
    // small per-stage scratch arenas (denoiser, text encoder): grown on demand, ordered against their last user's stream
    struct Scratch { char* p = nullptr; size_t bytes = 0; hipStream_t last = nullptr; bool last_valid = false; };
    Scratch dn_ws, enc_ws;
    float* zeros = nullptr;     // 4096 zero floats (stand-in bias for the fused kernels' unconditional loads)
    int max_steps = 64;         // Euler steps the time-grid buffers of the workspace are planned for (grows on demand)
    int* bad_ids_host = nullptr; int* bad_ids_dev = nullptr;   // mapped host word: count of out-of-range token ids seen by ev_text_encoder
    // workspace
    char* ws = nullptr; size_t ws_bytes = 0; size_t ws_used = 0; size_t ws_bytes_last = 0;
    int ws_B = -1, ws_Tp = -1, ws_Tv = -1;
    // profiling
    bool prof = false;
    bool fuse_pairs = true;     // EV_FUSE_PAIRS=0 disables resblock_pair_kernel (A/B runs)
    int fuse128 = 3;            // fuse C=128 pairs up to this kernel size (EV_FUSE128=0/3/7/11)
    bool fuse_attn = true;      // EV_FUSE_ATTN=0: attention and its output projection as separate launches (attention_kernel + a 1x1 conv)
    bool attn_h16 = true;       // EV_NO_ATTN_H16=1: the fused attention stays on the fp32 MFMA under arithmetic setting 16 too
    bool use_chain = true;      // ev_dbg_set_chain(h, 0) / EV_NO_CHAIN=1: ResBlock1 chains as three fused pairs instead of one launch
    bool fuse_mlp = true;       // EV_FUSE_MLP=0: LayerNorm / QKV / feed-forward of the transformer blocks as separate launches
    int gn_stats_tiles = 0;         // set by every launch_conv: row tiles whose GroupNorm statistics the launch left in Epi::gn_part (0 = none)
    int fuse_mlp_min_tiles = 96;    // EV_FUSE_MLP_MIN=<32-row tiles>: below this the separate (split-K) launches are used (measured with
                                    // tools/fuse_threshold.py at T = 516: batch 4 fused 13.4 / separate 12.1 ms, batch 8 15.1 / 15.2, batch 16 19.7 / 21.6)
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    hipStream_t ws_stream = nullptr; bool ws_stream_valid = false;   // stream of the last call that used the workspace
    float* temb_host[2] = {nullptr, nullptr}; size_t temb_cap[2] = {0, 0}; hipEvent_t temb_ev[2] = {nullptr, nullptr}; int temb_slot = 0;
    // Captured ev_cfm_decode calls stage their time embeddings in pinned memory that is NEVER recycled (a replay reads it whenever
    // the graph runs): EV_CAPTURE_SLOTS regions of one block allocated by ev_load_estimator, one consumed per captured call.
    float* cap_pool = nullptr; size_t cap_stride = 0; int cap_used = 0; int cap_nt[8] = {0};   // (cap_nt: the step count each region holds the time grid of)
    // The time MLP's output for a decode of n Euler steps (sinusoid -> linear_1 -> SiLU -> linear_2 -> Mish -> the six resnet projections:
    // n x 1536 floats) depends on n and the weights only — the time grid of solve_euler is torch.linspace(0, 1, n + 1) — so it is computed
    // once per step count and kept in device memory of the handle (at most 8 step counts, then the oldest goes).  Every later decode with
    // that n, eager or captured, reads it in place: four launches and one host-to-device copy fewer per decode, and — what made it
    // necessary — no copy node in a captured decode (see cfm_decode_impl).
    struct TprojEntry { int nt; float* d; };
    std::vector<TprojEntry> tproj_cache;
    bool captured = false; int cap_B = 0, cap_Tp = 0;   // a handle with a captured call stays bound to that (B, Tp): see ev_cfm_decode
    int64_t n_allocs = 0;       // device / pinned allocations made by the hot calls after loading (workspace and scratch growth, staging)
    double prof_flops = 0; int64_t prof_launches = 0;
    struct ProfRec { int kind, Cin, Cout, ntaps, nrows, cfg, lean; double flops; };
    std::vector<ProfRec> prof_recs;   // one per timed launch (EV_PROFILE_DUMP=<file> writes the per-shape table)
    hipStream_t stream = nullptr;
    // HiFi-GAN at small batch: the three ResBlock1 chains of an MRF level (kernel sizes 3 / 7 / 11, models.py:186-192) run on
    // three streams (the caller's + two of the handle's), ordered by events where they join the running sum
    hipStream_t mrf_stream[2] = {nullptr, nullptr};
    hipEvent_t mrf_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // balanced ("stream-K") launches, see SkCtl in ev_kernels.h: control words + flags (zeroed once) and the partial-tile slots
    int ncu = 0;                    // compute units of the device
    unsigned* sk_ctrl = nullptr;    // [0] epoch [1] arrivals [2] timed-out waits [3] shares taken over | flags from word 16 on | claims from word 16 + EV_SK_MAXWG on
    unsigned sk_seq = 0;            // sequence number of the balanced launches that use claims (conv_h16_bal_kernel, ln_mlp_h16_kernel)
    float* sk_part = nullptr;
    int split_terms = 16;           // arithmetic of the deep layers' products (ev_set_arithmetic; EV_SPLIT presets it): 16 = shipped: two block-scaled
                                    // fp16 pieces per operand, three products (fp32-grade); 6 = three bf16 pieces, six products (fp32-grade, no
                                    // range handling needed); 0 = fp32 MFMA everywhere; 3 = opt-in fast bf16 setting (not fp32-grade); 9 = A/B
    bool use_amax = true;           // EV_NO_AMAX=1 / ev_dbg_set_amax(h, 0): every fp16 tile pre-scans its input (A/B runs, tests of that path)
    bool amax_emitted = false;      // set by every launch_conv / launch_pair: the launch left per-granule bounds of its output in Epi::ymax
    int n_fp32_only_layers = 0;     // layers whose weights are not the exact sum of three bf16 pieces (tiny or non-finite): they keep the fp32 MFMA build
    int last_cfg = -1;              // build the last launch_conv / launch_pair took (ev_dbg_last_cfg: tests assert that a shape ran on the build they mean)
    bool sk_steal = true;           // EV_NO_SK_STEAL=1: owners wait for absent contributors (up to the spin limit) instead of taking their shares over (A/B)
    bool sk_balance = true;         // EV_NO_SK_BALANCE=1: every launch one tile per workgroup (A/B runs)
    bool sk_spread = false;         // EV_SK_SPREAD=1: launches of fewer row tiles than CUs (small batches) spread their units over up to 2 x CUs workgroups
    int sk_wgs = 2;                 // EV_SK_WGS=<1..3>: persistent workgroups per CU of a balanced ln_mlp launch (A/B runs)
    int sk_spin = 3000;             // EV_SK_SPIN=<polls> before an owner recomputes a contributor's share itself (~1.5 us per poll: at most ~4.5 ms)
    int mrf_max_frames = 16384;     // EV_MRF_STREAMS_MAX=<B*T mel frames>: calls up to this size use the three streams (0 = never).  Six more scratch
                                    // tensors per level; at batch 64 x 516 frames (21 GB) the two-stage pipeline of bench.py already fills the gaps:
                                    // -1.4 % on the vocoder alone, +0.6 % on the pipelined step
};

namespace {

int fail(ev_handle* h, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf;
    return 1;
}
#define HIPCHK(h, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(h, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

template <typename T>
int dev_upload(ev_handle* h, const std::vector<T>& v, T** out) {
    void* d = nullptr;
    HIPCHK(h, hipMalloc(&d, v.size() * sizeof(T) + 16));
    HIPCHK(h, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    h->owned.push_back(d);
    *out = (T*)d;
    return 0;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Make a scratch arena at least `need` bytes for a call on h->stream.  No device-wide synchronisation: only the stream of
// the arena's previous user (if it differs) and, when the arena must be replaced, the current stream are waited for, so a
// second handle working on another stream (BatchPipeline) is never stalled.
int scratch_acquire(ev_handle* h, ev_handle::Scratch& s, size_t need) {
    if (s.last_valid && s.last != h->stream) HIPCHK(h, hipStreamSynchronize(s.last));
    if (need > s.bytes) {
        if (s.last_valid) HIPCHK(h, hipStreamSynchronize(h->stream));
        if (s.p) HIPCHK(h, hipFree(s.p));
        const size_t want = std::max(need, s.bytes + s.bytes / 2);      // geometric growth: a stream of ever longer utterances
        s.p = nullptr; s.bytes = 0;                                     // re-allocates O(log) times, not once per new maximum
        if (hipMalloc((void**)&s.p, want) == hipSuccess) s.bytes = want;
        else { (void)hipGetLastError(); HIPCHK(h, hipMalloc((void**)&s.p, need)); s.bytes = need; }
        ++h->n_allocs;
    }
    s.last = h->stream; s.last_valid = true;
    return 0;
}

// finalize a layer whose packed host weights Wh [ntaps][Mpad][Kpad] and bias are given
int finish_layer(ev_handle* h, ConvLayer& L, const std::vector<float>& Wh, const std::vector<float>* bias) {
    int lo = 0, hi = 0;
    for (int i = 0; i < L.ntaps; ++i) { if (-L.off[i] > lo) lo = -L.off[i]; if (L.off[i] > hi) hi = L.off[i]; }
    L.halo_lo = lo; L.halo_hi = hi;
    if (lo + hi > EV_HALO) return fail(h, "conv halo %d exceeds EV_HALO", lo + hi);
    {   // re-order [tap][Mpad][Kpad] into MFMA-fragment order [tap][Mpad/32][Kpad/8][lane][4] (see conv_gemm_kernel)
        std::vector<float> Wf(Wh.size());
        const int MT32 = L.Mpad / 32, KG8 = L.Kpad / 8;
        for (int tap = 0; tap < L.ntaps; ++tap)
            for (int mt = 0; mt < MT32; ++mt)
                for (int kg = 0; kg < KG8; ++kg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int row = mt * 32 + (lane & 31), k = kg * 8 + 4 * (lane >> 5) + e;
                            Wf[((((size_t)tap * MT32 + mt) * KG8 + kg) * 64 + lane) * 4 + e] = Wh[((size_t)tap * L.Mpad + row) * L.Kpad + k];
                        }
        if (dev_upload(h, Wf, &L.W)) return 1;
    }
    if (L.Kpad % 16 == 0) {   // conv_h16_kernel: (w * 2^k) = h0 + h1 to 2^-22, two fp16 pieces; order [tap][Mpad/32][Kpad/16][piece][lane][8]
        float mx = 0.f;
        for (float w : Wh) if (std::isfinite(w)) mx = std::max(mx, std::fabs(w));
        int k = 0;
        if (mx > 0.f) { int ex; std::frexp(mx, &ex); k = 14 - ex; }          // mx = m * 2^ex, m in [0.5, 1): mx * 2^(14 - ex) in [8192, 16384)
        k = std::max(-40, std::min(40, k));                                  // (the product with an activation scale of up to 2^40 must stay far from the fp32 range)
        L.wh_scale = std::ldexp(1.0f, k);
        const int MT32 = L.Mpad / 32, KG16 = L.Kpad / 16;
        std::vector<unsigned short> Whp(Wh.size() * 2);
        auto hbits = [](_Float16 v) { unsigned short u; memcpy(&u, &v, 2); return u; };
        for (int tap = 0; tap < L.ntaps; ++tap)
            for (int mt = 0; mt < MT32; ++mt)
                for (int kg = 0; kg < KG16; ++kg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int row = mt * 32 + (lane & 31), kk = kg * 16 + 8 * (lane >> 5) + e;
                            const float w = Wh[((size_t)tap * L.Mpad + row) * L.Kpad + kk] * L.wh_scale;
                            const _Float16 h0 = (_Float16)w, h1 = (_Float16)(w - (float)h0);
                            const size_t base = ((((size_t)tap * MT32 + mt) * KG16 + kg) * 2) * 512 + (size_t)lane * 8 + e;
                            Whp[base] = hbits(h0); Whp[base + 512] = hbits(h1);
                        }
        if (dev_upload(h, Whp, &L.Wh)) return 1;
        if (L.Kpad % 32 == 0) {   // ... and in the 16 x 16 x 32 fragment order [tap][Mpad/16][Kpad/32][piece][lane][8]: lane = (row & 15) + 16 kgroup, k = 32 kg32 + 8 kgroup + e
            const int MT16 = L.Mpad / 16, KG32 = L.Kpad / 32;
            std::vector<unsigned short> Wqp(Wh.size() * 2);
            for (int tap = 0; tap < L.ntaps; ++tap)
                for (int mt = 0; mt < MT16; ++mt)
                    for (int kg = 0; kg < KG32; ++kg)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int row = mt * 16 + (lane & 15), kk = kg * 32 + 8 * (lane >> 4) + e;
                                const float w = Wh[((size_t)tap * L.Mpad + row) * L.Kpad + kk] * L.wh_scale;
                                const _Float16 h0 = (_Float16)w, h1 = (_Float16)(w - (float)h0);
                                const size_t base = ((((size_t)tap * MT16 + mt) * KG32 + kg) * 2) * 512 + (size_t)lane * 8 + e;
                                Wqp[base] = hbits(h0); Wqp[base + 512] = hbits(h1);
                            }
            if (dev_upload(h, Wqp, &L.Wq)) return 1;
        }
    }
    if (L.Kpad % 16 == 0) {   // conv_split_kernel / resblock_pair_split_kernel: w = w0 + w1 + w2 exactly, each piece the upper half of an fp32 word (bf16);
        // order [tap][Mpad/32][Kpad/16][piece][lane][8], lane = (row & 31) + 32 * half, element e <-> k = 16 kg + 8 half + e
        const int MT32 = L.Mpad / 32, KG16 = L.Kpad / 16;
        std::vector<unsigned short> Wx(Wh.size() * 3);
        auto bits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
        auto fl = [](unsigned u) { float f; memcpy(&f, &u, 4); return f; };
        // A weight whose residuals reach the subnormal range (|w| below ~1e-33) or that is not finite is NOT the exact sum of three truncated
        // bf16 pieces.  Such a checkpoint still loads (the reference loads it): the layer keeps no piece planes (Wx stays null), every gate of
        // the 16-bit-pipe builds tests Wx, so the layer runs on the exact fp32 MFMA build whatever the arithmetic setting.
        bool exact = true;
        for (int tap = 0; tap < L.ntaps && exact; ++tap)
            for (int mt = 0; mt < MT32; ++mt)
                for (int kg = 0; kg < KG16; ++kg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int row = mt * 32 + (lane & 31), k = kg * 16 + 8 * (lane >> 5) + e;
                            const float w = Wh[((size_t)tap * L.Mpad + row) * L.Kpad + k];
                            const unsigned p0 = bits(w) & 0xffff0000u;
                            const float r1 = w - fl(p0);
                            const unsigned p1 = bits(r1) & 0xffff0000u;
                            const unsigned p2 = bits(r1 - fl(p1)) & 0xffff0000u;
                            if (!(fl(p0) + fl(p1) + fl(p2) == w)) exact = false;
                            const size_t base = ((((size_t)tap * MT32 + mt) * KG16 + kg) * 3) * 512 + (size_t)lane * 8 + e;
                            Wx[base] = (unsigned short)(p0 >> 16); Wx[base + 512] = (unsigned short)(p1 >> 16); Wx[base + 1024] = (unsigned short)(p2 >> 16);
                        }
        if (exact) { if (dev_upload(h, Wx, &L.Wx)) return 1; }
        else ++h->n_fp32_only_layers;
    }
    if (bias) { if (dev_upload(h, *bias, &L.bias)) return 1; }
    // per-tile compact lists of non-zero taps (polyphase transposed convs have all-zero (phase, tap) slabs)
    const int bms[3] = {128, 64, 32};
    const int plane_bytes = (L.Mpad / 32) * (L.Kpad / 8) * 1024;   // tap-list entries carry the byte offset of the tap's weight plane
    std::vector<std::vector<int2>> lists[3];
    L.sparse_taps = false;
    for (int k = 0; k < 3; ++k) {
        const int BM = bms[k];
        const int mt = L.Mpad / BM;
        lists[k].resize(mt);
        for (int t = 0; t < mt; ++t)
            for (int tap = 0; tap < L.ntaps; ++tap) {
                bool nz = false;
                for (int m = t * BM; m < (t + 1) * BM && !nz; ++m) {
                    const float* row = &Wh[((size_t)tap * L.Mpad + m) * L.Kpad];
                    for (int c = 0; c < L.Kpad; ++c) if (row[c] != 0.f) { nz = true; break; }
                }
                if (nz) lists[k][t].push_back(make_int2(tap * plane_bytes, L.off[tap]));
                else if (t * BM < L.Cout) L.sparse_taps = true;
            }
    }
    L.pair256_uniform = L.Mpad % 256 == 0;
    for (int t = 0; t + 1 < L.Mpad / 128 && L.pair256_uniform; t += 2) {
        const auto &a = lists[0][t], &b = lists[0][t + 1];
        L.pair256_uniform = a.size() == b.size() && !a.empty();
        for (size_t i = 0; i < a.size() && L.pair256_uniform; ++i) L.pair256_uniform = a[i].x == b[i].x && a[i].y == b[i].y;
    }
    L.tile128_exact = true;
    for (int t = 0; t < L.Mpad / 128 && t * 128 < L.Cout; ++t)
        for (int hh = 0; hh < 2; ++hh)
            if ((2 * t + hh) * 64 < L.Cout && lists[1][2 * t + hh].size() != lists[0][t].size()) L.tile128_exact = false;
    if (L.sparse_taps) {   // the stacked pattern on the 32-channel tiling: full tiles first, then tiles with one and the same tap
        const int mt = (L.Cout + 31) / 32;
        int split = 0;
        while (split < mt && (int)lists[2][split].size() == L.ntaps) ++split;
        bool ok = split > 0 && split < mt;
        for (int t = split; t < mt && ok; ++t) ok = lists[2][t].size() == 1 && lists[2][t][0].x == lists[2][split][0].x;
        if (ok) { L.kstack_mt = split; L.kstack_tap = lists[2][split][0].x / plane_bytes; }
    }
    for (int k = 0; k < 3; ++k) {
        const int mt = L.Mpad / bms[k];
        if (!L.sparse_taps) {   // dense: one shared row holding every tap
            std::vector<int2> row(EV_MAX_TAPS, make_int2(0, 0));
            for (int tap = 0; tap < L.ntaps; ++tap) row[tap] = make_int2(tap * plane_bytes, L.off[tap]);
            if (dev_upload(h, row, &L.taplist[k])) return 1;
            L.nact[k] = nullptr;
        } else {
            std::vector<int2> tab((size_t)mt * EV_MAX_TAPS, make_int2(0, 0));
            std::vector<int> cnt(mt, 0);
            for (int t = 0; t < mt; ++t) {
                cnt[t] = (int)lists[k][t].size();
                for (size_t i = 0; i < lists[k][t].size(); ++i) tab[(size_t)t * EV_MAX_TAPS + i] = lists[k][t][i];
            }
            if (dev_upload(h, tab, &L.taplist[k])) return 1;
            if (dev_upload(h, cnt, &L.nact[k])) return 1;
            if (k == 1) for (int t = 0; t < mt && t < 16; ++t) L.nact64[t] = (unsigned char)std::min(cnt[t], 15);
        }
    }
    return 0;
}

// Conv1d weight (Cout, Cin, K), stride 1, "same" padding (K*d - d)/2, dilation d
int pack_conv(ev_handle* h, ConvLayer& L, const HostTensor& w, const HostTensor* b, int dilation) {
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], K = w.ndim == 3 ? (int)w.shape[2] : 1;
    if (K > EV_MAX_TAPS) return fail(h, "kernel size %d > %d", K, EV_MAX_TAPS);
    L.Cin = Cin; L.Cout = Cout; L.ntaps = K; L.Mpad = round_up(Cout, 128); L.Kpad = round_up(Cin, EV_BK);
    const int pad = (K * dilation - dilation) / 2;
    for (int k = 0; k < K; ++k) L.off[k] = k * dilation - pad;
    std::vector<float> Wh((size_t)K * L.Mpad * L.Kpad, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < K; ++k) Wh[((size_t)k * L.Mpad + co) * L.Kpad + ci] = w.p[((size_t)co * Cin + ci) * K + k];
    L.macs_per_row = (double)Cout * Cin * K;
    std::vector<float> bh;
    if (b) bh.assign(b->p, b->p + Cout);
    return finish_layer(h, L, Wh, b ? &bh : nullptr);
}

// Several (Cout_i, Cin) linears sharing one input stacked along the output axis
int pack_linear_stack(ev_handle* h, ConvLayer& L, const std::vector<const HostTensor*>& ws, const std::vector<const HostTensor*>& bs) {
    const int Cin = (int)ws[0]->shape[1];
    int Cout = 0;
    for (auto* w : ws) Cout += (int)w->shape[0];
    L.Cin = Cin; L.Cout = Cout; L.ntaps = 1; L.off[0] = 0; L.Mpad = round_up(Cout, 128); L.Kpad = round_up(Cin, EV_BK);
    std::vector<float> Wh((size_t)L.Mpad * L.Kpad, 0.f), bh((size_t)Cout, 0.f);
    int r0 = 0;
    for (size_t i = 0; i < ws.size(); ++i) {
        const int co_n = (int)ws[i]->shape[0];
        for (int co = 0; co < co_n; ++co) {
            for (int ci = 0; ci < Cin; ++ci) Wh[(size_t)(r0 + co) * L.Kpad + ci] = ws[i]->p[(size_t)co * Cin + ci];
            if (!bs.empty() && bs[i]) bh[r0 + co] = bs[i]->p[co];
        }
        r0 += co_n;
    }
    L.macs_per_row = (double)Cout * Cin;
    return finish_layer(h, L, Wh, bs.empty() ? nullptr : &bh);
}

// ConvTranspose1d weight (Cin, Cout, K), stride s, padding p as a polyphase conv over INPUT frames:
// view-row q holds the s output frames s*q .. s*q+s-1, column r*Cout + co;
//   out[s*q + r][co] = sum_{k == (r+p) mod s} W[ci][co][k] * x[q + (r + p - k)/s][ci]
int pack_convT(ev_handle* h, ConvLayer& L, const HostTensor& w, const HostTensor* b, int s, int p) {
    const int Cin = (int)w.shape[0], Cout = (int)w.shape[1], K = (int)w.shape[2];
    std::vector<int> offs;
    for (int r = 0; r < s; ++r)
        for (int k = 0; k < K; ++k)
            if ((r + p - k) % s == 0) {
                int o = (r + p - k) / s;
                bool found = false;
                for (int x : offs) if (x == o) found = true;
                if (!found) offs.push_back(o);
            }
    for (size_t i = 0; i < offs.size(); ++i) for (size_t j = i + 1; j < offs.size(); ++j) if (offs[j] < offs[i]) std::swap(offs[i], offs[j]);
    if ((int)offs.size() > EV_MAX_TAPS) return fail(h, "too many polyphase taps");
    L.Cin = Cin; L.Cout = s * Cout; L.ntaps = (int)offs.size(); L.Mpad = round_up(L.Cout, 128); L.Kpad = round_up(Cin, EV_BK);
    for (int i = 0; i < L.ntaps; ++i) L.off[i] = offs[i];
    std::vector<float> Wh((size_t)L.ntaps * L.Mpad * L.Kpad, 0.f), bh((size_t)L.Cout, 0.f);
    for (int r = 0; r < s; ++r)
        for (int k = 0; k < K; ++k) {
            if ((r + p - k) % s != 0) continue;
            const int o = (r + p - k) / s;
            int ti = 0;
            while (offs[ti] != o) ++ti;
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    Wh[((size_t)ti * L.Mpad + r * Cout + co) * L.Kpad + ci] = w.p[((size_t)ci * Cout + co) * K + k];
        }
    if (b) for (int r = 0; r < s; ++r) for (int co = 0; co < Cout; ++co) bh[r * Cout + co] = b->p[co];
    L.macs_per_row = (double)Cin * Cout * K;  // per input frame: every (ci, co, k) product happens exactly once
    return finish_layer(h, L, Wh, b ? &bh : nullptr);
}

// Conv1d (Cout, Cin, 3) stride 2 padding 1 over the PAIR view of its input (row t' = frames 2t', 2t'+1; 2*Cin columns):
//   out[t'] = W0 x[2t'-1] + W1 x[2t'] + W2 x[2t'+1]
int pack_conv_stride2(ev_handle* h, ConvLayer& L, const HostTensor& w, const HostTensor* b) {
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1], K = (int)w.shape[2];
    if (K != 3) return fail(h, "stride-2 conv expects k=3");
    L.Cin = 2 * Cin; L.Cout = Cout; L.ntaps = 2; L.off[0] = -1; L.off[1] = 0; L.Mpad = round_up(Cout, 128); L.Kpad = round_up(2 * Cin, EV_BK);
    std::vector<float> Wh((size_t)2 * L.Mpad * L.Kpad, 0.f), bh;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float* wk = &w.p[((size_t)co * Cin + ci) * 3];
            Wh[((size_t)0 * L.Mpad + co) * L.Kpad + Cin + ci] = wk[0];  // frame 2t'-1 = second half of pair t'-1
            Wh[((size_t)1 * L.Mpad + co) * L.Kpad + ci] = wk[1];        // frame 2t'
            Wh[((size_t)1 * L.Mpad + co) * L.Kpad + Cin + ci] = wk[2];  // frame 2t'+1
        }
    if (b) bh.assign(b->p, b->p + Cout);
    L.macs_per_row = (double)Cout * Cin * 3;
    return finish_layer(h, L, Wh, b ? &bh : nullptr);
}

const HostTensor* find(ev_handle* h, const TensorMap& m, const std::string& k) {
    auto it = m.find(k);
    if (it == m.end()) { fail(h, "missing tensor '%s'", k.c_str()); return nullptr; }
    return &it->second;
}

int upload_vec(ev_handle* h, const TensorMap& m, const std::string& k, float** out) {
    const HostTensor* t = find(h, m, k);
    if (!t) return 1;
    std::vector<float> v(t->p, t->p + t->numel());
    return dev_upload(h, v, out);
}

// ---------------------------------------------------------------------------
// launching
// ---------------------------------------------------------------------------
// Per-launch host-side state (no file-scope mutable state: different handles may be driven from different host threads)
struct LaunchOpts {
    int halo = EV_HALO;      // halo rows of the layer being launched (LDS is sized for BN + halo, not BN + EV_HALO)
    int kb = 1;              // k-chunks per stage (see conv_gemm_kernel: KB)
    int wgs_per_cu = 0;      // tools/conv_bench.py: cap workgroups per CU by over-allocating LDS (0 = off)
    int device = 0;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is sticky per (function, device): issue it only when a launch needs more
// than any earlier launch of THAT kernel asked for (it costs host time on the ~640-launch batch-1 chain).  The kernel is a
// non-type template parameter, so every instantiation has its own high-water marks (all conv instantiations share one
// function-pointer TYPE: a type-keyed cache would let one instantiation's grant hide another's request).
template <auto Kernel>
inline void ensure_dyn_smem(size_t smem, int device) {
    static std::atomic<int> granted[16];
    static std::mutex slow[16];           // the attribute is sticky and NOT monotonic: two threads raising it must not interleave
    const int d = device & 15;
    if (smem <= 65536 || (int)smem <= granted[d].load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(slow[d]);
    const int cur = granted[d].load(std::memory_order_relaxed);
    if ((int)smem <= cur) return;
    if (hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess)
        granted[d].store((int)smem, std::memory_order_release);
    else (void)hipGetLastError();         // (the launch that follows reports the failure through hipGetLastError)
}

template <int BM, int BN, int WM, int WN, bool PF, bool FULL, int LEAN, int KB = 1>
void launch_cfg2(const ConvParams& p, hipStream_t st, const LaunchOpts& lo);

// Dispatch on the epilogue flavour:
//   lean    : y = act(acc + bias) [+ R], act in {none, lrelu, SnakeBeta}, plain row-major Y/R, Cout % 4 == 0
//             (instruction-lean, bias preloaded into acc)
//   compact : any flag combination without a transcendental activation (code stays I-cache resident)
//   full    : tanh / SiLU / Mish / SnakeBeta
inline bool lean_acc(const ConvParams& p) { return p.accum || p.div3 || p.act2_lrelu; }
inline bool lean_ok(const ConvParams& p) {
    if (lean_acc(p) && (!p.R || p.act != ACT_NONE)) return false;
    return (p.act == ACT_NONE || p.act == ACT_LRELU || (p.act == ACT_SNAKE && (((size_t)p.act_a | (size_t)p.act_b) & 15) == 0)) &&
           ((!p.mask1 && !p.mask2) || (p.rowmask && p.mmul >= 1 && !(p.mask2 && lean_acc(p)))) && p.scale == 1.f &&
           !p.Y2 && (p.osplit_log2 >= 31 || (p.osplit_log2 >= 2 && (p.osstride & 3) == 0)) && (p.Cout & 3) == 0 && (p.ldy & 3) == 0 && (!p.R || (p.ldr & 3) == 0) &&
           (!p.bias || ((size_t)p.bias & 15) == 0) && !(p.dbg & 4) && p.S >= 4;   // (lean row walk: two wraps per 8-row pass)
}
template <int BM, int BN, int WM, int WN, bool PF = false>
void launch_cfg(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    static const bool no_lean = getenv("EV_NO_LEAN") != nullptr;
    const bool kb2 = !PF && lo.kb == 2;
    if (!no_lean && lean_ok(p)) {
        if (p.act == ACT_SNAKE) { if (kb2) launch_cfg2<BM, BN, WM, WN, false, false, 2, 2>(p, st, lo); else launch_cfg2<BM, BN, WM, WN, PF, false, 2>(p, st, lo); }
        else if (lean_acc(p)) { if (kb2) launch_cfg2<BM, BN, WM, WN, false, false, 3, 2>(p, st, lo); else launch_cfg2<BM, BN, WM, WN, PF, false, 3>(p, st, lo); }
        else { if (kb2) launch_cfg2<BM, BN, WM, WN, false, false, 1, 2>(p, st, lo); else launch_cfg2<BM, BN, WM, WN, PF, false, 1>(p, st, lo); }
    } else if (p.act == ACT_NONE || p.act == ACT_LRELU) {
        if (kb2) launch_cfg2<BM, BN, WM, WN, false, false, 0, 2>(p, st, lo); else launch_cfg2<BM, BN, WM, WN, PF, false, 0>(p, st, lo);
    } else launch_cfg2<BM, BN, WM, WN, PF, true, 0>(p, st, lo);
}

template <int BM, int BN, int WM, int WN, bool PF, bool FULL, int LEAN, int KB>
void launch_cfg2(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    // LDS holds the X tile during the K loop and, afterwards, one transposed 32-frame slab per wave for the epilogue
    const size_t xs = (size_t)(BN + ((lo.halo + 7) & ~7)) * (32 * KB + 4);
    constexpr size_t es = (size_t)4 * 32 * (BM / WM + 4);
    size_t smem = (xs > es ? xs : es) * sizeof(float);
    if (lo.wgs_per_cu > 0) { size_t cap = (size_t)(160 * 1024 / lo.wgs_per_cu) & ~(size_t)255; if (cap > smem) smem = cap; }
    ensure_dyn_smem<conv_gemm_kernel<BM, BN, WM, WN, PF, FULL, LEAN, KB>>(smem, lo.device);
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, PF, FULL, LEAN, KB>), dim3(p.mtiles * p.ntiles), dim3(256), smem, st, p);
}

// conv_split_kernel (fp32 contraction as six bf16 products per element pair): the lean epilogues only
template <int BM, int BN, int WM, int WN, int TERMS = 6>
void launch_split(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    const size_t xs = (size_t)(BN + ((lo.halo + 7) & ~7)) * EVX_RSB;
    constexpr size_t es = (size_t)4 * 32 * (BM / WM + 4) * sizeof(float);
    size_t smem = xs > es ? xs : es;
    {   // A/B: EV_SPLIT_WPC=<n> caps the deep-grid split builds at n workgroups per CU by padding the LDS request, which leaves the rest of
        // the CU (LDS and registers) to the kernels of another stream
        static const int wpc = getenv("EV_SPLIT_WPC") ? atoi(getenv("EV_SPLIT_WPC")) : 0;
        if (wpc > 0) smem = std::max(smem, (size_t)((160 * 1024 / (wpc + 1) + 1024) & ~255));
    }
    const dim3 grid(p.mtiles * p.ntiles);
    if (p.act == ACT_SNAKE) { ensure_dyn_smem<conv_split_kernel<BM, BN, WM, WN, 2, TERMS>>(smem, lo.device); hipLaunchKernelGGL((conv_split_kernel<BM, BN, WM, WN, 2, TERMS>), grid, dim3(256), smem, st, p); }
    else if (lean_acc(p)) { ensure_dyn_smem<conv_split_kernel<BM, BN, WM, WN, 3, TERMS>>(smem, lo.device); hipLaunchKernelGGL((conv_split_kernel<BM, BN, WM, WN, 3, TERMS>), grid, dim3(256), smem, st, p); }
    else { ensure_dyn_smem<conv_split_kernel<BM, BN, WM, WN, 1, TERMS>>(smem, lo.device); hipLaunchKernelGGL((conv_split_kernel<BM, BN, WM, WN, 1, TERMS>), grid, dim3(256), smem, st, p); }
}
// conv_h16_kernel (fp16 two-piece, block-scaled): the lean epilogues only
template <int BM, int BN, int WM, int WN>
void launch_h16(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    const size_t xs = (size_t)(BN + EV_HALO) * EVH_RSB + 16;            // (the waves' maxima sit behind the largest tile)
    constexpr size_t es = (size_t)4 * 32 * (BM / WM + 4) * sizeof(float);
    const size_t smem = xs > es ? xs : es;
    const dim3 grid(p.mtiles * p.ntiles);
    // the 16 x 16 x 32 K loop (conv_h16_kernel<..., Q = 1>) wherever the layer carries that fragment order; EV_H16Q=0: the 32 x 32 x 16 form (A/B)
    static const bool use_q = !(getenv("EV_H16Q") && atoi(getenv("EV_H16Q")) == 0);
    if (use_q && p.Wq && p.act != ACT_SNAKE) {
        if (lean_acc(p)) { ensure_dyn_smem<conv_h16_kernel<BM, BN, WM, WN, 3, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_kernel<BM, BN, WM, WN, 3, 1>), grid, dim3(256), smem, st, p); }
        else { ensure_dyn_smem<conv_h16_kernel<BM, BN, WM, WN, 1, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_kernel<BM, BN, WM, WN, 1, 1>), grid, dim3(256), smem, st, p); }
        return;
    }
    if (p.act == ACT_SNAKE) { ensure_dyn_smem<conv_h16_kernel<BM, BN, WM, WN, 2>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), smem, st, p); }
    else if (lean_acc(p)) { ensure_dyn_smem<conv_h16_kernel<BM, BN, WM, WN, 3>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), smem, st, p); }
    else { ensure_dyn_smem<conv_h16_kernel<BM, BN, WM, WN, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), smem, st, p); }
}
inline bool split_ok(const ConvLayer& L, const ConvParams& p0) {
    ConvParams p = p0;
    p.dbg &= ~4;                                        // (the no-epilogue ablation of tools/conv_bench.py exists in this build too)
    return L.Wx && lean_ok(p) && (p.isplit_log2 >= 31 || (p.isplit_log2 >= 6 && p.isplit_log2 < 31)) && L.Cin % EVX_KC == 0 && L.Cin == L.Kpad && L.Mpad % 128 == 0 && ((size_t)p.bias & 15) == 0;
}

// small-launch builds (conv_gemm_sk_kernel): 64 x 64 tiles, 16 waves, K split four ways inside the workgroup (TW = 4, KS = 4),
// or 32 x 32 tiles, 8 waves, K split eight ways (TW = 1, KS = 8) for launches of only a few dozen 64 x 64 tiles
template <bool FULL, int LEAN, int TW>
void launch_sk2(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    constexpr int KS = TW == 4 ? 4 : 8, BN = TW == 4 ? 64 : 32;
    // k-chunks per staging round: TW = 4 up to 8; TW = 1 the whole K when the X tile fits LDS (<= 150 KB) and the per-thread
    // staging registers (EV_SK_MAXF4), else the largest multiple of KS that does
    const int nchunks = p.Kpad / EV_BK, rows = BN + ((lo.halo + 7) & ~7);
    int kbs = nchunks < (TW == 4 ? 8 : 32) ? nchunks : (TW == 4 ? 8 : 32);
    while (kbs > 8 && ((size_t)rows * (32 * kbs + 4) * 4 > 150 * 1024 || rows * kbs * 8 > EV_SK_MAXF4(TW))) kbs = (kbs - 1) & ~7;
    const size_t xs = 2 * EV_MAX_TAPS + (size_t)rows * (32 * kbs + 4);
    const size_t red = (size_t)(KS - 1) * TW * 16 * 64, es = (size_t)TW * 32 * 36;
    size_t smem = std::max(xs, std::max(red, es)) * sizeof(float);
    ConvParams q = p;
    q.sk_kb = kbs;
    ensure_dyn_smem<conv_gemm_sk_kernel<KS, FULL, LEAN, TW>>(smem, lo.device);
    hipLaunchKernelGGL((conv_gemm_sk_kernel<KS, FULL, LEAN, TW>), dim3(p.mtiles * p.ntiles), dim3(64 * TW * KS), smem, st, q);
}
template <int TW>
void launch_sk(const ConvParams& p, hipStream_t st, const LaunchOpts& lo) {
    static const bool no_lean = getenv("EV_NO_LEAN") != nullptr;
    if (!no_lean && lean_ok(p)) {
        if (p.act == ACT_SNAKE) launch_sk2<false, 2, TW>(p, st, lo);
        else if (lean_acc(p)) launch_sk2<false, 3, TW>(p, st, lo);
        else launch_sk2<false, 1, TW>(p, st, lo);
    } else if (p.act == ACT_NONE || p.act == ACT_LRELU) launch_sk2<false, 0, TW>(p, st, lo);
    else launch_sk2<true, 0, TW>(p, st, lo);
}

constexpr int EV_CAPTURE_SLOTS = 8;       // captured ev_cfm_decode calls a handle can hold (pinned staging that is never recycled)
constexpr int EV_SK_MAXWG = 1024;          // persistent workgroups of a balanced launch (<= 4 per CU on 256 CUs)
constexpr int EV_SK_PART_FLOATS = 16384;   // largest partial accumulator tile handed over (64 KB: a 64 x 192 conv tile is 48 KB)
// Hand-off area of the balanced launches: allocated once per handle (never inside a stream capture: ev_load_estimator calls this)
int ensure_sk(ev_handle* h, bool hot_path = true) {
    if (h->sk_ctrl) return 0;
    if (hot_path) ++h->n_allocs;   // (every loader and ev_reserve call this first, so a hot call never gets here; if one ever does, ev_alloc_count shows it)
    const size_t words = 16 + 2 * EV_SK_MAXWG;         // control words | one flag per workgroup | one claim word per workgroup (work stealing)
    HIPCHK(h, hipMalloc((void**)&h->sk_ctrl, words * sizeof(unsigned)));
    HIPCHK(h, hipMemset(h->sk_ctrl, 0, words * sizeof(unsigned)));
    HIPCHK(h, hipMalloc((void**)&h->sk_part, (size_t)EV_SK_MAXWG * 2 * EV_SK_PART_FLOATS * sizeof(float)));   // (freed by ev_destroy)
    return 0;
}

// The balanced persistent build of a conv launch (conv_gemm_bal_kernel): G = wpc x CUs workgroups share the (tile, k-chunk) units.
template <int BM, int BN, int WM, int WN>
int launch_bal(ev_handle* h, ConvParams p, const LaunchOpts& lo, int wpc, unsigned long long wtab = 0) {
    if (ensure_sk(h)) return 1;
    const int nchunks = p.Kpad / EV_BK;
    const long U = (long)p.mtiles * p.ntiles * nchunks;
    const int G = wpc * h->ncu;
    p.sk.ctrl = h->sk_ctrl; p.sk.flags = h->sk_ctrl + 16; p.sk.part = h->sk_part; p.sk.part_floats = EV_SK_PART_FLOATS;
    p.sk.q = (int)(U / G); p.sk.r = (int)(U % G); p.sk.spin_limit = h->sk_spin;
    if (wtab) {            // weighted positions (stacked layers): a unit of M tile mt weighs its tap count
        int wsum = 0;
        for (int t = 0; t < p.mtiles; ++t) wsum += (int)((wtab >> (4 * t)) & 15ull);
        const long P = (long)p.ntiles * nchunks * wsum;
        p.sk.wsum = wsum; p.sk.wtab = wtab; p.sk.mtiles = p.mtiles; p.sk.nchunks = nchunks;
        p.sk.q = (int)(P / G); p.sk.r = (int)(P % G);
    }
    const size_t xs = (size_t)(BN + ((lo.halo + 7) & ~7)) * 36;
    constexpr size_t es = (size_t)4 * 32 * (BM / WM + 4);
    size_t smem = (xs > es ? xs : es) * sizeof(float);
    p.sk.lds_word = (int)smem;
    smem += 16;
    if (lean_acc(p)) { ensure_dyn_smem<conv_gemm_bal_kernel<BM, BN, WM, WN, 3>>(smem, lo.device); hipLaunchKernelGGL((conv_gemm_bal_kernel<BM, BN, WM, WN, 3>), dim3(G), dim3(256), smem, h->stream, p); }
    else { ensure_dyn_smem<conv_gemm_bal_kernel<BM, BN, WM, WN, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_gemm_bal_kernel<BM, BN, WM, WN, 1>), dim3(G), dim3(256), smem, h->stream, p); }
    return 0;
}
// ... and of the bf16-split build (conv_split_bal_kernel): a unit = (tile, 64-channel chunk)
template <int BM, int BN, int WM, int WN>
int launch_split_bal(ev_handle* h, ConvParams p, const LaunchOpts& lo, int wpc) {
    if (ensure_sk(h)) return 1;
    const int nchunks = p.Kpad / EVX_KC;
    const long U = (long)p.mtiles * p.ntiles * nchunks;
    const int G = wpc * h->ncu;
    p.sk.ctrl = h->sk_ctrl; p.sk.flags = h->sk_ctrl + 16; p.sk.part = h->sk_part; p.sk.part_floats = EV_SK_PART_FLOATS;
    p.sk.q = (int)(U / G); p.sk.r = (int)(U % G); p.sk.spin_limit = h->sk_spin;
    const size_t xs = (size_t)(BN + ((lo.halo + 7) & ~7)) * EVX_RSB;
    constexpr size_t es = (size_t)4 * 32 * (BM / WM + 4) * sizeof(float);
    size_t smem = xs > es ? xs : es;
    p.sk.lds_word = (int)smem;
    smem += 16;
    static_assert((size_t)BM * BN <= EV_SK_PART_FLOATS, "hand-off slot");
    if (lean_acc(p)) { ensure_dyn_smem<conv_split_bal_kernel<BM, BN, WM, WN, 3>>(smem, lo.device); hipLaunchKernelGGL((conv_split_bal_kernel<BM, BN, WM, WN, 3>), dim3(G), dim3(256), smem, h->stream, p); }
    else { ensure_dyn_smem<conv_split_bal_kernel<BM, BN, WM, WN, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_split_bal_kernel<BM, BN, WM, WN, 1>), dim3(G), dim3(256), smem, h->stream, p); }
    return 0;
}
template <int BM, int BN, int WM, int WN>
int launch_h16_bal(ev_handle* h, ConvParams p, const LaunchOpts& lo, int wpc) {
    if (ensure_sk(h)) return 1;
    const int nchunks = p.Kpad / EVX_KC;
    const long U = (long)p.mtiles * p.ntiles * nchunks;
    const int G = wpc * h->ncu;
    p.sk.ctrl = h->sk_ctrl; p.sk.flags = h->sk_ctrl + 16; p.sk.part = h->sk_part; p.sk.part_floats = EV_SK_PART_FLOATS;
    p.sk.q = (int)(U / G); p.sk.r = (int)(U % G); p.sk.spin_limit = h->sk_spin;
    if (h->sk_steal) { p.sk.claims = h->sk_ctrl + 16 + EV_SK_MAXWG; p.sk.seq = ++h->sk_seq; }
    constexpr int NW = WM * WN;                         // 4 waves: two workgroups per CU; 8 waves (256 x 128 tiles): one
    const size_t xs = (size_t)(BN + ((lo.halo + 7) & ~7)) * EVH_RSB;
    constexpr size_t es = (size_t)NW * 32 * (BM / WM + 4) * sizeof(float);
    size_t smem = xs > es ? xs : es;
    p.sk.lds_word = (int)smem;
    smem += 16 + 3 * NW * sizeof(float) + 16;           // the wait word + three sets of the waves' maxima
    static_assert((size_t)BM * BN <= 2 * EV_SK_PART_FLOATS, "hand-off slot");
    if ((size_t)BM * BN > EV_SK_PART_FLOATS) {          // 256 x 128 tiles: slots twice as large, half as many workgroups (the area is sized for 1024 x 2 x 64 KB)
        if ((size_t)G * 2 > EV_SK_MAXWG) return fail(h, "launch_h16_bal: %d workgroups of 256 x 128 tiles exceed the hand-off area", G);
        p.sk.part_floats = (int)((size_t)BM * BN);
    }
    static const char* stamp_file = getenv("EV_BAL_STAMPS");         // diagnostic: phase stamps of a few workgroups, once per layer shape
    static std::vector<long> stamped;
    const long sig = ((long)p.nrows << 24) ^ ((long)p.Kpad << 12) ^ p.Mpad ^ ((long)p.ntaps << 40);
    unsigned long long* d = nullptr;
    if (stamp_file && *stamp_file && stamped.size() < 8 && std::find(stamped.begin(), stamped.end(), sig) == stamped.end()) {
        stamped.push_back(sig);
        HIPCHK(h, hipMalloc((void**)&d, (size_t)G * 16 * sizeof(unsigned long long)));
        HIPCHK(h, hipMemsetAsync(d, 0, (size_t)G * 16 * sizeof(unsigned long long), h->stream));
        p.stamps = d;
    }
    {   // A/B: start stagger of the second workgroup of every CU, in steps of ~1 us (conv_h16_bal_kernel)
        static const int stag = getenv("EV_BAL_STAGGER") ? atoi(getenv("EV_BAL_STAGGER")) : 0;
        p.stagger_slots = (wpc == 2) ? stag : 0;
    }
    if constexpr (NW == 8) {                            // passes of 32 rows: 5 cover a 3-tap layer's tile, 6 the widest halo
        const bool narrow8 = BN + p.halo_lo + p.halo_hi <= 32 * 5;
        if (lean_acc(p)) {
            if (narrow8) { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 3, 5>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 3, 5>), dim3(G), dim3(512), smem, h->stream, p); }
            else { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 3, 6>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 3, 6>), dim3(G), dim3(512), smem, h->stream, p); }
        } else {
            if (narrow8) { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 1, 5>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 1, 5>), dim3(G), dim3(512), smem, h->stream, p); }
            else { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 1, 6>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 1, 6>), dim3(G), dim3(512), smem, h->stream, p); }
        }
    } else {
    const bool narrow = BN + p.halo_lo + p.halo_hi <= 16 * 9;         // a 3-tap layer: nine staging passes instead of twelve
    if (lean_acc(p)) {
        if (narrow) { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 3, 9>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 3, 9>), dim3(G), dim3(256), smem, h->stream, p); }
        else { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 3>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 3>), dim3(G), dim3(256), smem, h->stream, p); }
    } else {
        if (narrow) { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 1, 9>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 1, 9>), dim3(G), dim3(256), smem, h->stream, p); }
        else { ensure_dyn_smem<conv_h16_bal_kernel<BM, BN, WM, WN, 1>>(smem, lo.device); hipLaunchKernelGGL((conv_h16_bal_kernel<BM, BN, WM, WN, 1>), dim3(G), dim3(256), smem, h->stream, p); }
    }
    }
    if (d) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> st((size_t)G * 16);
        HIPCHK(h, hipMemcpy(st.data(), d, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        hipFree(d);
        if (FILE* f = fopen(stamp_file, "a")) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < G; ++w) for (int k = 0; k < 16; ++k) if (st[(size_t)w * 16 + k]) { t0 = std::min(t0, st[(size_t)w * 16 + k]); t1 = std::max(t1, st[(size_t)w * 16 + k]); }
            fprintf(f, "## conv_h16_bal_kernel rows=%d Kpad=%d Mpad=%d taps=%d: %d workgroups, %ld units (q=%d r=%d), first stamp -> last %.2f us; per workgroup: start offset | us between consecutive stamps (per pass: pre-scan | per chunk: staged, MFMAs | partial stored or contributors added, epilogue | end)\n",
                    p.nrows, p.Kpad, p.Mpad, p.ntaps, G, U, p.sk.q, p.sk.r, (double)(t1 - t0) / 100.0);
            for (int w : {0, 1, 2, 3, 100, 101, 255, 256, 511}) {
                if (w >= G || !st[(size_t)w * 16]) continue;
                fprintf(f, "  wg %3d: +%.2f |", w, (double)(st[(size_t)w * 16] - t0) / 100.0);
                for (int k = 1; k < 16 && st[(size_t)w * 16 + k]; ++k) fprintf(f, " %.2f", (double)(st[(size_t)w * 16 + k] - st[(size_t)w * 16 + k - 1]) / 100.0);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return 0;
}
inline bool split_bal_ok(const ev_handle* h, const ConvLayer& L, const ConvParams& p, long nwg, int wpc) {
    static const bool off = getenv("EV_NO_CONV_BALANCE") != nullptr;
    return !off && h->sk_balance && h->ncu > 0 && split_ok(L, p) && p.act != ACT_SNAKE && !p.dbg && !p.stamps && !p.gn_part &&
           nwg * 2 >= h->ncu && wpc * h->ncu <= EV_SK_MAXWG && (long)nwg * (p.Kpad / EVX_KC) >= (long)wpc * h->ncu;
}
// A conv launch takes the balanced build when it is dense (every tap in every M tile: equal units), has a lean non-transcendental
// epilogue, fills at least one tile per CU and at most a few rounds (deep grids balance by themselves), and the tile's accumulators fit
// a hand-off slot.
inline bool bal_ok(const ev_handle* h, const ConvLayer& L, const ConvParams& p, long nwg, int wpc) {
    static const bool off = getenv("EV_NO_CONV_BALANCE") != nullptr;
    // stacked layers (M tiles of unequal cost) keep one tile per workgroup unless EV_CONV_BALANCE_W=1: measured at batch 64
    // (tools/shape_profile.py) the weighted balanced grid is 3 % SLOWER on them (256 -> 512: 3.13 -> 3.24 ms, 512 -> 512: 1.88 -> 1.94):
    // 2080 tiles of mixed weight already even out over the CUs, the hand-offs do not pay
    static const bool no_w = getenv("EV_CONV_BALANCE_W") == nullptr;
    if (L.sparse_taps) {   // M tiles of unequal cost: weighted units (64-channel tiling only), every tile with at least one tap
        if (no_w || p.mtiles > 16 || !L.nact[1]) return false;
        long wsum = 0;
        for (int t = 0; t < p.mtiles; ++t) { if (L.nact64[t] < 1) return false; wsum += L.nact64[t]; }
        if ((nwg / p.mtiles) * (p.Kpad / EV_BK) * wsum < 16L * wpc * h->ncu) return false;   // every workgroup several units (none empty)
    }
    return !off && h->sk_balance && h->ncu > 0 && lean_ok(p) && p.act != ACT_SNAKE && !p.dbg && !p.stamps && !p.gn_part &&
           nwg >= h->ncu && nwg < 4L * wpc * h->ncu && wpc * h->ncu <= EV_SK_MAXWG && (long)nwg * (p.Kpad / EV_BK) >= (long)wpc * h->ncu;
}

constexpr int EV_GN_MAXTILES = 256;    // 32-row tiles of a launch that may leave GroupNorm statistics (EstBufs::GNP)
int launch_conv(ev_handle* h, const ConvLayer& L, const float* X, int ldx, float* Y, int ldy, const Geom& g, const Epi& e) {
    ConvParams p;
    memset(&p, 0, sizeof p);
    h->gn_stats_tiles = 0;
    p.X = X; p.ldx = ldx; p.Cin = L.Cin; p.isplit_log2 = e.isplit_log2; p.isstride = e.isstride;
    p.W = L.W; p.Wx = L.Wx; p.Wh = L.Wh; p.Wq = L.Wq; p.wh_scale = L.wh_scale; p.Mpad = L.Mpad; p.Kpad = L.Kpad; p.bias = L.bias;
    p.Y = Y; p.ldy = ldy; p.Cout = L.Cout; p.osplit_log2 = e.osplit_log2; p.osstride = e.osstride; p.mmul = e.mmul;
    p.nrows = g.nrows; p.S = g.S; p.P = g.P; p.T = g.T;
    p.ntaps = L.ntaps; for (int i = 0; i < L.ntaps; ++i) p.off[i] = L.off[i];
    if (!L.sparse_taps || L.kstack_mt > 0) {   // evenly spaced taps of a dense (or stacked) layer: the list is arithmetic, no fetch
        bool even = true;
        const int d = L.ntaps > 1 ? L.off[1] - L.off[0] : 0;
        for (int i = 2; i < L.ntaps; ++i) even = even && (L.off[i] - L.off[i - 1] == d);
        if (even) {
            p.plane_bytes = (L.Mpad / 32) * (L.Kpad / 8) * 1024; p.koff0 = L.off[0]; p.kdoff = d;
            if (!L.sparse_taps) p.ktaps_n = L.ntaps;
            else { p.kstack_mt = L.kstack_mt; p.kstack_tap = L.kstack_tap; }
        }
    }
    p.halo_lo = L.halo_lo; p.halo_hi = L.halo_hi;
    p.pro_lrelu = e.pro_slope >= 0.f; p.pro_slope = e.pro_slope;
    p.act = e.act; p.act_slope = e.act_slope; p.act_a = e.act_a; p.act_b = e.act_b;
    p.mask1 = e.mask1; p.scale = e.scale; p.R = e.R; p.ldr = e.ldr; p.accum = e.accum; p.div3 = e.div3;
    p.act2_lrelu = e.act2_lrelu; p.act2_slope = e.act2_slope; p.mask2 = e.mask2; p.rowmask = e.rowmask;
    p.Y2 = e.Y2; p.ldy2 = e.ldy2; p.dbg = e.dbg; p.stamps = e.stamps;
    p.xmax = e.xmax; p.xmax_n = e.xmax_n; p.rmax = e.rmax; p.yold = e.yold; p.ymax_mul = e.ymax_mul; h->amax_emitted = false;   // (p.ymax: set below, once the build is known)
    if ((ldx & 3) || (L.Cin & 3)) return fail(h, "conv input must be float4-aligned (ldx %d Cin %d)", ldx, L.Cin);
    {   // buffer (SRSRC) addressing uses 32-bit byte offsets: every tensor of a launch must stay below 4 GiB
        const double lim = 4294967296.0;
        const double ymax = (double)g.nrows * ldy * 4.0, xmax = (double)g.nrows * ldx * 4.0;
        if (xmax >= lim || ymax >= lim || (e.R && (double)g.nrows * e.ldr * 4.0 >= lim) || (e.Y2 && (double)g.nrows * e.ldy2 * 4.0 >= lim))
            return fail(h, "tensor of %.1f GiB exceeds the 4 GiB buffer-addressing limit: split the batch", (xmax > ymax ? xmax : ymax) / 1073741824.0);
    }

    // Tile choice.  cfg: 0 = 128x128, 1 = 64x128, 2 = 32x256, 5 = 64x192, 6 = 64x64 (channels x frames per workgroup).
    // A launch that fits the chip in about one round of workgroups is decided by its makespan: e.g. 1040 tiles on
    // 1024 slots run 16 stragglers after everyone else (measured: the matrix pipes idle for half of such a launch), and
    // 520 tiles on 256 CUs leave 8 CUs with three tiles.  So estimate, per candidate, the busiest CU's share of the
    // MFMA work and take the minimum; deep grids (HiFi-GAN) keep the 64x128 tile measured best by tools/conv_bench.py.
    int cfg;
    if (L.Cout <= 32) cfg = 2;
    else {
        struct Cand { int cfg, bm, bn, slots; };
        const Cand cands[3] = {{1, 64, 128, 4}, {5, 64, 192, 4}, {6, 64, 64, 5}};
        double best = 1e30;
        cfg = 1;
        for (const Cand& c : cands) {
            const long nwg = (long)((L.Cout + c.bm - 1) / c.bm) * ((g.nrows + c.bn - 1) / c.bn);
            const double w = (double)(c.bm / 32) * (c.bn / 32);           // MFMA tiles per workgroup
            double t;
            if (nwg <= 256L * c.slots) t = (double)((nwg + 255) / 256) * w;          // one round: the busiest CU
            else t = (double)nwg * w / 256.0 + (nwg < 256L * c.slots * 3 ? w : 0.0);  // deep grid: balanced + a short tail
            t *= (c.cfg == 6 ? 1.08 : 1.0);                               // small tiles re-read more operands
            if (t < best - 1e-9) { best = t; cfg = c.cfg; }
        }
    }
    // stacked layers (a 3-tap conv over a 1x1 conv: half of the M tiles carry one tap) in about one round of workgroups: the
    // 64 x 64 tile balances the heavy and the light tiles best (measured 3.6 -> 3.1 ms on the T/2-level resnets)
    if (L.sparse_taps && L.Cout > 32 && (long)((L.Cout + 63) / 64) * ((g.nrows + 63) / 64) <= 256L * 5 * 2) cfg = 6;
    // launches far below one round of workgroups are a latency chain per workgroup: use the build that prefetches the
    // next chunk's X tile through registers (measured 7-10 % on batch-1 decodes, nothing on full grids)
    if (cfg == 6 && (long)((L.Cout + 63) / 64) * ((g.nrows + 63) / 64) <= 320) cfg = 8;
    // deep grids (>= 6 rounds of 64x128 tiles, dense taps): larger tiles amortise the X staging and the weight-fragment stream over
    // twice the MFMAs — measured on the HiFi-GAN layers (tools/shape_profile.py with EV_FORCE_CFG): 128x128 is 2-3 % faster
    // for Cin <= 128 and for 3-tap layers, 64x192 for the 7 / 11-tap layers at Cin = 256
    if (cfg == 1 && L.Cout % 128 == 0 && !L.sparse_taps && (long)(L.Cout / 64) * ((g.nrows + 127) / 128) >= 256L * 4 * 6)
        cfg = (L.Cin > 128 && L.ntaps >= 7) ? 5 : 0;
    // launches far below one workgroup per CU with a K loop worth splitting: the 16-wave split-K build (batch-1 decodes)
    {
        static const bool no_sk = getenv("EV_NO_SK") != nullptr;
        const long wg64 = (long)((L.Cout + 63) / 64) * ((g.nrows + 63) / 64);
        if (!no_sk && (cfg == 6 || cfg == 8) && wg64 <= 192 && (L.Kpad / EV_BK) * L.ntaps >= 8) cfg = 9;
        // (a short K loop gains nothing from the split, but the 32 x 32 build for it is the instruction-lean conv_sk32_kernel)
        if (!no_sk && (cfg == 6 || cfg == 8) && wg64 <= 96 && L.Kpad == 128 && L.Cin == 128 && !L.sparse_taps) cfg = 9;
        // ... as 32 x 32 tiles on four times as many CUs (EV_SK32_MAX, A/B: 96 while the build held one workgroup per CU; since
        // conv_sk32_kernel fits two — 119 VGPRs — launches of up to 512 such tiles are still one round: 192; config-5 mean -1.9 %, p99 -4.5 %)
        static const int sk32_max = getenv("EV_SK32_MAX") ? atoi(getenv("EV_SK32_MAX")) : 192;
        if (cfg == 9 && wg64 <= sk32_max && L.Cout >= 32) cfg = 19;
    }
    {   // A/B override for the stacked sparse-tap layers only (a 3-tap conv over a 1x1 conv): EV_SPARSE_CFG=<cfg>[,<min rows>]
        static const char* senv = getenv("EV_SPARSE_CFG");
        if (senv && *senv && L.sparse_taps && L.Cout > 32) {
            int c = atoi(senv), minrows = 0;
            if (const char* comma = strchr(senv, ',')) minrows = atoi(comma + 1);
            if (g.nrows >= minrows) cfg = c;
        }
    }
    {   // deep grids of dense-channel layers: the bf16-split build (EV_SPLIT=0: fp32 MFMA everywhere; 3 / 9: products per element pair, A/B)
        const int split_terms = h->split_terms;
        // (polyphase transposed convs whose 64-channel M tiles carry different tap subsets — a 128-channel tile would compute the union — take
        // the 64 x 128 tile below)
        const long nwg128 = (long)(L.Mpad / 128) * ((g.nrows + 127) / 128);
        if (split_terms > 0 && (cfg == 0 || cfg == 1 || cfg == 5 || cfg == 6) && split_ok(L, p) && L.Cout % 128 == 0 && (!L.sparse_taps || (L.tile128_exact && L.kstack_mt == 0)) && nwg128 >= 2L * 2 * h->ncu && h->ncu > 0)
            cfg = split_terms == 3 ? 43 : split_terms == 9 ? 49 : split_terms == 16 ? 46 : 40;
        // polyphase transposed convs (M tiles of 64 channels with different tap subsets) on deep grids: 64 x 128 tiles of the split build
        else if ((split_terms == 6 || split_terms == 3 || split_terms == 16) && (cfg == 0 || cfg == 1 || cfg == 5 || cfg == 6) && split_ok(L, p) && L.Cout % 64 == 0 && L.kstack_mt == 0 && h->ncu > 0 &&
                 (L.sparse_taps || L.Cout % 128 != 0) &&
                 (long)((L.Cout + 63) / 64) * ((g.nrows + 127) / 128) >= 2L * 3 * h->ncu) cfg = 41;
        // launches of a few rounds (the U-Net convs of a large-batch decode): the balanced persistent grid of the split build
        else if ((split_terms == 6 || split_terms == 16) && (cfg == 1 || cfg == 5 || cfg == 6 || cfg == 0) && L.Cout == L.Mpad && split_bal_ok(h, L, p, nwg128, 2)) cfg = 60;
    }
    {   // debugging / test override: EV_FORCE_CFG=<0..3> forces one tile configuration for every conv launch
        static const char* env = getenv("EV_FORCE_CFG");
        if (env && *env) cfg = atoi(env);
    }
    LaunchOpts lo;
    lo.device = h->device;
    {   // A/B override: EV_KB=<1|2> forces the k-chunks per stage of every conv launch
        static const char* kenv = getenv("EV_KB");
        if (kenv && *kenv) lo.kb = atoi(kenv) == 2 ? 2 : 1;
    }
    if (e.force_cfg >= 0) { cfg = e.force_cfg % 100; lo.wgs_per_cu = (e.force_cfg / 100) % 10; lo.kb = e.force_cfg >= 1000 ? 2 : 1; }
    lo.halo = L.halo_lo + L.halo_hi;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
        }
        e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    {   // start stagger (see conv_gemm_kernel): only worth it when the grid is several rounds deep
        const int slots = cfg == 0 ? 3 : ((cfg == 1 || cfg == 5 || cfg == 7) ? 4 : ((cfg == 6 || cfg == 8) ? 5 : 3));
        const long wgs = (cfg == 5 || cfg == 7) ? (long)((L.Cout + 63) / 64) * ((g.nrows + 191) / 192) : (cfg == 6 || cfg == 8) ? (long)((L.Cout + 63) / 64) * ((g.nrows + 63) / 64) : cfg == 0 ? (long)(L.Mpad / 128) * ((g.nrows + 127) / 128) : (cfg == 1 ? (long)((L.Cout + 63) / 64) * ((g.nrows + 127) / 128) : (long)((L.Cout + 31) / 32) * ((g.nrows + 255) / 256));
        p.stagger_slots = (wgs >= 256L * slots * 3 && cfg < 10) ? slots : 0;
        if (e.stagger >= 0) p.stagger_slots = e.stagger;
        static const char* senv = getenv("EV_STAGGER");
        if (senv && *senv) p.stagger_slots = atoi(senv) ? p.stagger_slots : 0;
    }
    if ((cfg == 40 || cfg == 41 || cfg == 43 || cfg == 46 || cfg == 49 || cfg == 60) && !split_ok(L, p)) cfg = 0;
    if (cfg == 46 && !L.Wh) cfg = 40;
    {   // which builds leave bounds of their output: the fp16 builds 46 / 47 (cfg 41 becomes 47 below when the layer has fp16 pieces) and the
        // fp32 conv_gemm_kernel tiles with the lean non-SnakeBeta epilogues; a residual without slots of its own cannot be bounded
        const bool lean1or3 = lean_ok(p) && p.act != ACT_SNAKE && !getenv("EV_NO_LEAN");
        const bool h16 = (cfg == 46) || (cfg == 41 && h->split_terms == 16 && L.Wh);
        const bool gemm = (cfg == 0 || cfg == 1 || cfg == 5 || cfg == 6) && lean1or3 && lo.kb == 1;
        const bool no_amax = !h->use_amax;                                          // A/B: every fp16 tile pre-scans
        if (!no_amax && e.ymax && lean1or3 && (h16 || gemm) && (!e.R || e.rmax) && (!e.accum || e.yold)) { p.ymax = e.ymax; h->amax_emitted = true; }
        if (no_amax) p.xmax = nullptr;
    }
    if (cfg == 46) {   // 128 x 128 on the fp16 pipe, two block-scaled pieces, three products
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_h16<128, 128, 2, 2>(p, h->stream, lo);
    } else
    if (cfg == 41) {   // 64 x 128 on the bf16 pipe (per-tile tap lists of the 64-channel tiling)
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        if (h->split_terms == 16 && L.Wh) { launch_h16<64, 128, 2, 2>(p, h->stream, lo); cfg = 47; }
        else if (h->split_terms == 3) launch_split<64, 128, 2, 2, 3>(p, h->stream, lo); else launch_split<64, 128, 2, 2, 6>(p, h->stream, lo);
    } else
    if (cfg == 60) {   // 128 x 128 on the bf16 pipe, balanced persistent grid
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        static const int bal_wgs = getenv("EV_SPLIT_BAL_WGS") ? atoi(getenv("EV_SPLIT_BAL_WGS")) : 2;    // A/B: persistent workgroups per CU (1 or 2)
        // EV_BAL_WIDE=1 (A/B, measured null — not the default): layers of 256-channel multiples as ONE eight-wave workgroup per CU on 256 x 128
        // tiles, so that the X tile is staged once for both halves of the output channels instead of by two lockstep workgroups.  Same box, batch 64:
        // 256 -> 256 k3 2.585 -> 2.766 ms (T), 2.254 -> 2.426 ms (T / 2), 512 -> 512 1.447 -> 1.415 ms; decode 25.9 -> 26.5 ms (gpurun_out/s22).
        // Halving the bytes of the burst a chunk begins with does not shorten the chunk: what the two workgroups of a CU wait for in lockstep
        // is latency, and eight waves meet at every barrier instead of four.
        static const bool use_wide = getenv("EV_BAL_WIDE") && atoi(getenv("EV_BAL_WIDE")) != 0;
        bool wide = use_wide && h->split_terms == 16 && L.Wh && L.Mpad % 256 == 0 && L.Cout == L.Mpad && bal_wgs != 1;
        wide = wide && L.pair256_uniform;
        if (wide) {
            p.mtiles = L.Mpad / 256; p.mt_mul = 2;
            if (launch_h16_bal<256, 128, 4, 2>(h, p, lo, 1)) return 1;
            cfg = 68;
        } else
        if (h->split_terms == 16 && L.Wh) { if (launch_h16_bal<128, 128, 2, 2>(h, p, lo, bal_wgs == 1 ? 1 : 2)) return 1; cfg = 66; }
        else if (launch_split_bal<128, 128, 2, 2>(h, p, lo, bal_wgs == 1 ? 1 : 2)) return 1;
    } else
    if (cfg == 40 || cfg == 43 || cfg == 49) {   // 128 x 128 on the bf16 pipe
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        if (cfg == 40) launch_split<128, 128, 2, 2, 6>(p, h->stream, lo);
        else if (cfg == 43) launch_split<128, 128, 2, 2, 3>(p, h->stream, lo);
        else launch_split<128, 128, 2, 2, 9>(p, h->stream, lo);
    } else if (cfg == 0) {
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<128, 128, 2, 2>(p, h->stream, lo);
    } else if (cfg == 1) {
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<64, 128, 2, 2>(p, h->stream, lo);
    } else if (cfg == 5) {
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 191) / 192; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        static const int bal5 = getenv("EV_BAL5") ? atoi(getenv("EV_BAL5")) : 0;      // A/B: 0 = as is, 192 = balanced 64 x 192, 64 = balanced 64 x 64
        if (bal5 != 0) { p.ymax = nullptr; h->amax_emitted = false; }
        if (bal5 == 192 && lo.kb == 1 && bal_ok(h, L, p, (long)p.mtiles * p.ntiles, 3)) { if (launch_bal<64, 192, 2, 2>(h, p, lo, 3)) return 1; cfg = 55; }
        else if (bal5 == 64 && lo.kb == 1 && bal_ok(h, L, p, (long)((L.Cout + 63) / 64) * ((g.nrows + 63) / 64), 4)) {
            p.ntiles = (g.nrows + 63) / 64;
            if (launch_bal<64, 64, 2, 2>(h, p, lo, 4)) return 1;
            cfg = 56;
        } else launch_cfg<64, 192, 2, 2>(p, h->stream, lo);
    } else if (cfg == 7) {   // 64 x 192 with register-prefetched X staging
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 191) / 192; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<64, 192, 2, 2, true>(p, h->stream, lo);
    } else if (cfg == 8) {   // 64 x 64 with register-prefetched X staging
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 63) / 64; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<64, 64, 2, 2, true>(p, h->stream, lo);
    } else if (cfg == 9) {   // 64 x 64 tiles, 16 waves, split-K inside the workgroup
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 63) / 64; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_sk<4>(p, h->stream, lo);
    } else if (cfg == 19) {  // 32 x 32 tiles, 8 waves, split-K eight ways
        p.mtiles = (L.Cout + 31) / 32; p.ntiles = (g.nrows + 31) / 32; p.taplist = L.taplist[2]; p.nact_tab = L.nact[2]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        static const bool no_sk32_lean = getenv("EV_NO_SK32_LEAN") != nullptr;
        static const bool no_lean = getenv("EV_NO_LEAN") != nullptr;
        const int xr = 32 + p.halo_lo + p.halo_hi;
        // conv_sk32_kernel's preconditions (everything else: the general small-launch build)
        const bool fast = !no_sk32_lean && !no_lean && lean_ok(p) && (p.ktaps_n > 0 || p.kstack_mt > 0) && p.isplit_log2 >= 31 && !p.pro_lrelu && p.Cin == p.Kpad &&
                          (p.Kpad == 128 || p.Kpad == 256 || p.Kpad == 512 || p.Kpad == 1024) && (ldx % 4) == 0 && xr <= 16 * (512 / (p.Kpad / 4)) && (size_t)g.nrows * ldx * 4 < ((size_t)1 << 31) &&
                          (size_t)xr * (p.Kpad + 4) * 4 <= 150 * 1024;
        if (fast) {
            const size_t smem = std::max((size_t)xr * (p.Kpad + 4), (size_t)7 * 16 * 64) * sizeof(float);
            const dim3 grid(p.mtiles, p.ntiles);
            // GroupNorm statistics of the output for the layer that follows: one utterance, the tile stored as accumulated, one
            // 32-channel group per M tile over the (first) 256 output channels
            static const bool no_gn_stats = getenv("EV_NO_GN_STATS") != nullptr;
            if (!no_gn_stats && e.gn_part && g.nrows == g.S && p.ntiles <= EV_GN_MAXTILES && p.act == ACT_NONE && !p.R && !lean_acc(p) && !p.mask1 && !p.mask2 &&
                (p.kstack_mt == 8 || (p.kstack_mt == 0 && L.Cout == 256))) {
                p.gn_part = e.gn_part;
                h->gn_stats_tiles = p.ntiles;
            }
            if (p.act == ACT_SNAKE) { ensure_dyn_smem<conv_sk32_kernel<2>>(smem, h->device); hipLaunchKernelGGL(conv_sk32_kernel<2>, grid, dim3(512), smem, h->stream, p); }
            else if (lean_acc(p)) { ensure_dyn_smem<conv_sk32_kernel<3>>(smem, h->device); hipLaunchKernelGGL(conv_sk32_kernel<3>, grid, dim3(512), smem, h->stream, p); }
            else { ensure_dyn_smem<conv_sk32_kernel<1>>(smem, h->device); hipLaunchKernelGGL(conv_sk32_kernel<1>, grid, dim3(512), smem, h->stream, p); }
        } else launch_sk<1>(p, h->stream, lo);
    } else if (cfg == 6) {
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 63) / 64; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        if (lo.kb == 1 && bal_ok(h, L, p, (long)p.mtiles * p.ntiles, 4)) {
            p.ymax = nullptr; h->amax_emitted = false;          // (the balanced build leaves no bounds)
            unsigned long long wtab = 0;
            if (L.sparse_taps) for (int t = 0; t < p.mtiles; ++t) wtab |= (unsigned long long)L.nact64[t] << (4 * t);
            if (launch_bal<64, 64, 2, 2>(h, p, lo, 4, wtab)) return 1;
            cfg = L.sparse_taps ? 57 : 56;
        }
        else launch_cfg<64, 64, 2, 2>(p, h->stream, lo);
    } else if (cfg == 10 && !L.sparse_taps && L.Mpad % 128 == 0) {   // 128 x 192: less halo per MFMA for the wide-halo layers at Cout = 128
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 191) / 192; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = 0;
        launch_cfg<128, 192, 2, 2>(p, h->stream, lo);
    } else if (cfg == 20 && L.Mpad % 128 == 0) {   // 128 x 128 with the next chunk's X tile prefetched through registers
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<128, 128, 2, 2, true>(p, h->stream, lo);
    } else if (cfg == 13 && !L.sparse_taps && L.Mpad % 128 == 0) {   // 128 x 128 as 4 x 1 waves: a wave owns 32 channels x 128 frames, so a
        p.mtiles = L.Mpad / 128; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[0]; p.nact_tab = L.nact[0]; p.tl_stride = 0;   // weight fragment feeds 16 MFMAs (half the L2 stream of 2 x 2)
        launch_cfg<128, 128, 4, 1>(p, h->stream, lo);
    } else if (cfg == 14) {   // 64 x 256 as 2 x 2 waves: 32 channels x 128 frames per wave
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 255) / 256; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<64, 256, 2, 2>(p, h->stream, lo);
    } else if (cfg == 12 && !L.sparse_taps && L.Mpad % 256 == 0) {   // 256 x 64: every output channel of a 256-wide layer from one X tile
        p.mtiles = L.Mpad / 256; p.ntiles = (g.nrows + 63) / 64; p.taplist = L.taplist[0]; p.nact_tab = nullptr; p.tl_stride = 0;
        launch_cfg<256, 64, 4, 1>(p, h->stream, lo);
    } else if (cfg == 4) {   // 64 x 128 tile with register-prefetched X staging (single-round launches)
        p.mtiles = (L.Cout + 63) / 64; p.ntiles = (g.nrows + 127) / 128; p.taplist = L.taplist[1]; p.nact_tab = L.nact[1]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<64, 128, 2, 2, true>(p, h->stream, lo);
    } else {
        p.mtiles = (L.Cout + 31) / 32; p.ntiles = (g.nrows + 255) / 256; p.taplist = L.taplist[2]; p.nact_tab = L.nact[2]; p.tl_stride = L.sparse_taps ? EV_MAX_TAPS : 0;
        launch_cfg<32, 256, 1, 4>(p, h->stream, lo);
    }
    HIPCHK(h, hipGetLastError());
    h->last_cfg = cfg;
    if (h->prof) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        const double valid_rows = (double)(g.nrows / g.S) * g.T;
        h->prof_flops += 2.0 * L.macs_per_row * valid_rows;
        h->prof_launches += 1;
        h->prof_recs.push_back({0, L.Cin, L.Cout, L.ntaps, g.nrows, cfg, (int)lean_ok(p), 2.0 * L.macs_per_row * valid_rows});
    }
    return 0;
}

// One fused ResBlock1 pair  y = c2(lrelu(c1(lrelu(x)))) + x  (resblock_pair_kernel) for C = 32 / 64.
int launch_pair(ev_handle* h, const ConvLayer& L1, const ConvLayer& L2, const float* X, float* Y, int C, const Geom& g, const Epi& e) {
    PairParams pp;
    memset(&pp, 0, sizeof pp);
    ConvParams& p = pp.c2;
    p.X = X; p.ldx = C; p.Cin = C; p.isplit_log2 = 31;
    p.W = L2.W; p.Mpad = L2.Mpad; p.Kpad = L2.Kpad; p.bias = L2.bias;
    p.Y = Y; p.ldy = C; p.Cout = C; p.osplit_log2 = 31; p.mmul = 1;
    p.nrows = g.nrows; p.S = g.S; p.P = g.P; p.T = g.T;
    p.ntaps = L2.ntaps; p.taplist = L2.taplist[0]; p.tl_stride = 0; p.nact_tab = nullptr;
    p.pro_lrelu = 1; p.pro_slope = 0.1f;
    p.scale = 1.f; p.R = X; p.ldr = C; p.accum = e.accum; p.div3 = e.div3; p.act2_lrelu = e.act2_lrelu; p.act2_slope = e.act2_slope;
    pp.W1 = L1.W; pp.W1x = L1.Wx; p.Wx = L2.Wx; pp.W1h = L1.Wh; pp.w1h_scale = L1.wh_scale; p.Wh = L2.Wh; p.wh_scale = L2.wh_scale; pp.W1q = L1.Wq; p.Wq = L2.Wq; pp.b1 = L1.bias; pp.taplist1 = L1.taplist[0]; pp.ntaps1 = L1.ntaps;
    pp.h1 = L1.halo_lo; pp.h2 = L2.halo_lo; pp.mid_slope = 0.1f;
    p.rmax = nullptr; p.yold = e.yold; p.ymax_mul = 1; h->amax_emitted = false;   // (R = X: the fp16 pair kernels bound the residual by their own tile maximum; p.ymax below)
    if ((double)g.nrows * C * 4.0 >= 4294967296.0) return fail(h, "tensor exceeds the 4 GiB buffer-addressing limit: split the batch");
    if (L1.sparse_taps || L2.sparse_taps || L1.Kpad != C || L2.Kpad != C || L1.Kpad != L2.Kpad || L1.Mpad != L2.Mpad || L1.halo_lo != L1.halo_hi ||
        L2.halo_lo != L2.halo_hi || 2 * pp.h2 > 16 || !L1.bias || !L2.bias)
        return fail(h, "launch_pair: unsupported layer pair");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
        }
        e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    static const bool no_lean = getenv("EV_NO_LEAN") != nullptr;
    const int lean = no_lean ? 0 : ((e.accum || e.div3 || e.act2_lrelu) ? 3 : 1);
    const int split_terms = h->split_terms;
    const bool split = split_terms > 0 && lean != 0 && L1.Wx && L2.Wx && !(e.force_cfg == 0);
    const bool h16 = split && split_terms == 16 && L1.Wh && L2.Wh;
    if (h16) {   // the fp16 build: two block-scaled fp16 planes per LDS row
        if (e.ymax && (!e.accum || e.yold) && h->use_amax) { p.ymax = e.ymax; h->amax_emitted = true; }   // (the residual's bound: the pair's own tile maximum)
        const int NT = C == 32 ? 256 : C == 64 ? 128 : 64, RSB = 4 * C + 16;
        if (C != 32 && C != 64 && C != 128) return fail(h, "launch_pair: C must be 32, 64 or 128");
        pp.out_rows = NT - 2 * pp.h2; p.mtiles = 1; p.ntiles = (g.nrows + pp.out_rows - 1) / pp.out_rows;
        const size_t smem = std::max((size_t)(NT + EV_HALO) * RSB + 64, (size_t)4 * 32 * 36 * sizeof(float));   // (+ 16 floats: the waves' maxima)
        const dim3 grid(p.ntiles);
        // EV_PAIR_Q=1: both K loops on v_mfma_f32_16x16x32_f16 (resblock_pair_h16q_kernel).  Measured null, not the default: same box, config 2 —
        // every pair shape within 1 % of the 32 x 32 x 16 form (C = 128 k3 3 % slower: 150 registers admit three workgroups per CU instead of four),
        // while conv_h16_kernel gains 3-9 % from the same change (profiles/r04_mfma_shape_ab.txt): the pairs' K loops are too short (C <= 128 deep)
        // for the MFMA phase to be what the power limit throttles.
        static const bool use_q = getenv("EV_PAIR_Q") && atoi(getenv("EV_PAIR_Q")) != 0;
        const bool q = use_q && L1.Wq && L2.Wq;
#define EV_PAIR_H16(WM, WN) do { \
            if (q && lean == 1) { ensure_dyn_smem<resblock_pair_h16q_kernel<WM, WN, 1>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_h16q_kernel<WM, WN, 1>), grid, dim3(256), smem, h->stream, pp); } \
            else if (q) { ensure_dyn_smem<resblock_pair_h16q_kernel<WM, WN, 3>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_h16q_kernel<WM, WN, 3>), grid, dim3(256), smem, h->stream, pp); } \
            else if (lean == 1) { ensure_dyn_smem<resblock_pair_h16_kernel<WM, WN, 1>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_h16_kernel<WM, WN, 1>), grid, dim3(256), smem, h->stream, pp); } \
            else { ensure_dyn_smem<resblock_pair_h16_kernel<WM, WN, 3>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_h16_kernel<WM, WN, 3>), grid, dim3(256), smem, h->stream, pp); } } while (0)
        if (C == 32) EV_PAIR_H16(1, 4); else if (C == 64) EV_PAIR_H16(2, 2); else EV_PAIR_H16(4, 1);
#undef EV_PAIR_H16
    } else
    if (split) {   // the bf16-split build: LDS rows hold all C channels as three bf16 planes
        const int NT = C == 32 ? 256 : C == 64 ? 128 : 64, RSB = 6 * C + 16;
        if (C != 32 && C != 64 && C != 128) return fail(h, "launch_pair: C must be 32, 64 or 128");
        pp.out_rows = NT - 2 * pp.h2; p.mtiles = 1; p.ntiles = (g.nrows + pp.out_rows - 1) / pp.out_rows;
        const size_t xs = (size_t)(NT + ((2 * pp.h1 + 7) & ~7)) * RSB, ys = (size_t)(NT + 16) * RSB, es = (size_t)4 * 32 * 36 * sizeof(float);
        size_t smem = std::max(xs, std::max(ys, es));
        {
            static const int wpc = getenv("EV_SPLIT_WPC") ? atoi(getenv("EV_SPLIT_WPC")) : 0;
            if (wpc > 0) smem = std::max(smem, (size_t)((160 * 1024 / (wpc + 1) + 1024) & ~255));
        }
        const dim3 grid(p.ntiles);
#define EV_PAIR_SPLIT_T(WM, WN, TT) do { \
            if (lean == 1) { ensure_dyn_smem<resblock_pair_split_kernel<WM, WN, 1, TT>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_split_kernel<WM, WN, 1, TT>), grid, dim3(256), smem, h->stream, pp); } \
            else { ensure_dyn_smem<resblock_pair_split_kernel<WM, WN, 3, TT>>(smem, h->device); hipLaunchKernelGGL((resblock_pair_split_kernel<WM, WN, 3, TT>), grid, dim3(256), smem, h->stream, pp); } } while (0)
#define EV_PAIR_SPLIT(WM, WN) do { if (split_terms == 3) EV_PAIR_SPLIT_T(WM, WN, 3); else EV_PAIR_SPLIT_T(WM, WN, 6); } while (0)   // (9: the six-product build)
        if (C == 32) EV_PAIR_SPLIT(1, 4); else if (C == 64) EV_PAIR_SPLIT(2, 2); else EV_PAIR_SPLIT(4, 1);
#undef EV_PAIR_SPLIT
#undef EV_PAIR_SPLIT_T
    } else if (C == 32) {
        constexpr int NT = 256;
        pp.out_rows = NT - 2 * pp.h2; p.mtiles = 1; p.ntiles = (g.nrows + pp.out_rows - 1) / pp.out_rows;
        // exact X-tile rows (NT + 2*h1) instead of NT + EV_HALO: 39 KB instead of 46 KB for k = 3 / 7 -> 4 workgroups per CU
        const size_t xs = (size_t)(NT + ((2 * pp.h1 + 7) & ~7)) * EV_LDK, ys = (size_t)(NT + 16) * EV_LDK, es = (size_t)4 * 32 * 36;
        const size_t smem = std::max(xs, std::max(ys, es)) * sizeof(float);
        if (lean == 1) hipLaunchKernelGGL((resblock_pair_kernel<1, 4, 1>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else if (lean == 3) hipLaunchKernelGGL((resblock_pair_kernel<1, 4, 3>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else hipLaunchKernelGGL((resblock_pair_kernel<1, 4, 0>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
    } else if (C == 64) {
        constexpr int NT = 128;
        pp.out_rows = NT - 2 * pp.h2; p.mtiles = 1; p.ntiles = (g.nrows + pp.out_rows - 1) / pp.out_rows;
        const size_t xs = (size_t)(NT + ((2 * pp.h1 + 7) & ~7)) * EV_LDK, ys = (size_t)2 * (NT + 16) * EV_LDK;
        const size_t smem = (xs > ys ? xs : ys) * sizeof(float);
        if (lean == 1) hipLaunchKernelGGL((resblock_pair_kernel<2, 2, 1>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else if (lean == 3) hipLaunchKernelGGL((resblock_pair_kernel<2, 2, 3>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else hipLaunchKernelGGL((resblock_pair_kernel<2, 2, 0>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
    } else if (C == 128) {
        constexpr int NT = 64;    // 4 waves = 4 channel tiles; 64 compute rows keep the 128-channel intermediate in 46 KB of LDS
        pp.out_rows = NT - 2 * pp.h2; p.mtiles = 1; p.ntiles = (g.nrows + pp.out_rows - 1) / pp.out_rows;
        const size_t xs = (size_t)(NT + ((2 * pp.h1 + 7) & ~7)) * EV_LDK, ys = (size_t)4 * (NT + 16) * EV_LDK;
        const size_t smem = (xs > ys ? xs : ys) * sizeof(float);
        if (lean == 1) hipLaunchKernelGGL((resblock_pair_kernel<4, 1, 1>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else if (lean == 3) hipLaunchKernelGGL((resblock_pair_kernel<4, 1, 3>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
        else hipLaunchKernelGGL((resblock_pair_kernel<4, 1, 0>), dim3(p.ntiles), dim3(256), smem, h->stream, pp);
    } else {
        return fail(h, "launch_pair: C must be 32, 64 or 128");
    }
    HIPCHK(h, hipGetLastError());
    h->last_cfg = (h16 ? 160 : split ? 140 : 100) + L2.ntaps;
    if (h->prof) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        const double valid_rows = (double)(g.nrows / g.S) * g.T;
        h->prof_flops += 2.0 * (L1.macs_per_row + L2.macs_per_row) * valid_rows;
        h->prof_launches += 1;
        h->prof_recs.push_back({1, C, C, L1.ntaps, g.nrows, (h16 ? 160 : split ? 140 : 100) + L2.ntaps, lean, 2.0 * (L1.macs_per_row + L2.macs_per_row) * valid_rows});
    }
    return 0;
}

// A whole ResBlock1 — three (dilated conv, conv) pairs with their residual adds — in one launch (resblock_chain_h16_kernel): the narrow levels under
// arithmetic setting 16, kernel sizes whose summed halos leave most of a tile to store.  chain_ok() is the gate; the caller falls back to three
// launch_pair calls.  EV_NO_CHAIN=1 switches it off (A/B).
inline int chain_halo(const ConvLayer* L1, const ConvLayer* L2) { int hs = 0; for (int m = 0; m < 3; ++m) hs += L1[m].halo_lo + L2[m].halo_lo; return hs; }
inline bool chain_ok(const ev_handle* h, const ConvLayer* L1, const ConvLayer* L2, int C) {
    static const bool off = getenv("EV_NO_CHAIN") != nullptr;
    static const int max_halo = getenv("EV_CHAIN_MAXHALO") ? atoi(getenv("EV_CHAIN_MAXHALO")) : 12;
    if (off || !h->use_chain || h->split_terms != 16 || (C != 32 && C != 64)) return false;
    int hb = 0;
    for (int m = 0; m < 3; ++m) {
        const ConvLayer &a = L1[m], &b = L2[m];
        if (!a.Wh || !b.Wh || !a.bias || !b.bias || a.sparse_taps || b.sparse_taps || a.Kpad != C || b.Kpad != C || a.Mpad != b.Mpad || a.Mpad != L1[0].Mpad ||
            a.halo_lo != a.halo_hi || b.halo_lo != b.halo_hi || a.ntaps != L1[0].ntaps || b.ntaps != L2[0].ntaps || a.ntaps > 64 || b.ntaps > 64) return false;
        hb = std::max(hb, std::max(a.halo_lo, b.halo_lo));
    }
    const int NT = C == 32 ? 256 : 128, hs = chain_halo(L1, L2);
    return hs <= max_halo && 2 * hb <= EV_HALO && NT - 2 * hs >= NT / 2;
}
int launch_chain(ev_handle* h, const ConvLayer* L1, const ConvLayer* L2, const float* X, float* Y, int C, const Geom& g, const Epi& e) {
    ChainParams cp;
    memset(&cp, 0, sizeof cp);
    ConvParams& p = cp.c2;
    p.X = X; p.ldx = C; p.Cin = C; p.isplit_log2 = 31;
    p.Mpad = L2[2].Mpad; p.Kpad = L2[2].Kpad;
    p.Y = Y; p.ldy = C; p.Cout = C; p.osplit_log2 = 31; p.mmul = 1;
    p.nrows = g.nrows; p.S = g.S; p.P = g.P; p.T = g.T;
    p.pro_lrelu = 1; p.pro_slope = 0.1f;
    p.scale = 1.f; p.R = nullptr; p.ldr = C; p.accum = e.accum; p.div3 = e.div3; p.act2_lrelu = e.act2_lrelu; p.act2_slope = e.act2_slope;
    p.rmax = nullptr; p.yold = e.yold; p.ymax_mul = 1; h->amax_emitted = false;
    if (e.ymax && (!e.accum || e.yold) && h->use_amax) { p.ymax = e.ymax; h->amax_emitted = true; }
    if ((double)g.nrows * C * 4.0 >= 4294967296.0) return fail(h, "tensor exceeds the 4 GiB buffer-addressing limit: split the batch");
    int hb = 0;
    double macs = 0;
    for (int m = 0; m < 3; ++m) {
        cp.W1h[m] = L1[m].Wh; cp.W2h[m] = L2[m].Wh; cp.w1_scale[m] = L1[m].wh_scale; cp.w2_scale[m] = L2[m].wh_scale;
        cp.b1[m] = L1[m].bias; cp.b2[m] = L2[m].bias; cp.tl1[m] = L1[m].taplist[0]; cp.tl2[m] = L2[m].taplist[0];
        hb = std::max(hb, std::max(L1[m].halo_lo, L2[m].halo_lo));
        macs += L1[m].macs_per_row + L2[m].macs_per_row;
    }
    cp.ntaps1 = L1[0].ntaps; cp.ntaps2 = L2[0].ntaps; cp.hb = hb; cp.halo = chain_halo(L1, L2); cp.mid_slope = 0.1f;
    const int NT = C == 32 ? 256 : 128, RSB = 4 * C + 16;
    cp.out_rows = NT - 2 * cp.halo; p.mtiles = 1; p.ntiles = (g.nrows + cp.out_rows - 1) / cp.out_rows;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
        }
        e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    const int lean = (e.accum || e.div3 || e.act2_lrelu) ? 3 : 1;
    const size_t smem = std::max((size_t)(NT + EV_HALO) * RSB + 64, (size_t)4 * 32 * 36 * sizeof(float));
    const dim3 grid(p.ntiles);
#define EV_CHAIN(WM, WN) do { \
        if (lean == 1) { ensure_dyn_smem<resblock_chain_h16_kernel<WM, WN, 1>>(smem, h->device); hipLaunchKernelGGL((resblock_chain_h16_kernel<WM, WN, 1>), grid, dim3(256), smem, h->stream, cp); } \
        else { ensure_dyn_smem<resblock_chain_h16_kernel<WM, WN, 3>>(smem, h->device); hipLaunchKernelGGL((resblock_chain_h16_kernel<WM, WN, 3>), grid, dim3(256), smem, h->stream, cp); } } while (0)
    if (C == 32) EV_CHAIN(1, 4); else EV_CHAIN(2, 2);
#undef EV_CHAIN
    HIPCHK(h, hipGetLastError());
    h->last_cfg = 180 + L2[0].ntaps;
    if (h->prof) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        const double valid_rows = (double)(g.nrows / g.S) * g.T;
        h->prof_flops += 2.0 * macs * valid_rows;
        h->prof_launches += 1;
        h->prof_recs.push_back({1, C, C, L1[0].ntaps, g.nrows, 180 + L2[0].ntaps, lean, 2.0 * macs * valid_rows});
    }
    return 0;
}

int launch_gn(ev_handle* h, const float* X, int ldx, float* Y, int ldy, const float* gamma, const float* beta, const float* rowmask,
              const float* temb, const float* R, int ldr, const Geom& g, int C, int mode, const float* part = nullptr) {
    GNParams p;
    p.X = X; p.ldx = ldx; p.Y = Y; p.ldy = ldy; p.gamma = gamma; p.beta = beta; p.rowmask = rowmask; p.temb = temb; p.R = R; p.ldr = ldr;
    p.S = g.S; p.P = g.P; p.T = g.T; p.CG = C / 8; p.mode = mode; p.eps = 1e-5f;
    if (p.CG != 32) return fail(h, "groupnorm kernel expects 32 channels per group, got %d", p.CG);
    // one utterance whose producer (the conv launched just before) left per-tile statistics: a workgroup per (32 frames, group)
    if (part && h->gn_stats_tiles > 0 && g.nrows == g.S && h->gn_stats_tiles == (g.nrows + 31) / 32)
        hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(h->gn_stats_tiles, 8), dim3(256), 0, h->stream, p, part, h->gn_stats_tiles);
    // small batches: 1024 threads per (utterance, group) shorten the per-workgroup latency chain (8 workgroups at B = 1)
    else if (g.nrows / g.S < 32) hipLaunchKernelGGL(groupnorm_mish_kernel<1024>, dim3(g.nrows / g.S, 8), dim3(1024), 0, h->stream, p);
    else {
        // 512 threads per (utterance, group) slab: the workgroup's serial chain (load, two reductions, apply) is half as long as
        // with 256 (tools/decode_time.py, batch 64: 41.07 -> 40.60 ms per decode with 512, 40.94 with 1024); EV_GN_THREADS=<256|512|1024>
        // for A/B runs
        static const int gnt = getenv("EV_GN_THREADS") ? atoi(getenv("EV_GN_THREADS")) : 512;
        // EV_GN_PRE=1: residual / mask rows requested together with the slab (one memory round trip less per workgroup) — measured
        // SLOWER (40.8 -> 41.3 ms): 141 registers leave one 512-thread workgroup per CU where the plain build keeps two
        static const bool gnpre = getenv("EV_GN_PRE") ? atoi(getenv("EV_GN_PRE")) != 0 : false;
        if (gnt == 512 && gnpre) hipLaunchKernelGGL((groupnorm_mish_kernel<512, true>), dim3(g.nrows / g.S, 8), dim3(512), 0, h->stream, p);
        else if (gnt == 512) hipLaunchKernelGGL(groupnorm_mish_kernel<512>, dim3(g.nrows / g.S, 8), dim3(512), 0, h->stream, p);
        else if (gnt == 1024) hipLaunchKernelGGL(groupnorm_mish_kernel<1024>, dim3(g.nrows / g.S, 8), dim3(1024), 0, h->stream, p);
        else hipLaunchKernelGGL(groupnorm_mish_kernel<256>, dim3(g.nrows / g.S, 8), dim3(256), 0, h->stream, p);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int launch_ln(ev_handle* h, const float* X, int ldx, float* Y, int ldy, const float* gamma, const float* beta, const Geom& g) {
    LNParams p;
    p.X = X; p.ldx = ldx; p.Y = Y; p.ldy = ldy; p.gamma = gamma; p.beta = beta; p.nrows = g.nrows; p.S = g.S; p.P = g.P; p.T = g.T; p.eps = 1e-5f;
    hipLaunchKernelGGL(layernorm256_kernel, dim3((g.nrows + 3) / 4), dim3(256), 0, h->stream, p);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// LayerNorm + feed-forward (mode 0) or LayerNorm + QKV projection (mode 1) of one transformer block in one launch
// (ln_mlp_kernel).  Counted as ONE conv launch of the dominant-kernel family by the profiling hooks (its FLOPs are those of
// the linears it contains).
int launch_mlp(ev_handle* h, int mode, const float* X, const float* ln_g, const float* ln_b, const ConvLayer& L1, const ConvLayer* L2,
               const float* alpha, const float* binv, const float* R, const float* rowmask, float* Y, int ldy, const Geom& g,
               const float* qkv_scale = nullptr, int* qkv_packed = nullptr) {
    // qkv_scale (mode 1): the caller's attention can take q / k / v as fp16 piece pairs times these three powers of two; *qkv_packed tells
    // whether this launch wrote them so (only ln_qkv_h16_kernel does)
    if (qkv_packed) *qkv_packed = 0;
    MlpParams mp;
    memset(&mp, 0, sizeof mp);
    ConvParams& p = mp.ep;
    const ConvLayer& Lout = mode == 0 ? *L2 : L1;
    p.Y = Y; p.ldy = ldy; p.Cout = Lout.Cout; p.bias = mode == 0 ? Lout.bias : nullptr; p.osplit_log2 = 31; p.isplit_log2 = 31; p.mmul = 1;
    p.nrows = g.nrows; p.S = g.S; p.P = g.P; p.T = g.T; p.scale = 1.f;
    p.R = R; p.ldr = 256; p.rowmask = rowmask; p.mask2 = (mode == 0 && rowmask) ? 1 : 0;
    mp.X = X; mp.ldx = 256; mp.ln_g = ln_g; mp.ln_b = ln_b; mp.ln_eps = 1e-5f;
    if (!L1.bias && !h->zeros) {   // a layer without bias reads zeros: the kernel's bias loads are unconditional
        HIPCHK(h, hipMalloc((void**)&h->zeros, 4096 * sizeof(float)));
        HIPCHK(h, hipMemsetAsync(h->zeros, 0, 4096 * sizeof(float), h->stream));
        h->owned.push_back(h->zeros);
    }
    if (L1.Mpad > 4096) return fail(h, "launch_mlp: hidden width %d > 4096", L1.Mpad);
    mp.W1 = L1.W; mp.b1 = L1.bias ? L1.bias : h->zeros; mp.M1 = L1.Mpad; mp.alpha = alpha; mp.binv = binv; mp.W2 = L2 ? L2->W : nullptr;
    if (L1.Cin != 256 || L1.Kpad != 256 || L1.ntaps != 1 || (L1.Mpad % 128) || L1.Cout != L1.Mpad || (ldy & 3))
        return fail(h, "launch_mlp: first linear must be 256 -> multiple of 128 (got %d -> %d)", L1.Cin, L1.Cout);
    if (mode == 0 && (!L2 || L2->Cout != 256 || L2->Mpad != 256 || L2->Cin != L1.Cout || L2->Kpad != L1.Mpad || L2->ntaps != 1 || !L2->bias || !alpha || !binv))
        return fail(h, "launch_mlp: second linear must be %d -> 256 with bias", L1.Cout);
    if ((double)g.nrows * std::max(ldy, 256) * 4.0 >= 4294967296.0) return fail(h, "tensor exceeds the 4 GiB buffer-addressing limit: split the batch");
    if (g.S < 4 && g.nrows > 1) return fail(h, "launch_mlp: utterance stride %d < 4 rows is not supported by the lean row walk", g.S);
    const int nchunk = L1.Mpad / 128;
    {   // the feed-forward of a large batch on the bf16 pipe: 64-row tiles, one persistent workgroup per CU (ln_mlp_split_kernel)
        const int split_terms = h->split_terms;
        static const bool no_mlp_split = getenv("EV_NO_MLP_SPLIT") != nullptr;
        const int nt64 = (g.nrows + 63) / 64;
        if (mode == 0 && (split_terms == 6 || split_terms == 16) && !no_mlp_split && L1.Wx && L2->Wx && h->sk_balance && h->ncu > 0 && h->ncu <= EV_SK_MAXWG && nt64 >= h->ncu) {
            if (ensure_sk(h)) return 1;
            const long U = (long)nt64 * nchunk;
            const int grid = h->ncu;
            const bool h16 = split_terms == 16 && L1.Wh && L2->Wh;
            mp.W1x = L1.Wx; mp.W2x = L2->Wx; mp.ntiles = nt64;
            mp.W1h = L1.Wh; mp.W2h = L2->Wh; mp.w1_scale = L1.wh_scale; mp.w2_scale = L2->wh_scale;
            mp.sk.q = (int)(U / grid); mp.sk.r = (int)(U % grid); mp.sk.spin_limit = h->sk_spin;
            mp.sk.ctrl = h->sk_ctrl; mp.sk.flags = h->sk_ctrl + 16; mp.sk.part = h->sk_part; mp.sk.part_floats = EV_SK_PART_FLOATS;
            if (h16 && h->sk_steal) { mp.sk.claims = h->sk_ctrl + 16 + EV_SK_MAXWG; mp.sk.seq = ++h->sk_seq; }
            if (L1.Mpad > 1024) return fail(h, "launch_mlp: hidden width %d > 1024 (LDS table of the SnakeBeta vectors)", L1.Mpad);
            const size_t smem = h16 ? (size_t)64 * (4 * 256 + 16) + (size_t)64 * (4 * 128 + 16) + 16 + (size_t)2 * L1.Mpad * sizeof(float) + 64
                                    : (size_t)64 * (6 * 256 + 16) + (size_t)64 * (6 * 128 + 16) + 16 + (size_t)2 * L1.Mpad * sizeof(float);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->prof) {
                if (h->ev_used + 2 > h->ev_pool.size()) {
                    for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
                }
                e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
                HIPCHK(h, hipEventRecord(e0, h->stream));
            }
            if (h16) ensure_dyn_smem<ln_mlp_h16_kernel<0>>(smem, h->device); else ensure_dyn_smem<ln_mlp_split_kernel<6>>(smem, h->device);
            static const char* stamp_file = getenv("EV_MLP_STAMPS");     // diagnostic: phase stamps of a few workgroups of the first launches
            static int stamped = 0;
            if (stamp_file && *stamp_file && stamped < 2) {
                unsigned long long* d = nullptr;
                HIPCHK(h, hipMalloc((void**)&d, (size_t)grid * 32 * sizeof(unsigned long long)));
                HIPCHK(h, hipMemsetAsync(d, 0, (size_t)grid * 32 * sizeof(unsigned long long), h->stream));
                mp.ep.stamps = d;
                if (h16) hipLaunchKernelGGL((ln_mlp_h16_kernel<0>), dim3(grid), dim3(256), smem, h->stream, mp);
                else hipLaunchKernelGGL((ln_mlp_split_kernel<6>), dim3(grid), dim3(256), smem, h->stream, mp);
                HIPCHK(h, hipStreamSynchronize(h->stream));
                std::vector<unsigned long long> st((size_t)grid * 32);
                HIPCHK(h, hipMemcpy(st.data(), d, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                hipFree(d);
                mp.ep.stamps = nullptr;
                if (FILE* f = fopen(stamp_file, "a")) {
                    fprintf(f, "## ln_mlp_split_kernel rows=%d: %d workgroups, q=%d r=%d; per workgroup: us between consecutive stamps (start | staged+LN | per chunk: phase 1 | phase 2 + SnakeBeta | planes written ...)\n", g.nrows, grid, mp.sk.q, mp.sk.r);
                    for (int w : {0, 1, 100, 255}) {
                        if (w >= grid) continue;
                        fprintf(f, "  wg %3d:", w);
                        for (int k = 1; k < 32 && st[(size_t)w * 32 + k]; ++k) fprintf(f, " %.2f", (double)(st[(size_t)w * 32 + k] - st[(size_t)w * 32 + k - 1]) / 100.0);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                }
                ++stamped;
            } else if (h16) hipLaunchKernelGGL((ln_mlp_h16_kernel<0>), dim3(grid), dim3(256), smem, h->stream, mp);
            else hipLaunchKernelGGL((ln_mlp_split_kernel<6>), dim3(grid), dim3(256), smem, h->stream, mp);
            HIPCHK(h, hipGetLastError());
            if (h->prof) {
                HIPCHK(h, hipEventRecord(e1, h->stream));
                const double valid_rows = (double)(g.nrows / g.S) * g.T;
                const double fl = 2.0 * (L1.macs_per_row + L2->macs_per_row) * valid_rows;
                h->prof_flops += fl;
                h->prof_launches += 1;
                h->prof_recs.push_back({2, 256, Lout.Cout, 1, g.nrows, h16 ? 121 : 120, 1, fl});
            }
            return 0;
        }
    }
    {   // LayerNorm + QKV of a large batch on the fp16 pipe: one workgroup per 64-row tile (ln_qkv_h16_kernel)
        static const bool no_qkv_h16 = getenv("EV_NO_QKV_H16") != nullptr;
        const int nt64 = (g.nrows + 63) / 64;
        if (mode == 1 && h->split_terms == 16 && !no_qkv_h16 && L1.Wh && L1.Mpad == 384 && !R && !rowmask && h->ncu > 0 && nt64 >= h->ncu) {
            mp.W1h = L1.Wh; mp.w1_scale = L1.wh_scale; mp.ntiles = nt64;
            if (qkv_scale && qkv_packed && qkv_scale[0] > 0.f && qkv_scale[1] > 0.f && qkv_scale[2] > 0.f) {
                mp.qkv_pack = 1; *qkv_packed = 1;
                for (int i = 0; i < 3; ++i) mp.qkv_scale[i] = qkv_scale[i];
            }
            const size_t smem = (size_t)64 * (4 * 256 + 16) + 32;   // (+ 8 floats: the waves' maxima and their finite-only repeat)
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (h->prof) {
                if (h->ev_used + 2 > h->ev_pool.size()) {
                    for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
                }
                e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
                HIPCHK(h, hipEventRecord(e0, h->stream));
            }
            ensure_dyn_smem<ln_qkv_h16_kernel<0>>(smem, h->device); ensure_dyn_smem<ln_qkv_h16_kernel<1>>(smem, h->device);
            static const int qdbg = getenv("EV_QKV_DBG") ? atoi(getenv("EV_QKV_DBG")) : 0;
            static const char* stamp_file = getenv("EV_QKV_STAMPS");     // diagnostic: phase stamps of a few workgroups of the first launches
            static int stamped = 0;
            if (stamp_file && *stamp_file && stamped < 4) {
                unsigned long long* d = nullptr;
                HIPCHK(h, hipMalloc((void**)&d, (size_t)nt64 * 8 * sizeof(unsigned long long)));
                HIPCHK(h, hipMemsetAsync(d, 0, (size_t)nt64 * 8 * sizeof(unsigned long long), h->stream));
                mp.ep.stamps = d;
                { if (qdbg == 1) hipLaunchKernelGGL(ln_qkv_h16_kernel<1>, dim3(nt64), dim3(256), smem, h->stream, mp); else hipLaunchKernelGGL(ln_qkv_h16_kernel<0>, dim3(nt64), dim3(256), smem, h->stream, mp); }
                HIPCHK(h, hipStreamSynchronize(h->stream));
                std::vector<unsigned long long> st((size_t)nt64 * 8);
                HIPCHK(h, hipMemcpy(st.data(), d, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                hipFree(d);
                mp.ep.stamps = nullptr;
                if (FILE* f = fopen(stamp_file, "a")) {
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (int w = 0; w < nt64; ++w) if (st[(size_t)w * 8]) { t0 = std::min(t0, st[(size_t)w * 8]); t1 = std::max(t1, st[(size_t)w * 8 + 5]); }
                    fprintf(f, "## ln_qkv_h16_kernel rows=%d: %d workgroups, first start -> last end %.2f us; per workgroup: start offset | staged+LN | split+barrier | K loop | epilogue 1 | epilogue 2 (us)\n", g.nrows, nt64, (double)(t1 - t0) / 100.0);
                    for (int w : {0, 1, 2, 100, 255, 256, 300, 511, 512, 519}) {
                        if (w >= nt64 || !st[(size_t)w * 8]) continue;
                        fprintf(f, "  wg %3d: +%.2f |", w, (double)(st[(size_t)w * 8] - t0) / 100.0);
                        for (int k = 1; k < 6; ++k) fprintf(f, " %.2f", (double)(st[(size_t)w * 8 + k] - st[(size_t)w * 8 + k - 1]) / 100.0);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                }
                ++stamped;
            } else { if (qdbg == 1) hipLaunchKernelGGL(ln_qkv_h16_kernel<1>, dim3(nt64), dim3(256), smem, h->stream, mp); else hipLaunchKernelGGL(ln_qkv_h16_kernel<0>, dim3(nt64), dim3(256), smem, h->stream, mp); }
            HIPCHK(h, hipGetLastError());
            h->last_cfg = 122;
            if (h->prof) {
                HIPCHK(h, hipEventRecord(e1, h->stream));
                const double valid_rows = (double)(g.nrows / g.S) * g.T;
                const double fl = 2.0 * L1.macs_per_row * valid_rows;
                h->prof_flops += fl;
                h->prof_launches += 1;
                h->prof_recs.push_back({3, 256, Lout.Cout, 1, g.nrows, 122, 1, fl});
            }
            return 0;
        }
    }
    const int ntiles = (g.nrows + 31) / 32;
    size_t smem = (size_t)(32 * 260 + 4 * 32 * 36 + 4) * sizeof(float);
    // Balanced persistent grid (SkCtl): three workgroups per CU — the LDS request is padded so that exactly three fit, i.e. every
    // CU holds the same number of them — each taking an equal share of the (tile, 128-wide chunk) units.  From one tile per CU up;
    // below that the launch is a latency chain per workgroup and keeps one tile each.
    int grid = ntiles;
    mp.ntiles = ntiles;
    mp.sk.q = nchunk; mp.sk.r = 0; mp.sk.spin_limit = h->sk_spin;
    const int wpc = h->sk_wgs;
    const bool big = ntiles >= h->ncu && (long)ntiles * nchunk >= (long)wpc * h->ncu;
    const bool spread = !big && h->sk_spread && ntiles < h->ncu && (long)ntiles * nchunk >= 2;   // small batches: one or a few units per workgroup
    if (h->sk_balance && h->ncu > 0 && wpc * h->ncu <= EV_SK_MAXWG && (big || spread)) {
        if (ensure_sk(h)) return 1;
        const long U = (long)ntiles * nchunk;
        grid = (int)std::min<long>((long)wpc * h->ncu, U);
        mp.sk.q = (int)(U / grid); mp.sk.r = (int)(U % grid);
        mp.sk.ctrl = h->sk_ctrl; mp.sk.flags = h->sk_ctrl + 16; mp.sk.part = h->sk_part; mp.sk.part_floats = EV_SK_PART_FLOATS;
        // three per CU: the LDS request is padded so that exactly three fit.  Two per CU (the default): the 256-register build admits
        // exactly two workgroups per CU by itself, and the un-padded request leaves LDS for the workgroups of another stream (the
        // vocoder of the previous batch in the pipelined schedule) instead of locking them out of the CU.
        if (wpc != 2) smem = std::max(smem, (size_t)((160 * 1024 / wpc) & ~255));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
        }
        e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    // the balanced grid at two workgroups per CU has its own build (256 registers, no spills); so has every launch of at most two
    // tiles per CU (small batches: one tile per workgroup, at most two of them on a CU)
    const bool two = (mp.sk.ctrl != nullptr && wpc == 2) || (mp.sk.ctrl == nullptr && h->ncu > 0 && ntiles <= 2 * h->ncu);
    if (mode == 0 && two) { ensure_dyn_smem<ln_mlp_kernel<0, 2>>(smem, h->device); hipLaunchKernelGGL((ln_mlp_kernel<0, 2>), dim3(grid), dim3(256), smem, h->stream, mp); }
    else if (mode == 0) { ensure_dyn_smem<ln_mlp_kernel<0, 3>>(smem, h->device); hipLaunchKernelGGL((ln_mlp_kernel<0, 3>), dim3(grid), dim3(256), smem, h->stream, mp); }
    else if (two) { ensure_dyn_smem<ln_mlp_kernel<1, 2>>(smem, h->device); hipLaunchKernelGGL((ln_mlp_kernel<1, 2>), dim3(grid), dim3(256), smem, h->stream, mp); }
    else { ensure_dyn_smem<ln_mlp_kernel<1, 3>>(smem, h->device); hipLaunchKernelGGL((ln_mlp_kernel<1, 3>), dim3(grid), dim3(256), smem, h->stream, mp); }
    HIPCHK(h, hipGetLastError());
    if (h->prof) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        const double valid_rows = (double)(g.nrows / g.S) * g.T;
        const double fl = 2.0 * (L1.macs_per_row + (L2 ? L2->macs_per_row : 0.0)) * valid_rows;
        h->prof_flops += fl;
        h->prof_launches += 1;
        h->prof_recs.push_back({2 + mode, 256, Lout.Cout, 1, g.nrows, 20 + mode, 1, fl});
    }
    return 0;
}

// `part`: scratch for the split-key small-launch build, 4 * rows * (H*64 + 2*H) floats (null: always the one-launch kernel)
constexpr int EV_ATTN_MAXPARTS = 16;   // split-key attention (small launches): parts a query tile's key tiles may be spread over,
constexpr int EV_ATTN_MAXROWS = 8192;  // and the most rows such a launch has (<= 64 workgroups of 128 queries, plus padding)
int launch_attn(ev_handle* h, const float* QKV, int ld, float* O, int ldo, const float* rowmask, const Geom& g, int H, float* part = nullptr) {
    AttnParams p;
    p.QKV = QKV; p.ld = ld; p.O = O; p.ldo = ldo; p.rowmask = rowmask; p.S = g.S; p.P = g.P; p.T = g.T; p.H = H; p.scale = 0.125f;
    const int nwg = ((g.T + 127) / 128) * H * (g.nrows / g.S), nkt = (g.T + 31) / 32;
    static const bool no_sk = getenv("EV_NO_ATTN_SK") != nullptr;
    if (!no_sk && part && nwg <= 64 && nkt >= 4 && g.nrows <= EV_ATTN_MAXROWS) {   // far fewer workgroups than CUs: the key tiles of a query tile go to KS workgroups
        AttnPartParams pp;
        pp.a = p; pp.rows = g.nrows;
        static const int tpw = getenv("EV_ATTN_TPW") ? std::max(1, atoi(getenv("EV_ATTN_TPW"))) : 2;   // key tiles per workgroup (A/B)
        pp.KS = std::min(EV_ATTN_MAXPARTS, (nkt + tpw - 1) / tpw);
        pp.PO = part; pp.PML = part + (size_t)EV_ATTN_MAXPARTS * g.nrows * H * 64;
        hipLaunchKernelGGL(attention_part_kernel, dim3((g.T + 127) / 128, H, (g.nrows / g.S) * pp.KS), dim3(256), 0, h->stream, pp);
        const long tot = (long)g.nrows * H * 16;
        hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, pp);
    } else {
        hipLaunchKernelGGL(attention_kernel, dim3((g.T + 127) / 128, H, g.nrows / g.S), dim3(256), 0, h->stream, p);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// attn_out_h16_kernel adds the frame mask to the scores as one more MFMA: in accumulator units the mask is m 8 sq sk, an exact power of two that
// must split into two fp16 NUMBERS mask_a mask_b (normal range 2^-14 .. 2^15).  False: it does not (scales far outside anything a checkpoint
// produces) — the caller keeps the fp32 form.
bool attn_mask_split(const float* qkv_scale, float* mask_a, float* mask_b) {
    int eq = 0, ek = 0;
    if (!(qkv_scale[0] > 0.f) || !(qkv_scale[1] > 0.f) || std::frexp(qkv_scale[0], &eq) != 0.5f || std::frexp(qkv_scale[1], &ek) != 0.5f) return false;
    const int e = 3 + (eq - 1) + (ek - 1);              // log2(8 sq sk)
    const int ea = std::min(15, std::max(-14, e / 2)), eb = e - ea;
    if (eb < -14 || eb > 15) return false;
    *mask_a = (float)std::ldexp(1.0, ea); *mask_b = (float)std::ldexp(1.0, eb);
    return true;
}

// The powers of two by which ln_qkv_h16_kernel scales q, k, v before it splits them into fp16 pieces.  A LayerNorm output obeys
// |y_c| <= sqrt(C - 1) |gamma_c| + |beta_c| whatever its input (the normalised deviation of one of C numbers is at most sqrt(C - 1)), so
// |(W y)_j| <= sum_c |W_jc| (sqrt(C - 1) |gamma_c| + |beta_c|) =: bound_j.  scale = the power of two that maps max_j bound_j into (16384, 32768]:
// no finite input overflows fp16's 65504, and a value v keeps max(2^-22 |v|, 2^-25 / scale) — an absolute floor 2^-40 below the bound.
// A missing or non-finite operand leaves the scales at 0: the caller keeps the fp32 form.
void qkv_pack_scales(const HostTensor* g, const HostTensor* b, std::initializer_list<const HostTensor*> ws, float* out) {
    out[0] = out[1] = out[2] = 0.f;
    if (!g || !b) return;
    float tmp[3]; int t = 0;
    for (const HostTensor* W : ws) {
        if (!W || W->ndim != 2 || t >= 3) return;
        const int M = (int)W->shape[0], C = (int)W->shape[1];
        if ((int)g->shape[0] != C || (int)b->shape[0] != C) return;
        const double rc = std::sqrt((double)(C - 1));
        double worst = 0.0;
        for (int j = 0; j < M; ++j) {
            double acc = 0.0;
            for (int c = 0; c < C; ++c) acc += std::fabs((double)W->p[(size_t)j * C + c]) * (rc * std::fabs((double)g->p[c]) + std::fabs((double)b->p[c]));
            worst = std::max(worst, acc);
        }
        if (!(worst > 0.0) || !std::isfinite(worst)) return;
        int e = (int)std::floor(std::log2(32768.0 / worst));
        e = std::min(40, std::max(-40, e));
        tmp[t++] = (float)std::ldexp(1.0, e);
    }
    float ma, mb;
    if (t == 3 && attn_mask_split(tmp, &ma, &mb)) for (int i = 0; i < 3; ++i) out[i] = tmp[i];
}

// attention of both heads + output projection + residual in one launch (attn_out_kernel): H += Wout . attn(QKV) + bout, in place.
// Counted as ONE launch of the dominant-kernel family by the profiling hooks (its FLOPs: QK^T and PV of both heads + the projection).
inline bool attn_out_ok(const ev_handle* h, const ConvLayer& Lo, const Geom& g, int H) {
    const long wgs = (long)((g.T + 31) / 32) * (g.nrows / g.S);
    return h->fuse_attn && H == 2 && Lo.Cin == 128 && Lo.Kpad == 128 && Lo.Cout == 256 && Lo.Mpad == 256 && Lo.ntaps == 1 && !Lo.sparse_taps && Lo.bias &&
           g.S >= 4 && h->ncu > 0 && wgs >= h->ncu / 2;
}
// qkv_scale non-null: QKV holds fp16 piece pairs times these powers of two -> attn_out_h16_kernel
int launch_attn_out(ev_handle* h, const float* QKV, int ld, const ConvLayer& Lo, float* Hid, int ldh, const float* rowmask, const Geom& g, int H,
                    const float* qkv_scale = nullptr) {
    AttnOutParams p;
    memset(&p, 0, sizeof p);
    const bool h16 = qkv_scale != nullptr;
    if (h16) {
        p.inv_sq = 1.0f / qkv_scale[0]; p.inv_sk = 1.0f / qkv_scale[1]; p.inv_sv = 1.0f / qkv_scale[2]; p.Wouth = Lo.Wh; p.wo_scale = Lo.wh_scale;
        if (!attn_mask_split(qkv_scale, &p.mask_a, &p.mask_b) || !Lo.Wh) return fail(h, "launch_attn_out: q / k scales %g, %g outside the fp16 form's range", qkv_scale[0], qkv_scale[1]);
    }
    const int B = g.nrows / g.S;
    p.QKV = QKV; p.ld = ld; p.rowmask = rowmask; p.Wout = Lo.W; p.S = g.S; p.P = g.P; p.T = g.T; p.B = B; p.nq = (g.T + 31) / 32; p.scale = 0.125f;
    p.xcd_map = (B % 8 == 0) ? 1 : 0;
    {   // a short last query tile (<= AO_TAILQ queries) goes to one small 4 x 4-block workgroup per utterance, dispatched first
        static const bool no_tail = getenv("EV_ATTN_NO_TAIL") != nullptr;
        const int last = g.T - 32 * (p.nq - 1);
        if (!no_tail && p.nq >= 2 && last <= AO_TAILQ) { p.ntail = last; p.nq -= 1; }
    }
    ConvParams& e = p.ep;
    e.Y = Hid; e.ldy = ldh; e.Cout = Lo.Cout; e.bias = Lo.bias; e.R = Hid; e.ldr = ldh; e.osplit_log2 = 31; e.isplit_log2 = 31; e.mmul = 1; e.scale = 1.f;
    e.nrows = g.nrows; e.S = g.S; e.P = g.P; e.T = g.T;
    if ((ldh & 3) || (ld & 3) || ((size_t)Lo.bias & 15)) return fail(h, "launch_attn_out: unaligned operand");
    if ((double)g.nrows * std::max(ld, ldh) * 4.0 >= 4294967296.0) return fail(h, "tensor exceeds the 4 GiB buffer-addressing limit: split the batch");
    const size_t smem = h16 ? (size_t)4 * AOH_WB : (size_t)4 * 2 * 32 * AO_LDK * sizeof(float);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) {
        if (h->ev_used + 2 > h->ev_pool.size()) {
            for (int i = 0; i < 64; ++i) { hipEvent_t ev; HIPCHK(h, hipEventCreate(&ev)); h->ev_pool.push_back(ev); }
        }
        e0 = h->ev_pool[h->ev_used++]; e1 = h->ev_pool[h->ev_used++];
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    if (h16) ensure_dyn_smem<attn_out_h16_kernel>(smem, h->device); else ensure_dyn_smem<attn_out_kernel>(smem, h->device);
    auto launch = [&]() {
        if (h16) hipLaunchKernelGGL(attn_out_h16_kernel, dim3((unsigned)(p.nq * B + (p.ntail > 0 ? B : 0))), dim3(256), smem, h->stream, p);
        else hipLaunchKernelGGL(attn_out_kernel, dim3((unsigned)(p.nq * B + (p.ntail > 0 ? B : 0))), dim3(256), smem, h->stream, p);
    };
    static const char* stamp_file = getenv("EV_ATTN_STAMPS");      // diagnostic: per-workgroup phase stamps of the first launches, appended to this file
    static int stamped = 0;
    const int nwg = p.nq * B + (p.ntail > 0 ? B : 0);
    if (stamp_file && *stamp_file && stamped < 8 && B >= 32) {
        unsigned long long* d = nullptr;
        HIPCHK(h, hipMalloc((void**)&d, (size_t)nwg * 6 * sizeof(unsigned long long)));
        HIPCHK(h, hipMemsetAsync(d, 0, (size_t)nwg * 6 * sizeof(unsigned long long), h->stream));
        p.stamps = d;
        launch();
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<unsigned long long> st((size_t)nwg * 6);
        HIPCHK(h, hipMemcpy(st.data(), d, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        hipFree(d);
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < nwg; ++i) t0 = std::min(t0, st[(size_t)i * 6]);
        if (FILE* f = fopen(stamp_file, "a")) {
            const char* names[6] = {"start", "first key tile", "key loop done", "merged", "projection done", "end"};
            fprintf(f, "## %s T=%d B=%d: %d workgroups; us since the first workgroup started (100 MHz s_memrealtime)\n", h16 ? "attn_out_h16_kernel" : "attn_out_kernel", g.T, B, nwg);
            for (int k = 0; k < 6; ++k) {
                std::vector<double> v(nwg);
                for (int i = 0; i < nwg; ++i) v[i] = (double)(st[(size_t)i * 6 + k] - t0) / 100.0;
                std::sort(v.begin(), v.end());
                fprintf(f, "  %-16s min %7.1f  p10 %7.1f  p50 %7.1f  p90 %7.1f  max %7.1f\n", names[k], v[0], v[nwg / 10], v[nwg / 2], v[(size_t)nwg * 9 / 10], v[nwg - 1]);
            }
            for (int k = 0; k < 5; ++k) {              // per-workgroup phase durations
                std::vector<double> v(nwg);
                for (int i = 0; i < nwg; ++i) v[i] = (double)(st[(size_t)i * 6 + k + 1] - st[(size_t)i * 6 + k]) / 100.0;
                std::sort(v.begin(), v.end());
                fprintf(f, "  phase %-16s -> %-16s p10 %6.1f  p50 %6.1f  p90 %6.1f us\n", names[k], names[k + 1], v[nwg / 10], v[nwg / 2], v[(size_t)nwg * 9 / 10]);
            }
            if (p.ntail > 0) {                          // the short-tile workgroups (block ids 0 .. B-1) on their own
                std::vector<double> v0(B), v1(B);
                for (int i = 0; i < B; ++i) { v0[i] = (double)(st[(size_t)i * 6] - t0) / 100.0; v1[i] = (double)(st[(size_t)i * 6 + 5] - t0) / 100.0; }
                std::sort(v0.begin(), v0.end()); std::sort(v1.begin(), v1.end());
                fprintf(f, "  short last tile (%d queries, 4 x 4 blocks): start p50 %6.1f max %6.1f   end p50 %6.1f max %6.1f us\n", p.ntail, v0[B / 2], v0[B - 1], v1[B / 2], v1[B - 1]);
            }
            fclose(f);
        }
        ++stamped;
        p.stamps = nullptr;
    } else launch();
    HIPCHK(h, hipGetLastError());
    if (h->prof) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        const double fl = (double)B * ((double)H * 4.0 * g.T * (double)g.T * 64.0 + 2.0 * Lo.macs_per_row * g.T);
        h->prof_flops += fl;
        h->prof_launches += 1;
        h->prof_recs.push_back({4, 128, 256, 1, g.nrows, h16 ? 31 : 30, 1, fl});      // (31: attn_out_h16_kernel, counted with the fp16 builds)
    }
    return 0;
}

// ---------------------------------------------------------------------------
// workspace: bump allocator over one arena, zeroed when the shape changes
// ---------------------------------------------------------------------------
struct Bump {
    char* base; size_t off = 0; size_t cap;
    float* take(size_t nfloats) {
        size_t bytes = (nfloats * sizeof(float) + 255) & ~(size_t)255;
        float* p = (float*)(base ? base + off : nullptr);
        off += bytes;
        return p;
    }
};

struct EstBufs {
    Geom g0, g1, gt;  // level 0 (T), level 1 (T/2), time grid
    float *rm0, *rm1, *X0, *state, *A0, *B0, *R0, *H0, *LN0, *QKV0, *ATT0, *FF0, *CAT1, *U1, *F0, *G0, *V0;
    float *A1, *B1, *R1, *H1, *LN1, *QKV1, *ATT1, *FF1, *CAT0, *D1, *D2, *M1, *UU;
    float *tv, *temb_in, *temb_a, *temb_b, *tproj;
    float *GNP;          // per-tile GroupNorm statistics of a conv_sk32_kernel launch: EV_GN_MAXTILES x 8 groups x {count, mean, M2, -}
    float *ATTP;         // split-key attention partials: EV_ATTN_MAXPARTS x min(rows, EV_ATTN_MAXROWS) x (128 + 4) floats (the levels run one after the other)
    float *C1RMS;        // time-invariant (mu, spk) share of rn[0]'s [block1 conv | res_conv], 512 wide
    float *AR0, *AR1;    // [block1 conv | res_conv] outputs per level, 512 wide
};

void plan_est(Bump& b, int B, int Tp, int in_ch, int nsteps, EstBufs& e) {
    e.g0 = {B * (Tp + 4), Tp + 4, 2, Tp};
    e.g1 = {B * (Tp / 2 + 2), Tp / 2 + 2, 1, Tp / 2};
    e.gt = {nsteps, nsteps, 0, nsteps};
    const size_t n0 = e.g0.nrows, n1 = e.g1.nrows;
    e.rm0 = b.take(n0); e.rm1 = b.take(n1);
    e.X0 = b.take(n0 * in_ch); e.state = b.take(n0 * 80);
    e.A0 = b.take(n0 * 256); e.B0 = b.take(n0 * 256); e.R0 = b.take(n0 * 256); e.H0 = b.take(n0 * 256); e.LN0 = b.take(n0 * 256);
    e.QKV0 = b.take(n0 * 384); e.ATT0 = b.take(n0 * 128); e.FF0 = b.take(n0 * 1024); e.CAT1 = b.take(n0 * 512);
    e.U1 = b.take(n0 * 256); e.F0 = b.take(n0 * 256); e.G0 = b.take(n0 * 256); e.V0 = b.take(n0 * 80);
    e.A1 = b.take(n1 * 256); e.B1 = b.take(n1 * 256); e.R1 = b.take(n1 * 256); e.H1 = b.take(n1 * 256); e.LN1 = b.take(n1 * 256);
    e.QKV1 = b.take(n1 * 384); e.ATT1 = b.take(n1 * 128); e.FF1 = b.take(n1 * 1024); e.CAT0 = b.take(n1 * 512);
    e.D1 = b.take(n1 * 256); e.D2 = b.take(n1 * 256); e.M1 = b.take(n1 * 256); e.UU = b.take(n1 * 256);
    const size_t ns = nsteps;
    e.C1RMS = b.take(n0 * 512); e.AR0 = b.take(n0 * 512); e.AR1 = b.take(n1 * 512);
    e.ATTP = b.take(std::min<size_t>(n0, EV_ATTN_MAXROWS) * EV_ATTN_MAXPARTS * (128 + 4));
    e.GNP = b.take((size_t)EV_GN_MAXTILES * 8 * 4);
    e.tv = b.take(ns); e.temb_in = b.take(ns * in_ch); e.temb_a = b.take(ns * 1024); e.temb_b = b.take(ns * 1024); e.tproj = b.take(ns * 1536);
}

struct VocBufs {
    Geom g[5];
    float *M0, *C0;
    float *U[5], *XS[5], *T1[5], *Pa[5], *Pb[5];
    float *T1x[2][5], *Pax[2][5], *Pbx[2][5];   // scratch of the second and third ResBlock1 chain (three-stream MRF; null otherwise)
    struct { float* p; int C, l; unsigned* amax[2]; int amax_n; } zl[64]; int nz = 0;   // every tensor of the plan (for zero_pads_kernel) and its two sets of amax slots
    unsigned* slots = nullptr; size_t slot_words = 0;    // one block behind the tensors: per tensor 2 x ((rows >> 7) + 2) words, zeroed by every ev_hifigan call
    // slots of a tensor of this plan (null: not one of them)
    int index_of(const float* q) const { for (int k = 0; k < nz; ++k) if (zl[k].p == q) return k; return -1; }
};

void plan_voc(Bump& b, int B, int T, const int* ch, VocBufs& v, bool mrf_streams = false) {
    const int rates[4] = {8, 8, 2, 2};
    int Tl = T, Pl = 4;
    v.g[0] = {B * (Tl + 2 * Pl), Tl + 2 * Pl, Pl, Tl};
    for (int l = 1; l <= 4; ++l) {
        Tl *= rates[l - 1]; Pl *= rates[l - 1];
        v.g[l] = {B * (Tl + 2 * Pl), Tl + 2 * Pl, Pl, Tl};
    }
    v.nz = 0;
    auto take = [&](size_t n, int C, int l) { float* q = b.take(n); v.zl[v.nz].p = q; v.zl[v.nz].C = C; v.zl[v.nz].l = l; ++v.nz; return q; };
    v.M0 = take((size_t)v.g[0].nrows * 80, 80, 0);
    v.C0 = take((size_t)v.g[0].nrows * ch[0], ch[0], 0);
    for (int l = 1; l <= 4; ++l) {
        const size_t n = (size_t)v.g[l].nrows * ch[l];
        v.U[l] = take(n, ch[l], l); v.XS[l] = take(n, ch[l], l); v.T1[l] = take(n, ch[l], l); v.Pa[l] = take(n, ch[l], l); v.Pb[l] = take(n, ch[l], l);
        for (int c = 0; c < 2; ++c) {
            v.T1x[c][l] = mrf_streams ? take(n, ch[l], l) : nullptr; v.Pax[c][l] = mrf_streams ? take(n, ch[l], l) : nullptr;
            v.Pbx[c][l] = mrf_streams ? take(n, ch[l], l) : nullptr;
        }
    }
    v.slot_words = 0;
    for (int k = 0; k < v.nz; ++k) { v.zl[k].amax_n = (v.g[v.zl[k].l].nrows >> 7) + 2; v.slot_words += 2 * (size_t)v.zl[k].amax_n; }
    v.slots = (unsigned*)b.take(v.slot_words);
    size_t off = 0;
    for (int k = 0; k < v.nz; ++k)
        for (int c = 0; c < 2; ++c) { v.zl[k].amax[c] = v.slots ? v.slots + off : nullptr; off += (size_t)v.zl[k].amax_n; }
}
inline bool mrf_streams_for(const ev_handle* h, int B, int Tv) { return h->mrf_max_frames > 0 && (long)B * Tv <= h->mrf_max_frames; }


size_t plan_all(ev_handle* h, char* base, int B, int Tp, int Tv, EstBufs* eb, VocBufs* vb, size_t* voc_off = nullptr) {
    Bump b{base, 0, 0};
    EstBufs e; VocBufs v;
    if (Tp > 0) plan_est(b, B, Tp, h->est.loaded ? h->est.in_ch : 2 * h->dims.n_feats + h->dims.spk_emb_dim, h->max_steps, e);
    if (voc_off) *voc_off = b.off;
    if (Tv > 0) {
        int ch[5] = {512, 256, 128, 64, 32};
        plan_voc(b, B, Tv, h->voc.loaded ? h->voc.ch : ch, v, mrf_streams_for(h, B, Tv));
    }
    if (eb) *eb = e;
    if (vb) *vb = v;
    return b.off;
}

// Replace the arena by one of at least `need` bytes.  Growth on the request path is geometric (x1.5): a stream of utterances with ever
// new maximum lengths re-allocates O(log) times; ev_reserve sizes it once up front (exact).
int grow_ws(ev_handle* h, size_t need, bool geometric) {
    if (h->ws_stream_valid) HIPCHK(h, hipStreamSynchronize(h->ws_stream));
    if (h->stream && (!h->ws_stream_valid || h->stream != h->ws_stream)) HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->ws) HIPCHK(h, hipFree(h->ws));
    h->ws = nullptr; h->ws_bytes = 0; h->ws_B = -1;
    const size_t want = geometric ? std::max(need, h->ws_bytes_last + h->ws_bytes_last / 2) : need;
    if (want > need && hipMalloc((void**)&h->ws, want) == hipSuccess) h->ws_bytes = want;
    else { (void)hipGetLastError(); HIPCHK(h, hipMalloc((void**)&h->ws, need)); h->ws_bytes = need; }
    h->ws_bytes_last = h->ws_bytes;
    ++h->n_allocs;
    return 0;
}

// make sure the arena fits (B, Tp, Tv); zero it when the geometry changes (pad rows must be zero)
int ensure_ws(ev_handle* h, int B, int Tp, int Tv, EstBufs* eb, VocBufs* vb) {
    // keep the other path's last shape so alternating cfm/hifigan calls do not thrash
    if (Tp <= 0) Tp = h->ws_Tp > 0 && h->ws_B == B ? h->ws_Tp : 0;
    if (Tv <= 0) Tv = h->ws_Tv > 0 && h->ws_B == B ? h->ws_Tv : 0;
    // the workspace is shared by every call on this handle: a caller that moves to another stream must not overtake the
    // previous call's kernels (one handle = one stream user at a time, include/emojivoice.h)
    if (h->ws_stream_valid && h->ws_stream != h->stream) HIPCHK(h, hipStreamSynchronize(h->ws_stream));
    h->ws_stream = h->stream; h->ws_stream_valid = true;
    const size_t need = plan_all(h, nullptr, B, Tp, Tv, nullptr, nullptr);
    bool rezero = (B != h->ws_B || Tp != h->ws_Tp || Tv != h->ws_Tv);
    if (need > h->ws_bytes) {
        if (h->captured) return fail(h, "this handle holds a captured ev_cfm_decode: its workspace cannot be replaced (needs %zu bytes, has %zu); "
                                        "use ev_reserve before capturing, or another handle", need, h->ws_bytes);
        if (grow_ws(h, need, true)) return 1;
        rezero = true;
    }
    size_t voc_off = 0;
    VocBufs vplan;
    plan_all(h, h->ws, B, Tp, Tv, eb, &vplan, &voc_off);
    if (vb) *vb = vplan;
    if (rezero) {
        // the estimator's part is zeroed whole (small); of the vocoder's tensors — 0.3-1.1 GB for one streaming utterance — only
        // the pad rows, which is all the convolutions need (zero_pads_kernel)
        static const bool full_zero = getenv("EV_FULL_REZERO") != nullptr;
        const size_t zbytes = (full_zero || Tv <= 0) ? need : voc_off;
        if (zbytes) {
            hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
            const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive;
            if (capturing || h->captured) {      // (kernel nodes only in a captured call — and the same path for the eager calls of a handle that holds graphs)
                const size_t n16 = (zbytes + 15) / 16;
                hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 2048)), dim3(256), 0, h->stream, (f32x4*)h->ws, n16);
                HIPCHK(h, hipGetLastError());
            } else HIPCHK(h, hipMemsetAsync(h->ws, 0, zbytes, h->stream));
        }
        if (!full_zero && Tv > 0 && vplan.nz > 0) {
            ZeroPadParams zp;
            memset(&zp, 0, sizeof zp);
            zp.B = B; zp.n = vplan.nz;
            for (int k = 0; k < vplan.nz; ++k) { zp.p[k] = vplan.zl[k].p; zp.C[k] = vplan.zl[k].C; zp.lvl[k] = vplan.zl[k].l; }
            long most = 0;
            for (int l = 0; l < 5; ++l) {
                zp.S[l] = vplan.g[l].S; zp.P[l] = vplan.g[l].P; zp.T[l] = vplan.g[l].T;
            }
            for (int k = 0; k < vplan.nz; ++k) most = std::max(most, (long)B * (zp.S[zp.lvl[k]] - zp.T[zp.lvl[k]]) * (zp.C[k] / 4));
            const unsigned gx = (unsigned)std::min<long>((most + 255) / 256, 512);
            hipLaunchKernelGGL(zero_pads_kernel, dim3(gx ? gx : 1, vplan.nz), dim3(256), 0, h->stream, zp);
            HIPCHK(h, hipGetLastError());
        }
        h->ws_B = B; h->ws_Tp = Tp; h->ws_Tv = Tv;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// estimator forward (decoder.py:363-443).  On entry X0 holds [x*m | mu*m | spk*m].
// ---------------------------------------------------------------------------
struct LevelBufs { float *A, *Bf, *R, *H, *LN, *QKV, *ATT, *FF; const float* rm; Geom g; float* AR; float* ATTP; float* GNP; };
inline int g_rows32(const Geom& g) { return (g.nrows + 31) / 32; }

int run_resnet(ev_handle* h, const ResnetW& w, const float* X, int ldx, const LevelBufs& L, const float* temb) {
    Epi e; e.gn_part = L.GNP;
    if (launch_conv(h, w.c1r, X, ldx, L.AR, 512, L.g, e)) return 1;                       // [block1 conv | res_conv]
    if (launch_gn(h, L.AR, 512, L.Bf, 256, w.g1, w.b1, L.rm, temb, nullptr, 0, L.g, 256, 1, L.GNP)) return 1;
    if (launch_conv(h, w.c2, L.Bf, 256, L.A, 256, L.g, e)) return 1;
    return launch_gn(h, L.A, 256, L.H, 256, w.g2, w.b2, L.rm, nullptr, L.AR + 256, 512, L.g, 256, 2, L.GNP);
}

// rn[0] with the time-invariant input share hoisted out of the Euler loop (see EstimatorW): X = X0 (x in columns [0, n_feats))
int run_resnet0(ev_handle* h, const EstimatorW& W, const float* X, int ldx, const LevelBufs& L, const float* temb, const float* C1RMS) {
    const ResnetW& w = W.rn[0];
    Epi e;
    { Epi e1; e1.R = C1RMS; e1.ldr = 512; if (launch_conv(h, W.rn0_c1r_x, X, ldx, L.AR, 512, L.g, e1)) return 1; }
    if (launch_gn(h, L.AR, 512, L.Bf, 256, w.g1, w.b1, L.rm, temb, nullptr, 0, L.g, 256, 1)) return 1;   // (c1r adds the hoisted share: no statistics)
    e.gn_part = L.GNP;
    if (launch_conv(h, w.c2, L.Bf, 256, L.A, 256, L.g, e)) return 1;
    return launch_gn(h, L.A, 256, L.H, 256, w.g2, w.b2, L.rm, nullptr, L.AR + 256, 512, L.g, 256, 2, L.GNP);
}

// BasicTransformerBlock (transformer.py:243-316) on L.H; result (masked) -> Z with row stride ldz
int run_transformer(ev_handle* h, const TransW& w, const LevelBufs& L, float* Z, int ldz, int heads) {
    Epi e;
    // row tiles of 32 frames: the fused LayerNorm + linear kernels need about a round of workgroups to pay off (a batch-1
    // decode has 9-27 such tiles: it keeps the split-K small-launch build of the separate linears)
    const bool fuse = h->fuse_mlp && (g_rows32(L.g) >= h->fuse_mlp_min_tiles || (h->sk_spread && h->sk_balance));
    const bool aok = attn_out_ok(h, w.out, L.g, heads);
    int packed = 0;                                     // q / k / v left as fp16 piece pairs (ln_qkv_h16_kernel -> attn_out_h16_kernel)
    if (fuse) {
        if (launch_mlp(h, 1, L.H, w.ln1g, w.ln1b, w.qkv, nullptr, nullptr, nullptr, nullptr, nullptr, L.QKV, 384, L.g,
                       (aok && h->attn_h16 && w.out.Wh) ? w.qkv_scale : nullptr, &packed)) return 1;
    } else {
        if (launch_ln(h, L.H, 256, L.LN, 256, w.ln1g, w.ln1b, L.g)) return 1;
        if (launch_conv(h, w.qkv, L.LN, 256, L.QKV, 384, L.g, e)) return 1;
    }
    if (aok) {
        if (launch_attn_out(h, L.QKV, 384, w.out, L.H, 256, L.rm, L.g, heads, packed ? w.qkv_scale : nullptr)) return 1;   // H <- H + Wout . attn + b, one launch
    } else {
        if (launch_attn(h, L.QKV, 384, L.ATT, 128, L.rm, L.g, heads, L.ATTP)) return 1;
        Epi eo; eo.R = L.H; eo.ldr = 256;
        if (launch_conv(h, w.out, L.ATT, 128, L.H, 256, L.g, eo)) return 1;  // H <- attn + H (in place, element-wise aliasing only)
    }
    if (fuse) return launch_mlp(h, 0, L.H, w.ln3g, w.ln3b, w.ff1, &w.ff2, w.alpha, w.binv, L.H, L.rm, Z, ldz, L.g);
    if (launch_ln(h, L.H, 256, L.LN, 256, w.ln3g, w.ln3b, L.g)) return 1;
    Epi e1; e1.act = ACT_SNAKE; e1.act_a = w.alpha; e1.act_b = w.binv;
    if (launch_conv(h, w.ff1, L.LN, 256, L.FF, 1024, L.g, e1)) return 1;
    Epi e2; e2.R = L.H; e2.ldr = 256; e2.mask2 = 1; e2.rowmask = L.rm;
    return launch_conv(h, w.ff2, L.FF, 1024, Z, ldz, L.g, e2);
}

// One estimator evaluation.  euler: state += dt * v, X0[:, :80] = state * m ; else V0 = v.
int run_estimator(ev_handle* h, EstBufs& b, int step, float dt, bool euler, const float* tproj = nullptr) {
    const EstimatorW& w = h->est;
    const int heads = h->dims.heads;
    LevelBufs L0{b.A0, b.B0, b.R0, b.H0, b.LN0, b.QKV0, b.ATT0, b.FF0, b.rm0, b.g0, b.AR0, b.ATTP, b.GNP};
    LevelBufs L1{b.A1, b.B1, b.R1, b.H1, b.LN1, b.QKV1, b.ATT1, b.FF1, b.rm1, b.g1, b.AR1, b.ATTP, b.GNP};
    const float* tp = (tproj ? tproj : b.tproj) + (size_t)step * 1536;
    // down 0 @T
    if (run_resnet0(h, w, b.X0, w.in_ch, L0, tp + 0 * 256, b.C1RMS)) return 1;
    if (run_transformer(h, w.tr[0], L0, b.CAT1 + 256, 512, heads)) return 1;      // hidden 0
    {   // Downsample1D k3 s2 p1 over the pair view of CAT1[:, 256:512]
        Epi e; e.isplit_log2 = 8; e.isstride = 512; e.mask1 = 1; e.rowmask = b.rm1;
        if (launch_conv(h, w.down0, b.CAT1 + 256, 1024, b.D1, 256, b.g1, e)) return 1;
    }
    // down 1 @T/2
    if (run_resnet(h, w.rn[1], b.D1, 256, L1, tp + 1 * 256)) return 1;
    if (run_transformer(h, w.tr[1], L1, b.CAT0 + 256, 512, heads)) return 1;      // hidden 1
    { Epi e; e.mask1 = 1; e.rowmask = b.rm1; if (launch_conv(h, w.down1, b.CAT0 + 256, 512, b.D2, 256, b.g1, e)) return 1; }
    // mid
    if (run_resnet(h, w.rn[2], b.D2, 256, L1, tp + 2 * 256)) return 1;
    if (run_transformer(h, w.tr[2], L1, b.M1, 256, heads)) return 1;
    if (run_resnet(h, w.rn[3], b.M1, 256, L1, tp + 3 * 256)) return 1;
    if (run_transformer(h, w.tr[3], L1, b.CAT0, 512, heads)) return 1;
    // up 0 @T/2: cat([x, hidden1])
    if (run_resnet(h, w.rn[4], b.CAT0, 512, L1, tp + 4 * 256)) return 1;
    if (run_transformer(h, w.tr[4], L1, b.UU, 256, heads)) return 1;
    {   // ConvTranspose1d k4 s2 p1: view-row q -> frames 2q, 2q+1 of CAT1[:, 0:256]
        Epi e; e.osplit_log2 = 8; e.osstride = 512; e.mask1 = 1; e.rowmask = b.rm0; e.mmul = 2;
        if (launch_conv(h, w.up0, b.UU, 256, b.CAT1, 1024, b.g1, e)) return 1;
    }
    // up 1 @T: cat([x, hidden0])
    if (run_resnet(h, w.rn[5], b.CAT1, 512, L0, tp + 5 * 256)) return 1;
    if (run_transformer(h, w.tr[5], L0, b.U1, 256, heads)) return 1;
    { Epi e; e.mask1 = 1; e.rowmask = b.rm0; if (launch_conv(h, w.up1, b.U1, 256, b.F0, 256, b.g0, e)) return 1; }
    // final block + projection
    { Epi e; if (launch_conv(h, w.fin_conv, b.F0, 256, b.A0, 256, b.g0, e)) return 1; }
    if (launch_gn(h, b.A0, 256, b.G0, 256, w.fin_g, w.fin_b, b.rm0, nullptr, nullptr, 0, b.g0, 256, 0)) return 1;
    if (euler) {
        Epi e; e.mask1 = 1; e.rowmask = b.rm0; e.scale = dt; e.R = b.state; e.ldr = 80; e.Y2 = b.X0; e.ldy2 = w.in_ch;
        return launch_conv(h, w.fin_proj, b.G0, 256, b.state, 80, b.g0, e);
    }
    Epi e; e.mask1 = 1; e.rowmask = b.rm0;
    return launch_conv(h, w.fin_proj, b.G0, 256, b.V0, 80, b.g0, e);
}

// sinusoidal embedding + time MLP + the six resnet time projections for a list of times
int run_time_mlp(ev_handle* h, EstBufs& b, const std::vector<float>& ts) {
    const EstimatorW& w = h->est;
    const int nt = (int)ts.size(), dim = w.in_ch, half = dim / 2;
    std::vector<float> emb((size_t)nt * dim);
    const float ne = -(float)(log(10000.0) / (double)(half - 1));   // decoder.py:24-25, evaluated in float like torch
    for (int i = 0; i < nt; ++i) {
        const float st = 1000.0f * ts[i];
        for (int k = 0; k < half; ++k) {
            const float f = expf((float)k * ne);
            const float a = st * f;
            emb[(size_t)i * dim + k] = sinf(a);
            emb[(size_t)i * dim + half + k] = cosf(a);
        }
    }
    {   // staged through a two-slot pinned ring owned by the handle: no host/stream synchronisation per call, so a caller
        // that pipelines consecutive batches on two streams (emojivoice_amd/pipeline.py) keeps enqueueing ahead of the GPU.
        // A call under stream capture gets a region of the capture pool instead, which is never written again: a replay may
        // read it at any later time, whatever eager calls (other n_steps, other ring contents) happen in between.
        const size_t bytes = emb.size() * sizeof(float);
        hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
        const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive;
        const float* src = nullptr;
        if (capturing) {
            if (!h->cap_pool || bytes > h->cap_stride) return fail(h, "ev_cfm_decode under stream capture: %d Euler steps exceed the capture staging (%zu bytes per call)", nt, h->cap_stride);
            // the time grid depends on the number of Euler steps only (flow_matching.py:52): captured calls with the same step count share
            // ONE region (written once, never again), so the pool bounds the distinct step counts a handle captures, not the number of graphs
            int slot = -1;
            for (int k = 0; k < h->cap_used; ++k) if (h->cap_nt[k] == nt) slot = k;
            if (slot < 0) {
                if (h->cap_used >= EV_CAPTURE_SLOTS) return fail(h, "ev_cfm_decode under stream capture: this handle has already captured calls with %d different step counts (their staging is never recycled)", EV_CAPTURE_SLOTS);
                slot = h->cap_used++;
                h->cap_nt[slot] = nt;
                memcpy((char*)h->cap_pool + (size_t)slot * h->cap_stride, emb.data(), bytes);
            }
            src = (const float*)((const char*)h->cap_pool + (size_t)slot * h->cap_stride);
        } else {
            const int slot = (h->temb_slot ^= 1);
            if (h->temb_cap[slot] < bytes) {
                if (h->temb_host[slot]) { HIPCHK(h, hipEventSynchronize(h->temb_ev[slot])); HIPCHK(h, hipHostFree(h->temb_host[slot])); h->temb_host[slot] = nullptr; h->temb_cap[slot] = 0; }
                const size_t want = std::max(bytes, (size_t)64 * dim * sizeof(float));
                HIPCHK(h, hipHostMalloc((void**)&h->temb_host[slot], want, hipHostMallocDefault));
                h->temb_cap[slot] = want;
                ++h->n_allocs;
                if (!h->temb_ev[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->temb_ev[slot], hipEventDisableTiming));
            } else {
                HIPCHK(h, hipEventSynchronize(h->temb_ev[slot]));   // the copy issued two calls ago has read this slot
            }
            memcpy(h->temb_host[slot], emb.data(), bytes);
            src = h->temb_host[slot];
        }
        HIPCHK(h, hipMemcpyAsync(b.temb_in, src, bytes, hipMemcpyHostToDevice, h->stream));
        if (!capturing) HIPCHK(h, hipEventRecord(h->temb_ev[h->temb_slot], h->stream));
    }
    Geom gt{nt, nt, 0, nt};
    Epi e1; e1.act = ACT_SILU;
    if (launch_conv(h, w.t1, b.temb_in, dim, b.temb_a, 1024, gt, e1)) return 1;
    Epi e2; e2.act = ACT_MISH;   // every consumer applies Mish first (decoder.py:49)
    if (launch_conv(h, w.t2, b.temb_a, 1024, b.temb_b, 1024, gt, e2)) return 1;
    Epi e3;
    return launch_conv(h, w.tmlp, b.temb_b, 1024, b.tproj, 1536, gt, e3);
}

int prep_inputs(ev_handle* h, EstBufs& b, const float* d_x, const float* d_mu, const int32_t* d_len, const float* d_spk, int B, int Tp) {
    const EstimatorW& w = h->est;
    hipStream_t st = h->stream;
    hipLaunchKernelGGL(rowmask_kernel, dim3((b.g0.nrows + 255) / 256), dim3(256), 0, st, b.rm0, d_len, b.g0.nrows, b.g0.S, b.g0.P, b.g0.T, 1);
    hipLaunchKernelGGL(rowmask_kernel, dim3((b.g1.nrows + 255) / 256), dim3(256), 0, st, b.rm1, d_len, b.g1.nrows, b.g1.S, b.g1.P, b.g1.T, 2);
    dim3 grid((Tp + 31) / 32, (80 + 31) / 32, B);
    // ODE state (unmasked) and the masked estimator input [x*m | mu*m | spk*m]
    hipLaunchKernelGGL(cm_to_fm_kernel, grid, dim3(256), 0, st, d_x, b.state, 80, 0, 80, Tp, b.g0.S, b.g0.P, (const float*)nullptr, 1.0f);
    hipLaunchKernelGGL(cm_to_fm_kernel, grid, dim3(256), 0, st, d_x, b.X0, w.in_ch, 0, 80, Tp, b.g0.S, b.g0.P, (const float*)b.rm0, 1.0f);
    hipLaunchKernelGGL(cm_to_fm_kernel, grid, dim3(256), 0, st, d_mu, b.X0, w.in_ch, 80, 80, Tp, b.g0.S, b.g0.P, (const float*)b.rm0, 1.0f);
    if (w.in_ch > 160) {
        const int C = w.in_ch - 160;
        const long tot = (long)b.g0.nrows * C;
        hipLaunchKernelGGL(bcast_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_spk, b.X0, w.in_ch, 160, C, b.g0.nrows, b.g0.S, b.g0.P, b.g0.T, (const float*)b.rm0);
    }
    HIPCHK(h, hipGetLastError());
    {   // time-invariant share of rn[0]: columns [n_feats, in_ch) of X0 = [mu*m | spk*m]
        Epi e;
        if (launch_conv(h, w.rn0_c1r_ms, b.X0 + h->dims.n_feats, w.in_ch, b.C1RMS, 512, b.g0, e)) return 1;
    }
    return 0;
}

int launch_cln(ev_handle* h, const float* X, int ldx, const float* R, int ldr, const float* g, const float* b, const float* rowmask, float* Y, int ldy,
               const Geom& geo, int C, int relu) {
    CLNParams p;
    p.X = X; p.ldx = ldx; p.R = R; p.ldr = ldr; p.gamma = g; p.beta = b; p.rowmask = rowmask; p.Y = Y; p.ldy = ldy;
    p.nrows = geo.nrows; p.S = geo.S; p.P = geo.P; p.T = geo.T; p.relu = relu; p.eps = 1e-4f;
    if (C == 192) hipLaunchKernelGGL(chan_layernorm_kernel<3>, dim3((geo.nrows + 3) / 4), dim3(256), 0, h->stream, p);
    else if (C == 256) hipLaunchKernelGGL(chan_layernorm_kernel<4>, dim3((geo.nrows + 3) / 4), dim3(256), 0, h->stream, p);
    else return fail(h, "channel LayerNorm width %d not supported (192 / 256)", C);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// TextEncoder.forward (text_encoder.py:378-410) + DurationPredictor (:70-94).  Every activation is kept multiplied by the
// frame mask: convolutions and attention only ever see masked inputs in the reference too, and the per-frame ops
// (LayerNorm, 1x1 convs, residual adds) cannot carry a padded frame's value into a valid one, so valid frames are unchanged.
int run_text_encoder(ev_handle* h, const int64_t* d_ids, const int32_t* d_len, const float* d_spk, int B, int Tx, float* d_mu, float* d_logw) {
    const TextEncW& w = h->enc;
    const int P = 2, S = Tx + 2 * P;
    const Geom g{B * S, S, P, Tx};
    const size_t n = (size_t)g.nrows;
    const int C = w.C, nc = w.nch;
    // workspace (floats): rm | E | Xm | A | H | H1 | Y | D | D2 | QKV | ATT | FF | MU | LW
    const size_t need = (n * (1 + 3 * nc + 3 * C + 2 * w.dp1.Cout + 3 * C + C + w.ffc + 80 + 4) + 1024) * sizeof(float);
    if (scratch_acquire(h, h->enc_ws, need)) return 1;
    HIPCHK(h, hipMemsetAsync(h->enc_ws.p, 0, need, h->stream));     // pad rows are the convolutions' zero padding
    Bump bp; bp.base = h->enc_ws.p; bp.off = 0;
    float* rm = bp.take(n); float* E = bp.take(n * nc); float* Xm = bp.take(n * nc); float* A = bp.take(n * nc);
    float* H = bp.take(n * C); float* H1 = bp.take(n * C); float* Y = bp.take(n * C);
    float* D = bp.take(n * w.dp1.Cout); float* D2 = bp.take(n * w.dp1.Cout);
    float* QKV = bp.take(n * 3 * C); float* ATT = bp.take(n * C); float* FF = bp.take(n * w.ffc);
    float* MU = bp.take(n * 80); float* LW = bp.take(n * 4);
    hipStream_t st = h->stream;
    hipLaunchKernelGGL(rowmask_kernel, dim3((g.nrows + 255) / 256), dim3(256), 0, st, rm, d_len, g.nrows, g.S, g.P, g.T, 1);
    {
        const long tot = (long)B * Tx * (nc / 4);
        hipLaunchKernelGGL(enc_embed_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_ids, d_len, (const float*)w.emb, w.nvocab, sqrtf((float)nc),
                           E, Xm, nc, B, Tx, S, P, h->bad_ids_dev);
    }
    HIPCHK(h, hipGetLastError());
    // prenet ConvReluNorm (:36-67): 3 x [conv k5 -> LayerNorm -> ReLU], then x_org + proj(x), masked
    for (int i = 0; i < 3; ++i) {
        { Epi e; if (launch_conv(h, w.pre[i], Xm, nc, A, nc, g, e)) return 1; }
        if (launch_cln(h, A, nc, nullptr, 0, w.pre_g[i], w.pre_b[i], rm, Xm, nc, g, nc, 1)) return 1;
    }
    { Epi e; e.R = E; e.ldr = nc; e.mask2 = 1; e.rowmask = rm; if (launch_conv(h, w.pre_proj, Xm, nc, H, C, g, e)) return 1; }
    if (C > nc) {
        const long tot = (long)g.nrows * (C - nc);
        hipLaunchKernelGGL(bcast_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_spk, H, C, nc, C - nc, g.nrows, g.S, g.P, g.T, (const float*)rm);
        HIPCHK(h, hipGetLastError());
    }
    // Encoder (:276-325)
    const int kc = C / w.heads;
    for (int l = 0; l < w.nlayers; ++l) {
        const EncLayerW& L = w.layers[l];
        { Epi e; if (launch_conv(h, L.qkv, H, C, QKV, 3 * C, g, e)) return 1; }
        {
            const long tot = (long)g.nrows * 2 * w.heads * (kc / 4);
            hipLaunchKernelGGL(enc_rope_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, QKV, 3 * C, (const float*)w.theta, w.heads, kc, kc / 2,
                               g.nrows, g.S, g.P, g.T);
            const size_t att_smem = (size_t)(64 * (kc | 4) + 4 * kc + 4 * ((Tx + 63) & ~63)) * sizeof(float);
            if (kc % 4 != 0 || kc > 128) return fail(h, "ev_text_encoder: channels per head must be a multiple of 4, at most 128");
            if (att_smem > 160 * 1024) return fail(h, "ev_text_encoder: text too long for the attention workspace in LDS");
            ensure_dyn_smem<enc_attention_kernel>(att_smem, h->device);
            hipLaunchKernelGGL(enc_attention_kernel, dim3((Tx + 3) / 4, w.heads, B), dim3(256), att_smem, st, (const float*)QKV, 3 * C, d_len, ATT, C, w.heads, kc, B, Tx, S, P,
                               sqrtf((float)kc));
            HIPCHK(h, hipGetLastError());
        }
        { Epi e; e.R = H; e.ldr = C; if (launch_conv(h, L.out, ATT, C, Y, C, g, e)) return 1; }                  // x + attn(x)
        if (launch_cln(h, Y, C, nullptr, 0, L.g1, L.b1, rm, H1, C, g, C, 0)) return 1;
        { Epi e; e.act = ACT_LRELU; e.act_slope = 0.f; e.mask1 = 1; e.rowmask = rm; if (launch_conv(h, L.ff1, H1, C, FF, w.ffc, g, e)) return 1; }   // relu
        { Epi e; e.mask1 = 1; e.rowmask = rm; e.R = H1; e.ldr = C; if (launch_conv(h, L.ff2, FF, w.ffc, Y, C, g, e)) return 1; }                    // x + ffn(x)*m
        if (launch_cln(h, Y, C, nullptr, 0, L.g2, L.b2, rm, H, C, g, C, 0)) return 1;
    }
    { Epi e; e.mask1 = 1; e.rowmask = rm; if (launch_conv(h, w.proj_m, H, C, MU, 80, g, e)) return 1; }
    {
        dim3 grid((Tx + 31) / 32, (80 + 31) / 32, B);
        hipLaunchKernelGGL(fm_to_cm_kernel, grid, dim3(256), 0, st, (const float*)MU, 80, 0, d_mu, 80, Tx, S, P, 1.0f, 0.0f);
    }
    // DurationPredictor (:70-94): conv k3 -> ReLU -> LayerNorm (x2) at filter_channels_dp, 1x1 projection, masked
    const int fd = w.dp1.Cout;
    { Epi e; e.act = ACT_LRELU; e.act_slope = 0.f; if (launch_conv(h, w.dp1, H, C, D, fd, g, e)) return 1; }
    if (launch_cln(h, D, fd, nullptr, 0, w.dp_g1, w.dp_b1, rm, D2, fd, g, fd, 0)) return 1;
    { Epi e; e.act = ACT_LRELU; e.act_slope = 0.f; if (launch_conv(h, w.dp2, D2, fd, D, fd, g, e)) return 1; }
    if (launch_cln(h, D, fd, nullptr, 0, w.dp_g2, w.dp_b2, rm, D2, fd, g, fd, 0)) return 1;
    { Epi e; e.mask1 = 1; e.rowmask = rm; if (launch_conv(h, w.dp_proj, D2, fd, LW, 1, g, e)) return 1; }
    {
        const size_t total = (size_t)B * Tx;
        hipLaunchKernelGGL(strip_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)LW, d_logw, Tx, S, P, total);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Build the two DFT-basis layers once (double-precision trigonometry, rounded to fp32).
int denoiser_init(ev_handle* h) {
    DenoiserW& w = h->dn;
    if (w.ready) return 0;
    const int N = 1024, NB = 513, HC = 516;                   // bins padded to 516 per half: [re | im] = 1032 columns
    const double PI = 3.14159265358979323846;
    std::vector<double> win(N);
    for (int n = 0; n < N; ++n) win[n] = 0.5 - 0.5 * cos(2.0 * PI * n / N);       // torch.hann_window(1024), periodic
    {   // forward: out[f][k] = sum_n w[n] a[256 f + n] e^{-2 pi i k n / N}; taps j = 0..3 over rows f + j, 256 columns each
        ConvLayer& L = w.fwd;
        L.Cin = 256; L.Cout = 2 * HC; L.ntaps = 4; L.Mpad = round_up(L.Cout, 128); L.Kpad = 256;
        for (int j = 0; j < 4; ++j) L.off[j] = j;
        std::vector<float> Wh((size_t)4 * L.Mpad * L.Kpad, 0.f);
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < NB; ++k)
                for (int c = 0; c < 256; ++c) {
                    const int n = 256 * j + c;
                    const double ang = 2.0 * PI * (double)((long)k * n % N) / N;
                    Wh[((size_t)j * L.Mpad + k) * L.Kpad + c] = (float)(win[n] * cos(ang));
                    Wh[((size_t)j * L.Mpad + HC + k) * L.Kpad + c] = (float)(-win[n] * sin(ang));
                }
        L.macs_per_row = (double)2 * NB * N;
        if (finish_layer(h, L, Wh, nullptr)) return 1;
    }
    {   // inverse + overlap-add: row r, column c <- sum_j frame (r - j), sample n = 256 j + c of w[n] * irfft(X)[n]
        ConvLayer& L = w.inv;
        L.Cin = 2 * HC; L.Cout = 256; L.ntaps = 4; L.Mpad = round_up(L.Cout, 128); L.Kpad = round_up(L.Cin, EV_BK);
        for (int j = 0; j < 4; ++j) L.off[j] = -j;
        std::vector<float> Wh((size_t)4 * L.Mpad * L.Kpad, 0.f);
        for (int j = 0; j < 4; ++j)
            for (int c = 0; c < 256; ++c) {
                const int n = 256 * j + c;
                for (int k = 0; k < NB; ++k) {
                    const double ck = (k == 0 || k == N / 2) ? 1.0 : 2.0;
                    const double ang = 2.0 * PI * (double)((long)k * n % N) / N;
                    Wh[((size_t)j * L.Mpad + c) * L.Kpad + k] = (float)(win[n] * ck * cos(ang) / N);
                    if (k != 0 && k != N / 2) Wh[((size_t)j * L.Mpad + c) * L.Kpad + HC + k] = (float)(-win[n] * ck * sin(ang) / N);
                }
            }
        L.macs_per_row = (double)2 * NB * N;
        if (finish_layer(h, L, Wh, nullptr)) return 1;
    }
    std::vector<float> w2(N);
    for (int n = 0; n < N; ++n) w2[n] = (float)(win[n] * win[n]);
    if (dev_upload(h, w2, &w.win2)) return 1;
    w.ready = true;
    return 0;
}

// STFT -> (optional magnitude output) -> (optional spectral gain + inverse STFT)
int run_denoiser(ev_handle* h, const float* d_audio, int B, int L, const float* d_bias, float strength, float* d_out, float* d_mag) {
    if (denoiser_init(h)) return 1;
    const DenoiserW& w = h->dn;
    const int T = L / 256, R = T + 4, F = T + 1, P = 4, S = R + 2 * P;
    const Geom g{B * S, S, P, R};
    const size_t n = (size_t)g.nrows;
    const size_t need = (n * (256 + 1032 + 256) + 1024) * sizeof(float);
    if (scratch_acquire(h, h->dn_ws, need)) return 1;
    HIPCHK(h, hipMemsetAsync(h->dn_ws.p, 0, need, h->stream));
    Bump bp; bp.base = h->dn_ws.p; bp.off = 0;
    float* SIG = bp.take(n * 256); float* SPEC = bp.take(n * 1032); float* OUT = bp.take(n * 256);
    hipStream_t st = h->stream;
    {
        const size_t tot = (size_t)B * (L + 1024);
        hipLaunchKernelGGL(dn_pad_reflect_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_audio, SIG, B, L, S, P);
    }
    { Epi e; if (launch_conv(h, w.fwd, SIG, 256, SPEC, 1032, g, e)) return 1; }
    {
        const size_t tot = n * 516;
        hipLaunchKernelGGL(dn_gain_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, SPEC, d_bias, strength, d_mag, g.nrows, S, P, F);
    }
    HIPCHK(h, hipGetLastError());
    if (!d_out) return 0;
    { Epi e; if (launch_conv(h, w.inv, SPEC, 1032, OUT, 256, g, e)) return 1; }
    {
        const size_t tot = (size_t)B * L;
        hipLaunchKernelGGL(dn_crop_norm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)OUT, (const float*)w.win2, d_out, B, L, S, P, F);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int ev_abi_version(void) { return EV_ABI_VERSION; }

int ev_create(ev_handle** out, int device, const ev_model_dims* dims) {
    if (!out || !dims) return 1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return 2;
    if (hipSetDevice(device) != hipSuccess) return 3;
    ev_handle* h = new ev_handle();
    h->device = device;
    h->dims = *dims;
    { const char* fp = getenv("EV_FUSE_PAIRS"); if (fp && *fp == '0') h->fuse_pairs = false; }
    { const char* fp = getenv("EV_FUSE128"); if (fp && *fp) h->fuse128 = atoi(fp); }
    { const char* fp = getenv("EV_FUSE_MLP"); if (fp && *fp == '0') h->fuse_mlp = false; }
    { const char* fp = getenv("EV_FUSE_ATTN"); if (fp && *fp == '0') h->fuse_attn = false; }
    if (getenv("EV_NO_ATTN_H16")) h->attn_h16 = false;
    { const char* fp = getenv("EV_FUSE_MLP_MIN"); if (fp && *fp) h->fuse_mlp_min_tiles = atoi(fp); }
    { const char* fp = getenv("EV_MRF_STREAMS_MAX"); if (fp && *fp) h->mrf_max_frames = atoi(fp); }
    if (getenv("EV_NO_AMAX")) h->use_amax = false;
    { const char* sp = getenv("EV_SPLIT"); if (sp && *sp) { const int t = atoi(sp); h->split_terms = (t == 0 || t == 3 || t == 6 || t == 9 || t == 16) ? t : 16; } }
    { const char* fp = getenv("EV_NO_SK_BALANCE"); if (fp && *fp && *fp != '0') h->sk_balance = false; }
    { const char* fp = getenv("EV_SK_SPIN"); if (fp && *fp) h->sk_spin = atoi(fp); }
    if (getenv("EV_NO_SK_STEAL")) h->sk_steal = false;
    { const char* fp = getenv("EV_SK_WGS"); if (fp && *fp) h->sk_wgs = std::min(3, std::max(1, atoi(fp))); }
    { const char* fp = getenv("EV_SK_SPREAD"); if (fp && *fp) h->sk_spread = *fp != '0'; }
    if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) h->ncu = 0;
    // the shipped decoder configuration (configs/model/decoder/default.yaml: 2 heads x 64) is the only one the workspace
    // plan and the transformer launch sequence are laid out for
    if (dims->n_feats != 80 || dims->channels != 256 || dims->head_dim != 64 || dims->heads != 2) {
        delete h;
        return 4;
    }
    *out = h;
    return 0;
}

void ev_destroy(ev_handle* h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    for (void* p : h->owned) hipFree(p);
    if (h->ws) hipFree(h->ws);
    if (h->cap_pool) hipHostFree(h->cap_pool);
    for (auto& c : h->tproj_cache) hipFree(c.d);
    h->tproj_cache.clear();
    if (h->sk_ctrl) hipFree(h->sk_ctrl);
    if (h->sk_part) hipFree(h->sk_part);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->enc_ws.p) hipFree(h->enc_ws.p);
    if (h->dn_ws.p) hipFree(h->dn_ws.p);
    if (h->bad_ids_host) hipHostFree(h->bad_ids_host);
    for (int i = 0; i < 2; ++i) { if (h->temb_ev[i]) hipEventDestroy(h->temb_ev[i]); if (h->temb_host[i]) hipHostFree(h->temb_host[i]); }
    for (int i = 0; i < 2; ++i) if (h->mrf_stream[i]) hipStreamDestroy(h->mrf_stream[i]);
    for (int i = 0; i < 4; ++i) if (h->mrf_ev[i]) hipEventDestroy(h->mrf_ev[i]);
    delete h;
}

const char* ev_last_error(ev_handle* h) { return h ? h->err.c_str() : "null handle"; }

static int build_map(ev_handle* h, const float* blob, const ev_tensor_index* index, size_t n, TensorMap& m) {
    if (!blob || !index) return fail(h, "null blob/index");
    for (size_t i = 0; i < n; ++i) {
        HostTensor t;
        t.p = blob + index[i].offset;
        t.ndim = index[i].ndim;
        if (t.ndim < 0 || t.ndim > 4) return fail(h, "bad ndim for %s", index[i].name);
        for (int d = 0; d < t.ndim; ++d) t.shape[d] = index[i].shape[d];
        m[index[i].name] = t;
    }
    return 0;
}

int ev_load_estimator(ev_handle* h, const float* blob, const ev_tensor_index* index, size_t n) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    TensorMap m;
    if (build_map(h, blob, index, n, m)) return 1;
    if (ensure_sk(h, false)) return 1;
    if (!h->tproj_cache.empty()) {                       // new weights: the cached time-MLP outputs are those of the old ones
        HIPCHK(h, hipDeviceSynchronize());
        for (auto& c : h->tproj_cache) hipFree(c.d);
        h->tproj_cache.clear();
    }
    EstimatorW& w = h->est;
    w.loaded = false;
#define T_(k) find(h, m, k)
#define REQ(x) do { if (x) return 1; } while (0)
    const HostTensor* l1 = T_("time_mlp.linear_1.weight");
    if (!l1) return 1;
    w.in_ch = (int)l1->shape[1];
    if (w.in_ch != 2 * h->dims.n_feats + h->dims.spk_emb_dim) return fail(h, "estimator in_channels %d != 2*n_feats + spk_emb_dim", w.in_ch);
    {
        const HostTensor *b1 = T_("time_mlp.linear_1.bias"), *l2 = T_("time_mlp.linear_2.weight"), *b2 = T_("time_mlp.linear_2.bias");
        if (!b1 || !l2 || !b2) return 1;
        REQ(pack_linear_stack(h, w.t1, {l1}, {b1}));
        REQ(pack_linear_stack(h, w.t2, {l2}, {b2}));
    }
    const char* rn_names[6] = {"down_blocks.0.0", "down_blocks.1.0", "mid_blocks.0.0", "mid_blocks.1.0", "up_blocks.0.0", "up_blocks.1.0"};
    const char* tr_names[6] = {"down_blocks.0.1.0", "down_blocks.1.1.0", "mid_blocks.0.1.0", "mid_blocks.1.1.0", "up_blocks.0.1.0", "up_blocks.1.1.0"};
    std::vector<const HostTensor*> mw, mb;
    for (int i = 0; i < 6; ++i) {
        const std::string p = rn_names[i];
        const HostTensor *c1w = T_(p + ".block1.block.0.weight"), *c1b = T_(p + ".block1.block.0.bias");
        const HostTensor *c2w = T_(p + ".block2.block.0.weight"), *c2b = T_(p + ".block2.block.0.bias");
        const HostTensor *rw = T_(p + ".res_conv.weight"), *rb = T_(p + ".res_conv.bias");
        const HostTensor *tw = T_(p + ".mlp.1.weight"), *tb = T_(p + ".mlp.1.bias");
        if (!c1w || !c1b || !c2w || !c2b || !rw || !rb || !tw || !tb) return 1;
        // [block1 conv (k = 3) | res_conv (1x1 -> centre tap)] stacked along the output axis, input channels [c0, c1)
        // npad > c1 - c0: the layer is declared npad channels wide with ZERO weights on the channels past c1 - the fp16 builds take inputs in
        // whole 64-channel chunks, and what lies behind the slice in the same row (rn[0]'s x share: the first channels of mu) then meets zeros
        auto stack = [&](int c0, int c1, bool with_bias, ConvLayer& L, int npad = 0) -> int {
            const int Co = (int)c1w->shape[0], Ci = (int)c1w->shape[1], K = (int)c1w->shape[2], Cr = (int)rw->shape[0], nr = c1 - c0, n = std::max(nr, npad);
            std::vector<float> wt((size_t)(Co + Cr) * n * K, 0.f), bt((size_t)(Co + Cr), 0.f);
            for (int co = 0; co < Co; ++co)
                for (int ci = 0; ci < nr; ++ci)
                    for (int k = 0; k < K; ++k) wt[((size_t)co * n + ci) * K + k] = c1w->p[((size_t)co * Ci + c0 + ci) * K + k];
            for (int co = 0; co < Cr; ++co)
                for (int ci = 0; ci < nr; ++ci) wt[((size_t)(Co + co) * n + ci) * K + K / 2] = rw->p[(size_t)co * Ci + c0 + ci];
            for (int co = 0; co < Co; ++co) bt[co] = c1b->p[co];
            for (int co = 0; co < Cr; ++co) bt[Co + co] = rb->p[co];
            HostTensor wh, bh;
            wh.p = wt.data(); wh.ndim = 3; wh.shape[0] = Co + Cr; wh.shape[1] = n; wh.shape[2] = K;
            bh.p = bt.data(); bh.ndim = 1; bh.shape[0] = Co + Cr;
            if (pack_conv(h, L, wh, with_bias ? &bh : nullptr, 1)) return 1;
            L.macs_per_row = (double)Co * nr * K + (double)Cr * nr;        // the reference's arithmetic (zero taps and zero channels are not work)
            return 0;
        };
        if ((int)rw->shape[1] != (int)c1w->shape[1] || (int)c1w->shape[2] != 3) return fail(h, "resnet %d: unexpected conv shapes", i);
        REQ(stack(0, (int)c1w->shape[1], true, w.rn[i].c1r));
        REQ(pack_conv(h, w.rn[i].c2, *c2w, c2b, 1));
        if (i == 0) {   // split along Cin at n_feats: [x | mu, spk]
            const int nf = h->dims.n_feats, cin = (int)c1w->shape[1];
            // x share: n_feats = 80 channels, read as two 64-channel chunks of the [x | mu | spk] row (EV_RN0_PAD=0: as 80 channels on the fp32 MFMA)
            static const bool no_pad = getenv("EV_RN0_PAD") && atoi(getenv("EV_RN0_PAD")) == 0;
            const int npad = (!no_pad && nf % 64 != 0 && ((nf + 63) & ~63) <= cin) ? ((nf + 63) & ~63) : 0;
            REQ(stack(0, nf, false, w.rn0_c1r_x, npad));
            REQ(stack(nf, cin, true, w.rn0_c1r_ms));
        }
        REQ(upload_vec(h, m, p + ".block1.block.1.weight", &w.rn[i].g1));
        REQ(upload_vec(h, m, p + ".block1.block.1.bias", &w.rn[i].b1));
        REQ(upload_vec(h, m, p + ".block2.block.1.weight", &w.rn[i].g2));
        REQ(upload_vec(h, m, p + ".block2.block.1.bias", &w.rn[i].b2));
        mw.push_back(tw); mb.push_back(tb);
    }
    REQ(pack_linear_stack(h, w.tmlp, mw, mb));
    for (int i = 0; i < 6; ++i) {
        const std::string p = tr_names[i];
        const HostTensor *q = T_(p + ".attn1.to_q.weight"), *k = T_(p + ".attn1.to_k.weight"), *v = T_(p + ".attn1.to_v.weight");
        const HostTensor *ow = T_(p + ".attn1.to_out.0.weight"), *ob = T_(p + ".attn1.to_out.0.bias");
        const HostTensor *f1w = T_(p + ".ff.net.0.proj.weight"), *f1b = T_(p + ".ff.net.0.proj.bias");
        const HostTensor *f2w = T_(p + ".ff.net.2.weight"), *f2b = T_(p + ".ff.net.2.bias");
        if (!q || !k || !v || !ow || !ob || !f1w || !f1b || !f2w || !f2b) return 1;
        REQ(pack_linear_stack(h, w.tr[i].qkv, {q, k, v}, {}));
        REQ(pack_linear_stack(h, w.tr[i].out, {ow}, {ob}));
        REQ(pack_linear_stack(h, w.tr[i].ff1, {f1w}, {f1b}));
        REQ(pack_linear_stack(h, w.tr[i].ff2, {f2w}, {f2b}));
        REQ(upload_vec(h, m, p + ".norm1.weight", &w.tr[i].ln1g));
        REQ(upload_vec(h, m, p + ".norm1.bias", &w.tr[i].ln1b));
        REQ(upload_vec(h, m, p + ".norm3.weight", &w.tr[i].ln3g));
        REQ(upload_vec(h, m, p + ".norm3.bias", &w.tr[i].ln3b));
        // derived by the host loader with torch ops (transformer.py:71-78): exp(alpha), 1/(exp(beta)+1e-9)
        REQ(upload_vec(h, m, p + ".ff.net.0.alpha_exp", &w.tr[i].alpha));
        REQ(upload_vec(h, m, p + ".ff.net.0.beta_inv", &w.tr[i].binv));
        if (w.tr[i].qkv.Cout != 3 * h->dims.heads * 64) return fail(h, "qkv width %d != 3*heads*64", w.tr[i].qkv.Cout);
        qkv_pack_scales(T_(p + ".norm1.weight"), T_(p + ".norm1.bias"), {q, k, v}, w.tr[i].qkv_scale);
    }
    {
        const HostTensor *d0w = T_("down_blocks.0.2.conv.weight"), *d0b = T_("down_blocks.0.2.conv.bias");
        const HostTensor *d1w = T_("down_blocks.1.2.weight"), *d1b = T_("down_blocks.1.2.bias");
        const HostTensor *u0w = T_("up_blocks.0.2.conv.weight"), *u0b = T_("up_blocks.0.2.conv.bias");
        const HostTensor *u1w = T_("up_blocks.1.2.weight"), *u1b = T_("up_blocks.1.2.bias");
        const HostTensor *fw = T_("final_block.block.0.weight"), *fb = T_("final_block.block.0.bias");
        const HostTensor *pw = T_("final_proj.weight"), *pb = T_("final_proj.bias");
        if (!d0w || !d0b || !d1w || !d1b || !u0w || !u0b || !u1w || !u1b || !fw || !fb || !pw || !pb) return 1;
        REQ(pack_conv_stride2(h, w.down0, *d0w, d0b));
        REQ(pack_conv(h, w.down1, *d1w, d1b, 1));
        REQ(pack_convT(h, w.up0, *u0w, u0b, 2, 1));
        REQ(pack_conv(h, w.up1, *u1w, u1b, 1));
        REQ(pack_conv(h, w.fin_conv, *fw, fb, 1));
        REQ(pack_conv(h, w.fin_proj, *pw, pb, 1));
        REQ(upload_vec(h, m, "final_block.block.1.weight", &w.fin_g));
        REQ(upload_vec(h, m, "final_block.block.1.bias", &w.fin_b));
    }
    if (!h->zeros) {      // stand-in bias of the fused kernels' unconditional loads (a layer without bias reads zeros)
        HIPCHK(h, hipMalloc((void**)&h->zeros, 4096 * sizeof(float)));
        HIPCHK(h, hipMemset(h->zeros, 0, 4096 * sizeof(float)));
        h->owned.push_back(h->zeros);
    }
    if (!h->cap_pool) {   // pinned staging of captured ev_cfm_decode calls (see run_time_mlp): EV_CAPTURE_SLOTS x (64 steps x in_ch floats)
        h->cap_stride = ((size_t)64 * w.in_ch * sizeof(float) + 255) & ~(size_t)255;
        HIPCHK(h, hipHostMalloc((void**)&h->cap_pool, h->cap_stride * EV_CAPTURE_SLOTS, hipHostMallocDefault));
    }
    w.loaded = true;
    return 0;
}

int ev_load_vocoder(ev_handle* h, const float* blob, const ev_tensor_index* index, size_t n) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    TensorMap m;
    if (build_map(h, blob, index, n, m)) return 1;
    VocoderW& v = h->voc;
    v.loaded = false;
    const int rates[4] = {8, 8, 2, 2}, ksz[4] = {16, 16, 4, 4}, rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
    const HostTensor *pw = T_("conv_pre.weight"), *pb = T_("conv_pre.bias"), *qw = T_("conv_post.weight"), *qb = T_("conv_post.bias");
    if (!pw || !pb || !qw || !qb) return 1;
    REQ(pack_conv(h, v.pre, *pw, pb, 1));
    REQ(pack_conv(h, v.post, *qw, qb, 1));
    {   // conv_post_kernel's copy: (1, C, K) -> [K][C]
        const int C = (int)qw->shape[1], K = (int)qw->shape[2];
        std::vector<float> wt((size_t)K * C);
        for (int c = 0; c < C; ++c) for (int k = 0; k < K; ++k) wt[(size_t)k * C + c] = qw->p[(size_t)c * K + k];
        REQ(dev_upload(h, wt, &v.post_w));
        v.post_b = qb->p[0]; v.post_k = K;
    }
    v.ch[0] = (int)pw->shape[0];
    for (int i = 0; i < 4; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "ups.%d.weight", i); const HostTensor* uw = T_(nm);
        snprintf(nm, sizeof nm, "ups.%d.bias", i); const HostTensor* ub = T_(nm);
        if (!uw || !ub) return 1;
        if ((int)uw->shape[2] != ksz[i]) return fail(h, "ups.%d kernel %d != %d", i, (int)uw->shape[2], ksz[i]);
        REQ(pack_convT(h, v.ups[i], *uw, ub, rates[i], (ksz[i] - rates[i]) / 2));
        v.ch[i + 1] = (int)uw->shape[1];
        for (int j = 0; j < 3; ++j)
            for (int mm = 0; mm < 3; ++mm) {
                snprintf(nm, sizeof nm, "resblocks.%d.convs1.%d.weight", i * 3 + j, mm); const HostTensor* w1 = T_(nm);
                snprintf(nm, sizeof nm, "resblocks.%d.convs1.%d.bias", i * 3 + j, mm); const HostTensor* b1 = T_(nm);
                snprintf(nm, sizeof nm, "resblocks.%d.convs2.%d.weight", i * 3 + j, mm); const HostTensor* w2 = T_(nm);
                snprintf(nm, sizeof nm, "resblocks.%d.convs2.%d.bias", i * 3 + j, mm); const HostTensor* b2 = T_(nm);
                if (!w1 || !b1 || !w2 || !b2) return 1;
                if ((int)w1->shape[2] != rk[j]) return fail(h, "resblock kernel mismatch");
                REQ(pack_conv(h, v.c1[i * 3 + j][mm], *w1, b1, rd[mm]));
                REQ(pack_conv(h, v.c2[i * 3 + j][mm], *w2, b2, 1));
            }
    }
    REQ(ensure_sk(h, false));   // the balanced builds' hand-off area: a mid-size ev_hifigan call (e.g. 8 x 516 frames) takes them — never allocate on the request path
    v.loaded = true;
    return 0;
}
#undef T_
#undef REQ

#define T_(k) find(h, m, k)
#define REQ(x) do { if (x) return 1; } while (0)
int ev_load_text_encoder(ev_handle* h, const float* blob, const ev_tensor_index* index, size_t n) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    TensorMap m;
    if (build_map(h, blob, index, n, m)) return 1;
    TextEncW& w = h->enc;
    w.loaded = false;
    w.layers.clear();
    const HostTensor* emb = T_("emb.weight");
    const HostTensor* th = T_("rope_theta");
    if (!emb || !th) return 1;
    w.nvocab = (int)emb->shape[0]; w.nch = (int)emb->shape[1];
    REQ(upload_vec(h, m, "emb.weight", &w.emb));
    REQ(upload_vec(h, m, "rope_theta", &w.theta));
    for (int i = 0; i < 3; ++i) {
        const std::string c = "prenet.conv_layers." + std::to_string(i), nn = "prenet.norm_layers." + std::to_string(i);
        const HostTensor *cw = T_(c + ".weight"), *cb = T_(c + ".bias");
        if (!cw || !cb) return 1;
        REQ(pack_conv(h, w.pre[i], *cw, cb, 1));
        REQ(upload_vec(h, m, nn + ".gamma", &w.pre_g[i]));
        REQ(upload_vec(h, m, nn + ".beta", &w.pre_b[i]));
    }
    {
        const HostTensor *pw = T_("prenet.proj.weight"), *pb = T_("prenet.proj.bias");
        if (!pw || !pb) return 1;
        REQ(pack_conv(h, w.pre_proj, *pw, pb, 1));
    }
    int nl = 0;
    while (m.count("encoder.attn_layers." + std::to_string(nl) + ".conv_q.weight")) ++nl;
    if (nl == 0) return fail(h, "text encoder: no encoder.attn_layers.* tensors");
    w.nlayers = nl;
    w.layers.resize(nl);
    for (int l = 0; l < nl; ++l) {
        const std::string a = "encoder.attn_layers." + std::to_string(l), f = "encoder.ffn_layers." + std::to_string(l);
        const HostTensor *q = T_(a + ".conv_q.weight"), *k = T_(a + ".conv_k.weight"), *v = T_(a + ".conv_v.weight");
        const HostTensor *qb = T_(a + ".conv_q.bias"), *kb = T_(a + ".conv_k.bias"), *vb = T_(a + ".conv_v.bias");
        const HostTensor *ow = T_(a + ".conv_o.weight"), *ob = T_(a + ".conv_o.bias");
        const HostTensor *f1w = T_(f + ".conv_1.weight"), *f1b = T_(f + ".conv_1.bias"), *f2w = T_(f + ".conv_2.weight"), *f2b = T_(f + ".conv_2.bias");
        if (!q || !k || !v || !qb || !kb || !vb || !ow || !ob || !f1w || !f1b || !f2w || !f2b) return 1;
        EncLayerW& L = w.layers[l];
        REQ(pack_linear_stack(h, L.qkv, {q, k, v}, {qb, kb, vb}));
        REQ(pack_linear_stack(h, L.out, {ow}, {ob}));
        REQ(pack_conv(h, L.ff1, *f1w, f1b, 1));
        REQ(pack_conv(h, L.ff2, *f2w, f2b, 1));
        REQ(upload_vec(h, m, "encoder.norm_layers_1." + std::to_string(l) + ".gamma", &L.g1));
        REQ(upload_vec(h, m, "encoder.norm_layers_1." + std::to_string(l) + ".beta", &L.b1));
        REQ(upload_vec(h, m, "encoder.norm_layers_2." + std::to_string(l) + ".gamma", &L.g2));
        REQ(upload_vec(h, m, "encoder.norm_layers_2." + std::to_string(l) + ".beta", &L.b2));
    }
    w.C = w.layers[0].out.Cout;
    w.ffc = w.layers[0].ff1.Cout;
    w.heads = 2;                                           // TextEncoder's n_heads (configs/model/encoder/default.yaml)
    {
        const int kc = w.C / w.heads;                      // 128 (multi-speaker) or 96 (single-speaker); rotary on int(kc * 0.5) features
        if (w.C % w.heads || kc > 128 || (kc & 3) || (int)th->numel() != kc / 4)
            return fail(h, "text encoder: head width %d with a rope table of %d entries is not supported", kc, (int)th->numel());
    }
    if (w.C != w.nch + h->dims.spk_emb_dim && w.C != w.nch) return fail(h, "text encoder width %d != n_channels %d (+ spk_emb_dim)", w.C, w.nch);
    {
        const HostTensor *mw = T_("proj_m.weight"), *mb = T_("proj_m.bias");
        const HostTensor *d1w = T_("proj_w.conv_1.weight"), *d1b = T_("proj_w.conv_1.bias"), *d2w = T_("proj_w.conv_2.weight"), *d2b = T_("proj_w.conv_2.bias");
        const HostTensor *dpw = T_("proj_w.proj.weight"), *dpb = T_("proj_w.proj.bias");
        if (!mw || !mb || !d1w || !d1b || !d2w || !d2b || !dpw || !dpb) return 1;
        REQ(pack_conv(h, w.proj_m, *mw, mb, 1));
        REQ(pack_conv(h, w.dp1, *d1w, d1b, 1));
        REQ(pack_conv(h, w.dp2, *d2w, d2b, 1));
        REQ(pack_conv(h, w.dp_proj, *dpw, dpb, 1));
        REQ(upload_vec(h, m, "proj_w.norm_1.gamma", &w.dp_g1));
        REQ(upload_vec(h, m, "proj_w.norm_1.beta", &w.dp_b1));
        REQ(upload_vec(h, m, "proj_w.norm_2.gamma", &w.dp_g2));
        REQ(upload_vec(h, m, "proj_w.norm_2.beta", &w.dp_b2));
    }
    if (w.proj_m.Cout != 80) return fail(h, "proj_m width %d != 80", w.proj_m.Cout);
    w.loaded = true;
    return 0;
}

#undef T_
#undef REQ

int ev_text_encoder(ev_handle* h, const int64_t* d_ids, const int32_t* d_lengths, const float* d_spk, int B, int Tx, float* d_mu, float* d_logw, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->enc.loaded) return fail(h, "text encoder weights not loaded");
    if (B <= 0 || Tx <= 0 || !d_ids || !d_lengths || !d_mu || !d_logw) return fail(h, "bad arguments B=%d Tx=%d", B, Tx);
    if (h->enc.C > h->enc.nch && !d_spk) return fail(h, "speaker embedding required by a multi-speaker text encoder");
    if ((double)B * (Tx + 4) * 3 * h->enc.C * 4.0 >= 4294967296.0) return fail(h, "text batch exceeds the 4 GiB buffer-addressing limit: split the batch");
    h->stream = (hipStream_t)stream;
    if (!h->bad_ids_host) {
        HIPCHK(h, hipHostMalloc((void**)&h->bad_ids_host, 64, hipHostMallocMapped));
        *h->bad_ids_host = 0;
        HIPCHK(h, hipHostGetDevicePointer((void**)&h->bad_ids_dev, h->bad_ids_host, 0));
    }
    return run_text_encoder(h, d_ids, d_lengths, d_spk, B, Tx, d_mu, d_logw);
}

int ev_text_encoder_status(ev_handle* h, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    if (h->bad_ids_host && *(volatile int*)h->bad_ids_host) {
        *(volatile int*)h->bad_ids_host = 0;
        return fail(h, "index out of range in self: a token id passed to ev_text_encoder is outside [0, %d)", h->enc.nvocab);
    }
    return 0;
}

int ev_set_mrf_streams_max(ev_handle* h, int max_frames) {
    if (!h) return 1;
    if (max_frames < 0) return fail(h, "ev_set_mrf_streams_max: negative limit");
    h->mrf_max_frames = max_frames;
    h->ws_B = -1;   // the plan depends on the limit (scratch of the second and third chain): re-plan on the next call
    return 0;
}

size_t ev_workspace_bytes(ev_handle* h, int B, int Tp_cfm, int T_voc) {
    if (!h || B <= 0) return 0;
    return plan_all(h, nullptr, B, Tp_cfm, T_voc, nullptr, nullptr);
}

static int check_cfm_args(ev_handle* h, int B, int Tp) {
    if (!h->est.loaded) return fail(h, "estimator weights not loaded");
    if (B <= 0 || Tp <= 0 || (Tp & 3)) return fail(h, "bad shape B=%d Tp=%d (Tp must be a positive multiple of 4)", B, Tp);
    return 0;
}

static int cfm_decode_impl(ev_handle* h, const float* d_mu, const int32_t* d_lengths, const float* d_spk, const float* d_z,
                           int B, int Tp, int n_steps, float* d_dec, float out_scale, float out_shift, float* d_out, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (check_cfm_args(h, B, Tp)) return 1;
    if (n_steps <= 0 || n_steps > 65536) return fail(h, "n_steps %d must be positive (cli.py:143) and at most 65536", n_steps);
    if (!d_mu || !d_z || (!d_out && !d_dec) || (h->dims.spk_emb_dim > 0 && !d_spk)) return fail(h, "null tensor argument");
    h->stream = (hipStream_t)stream;
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    const bool capturing = h->stream && hipStreamIsCapturing(h->stream, &cst) == hipSuccess && cst == hipStreamCaptureStatusActive;
    // A captured call's kernels address the workspace as planned for its (B, Tp).  The handle may hold captured calls of SEVERAL shapes
    // (a serving loop keeps one graph per utterance length, emojivoice_amd.matcha_tts.DecodeGraphs) and serve eager calls of any shape
    // in between: every call — captured or eager — of a handle that holds graphs starts by re-zeroing the estimator's part of the
    // arena for its own plan (below), so no call depends on what an earlier one left in the pad rows.  What stays forbidden is anything
    // that would REPLACE the arena (growth beyond the reservation, more Euler steps than planned): the graphs address it.
    if (n_steps > h->max_steps) {
        // the reference has no upper limit on n_timesteps: the per-step time-embedding buffers are re-planned when a call asks for more
        if (h->captured || capturing) return fail(h, "n_steps %d exceeds the %d the workspace is planned for and the handle is captured / capturing: call once eagerly first", n_steps, h->max_steps);
        while (n_steps > h->max_steps) h->max_steps *= 2;
        h->ws_B = -1;
    }
    if (capturing) {
        if (plan_all(h, nullptr, B, Tp, 0, nullptr, nullptr) > h->ws_bytes)
            return fail(h, "ev_cfm_decode under stream capture at (B=%d, Tp=%d) needs a workspace that is already large enough: ev_reserve (or one eager call at "
                           "the largest shape) first — planning must not allocate under capture", B, Tp);
        h->captured = true; h->cap_B = B; h->cap_Tp = Tp;
    }
    if (h->captured) h->ws_B = -1;          // (forces the re-zero of this call's plan: a memset + nothing else, capturable)
    EstBufs b;
    if (ensure_ws(h, B, Tp, 0, &b, nullptr)) return 1;
    // time grid exactly as torch.linspace(0, 1, n+1) + the running t / dt of solve_euler (flow_matching.py:52,70-83), in fp32
    const int steps = n_steps + 1;
    std::vector<float> span(steps);
    {
        const float stepf = (1.0f - 0.0f) / (float)(steps - 1);
        const int halfway = steps / 2;
        for (int i = 0; i < steps; ++i) span[i] = (i < halfway) ? (0.0f + stepf * (float)i) : (1.0f - stepf * (float)(steps - i - 1));
    }
    std::vector<float> ts(n_steps), dts(n_steps);
    {
        float t = span[0], dt = span[1] - span[0];
        for (int s = 1; s <= n_steps; ++s) {
            ts[s - 1] = t; dts[s - 1] = dt;
            t = t + dt;
            if (s < n_steps) dt = span[s + 1] - t;
        }
    }
    if (prep_inputs(h, b, d_z, d_mu, d_lengths, d_spk, B, Tp)) return 1;
    // the time MLP's output for this step count: from the handle's cache, or computed now and cached for the next call.  A CAPTURED call
    // must find it there: a host-to-device copy captured from pinned memory replays from a staging buffer of the HIP runtime that later
    // eager copies recycle (measured: the second eager decode after a capture corrupts every later replay, tools/graph_debug2.py,
    // profiles/r04_graph_capture_h2d_hazard.txt) — so a captured decode contains kernels and one memset, nothing else.
    const float* tproj = nullptr;
    for (auto& c : h->tproj_cache) if (c.nt == n_steps) tproj = c.d;
    if (!tproj) {
        if (capturing) return fail(h, "ev_cfm_decode under stream capture: no eager decode with %d Euler steps has run on this handle yet (the time embeddings of a "
                                      "captured call must already be on the device: call once eagerly with this step count first)", n_steps);
        if (run_time_mlp(h, b, ts)) return 1;
        // keep it for the next decode with this step count: at most 8 step counts; a handle that holds graphs never drops an entry (a graph may
        // address it) and simply stops caching new ones, any other handle drops its oldest
        const size_t bytes = (size_t)n_steps * 1536 * sizeof(float);
        bool room = h->tproj_cache.size() < 8;
        if (!room && !h->captured) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            hipFree(h->tproj_cache.front().d);
            h->tproj_cache.erase(h->tproj_cache.begin());
            room = true;
        }
        float* d = nullptr;
        if (room && n_steps <= 4096) {
            if (hipMalloc((void**)&d, bytes) == hipSuccess) {
                ++h->n_allocs;
                HIPCHK(h, hipMemcpyAsync(d, b.tproj, bytes, hipMemcpyDeviceToDevice, h->stream));
                h->tproj_cache.push_back({n_steps, d});
            } else (void)hipGetLastError();
        }
    }
    for (int s = 0; s < n_steps; ++s)
        if (run_estimator(h, b, s, dts[s], true, tproj)) return 1;
    dim3 grid((Tp + 31) / 32, (80 + 31) / 32, B);
    if (d_dec) hipLaunchKernelGGL(fm_to_cm_kernel, grid, dim3(256), 0, h->stream, (const float*)b.state, 80, 0, d_dec, 80, Tp, b.g0.S, b.g0.P, 1.0f, 0.0f);
    if (d_out) hipLaunchKernelGGL(fm_to_cm_kernel, grid, dim3(256), 0, h->stream, (const float*)b.state, 80, 0, d_out, 80, Tp, b.g0.S, b.g0.P, out_scale, out_shift);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int ev_cfm_decode(ev_handle* h, const float* d_mu, const int32_t* d_lengths, const float* d_spk, const float* d_z,
                  int B, int Tp, int n_steps, float out_scale, float out_shift, float* d_out, void* stream) {
    if (h && !d_out) return fail(h, "null tensor argument");
    return cfm_decode_impl(h, d_mu, d_lengths, d_spk, d_z, B, Tp, n_steps, nullptr, out_scale, out_shift, d_out, stream);
}

int ev_cfm_decode2(ev_handle* h, const float* d_mu, const int32_t* d_lengths, const float* d_spk, const float* d_z,
                   int B, int Tp, int n_steps, float* d_dec, float mel_std, float mel_mean, float* d_mel, void* stream) {
    return cfm_decode_impl(h, d_mu, d_lengths, d_spk, d_z, B, Tp, n_steps, d_dec, mel_std, mel_mean, d_mel, stream);
}

int ev_reserve(ev_handle* h, int B, int Tx_max, int Tp_max, int T_voc_max, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (B <= 0 || Tx_max < 0 || Tp_max < 0 || T_voc_max < 0 || (Tp_max & 3)) return fail(h, "ev_reserve: bad shape B=%d Tx=%d Tp=%d (multiple of 4) T_voc=%d", B, Tx_max, Tp_max, T_voc_max);
    h->stream = (hipStream_t)stream;
    if (ensure_sk(h, false)) return 1;                // (normally there since the loaders; a handle reserved before any load gets it here)
    if (Tp_max > 0 || T_voc_max > 0) {
        // the plan is monotonic in every argument except for the three-stream scratch of small vocoder calls: take the larger
        size_t need = plan_all(h, nullptr, B, Tp_max, T_voc_max, nullptr, nullptr);
        if (T_voc_max > 0 && h->mrf_max_frames > 0) {
            const int tv_ms = (int)std::min<long>(T_voc_max, h->mrf_max_frames / B);
            if (tv_ms > 0) need = std::max(need, plan_all(h, nullptr, B, Tp_max, tv_ms, nullptr, nullptr));
        }
        if (need > h->ws_bytes) {
            if (h->captured) return fail(h, "ev_reserve: this handle holds a captured ev_cfm_decode, its workspace cannot be replaced");
            if (grow_ws(h, need, false)) return 1;
        }
    }
    if (Tx_max > 0 && h->enc.loaded) {
        const size_t n = (size_t)B * (Tx_max + 4);
        const TextEncW& w = h->enc;
        const size_t need = (n * (1 + 3 * w.nch + 3 * w.C + 2 * w.dp1.Cout + 3 * w.C + w.C + w.ffc + 80 + 4) + 1024) * sizeof(float);
        if (scratch_acquire(h, h->enc_ws, need)) return 1;
    }
    if (T_voc_max > 0) {   // denoiser scratch for L = 256 * T_voc_max (run_denoiser)
        const size_t n = (size_t)B * (T_voc_max + 4 + 8);
        const size_t need = (n * (256 + 1032 + 256) + 1024) * sizeof(float);
        if (scratch_acquire(h, h->dn_ws, need)) return 1;
        if (denoiser_init(h)) return 1;
    }
    if (h->est.loaded) {   // both slots of the pinned time-embedding ring, for the Euler steps the workspace is planned for
        const size_t bytes = (size_t)std::max(64, h->max_steps) * h->est.in_ch * sizeof(float);
        for (int slot = 0; slot < 2; ++slot) {
            if (h->temb_cap[slot] >= bytes) continue;
            if (h->temb_host[slot]) { HIPCHK(h, hipEventSynchronize(h->temb_ev[slot])); HIPCHK(h, hipHostFree(h->temb_host[slot])); h->temb_host[slot] = nullptr; h->temb_cap[slot] = 0; }
            HIPCHK(h, hipHostMalloc((void**)&h->temb_host[slot], bytes, hipHostMallocDefault));
            h->temb_cap[slot] = bytes;
            if (!h->temb_ev[slot]) HIPCHK(h, hipEventCreateWithFlags(&h->temb_ev[slot], hipEventDisableTiming));
            HIPCHK(h, hipEventRecord(h->temb_ev[slot], h->stream));
        }
    }
    if (h->voc.loaded && h->mrf_max_frames > 0 && !h->mrf_stream[0]) {   // streams / events of the three-chain fan-out
        for (int c = 0; c < 2; ++c) HIPCHK(h, hipStreamCreateWithFlags(&h->mrf_stream[c], hipStreamNonBlocking));
        for (int c = 0; c < 4; ++c) HIPCHK(h, hipEventCreateWithFlags(&h->mrf_ev[c], hipEventDisableTiming));
    }
    return 0;
}

int64_t ev_alloc_count(ev_handle* h) { return h ? h->n_allocs : -1; }

int ev_dbg_sk_stats(ev_handle* h, uint32_t* out3) {
    if (!h || !out3) return 1;
    out3[0] = out3[1] = out3[2] = 0;
    if (!h->sk_ctrl) return 0;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out3, h->sk_ctrl, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

// Diagnostic (ABI 4): contributor shares that owners of the balanced launches took over because the contributor had not started (work
// stealing, ev_kernels.h sk_wait_many), since the handle was created; -1 on error.  Synchronises the device.
int64_t ev_dbg_sk_taken(ev_handle* h) {
    if (!h) return -1;
    if (!h->sk_ctrl) return 0;
    uint32_t v = 0;
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, h->sk_ctrl + 3, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (int64_t)v;
}

int ev_estimator(ev_handle* h, const float* d_x, const float* d_mu, const int32_t* d_lengths, const float* d_spk, float t,
                 int B, int Tp, float* d_v, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (check_cfm_args(h, B, Tp)) return 1;
    h->stream = (hipStream_t)stream;
    EstBufs b;
    if (ensure_ws(h, B, Tp, 0, &b, nullptr)) return 1;
    if (prep_inputs(h, b, d_x, d_mu, d_lengths, d_spk, B, Tp)) return 1;
    if (run_time_mlp(h, b, std::vector<float>(1, t))) return 1;
    if (run_estimator(h, b, 0, 0.f, false)) return 1;
    dim3 grid((Tp + 31) / 32, (80 + 31) / 32, B);
    hipLaunchKernelGGL(fm_to_cm_kernel, grid, dim3(256), 0, h->stream, (const float*)b.V0, 80, 0, d_v, 80, Tp, b.g0.S, b.g0.P, 1.0f, 0.0f);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int ev_align(ev_handle* h, const float* d_wceil, const float* d_mu_x, const int32_t* d_xlen, const int64_t* d_ylen, int B, int Tx, int Tp,
             float* d_mu_y, float* d_attn, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (B <= 0 || Tx <= 0 || Tp <= 0 || Tx > 8192 || !d_wceil || !d_mu_x || !d_xlen || !d_ylen || !d_mu_y) return fail(h, "bad arguments B=%d Tx=%d Tp=%d", B, Tx, Tp);
    hipLaunchKernelGGL(enc_align_kernel, dim3(B), dim3(256), (size_t)Tx * sizeof(float), (hipStream_t)stream, d_wceil, d_mu_x, d_xlen, d_ylen, d_mu_y, d_attn,
                       h->dims.n_feats, Tx, Tp);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int ev_stft_magnitude(ev_handle* h, const float* d_audio, int B, int L, float* d_mag, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (B <= 0 || L < 768 || (L & 255) || !d_audio || !d_mag) return fail(h, "bad arguments B=%d L=%d (L must be a multiple of 256, >= 768: reflect padding of 512 needs more than 512 samples, as in torch.stft)", B, L);
    h->stream = (hipStream_t)stream;
    return run_denoiser(h, d_audio, B, L, nullptr, 0.f, nullptr, d_mag);
}

int ev_denoise(ev_handle* h, const float* d_audio, int B, int L, const float* d_bias_spec, float strength, float* d_out, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (B <= 0 || L < 768 || (L & 255) || !d_audio || !d_bias_spec || !d_out) return fail(h, "bad arguments B=%d L=%d (L must be a multiple of 256, >= 768: reflect padding of 512 needs more than 512 samples, as in torch.stft)", B, L);
    if ((double)B * (L / 256 + 12) * 1032 * 4.0 >= 4294967296.0) return fail(h, "audio batch exceeds the 4 GiB buffer-addressing limit: split the batch");
    h->stream = (hipStream_t)stream;
    return run_denoiser(h, d_audio, B, L, d_bias_spec, strength, d_out, nullptr);
}

int ev_hifigan(ev_handle* h, const float* d_mel, int B, int T, float* d_wav, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->voc.loaded) return fail(h, "vocoder weights not loaded");
    if (B <= 0 || T <= 0 || !d_mel || !d_wav) return fail(h, "bad arguments B=%d T=%d", B, T);
    h->stream = (hipStream_t)stream;
    VocBufs v;
    if (ensure_ws(h, B, 0, T, nullptr, &v)) return 1;
    const VocoderW& w = h->voc;
    dim3 grid((T + 31) / 32, (80 + 31) / 32, B);
    hipLaunchKernelGGL(cm_to_fm_kernel, grid, dim3(256), 0, h->stream, d_mel, v.M0, 80, 0, 80, T, v.g[0].S, v.g[0].P, (const float*)nullptr, 1.0f);
    HIPCHK(h, hipGetLastError());
    // amax slots: every tensor of the plan has per-128-row bounds of what this call stores in it (ev_kernels.h, "Tile maxima from the producer");
    // zeroed here, left by each producer, read by the fp16 builds in place of their pre-scan.  `has[k]`: tensor k's slots hold bounds of its
    // present contents (false until a build that leaves them has written it, and again after one that does not).
    // Every tensor has TWO sets of slots and `cur[k]` says which one describes it: a launch that adds to a running sum (accum) reads the old
    // bound from the current set and leaves the new one in the OTHER set, then the sets swap — a bound read from slots the same launch is
    // updating would depend on timing, and the tile scales (hence the bits of the result) with it.
    bool has[64] = {false};
    int cur[64] = {0};
    if (v.slots) HIPCHK(h, hipMemsetAsync(v.slots, 0, v.slot_words * sizeof(unsigned), h->stream));
    auto with_slots = [&](Epi& e, const float* x, const float* r, float* y, int ymul) {   // before a launch
        const int ix = v.index_of(x), ir = r ? v.index_of(r) : -1, iy = v.index_of(y);
        if (!v.slots) return;
        if (ix >= 0 && has[ix]) { e.xmax = v.zl[ix].amax[cur[ix]]; e.xmax_n = v.zl[ix].amax_n; }
        if (ir >= 0 && has[ir]) e.rmax = v.zl[ir].amax[cur[ir]];
        if (iy >= 0 && !e.accum) { e.ymax = v.zl[iy].amax[cur[iy]]; e.ymax_mul = ymul; }
        if (iy >= 0 && e.accum && has[iy]) { e.yold = v.zl[iy].amax[cur[iy]]; e.ymax = v.zl[iy].amax[cur[iy] ^ 1]; e.ymax_mul = ymul; }
    };
    auto wrote = [&](float* y, bool accum) {                  // after it
        const int iy = v.index_of(y);
        if (iy < 0) return;
        has[iy] = h->amax_emitted;
        if (accum && h->amax_emitted) cur[iy] ^= 1;
    };
    {   // conv_pre, with the first leaky_relu (models.py:184) fused as the epilogue: only ups[0] consumes it
        Epi e; e.act = ACT_LRELU; e.act_slope = 0.1f;
        with_slots(e, v.M0, nullptr, v.C0, 1);
        if (launch_conv(h, w.pre, v.M0, 80, v.C0, w.ch[0], v.g[0], e)) return 1;
        wrote(v.C0, false);
    }
    const float* xin = v.C0;
    int cin = w.ch[0];
    // The three ResBlock1 chains of a level are independent until they join the running sum, so they go to three streams and fill
    // each other's gaps: a streaming utterance is 1-4 workgroups per CU and launch, every launch a lock-step stage -> MFMA ->
    // epilogue sequence with a quantisation tail (batch 1: -13 % at T = 516; batch 16: -3 %).  The joins keep the serial order
    // (x3 + x7 + x11) / 3.
    const bool ms = mrf_streams_for(h, B, T) && !h->prof && v.Pax[0][1] != nullptr;
    if (ms && !h->mrf_stream[0]) {
        for (int c = 0; c < 2; ++c) HIPCHK(h, hipStreamCreateWithFlags(&h->mrf_stream[c], hipStreamNonBlocking));
        for (int c = 0; c < 4; ++c) HIPCHK(h, hipEventCreateWithFlags(&h->mrf_ev[c], hipEventDisableTiming));
    }
    // Work queued on the two extra streams BEHIND a busy caller's stream slows that stream down: with barrier packets pending in
    // two more hardware queues every dispatch of the caller's stream took ~5 us longer (a 6.3 ms CFM decode in front of the
    // vocoder became 8.4 ms; tools/stream_split.py).  So the call first waits for the caller's stream to drain — the host has
    // nothing else to enqueue for this utterance anyway — and only then fans out.
    if (ms) HIPCHK(h, hipStreamSynchronize(h->stream));
    hipStream_t s0 = h->stream;
    // (the fan-out runs three chains of this handle concurrently: the balanced builds' hand-off area — one per handle, launches
    // ordered by ONE stream — is off limits for them; a failure between fork and join waits for the side streams before returning)
    struct StreamGuard {
        ev_handle* h; hipStream_t s; bool bal; bool ms; bool done;
        ~StreamGuard() {
            h->stream = s; h->sk_balance = bal;
            if (ms && !done) { for (int c = 0; c < 2; ++c) if (h->mrf_stream[c]) (void)hipStreamSynchronize(h->mrf_stream[c]); }
        }
    } guard{h, s0, h->sk_balance, ms, false};
    if (ms) h->sk_balance = false;
    for (int i = 0; i < 4; ++i) {
        const int l = i + 1, C = w.ch[l], s = w.ups[i].Cout / C;
        {   // transposed conv: input frames of level l-1 -> view rows of s output frames each
            Epi e;
            with_slots(e, xin, nullptr, v.U[l], s);         // (a row of this launch = s rows of U[l]'s axis)
            if (launch_conv(h, w.ups[i], xin, cin, v.U[l], s * C, v.g[l - 1], e)) return 1;
            wrote(v.U[l], false);
        }
        if (ms) {
            HIPCHK(h, hipEventRecord(h->mrf_ev[3], s0));
            for (int c = 0; c < 2; ++c) HIPCHK(h, hipStreamWaitEvent(h->mrf_stream[c], h->mrf_ev[3], 0));
        }
        for (int j = 0; j < 3; ++j) {
            const float* x = v.U[l];
            const bool own = ms && j > 0;                      // chain j on its own stream with its own scratch
            h->stream = own ? h->mrf_stream[j - 1] : s0;
            float* pp[2] = {own ? v.Pax[j - 1][l] : v.Pa[l], own ? v.Pbx[j - 1][l] : v.Pb[l]};
            float* t1 = own ? v.T1x[j - 1][l] : v.T1[l];
            if (h->fuse_pairs && chain_ok(h, w.c1[i * 3 + j], w.c2[i * 3 + j], C)) {
                // the whole ResBlock in one launch: x never leaves the CU between its three pairs (resblock_chain_h16_kernel)
                if (ms && j > 0) HIPCHK(h, hipStreamWaitEvent(h->stream, h->mrf_ev[j - 1], 0));
                Epi e2;
                float* y = v.XS[l];
                e2.accum = (j > 0);
                if (j == 2) { e2.div3 = 1; e2.act2_lrelu = 1; e2.act2_slope = (i == 3) ? 0.01f : 0.1f; }
                if (l < 4) with_slots(e2, x, nullptr, y, 1);
                if (launch_chain(h, w.c1[i * 3 + j], w.c2[i * 3 + j], x, y, C, v.g[l], e2)) return 1;
                wrote(y, e2.accum != 0);
                if (ms) HIPCHK(h, hipEventRecord(h->mrf_ev[j], h->stream));
                continue;
            }
            for (int mm = 0; mm < 3; ++mm) {
                if (ms && mm == 2 && j > 0) HIPCHK(h, hipStreamWaitEvent(h->stream, h->mrf_ev[j - 1], 0));   // the sum so far is in XS
                Epi e2; e2.R = x; e2.ldr = C;                                          // c2 + x
                float* y = pp[mm & 1];
                if (mm == 2) {   // resblock output joins the running mean over the 3 kernel sizes (models.py:186-192)
                    y = v.XS[l];
                    e2.accum = (j > 0);
                    if (j == 2) { e2.div3 = 1; e2.act2_lrelu = 1; e2.act2_slope = (i == 3) ? 0.01f : 0.1f; }  // next consumer's leaky_relu
                }
                if (h->fuse_pairs && (C == 32 || C == 64 || (C == 128 && h->fuse128 && w.c1[i * 3 + j][mm].ntaps <= h->fuse128))) {
                    // narrow stages: both convs of the pair in one launch, intermediate kept in LDS
                    // (a fused pair finds its own tile maximum and bounds its residual with it: it needs no slots of x, and only the pair that
                    // closes a chain into the running sum of a level whose sum an fp16 build consumes — the upsampler of levels 1..3 — leaves
                    // bounds at all: an atomic per wave costs the narrow pairs 3-10 %, profiles/r04_amax_ab.txt)
                    if (mm == 2 && l < 4) with_slots(e2, x, nullptr, y, 1);
                    if (launch_pair(h, w.c1[i * 3 + j][mm], w.c2[i * 3 + j][mm], x, y, C, v.g[l], e2)) return 1;
                    wrote(y, e2.accum != 0);
                } else {
                    Epi e1; e1.pro_slope = 0.1f; e1.act = ACT_LRELU; e1.act_slope = 0.1f;   // lrelu -> c1 -> lrelu
                    with_slots(e1, x, nullptr, t1, 1);
                    if (launch_conv(h, w.c1[i * 3 + j][mm], x, C, t1, C, v.g[l], e1)) return 1;
                    wrote(t1, false);
                    with_slots(e2, t1, x, y, 1);
                    if (launch_conv(h, w.c2[i * 3 + j][mm], t1, C, y, C, v.g[l], e2)) return 1;
                    wrote(y, e2.accum != 0);
                }
                if (ms && mm == 2) HIPCHK(h, hipEventRecord(h->mrf_ev[j], h->stream));
                x = y;
            }
        }
        h->stream = s0;
        if (ms) HIPCHK(h, hipStreamWaitEvent(s0, h->mrf_ev[2], 0));   // (chain 2's join waited for chain 1's, which waited for chain 0's)
        xin = v.XS[l];
        cin = C;
    }
    {   // conv_post + tanh straight into (B, 256 T)
        const Geom& g4 = v.g[4];
        if (w.ch[4] == 32 && w.post_k == 7 && w.post_w && g4.P >= 3) {
            const int tiles = (g4.T + 255) / 256;
            hipLaunchKernelGGL((conv_post_kernel<32, 7>), dim3((unsigned)(B * tiles)), dim3(256), 0, h->stream, (const float*)v.XS[4], (const float*)w.post_w,
                               w.post_b, d_wav, g4.T, g4.S, g4.P);
            HIPCHK(h, hipGetLastError());
        } else {   // other channel / tap counts: the generic conv + a strip pass
            Epi e; e.act = ACT_TANH;
            if (launch_conv(h, w.post, v.XS[4], w.ch[4], v.T1[4], 1, g4, e)) return 1;
            const size_t total = (size_t)B * g4.T;
            hipLaunchKernelGGL(strip_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, (const float*)v.T1[4], d_wav, g4.T, g4.S, g4.P, total);
            HIPCHK(h, hipGetLastError());
        }
    }
    guard.done = true;
    return 0;
}

int ev_profile_enable(ev_handle* h, int on) {
    if (!h) return 1;
    h->prof = on != 0;
    h->ev_used = 0; h->prof_flops = 0; h->prof_launches = 0; h->prof_recs.clear();
    return 0;
}

int ev_profile_read(ev_handle* h, double* conv_ms, double* conv_flops, int64_t* conv_launches, int reset) {
    if (!h) return 1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double ms = 0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, h->ev_pool[i], h->ev_pool[i + 1]));
        ms += t;
    }
    if (const char* dump = getenv("EV_PROFILE_DUMP")) {
        if (*dump && h->prof_recs.size() * 2 == h->ev_used) {
            struct Agg { ev_handle::ProfRec r; double ms = 0, fl = 0; int n = 0; };
            std::vector<Agg> aggs;
            for (size_t i = 0; i < h->prof_recs.size(); ++i) {
                const auto& r = h->prof_recs[i];
                float t = 0;
                hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]);
                Agg* a = nullptr;
                for (auto& x : aggs)
                    if (x.r.kind == r.kind && x.r.Cin == r.Cin && x.r.Cout == r.Cout && x.r.ntaps == r.ntaps && x.r.nrows == r.nrows && x.r.cfg == r.cfg && x.r.lean == r.lean) a = &x;
                if (!a) { aggs.push_back(Agg{r}); a = &aggs.back(); }
                a->ms += t; a->fl += r.flops; a->n += 1;
            }
            if (FILE* f = fopen(dump, "a")) {
                fprintf(f, "# kind Cin Cout ntaps nrows cfg lean launches total_ms TFLOP/s\n");
                for (auto& a : aggs)
                    fprintf(f, "%s %4d %4d %3d %9d %3d %d %5d %9.3f %7.1f\n", a.r.kind == 0 ? "conv" : (a.r.kind == 1 ? "pair" : (a.r.kind == 2 ? "lnff" : (a.r.kind == 3 ? "lnqkv" : "attn "))), a.r.Cin, a.r.Cout, a.r.ntaps, a.r.nrows, a.r.cfg, a.r.lean,
                            a.n, a.ms, a.fl / (a.ms * 1e9));
                fclose(f);
            }
        }
    }
    if (conv_ms) *conv_ms = ms;
    if (conv_flops) *conv_flops = h->prof_flops;
    if (conv_launches) *conv_launches = h->prof_launches;
    if (reset) { h->ev_used = 0; h->prof_flops = 0; h->prof_launches = 0; h->prof_recs.clear(); }
    return 0;
}

int ev_dbg_last_cfg(ev_handle* h) { return h ? h->last_cfg : -1; }

// Arithmetic of the deep layers' products: 6 (default) = every fp32 product from six exact bf16 x bf16 products on the bf16 matrix pipe,
// 0 = every layer on the fp32 MFMA.  Takes effect with the next call on this handle; results of the two settings agree to fp32 rounding.
int ev_set_arithmetic(ev_handle* h, int bf16_products) {
    if (!h) return 1;
    if (bf16_products != 0 && bf16_products != 3 && bf16_products != 6 && bf16_products != 9 && bf16_products != 16)
        return fail(h, "ev_set_arithmetic: 0, 3, 6, 9 or 16, got %d", bf16_products);
    h->split_terms = bf16_products;
    return 0;
}
int ev_get_arithmetic(ev_handle* h) { return h ? h->split_terms : -1; }

// Diagnostic / A-B switch (ABI v4): 1 (default) = the fp16 builds take their tile scales from the producers' amax slots, 0 = they pre-scan
int ev_dbg_set_amax(ev_handle* h, int on) {
    if (!h) return 1;
    h->use_amax = on != 0;
    return 0;
}

// Diagnostic / A-B switch (ABI v4): 1 (default) = ev_hifigan runs the ResBlock1 chains that qualify (narrow levels, k = 3) as ONE launch each
// (resblock_chain_h16_kernel), 0 = as three fused pairs
int ev_dbg_set_chain(ev_handle* h, int on) {
    if (!h) return 1;
    h->use_chain = on != 0;
    return 0;
}

// Diagnostic / A-B switch (ABI v4): 1 (default) = under arithmetic setting 16 the fused attention runs on the fp16 pipe (attn_out_h16_kernel, q / k / v
// left as fp16 piece pairs by ln_qkv_h16_kernel), 0 = it stays on the fp32 MFMA (attn_out_kernel).  EV_NO_ATTN_H16=1 presets 0.
int ev_dbg_set_attn_h16(ev_handle* h, int on) {
    if (!h) return 1;
    h->attn_h16 = on != 0;
    return 0;
}

// The launches of the bf16-split builds (conv_split_kernel, conv_split_bal_kernel, resblock_pair_split_kernel) among those recorded
// since the last reset: call before ev_profile_read(..., reset = 1).
int ev_profile_read_split(ev_handle* h, double* ms_out, double* flops_out, int64_t* launches_out) {
    if (!h) return 1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double ms = 0, fl = 0;
    int64_t n = 0;
    if (h->prof_recs.size() * 2 == h->ev_used)
        for (size_t i = 0; i < h->prof_recs.size(); ++i) {
            const auto& r = h->prof_recs[i];
            const bool split = (r.kind == 0 && (r.cfg == 40 || r.cfg == 41 || r.cfg == 46 || r.cfg == 47 || r.cfg == 66 || r.cfg == 68 || r.cfg == 43 || r.cfg == 49 || r.cfg == 60)) || (r.kind == 1 && r.cfg >= 140) || (r.kind == 2 && (r.cfg == 120 || r.cfg == 121)) || (r.kind == 3 && r.cfg == 122) || (r.kind == 4 && r.cfg == 31);
            if (!split) continue;
            float t = 0;
            HIPCHK(h, hipEventElapsedTime(&t, h->ev_pool[2 * i], h->ev_pool[2 * i + 1]));
            ms += t; fl += r.flops; n += 1;
        }
    if (ms_out) *ms_out = ms;
    if (flops_out) *flops_out = fl;
    if (launches_out) *launches_out = n;
    return 0;
}

// Kernel microbenchmark hook (tools/conv_bench.py): times `iters` launches of one resblock-style conv
// (prologue lrelu, bias, residual) at a given geometry with HIP events; dbg = ablation bits, cfg = forced tile config.
int ev_dbg_conv_bench(ev_handle* h, int Cin, int Cout, int K, int dil, int B, int T, int P, int iters, int dbg, int cfg, float* ms_out) {
    const int stagger = (dbg & 64) ? 0 : -1;   // bit 64: disable the start stagger
    const bool planar = (dbg & 32) != 0;      // bit 32: channel-chunk-planar activations
    const bool no_res = (dbg & 128) != 0;     // bit 128: no residual input
    const bool no_pro = (dbg & 512) != 0;     // bit 512: no prologue activation (the estimator's layers)
    dbg &= ~(64 | 32 | 128 | 512);
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = nullptr;
    std::vector<float> w((size_t)Cout * Cin * K), b(Cout);
    unsigned s = 12345u;
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f; }
    for (auto& v : b) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f); }
    HostTensor wt, bt;
    wt.p = w.data(); wt.ndim = 3; wt.shape[0] = Cout; wt.shape[1] = Cin; wt.shape[2] = K;
    bt.p = b.data(); bt.ndim = 1; bt.shape[0] = Cout;
    ConvLayer L;
    size_t owned0 = h->owned.size();
    if (pack_conv(h, L, wt, &bt, dil)) return 1;
    Geom g{B * (T + 2 * P), T + 2 * P, P, T};
    float *X = nullptr, *Y = nullptr;
    const size_t nx = (size_t)g.nrows * Cin, ny = (size_t)g.nrows * Cout;
    HIPCHK(h, hipMalloc((void**)&X, nx * 4));
    HIPCHK(h, hipMalloc((void**)&Y, ny * 4));
    std::vector<float> xh(std::min(nx, (size_t)1 << 22));
    for (auto& v : xh) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    for (size_t o = 0; o < nx; o += xh.size()) HIPCHK(h, hipMemcpy(X + o, xh.data(), std::min(xh.size(), nx - o) * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(Y, 0, ny * 4));
    Epi e; e.pro_slope = no_pro ? -1.f : 0.1f; e.dbg = dbg; e.force_cfg = cfg; e.stagger = stagger;
    unsigned long long* d_st = nullptr;
    const size_t max_wgs = 1 << 16;
    if (dbg & 16) { HIPCHK(h, hipMalloc((void**)&d_st, max_wgs * 4 * 8)); HIPCHK(h, hipMemset(d_st, 0, max_wgs * 4 * 8)); e.stamps = d_st; }
    if (Cin == Cout && !no_res) { e.R = X; e.ldr = Cin; }
    int ldx_b = Cin, ldy_b = Cout;
    if (planar) {   // channel-chunk-planar layout: [C/32 planes][rows][32]
        e.isplit_log2 = 5; e.isstride = g.nrows * 32; e.osplit_log2 = 5; e.osstride = g.nrows * 32; ldx_b = 32; ldy_b = 32;
        if (e.R) e.ldr = 32;
    }
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    int rc = launch_conv(h, L, X, ldx_b, Y, ldy_b, g, e);   // warm-up
    HIPCHK(h, hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters && !rc; ++i) rc = launch_conv(h, L, X, ldx_b, Y, ldy_b, g, e);
    HIPCHK(h, hipEventRecord(e1, nullptr));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / (float)iters;
    if (d_st) {   // timeline of the LAST launch: percentiles of the four per-workgroup stamps relative to the earliest start
        std::vector<unsigned long long> st(max_wgs * 4);
        HIPCHK(h, hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> v[4];
        unsigned long long t0 = ~0ull;
        for (size_t i = 0; i < max_wgs; ++i) if (st[4 * i] && st[4 * i] < t0) t0 = st[4 * i];
        for (size_t i = 0; i < max_wgs; ++i) if (st[4 * i]) for (int k = 0; k < 4; ++k) v[k].push_back((double)(st[4 * i + k] - t0) / 100.0);  // 100 MHz -> us
        const char* nm[4] = {"start", "stage0", "kloop", "end"};
        for (int k = 0; k < 4; ++k) {
            std::sort(v[k].begin(), v[k].end());
            const size_t n = v[k].size();
            if (n) fprintf(stderr, "  stamp %-6s n=%zu  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us\n", nm[k], n, v[k][0], v[k][n / 10], v[k][n / 2], v[k][n * 9 / 10], v[k][n - 1]);
        }
        hipFree(d_st);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(X); hipFree(Y);
    while (h->owned.size() > owned0) { hipFree(h->owned.back()); h->owned.pop_back(); }
    return rc;
}

// ---------------------------------------------------------------------------
// operator-level entry points for unit parity tests
// ---------------------------------------------------------------------------
int ev_op_conv1d(ev_handle* h, const float* d_x, const float* w, const float* bias, int B, int Cin, int T, int Cout, int K,
                 int dilation, int transposed, int stride, int padding, float pre_lrelu_slope, float* d_y, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    ConvLayer L;
    HostTensor wt, bt;
    wt.p = w; wt.ndim = 3;
    bt.p = bias; bt.ndim = 1;
    int Tout = T, P = 0;
    size_t owned0 = h->owned.size();
    if (!transposed && stride == 1) {
        wt.shape[0] = Cout; wt.shape[1] = Cin; wt.shape[2] = K; bt.shape[0] = Cout;
        if (padding != (K * dilation - dilation) / 2) return fail(h, "only 'same' padding is supported");
        if (pack_conv(h, L, wt, bias ? &bt : nullptr, dilation)) return 1;
    } else if (!transposed && stride == 2) {
        wt.shape[0] = Cout; wt.shape[1] = Cin; wt.shape[2] = K; bt.shape[0] = Cout;
        if (K != 3 || padding != 1 || (T & 1)) return fail(h, "stride-2 conv: k=3, p=1, even T");
        if (pack_conv_stride2(h, L, wt, bias ? &bt : nullptr)) return 1;
        Tout = T / 2;
    } else {
        wt.shape[0] = Cin; wt.shape[1] = Cout; wt.shape[2] = K; bt.shape[0] = Cout;
        if (pack_convT(h, L, wt, bias ? &bt : nullptr, stride, padding)) return 1;
        if ((K - 2 * padding) != stride) return fail(h, "transposed conv must satisfy K - 2p == stride");
        Tout = T * stride;
    }
    // geometry: pads large enough for the halo, compatible with the view factor
    const int vf = (!transposed && stride == 2) ? 2 : 1;
    P = 32;
    int rc = 0;
    float *X = nullptr, *Y = nullptr;
    unsigned* XM = nullptr;
    do {
        Geom gin{B * (T + 2 * P), T + 2 * P, P, T};
        const int Pout = transposed ? P * stride : P / vf;
        Geom gout{B * (Tout + 2 * Pout), Tout + 2 * Pout, Pout, Tout};
        const size_t nx = (size_t)gin.nrows * Cin, ny = (size_t)gout.nrows * Cout;
        if (hipMalloc((void**)&X, nx * 4) != hipSuccess || hipMalloc((void**)&Y, ny * 4) != hipSuccess) { rc = fail(h, "hipMalloc failed"); break; }
        hipMemsetAsync(X, 0, nx * 4, h->stream);
        hipMemsetAsync(Y, 0, ny * 4, h->stream);
        dim3 g1((T + 31) / 32, (Cin + 31) / 32, B);
        hipLaunchKernelGGL(cm_to_fm_kernel, g1, dim3(256), 0, h->stream, d_x, X, Cin, 0, Cin, T, gin.S, gin.P, (const float*)nullptr, 1.0f);
        Epi e; e.pro_slope = pre_lrelu_slope;
        if (h->use_amax && !(!transposed && stride == 2)) {   // the input's amax slots, exact (amax_rows_kernel): the fp16 builds then take their tile scale from them
            const int ns = (gin.nrows >> 7) + 2;
            if (hipMalloc((void**)&XM, (size_t)ns * 4) != hipSuccess) { rc = fail(h, "hipMalloc failed"); break; }
            hipMemsetAsync(XM, 0, (size_t)ns * 4, h->stream);
            hipLaunchKernelGGL(amax_rows_kernel, dim3((gin.nrows + 127) / 128), dim3(256), 0, h->stream, (const float*)X, Cin, Cin, gin.nrows, XM);
            e.xmax = XM; e.xmax_n = ns;
        }
        if (transposed) {
            rc = launch_conv(h, L, X, Cin, Y, stride * Cout, gin, e);   // view rows = input frames
        } else if (stride == 2) {
            Geom gv{gin.nrows / 2, gin.S / 2, gin.P / 2, T / 2};
            rc = launch_conv(h, L, X, 2 * Cin, Y, Cout, gv, e);
        } else {
            rc = launch_conv(h, L, X, Cin, Y, Cout, gin, e);
        }
        if (rc) break;
        dim3 g2((Tout + 31) / 32, (Cout + 31) / 32, B);
        hipLaunchKernelGGL(fm_to_cm_kernel, g2, dim3(256), 0, h->stream, (const float*)Y, Cout, 0, d_y, Cout, Tout, gout.S, gout.P, 1.0f, 0.0f);
        if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, "sync failed: %s", hipGetErrorString(hipGetLastError()));
    } while (0);
    if (X) hipFree(X);
    if (Y) hipFree(Y);
    if (XM) hipFree(XM);
    while (h->owned.size() > owned0) { hipFree(h->owned.back()); h->owned.pop_back(); }
    return rc;
}

int ev_op_groupnorm_mish(ev_handle* h, const float* d_x, const float* d_gamma, const float* d_beta, const int32_t* d_lengths,
                         int B, int C, int T, int groups, float* d_y, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    if (groups != 8 || C != 256) return fail(h, "groupnorm op: C=256, groups=8 only");
    Geom g{B * (T + 4), T + 4, 2, T};
    float *X = nullptr, *Y = nullptr, *rm = nullptr;
    const size_t n = (size_t)g.nrows * C;
    HIPCHK(h, hipMalloc((void**)&X, n * 4)); HIPCHK(h, hipMalloc((void**)&Y, n * 4)); HIPCHK(h, hipMalloc((void**)&rm, g.nrows * 4));
    hipMemsetAsync(X, 0, n * 4, h->stream); hipMemsetAsync(Y, 0, n * 4, h->stream);
    hipLaunchKernelGGL(rowmask_kernel, dim3((g.nrows + 255) / 256), dim3(256), 0, h->stream, rm, d_lengths, g.nrows, g.S, g.P, g.T, 1);
    dim3 g1((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(cm_to_fm_kernel, g1, dim3(256), 0, h->stream, d_x, X, C, 0, C, T, g.S, g.P, (const float*)nullptr, 1.0f);
    int rc = launch_gn(h, X, C, Y, C, d_gamma, d_beta, rm, nullptr, nullptr, 0, g, C, 0);
    hipLaunchKernelGGL(fm_to_cm_kernel, g1, dim3(256), 0, h->stream, (const float*)Y, C, 0, d_y, C, T, g.S, g.P, 1.0f, 0.0f);
    hipStreamSynchronize(h->stream);
    hipFree(X); hipFree(Y); hipFree(rm);
    return rc;
}

int ev_op_split_pieces(ev_handle* h, const float* d_x, int n, float* d_pieces, void* stream) {
    if (!h || !d_x || !d_pieces || n <= 0) return fail(h, "ev_op_split_pieces: bad arguments");
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(evx_split_check_kernel, dim3((n + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, d_x, d_pieces, n);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int ev_op_layernorm(ev_handle* h, const float* d_x, const float* d_gamma, const float* d_beta, int rows, int C, float* d_y, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    if (C != 256) return fail(h, "layernorm op: C=256 only");
    Geom g{rows, rows, 0, rows};
    return launch_ln(h, d_x, C, d_y, C, d_gamma, d_beta, g);
}

// LayerNorm(256) -> Linear(256 -> M1) [-> SnakeBeta -> Linear(M1 -> 256) + residual x, * mask]: ln_mlp_kernel on (rows, 256) rows
int ev_op_ln_mlp(ev_handle* h, const float* d_x, const float* d_ln_g, const float* d_ln_b, const float* w1, const float* b1,
                 const float* d_alpha_exp, const float* d_beta_inv, const float* w2, const float* b2, const float* d_rowmask,
                 int rows, int M1, int mode, float* d_y, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    if (rows <= 0 || (M1 % 128) || !d_x || !w1 || !d_y || (mode == 0 && (!w2 || !b2 || !b1 || !d_alpha_exp || !d_beta_inv))) return fail(h, "ev_op_ln_mlp: bad arguments");
    size_t owned0 = h->owned.size();
    ConvLayer L1, L2;
    HostTensor t1, tb1, t2, tb2;
    t1.p = w1; t1.ndim = 2; t1.shape[0] = M1; t1.shape[1] = 256;
    tb1.p = b1; tb1.ndim = 1; tb1.shape[0] = M1;
    int rc = pack_linear_stack(h, L1, {&t1}, b1 ? std::vector<const HostTensor*>{&tb1} : std::vector<const HostTensor*>{});
    if (!rc && mode == 0) {
        t2.p = w2; t2.ndim = 2; t2.shape[0] = 256; t2.shape[1] = M1;
        tb2.p = b2; tb2.ndim = 1; tb2.shape[0] = 256;
        rc = pack_linear_stack(h, L2, {&t2}, {&tb2});
    }
    Geom g{rows, rows, 0, rows};
    if (!rc) rc = launch_mlp(h, mode, d_x, d_ln_g, d_ln_b, L1, mode == 0 ? &L2 : nullptr, d_alpha_exp, d_beta_inv, mode == 0 ? d_x : nullptr,
                             mode == 0 ? d_rowmask : nullptr, d_y, mode == 0 ? 256 : M1, g);
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, "sync failed");
    while (h->owned.size() > owned0) {
        if (h->owned.back() == (void*)h->zeros) break;      // (the handle's zero buffer may have been created by this call: keep it)
        hipFree(h->owned.back()); h->owned.pop_back();
    }
    return rc;
}

int ev_op_attention(ev_handle* h, const float* d_qkv, const int32_t* d_lengths, int B, int T, int heads, float* d_out, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    Geom g{B * T, T, 0, T};
    float* rm = nullptr;
    HIPCHK(h, hipMalloc((void**)&rm, (size_t)g.nrows * 4));
    hipLaunchKernelGGL(rowmask_kernel, dim3((g.nrows + 255) / 256), dim3(256), 0, h->stream, rm, d_lengths, g.nrows, g.S, g.P, g.T, 1);
    int rc = launch_attn(h, d_qkv, 3 * heads * 64, d_out, heads * 64, rm, g, heads);
    hipStreamSynchronize(h->stream);
    hipFree(rm);
    return rc;
}

// attn_out_kernel: d_hid (B*T, 256) <- d_hid + Wout . attention(d_qkv) + bout; w_out (256, 128), b_out (256) are HOST pointers
int ev_op_attn_out(ev_handle* h, const float* d_qkv, const int32_t* d_lengths, int B, int T, const float* w_out, const float* b_out, float* d_hid, void* stream) {
    if (!h) return 1;
    HIPCHK(h, hipSetDevice(h->device));
    h->stream = (hipStream_t)stream;
    if (B <= 0 || T <= 0 || !d_qkv || !w_out || !b_out || !d_hid) return fail(h, "ev_op_attn_out: bad arguments");
    size_t owned0 = h->owned.size();
    ConvLayer Lo;
    HostTensor tw, tb;
    tw.p = w_out; tw.ndim = 2; tw.shape[0] = 256; tw.shape[1] = 128;
    tb.p = b_out; tb.ndim = 1; tb.shape[0] = 256;
    int rc = pack_linear_stack(h, Lo, {&tw}, {&tb});
    Geom g{B * T, T, 0, T};
    float* rm = nullptr;
    if (!rc && hipMalloc((void**)&rm, (size_t)g.nrows * 4) != hipSuccess) rc = fail(h, "ev_op_attn_out: out of memory");
    float* packed = nullptr;
    float sc[3] = {0.f, 0.f, 0.f};
    if (!rc && h->split_terms == 16 && h->attn_h16 && Lo.Wh) {
        // arithmetic setting 16: the op runs what the model runs — q / k / v as fp16 piece pairs (the model's ln_qkv_h16_kernel writes them so;
        // here a copy is packed, with scales from the data's own maxima where the loader uses its weight bound) and attn_out_h16_kernel
        std::vector<float> hq((size_t)g.nrows * 384);
        if (hipMemcpy(hq.data(), d_qkv, hq.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, "ev_op_attn_out: copy failed");
        float mx[3] = {0.f, 0.f, 0.f};
        for (size_t i = 0; i < hq.size(); ++i) { const float a = std::fabs(hq[i]); if (std::isfinite(a)) { float& m = mx[(i % 384) / 128]; m = std::max(m, a); } }
        for (int i = 0; i < 3; ++i) sc[i] = mx[i] > 0.f ? (float)std::ldexp(1.0, std::min(40, std::max(-40, (int)std::floor(std::log2(32768.0 / mx[i]))))) : 1.f;
        float ma, mb;
        const bool fits = attn_mask_split(sc, &ma, &mb);                   // (else: the fp32 form, as the model would)
        if (!rc && fits && hipMalloc((void**)&packed, hq.size() * 4) != hipSuccess) rc = fail(h, "ev_op_attn_out: out of memory");
        if (!rc && fits) {
            const size_t npairs = hq.size() / 2;
            hipLaunchKernelGGL(qkv_pack_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, h->stream, d_qkv, packed, npairs, sc[0], sc[1], sc[2]);
        }
    }
    if (!rc) {
        hipLaunchKernelGGL(rowmask_kernel, dim3((g.nrows + 255) / 256), dim3(256), 0, h->stream, rm, d_lengths, g.nrows, g.S, g.P, g.T, 1);
        rc = launch_attn_out(h, packed ? packed : d_qkv, 384, Lo, d_hid, 256, rm, g, 2, packed ? sc : nullptr);
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, "sync failed");
    if (rm) hipFree(rm);
    if (packed) hipFree(packed);
    while (h->owned.size() > owned0) {
        if (h->owned.back() == (void*)h->zeros) break;
        hipFree(h->owned.back()); h->owned.pop_back();
    }
    return rc;
}

}  // extern "C"
