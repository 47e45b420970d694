/*
 * emojivoice.h — C ABI of the MI355X-native EmojiVoice TTS hot path.
 *
 * The reference (rosielab/emojivoice, vendored Matcha-TTS) is pure Python on
 * stock torch ops; it has no FFI of its own.  The entry points below are what a
 * binding for its hot path replaces (paths relative to Matcha-TTS/matcha/):
 *
 *   ev_cfm_decode   <-  BASECFM.forward / solve_euler      models/components/flow_matching.py:32-85
 *                       driving Decoder.forward            models/components/decoder.py:363-443
 *                       (BasicTransformerBlock             models/components/transformer.py:243-316,
 *                        diffusers Attention, SnakeBeta    transformer.py:17-80)
 *   ev_hifigan      <-  Generator.forward                  hifigan/models.py:181-197
 *                       (ResBlock1.forward                 hifigan/models.py:90-97)
 *   ev_text_encoder <-  TextEncoder.forward                models/components/text_encoder.py:378-410
 *                       (ConvReluNorm :36-67, Encoder :276-325, MultiHeadAttention + RoPE :97-246, FFN :255-273,
 *                        DurationPredictor :70-94, channel LayerNorm :15-33) — the caller of the hot path
 *   ev_load_estimator <- MatchaTTS.load_from_checkpoint -> state_dict["decoder.estimator.*"]   cli.py:110-118
 *   ev_load_text_encoder <- same checkpoint, state_dict["encoder.*"] + the derived "rope_theta" table
 *   ev_load_vocoder   <- Generator.load_state_dict(ckpt["generator"]) + remove_weight_norm()   cli.py:84-90
 *
 * Conventions
 *   - All tensors are fp32.  Pointers named d_* are DEVICE pointers owned by the
 *     caller (PyTorch-ROCm in the shipped host layer); they are borrowed for the
 *     duration of the call and all work is enqueued asynchronously on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream).
 *   - The handle owns the re-laid-out weights and a workspace that grows on demand, geometrically (growth = a wait for the
 *     handle's streams + hipFree + hipMalloc).  ev_reserve sizes everything once, up front, so that no later call at or below
 *     the reserved shape allocates or waits (a streaming server reserves its longest utterance: ev_alloc_count stays put).
 *   - Graph capture: ev_cfm_decode is capturable at any (B, Tp) the workspace already holds (ev_reserve, or an earlier eager call at the
 *     largest shape: planning must not allocate under capture): it enqueues kernels only and never waits on the host.  A handle may hold captured calls of MANY shapes (ABI 4; a serving loop keeps one graph per utterance length)
 *     and serve eager calls of any shape in between: once a call has been captured, every call on the handle — captured or eager —
 *     begins by re-zeroing the estimator's part of the workspace for its own plan, so none depends on what another left there.  A
 *     captured call consists of kernel nodes only (its re-zeroing is a kernel too): the time-MLP output for its step count must already be on the device, i.e.
 *     one EAGER ev_cfm_decode with the same n_steps must have run on the handle before (the handle keeps that output per step count;
 *     a host-to-device copy captured from pinned memory, and a captured memset, are not safe to replay on this runtime: later eager copies /
 *     memsets recycle their staging — profiles/r04_graph_capture_h2d_hazard.txt).
 *     What returns an error instead of pulling memory from under the graphs: a growth of the workspace beyond what is reserved, more
 *     Euler steps than the workspace is planned for (64, or the largest n_steps of an earlier call), a captured call with a step
 *     count no eager call has used.
 *   - Layout at the boundary is the reference's: mel-like tensors are (B, 80, T)
 *     channel-major contiguous; waveforms are (B, 256*T) contiguous.
 *   - Every function returns 0 on success, non-zero on failure; the message is
 *     available from ev_last_error().  Nothing throws across this boundary.
 *   - One handle per (device, stream user); calls on one handle are not thread-safe.  Different handles are independent:
 *     the library keeps no mutable process-global launch state, so two host threads may drive two handles concurrently.
 */
#ifndef EMOJIVOICE_H
#define EMOJIVOICE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EV_ABI_VERSION 4   /* 4: ev_dbg_set_amax, ev_dbg_set_attn_h16, ev_dbg_set_chain, ev_dbg_sk_taken, captured decodes of many shapes; 3: ev_set_arithmetic / ev_get_arithmetic, ev_profile_read_split, test hooks; everything of earlier versions unchanged */

typedef struct ev_handle ev_handle;

/* One tensor of a weight blob: `name` is the reference state_dict key with the
 * module prefix stripped ("down_blocks.0.0.block1.block.0.weight", "conv_pre.weight" ...),
 * `offset` is in floats from the start of the HOST blob. */
typedef struct ev_tensor_index {
    const char *name;
    uint64_t offset;
    int32_t ndim;
    int64_t shape[4];
} ev_tensor_index;

/* Static model dimensions (reference configs/model/decoder/default.yaml, hifigan/config.py:1-28). */
typedef struct ev_model_dims {
    int32_t n_feats;       /* 80 */
    int32_t spk_emb_dim;   /* 64 (0 for a single-speaker checkpoint) */
    int32_t channels;      /* 256 */
    int32_t heads;         /* 2 */
    int32_t head_dim;      /* 64 */
} ev_model_dims;

int ev_abi_version(void);
int ev_create(ev_handle **out, int device, const ev_model_dims *dims);
void ev_destroy(ev_handle *h);
const char *ev_last_error(ev_handle *h);

/* Weights: host blob + index; copied and re-laid-out into library-owned device memory. */
int ev_load_estimator(ev_handle *h, const float *blob, const ev_tensor_index *index, size_t n);
int ev_load_vocoder(ev_handle *h, const float *blob, const ev_tensor_index *index, size_t n);
/* Text-encoder keys: state_dict["encoder.<key>"] (matcha_tts.py:52-60) plus "rope_theta" =
 * 1 / (10000 ** (arange(0, d, 2) / d)), d = 64  (text_encoder.py:115-117), computed by the loader with the reference's ops. */
int ev_load_text_encoder(ev_handle *h, const float *blob, const ev_tensor_index *index, size_t n);

/* Bytes of device workspace the two hot calls need at a given shape (0 on bad args). */
size_t ev_workspace_bytes(ev_handle *h, int B, int Tp_cfm, int T_voc);

/* Conditional-flow-matching ODE decode: n_steps Euler steps of the U-Net estimator.
 *   d_mu      (B, 80, Tp)  aligned encoder output mu_y          (matcha_tts.py:134-135)
 *   d_lengths (B) int32    valid frames per utterance (y_lengths; mask = t < length)
 *   d_spk     (B, 64)      speaker/emoji embedding rows (NULL iff spk_emb_dim == 0)
 *   d_z       (B, 80, Tp)  initial state x0 = noise * temperature (flow_matching.py:51)
 *   d_out     (B, 80, Tp)  decoder output; out_scale/out_shift fuse denormalize()
 *                          (utils/model.py:71-90): out = dec*out_scale + out_shift
 *   Tp must be a multiple of 4 (fix_len_compatibility, utils/model.py:14-20). */
int ev_cfm_decode(ev_handle *h, const float *d_mu, const int32_t *d_lengths, const float *d_spk, const float *d_z,
                  int B, int Tp, int n_steps, float out_scale, float out_shift, float *d_out, void *stream);

/* The same decode with both of the reference's outputs (matcha_tts.py:139-152): d_dec = "decoder_outputs" (may be NULL),
 * d_mel = "mel" = denormalize(decoder_outputs, mel_mean, mel_std) = dec * mel_std + mel_mean (may be NULL; not both). */
int ev_cfm_decode2(ev_handle *h, const float *d_mu, const int32_t *d_lengths, const float *d_spk, const float *d_z,
                   int B, int Tp, int n_steps, float *d_dec, float mel_std, float mel_mean, float *d_mel, void *stream);

/* Pre-size everything the hot calls allocate on demand, for batches of B utterances of up to Tx_max tokens (ev_text_encoder),
 * Tp_max mel frames (ev_cfm_decode / ev_estimator; multiple of 4) and T_voc_max mel frames (ev_hifigan, ev_denoise at
 * L = 256 * T_voc_max): the workspace arena, the text-encoder and denoiser scratch, the pinned time-embedding ring, the
 * denoiser's DFT bases and the side streams of the small-call vocoder.  0 skips a stage.  After it, calls with the same B and
 * lengths up to the reserved ones never allocate (ev_alloc_count does not move) and never wait for the device to re-plan.
 * The reference has no counterpart: torch's caching allocator amortises the same cost (feel_me.py:181-203). */
int ev_reserve(ev_handle *h, int B, int Tx_max, int Tp_max, int T_voc_max, void *stream);
/* Device / pinned allocations made by the hot calls since ev_create (workspace growth, scratch growth, staging). */
int64_t ev_alloc_count(ev_handle *h);

/* One estimator evaluation v = Decoder(x, mask, mu, t, spk)  (decoder.py:363-443). */
int ev_estimator(ev_handle *h, const float *d_x, const float *d_mu, const int32_t *d_lengths, const float *d_spk,
                 float t, int B, int Tp, float *d_v, void *stream);

/* Text encoder + duration predictor (the stage in front of ev_cfm_decode; matcha_tts.py:118-121):
 *   d_ids     (B, Tx) int64  phoneme ids (torch.long, as the reference passes them)
 *   d_lengths (B) int32      valid tokens per utterance (x_lengths; mask = t < length)
 *   d_spk     (B, 64)        speaker/emoji embedding rows (NULL iff the model is single-speaker)
 *   d_mu      (B, 80, Tx)    mu_x, masked          d_logw (B, Tx)  log-durations, masked
 * Duration rounding, the monotonic alignment path and mu_y = attn^T mu_x stay with the caller (data-dependent sizes). */
int ev_text_encoder(ev_handle *h, const int64_t *d_ids, const int32_t *d_lengths, const float *d_spk, int B, int Tx,
                    float *d_mu, float *d_logw, void *stream);

/* The reference's nn.Embedding raises IndexError for a token id outside [0, n_vocab) (text_encoder.py:395).  The device
 * stage cannot raise mid-stream: it computes from a clamped id and records the event.  This call waits for `stream` and
 * returns non-zero ("index out of range in self") if any ev_text_encoder call since the last check saw such an id among
 * the valid tokens; the host layer calls it at its first natural synchronisation point (the read of max(y_lengths)). */
int ev_text_encoder_status(ev_handle *h, void *stream);

/* Hard monotonic alignment and expansion (utils/model.py:29-41 generate_path; matcha_tts.py:131-135):
 *   d_wceil (B, Tx) f32   ceil(exp(logw) * x_mask) * length_scale        d_mu_x (B, 80, Tx)
 *   d_xlen (B) int32, d_ylen (B) int64 (= clamp_min(sum(w_ceil), 1).long(), computed by the caller: its maximum fixes Tp)
 *   d_mu_y (B, 80, Tp) = attn^T mu_x;   d_attn (B, Tx, Tp) 0/1 path, may be NULL */
int ev_align(ev_handle *h, const float *d_wceil, const float *d_mu_x, const int32_t *d_xlen, const int64_t *d_ylen,
             int B, int Tx, int Tp, float *d_mu_y, float *d_attn, void *stream);

/* Largest ev_hifigan call (B*T mel frames) that fans its ResBlock1 chains out over three streams (see ev_hifigan); 0 = never.
 * Default 16384, or EV_MRF_STREAMS_MAX.  A caller that already overlaps the vocoder with other work on a second stream, and
 * must not have the host wait inside the call, sets 0 (emojivoice_amd/pipeline.py does). */
int ev_set_mrf_streams_max(ev_handle *h, int max_frames);

/* HiFi-GAN V1 generator: d_mel (B, 80, T) -> d_wav (B, 256*T), tanh output, no clamp/denoiser.
 * Calls of B*T <= 16384 mel frames (EV_MRF_STREAMS_MAX) first wait for `stream` to drain on the host, then run the three
 * ResBlock1 chains of each level on `stream` and two streams of the handle, joined back into `stream` by events before the
 * call returns: the result is ordered on `stream` like that of any other call.  Such a call is not graph-capturable. */
int ev_hifigan(ev_handle *h, const float *d_mel, int B, int T, float *d_wav, void *stream);

/* Denoiser stage that every reference caller applies to the vocoder output (hifigan/denoiser.py:10-64; cli.py:121-126,
 * feel_me.py:181-187): torch.stft / torch.istft semantics at n_fft = win = 1024, hop 256, periodic Hann window, centred
 * with reflect padding, evaluated as two DFT-basis convolutions on the matrix cores.
 *   ev_stft_magnitude: d_audio (B, L) -> d_mag (B, 513, L/256 + 1)     (Denoiser.__init__'s bias spectrum)
 *   ev_denoise:        d_out = ISTFT(clamp(|X| - bias*strength, 0) * exp(i*angle(X))), X = STFT(d_audio); d_bias_spec (513)
 * L must be a multiple of 256 (it is 256 * mel frames) and >= 768: like torch.stft's reflect padding (512 each side), which
 * the reference relies on, inputs of 512 samples or fewer are an error. */
int ev_stft_magnitude(ev_handle *h, const float *d_audio, int B, int L, float *d_mag, void *stream);
int ev_denoise(ev_handle *h, const float *d_audio, int B, int L, const float *d_bias_spec, float strength, float *d_out, void *stream);

/* Timing hooks for bench.py: HIP-event time (ms) of the dominant kernel family
 * (implicit-GEMM convs, fused pairs, fused LayerNorm + MLP, fused attention) accumulated over the calls since the last reset,
 * measured on the stream the kernels run on. */
int ev_profile_enable(ev_handle *h, int on);
int ev_profile_read(ev_handle *h, double *conv_ms, double *conv_flops, int64_t *conv_launches, int reset);
/* ... and, of those, the launches that ran on the bf16 matrix pipe (conv_split_kernel, conv_split_bal_kernel,
 * resblock_pair_split_kernel: every fp32 product as six exact bf16 products, fp32 accumulation); call before a resetting
 * ev_profile_read.  The rest of the family runs on v_mfma_f32_32x32x2_f32. */
int ev_profile_read_split(ev_handle *h, double *ms, double *flops, int64_t *launches);
/* Arithmetic of the contractions.  Tensors are fp32 and every accumulation is fp32 in every setting.
 *   16 (default): layers deep enough to pay for it form each fp32 product from THREE fp16 x fp16 products on the fp16 matrix pipe: an
 *                 operand, times a power-of-two block scale (weights: one per layer; activations: one per workgroup tile, from the
 *                 tile's maximum), is cut into two fp16 pieces h0 + h1 (22-23 significand bits), the product is h0 g0 + h0 g1 + h1 g0.
 *                 Errors against fp64 at the level of the fp32 FMA chain (one conv layer, tools/arith_accuracy.py: rms 3.4e-7 of the
 *                 output scale against 5.4e-7 for the fp32 MFMA), independent of the activations' scale; ~2.1x the fp32 MFMA's throughput
 *   6:            three bf16 pieces per operand (exact split, no range handling), the six products of weight <= 2: fp32-grade too, ~1.45x
 *   0:            every layer on the fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain)
 *   3:            opt-in fast bf16 setting of the vocoder's deep layers: three products of weight <= 1, ~16 significand bits per product
 *                 (waveform RMS difference to the fp32 MFMA result 7e-5 against 1.7e-6 for 16 and 6): inside the 1e-3 gate, NOT fp32-grade
 *   9:            accuracy A/B of conv_split_kernel only (tools/bf16_split_probe.hip); the other split builds run 6
 * The environment variable EV_SPLIT presets it for handles created afterwards.  Takes effect with the next call on the handle. */
int ev_set_arithmetic(ev_handle *h, int setting);
int ev_get_arithmetic(ev_handle *h);

/* Test hook: the build the last conv / fused-pair launch of this handle took (tile configuration id: 40 / 60 = conv_split_kernel /
 * its balanced grid, 140 + taps = resblock_pair_split_kernel, 100 + taps = resblock_pair_kernel, others: see launch_conv). */
int ev_dbg_last_cfg(ev_handle *h);

/* Kernel microbenchmark hook (tools/conv_bench.py, not part of the product path): times `iters` launches of one
 * resblock-style conv (prologue leaky-relu, bias, residual) at a given geometry with HIP events on the default stream;
 * dbg = ablation bits, cfg = forced tile configuration (< 0: the engine's own choice). */
int ev_dbg_conv_bench(ev_handle *h, int Cin, int Cout, int K, int dil, int B, int T, int P, int iters, int dbg, int cfg, float *ms_out);

/* Diagnostic / A-B switch: the fp16 builds need max |x| over the rows a tile stages.  on = 1 (default): they take it from the bounds their
 * producers left per 128-row granule (every launch of ev_hifigan's chain leaves an upper bound of |y| from its accumulators — no second read
 * of the input); on = 0: every tile pre-scans its input (the behaviour before ABI 4; also what inputs without bounds get).  The environment
 * variable EV_NO_AMAX=1 presets 0 for handles created afterwards.  Results differ only through the choice of the power-of-two block scale. */
int ev_dbg_set_amax(ev_handle *h, int on);
/* Diagnostic / A-B switch (ABI 4): on = 1 (default): under arithmetic setting 16 the self-attention of the U-Net's transformer blocks (transformer.py:262-271)
 * runs on the fp16 matrix pipe — q, k, v leave the LayerNorm + projection kernel as fp16 piece pairs times a power of two that the loader derives from a
 * bound no input can exceed, and both products of the attention use all piece products (22-bit operands, fp32 accumulation); on = 0: the attention
 * stays on the fp32 MFMA as before ABI 4.  EV_NO_ATTN_H16=1 presets 0 for handles created afterwards.  Arithmetic settings 6 / 0 never take this path. */
int ev_dbg_set_attn_h16(ev_handle *h, int on);
/* Diagnostic / A-B switch (ABI 4): on = 1 (default): under arithmetic setting 16, ev_hifigan runs a whole ResBlock1 (hifigan/models.py:90-97: three
 * (dilated conv, conv) pairs with their residual adds) as ONE launch where the level is narrow (32 / 64 channels) and the kernel size small enough
 * for the summed halos (k = 3): the running x stays in registers between the pairs; on = 0: three fused-pair launches as before.  EV_NO_CHAIN=1
 * presets 0.  Results differ by rounding only (other tile boundaries, hence other power-of-two tile scales). */
int ev_dbg_set_chain(ev_handle *h, int on);

/* Diagnostic: the control words of the balanced ("stream-K") launches (ev_kernels.h, SkCtl) after a device synchronisation:
 * out3 = {launches so far (epoch), arrivals of an unfinished launch (0), hand-off waits that ran out and were recomputed}. */
int ev_dbg_sk_stats(ev_handle *h, uint32_t *out3);
/* Diagnostic (ABI 4): contributor shares the owners of balanced launches took over because the contributor had not started yet (an owner no
 * longer waits for a workgroup that is not resident — with two pipelines in flight the vocoder holds CU slots — it computes the share itself,
 * with the bits the contributor would have delivered), since the handle was created; -1 on error.  Synchronises the device. */
int64_t ev_dbg_sk_taken(ev_handle *h);

/* ---- operator-level entry points (unit parity tests call these) ------------------
 * Activations here are frame-major (rows, C) fp32 with an explicit row stride. */
int ev_op_conv1d(ev_handle *h, const float *d_x /*(B,Cin,T)*/, const float *w /*HOST (Cout,Cin,K)*/,
                 const float *bias /*HOST (Cout) or NULL*/, int B, int Cin, int T, int Cout, int K, int dilation,
                 int transposed, int stride, int padding, float pre_lrelu_slope /*<0: none*/, float *d_y, void *stream);
int ev_op_groupnorm_mish(ev_handle *h, const float *d_x /*(B,C,T)*/, const float *d_gamma, const float *d_beta,
                         const int32_t *d_lengths, int B, int C, int T, int groups, float *d_y, void *stream);
/* The three bf16 pieces (as fp32 values, (3, n)) the split builds cut every fp32 operand into: p0 + p1 + p2 == x exactly. */
int ev_op_split_pieces(ev_handle *h, const float *d_x, int n, float *d_pieces, void *stream);
int ev_op_layernorm(ev_handle *h, const float *d_x /*(rows,C)*/, const float *d_gamma, const float *d_beta, int rows,
                    int C, float *d_y, void *stream);
/* ln_mlp_kernel: y = x + W2.SnakeBeta(W1.LN(x) + b1) + b2, rows * mask (mode 0; transformer.py:300-316) or y = W1.LN(x) [+ b1]
 * (mode 1: the QKV projection, y is (rows, M1)); x (rows, 256); alpha_exp = exp(alpha), beta_inv = 1/(exp(beta)+1e-9) (M1);
 * w1 (M1, 256) and w2 (256, M1) are HOST pointers; M1 a multiple of 128; rowmask (rows) or NULL. */
int ev_op_ln_mlp(ev_handle *h, const float *d_x, const float *d_ln_g, const float *d_ln_b, const float *w1, const float *b1,
                 const float *d_alpha_exp, const float *d_beta_inv, const float *w2, const float *b2, const float *d_rowmask,
                 int rows, int M1, int mode, float *d_y, void *stream);
int ev_op_attention(ev_handle *h, const float *d_qkv /*(B,T,3*heads*64)*/, const int32_t *d_lengths, int B, int T,
                    int heads, float *d_out /*(B,T,heads*64)*/, void *stream);

/* attn_out_kernel: d_hid (B*T, 256) <- d_hid + Wout . Attention(d_qkv) + bout, both heads, additive float mask (transformer.py:262-271);
 * d_qkv (B, T, 384) = [q | k | v] x (2 heads x 64); w_out (256, 128) and b_out (256) are HOST pointers. */
int ev_op_attn_out(ev_handle *h, const float *d_qkv, const int32_t *d_lengths, int B, int T, const float *w_out, const float *b_out,
                   float *d_hid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EMOJIVOICE_H */
