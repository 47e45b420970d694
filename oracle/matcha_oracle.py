"""CPU oracle for the EmojiVoice hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the shipped path (``emojivoice_amd``) never does and
fails loudly when its HIP library is missing.

This is a restatement in plain ``torch`` (fp32, CPU, functional — no
lightning / hydra / diffusers / einops / conformer) of the reference algorithm.
Every function cites the reference lines it follows (paths relative to
``/root/reference/Matcha-TTS/matcha`` unless they start with ``feel_me.py``).
Weights come in as a flat ``{reference_state_dict_key: tensor}`` mapping.

Pinning (see tests/golden/make_golden.py, tests/test_oracle_golden.py):
  * HiFi-GAN (``hifigan_forward``, ``denoiser``): pinned against the reference's
    own ``matcha.hifigan`` modules imported unmodified in the build container.
  * Matcha (``text_encoder``, ``synthesise``, ``cfm_decode``, ``estimator``):
    pinned against the reference's own ``matcha_tts.py / flow_matching.py /
    decoder.py / transformer.py / text_encoder.py / utils/model.py`` imported
    unmodified, EXCEPT the arithmetic inside ``diffusers.models.
    attention_processor.Attention`` (third-party, diffusers==0.25.0 per
    Matcha-TTS/requirements.txt:40, absent offline).  That one class is restated
    from its published algorithm (``attention`` below) -> **parity unpinned at
    the attention boundary**; the reference-side check available for it is the
    parameter count 18,204,193 (synthesis.ipynb:127), which the restated
    structure reproduces.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ---------------------------------------------------------------------------
# utils/model.py
# ---------------------------------------------------------------------------
def sequence_mask(length: Tensor, max_length: Optional[int] = None) -> Tensor:
    """utils/model.py:7-11."""
    if max_length is None:
        max_length = int(length.max())
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def fix_len_compatibility(length: int, num_downsamplings_in_unet: int = 2) -> int:
    """utils/model.py:14-20 — round up to a multiple of 2**n."""
    factor = 2**num_downsamplings_in_unet
    return int(math.ceil(float(length) / factor) * factor)


def generate_path(duration: Tensor, mask: Tensor) -> Tensor:
    """utils/model.py:29-41 — hard monotonic alignment from cumulative durations."""
    b, t_x, t_y = mask.shape
    cum_duration = torch.cumsum(duration, 1)
    cum_flat = cum_duration.view(b * t_x)
    path = sequence_mask(cum_flat, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path * mask


def denormalize(data: Tensor, mu, std) -> Tensor:
    """utils/model.py:71-90 with the scalar buffers of baselightningmodule.py:20-28."""
    return data * std + mu


# ---------------------------------------------------------------------------
# text_encoder.py
# ---------------------------------------------------------------------------
def _chan_layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-4) -> Tensor:
    """text_encoder.py:15-33 (channel-wise LayerNorm over dim 1)."""
    mean = torch.mean(x, 1, keepdim=True)
    variance = torch.mean((x - mean) ** 2, 1, keepdim=True)
    x = (x - mean) * torch.rsqrt(variance + eps)
    return x * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def _rope(x: Tensor, d: int, base: float = 10000.0) -> Tensor:
    """text_encoder.py:97-172 — rotary embedding on the first ``d`` features of (b,h,t,c)."""
    t = x.shape[2]
    theta = 1.0 / (base ** (torch.arange(0, d, 2).float() / d))
    seq_idx = torch.arange(t).float()
    idx_theta = torch.einsum("n,d->nd", seq_idx, theta)
    idx_theta2 = torch.cat([idx_theta, idx_theta], dim=1)
    cos = idx_theta2.cos()[None, None, :, :]
    sin = idx_theta2.sin()[None, None, :, :]
    x_rope, x_pass = x[..., :d], x[..., d:]
    d_2 = d // 2
    neg_half = torch.cat([-x_rope[..., d_2:], x_rope[..., :d_2]], dim=-1)
    x_rope = (x_rope * cos) + (neg_half * sin)
    return torch.cat((x_rope, x_pass), dim=-1)


def _enc_mha(sd: SD, p: str, x: Tensor, attn_mask: Tensor, n_heads: int) -> Tensor:
    """text_encoder.py:175-246 (MultiHeadAttention with RoPE on half the head dim)."""
    q = F.conv1d(x, sd[f"{p}.conv_q.weight"], sd[f"{p}.conv_q.bias"])
    k = F.conv1d(x, sd[f"{p}.conv_k.weight"], sd[f"{p}.conv_k.bias"])
    v = F.conv1d(x, sd[f"{p}.conv_v.weight"], sd[f"{p}.conv_v.bias"])
    b, d, t = k.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    q = _rope(q, int(kc * 0.5))
    k = _rope(k, int(kc * 0.5))
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(kc)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(out, sd[f"{p}.conv_o.weight"], sd[f"{p}.conv_o.bias"])


def text_encoder(sd: SD, x: Tensor, x_lengths: Tensor, spks: Optional[Tensor], n_heads: int = 2, n_layers: int = 6, prefix: str = "encoder"):
    """TextEncoder.forward, text_encoder.py:378-410.  Returns (mu, logw, x_mask)."""
    p = prefix
    n_channels = sd[f"{p}.emb.weight"].shape[1]
    h = F.embedding(x, sd[f"{p}.emb.weight"]) * math.sqrt(n_channels)
    h = torch.transpose(h, 1, -1)
    x_mask = torch.unsqueeze(sequence_mask(x_lengths, h.size(2)), 1).to(h.dtype)
    # prenet ConvReluNorm :36-67 (dropout = identity in eval)
    h_org = h
    for i in range(3):
        w = sd[f"{p}.prenet.conv_layers.{i}.weight"]
        h = F.conv1d(h * x_mask, w, sd[f"{p}.prenet.conv_layers.{i}.bias"], padding=w.shape[2] // 2)
        h = _chan_layer_norm(h, sd[f"{p}.prenet.norm_layers.{i}.gamma"], sd[f"{p}.prenet.norm_layers.{i}.beta"])
        h = torch.relu(h)
    h = h_org + F.conv1d(h, sd[f"{p}.prenet.proj.weight"], sd[f"{p}.prenet.proj.bias"])
    h = h * x_mask
    if spks is not None:
        h = torch.cat([h, spks.unsqueeze(-1).repeat(1, 1, h.shape[-1])], dim=1)
    # Encoder :276-325
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    e = f"{p}.encoder"
    for i in range(n_layers):
        h = h * x_mask
        y = _enc_mha(sd, f"{e}.attn_layers.{i}", h, attn_mask, n_heads)
        h = _chan_layer_norm(h + y, sd[f"{e}.norm_layers_1.{i}.gamma"], sd[f"{e}.norm_layers_1.{i}.beta"])
        w1 = sd[f"{e}.ffn_layers.{i}.conv_1.weight"]
        w2 = sd[f"{e}.ffn_layers.{i}.conv_2.weight"]
        y = F.conv1d(h * x_mask, w1, sd[f"{e}.ffn_layers.{i}.conv_1.bias"], padding=w1.shape[2] // 2)
        y = torch.relu(y)
        y = F.conv1d(y * x_mask, w2, sd[f"{e}.ffn_layers.{i}.conv_2.bias"], padding=w2.shape[2] // 2)
        y = y * x_mask
        h = _chan_layer_norm(h + y, sd[f"{e}.norm_layers_2.{i}.gamma"], sd[f"{e}.norm_layers_2.{i}.beta"])
    h = h * x_mask
    mu = F.conv1d(h, sd[f"{p}.proj_m.weight"], sd[f"{p}.proj_m.bias"]) * x_mask
    # DurationPredictor :70-94
    w = f"{p}.proj_w"
    d = F.conv1d(h * x_mask, sd[f"{w}.conv_1.weight"], sd[f"{w}.conv_1.bias"], padding=sd[f"{w}.conv_1.weight"].shape[2] // 2)
    d = torch.relu(d)
    d = _chan_layer_norm(d, sd[f"{w}.norm_1.gamma"], sd[f"{w}.norm_1.beta"])
    d = F.conv1d(d * x_mask, sd[f"{w}.conv_2.weight"], sd[f"{w}.conv_2.bias"], padding=sd[f"{w}.conv_2.weight"].shape[2] // 2)
    d = torch.relu(d)
    d = _chan_layer_norm(d, sd[f"{w}.norm_2.gamma"], sd[f"{w}.norm_2.beta"])
    logw = F.conv1d(d * x_mask, sd[f"{w}.proj.weight"], sd[f"{w}.proj.bias"]) * x_mask
    return mu, logw, x_mask


# ---------------------------------------------------------------------------
# decoder.py / transformer.py — the CFM estimator
# ---------------------------------------------------------------------------
def sinusoidal_pos_emb(t: Tensor, dim: int, scale: float = 1000) -> Tensor:
    """decoder.py:14-29."""
    if t.ndim < 1:
        t = t.unsqueeze(0)
    half_dim = dim // 2
    emb = math.log(10000) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim).float() * -emb)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def _block1d(sd: SD, p: str, x: Tensor, mask: Tensor, groups: int = 8) -> Tensor:
    """Block1D, decoder.py:32-43: Conv1d k3 p1 -> GroupNorm(8) -> Mish, masked in and out."""
    h = F.conv1d(x * mask, sd[f"{p}.block.0.weight"], sd[f"{p}.block.0.bias"], padding=1)
    h = F.group_norm(h, groups, sd[f"{p}.block.1.weight"], sd[f"{p}.block.1.bias"], eps=1e-5)
    h = F.mish(h)
    return h * mask


def _resnet(sd: SD, p: str, x: Tensor, mask: Tensor, t_emb: Tensor) -> Tensor:
    """ResnetBlock1D, decoder.py:46-61."""
    h = _block1d(sd, f"{p}.block1", x, mask)
    h = h + F.linear(F.mish(t_emb), sd[f"{p}.mlp.1.weight"], sd[f"{p}.mlp.1.bias"]).unsqueeze(-1)
    h = _block1d(sd, f"{p}.block2", h, mask)
    return h + F.conv1d(x * mask, sd[f"{p}.res_conv.weight"], sd[f"{p}.res_conv.bias"])


def attention(sd: SD, p: str, x: Tensor, mask: Tensor, heads: int = 2) -> Tensor:
    """diffusers==0.25.0 ``Attention`` + default ``AttnProcessor2_0`` as constructed at
    transformer.py:196-204 and called at :266-271 (self-attention, no bias on q/k/v,
    bias on to_out[0], scale = head_dim**-0.5).

    PARITY UNPINNED (third-party code absent offline; restated from the published
    algorithm).  The float ``(B,T)`` 0/1 frame mask is ``prepare_attention_mask``-ed to
    ``(B,heads,1,T)`` and handed to ``F.scaled_dot_product_attention`` as a FLOAT mask,
    i.e. it is ADDED to the scores (valid keys +1.0, padded keys +0.0): padded frames
    remain live keys/values.
    """
    b, t, _ = x.shape
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    k = F.linear(x, sd[f"{p}.to_k.weight"])
    v = F.linear(x, sd[f"{p}.to_v.weight"])
    hd = q.shape[-1] // heads
    q = q.view(b, t, heads, hd).transpose(1, 2)
    k = k.view(b, t, heads, hd).transpose(1, 2)
    v = v.view(b, t, heads, hd).transpose(1, 2)
    am = mask.repeat_interleave(heads, dim=0).view(b, heads, -1, mask.shape[-1])
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=am, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(b, t, heads * hd)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def _snake_beta(sd: SD, p: str, x: Tensor) -> Tensor:
    """SnakeBeta.forward, transformer.py:63-80 (alpha_logscale=True)."""
    x = F.linear(x, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])
    alpha = torch.exp(sd[f"{p}.alpha"])
    beta = torch.exp(sd[f"{p}.beta"])
    return x + (1.0 / (beta + 0.000000001)) * torch.pow(torch.sin(x * alpha), 2)


def _transformer(sd: SD, p: str, x: Tensor, mask: Tensor) -> Tensor:
    """BasicTransformerBlock.forward, transformer.py:243-316, reachable branch only
    (layer_norm, self-attention, no cross-attention, SnakeBeta FF, dropout = identity)."""
    dim = x.shape[-1]
    n = F.layer_norm(x, (dim,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps=1e-5)
    x = attention(sd, f"{p}.attn1", n, mask) + x
    n = F.layer_norm(x, (dim,), sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], eps=1e-5)
    ff = _snake_beta(sd, f"{p}.ff.net.0", n)
    ff = F.linear(ff, sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    return ff + x


def estimator(sd: SD, x: Tensor, mask: Tensor, mu: Tensor, t: Tensor, spks: Optional[Tensor], prefix: str = "decoder.estimator") -> Tensor:
    """Decoder.forward, decoder.py:363-443, for channels=(256,256), n_blocks=1,
    num_mid_blocks=2 (configs/model/decoder/default.yaml)."""
    p = prefix
    in_ch = sd[f"{p}.time_mlp.linear_1.weight"].shape[1]
    te = sinusoidal_pos_emb(t, in_ch)
    te = F.linear(te, sd[f"{p}.time_mlp.linear_1.weight"], sd[f"{p}.time_mlp.linear_1.bias"])
    te = F.silu(te)
    te = F.linear(te, sd[f"{p}.time_mlp.linear_2.weight"], sd[f"{p}.time_mlp.linear_2.bias"])

    x = torch.cat([x, mu], dim=1)
    if spks is not None:
        x = torch.cat([x, spks.unsqueeze(-1).expand(-1, -1, x.shape[-1])], dim=1)

    hiddens = []
    masks = [mask]
    for i in range(2):
        mask_down = masks[-1]
        x = _resnet(sd, f"{p}.down_blocks.{i}.0", x, mask_down, te)
        x = _transformer(sd, f"{p}.down_blocks.{i}.1.0", x.transpose(1, 2), mask_down[:, 0]).transpose(1, 2)
        hiddens.append(x)
        if i == 0:
            x = F.conv1d(x * mask_down, sd[f"{p}.down_blocks.0.2.conv.weight"], sd[f"{p}.down_blocks.0.2.conv.bias"], stride=2, padding=1)
        else:
            x = F.conv1d(x * mask_down, sd[f"{p}.down_blocks.1.2.weight"], sd[f"{p}.down_blocks.1.2.bias"], padding=1)
        masks.append(mask_down[:, :, ::2])
    masks = masks[:-1]
    mask_mid = masks[-1]
    for i in range(2):
        x = _resnet(sd, f"{p}.mid_blocks.{i}.0", x, mask_mid, te)
        x = _transformer(sd, f"{p}.mid_blocks.{i}.1.0", x.transpose(1, 2), mask_mid[:, 0]).transpose(1, 2)
    for i in range(2):
        mask_up = masks.pop()
        x = _resnet(sd, f"{p}.up_blocks.{i}.0", torch.cat([x, hiddens.pop()], dim=1), mask_up, te)
        x = _transformer(sd, f"{p}.up_blocks.{i}.1.0", x.transpose(1, 2), mask_up[:, 0]).transpose(1, 2)
        if i == 0:
            x = F.conv_transpose1d(x * mask_up, sd[f"{p}.up_blocks.0.2.conv.weight"], sd[f"{p}.up_blocks.0.2.conv.bias"], stride=2, padding=1)
        else:
            x = F.conv1d(x * mask_up, sd[f"{p}.up_blocks.1.2.weight"], sd[f"{p}.up_blocks.1.2.bias"], padding=1)
    x = _block1d(sd, f"{p}.final_block", x, mask_up)
    out = F.conv1d(x * mask_up, sd[f"{p}.final_proj.weight"], sd[f"{p}.final_proj.bias"])
    return out * mask


# ---------------------------------------------------------------------------
# flow_matching.py
# ---------------------------------------------------------------------------
def solve_euler(sd: SD, z: Tensor, mu: Tensor, mask: Tensor, n_timesteps: int, spks: Optional[Tensor], return_all: bool = False):
    """BASECFM.solve_euler, flow_matching.py:55-85, with t_span = linspace(0,1,n+1) (:52)."""
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    t, dt = t_span[0], t_span[1] - t_span[0]
    x = z
    sol = []
    for step in range(1, len(t_span)):
        dphi_dt = estimator(sd, x, mask, mu, t, spks)
        x = x + dt * dphi_dt
        t = t + dt
        sol.append(x)
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return sol if return_all else sol[-1]


def cfm_decode(sd: SD, mu: Tensor, mask: Tensor, n_timesteps: int, temperature: float = 1.0, spks: Optional[Tensor] = None, z: Optional[Tensor] = None) -> Tensor:
    """BASECFM.forward, flow_matching.py:32-53.  ``z`` (unit normal, before the
    temperature scale) may be supplied for bit-reproducible parity runs."""
    if z is None:
        z = torch.randn_like(mu)
    return solve_euler(sd, z * temperature, mu, mask, n_timesteps, spks)


# ---------------------------------------------------------------------------
# matcha_tts.py
# ---------------------------------------------------------------------------
def synthesise(sd: SD, x: Tensor, x_lengths: Tensor, n_timesteps: int, temperature: float = 1.0, spks: Optional[Tensor] = None, length_scale: float = 1.0, z: Optional[Tensor] = None) -> dict:
    """MatchaTTS.synthesise, matcha_tts.py:77-152 (rtf omitted: wall clock)."""
    n_spks = sd["spk_emb.weight"].shape[0] if "spk_emb.weight" in sd else 1
    spk = F.embedding(spks.long(), sd["spk_emb.weight"]) if n_spks > 1 else None
    mu_x, logw, x_mask = text_encoder(sd, x, x_lengths, spk)
    w = torch.exp(logw) * x_mask
    w_ceil = torch.ceil(w) * length_scale
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
    y_max_length = int(y_lengths.max())
    y_max_length_ = fix_len_compatibility(y_max_length)
    y_mask = sequence_mask(y_lengths, y_max_length_).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)
    attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
    mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
    encoder_outputs = mu_y[:, :, :y_max_length]
    dec = cfm_decode(sd, mu_y, y_mask, n_timesteps, temperature, spk, z=z)
    dec = dec[:, :, :y_max_length]
    return {
        "encoder_outputs": encoder_outputs,
        "decoder_outputs": dec,
        "attn": attn[:, :, :y_max_length],
        "mel": denormalize(dec, sd["mel_mean"], sd["mel_std"]),
        "mel_lengths": y_lengths,
        "mu_y": mu_y,
        "y_mask": y_mask,
    }


# ---------------------------------------------------------------------------
# hifigan/models.py
# ---------------------------------------------------------------------------
LRELU_SLOPE = 0.1  # hifigan/models.py:11


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """hifigan/xutils.py:37-38."""
    return int((kernel_size * dilation - dilation) / 2)


def _resblock1(sd: SD, p: str, x: Tensor, k: int, dil) -> Tensor:
    """ResBlock1.forward, hifigan/models.py:90-97."""
    for m, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"], dilation=d, padding=get_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"], padding=get_padding(k, 1))
        x = xt + x
    return x


def hifigan_forward(sd: SD, mel: Tensor, h: dict, return_stages: bool = False):
    """Generator.forward, hifigan/models.py:181-197, weight-norm folded."""
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    rk, rd = h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]
    nk = len(rk)
    stages = []
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    stages.append(x)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            r = _resblock1(sd, f"resblocks.{i * nk + j}", x, rk[j], rd[j])
            xs = r if xs is None else xs + r
        x = xs / nk
        stages.append(x)
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:193)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    x = torch.tanh(x)
    return (x, stages) if return_stages else x


# ---------------------------------------------------------------------------
# hifigan/denoiser.py  (next row f-1)
# ---------------------------------------------------------------------------
def denoiser_bias_spec(sd: SD, h: dict, filter_length: int = 1024, n_overlap: int = 4, win_length: int = 1024) -> Tensor:
    """Denoiser.__init__, denoiser.py:10-56 (mode="zeros")."""
    mel_input = torch.zeros((1, 80, 88))
    bias_audio = hifigan_forward(sd, mel_input, h).float().squeeze(0)
    spec = torch.stft(bias_audio, n_fft=filter_length, hop_length=filter_length // n_overlap, win_length=win_length,
                      window=torch.hann_window(win_length), return_complex=True)
    mag = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1))
    return mag[:, :, 0][:, :, None]


def denoiser(audio: Tensor, bias_spec: Tensor, strength: float = 0.0005, filter_length: int = 1024, n_overlap: int = 4, win_length: int = 1024) -> Tensor:
    """Denoiser.forward, denoiser.py:58-64."""
    hop = filter_length // n_overlap
    win = torch.hann_window(win_length)
    spec = torch.stft(audio, n_fft=filter_length, hop_length=hop, win_length=win_length, window=win, return_complex=True)
    sr = torch.view_as_real(spec)
    mag = torch.sqrt(sr.pow(2).sum(-1))
    ang = torch.atan2(sr[..., -1], sr[..., 0])
    mag = torch.clamp(mag - bias_spec * strength, 0.0)
    return torch.istft(torch.complex(mag * torch.cos(ang), mag * torch.sin(ang)), n_fft=filter_length, hop_length=hop,
                       win_length=win_length, window=win)


def to_waveform(sd_voc: SD, h: dict, mel: Tensor, bias_spec: Optional[Tensor] = None) -> Tensor:
    """cli.py:121-126 / feel_me.py:181-187."""
    audio = hifigan_forward(sd_voc, mel, h).clamp(-1, 1)
    if bias_spec is not None:
        audio = denoiser(audio.squeeze(), bias_spec, strength=0.00025).squeeze()
    return audio.squeeze()


# ---------------------------------------------------------------------------
# feel_me.py emoji -> speaker rule
# ---------------------------------------------------------------------------
EMOJI_MAPPING = {  # feel_me.py:84-96
    "\U0001F60D": 107, "\U0001F621": 58, "\U0001F60E": 79, "\U0001F62D": 103, "\U0001F644": 66, "\U0001F601": 18,
    "\U0001F642": 12, "\U0001F923": 15, "\U0001F62E": 54, "\U0001F605": 22, "\U0001F914": 17,
}


def parse_emoji_response(response: str, is_emoji, replace_emoji, default_spk: int = 0):
    """feel_me.py:298-317: first mapped emoji in order of appearance wins (default 0);
    strip all emojis and brackets; empty -> 'nice'.  ``is_emoji``/``replace_emoji`` are
    the two functions the reference takes from the third-party ``emoji`` package."""
    spk = default_spk
    for ch in response:
        if is_emoji(ch) and ch in EMOJI_MAPPING:
            spk = EMOJI_MAPPING[ch]
            break
    text = replace_emoji(response, "")
    text = text.replace(")", "").replace("(", "")
    if text == "":
        text = "nice"
    return text, spk
