"""Parameter tables and synthetic (key-seeded) weights for the EmojiVoice hot path.

No trained checkpoint exists offline, so benchmarks, smoke and
parity tests use random-init weights of the reference architecture.  Every tensor
is generated from ``crc32(key)`` so that any process (this container, the GPU
box, every rank of a data-parallel job) regenerates bit-identical weights
without shipping files.

Key names and shapes are the reference ``state_dict`` names:
  * Matcha: ``matcha/models/matcha_tts.py:30-75`` (``spk_emb``, ``encoder.*``,
    ``decoder.estimator.*``, buffers ``mel_mean``/``mel_std`` from
    ``baselightningmodule.py:20-28``).
  * HiFi-GAN generator after ``remove_weight_norm()``
    (``matcha/hifigan/models.py:148-206``): ``conv_pre``, ``ups.N``,
    ``resblocks.N.convs{1,2}.M``, ``conv_post``.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

# ----------------------------------------------------------------------------
# Hyper-parameters restated from the reference config tree
# (configs/model/matcha.yaml, encoder/default.yaml, decoder/default.yaml,
#  configs/data/emoji_multi.yaml, hifigan/config.py:1-28)
# ----------------------------------------------------------------------------
N_FEATS = 80
SPK_EMB_DIM = 64
N_SPKS_EMOJI = 109
N_VOCAB_DEFAULT = 178
ENC_CHANNELS = 192
ENC_FILTER = 768
ENC_FILTER_DP = 256
ENC_HEADS = 2
ENC_LAYERS = 6
ENC_KERNEL = 3
DEC_CH = 256
DEC_HEADS = 2
DEC_HEAD_DIM = 64
DEC_TIME_DIM = 1024
DEC_FF = 1024
MEL_MEAN_EMOJI = -6.856600761413574
MEL_STD_EMOJI = 2.609809160232544

HIFIGAN_V1 = {
    "resblock": "1",
    "upsample_rates": [8, 8, 2, 2],
    "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512,
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "num_mels": 80,
    "sampling_rate": 22050,
    "hop_size": 256,
}

Shape = Tuple[int, ...]


def _resnet_shapes(p: str, cin: int, cout: int, out: "OrderedDict[str, Shape]") -> None:
    out[f"{p}.mlp.1.weight"] = (cout, DEC_TIME_DIM)
    out[f"{p}.mlp.1.bias"] = (cout,)
    out[f"{p}.block1.block.0.weight"] = (cout, cin, 3)
    out[f"{p}.block1.block.0.bias"] = (cout,)
    out[f"{p}.block1.block.1.weight"] = (cout,)
    out[f"{p}.block1.block.1.bias"] = (cout,)
    out[f"{p}.block2.block.0.weight"] = (cout, cout, 3)
    out[f"{p}.block2.block.0.bias"] = (cout,)
    out[f"{p}.block2.block.1.weight"] = (cout,)
    out[f"{p}.block2.block.1.bias"] = (cout,)
    out[f"{p}.res_conv.weight"] = (cout, cin, 1)
    out[f"{p}.res_conv.bias"] = (cout,)


def _transformer_shapes(p: str, dim: int, out: "OrderedDict[str, Shape]") -> None:
    inner = DEC_HEADS * DEC_HEAD_DIM
    out[f"{p}.norm1.weight"] = (dim,)
    out[f"{p}.norm1.bias"] = (dim,)
    out[f"{p}.attn1.to_q.weight"] = (inner, dim)
    out[f"{p}.attn1.to_k.weight"] = (inner, dim)
    out[f"{p}.attn1.to_v.weight"] = (inner, dim)
    out[f"{p}.attn1.to_out.0.weight"] = (dim, inner)
    out[f"{p}.attn1.to_out.0.bias"] = (dim,)
    out[f"{p}.norm3.weight"] = (dim,)
    out[f"{p}.norm3.bias"] = (dim,)
    out[f"{p}.ff.net.0.proj.weight"] = (DEC_FF, dim)
    out[f"{p}.ff.net.0.proj.bias"] = (DEC_FF,)
    out[f"{p}.ff.net.0.alpha"] = (DEC_FF,)
    out[f"{p}.ff.net.0.beta"] = (DEC_FF,)
    out[f"{p}.ff.net.2.weight"] = (dim, DEC_FF)
    out[f"{p}.ff.net.2.bias"] = (dim,)


def estimator_shapes(n_spks: int = N_SPKS_EMOJI, prefix: str = "decoder.estimator") -> "OrderedDict[str, Shape]":
    """Decoder (U-Net estimator) parameters, reference decoder.py:200-316."""
    cin0 = 2 * N_FEATS + (SPK_EMB_DIM if n_spks > 1 else 0)
    o: "OrderedDict[str, Shape]" = OrderedDict()
    p = prefix
    o[f"{p}.time_mlp.linear_1.weight"] = (DEC_TIME_DIM, cin0)
    o[f"{p}.time_mlp.linear_1.bias"] = (DEC_TIME_DIM,)
    o[f"{p}.time_mlp.linear_2.weight"] = (DEC_TIME_DIM, DEC_TIME_DIM)
    o[f"{p}.time_mlp.linear_2.bias"] = (DEC_TIME_DIM,)
    # down blocks: (resnet, [transformer], downsample)
    _resnet_shapes(f"{p}.down_blocks.0.0", cin0, DEC_CH, o)
    _transformer_shapes(f"{p}.down_blocks.0.1.0", DEC_CH, o)
    o[f"{p}.down_blocks.0.2.conv.weight"] = (DEC_CH, DEC_CH, 3)  # Downsample1D k3 s2 p1
    o[f"{p}.down_blocks.0.2.conv.bias"] = (DEC_CH,)
    _resnet_shapes(f"{p}.down_blocks.1.0", DEC_CH, DEC_CH, o)
    _transformer_shapes(f"{p}.down_blocks.1.1.0", DEC_CH, o)
    o[f"{p}.down_blocks.1.2.weight"] = (DEC_CH, DEC_CH, 3)  # plain Conv1d k3 p1
    o[f"{p}.down_blocks.1.2.bias"] = (DEC_CH,)
    for i in range(2):
        _resnet_shapes(f"{p}.mid_blocks.{i}.0", DEC_CH, DEC_CH, o)
        _transformer_shapes(f"{p}.mid_blocks.{i}.1.0", DEC_CH, o)
    _resnet_shapes(f"{p}.up_blocks.0.0", 2 * DEC_CH, DEC_CH, o)
    _transformer_shapes(f"{p}.up_blocks.0.1.0", DEC_CH, o)
    o[f"{p}.up_blocks.0.2.conv.weight"] = (DEC_CH, DEC_CH, 4)  # ConvTranspose1d (Cin, Cout, 4) s2 p1
    o[f"{p}.up_blocks.0.2.conv.bias"] = (DEC_CH,)
    _resnet_shapes(f"{p}.up_blocks.1.0", 2 * DEC_CH, DEC_CH, o)
    _transformer_shapes(f"{p}.up_blocks.1.1.0", DEC_CH, o)
    o[f"{p}.up_blocks.1.2.weight"] = (DEC_CH, DEC_CH, 3)
    o[f"{p}.up_blocks.1.2.bias"] = (DEC_CH,)
    o[f"{p}.final_block.block.0.weight"] = (DEC_CH, DEC_CH, 3)
    o[f"{p}.final_block.block.0.bias"] = (DEC_CH,)
    o[f"{p}.final_block.block.1.weight"] = (DEC_CH,)
    o[f"{p}.final_block.block.1.bias"] = (DEC_CH,)
    o[f"{p}.final_proj.weight"] = (N_FEATS, DEC_CH, 1)
    o[f"{p}.final_proj.bias"] = (N_FEATS,)
    return o


def text_encoder_shapes(n_vocab: int = N_VOCAB_DEFAULT, n_spks: int = N_SPKS_EMOJI, prefix: str = "encoder") -> "OrderedDict[str, Shape]":
    """TextEncoder parameters, reference text_encoder.py:328-376."""
    o: "OrderedDict[str, Shape]" = OrderedDict()
    p = prefix
    c = ENC_CHANNELS
    h = c + (SPK_EMB_DIM if n_spks > 1 else 0)
    o[f"{p}.emb.weight"] = (n_vocab, c)
    for i in range(3):
        o[f"{p}.prenet.conv_layers.{i}.weight"] = (c, c, 5)
        o[f"{p}.prenet.conv_layers.{i}.bias"] = (c,)
        o[f"{p}.prenet.norm_layers.{i}.gamma"] = (c,)
        o[f"{p}.prenet.norm_layers.{i}.beta"] = (c,)
    o[f"{p}.prenet.proj.weight"] = (c, c, 1)
    o[f"{p}.prenet.proj.bias"] = (c,)
    for i in range(ENC_LAYERS):
        for n in ("q", "k", "v", "o"):
            o[f"{p}.encoder.attn_layers.{i}.conv_{n}.weight"] = (h, h, 1)
            o[f"{p}.encoder.attn_layers.{i}.conv_{n}.bias"] = (h,)
        o[f"{p}.encoder.norm_layers_1.{i}.gamma"] = (h,)
        o[f"{p}.encoder.norm_layers_1.{i}.beta"] = (h,)
        o[f"{p}.encoder.ffn_layers.{i}.conv_1.weight"] = (ENC_FILTER, h, ENC_KERNEL)
        o[f"{p}.encoder.ffn_layers.{i}.conv_1.bias"] = (ENC_FILTER,)
        o[f"{p}.encoder.ffn_layers.{i}.conv_2.weight"] = (h, ENC_FILTER, ENC_KERNEL)
        o[f"{p}.encoder.ffn_layers.{i}.conv_2.bias"] = (h,)
        o[f"{p}.encoder.norm_layers_2.{i}.gamma"] = (h,)
        o[f"{p}.encoder.norm_layers_2.{i}.beta"] = (h,)
    o[f"{p}.proj_m.weight"] = (N_FEATS, h, 1)
    o[f"{p}.proj_m.bias"] = (N_FEATS,)
    o[f"{p}.proj_w.conv_1.weight"] = (ENC_FILTER_DP, h, 3)
    o[f"{p}.proj_w.conv_1.bias"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.norm_1.gamma"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.norm_1.beta"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.conv_2.weight"] = (ENC_FILTER_DP, ENC_FILTER_DP, 3)
    o[f"{p}.proj_w.conv_2.bias"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.norm_2.gamma"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.norm_2.beta"] = (ENC_FILTER_DP,)
    o[f"{p}.proj_w.proj.weight"] = (1, ENC_FILTER_DP, 1)
    o[f"{p}.proj_w.proj.bias"] = (1,)
    return o


def matcha_shapes(n_vocab: int = N_VOCAB_DEFAULT, n_spks: int = N_SPKS_EMOJI) -> "OrderedDict[str, Shape]":
    o: "OrderedDict[str, Shape]" = OrderedDict()
    if n_spks > 1:
        o["spk_emb.weight"] = (n_spks, SPK_EMB_DIM)
    o.update(text_encoder_shapes(n_vocab, n_spks))
    o.update(estimator_shapes(n_spks))
    return o


def hifigan_shapes(h: dict = HIFIGAN_V1) -> "OrderedDict[str, Shape]":
    """Generator parameters with weight-norm already folded (models.py:148-179)."""
    o: "OrderedDict[str, Shape]" = OrderedDict()
    c0 = h["upsample_initial_channel"]
    o["conv_pre.weight"] = (c0, h["num_mels"], 7)
    o["conv_pre.bias"] = (c0,)
    ch = c0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        o[f"ups.{i}.weight"] = (c0 // (2**i), c0 // (2 ** (i + 1)), k)  # ConvTranspose1d (Cin, Cout, k)
        o[f"ups.{i}.bias"] = (c0 // (2 ** (i + 1)),)
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(h["upsample_rates"])):
        ch = c0 // (2 ** (i + 1))
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            for m in range(3):
                for cs in ("convs1", "convs2"):
                    o[f"resblocks.{i * nk + j}.{cs}.{m}.weight"] = (ch, ch, k)
                    o[f"resblocks.{i * nk + j}.{cs}.{m}.bias"] = (ch,)
    o["conv_post.weight"] = (1, ch, 7)
    o["conv_post.bias"] = (1,)
    return o


def count_params(shapes: Dict[str, Shape]) -> int:
    return sum(int(math.prod(s)) for s in shapes.values())


# ----------------------------------------------------------------------------
# Key-seeded synthetic weights
# ----------------------------------------------------------------------------
def _gen(key: str, salt: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32((salt + "/" + key).encode("utf-8")) & 0x7FFFFFFF)
    return g


def _randn(key: str, shape: Shape, std: float, salt: str, mean: float = 0.0) -> torch.Tensor:
    t = torch.randn(shape, generator=_gen(key, salt), dtype=torch.float32)
    return t * std + mean


def synthetic_matcha_state(n_vocab: int = N_VOCAB_DEFAULT, n_spks: int = N_SPKS_EMOJI, salt: str = "ev0") -> "OrderedDict[str, torch.Tensor]":
    """Random weights for the full MatchaTTS module (encoder + estimator + spk_emb).

    Scales keep activations O(1): matrices ~ N(0, 1/fan_in); norm gains ~ 1;
    biases and SnakeBeta log-params small but non-zero so every term is exercised.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, s in matcha_shapes(n_vocab, n_spks).items():
        leaf = k.rsplit(".", 1)[-1]
        if k == "spk_emb.weight":
            t = _randn(k, s, 1.0, salt)
        elif k.endswith("emb.weight"):
            t = _randn(k, s, ENC_CHANNELS**-0.5, salt)
        elif leaf in ("alpha", "beta") and ".ff.net.0." in k:
            t = _randn(k, s, 0.2, salt)
        elif leaf in ("gamma",) or (leaf == "weight" and len(s) == 1):
            t = _randn(k, s, 0.1, salt, mean=1.0)
        elif leaf in ("beta", "bias"):
            t = _randn(k, s, 0.1, salt)
        else:  # conv / linear matrices
            fan_in = int(math.prod(s[1:]))
            if k.endswith("up_blocks.0.2.conv.weight"):  # ConvTranspose1d: (Cin, Cout, k), 2 taps hit each output
                fan_in = s[0] * 2
            t = _randn(k, s, 1.0 / math.sqrt(fan_in), salt)
        sd[k] = t
    sd["mel_mean"] = torch.tensor(MEL_MEAN_EMOJI, dtype=torch.float32)
    sd["mel_std"] = torch.tensor(MEL_STD_EMOJI, dtype=torch.float32)
    return sd


def synthetic_hifigan_state(h: dict = HIFIGAN_V1, salt: str = "ev0") -> "OrderedDict[str, torch.Tensor]":
    """Random folded-weight-norm HiFi-GAN generator weights.

    The reference ``init_weights`` (xutils.py:25-28, std 0.01) yields ~1e-2 RMS
    audio that makes the 1e-3 waveform gate vacuous, so matrices
    are drawn at variance-preserving scale instead: the residual branch of each
    ResBlock1 pair is damped (0.5/sqrt(fan_in) on its second conv) and
    ``conv_post`` targets a pre-tanh std of ~0.5.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    strides = h["upsample_rates"]
    for k, s in hifigan_shapes(h).items():
        if k.endswith(".bias"):
            t = _randn(k, s, 0.05, salt)
        elif k.startswith("ups."):
            i = int(k.split(".")[1])
            fan_in = s[0] * s[2] // strides[i]
            t = _randn(k, s, 1.3 / math.sqrt(fan_in), salt)
        elif ".convs2." in k:
            t = _randn(k, s, 0.5 / math.sqrt(s[1] * s[2]), salt)
        elif k == "conv_post.weight":
            t = _randn(k, s, 0.5 / math.sqrt(s[1] * s[2]), salt)
        else:
            t = _randn(k, s, 1.3 / math.sqrt(s[1] * s[2]), salt)
        sd[k] = t
    return sd


def weight_norm_split(sd: Dict[str, torch.Tensor], salt: str = "ev0") -> "OrderedDict[str, torch.Tensor]":
    """Inverse of ``remove_weight_norm``: turn folded weights into the
    ``weight_g``/``weight_v`` pairs a real HiFi-GAN ``generator`` checkpoint
    stores (torch.nn.utils.weight_norm, dim=0).  ``weight_v`` is a randomly
    rescaled copy so the fold is a real computation in tests."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, w in sd.items():
        if k.endswith(".weight"):
            base = k[: -len(".weight")]
            scale = torch.rand((w.shape[0],) + (1,) * (w.dim() - 1), generator=_gen(k, salt + "wn")) + 0.5
            v = w * scale
            g = w.flatten(1).norm(dim=1).view_as(scale)
            out[base + ".weight_g"] = g
            out[base + ".weight_v"] = v
        else:
            out[k] = w
    return out


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """``remove_weight_norm`` (hifigan/models.py:199-206, cli.py:84-90) on a raw
    ``generator`` state dict: w = g * v / ||v|| with the norm over all dims but 0."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, t in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            v = sd[base + ".weight_v"].float()
            g = t.float()
            n = v.flatten(1).norm(dim=1).view((-1,) + (1,) * (v.dim() - 1))
            out[base + ".weight"] = v * (g / n)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = t
    return out
