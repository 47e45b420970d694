"""Host stage of ``MatchaTTS.synthesise``: text encoder, duration predictor and the
hard monotonic alignment.  north_star keeps this stage on
plain PyTorch ("run once on host"); it is <1 % of the path's FLOPs.  Functional
torch code over the reference state-dict names, runs on whatever device the
weights live on.

Follows reference ``matcha/models/components/text_encoder.py`` (TextEncoder.forward
:378-410 with ConvReluNorm :36-67, Encoder :276-325, MultiHeadAttention + RoPE
:97-246, FFN :255-273, DurationPredictor :70-94, channel LayerNorm :15-33) and
``matcha/utils/model.py`` (sequence_mask :7-11, fix_len_compatibility :14-20,
generate_path :29-41).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def sequence_mask(length: Tensor, max_length: Optional[int] = None) -> Tensor:
    if max_length is None:
        max_length = int(length.max())
    pos = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return pos[None, :] < length[:, None]


def fix_len_compatibility(length: int, num_downsamplings_in_unet: int = 2) -> int:
    f = 2**num_downsamplings_in_unet
    return int(math.ceil(float(length) / f) * f)


def generate_path(duration: Tensor, mask: Tensor) -> Tensor:
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1).view(b * t_x)
    path = sequence_mask(cum, t_y).to(mask.dtype).view(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0, 0, 0))[:, :-1]
    return path * mask


def _cln(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-4) -> Tensor:
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def _rotary(x: Tensor, d: int, base: float = 10000.0) -> Tensor:
    t = x.shape[2]
    theta = 1.0 / (base ** (torch.arange(0, d, 2, device=x.device).float() / d))
    ang = torch.einsum("n,d->nd", torch.arange(t, device=x.device).float(), theta)
    ang = torch.cat([ang, ang], dim=1)
    cos, sin = ang.cos()[None, None], ang.sin()[None, None]
    xr, xp = x[..., :d], x[..., d:]
    rot = torch.cat([-xr[..., d // 2:], xr[..., : d // 2]], dim=-1)
    return torch.cat([xr * cos + rot * sin, xp], dim=-1)


class TextEncoder:
    def __init__(self, sd: Dict[str, Tensor], n_heads: int = 2, n_layers: int = 6, prefix: str = "encoder"):
        self.sd, self.n_heads, self.n_layers, self.p = sd, n_heads, n_layers, prefix

    def _conv(self, name: str, x: Tensor) -> Tensor:
        w = self.sd[f"{name}.weight"]
        b = self.sd[f"{name}.bias"]
        k = w.shape[2]
        if not x.is_cuda:
            return F.conv1d(x, w, b, padding=k // 2)
        # On the GPU a "same" Conv1d is evaluated as k shifted GEMMs (rocBLAS): MIOpen's conv1d searches / compiles a
        # solver for every new (B, L), which costs 50-500 ms the first time a streaming caller sees a text length.
        if k == 1:
            return torch.matmul(w[:, :, 0], x) + b[:, None]
        xp = F.pad(x, (k // 2, k // 2))
        L = x.shape[-1]
        cols = torch.cat([xp[:, :, i:i + L] for i in range(k)], dim=1)          # (B, k*Cin, L), tap-major
        wk = w.permute(0, 2, 1).reshape(w.shape[0], -1)                         # (Cout, k*Cin)
        return torch.matmul(wk, cols) + b[:, None]

    def _attn(self, name: str, x: Tensor, attn_mask: Tensor) -> Tensor:
        q, k, v = (self._conv(f"{name}.conv_{n}", x) for n in "qkv")
        b, d, t = k.shape
        h = self.n_heads
        kc = d // h
        q, k, v = (a.view(b, h, kc, t).transpose(2, 3) for a in (q, k, v))
        q, k = _rotary(q, int(kc * 0.5)), _rotary(k, int(kc * 0.5))
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(kc)
        scores = scores.masked_fill(attn_mask == 0, -1e4)
        o = torch.matmul(F.softmax(scores, dim=-1), v)
        o = o.transpose(2, 3).contiguous().view(b, d, t)
        return self._conv(f"{name}.conv_o", o)

    @torch.inference_mode()
    def __call__(self, x: Tensor, x_lengths: Tensor, spks: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
        sd, p = self.sd, self.p
        c = sd[f"{p}.emb.weight"].shape[1]
        h = F.embedding(x, sd[f"{p}.emb.weight"]) * math.sqrt(c)
        h = h.transpose(1, -1)
        x_mask = sequence_mask(x_lengths, h.size(2)).unsqueeze(1).to(h.dtype)
        res = h
        for i in range(3):
            h = self._conv(f"{p}.prenet.conv_layers.{i}", h * x_mask)
            h = torch.relu(_cln(h, sd[f"{p}.prenet.norm_layers.{i}.gamma"], sd[f"{p}.prenet.norm_layers.{i}.beta"]))
        h = (res + self._conv(f"{p}.prenet.proj", h)) * x_mask
        if spks is not None:
            h = torch.cat([h, spks.unsqueeze(-1).repeat(1, 1, h.shape[-1])], dim=1)
        am = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
        e = f"{p}.encoder"
        for i in range(self.n_layers):
            h = h * x_mask
            y = self._attn(f"{e}.attn_layers.{i}", h, am)
            h = _cln(h + y, sd[f"{e}.norm_layers_1.{i}.gamma"], sd[f"{e}.norm_layers_1.{i}.beta"])
            y = torch.relu(self._conv(f"{e}.ffn_layers.{i}.conv_1", h * x_mask))
            y = self._conv(f"{e}.ffn_layers.{i}.conv_2", y * x_mask) * x_mask
            h = _cln(h + y, sd[f"{e}.norm_layers_2.{i}.gamma"], sd[f"{e}.norm_layers_2.{i}.beta"])
        h = h * x_mask
        mu = self._conv(f"{p}.proj_m", h) * x_mask
        w = f"{p}.proj_w"
        d = torch.relu(self._conv(f"{w}.conv_1", h * x_mask))
        d = _cln(d, sd[f"{w}.norm_1.gamma"], sd[f"{w}.norm_1.beta"])
        d = torch.relu(self._conv(f"{w}.conv_2", d * x_mask))
        d = _cln(d, sd[f"{w}.norm_2.gamma"], sd[f"{w}.norm_2.beta"])
        logw = self._conv(f"{w}.proj", d * x_mask) * x_mask
        return mu, logw, x_mask
