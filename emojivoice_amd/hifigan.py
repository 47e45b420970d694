"""Drop-in for ``matcha.hifigan.models.Generator`` (+ ``config.v1``, ``env.AttrDict``).

Call surface kept from the reference (hifigan/models.py:148-206, cli.py:84-90):

    h = AttrDict(v1); g = Generator(h).to(device)
    g.load_state_dict(torch.load(path)["generator"]); g.eval(); g.remove_weight_norm()
    wav = g(mel)            # (B, 80, T) -> (B, 1, 256 T), fp32, tanh output

The forward runs entirely in the HIP library (``ev_hifigan``); there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import weights as W
from ._lib import Engine, EvLibraryError

v1 = dict(W.HIFIGAN_V1, num_gpus=0, batch_size=16, learning_rate=0.0004, adam_b1=0.8, adam_b2=0.99, lr_decay=0.999, seed=1234,
          resblock_initial_channel=256, segment_size=8192, num_freq=1025, n_fft=1024, win_size=1024, fmin=0, fmax=8000,
          fmax_loss=None, num_workers=4)


class AttrDict(dict):
    """hifigan/env.py:7-10."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


class Generator:
    def __init__(self, h):
        self.h = h
        if str(h["resblock"]) != "1" or list(h["upsample_rates"]) != [8, 8, 2, 2] or list(h["resblock_kernel_sizes"]) != [3, 7, 11]:
            raise ValueError("only the HiFi-GAN V1 configuration (hifigan/config.py:1-28) is implemented")
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        self._raw: Optional[Dict[str, torch.Tensor]] = None
        self._folded: Optional[Dict[str, torch.Tensor]] = None
        self.engine: Optional[Engine] = None
        self.device = torch.device("cpu")
        self._dirty = True

    # ---- nn.Module-like surface -------------------------------------------------
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise EvLibraryError("emojivoice_amd.hifigan.Generator runs on a ROCm GPU only (no CPU fallback)")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._dirty = True
        return self

    def eval(self):
        return self

    def parameters(self):
        sd = self._folded or self._raw or {}
        for v in sd.values():
            yield v.to(self.device) if self.device.type == "cuda" else v

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        has_wn = any(k.endswith(".weight_g") for k in sd)
        folded = W.fold_weight_norm(sd) if has_wn else sd
        if strict:
            want = W.hifigan_shapes(self.h)
            missing = [k for k in want if k not in folded]
            bad = [k for k in want if k in folded and tuple(folded[k].shape) != tuple(want[k])]
            extra = [k for k in folded if k not in want]
            if missing or bad or extra:
                raise RuntimeError(f"Generator.load_state_dict: missing={missing[:4]} shape-mismatch={bad[:4]} unexpected={extra[:4]}")
        self._raw = sd
        self._folded = None if has_wn else folded
        self._pending_fold = folded
        self._dirty = True
        return self

    def remove_weight_norm(self):
        """hifigan/models.py:199-206: fold ``weight_g * weight_v / ||weight_v||`` into plain weights."""
        print("Removing weight norm...")
        if self._raw is None:
            raise RuntimeError("load_state_dict first")
        self._folded = self._pending_fold
        self._dirty = True

    def state_dict(self):
        return dict(self._folded if self._folded is not None else (self._raw or {}))

    # ---- the hot call --------------------------------------------------------------
    def _sync_engine(self):
        if self.device.type != "cuda":
            raise EvLibraryError("Generator must be moved to a ROCm GPU with .to('cuda') (no CPU fallback)")
        if self._raw is None:
            raise RuntimeError("Generator has no weights: call load_state_dict")
        if self._folded is None:
            # the reference would run with live weight-norm parametrisation: identical values
            self._folded = self._pending_fold
        if self.engine is None or self._dirty:
            if self.engine is not None:
                self.engine.close()
            self.engine = Engine(self.device.index, spk_emb_dim=64)
            self.engine.load_vocoder(self._folded)
            self._dirty = False

    @torch.inference_mode()
    def warmup(self, frames: int = 8, max_frames: int = 0, batch: int = 1) -> None:
        """One tiny call: loads the code objects and allocates a first workspace (see MatchaTTS.warmup).  ``max_frames``: pre-size
        the workspace (and the denoiser's scratch) for utterances of up to that many mel frames through ``ev_reserve``."""
        self._sync_engine()
        if max_frames > 0:
            self.engine.reserve(batch, 0, 0, max_frames)
        self.forward(torch.zeros((1, 80, frames), dtype=torch.float32, device=self.device))
        torch.cuda.synchronize(self.device)

    @torch.inference_mode()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._sync_engine()
        return self.engine.hifigan(x.to(self.device))

    __call__ = forward


def synthetic(device="cuda:0") -> Generator:
    g = Generator(AttrDict(v1)).to(device)
    g.load_state_dict(W.synthetic_hifigan_state())
    return g


@torch.inference_mode()
def to_waveform(mel, vocoder, denoiser=None):
    """cli.py:121-126 / feel_me.py:181-187."""
    audio = vocoder(mel).clamp(-1, 1)
    if denoiser is not None:
        audio = denoiser(audio.squeeze(), strength=0.00025).cpu().squeeze()
    return audio.cpu().squeeze()
