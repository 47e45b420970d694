"""Drop-in for ``matcha.hifigan.denoiser.Denoiser``: STFT-domain bias subtraction that every reference
caller applies right after the vocoder (cli.py:121-126, feel_me.py:181-187).

Follows hifigan/denoiser.py:10-64: bias spectrum = |STFT(vocoder(zeros(1, 80, 88)))|[:, :, 0]; forward =
ISTFT(clamp(|STFT(audio)| - bias * strength, 0) * exp(i * angle)).  Both the vocoder call and the STFT pair go through the
HIP library (``ev_stft_magnitude`` / ``ev_denoise``: the windowed DFT bases as two 4-tap convolutions on the matrix cores).
Waveforms whose length is not a multiple of 256 (never produced by the vocoder) are not supported.
"""
from __future__ import annotations

import torch


class Denoiser:
    def __init__(self, vocoder, filter_length: int = 1024, n_overlap: int = 4, win_length: int = 1024, mode: str = "zeros"):
        if filter_length != 1024 or n_overlap != 4 or win_length != 1024:
            raise ValueError("the HIP denoiser implements the reference's only configuration: filter 1024, overlap 4, window 1024")
        self.filter_length, self.hop_length, self.win_length = filter_length, filter_length // n_overlap, win_length
        self.device = device = vocoder.device
        self.vocoder = vocoder                      # the STFT pair runs on the vocoder's native handle
        if mode == "zeros":
            mel_input = torch.zeros((1, 80, 88), dtype=torch.float32, device=device)
        elif mode == "normal":
            mel_input = torch.randn((1, 80, 88), dtype=torch.float32, device=device)
        else:
            raise Exception(f"Mode {mode} if not supported")
        with torch.no_grad():
            bias_audio = vocoder(mel_input).float().squeeze(0)          # (1, L)
            bias_spec = vocoder.engine.stft_magnitude(bias_audio)       # (1, 513, F)
        self.bias_spec = bias_spec[:, :, 0][:, :, None]

    @torch.inference_mode()
    def forward(self, audio, strength: float = 0.0005):
        audio = audio.to(self.device)
        shape = audio.shape
        self.vocoder._sync_engine()
        out = self.vocoder.engine.denoise(audio.reshape(-1, shape[-1]), self.bias_spec.reshape(-1), strength)
        # the reference's (1, 513, 1) bias spectrum broadcasts a 1-D input's spectrum to a batch of one: (L,) -> (1, L), (B, L) -> (B, L)
        return out.reshape((1, shape[-1])) if audio.dim() == 1 else out.reshape(shape)

    __call__ = forward
