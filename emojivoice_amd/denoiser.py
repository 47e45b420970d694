"""Drop-in for ``matcha.hifigan.denoiser.Denoiser``: STFT-domain bias subtraction that every reference
caller applies right after the vocoder (cli.py:121-126, feel_me.py:181-187).

Follows hifigan/denoiser.py:10-64: bias spectrum = |STFT(vocoder(zeros(1, 80, 88)))|[:, :, 0]; forward =
ISTFT(clamp(|STFT(audio)| - bias * strength, 0) * exp(i * angle)).  The vocoder call goes through the HIP path; the
STFT/ISTFT are torch ops on the GPU (rocFFT), this stage is ~0.1 % of the path's FLOPs.
"""
from __future__ import annotations

import torch


class Denoiser:
    def __init__(self, vocoder, filter_length: int = 1024, n_overlap: int = 4, win_length: int = 1024, mode: str = "zeros"):
        self.filter_length = filter_length
        self.hop_length = int(filter_length / n_overlap)
        self.win_length = win_length
        self.device = device = vocoder.device
        if mode == "zeros":
            mel_input = torch.zeros((1, 80, 88), dtype=torch.float32, device=device)
        elif mode == "normal":
            mel_input = torch.randn((1, 80, 88), dtype=torch.float32, device=device)
        else:
            raise Exception(f"Mode {mode} if not supported")
        self.window = torch.hann_window(win_length, device=device)
        with torch.no_grad():
            bias_audio = vocoder(mel_input).float().squeeze(0)
            bias_spec, _ = self._stft(bias_audio)
        self.bias_spec = bias_spec[:, :, 0][:, :, None]

    def _stft(self, audio):
        spec = torch.stft(audio, n_fft=self.filter_length, hop_length=self.hop_length, win_length=self.win_length,
                          window=self.window, return_complex=True)
        re = torch.view_as_real(spec)
        return torch.sqrt(re.pow(2).sum(-1)), torch.atan2(re[..., -1], re[..., 0])

    def _istft(self, mag, ang):
        return torch.istft(torch.complex(mag * torch.cos(ang), mag * torch.sin(ang)), n_fft=self.filter_length,
                           hop_length=self.hop_length, win_length=self.win_length, window=self.window)

    @torch.inference_mode()
    def forward(self, audio, strength: float = 0.0005):
        mag, ang = self._stft(audio.to(self.device))
        mag = torch.clamp(mag - self.bias_spec * strength, 0.0)
        return self._istft(mag, ang)

    __call__ = forward
