"""Counterpart of the ``matcha-tts`` CLI (reference Matcha-TTS/matcha/cli.py:160-250) for the MI355X path
(SURVEY §8 f-2).  The phonemiser front end (espeak-ng) is out of scope, so utterances are given as phoneme-id
sequences (``--ids "12 0 45 ..."`` or ``--file`` with one sequence per line, optional ``|speaker`` suffix like
cli.py:332-336) or as pre-phonemised IPA text (``--phonemes "həlˈoʊ wˈɜːld"``, mapped through the reference's symbol table,
emojivoice_amd/text.py); everything after that point mirrors the reference: validate_args (:138-158), load_matcha / load_vocoder
(:84-118), unbatched / batched synthesis (:277-317, :389-425), ``to_waveform`` (:121-126), PCM_24 wav files (:134).

    python -m emojivoice_amd.cli --checkpoint_path model.ckpt --vocoder_path g_02500000 --ids "0 23 0 51 0" --spk 12
    python -m emojivoice_amd.cli --synthetic --emoji-text "Hello world 🙂" --ids "0 23 0 51 0"
"""
from __future__ import annotations

import argparse
import datetime as dt
import os
import struct
import sys
import warnings
from pathlib import Path

import numpy as np
import torch


from .text import cleaned_text_to_sequence, intersperse  # noqa: E402  (utils/utils.py:131-135, text/__init__.py:27-35)


def write_wav_pcm24(path, wav: np.ndarray, sr: int = 22050):
    """soundfile.write(..., 'PCM_24') equivalent (cli.py:134) with the stdlib only."""
    x = np.clip(np.asarray(wav, dtype=np.float64), -1.0, 1.0)
    q = np.round(x * (2**23 - 1)).astype(np.int32)
    b = (q & 0xFFFFFF).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(b)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 3, 3, 24))
        f.write(b"data" + struct.pack("<I", len(b)) + b)


def validate_args(args):
    assert args.ids or args.file or args.phonemes, "One of --ids, --phonemes or --file must be provided"
    assert args.temperature >= 0, "Sampling temperature cannot be negative"
    assert args.steps > 0, "Number of ODE steps must be greater than 0"
    if args.speaking_rate is None:
        args.speaking_rate = 1.0
    if args.batched:
        assert args.batch_size > 0, "Batch size must be greater than 0"
    assert args.speaking_rate > 0, "Speaking rate must be greater than 0"
    return args


def load_models(args, device):
    from . import weights as W
    from .denoiser import Denoiser
    from .hifigan import AttrDict, Generator, v1
    from .matcha_tts import MatchaTTS

    if args.synthetic:
        model = MatchaTTS(W.synthetic_matcha_state(), device=device)
        voc_sd = W.synthetic_hifigan_state()
    else:
        model = MatchaTTS.load_from_checkpoint(args.checkpoint_path, map_location=device)
        voc_sd = torch.load(args.vocoder_path, map_location="cpu")["generator"]
    vocoder = Generator(AttrDict(v1)).to(device)
    vocoder.load_state_dict(voc_sd)
    vocoder.eval()
    vocoder.remove_weight_norm()
    denoiser = Denoiser(vocoder, mode="zeros") if args.denoiser_strength > 0 else None
    return model.eval(), vocoder, denoiser


@torch.inference_mode()
def to_waveform(mel, vocoder, denoiser=None, strength=0.00025):
    audio = vocoder(mel).clamp(-1, 1)
    if denoiser is not None:
        audio = denoiser(audio.squeeze(), strength=strength).cpu().squeeze()
    return audio.cpu().squeeze()


def parse_lines(args):
    if args.phonemes is not None:
        # pre-phonemised (IPA) text -> ids exactly as process_text does after its cleaner (cli.py:52-56): always with blanks
        return [(intersperse(cleaned_text_to_sequence(args.phonemes), 0), None)]
    lines = [args.ids] if args.ids else open(args.file, encoding="utf-8").read().splitlines()
    out = []
    for ln in lines:
        ln = ln.strip()
        if not ln:
            continue
        spk = None
        if "|" in ln:
            ln, s = ln.rsplit("|", 1)
            spk = int(s)
        if args.file_phonemes:
            ids = intersperse(cleaned_text_to_sequence(ln), 0)
        else:
            ids = [int(t) for t in ln.split()]
            if args.add_blank:
                ids = intersperse(ids, 0)
        out.append((ids, spk))
    return out


@torch.inference_mode()
def cli(argv=None):
    p = argparse.ArgumentParser(description="Matcha-TTS / EmojiVoice synthesis on MI355X")
    p.add_argument("--checkpoint_path", type=str, default=None)
    p.add_argument("--vocoder_path", type=str, default=None, help="HiFi-GAN generator checkpoint (dict with 'generator')")
    p.add_argument("--synthetic", action="store_true", help="random-init weights (no checkpoint is available offline)")
    p.add_argument("--ids", type=str, default=None, help="phoneme ids of one utterance, space separated")
    p.add_argument("--phonemes", type=str, default=None, help="one pre-phonemised (IPA) utterance, e.g. the output of english_cleaners2; "
                   "mapped through the 198-symbol table and interspersed with blanks like the reference front end")
    p.add_argument("--file", type=str, default=None, help="one id sequence per line, optional '|speaker'")
    p.add_argument("--file_phonemes", action="store_true", help="lines of --file are IPA strings, not ids")
    p.add_argument("--add_blank", action="store_true", help="intersperse ids with 0 like the reference front end")
    p.add_argument("--emoji-text", type=str, default=None, help="LLM-style text; its first mapped emoji selects the speaker (feel_me.py)")
    p.add_argument("--spk", type=int, default=None)
    p.add_argument("--temperature", type=float, default=0.667)
    p.add_argument("--speaking_rate", type=float, default=None)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--denoiser_strength", type=float, default=0.00025)
    p.add_argument("--output_folder", type=str, default=os.getcwd())
    p.add_argument("--batched", action="store_true")
    p.add_argument("--batch_size", type=int, default=32)
    args = validate_args(p.parse_args(argv))
    if not args.synthetic:
        assert args.checkpoint_path and args.vocoder_path, "--checkpoint_path and --vocoder_path are required (or --synthetic)"
    if not torch.cuda.is_available():
        sys.exit("[-] No ROCm GPU visible: this CLI drives the MI355X path only (no CPU fallback)")
    device = torch.device("cuda", 0)
    model, vocoder, denoiser = load_models(args, device)
    spk_default = args.spk
    if args.emoji_text is not None:
        from .emoji import parse_response

        _, spk_default = parse_response(args.emoji_text)
        print(f"[emoji] speaker {spk_default} selected from {args.emoji_text!r}")
    if spk_default is None:
        warnings.warn("[-] No speaker provided, using speaker number 0.", UserWarning)
        spk_default = 0
    items = parse_lines(args)
    folder = Path(args.output_folder)
    folder.mkdir(exist_ok=True, parents=True)
    rtfs = []
    bs = args.batch_size if args.batched else 1
    for b0 in range(0, len(items), bs):
        chunk = items[b0:b0 + bs]
        lens = torch.tensor([len(i) for i, _ in chunk], dtype=torch.long)
        x = torch.zeros(len(chunk), int(lens.max()), dtype=torch.long)
        for r, (ids, _) in enumerate(chunk):
            x[r, :len(ids)] = torch.tensor(ids)
        spks = torch.tensor([s if s is not None else spk_default for _, s in chunk], dtype=torch.long)
        t0 = dt.datetime.now()
        out = model.synthesise(x.to(device), lens.to(device), n_timesteps=args.steps, temperature=args.temperature,
                               spks=spks.to(device), length_scale=args.speaking_rate)
        wav = to_waveform(out["mel"], vocoder, denoiser, args.denoiser_strength)
        t = (dt.datetime.now() - t0).total_seconds()
        wav = wav.reshape(len(chunk), -1)
        rtf_w = t * 22050 / wav.shape[-1] / len(chunk)
        rtfs.append(rtf_w)
        print(f"[batch {b0 // bs + 1}] Matcha-TTS RTF: {out['rtf']:.4f}  + VOCODER RTF: {rtf_w:.4f}")
        for r in range(len(chunk)):
            n = int(out["mel_lengths"][r])
            name = f"utterance_{b0 + r + 1:03d}_speaker_{int(spks[r]):03d}"
            np.save(folder / name, out["mel"][r, :, :n].cpu().numpy())
            write_wav_pcm24(folder / f"{name}.wav", wav[r, : n * 256].numpy())
            print(f"[+] Waveform saved: {(folder / (name + '.wav')).resolve()}  ({n * 256 / 22050:.2f} s)")
    print(f"[avg] Matcha-TTS + VOCODER RTF: {np.mean(rtfs):.4f} ± {np.std(rtfs):.4f}")


if __name__ == "__main__":
    cli()
