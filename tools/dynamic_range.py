#!/usr/bin/env python3
"""Within-tile dynamic range of the block-scaled fp16 datapath.

The fp16 builds take ONE power-of-two scale per 128-row tile (per 64-channel chunk in the balanced builds), so a row that is many octaves
quieter than the loudest row of its tile keeps fewer significand bits (fp16's subnormal floor, DESIGN section 3 "Error bound").  This probe
measures the case the per-utterance scale test cannot see: ONE utterance whose frames alternate in blocks of 16 between scale 1 and
scale 2^-k (k = 4 .. 24), through a 3-tap conv layer (rows 1..14 of a quiet block see only quiet inputs), error PER OUTPUT ROW relative
to that row's own fp64 RMS, for the arithmetic settings 16 (shipped) / 6 (bf16 six products) / 0 (exact fp32 MFMA) side by side; and a
ragged batch whose padded frames carry `mel_mean` through the whole vocoder, error per utterance over its valid samples.

    python tools/dynamic_range.py            prints the table (committed as profiles/r04_within_tile_dynamic_range.txt)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def conv_rows_table(eng, ks=(0, 4, 8, 12, 16, 20, 24), T=140000, C=128):
    """{k: {arith: (worst quiet-row error, rms quiet-row error, worst loud-row error)}}, errors relative to each row's own fp64 RMS."""
    g = torch.Generator().manual_seed(77)
    w = torch.randn(C, C, 3, generator=g) / (C * 3) ** 0.5
    base = torch.randn(1, C, T, generator=g)
    blk = (torch.arange(T) // 16) % 2                      # 0 = loud block, 1 = quiet block
    pos = torch.arange(T) % 16
    pure_quiet = (blk == 1) & (pos >= 1) & (pos <= 14)     # rows whose three taps all read quiet frames
    pure_loud = (blk == 0) & (pos >= 1) & (pos <= 14)
    out = {}
    orig = eng.arithmetic()
    try:
        for k in ks:
            scale = torch.where(blk == 1, torch.tensor(2.0 ** -k), torch.tensor(1.0)).view(1, 1, T)
            x = base * scale
            ref = F.conv1d(x.double(), w.double(), None, padding=1)[0]          # (C, T)
            row_rms = ref.pow(2).mean(dim=0).sqrt()
            rec = {}
            for a in (16, 6, 0):
                eng.set_arithmetic(a)
                y = eng.op_conv1d(x.cuda(), w, None, dilation=1, padding=1).cpu().double()[0]
                cfg = eng.last_cfg()
                e = (y - ref).abs().max(dim=0).values / row_rms
                rec[a] = (float(e[pure_quiet].max()), float(e[pure_quiet].pow(2).mean().sqrt()), float(e[pure_loud].max()), cfg)
            out[k] = rec
    finally:
        eng.set_arithmetic(orig)
    return out


def vocoder_ragged_table(lengths=(516, 300, 120, 40)):
    """{arith: [per-utterance RMS error over the valid samples / that utterance's fp64 RMS]} for a ragged batch padded with mel_mean."""
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from oracle import matcha_oracle as O

    voc_sd = W.synthetic_hifigan_state()
    g = torch.Generator().manual_seed(78)
    T = max(lengths)
    mel = torch.randn(len(lengths), 80, T, generator=g) * 2.0 - 5.0
    for i, n in enumerate(lengths):
        mel[i, :, n:] = -6.8566                               # what the decoder's masked frames denormalise to (emoji_multi.yaml mel_mean)
    ref = O.hifigan_forward({k: v.double() for k, v in voc_sd.items()}, mel.double(), W.HIFIGAN_V1)[:, 0]
    voc = Generator(AttrDict(v1)).to("cuda:0")
    voc.load_state_dict(voc_sd)
    out = {}
    voc._sync_engine()
    orig = voc.engine.arithmetic()
    try:
        for a in (16, 6, 0):
            voc.engine.set_arithmetic(a)
            y = voc(mel.cuda()).cpu().double()[:, 0]
            out[a] = [float(((y[i, :256 * n] - ref[i, :256 * n]).pow(2).mean() / ref[i, :256 * n].pow(2).mean()).sqrt()) for i, n in enumerate(lengths)]
    finally:
        voc.engine.set_arithmetic(orig)
        voc.engine.close()
    return out


def main():
    from emojivoice_amd._lib import Engine

    eng = Engine(0, spk_emb_dim=64)
    tab = conv_rows_table(eng)
    eng.close()
    print("# conv 128 -> 128, k = 3, one utterance of 140000 frames, blocks of 16 frames alternating between scale 1 and scale 2^-k;")
    print("# error per output row = max over channels |y - fp64| / (that row's fp64 RMS); quiet rows = rows 1..14 of a quiet block")
    print("# k | arithmetic 16: worst quiet, rms quiet, worst loud | arithmetic 6: ... | arithmetic 0 (exact fp32 MFMA): ... | builds")
    for k, rec in tab.items():
        print(f"{k:3d} | " + " | ".join(f"{rec[a][0]:.2e} {rec[a][1]:.2e} {rec[a][2]:.2e}" for a in (16, 6, 0)) + " | cfg " + "/".join(str(rec[a][3]) for a in (16, 6, 0)))
    vt = vocoder_ragged_table()
    print("# HiFi-GAN V1 on a ragged batch (516 / 300 / 120 / 40 valid frames of 516, padded frames = mel_mean): RMS error over the valid samples")
    print("# relative to that utterance's fp64 RMS, per utterance")
    for a in (16, 6, 0):
        print(f"arithmetic {a:2d}: " + "  ".join(f"{v:.2e}" for v in vt[a]))


if __name__ == "__main__":
    main()
