#!/usr/bin/env python3
"""Differential fuzz of the 16-bit-pipe builds against the fp32 MFMA builds on ONE handle (ev_set_arithmetic 16 / 6 vs 0):
random layer shapes on deep and shallow grids, odd lengths, prologue leaky-relu on / off, inputs scaled by 10^U(-5, 5) with a few
outliers, LayerNorm + QKV / LayerNorm + feed-forward rows, the fused attention on the fp16 pipe against the fp32-MFMA attention (ragged lengths, q / k / v of
random magnitude), and the whole vocoder with its k = 3 ResBlocks as one launch against three fused pairs.  Prints the worst error relative to the
output scale per build.
    python tools/fuzz_h16.py [n_cases] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd._lib import Engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
eng = Engine(0)
worst = {}
SPLIT = (40, 41, 60, 46, 47, 66)


def note(cfg, arith, y, y0, what):
    err = (y.double() - y0.double()).abs().max().item()
    scale = max(y0.double().abs().max().item(), 1e-30)
    rel = err / scale
    key = (arith, cfg)
    if rel > worst.get(key, (0.0, ""))[0]:
        worst[key] = (rel, what)
    if not (rel <= 2e-5):
        print(f"FAIL arith {arith} cfg {cfg}: rel {rel:.3e}  {what}", flush=True)
        return 1
    return 0


bad = 0
for case in range(n_cases):
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    kind = case % 6
    mag = 10.0 ** float(rng.uniform(-5, 5))
    if kind < 3:
        cin = int(rng.choice([64, 128, 192, 256, 512])); cout = int(rng.choice([64, 128, 192, 256, 384, 512]))
        K = int(rng.choice([1, 3, 7, 11])); dil = int(rng.choice([1, 3, 5])) if K > 1 else 1
        slope = float(rng.choice([-1.0, 0.1]))
        deep = kind != 2
        rows = (int(170000 * 128 / max(cout, 128)) if deep else 20000) + int(rng.integers(0, 999))
        B = int(rng.integers(1, 5)); T = rows // B
        x = torch.randn(B, cin, T, generator=g) * mag
        if case % 5 == 0:                                   # a few outliers far above the rest
            idx = torch.randint(0, x.numel(), (7,), generator=g)
            x.view(-1)[idx] *= 3000.0
        w = torch.randn(cout, cin, K, generator=g) / (cin * K) ** 0.5
        b = torch.randn(cout, generator=g) * mag
        pad = dil * (K - 1) // 2
        what = f"conv {cin}->{cout} k{K} d{dil} B{B} T{T} slope {slope} mag {mag:.1e}"
        eng.set_arithmetic(0)
        xc = x.cuda()
        y0 = eng.op_conv1d(xc, w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope).cpu()
        for arith in (16, 6):
            eng.set_arithmetic(arith)
            y = eng.op_conv1d(xc, w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope).cpu()
            cfg = eng.last_cfg()
            if cfg in SPLIT:
                bad += note(cfg, arith, y, y0, what)
            elif not torch.equal(y, y0):
                print(f"FAIL arith {arith}: fp32 build {cfg} differs from itself  {what}", flush=True); bad += 1
        del xc
    elif kind == 4:
        # attn_out_h16_kernel (cfg 31) against attn_out_kernel on the same fp32 q | k | v: scores of moderate size, values of any size
        B = int(rng.choice([1, 2, 3, 8, 17, 64])); T = int(rng.choice([32, 33, 67, 100, 132, 258, 260, 516])) if B < 17 else int(rng.choice([36, 98, 132]))
        qkv = torch.randn(B, T, 384, generator=g)
        sq = 10.0 ** float(rng.uniform(-2, 2))
        qkv[..., :128] *= sq * float(rng.uniform(0.5, 3.0)); qkv[..., 128:256] /= sq; qkv[..., 256:] *= mag
        if case % 3 == 0:
            qkv[:, ::3, 128:] *= 2.0 ** -10                 # keys / values far below their neighbours
        hid = torch.randn(B, T, 256, generator=g) * mag
        w_out = torch.randn(256, 128, generator=g) / 128 ** 0.5; b_out = torch.randn(256, generator=g) * 0.1 * mag
        lengths = torch.randint(1, T + 1, (B,), generator=g); lengths[0] = T
        eng.set_arithmetic(16)
        eng.set_attn_h16(False)
        y0 = eng.op_attn_out(qkv.cuda(), lengths.cuda(), w_out, b_out, hid.cuda()).cpu()
        eng.set_attn_h16(True)
        y = eng.op_attn_out(qkv.cuda(), lengths.cuda(), w_out, b_out, hid.cuda()).cpu()
        valid = (torch.arange(T)[None] < lengths[:, None])
        bad += note(31, 16, (y - hid)[valid], (y0 - hid)[valid], f"attention B{B} T{T} |q| x{sq:.1e} |v| x{mag:.1e}")
    elif kind == 5:
        # the vocoder with resblock_chain_h16_kernel (cfg 183) against the same call with three fused pairs per ResBlock
        if "voc" not in globals():
            from emojivoice_amd import weights as W
            from emojivoice_amd.hifigan import AttrDict, Generator, v1
            voc = Generator(AttrDict(v1)).to("cuda:0"); voc.load_state_dict(W.synthetic_hifigan_state()); voc._sync_engine()
        B = int(rng.choice([1, 2, 5])); T = int(rng.integers(8, 300))
        mel = torch.randn(B, 80, T, generator=g) * float(rng.uniform(0.5, 3.0)) - float(rng.uniform(2, 8))
        if case % 2:
            mel[B - 1, :, : T // 2] = -11.5
        voc.engine.set_chain(False); y0 = voc(mel.cuda()).cpu()
        voc.engine.set_chain(True); y = voc(mel.cuda()).cpu()
        bad += note(183, 16, y, y0, f"vocoder chains B{B} T{T}")
    else:
        rows = int(rng.choice([16384, 16640, 20011, 33280, 40000])) + int(rng.integers(0, 64))
        x = (torch.randn(rows, 256, generator=g) * float(rng.uniform(0.2, 3.0)) + float(rng.uniform(-1, 1))) * mag
        ln_g = (torch.rand(256, generator=g) + 0.5) * 10.0 ** float(rng.uniform(-3, 3)); ln_b = torch.randn(256, generator=g) * 0.1
        wq = torch.randn(384, 256, generator=g) / 16.0
        w1 = torch.randn(1024, 256, generator=g) / 16.0; b1 = torch.randn(1024, generator=g) * 0.1
        alpha, beta = torch.randn(1024, generator=g) * 0.3, torch.randn(1024, generator=g) * 0.3
        w2 = torch.randn(256, 1024, generator=g) / 32.0; b2 = torch.randn(256, generator=g) * 0.1
        mask = (torch.rand(rows, generator=g) > 0.2).float().cuda()
        xc, gc, bc = x.cuda(), ln_g.cuda(), ln_b.cuda()
        eng.set_arithmetic(0)
        q0 = eng.op_ln_mlp(xc, gc, bc, wq, None).cpu()
        f0 = eng.op_ln_mlp(xc, gc, bc, w1, b1, alpha, beta, w2, b2, mask).cpu()
        eng.set_arithmetic(16)
        q = eng.op_ln_mlp(xc, gc, bc, wq, None).cpu()
        bad += note(eng.last_cfg(), 16, q, q0, f"ln+qkv rows {rows} mag {mag:.1e}")
        f = eng.op_ln_mlp(xc, gc, bc, w1, b1, alpha, beta, w2, b2, mask).cpu()
        bad += note(121, 16, f, f0, f"ln+ff rows {rows} mag {mag:.1e}")
        del xc
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} failures", flush=True)
eng.close()
for (arith, cfg), (rel, what) in sorted(worst.items()):
    print(f"arith {arith:2d} cfg {cfg:3d}: worst {rel:.2e} of the output scale  ({what})")
print(f"{n_cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
