#!/usr/bin/env python3
"""Randomised parity fuzz of ev_text_encoder / ev_denoise / ev_stft_magnitude against the CPU oracle (random batch sizes,
token counts incl. 1, ragged lengths, both speaker configurations)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS
from oracle import matcha_oracle as O

torch.set_num_threads(16)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for n_spks in (109, 1):
    sd = W.synthetic_matcha_state(178, n_spks)
    m = MatchaTTS(sd, device="cuda:0")
    for case in range(8):
        B = int(rng.integers(1, 7)); Tx = int(rng.choice([1, 2, 3, 7, 33, 64, 65, 130, 257, 300]))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        ids = torch.randint(0, 178, (B, Tx), generator=g)
        lens = torch.randint(1, Tx + 1, (B,), generator=g); lens[int(rng.integers(B))] = Tx
        sid = torch.randint(0, max(n_spks, 1), (B,), generator=g)
        spk_cpu = torch.nn.functional.embedding(sid, sd["spk_emb.weight"]) if n_spks > 1 else None
        mu, logw = m.engine.text_encoder(ids.cuda(), lens.cuda(), spk_cpu.cuda() if spk_cpu is not None else None)
        rmu, rlogw, _ = O.text_encoder(sd, ids, lens, spk_cpu)
        e1, e2 = float((mu.cpu() - rmu).abs().max()), float((logw.cpu() - rlogw).abs().max())
        worst = max(worst, e1, e2)
        if e1 > 1e-4 or e2 > 1e-4:
            print(f"MISMATCH n_spks={n_spks} B={B} Tx={Tx} lens={lens.tolist()} mu {e1:.2e} logw {e2:.2e}"); sys.exit(1)
    m.engine.close()
print(f"text encoder: 16 cases ok, worst abs err {worst:.2e}")
voc = Generator(AttrDict(v1)).to("cuda:0"); voc.load_state_dict(W.synthetic_hifigan_state()); voc._sync_engine()
worst = 0.0
for case in range(8):
    B = int(rng.integers(1, 5)); L = 256 * int(rng.choice([4, 5, 8, 31, 100, 516]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    a = (torch.randn(B, L, generator=g) * 0.3).clamp(-1, 1); bias = torch.rand(513, generator=g)
    ref = O.denoiser(a, bias[None, :, None], strength=0.003)
    got = voc.engine.denoise(a.cuda(), bias.cuda(), 0.003).cpu()
    e = float((got - ref).abs().max()); worst = max(worst, e)
    if e > 3e-5:
        print(f"MISMATCH denoise B={B} L={L} err {e:.2e}"); sys.exit(1)
print(f"denoiser: 8 cases ok, worst abs err {worst:.2e}")
