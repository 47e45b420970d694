#!/usr/bin/env python3
"""Randomised parity fuzz of ev_cfm_decode and ev_hifigan against the CPU oracle: random batch sizes, padded lengths (multiples
of 4 from 4 up), ragged utterance lengths incl. 1, few Euler steps."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS
from oracle import matcha_oracle as O

torch.set_num_threads(16)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(seed)
sd = W.synthetic_matcha_state(); vsd = W.synthetic_hifigan_state()
m = MatchaTTS(sd, device="cuda:0")
voc = Generator(AttrDict(v1)).to("cuda:0"); voc.load_state_dict(vsd)
worst_m = worst_w = 0.0
for case in range(ncase):
    B = int(rng.integers(1, 8)); Tp = 4 * int(rng.choice([1, 1, 2, 3, 5, 8, 16, 17, 33, 48]))
    steps = int(rng.integers(1, 4))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    lens = torch.randint(1, Tp + 1, (B,), generator=g)
    if rng.random() < 0.5:
        lens[int(rng.integers(B))] = Tp
    mask = O.sequence_mask(lens, Tp).unsqueeze(1).float()
    mu = torch.randn(B, 80, Tp, generator=g) * mask; z = torch.randn(B, 80, Tp, generator=g) * 0.667
    sid = torch.randint(0, 109, (B,), generator=g)
    spk = torch.nn.functional.embedding(sid, sd["spk_emb.weight"])
    ref = O.solve_euler(sd, z, mu, mask, steps, spk)
    dec = m.engine.cfm_decode(mu.cuda(), lens.cuda(), spk.cuda(), z.cuda(), steps)
    e = float((dec.cpu() - ref).abs().max()); worst_m = max(worst_m, e)
    if e > 5e-5:
        print(f"MISMATCH cfm B={B} Tp={Tp} lens={lens.tolist()} steps={steps} err {e:.2e}"); sys.exit(1)
    T = int(rng.choice([1, 2, 3, 5, 9, 20, 33]))
    mel = torch.randn(B, 80, T, generator=g) * 2 - 5
    rw = O.hifigan_forward(vsd, mel, dict(v1))
    gw = voc(mel.cuda()).cpu()
    e = float((gw - rw).pow(2).mean().sqrt()); worst_w = max(worst_w, e)
    if e > 1e-4 or not tuple(gw.shape) == tuple(rw.shape):
        print(f"MISMATCH hifigan B={B} T={T} rms {e:.2e}"); sys.exit(1)
print(f"seed {seed}: {ncase} cases ok, worst mel Linf {worst_m:.2e}, worst wav rms {worst_w:.2e}")
