#!/usr/bin/env python3
"""One-off robustness check: a 24-second utterance (Tp = 2048 frames) through CFM (2 Euler steps) + HiFi-GAN vs the CPU oracle."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS
from oracle import matcha_oracle as O

torch.set_num_threads(16)
sd = W.synthetic_matcha_state(); vsd = W.synthetic_hifigan_state()
dev = torch.device("cuda", 0)
m = MatchaTTS(sd, device=dev)
voc = Generator(AttrDict(v1)).to(dev); voc.load_state_dict(vsd)
g = torch.Generator().manual_seed(3)
NS = int(os.environ.get("NS", "2"))
B, Tp = 2, 2048
mu = torch.randn(B, 80, Tp, generator=g); z = torch.randn(B, 80, Tp, generator=g) * 0.667
lengths = torch.tensor([2048, 1777])
sid = torch.tensor([5, 9])
spk = torch.nn.functional.embedding(sid, sd["spk_emb.weight"])
mel = m.engine.cfm_decode(mu.to(dev), lengths.to(dev), spk.to(dev), z.to(dev), NS, m.mel_std, m.mel_mean)
wav = voc(mel)
torch.cuda.synchronize()
t0 = time.time()
mask = O.sequence_mask(lengths, Tp).unsqueeze(1).float()
ref = O.solve_euler(sd, z, mu, mask, NS, spk)
ref_mel = O.denormalize(ref, sd["mel_mean"], sd["mel_std"])
ref_wav = O.hifigan_forward(vsd, mel.cpu(), dict(v1))
print(f"oracle {time.time() - t0:.1f} s")
print("mel Linf", float((mel.cpu() - ref_mel).abs().max()), "wav rms", float((wav.cpu() - ref_wav).pow(2).mean().sqrt()), "wav shape", tuple(wav.shape))
