#!/usr/bin/env python3
"""Per-shape conv timing of one config-2 step (HIP events around every conv launch; EV_PROFILE_DUMP table).

    python tools/shape_profile.py [B] [out_file]
"""
import os
import sys

out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/shape_profile.txt"
os.environ["EV_PROFILE_DUMP"] = out
if os.path.exists(out):
    os.remove(out)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 516
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
voc = Generator(AttrDict(v1)).to(dev)
voc.load_state_dict(W.synthetic_hifigan_state())
m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
mel = (torch.randn(B, 80, T, generator=g) * 2 - 5).to(dev)
mu = torch.randn(B, 80, T, generator=g).to(dev)
z = torch.randn(B, 80, T, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32).to(dev)
spk = m._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]
for it in range(2):
    if it == 1:
        voc._sync_engine()
        voc.engine.profile_enable(True)
        m.engine.profile_enable(True)
    if not os.environ.get("EV_SP_NOVOC"):
        wav = voc(mel)
    dec = m.engine.cfm_decode(mu, lengths, spk, z, 10)
    torch.cuda.synchronize()
print("hifigan", voc.engine.profile_read())
print("cfm", m.engine.profile_read())
print("sk_stats (launches, arrivals, timed-out waits)", m.engine.sk_stats())
print(open(out).read())
