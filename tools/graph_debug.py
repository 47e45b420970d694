"""Bisect of the decode-graph NaN (round 4): which intervening call breaks the second replay of the largest graph?"""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.matcha_tts import MatchaTTS
DEV = "cuda:0"
sd = W.synthetic_matcha_state()
eager = MatchaTTS(sd, device=DEV); eager.warmup(max_frames=400, max_tokens=300)
g = torch.Generator().manual_seed(21)
cases = {}
for Tp in (64, 200, 132, 396):
    mu = torch.randn(1, 80, Tp, generator=g).to(DEV); z = torch.randn(1, 80, Tp, generator=g).to(DEV)
    lengths = torch.tensor([Tp - 3], device=DEV); spk = eager._sd["spk_emb.weight"][torch.tensor([Tp % 109], device=DEV)]
    cases[Tp] = (mu, z, lengths, spk, eager.decode(mu, lengths, 10, 0.667, spk, z=z))

def trial(name, order, eager_after=None, eager_steps=4, eager_T=64):
    model = MatchaTTS(sd, device=DEV); model.warmup(max_frames=400, max_tokens=300)
    dg = model.enable_decode_graphs()
    bad = []
    for i, Tp in enumerate(order):
        mu, z, lengths, spk, ref = cases[Tp]
        dec, mel = model.decode(mu, lengths, 10, 0.667, spk, z=z)
        torch.cuda.synchronize()
        if not (torch.equal(dec, ref[0]) and torch.equal(mel, ref[1])):
            bad.append((i, Tp, bool(torch.isnan(dec).any()), float((dec - ref[0]).abs().nan_to_num(1e9).max())))
        if eager_after is not None and i == eager_after:
            keep, model.decode_graphs = model.decode_graphs, None
            c = cases[eager_T]
            model.decode(c[0], c[2], eager_steps, 0.667, c[3], z=c[1]); torch.cuda.synchronize()
            model.decode_graphs = keep
    print(f"{name:60s} bad={bad}", flush=True)
    model.engine.close()

trial("396 twice, nothing else", [396, 396, 396])
trial("396, 64, 396", [396, 64, 396])
trial("64, 396, 64, 396", [64, 396, 64, 396])
trial("396, eager 64/4 steps, 396", [396, 396], eager_after=0)
trial("396, eager 64/10 steps, 396", [396, 396], eager_after=0, eager_steps=10)
trial("396, eager 396/4 steps, 396", [396, 396], eager_after=0, eager_T=396)
trial("full order of the test", [64, 200, 64, 132, 396, 200, 132, 64, 396], eager_after=4)
trial("full order, no eager", [64, 200, 64, 132, 396, 200, 132, 64, 396])
trial("200, 396, 200, 396", [200, 396, 200, 396])

print("---- second batch of probes", flush=True)
def trial2(name, steps):
    """steps: list of ('g', Tp) graph decode / ('e', Tp, nsteps) eager decode on the same handle / ('c', Tp) capture only"""
    model = MatchaTTS(sd, device=DEV); model.warmup(max_frames=400, max_tokens=300)
    dg = model.enable_decode_graphs()
    out = []
    for st in steps:
        kind, Tp = st[0], st[1]
        mu, z, lengths, spk, ref = cases[Tp]
        if kind == 'g':
            dec, mel = model.decode(mu, lengths, 10, 0.667, spk, z=z)
        elif kind == 'c':
            x0 = (z * 0.667).contiguous()
            dg._cache[((1, 80, Tp), 10)] = dg._capture(None, mu.contiguous(), lengths, spk, x0, 10); dg._order.append(((1, 80, Tp), 10))
            continue
        else:
            keep, model.decode_graphs = model.decode_graphs, None
            dec, mel = model.decode(mu, lengths, st[2], 0.667, spk, z=z)
            model.decode_graphs = keep
            if st[2] != 10:
                continue
        torch.cuda.synchronize()
        d = (dec - ref[0]).abs().nan_to_num(1e9)
        nb = int((d > 0).sum())
        rows = (d > 0).any(dim=1)[0].nonzero().flatten()
        out.append((kind, Tp, nb, (int(rows.min()), int(rows.max())) if nb else None))
    print(f"{name:50s} {out}", flush=True)
    model.engine.close()

trial2("g396, capture-only 64, g396", [('g', 396), ('c', 64), ('g', 396)])
trial2("g396, g64, g396 (where)", [('g', 396), ('g', 64), ('g', 396)])
trial2("g396, e64/4, e396/10, g396", [('g', 396), ('e', 64, 4), ('e', 396, 10), ('g', 396)])
trial2("g396, e64/4, g396, g396", [('g', 396), ('e', 64, 4), ('g', 396), ('g', 396)])
trial2("e396 first, g64, e396/10", [('e', 396, 10), ('g', 64), ('e', 396, 10)])
