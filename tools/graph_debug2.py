"""Second bisect of the decode-graph corruption: manual capture (no DecodeGraphs), engine-level calls only."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.matcha_tts import MatchaTTS
DEV = "cuda:0"
sd = W.synthetic_matcha_state()
g = torch.Generator().manual_seed(21)
def inputs(Tp):
    mu = torch.randn(1, 80, Tp, generator=g).to(DEV); z = torch.randn(1, 80, Tp, generator=g).to(DEV)
    return mu, z, torch.tensor([Tp - 3], device=DEV, dtype=torch.int32), None
I = {Tp: inputs(Tp) for Tp in (64, 396)}

def run(name, between):
    model = MatchaTTS(sd, device=DEV); model.warmup(max_frames=400, max_tokens=300)
    eng = model.engine
    spk = model._sd["spk_emb.weight"][torch.tensor([5], device=DEV)].contiguous()
    mu, z, ln, _ = I[396]
    ref = eng.cfm_decode(mu, ln, spk, z, 10).clone()
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = eng.cfm_decode(mu, ln, spk, z, 10)
    torch.cuda.current_stream().wait_stream(s)
    out.zero_(); gr.replay(); torch.cuda.synchronize()
    ok0 = torch.equal(out, ref)
    between(eng, spk)
    torch.cuda.synchronize()
    out.zero_(); gr.replay(); torch.cuda.synchronize()
    ok1 = torch.equal(out, ref)
    e = eng.cfm_decode(mu, ln, spk, z, 10); torch.cuda.synchronize()
    ok2 = torch.equal(e, ref)
    out.zero_(); gr.replay(); torch.cuda.synchronize()
    ok3 = torch.equal(out, ref)
    print(f"{name:44s} first replay {ok0}  after: replay {ok1}  eager {ok2}  replay again {ok3}  nan {bool(torch.isnan(out).any())}", flush=True)
    eng.close()

m64, z64, l64, _ = I[64]
def n_calls(n, keep):
    def f(e, s):
        outs = []
        for _ in range(n):
            o = e.cfm_decode(m64, l64, s, z64, 10)
            if keep: outs.append(o)
        return outs
    return f
for n in (1, 2, 3, 4, 5, 6):
    run(f"{n} eager calls 64/10, outputs kept", n_calls(n, True))
for n in (2, 4):
    run(f"{n} eager calls 64/10, outputs dropped", n_calls(n, False))
