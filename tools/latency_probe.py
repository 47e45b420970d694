import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS
dev = torch.device("cuda", 0)
model = MatchaTTS(W.synthetic_matcha_state(), device=dev)
voc = Generator(AttrDict(v1)).to(dev); voc.load_state_dict(W.synthetic_hifigan_state())
def run(T, tag):
    Tp = (T + 3)//4*4
    mu = torch.randn(1, 80, Tp).to(dev); z = torch.randn(1, 80, Tp).to(dev)
    spk = model._sd["spk_emb.weight"][torch.tensor([12], device=dev)]
    L = torch.tensor([T], device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec = model.engine.cfm_decode(mu, L, spk, z, 10); torch.cuda.synchronize(); t1 = time.perf_counter()
    wav = voc(dec); torch.cuda.synchronize(); t2 = time.perf_counter()
    x = torch.randint(1,178,(1, T//4)).to(dev)
    model.encode(x, torch.tensor([T//4], device=dev), spk); torch.cuda.synchronize(); t3 = time.perf_counter()
    model.encoder(x, torch.tensor([T//4], device=dev), spk); torch.cuda.synchronize(); t3b = time.perf_counter()
    zz = model.draw_noise(1, Tp); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"{tag} T={T}: cfm {1e3*(t1-t0):.1f} ms, hifigan {1e3*(t2-t1):.1f} ms, encoder device {1e3*(t3-t2):.1f} ms / host {1e3*(t3b-t3):.1f} ms, noise {1e3*(t4-t3b):.1f} ms")
for T in (516, 516, 516, 300, 300, 700, 700, 516, 100, 860, 860):
    run(T, "run")
