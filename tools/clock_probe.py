#!/usr/bin/env python3
"""Sample sclk / power (rocm-smi) while the vocoder runs back to back: is the fp32-MFMA roof clock-throttled?"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1

dev = torch.device("cuda", 0)
voc = Generator(AttrDict(v1)).to(dev); voc.load_state_dict(W.synthetic_hifigan_state())
mel = (torch.randn(64, 80, 516) * 2 - 5).to(dev)
voc(mel); torch.cuda.synchronize()
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in o.splitlines() if ("sclk" in l or "mclk" in l or "Power" in l or "fclk" in l)]
            print(time.strftime("%H:%M:%S"), " | ".join(keep), flush=True)
        except Exception as e:  # noqa
            print("smi failed", e, flush=True)
        time.sleep(1.0)
print("idle:"); 
th = threading.Thread(target=sampler); th.start(); time.sleep(2.5)
print("busy:", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 12:
    voc(mel); n += 1
torch.cuda.synchronize()
print(f"{n} calls, {(time.time() - t0) / n * 1e3:.1f} ms each")
stop = True; th.join()
