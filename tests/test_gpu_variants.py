"""The kernel A/B switches (tile configuration, two-chunk staging, generic instead of lean epilogue, unfused resblock pairs,
fused LayerNorm + QKV / feed-forward kernels forced on at a small batch or switched off, split-key attention switched off,
the three ResBlock1 chains of an MRF level on one stream instead of three, the workspace zeroed whole instead of its pad rows,
the balanced persistent grids and the fused attention + projection kernel at batch 64, and EV_SPLIT=0: every conv on the fp32 MFMA
instead of the bf16-split builds)
must not change results: every fp32-MFMA build accumulates in the same (chunk, tap, k-group) order, so those variants agree with the
default path to fp32 rounding of the epilogue (bias added before vs after the K loop); the split builds form each product from six
exact bf16 products and differ from the fp32 MFMA by less than one rounding of the product.  The switches are read once per
process, hence one child process per variant."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys, torch
sys.path.insert(0, %r)
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(7)
B, T = 3, 44
if len(sys.argv) > 2 and sys.argv[2] == "b1": B, T = 1, 100
if len(sys.argv) > 2 and sys.argv[2] == "b64": B, T = 64, 260
if len(sys.argv) > 2 and sys.argv[2] == "b8": B, T = 8, 516
m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
voc = Generator(AttrDict(v1)).to(dev); voc.load_state_dict(W.synthetic_hifigan_state())
mu = torch.randn(B, 80, T, generator=g).to(dev); z = (torch.randn(B, 80, T, generator=g) * 0.667).to(dev)
lengths = torch.tensor([44, 31, 17] if B == 3 else ([T] if B == 1 else [T - (7 * i) %% 90 for i in range(B)]), dtype=torch.int32).to(dev)
spk = m._sd["spk_emb.weight"][(torch.arange(B, device=dev) * 4 + 1) %% 109]
mel = m.engine.cfm_decode(mu, lengths, spk, z, 4 if B < 8 else 2, m.mel_std, m.mel_mean)
wav = voc(mel if B < 64 else mel[:4])
torch.save({"mel": mel.cpu(), "wav": wav.cpu()}, sys.argv[1])
""" % REPO

VARIANTS = [{"EV_KB": "2"}, {"EV_NO_LEAN": "1"}, {"EV_FUSE_PAIRS": "0"}, {"EV_FORCE_CFG": "0"}, {"EV_FORCE_CFG": "5"}, {"EV_FORCE_CFG": "6"},
            {"EV_FORCE_CFG": "4"}, {"EV_FUSE_MLP_MIN": "1"}, {"EV_FUSE_MLP": "0"}, {"EV_NO_ATTN_SK": "1"},
            {"EV_MRF_STREAMS_MAX": "0"}, {"EV_FULL_REZERO": "1"}, {"EV_SPLIT": "0"}, {"EV_SPLIT": "6"}]


# the single-utterance builds: conv_sk32_kernel, per-tile GroupNorm statistics + groupnorm_apply_kernel, split-key attention
B1_VARIANTS = [{"EV_SPLIT": "0"}, {"EV_NO_GN_STATS": "1"}, {"EV_NO_SK32_LEAN": "1"}, {"EV_ATTN_TPW": "1"}, {"EV_ATTN_TPW": "3"}, {"EV_NO_ATTN_SK": "1"}, {"EV_NO_SK": "1"}]


def _run(tmp_path, name, extra, shape="b3"):
    out = tmp_path / f"{name}.pt"
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CHILD, str(out), shape], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch

    return torch.load(out)


def test_kernel_variants_agree(tmp_path):
    ref = _run(tmp_path, "default", {})
    assert float(ref["wav"].abs().max()) > 1e-3 and float(ref["mel"].abs().max()) > 1e-1
    for i, extra in enumerate(VARIANTS):
        got = _run(tmp_path, f"v{i}", extra)
        dmel = float((got["mel"] - ref["mel"]).abs().max())
        dwav = float((got["wav"] - ref["wav"]).abs().max())
        assert dmel <= 2e-5 and dwav <= 2e-5, (extra, dmel, dwav)


def test_batch1_variants_agree(tmp_path):
    """One utterance takes builds of its own (far fewer workgroups than CUs): they agree with the general ones.  The per-tile
    GroupNorm statistics are merged pairwise instead of summed in two passes: a few 1e-6 on the mel."""
    ref = _run(tmp_path, "b1_default", {}, "b1")
    assert ref["mel"].shape == (1, 80, 100)
    for i, extra in enumerate(B1_VARIANTS):
        got = _run(tmp_path, f"b1_v{i}", extra, "b1")
        dmel = float((got["mel"] - ref["mel"]).abs().max())
        dwav = float((got["wav"] - ref["wav"]).abs().max())
        assert dmel <= 2e-5 and dwav <= 5e-5, (extra, dmel, dwav)


# batch 64: the balanced persistent builds (SkCtl in ev_kernels.h: ln_mlp_kernel, conv_gemm_bal_kernel with equal and with weighted
# units) and the fused attention + projection kernel.  An owner that never waits (EV_SK_SPIN=0) recomputes every contributor's share
# as a separate partial sum: the SAME bits.  The one-tile-per-workgroup builds differ by the order of partial sums only.
def test_batch64_balanced_builds(tmp_path):
    ref = _run(tmp_path, "b64_default", {}, "b64")
    assert ref["mel"].shape == (64, 80, 260) and float(ref["mel"].abs().max()) > 1e-1
    import torch

    nowait = _run(tmp_path, "b64_nowait", {"EV_SK_SPIN": "0"}, "b64")
    assert torch.equal(nowait["mel"], ref["mel"]) and torch.equal(nowait["wav"], ref["wav"])
    # (run-to-run equality of the default build is covered in-process by tests/test_gpu_ops.py; every switch here costs a batch-64 child)
    for i, extra in enumerate([{"EV_NO_SK_BALANCE": "1"}, {"EV_CONV_BALANCE_W": "1", "EV_BAL5": "64"}, {"EV_FUSE_ATTN": "0", "EV_SK_WGS": "3"}, {"EV_SPLIT": "0"}, {"EV_SPLIT": "6"},
                               {"EV_NO_QKV_H16": "1"}]):     # (LayerNorm + QKV back on the fp32 MFMA build)
        got = _run(tmp_path, f"b64_v{i}", extra, "b64")
        dmel = float((got["mel"] - ref["mel"]).abs().max())
        dwav = float((got["wav"] - ref["wav"]).abs().max())
        assert dmel <= 2e-5 and dwav <= 5e-5, (extra, dmel, dwav)


def test_mid_batch_split_builds_agree_with_fp32_mfma(tmp_path):
    """Batch 8 x 516 frames: the vocoder's C = 256 level is a few rounds of workgroups (conv_split_bal_kernel), its C = 128 level a deep
    grid (conv_split_kernel), the C = 32 / 64 levels the fused split pairs; the U-Net keeps fp32-MFMA builds.  Against EV_SPLIT=0."""
    ref = _run(tmp_path, "b8_fp32", {"EV_SPLIT": "0"}, "b8")
    got = _run(tmp_path, "b8_split", {}, "b8")
    assert got["wav"].shape == (8, 1, 516 * 256) and float(ref["wav"].abs().max()) > 1e-3
    dmel = float((got["mel"] - ref["mel"]).abs().max())
    dwav = float((got["wav"] - ref["wav"]).abs().max())
    assert dmel <= 2e-5 and dwav <= 5e-5, (dmel, dwav)
