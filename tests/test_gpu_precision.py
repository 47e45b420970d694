"""Hardening of the block-scaled fp16 datapath (arithmetic 16) on a real MI355X, through the C ABI:

  * rows of very different magnitude INSIDE one 128-row tile: error per output row relative to that row's own
    fp64 magnitude, side by side for arithmetic 16 / 6 / 0 (tools/dynamic_range.py prints the full table);
  * a ragged batch padded with `mel_mean` through the whole vocoder;
  * an Inf / a NaN in one utterance must not touch the utterance that shares its tile;
  * 64-channel chunks of very different magnitude on the balanced build (the accumulators' unit changes by a factor of up to 2^80);
  * weights that are not the exact sum of three bf16 pieces (1e-39, 1e-35): the checkpoint loads and the layer runs on the fp32 build;
  * a whole ResBlock1 in one launch (resblock_chain_h16_kernel) against the three fused pairs it replaces;
  * the self-attention on the fp16 pipe (attn_out_h16_kernel): keys / values of very different magnitude, peaked and flat softmaxes, and the whole
    decoder with that path on and off.
"""
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def eng():
    from emojivoice_amd._lib import Engine

    e = Engine(0, spk_emb_dim=64)
    yield e
    e.close()


def test_within_tile_dynamic_range_per_row(eng):
    """north_star's tolerance is 1e-4 (mel L-inf); a row 2^-20 below its tile's maximum must still meet it RELATIVE TO ITS OWN MAGNITUDE.
    By the error bound of DESIGN section 3 such a row keeps ~2^-17 per element; measured here per row (max over channels / row RMS)."""
    tab = _tool("dynamic_range").conv_rows_table(eng, ks=(0, 8, 16, 20))
    for k, rec in tab.items():
        q16, r16, l16, cfg16 = rec[16]
        q0, r0, l0, cfg0 = rec[0]
        assert cfg16 == 46 and cfg0 not in (40, 41, 46, 47, 60, 66), (cfg16, cfg0)
        assert l16 <= 2e-5 and l0 <= 2e-5, (k, l16, l0)                    # loud rows: fp32-grade under every setting
        assert q0 <= 2e-5, (k, q0)                                          # exact fp32: every row at fp32 grade whatever its neighbours
        assert q16 <= 1e-4, f"rows 2^-{k} below their tile's maximum: worst row-relative error {q16:.2e} exceeds the 1e-4 tolerance"
        if k <= 8:
            assert q16 <= 2e-5, (k, q16)                                    # within 2^-8 of the tile maximum: indistinguishable from fp32


def test_vocoder_on_a_mel_mean_padded_ragged_batch():
    vt = _tool("dynamic_range").vocoder_ragged_table()
    for a in (16, 6, 0):
        assert max(vt[a]) <= 1e-4, (a, vt[a])                               # every utterance, the 40-frame one beside 476 padded frames included
    assert max(vt[16]) <= 3 * max(vt[0]) + 1e-6, (vt[16], vt[0])            # ... and not worse than the fp32 MFMA chain by more than rounding noise


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_non_finite_value_stays_in_its_own_rows(eng, bad):
    """Utterance 0 ends in a non-finite value; utterance 1 — whose first rows share a 128-row tile with utterance 0's last rows — holds
    finite values above 65504.  With the tile's scale taken from a non-finite maximum (scale 1) those would overflow in fp16 and the
    small ones would lose their second piece; the finite-only repeat of the search keeps utterance 1 at fp32 grade."""
    g = torch.Generator().manual_seed(9)
    B, C, T, K = 3, 128, 50000, 3
    w = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, C, T, generator=g)
    x[1] *= 1e5                                             # finite, far above fp16's largest value at scale 1
    x[0, 5, T - 1] = bad
    ref = F.conv1d(x[1:2].double(), w.double(), b.double(), padding=1)
    orig = eng.arithmetic()
    try:
        eng.set_arithmetic(16)
        y = eng.op_conv1d(x.cuda(), w, b, dilation=1, padding=1).cpu().double()
        assert eng.last_cfg() == 46
    finally:
        eng.set_arithmetic(orig)
    e = (y[1:2] - ref).abs() / ref.pow(2).mean().sqrt()
    assert bool(torch.isfinite(y[1]).all()) and float(e.max()) <= 2e-5, float(e.max())
    assert bool(torch.isfinite(y[0, :, : T - 2]).all())     # utterance 0 itself: only the rows whose taps touch the bad element are lost
    assert not bool(torch.isfinite(y[0, :, T - 1]).all())


def test_chunks_of_very_different_magnitude_on_the_balanced_build(eng):
    """conv_h16_bal_kernel takes one scale per 64-channel chunk and moves its running sums from one chunk's unit to the next by an exact
    power of two: channels 0..63 at 1e15, 64..127 at 1e-15, the rest at unit scale — finite, and right against fp64 at the output's scale."""
    g = torch.Generator().manual_seed(10)
    B, C, T, K = 2, 256, 16600, 3
    w = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    b = torch.randn(C, generator=g)
    x = torch.randn(B, C, T, generator=g)
    for order in ((1e15, 1e-15), (1e-15, 1e15)):
        xs = x.clone()
        xs[:, :64] *= order[0]
        xs[:, 64:128] *= order[1]
        ref = F.conv1d(xs.double(), w.double(), b.double(), padding=1)
        orig = eng.arithmetic()
        try:
            eng.set_arithmetic(16)
            y = eng.op_conv1d(xs.cuda(), w, b, dilation=1, padding=1).cpu().double()
            assert eng.last_cfg() == 66, eng.last_cfg()
        finally:
            eng.set_arithmetic(orig)
        assert bool(torch.isfinite(y).all())
        assert float((y - ref).abs().max() / ref.pow(2).mean().sqrt()) <= 2e-5


def test_checkpoint_with_tiny_weights_loads_and_runs_on_the_fp32_build():
    """A weight of 1e-39 (subnormal) or 1e-35 is not the exact sum of three truncated bf16 pieces; the loader used to refuse the whole
    checkpoint.  It now leaves that layer without piece planes: the layer runs on the exact fp32 MFMA build."""
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from oracle import matcha_oracle as O

    sd = {k: v.clone() for k, v in W.synthetic_hifigan_state().items()}
    sd["resblocks.0.convs1.0.weight"][3, 4, 1] = 1e-39
    sd["resblocks.4.convs2.1.weight"][0, 0, 0] = 1e-35
    voc = Generator(AttrDict(v1)).to("cuda:0")
    voc.load_state_dict(sd)
    mel = torch.randn(2, 80, 40, generator=torch.Generator().manual_seed(12)) * 2 - 5
    wav = voc(mel.cuda()).cpu()
    ref = O.hifigan_forward(sd, mel, W.HIFIGAN_V1)
    assert float((wav - ref).pow(2).mean().sqrt()) <= 1e-4
    voc.engine.close()


def test_tile_maximum_in_a_neighbour_tiles_halo(eng):
    """The fp16 builds take a tile's scale from the producers' per-granule bounds (amax slots) of EVERY granule the tile stages, halo rows
    included.  Here the largest value by far sits in the LAST row of one 128-row granule, i.e. in the halo of the
    next tile only: a scale taken from that tile's own granule would be 2^24 too large and the halo row would overflow in fp16."""
    g = torch.Generator().manual_seed(14)
    B, C, T, K, d = 1, 128, 140000, 7, 3                    # halo 9 on either side; op geometry pads every utterance with 32 rows
    w = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    b = torch.zeros(C)
    x = torch.randn(B, C, T, generator=g) * 1e-3
    t_spike = 127 - 32 + 128 * 40                           # flattened row 128 * 40 + 127: last row of granule 40
    x[0, :, t_spike] = 3e4
    ref = F.conv1d(x.double(), w.double(), b.double(), padding=d * (K - 1) // 2, dilation=d)[0]
    near = torch.zeros(T, dtype=torch.bool)
    near[t_spike - 9: t_spike + 10] = True
    orig = eng.arithmetic()
    try:
        eng.set_arithmetic(16)
        for amax in (True, False):
            eng.set_amax(amax)
            y = eng.op_conv1d(x.cuda(), w, b, dilation=d, padding=d * (K - 1) // 2).cpu().double()[0]
            assert eng.last_cfg() == 46
            assert bool(torch.isfinite(y).all()), f"amax={amax}: overflow"
            e = (y - ref).abs()
            assert float(e[:, near].max() / ref[:, near].abs().max()) <= 1e-5, (amax, float(e[:, near].max()))
            far = ~near
            far[128 * 39 - 32: 128 * 42] = False            # the three tiles whose scale the spike sets (their quiet rows: see the dynamic-range test)
            assert float(e[:, far].max() / ref[:, far].pow(2).mean().sqrt()) <= 2e-5, (amax, float(e[:, far].max()))
    finally:
        eng.set_amax(True)
        eng.set_arithmetic(orig)


def test_vocoder_with_and_without_amax_slots_agree():
    """ev_hifigan end to end, 16 x 516 frames (deep grids: conv_h16_kernel on every level-1 / level-2 layer): the chain whose tile scales
    come from the producers' bounds against the same chain with every tile pre-scanning its input, and both against the oracle.  The
    first half of every other utterance is near-silent (mel at -11.5): tiles straddle silence and speech."""
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from oracle import matcha_oracle as O

    voc_sd = W.synthetic_hifigan_state()
    voc = Generator(AttrDict(v1)).to("cuda:0")
    voc.load_state_dict(voc_sd)
    g = torch.Generator().manual_seed(15)
    mel = torch.randn(16, 80, 516, generator=g) * 2.0 - 5.0
    mel[::2, :, :258] = -11.5
    voc._sync_engine()
    voc.engine.set_amax(True)
    y_on = voc(mel.cuda()).cpu()
    voc.engine.set_amax(False)
    y_off = voc(mel.cuda()).cpu()
    voc.engine.set_amax(True)
    ref = O.hifigan_forward(voc_sd, mel[:3], W.HIFIGAN_V1)
    assert float((y_on - y_off).pow(2).mean().sqrt()) <= 1e-5
    for y in (y_on, y_off):
        assert float((y[:3] - ref).pow(2).mean().sqrt()) <= 1e-4
    voc.engine.close()


def _attn_ref64(qkv, lengths, w_out, b_out, hid):
    B, T, _ = qkv.shape
    q, k, v = (qkv[..., i * 128:(i + 1) * 128].double().view(B, T, 2, 64).transpose(1, 2) for i in range(3))
    mask = (torch.arange(T)[None] < lengths[:, None]).double()
    s = q @ k.transpose(-1, -2) / 8.0 + mask.repeat_interleave(2, dim=0).view(B, 2, 1, T)      # FLOAT mask, added (transformer.py:266-271)
    att = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, 128)
    return hid.double() + att @ w_out.double().T + b_out.double()


@pytest.mark.parametrize("case", ["plain", "quiet_keys", "quiet_values", "peaked", "large"])
def test_attention_on_the_fp16_pipe_vs_fp64(eng, case):
    """attn_out_h16_kernel against fp64, beside the fp32-MFMA kernel on the same data.  q / k / v are stored as two fp16 pieces times ONE power of
    two per tensor, so what matters is data far below the tensor's maximum: half of the keys (values) 2^-12 below the others, softmaxes peaked
    on one key (scores +-60), and magnitudes near the top of the scaled range.  Error is taken relative to the RMS of the attention term."""
    g = torch.Generator().manual_seed(len(case))
    B, T = 4, 132
    qkv = torch.randn(B, T, 384, generator=g)
    if case == "quiet_keys":
        qkv[:, ::2, 128:256] *= 2.0 ** -12
    elif case == "quiet_values":
        qkv[:, ::2, 256:384] *= 2.0 ** -12
    elif case == "peaked":
        qkv[..., :256] *= 5.0
    elif case == "large":
        qkv *= 300.0
        qkv[..., :128] *= 1e-3                                              # (scores stay moderate)
    hid = torch.randn(B, T, 256, generator=g)
    w_out = torch.randn(256, 128, generator=g) / 128 ** 0.5
    b_out = torch.randn(256, generator=g) * 0.1
    lengths = torch.tensor([132, 100, 1, 77])
    ref = _attn_ref64(qkv, lengths, w_out, b_out, hid)
    term = (ref - hid.double() - b_out.double())
    valid = (torch.arange(T)[None] < lengths[:, None])
    scale = float(term[valid].pow(2).mean().sqrt())
    errs = {}
    for on in (True, False):
        eng.set_attn_h16(on)
        got = eng.op_attn_out(qkv.cuda(), lengths.cuda(), w_out, b_out, hid.cuda()).cpu().double()
        errs[on] = float((got - ref)[valid].abs().max()) / scale
    eng.set_attn_h16(True)
    # fp32-grade: within 2e-5 of the attention term's RMS — or, where scores of +-300 make the fp32 chain itself coarser than that ("large":
    # fp32 rounding of a score near 300 is 3e-5, and it sits in an exponent), no worse than the fp32-MFMA kernel on the same data
    assert errs[True] <= max(2e-5, 1.5 * errs[False]), (case, errs)
    assert errs[True] <= 4 * errs[False] + 2e-6, (case, errs)              # the class of the fp32 chain, not merely inside the tolerance


def test_decoder_with_attention_on_the_fp16_and_fp32_pipes_agree():
    """The whole CFM decode (batch 64 x 260 frames: the full-resolution transformer blocks have the 256 row tiles of 64 that put LayerNorm + QKV
    on ln_qkv_h16_kernel, hence the attention on attn_out_h16_kernel, with a short last query tile of 4) against the same decode with the
    fp32-MFMA attention: they differ by rounding only — far inside the 1e-4 mel tolerance — and the fp16 path is deterministic."""
    from emojivoice_amd import weights as W
    from emojivoice_amd.matcha_tts import MatchaTTS

    model = MatchaTTS(W.synthetic_matcha_state(), device="cuda:0")
    g = torch.Generator().manual_seed(3)
    B, T = 64, 260
    mu = torch.randn(B, 80, T, generator=g).cuda()
    lengths = torch.tensor([260, 200, 17, 260] * 16).cuda()
    spk = torch.randn(B, model.spk_emb_dim, generator=g).cuda()
    z = torch.randn(B, 80, T, generator=g).cuda()
    outs = {True: [], False: []}
    for on in (True, False, True):
        model.engine.set_attn_h16(on)
        outs[on].append(model.decode(mu, lengths, 2, spk=spk, z=z)[1].cpu())
    model.engine.set_attn_h16(True)
    assert torch.equal(outs[True][0], outs[True][1])
    diff = float((outs[True][0] - outs[False][0]).abs().max())
    assert 0.0 < diff <= 5e-5, diff                                        # (0 would mean the switch did nothing)
    model.engine.close()


@pytest.mark.parametrize("B,T", [(16, 516), (1, 132), (3, 40)])
def test_vocoder_with_and_without_resblock_chains_agree(B, T):
    """ev_hifigan with the k = 3 ResBlocks of the 64- and 32-channel levels as ONE launch each (the running x stays in registers between the three
    (dilated conv, conv) pairs; tiles of 128 / 256 frames of which 12 a side are computed but not stored) against the same call with three fused
    pairs per ResBlock, and both against the oracle.  Batch 1 and a 40-frame batch take the small-launch schedule (three streams); utterance
    starts and ends fall inside tiles (T * 64 and T * 256 frames are no multiples of the stored windows of 104 / 232)."""
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from oracle import matcha_oracle as O

    voc_sd = W.synthetic_hifigan_state()
    voc = Generator(AttrDict(v1)).to("cuda:0")
    voc.load_state_dict(voc_sd)
    g = torch.Generator().manual_seed(B * 1000 + T)
    mel = torch.randn(B, 80, T, generator=g) * 2.0 - 5.0
    mel[B - 1, :, : T // 2] = -11.5        # near-silence beside speech inside one tile: with uniform data every tile of either form takes the same
                                           # power-of-two scale and the two forms agree bit for bit — the switch would show no effect
    voc._sync_engine()
    outs = {}
    for on in (True, False, True):
        voc.engine.set_chain(on)
        y = voc(mel.cuda()).cpu()
        if on and True in outs:
            assert torch.equal(outs[True], y)                               # deterministic
        outs[on] = y
    # the switch is real: two chains (levels of 64 and 32 channels, k = 3) replace 2 x 3 pair launches.  (The OUTPUTS may well agree bit for bit:
    # a power-of-two tile scale commutes with the rounding of the pieces while both stay normal fp16 numbers, so the two forms differ only
    # where a tile's quiet rows push a low piece into the subnormals.)
    counts = {}
    for on in (True, False):
        voc.engine.set_chain(on)
        voc.engine.profile_enable(True)
        voc(mel.cuda())
        counts[on] = voc.engine.profile_read()[2]
        voc.engine.profile_enable(False)
    voc.engine.set_chain(True)
    assert counts[False] - counts[True] == 4, counts
    rms = float((outs[True] - outs[False]).pow(2).mean().sqrt())
    assert rms <= 1e-5, rms
    nref = min(B, 2)
    ref = O.hifigan_forward(voc_sd, mel[:nref], W.HIFIGAN_V1)
    for y in outs.values():
        assert float((y[:nref] - ref).pow(2).mean().sqrt()) <= 1e-4
    voc.engine.close()
