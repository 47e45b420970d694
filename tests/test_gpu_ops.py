"""Operator-level parity on a real MI355X: each HIP kernel, called through the C ABI
(ev_op_*), against the same op of the CPU oracle / plain torch fp32 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from emojivoice_amd._lib import Engine

    e = Engine(0, spk_emb_dim=64)
    yield e
    e.close()


def _close(got, ref, rtol=3e-5, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= rtol * scale + 1e-6, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


CONV_CASES = [
    # B, Cin, T, Cout, K, dil, pre_slope
    (2, 80, 40, 512, 7, 1, -1.0),      # conv_pre
    (3, 256, 150, 256, 3, 1, 0.1),     # resblock k3 (BM=64 path: small grid)
    (2, 256, 700, 256, 3, 3, 0.1),
    (2, 128, 300, 128, 7, 5, 0.1),
    (2, 64, 333, 64, 11, 5, 0.1),      # BM=64
    (2, 32, 1000, 32, 11, 5, 0.1),     # BM=32, widest halo
    (2, 32, 515, 32, 3, 1, -1.0),
    (2, 32, 300, 1, 7, 1, 0.01),       # conv_post
    (2, 224, 64, 256, 3, 1, -1.0),     # estimator block conv, Cin not a multiple of 32
    (2, 224, 64, 256, 1, 1, -1.0),     # res_conv 1x1
    (1, 1024, 36, 256, 1, 1, -1.0),    # FF2-like linear
    (2, 256, 33, 80, 1, 1, -1.0),      # final_proj
]


@pytest.mark.parametrize("B,Cin,T,Cout,K,dil,slope", CONV_CASES)
def test_conv1d(eng, B, Cin, T, Cout, K, dil, slope):
    g = torch.Generator().manual_seed(B * 1000 + Cin + T + Cout + K)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K * dil - dil) // 2
    xin = F.leaky_relu(x, slope) if slope >= 0 else x
    ref = F.conv1d(xin, w, b, dilation=dil, padding=pad)
    got = eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope)
    _close(got, ref, what=f"conv {Cin}->{Cout} k{K} d{dil}")


def test_split_pieces_are_exact(eng):
    """Every fp32 operand of the split builds is cut into three bf16 pieces whose sum is the operand, bit for bit — the products the
    bf16 pipe forms are then exact products of exact pieces (values across the exponent range, signs, powers of two, zero)."""
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(4096, generator=g) * 10.0 ** torch.randint(-20, 20, (4096,), generator=g).float(),
                   torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -100, 3.0e38, -3.0e38, 1.0 + 2.0 ** -23, 0.1, 65504.0, 1e-30])])
    p = eng.op_split_pieces(x.cuda()).cpu()
    assert torch.equal((p[0].double() + p[1].double() + p[2].double()).float(), x), "pieces do not sum to the operand"
    assert torch.equal(p[0] + p[1] + p[2], x)                      # ... in fp32 arithmetic too (what the accumulator sees is exact)
    bits = p.view(torch.int32)
    assert int((bits & 0xFFFF).abs().max()) == 0, "a piece is not a bf16 value"
    assert float((p[1].abs() - p[0].abs() * 2.0 ** -7).clamp(min=0).max()) == 0 and float((p[2].abs() - p[0].abs() * 2.0 ** -14).clamp(min=0).max()) == 0


@pytest.mark.parametrize("B,Cin,T,Cout,K,dil,slope,cfg", [(5, 128, 30000, 128, 7, 3, 0.1, 40), (3, 256, 23000, 256, 11, 5, 0.1, 40), (2, 128, 40000, 256, 3, 1, -1.0, 40),
                                                          (16, 256, 600, 256, 3, 1, -1.0, 60), (40, 512, 258, 256, 1, 1, 0.1, 60)])
def test_conv1d_split_builds(eng, B, Cin, T, Cout, K, dil, slope, cfg):
    """conv_split_kernel (cfg 40: deep grids) and conv_split_bal_kernel (cfg 60: a few rounds of workgroups): fp32 convs whose products
    are formed on the bf16 matrix pipe — every fp32 operand is the exact sum of three bf16 pieces, the six products of weight <= 2 are
    accumulated in fp32 (what is left out is below one fp32 rounding of the product: tools/bf16_split_probe.hip) — and their fp16 forms
    conv_h16_kernel (46) / conv_h16_bal_kernel (66): two block-scaled fp16 pieces, three products (the default).  Same tolerance against
    torch's fp32 conv as the fp32-MFMA builds, and bit-reproducible."""
    g = torch.Generator().manual_seed(B + Cin + T)
    x = torch.randn(B, Cin, T, generator=g) * 1.5
    x[:, :, ::7] *= 1e-3                                   # operands of very different magnitude in one dot product
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = dil * (K - 1) // 2
    xin = F.leaky_relu(x, slope) if slope >= 0 else x
    ref = F.conv1d(xin, w, b, padding=pad, dilation=dil)
    orig = eng.arithmetic()
    try:
        for arith, want in ((6, cfg), (16, cfg + 6)):
            eng.set_arithmetic(arith)
            got = eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope)
            assert eng.last_cfg() == want, f"this shape was expected to take build {want}, took {eng.last_cfg()}"
            _close(got, ref, what=f"split conv {Cin}->{Cout} k{K} d{dil}, arithmetic {arith}")
            assert torch.equal(eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope), got)
    finally:
        eng.set_arithmetic(orig)


@pytest.mark.parametrize("B,Cin,T,Cout,K,slope", [(64, 256, 196, 256, 3, -1.0), (64, 256, 194, 256, 3, 0.1), (64, 512, 196, 256, 1, -1.0), (48, 256, 283, 256, 3, -1.0)])
def test_conv1d_balanced_grid(eng, B, Cin, T, Cout, K, slope, monkeypatch):
    """Dense conv launches of about one round of workgroups take conv_gemm_bal_kernel (a balanced persistent grid over (tile,
    k-chunk) units with in-launch hand-off of partial accumulator tiles, SkCtl in ev_kernels.h): against torch, against a handle
    whose owners never wait (EV_SK_SPIN=0: every hand-off takes the recompute path and must deliver the same bits), launch after
    launch, and against the one-tile-per-workgroup build."""
    from emojivoice_amd._lib import Engine

    g = torch.Generator().manual_seed(B + Cin + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K - 1) // 2
    xin = F.leaky_relu(x, slope) if slope >= 0 else x
    ref = F.conv1d(xin, w, b, padding=pad)
    s0 = eng.sk_stats()[0]
    got = eng.op_conv1d(x.cuda(), w, b, padding=pad, pre_lrelu_slope=slope)
    assert eng.sk_stats()[0] == s0 + 1, "this shape was expected to take the balanced build"
    _close(got, ref, what=f"balanced conv {Cin}->{Cout} k{K}")
    for _ in range(2):
        assert torch.equal(eng.op_conv1d(x.cuda(), w, b, padding=pad, pre_lrelu_slope=slope), got)
    monkeypatch.setenv("EV_SK_SPIN", "0")
    e2 = Engine(0)
    try:
        assert torch.equal(e2.op_conv1d(x.cuda(), w, b, padding=pad, pre_lrelu_slope=slope), got)
    finally:
        e2.close()
    monkeypatch.setenv("EV_NO_SK_BALANCE", "1")
    e3 = Engine(0)
    try:
        _close(e3.op_conv1d(x.cuda(), w, b, padding=pad, pre_lrelu_slope=slope), got, what="one tile per workgroup vs balanced")
        assert e3.sk_stats()[0] == 0
    finally:
        e3.close()


@pytest.mark.parametrize("B,Cin,T,Cout,K,s,p", [(2, 512, 37, 256, 16, 8, 4), (2, 256, 70, 128, 16, 8, 4), (2, 128, 130, 64, 4, 2, 1),
                                                (3, 64, 257, 32, 4, 2, 1), (2, 256, 66, 256, 4, 2, 1)])
def test_conv_transpose1d(eng, B, Cin, T, Cout, K, s, p):
    g = torch.Generator().manual_seed(Cin + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / s) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=p)
    got = eng.op_conv1d(x.cuda(), w, b, transposed=True, stride=s, padding=p, pre_lrelu_slope=0.1)
    _close(got, ref, what=f"convT {Cin}->{Cout} k{K} s{s}")


def test_arithmetic_switch_in_process(eng):
    """ev_set_arithmetic: the same handle, the same launch geometry, products on the bf16 pipe (6) or on the fp32 MFMA (0); the two agree to
    fp32 rounding and each is bit-reproducible."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 128, 33000, generator=g)
    w = torch.randn(128, 128, 7, generator=g) / (128 * 7) ** 0.5
    b = torch.randn(128, generator=g)
    orig = eng.arithmetic()
    eng.set_arithmetic(6)
    y6 = eng.op_conv1d(x.cuda(), w, b, dilation=3, padding=9, pre_lrelu_slope=0.1)
    assert eng.last_cfg() == 40
    try:
        eng.set_arithmetic(16)
        y16 = eng.op_conv1d(x.cuda(), w, b, dilation=3, padding=9, pre_lrelu_slope=0.1)
        assert eng.last_cfg() == 46
        assert torch.equal(eng.op_conv1d(x.cuda(), w, b, dilation=3, padding=9, pre_lrelu_slope=0.1), y16)
        eng.set_arithmetic(0)
        y0 = eng.op_conv1d(x.cuda(), w, b, dilation=3, padding=9, pre_lrelu_slope=0.1)
        assert eng.last_cfg() not in (40, 41, 60)
        assert torch.equal(eng.op_conv1d(x.cuda(), w, b, dilation=3, padding=9, pre_lrelu_slope=0.1), y0)
    finally:
        eng.set_arithmetic(orig)
    _close(y6, y0, rtol=1e-5, what="bf16-split products vs fp32 MFMA")
    _close(y16, y0, rtol=1e-5, what="fp16 block-scaled products vs fp32 MFMA")
    with pytest.raises(Exception):
        eng.set_arithmetic(4)


def test_h16_block_scaling_is_scale_invariant(eng):
    """The fp16 form (arithmetic 16) has fp16's range only through its power-of-two block scales: a deep conv layer against fp64 with
    activations that are tiny, huge, and of very different scale from utterance to utterance and from channel to channel — the error,
    relative to each utterance's output RMS, stays at the level of the fp32 FMA chain in every case (tools/arith_accuracy.py prints the
    same table for all settings)."""
    g = torch.Generator().manual_seed(5)
    B, C, T, K, d = 3, 128, 50000, 7, 3
    w = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    base = torch.randn(B, C, T, generator=g)
    cases = {"unit": base, "tiny": base * 1e-4, "huge": base * 3e3, "beyond fp16": base * 1e7, "rows": base * torch.tensor([1e-3, 1.0, 1e3]).view(3, 1, 1),
             "channels": base * (10.0 ** torch.linspace(-3, 3, C)).view(1, C, 1)}
    orig = eng.arithmetic()
    try:
        eng.set_arithmetic(16)
        for name, x in cases.items():
            ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), padding=d * (K - 1) // 2, dilation=d)
            y = eng.op_conv1d(x.cuda(), w, b, dilation=d, padding=d * (K - 1) // 2, pre_lrelu_slope=0.1).cpu().double()
            assert eng.last_cfg() == 46
            e = (y - ref).abs() / ref.pow(2).mean(dim=(1, 2), keepdim=True).sqrt()
            assert float(e.max()) <= 2e-5 and float(e.pow(2).mean().sqrt()) <= 1.5e-6, (name, float(e.max()), float(e.pow(2).mean().sqrt()))
        # degenerate tiles: all zeros and vanishingly small activations give the bias (no NaN from the scale arithmetic)
        for x in (torch.zeros(B, C, T), base * 1e-36):
            y = eng.op_conv1d(x.cuda(), w, b, dilation=d, padding=d * (K - 1) // 2, pre_lrelu_slope=0.1).cpu()
            assert bool(torch.isfinite(y).all()) and float((y - b.view(1, C, 1)).abs().max()) <= 1e-6
    finally:
        eng.set_arithmetic(orig)


def test_split_builds_match_fp32_mfma_on_random_shapes(eng):
    """Differential test of the two datapaths on one handle: random layer shapes (channels, taps, dilations, odd lengths, ragged last tiles,
    with and without the prologue leaky-relu) that land on conv_split_kernel (40), its 64-channel tile (41) or the balanced grid (60)."""
    rng = torch.Generator().manual_seed(2024)
    seen = set()
    orig = eng.arithmetic()
    for case in range(14):
        cin = [64, 128, 256, 512][int(torch.randint(0, 4, (1,), generator=rng))]
        cout = [64, 128, 192, 256, 512][int(torch.randint(0, 5, (1,), generator=rng))]
        K = [1, 3, 7, 11][int(torch.randint(0, 4, (1,), generator=rng))]
        dil = [1, 3, 5][int(torch.randint(0, 3, (1,), generator=rng))] if K > 1 else 1
        slope = 0.1 if case % 2 else -1.0
        deep = case % 3 != 2                               # two of three cases on a deep grid, the third on a few rounds of workgroups
        rows = ((220000 if cout < 128 else 170000 * 128 // cout) if deep else 20000) + int(torch.randint(0, 999, (1,), generator=rng))
        B = 3
        T = rows // B
        x = torch.randn(B, cin, T, generator=rng)
        w = torch.randn(cout, cin, K, generator=rng) / (cin * K) ** 0.5
        b = torch.randn(cout, generator=rng)
        pad = dil * (K - 1) // 2
        try:
            eng.set_arithmetic(0)
            y0 = eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope)
            assert eng.last_cfg() not in (40, 41, 60, 46, 47, 66)
            for arith in (6, 16):
                eng.set_arithmetic(arith)
                y = eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=pad, pre_lrelu_slope=slope)
                cfg = eng.last_cfg()
                seen.add(cfg)
                if cfg in (40, 41, 60, 46, 47, 66):
                    _close(y, y0, rtol=1e-5, what=f"case {case}: {cin}->{cout} k{K} d{dil} T{T} cfg {cfg}")
                else:
                    assert torch.equal(y, y0), f"case {case}: both settings took the fp32 build {cfg} and must agree bit for bit"
        finally:
            eng.set_arithmetic(orig)
    assert {40, 41, 60, 46, 47, 66} <= seen, f"the shapes were meant to cover all split builds, saw {sorted(seen)}"


def test_conv_transpose1d_split_build(eng):
    """A polyphase transposed conv on a deep grid takes the 64 x 128 tile of the bf16-split build (cfg 41: the two phases' 64-channel
    M tiles carry different tap subsets)."""
    g = torch.Generator().manual_seed(11)
    B, Cin, T, Cout, K, s, p = 3, 128, 33000, 64, 4, 2, 1
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K / s) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=p)
    orig = eng.arithmetic()
    try:
        for arith, want in ((6, 41), (16, 47)):
            eng.set_arithmetic(arith)
            got = eng.op_conv1d(x.cuda(), w, b, transposed=True, stride=s, padding=p, pre_lrelu_slope=0.1)
            assert eng.last_cfg() == want, eng.last_cfg()
            _close(got, ref, what=f"convT 128->64 k4 s2, split build, arithmetic {arith}")
    finally:
        eng.set_arithmetic(orig)


def test_conv_stride2(eng):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 256, 68, generator=g)
    w = torch.randn(256, 256, 3, generator=g) / (256 * 3) ** 0.5
    b = torch.randn(256, generator=g)
    ref = F.conv1d(x, w, b, stride=2, padding=1)
    got = eng.op_conv1d(x.cuda(), w, b, stride=2, padding=1)
    _close(got, ref, what="conv k3 s2")


def test_groupnorm_mish(eng):
    g = torch.Generator().manual_seed(6)
    B, C, T = 3, 256, 77
    x = torch.randn(B, C, T, generator=g) * 2 + 0.5
    gamma = torch.randn(C, generator=g) * 0.1 + 1
    beta = torch.randn(C, generator=g) * 0.1
    lengths = torch.tensor([77, 40, 1])
    mask = (torch.arange(T)[None] < lengths[:, None]).float().unsqueeze(1)
    ref = F.mish(F.group_norm(x, 8, gamma, beta, eps=1e-5)) * mask     # decoder.py:41-43: stats include padded frames
    got = eng.op_groupnorm_mish(x.cuda(), gamma.cuda(), beta.cuda(), lengths.cuda())
    _close(got, ref, rtol=1e-5, what="groupnorm+mish")


def test_layernorm(eng):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(101, 256, generator=g) * 3 + 1
    gamma = torch.randn(256, generator=g) * 0.1 + 1
    beta = torch.randn(256, generator=g) * 0.1
    ref = F.layer_norm(x, (256,), gamma, beta, eps=1e-5)
    got = eng.op_layernorm(x.cuda(), gamma.cuda(), beta.cuda())
    _close(got, ref, rtol=1e-5, what="layernorm")


@pytest.mark.parametrize("B,T,lens", [(2, 70, [70, 33]), (1, 32, [32]), (3, 258, [258, 200, 7]), (1, 130, [130])])
def test_attention(eng, B, T, lens):
    g = torch.Generator().manual_seed(T)
    heads = 2
    qkv = torch.randn(B, T, 3 * heads * 64, generator=g)
    lengths = torch.tensor(lens)
    mask = (torch.arange(T)[None] < lengths[:, None]).float()
    q, k, v = (qkv[..., i * 128:(i + 1) * 128].view(B, T, heads, 64).transpose(1, 2) for i in range(3))
    am = mask.repeat_interleave(heads, dim=0).view(B, heads, 1, T)     # FLOAT mask: added to the scores
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=am).transpose(1, 2).reshape(B, T, heads * 64)
    got = eng.op_attention(qkv.cuda(), lengths.cuda(), heads)
    _close(got, ref, rtol=2e-5, what="attention")


@pytest.mark.parametrize("B,T,lens", [(3, 70, [70, 33, 1]), (2, 516, [516, 400]), (8, 258, [258, 1, 100, 258, 31, 32, 33, 257]), (1, 32, [32]),
                                      (2, 4, [4, 2]), (64, 132, None), (3, 33, [33, 32, 5]), (5, 67, [67, 66, 65, 64, 1]), (16, 98, None)])
def test_attn_out_fused(eng, B, T, lens):
    """attn_out_kernel (both heads of the additive-mask attention + the 128 -> 256 output projection + the residual in one launch,
    transformer.py:262-271) against plain torch fp32 (SDPA with the FLOAT mask added, then Linear with bias, then + hidden).
    T = 516, 258, 132, 33, 67 and 98 end in a short tile of 4, 2, 4, 1, 3 and 2 queries: the 4 x 4-block path (attn_tail_path)."""
    g = torch.Generator().manual_seed(T * 7 + B)
    heads = 2
    qkv = torch.randn(B, T, 3 * heads * 64, generator=g)
    hid = torch.randn(B, T, 256, generator=g)
    w_out = torch.randn(256, 128, generator=g) / 128 ** 0.5
    b_out = torch.randn(256, generator=g) * 0.1
    lengths = torch.tensor(lens) if lens is not None else torch.randint(1, T + 1, (B,), generator=g)
    mask = (torch.arange(T)[None] < lengths[:, None]).float()
    q, k, v = (qkv[..., i * 128:(i + 1) * 128].view(B, T, heads, 64).transpose(1, 2) for i in range(3))
    am = mask.repeat_interleave(heads, dim=0).view(B, heads, 1, T)     # FLOAT mask: added to the scores
    att = F.scaled_dot_product_attention(q, k, v, attn_mask=am).transpose(1, 2).reshape(B, T, heads * 64)
    ref = hid + att @ w_out.T + b_out
    got = eng.op_attn_out(qkv.cuda(), lengths.cuda(), w_out, b_out, hid.cuda())
    _close(got, ref, rtol=2e-5, what="attention + out-proj + residual")
    assert torch.equal(eng.op_attn_out(qkv.cuda(), lengths.cuda(), w_out, b_out, hid.cuda()), got)      # deterministic


@pytest.mark.parametrize("rows,M1", [(70, 1024), (1, 1024), (333, 384), (32, 128)])
def test_ln_mlp_fused(eng, rows, M1):
    """ln_mlp_kernel (LayerNorm + feed-forward / LayerNorm + projection in one launch) against plain torch fp32:
    x + W2 . SnakeBeta(W1 . LN(x) + b1) + b2, * mask (transformer.py:17-80, 300-316) and W . LN(x)."""
    g = torch.Generator().manual_seed(rows + M1)
    x = torch.randn(rows, 256, generator=g) * 1.7 + 0.3
    ln_g, ln_b = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    w1 = torch.randn(M1, 256, generator=g) / 16.0
    b1 = torch.randn(M1, generator=g) * 0.1
    xn = F.layer_norm(x, (256,), ln_g, ln_b, eps=1e-5)
    # projection only (the QKV use has no bias; with bias checked too)
    _close(eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), w1, None), xn @ w1.T, what="ln + linear")
    _close(eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), w1, b1), xn @ w1.T + b1, what="ln + linear + bias")
    # feed-forward
    alpha, beta = torch.randn(M1, generator=g) * 0.3, torch.randn(M1, generator=g) * 0.3
    w2 = torch.randn(256, M1, generator=g) / M1 ** 0.5
    b2 = torch.randn(256, generator=g) * 0.1
    mask = (torch.rand(rows, generator=g) > 0.2).float()
    h = xn @ w1.T + b1
    h = h + (1.0 / (torch.exp(beta) + 0.000000001)) * torch.sin(h * torch.exp(alpha)) ** 2
    ref = (x + h @ w2.T + b2) * mask[:, None]
    got = eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), w1, b1, alpha, beta, w2, b2, mask.cuda())
    _close(got, ref, what="ln + ff")


def _ln_mlp_case(rows, M1, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, 256, generator=g) * 1.7 + 0.3
    ln_g, ln_b = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.1
    w1 = torch.randn(M1, 256, generator=g) / 16.0
    b1 = torch.randn(M1, generator=g) * 0.1
    alpha, beta = torch.randn(M1, generator=g) * 0.3, torch.randn(M1, generator=g) * 0.3
    w2 = torch.randn(256, M1, generator=g) / M1 ** 0.5
    b2 = torch.randn(256, generator=g) * 0.1
    mask = (torch.rand(rows, generator=g) > 0.2).float()
    return x, ln_g, ln_b, w1, b1, alpha, beta, w2, b2, mask


@pytest.mark.parametrize("rows", [8192 + 808, 16640, 20011, 33280])
def test_ln_mlp_balanced_grid(eng, rows, monkeypatch):
    """From one 32-row tile per CU up, ln_mlp_kernel runs as a balanced persistent grid (SkCtl in ev_kernels.h): row tiles are
    split between workgroups along the hidden width and the partial tiles handed over inside the launch.  From one 64-row tile per CU
    up (the last three cases; 20011 rows end in a partial tile) the feed-forward takes ln_mlp_split_kernel: the same grid logic around
    the bf16-split products (six exact bf16 products per fp32 product, fp32 accumulation).  Checked against plain
    torch fp32, against a second handle whose owners never wait (EV_SK_SPIN=0: every hand-off takes the recompute path, which must
    deliver the SAME bits), and launch after launch (the flags carry a device-side epoch, nothing is re-zeroed)."""
    from emojivoice_amd._lib import Engine

    x, ln_g, ln_b, w1, b1, alpha, beta, w2, b2, mask = _ln_mlp_case(rows, 1024, rows)
    xn = F.layer_norm(x, (256,), ln_g, ln_b, eps=1e-5)
    h = xn @ w1.T + b1
    h = h + (1.0 / (torch.exp(beta) + 0.000000001)) * torch.sin(h * torch.exp(alpha)) ** 2
    ref = (x + h @ w2.T + b2) * mask[:, None]
    args = (x.cuda(), ln_g.cuda(), ln_b.cuda(), w1, b1, alpha, beta, w2, b2, mask.cuda())
    got = eng.op_ln_mlp(*args)
    _close(got, ref, what="balanced ln + ff")
    for _ in range(3):
        assert torch.equal(eng.op_ln_mlp(*args), got)
    wq = torch.randn(384, 256, generator=torch.Generator().manual_seed(1)) / 16.0
    _close(eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), wq, None), xn @ wq.T, what="balanced ln + qkv")
    if rows >= 64 * 256 and eng.arithmetic() == 16:
        # from one 64-row tile per CU up the 384-wide projection takes ln_qkv_h16_kernel (fp16 pieces, block-scaled); fp32-grade against
        # fp64, with a bias, and with rows scaled far outside the fp16 range (LayerNorm removes the scale, the tile scale the rest)
        assert eng.last_cfg() == 122
        bq = torch.randn(384, generator=torch.Generator().manual_seed(2)) * 0.1
        ref64 = F.layer_norm(x.double(), (256,), ln_g.double(), ln_b.double(), eps=1e-5) @ wq.double().T + bq.double()
        got_q = eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), wq, bq)
        assert eng.last_cfg() == 122
        err = (got_q.cpu().double() - ref64).abs().max().item()
        assert err <= 2e-5 * ref64.abs().max().item(), err
        assert torch.equal(eng.op_ln_mlp(x.cuda(), ln_g.cuda(), ln_b.cuda(), wq, bq), got_q)
        big_g = ln_g * 3.0e6
        ref_big = F.layer_norm(x.double(), (256,), big_g.double(), ln_b.double(), eps=1e-5) @ wq.double().T
        got_big = eng.op_ln_mlp(x.cuda(), big_g.cuda(), ln_b.cuda(), wq, None)
        assert (got_big.cpu().double() - ref_big).abs().max().item() <= 2e-5 * ref_big.abs().max().item()
    monkeypatch.setenv("EV_SK_SPIN", "0")
    e2 = Engine(0)
    try:
        assert torch.equal(e2.op_ln_mlp(*args), got), "the recompute path of a timed-out hand-off must deliver the contributor's bits"
    finally:
        e2.close()
    monkeypatch.setenv("EV_NO_SK_BALANCE", "1")
    e3 = Engine(0)
    try:
        _close(e3.op_ln_mlp(*args), got, what="one tile per workgroup vs balanced")
    finally:
        e3.close()
