#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  Usage:  python tests/golden/make_golden.py

What is the reference's arithmetic and what is not:
  * ``matcha.hifigan.{models,config,env,xutils,denoiser}`` import unmodified.
  * ``matcha.models.matcha_tts / components.{flow_matching,decoder,transformer,
    text_encoder} / utils.model`` import unmodified once the third-party packages
    that are absent offline are pre-registered as inert stand-ins in
    ``sys.modules`` (hydra, omegaconf, lightning, gdown, wget, conformer and the
    compiled Cython ``monotonic_align.core`` — none of them computes anything on
    the inference path).
  * ``diffusers`` (0.25.0, absent) is the one stand-in that carries arithmetic:
    ``Attention`` is restated here from its published algorithm (default
    AttnProcessor2_0 -> ``F.scaled_dot_product_attention`` with the float mask
    ADDED).  Parity at that boundary is therefore UNPINNED.

Weights are the key-seeded synthetic recipe of ``emojivoice_amd.weights`` loaded
into the reference modules with ``load_state_dict(strict=True)``, which also pins
every parameter name and shape of the restated tables.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Matcha-TTS"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins():
    class _Any:  # generic placeholder type
        def __init__(self, *a, **k):
            pass

    _mod("hydra", utils=types.SimpleNamespace(instantiate=None))
    _mod("hydra.core")
    _mod("hydra.core.hydra_config", HydraConfig=_Any)
    _mod("omegaconf", DictConfig=dict, OmegaConf=_Any, open_dict=_Any)
    _mod("gdown")
    _mod("wget")
    _mod("rootutils")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

    _mod("lightning", LightningModule=LightningModule, Callback=_Any)
    _mod("lightning.pytorch")
    _mod("lightning.pytorch.loggers", Logger=_Any)
    _mod("lightning.pytorch.utilities", rank_zero_only=lambda f: f, grad_norm=lambda *a, **k: {})
    _mod("conformer", ConformerBlock=nn.Module)
    _mod("matcha.utils.monotonic_align.core", maximum_path_c=None)

    # ---- diffusers stand-in (the only one with arithmetic) -------------------
    class Attention(nn.Module):
        """diffusers 0.25.0 Attention(query_dim, heads, dim_head, dropout, bias=False,
        cross_attention_dim=None, upcast_attention=False) + AttnProcessor2_0."""

        def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                     upcast_attention=False, **kw):
            super().__init__()
            inner = dim_head * heads
            self.heads = heads
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(query_dim, inner, bias=bias)
            self.to_v = nn.Linear(query_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            b, t, _ = hidden_states.shape
            if attention_mask is not None:
                attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
                attention_mask = attention_mask.view(b, self.heads, -1, attention_mask.shape[-1])
            q = self.to_q(hidden_states)
            k = self.to_k(hidden_states)
            v = self.to_v(hidden_states)
            hd = q.shape[-1] // self.heads
            q = q.view(b, -1, self.heads, hd).transpose(1, 2)
            k = k.view(b, -1, self.heads, hd).transpose(1, 2)
            v = v.view(b, -1, self.heads, hd).transpose(1, 2)
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(b, -1, self.heads * hd)
            o = self.to_out[0](o)
            return self.to_out[1](o)

    def get_activation(name):
        return {"silu": nn.SiLU(), "swish": nn.SiLU(), "mish": nn.Mish(), "gelu": nn.GELU(), "relu": nn.ReLU()}[name]

    _mod("diffusers")
    _mod("diffusers.models")
    _mod("diffusers.models.activations", get_activation=get_activation)
    _mod("diffusers.models.attention", GEGLU=_Any, GELU=_Any, AdaLayerNorm=_Any, AdaLayerNormZero=_Any, ApproximateGELU=_Any)
    _mod("diffusers.models.attention_processor", Attention=Attention)
    _mod("diffusers.models.lora", LoRACompatibleLinear=nn.Linear)
    _mod("diffusers.utils")
    _mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)


def build_reference_matcha(n_vocab, n_spks, sd):
    from matcha.models.matcha_tts import MatchaTTS  # reference, unmodified

    NS = types.SimpleNamespace
    enc_params = NS(n_feats=80, n_channels=192, filter_channels=768, filter_channels_dp=256, n_heads=2, n_layers=6,
                    kernel_size=3, p_dropout=0.1, spk_emb_dim=64, n_spks=1, prenet=True)
    encoder = NS(encoder_type="RoPE Encoder", encoder_params=enc_params,
                 duration_predictor_params=NS(filter_channels_dp=256, kernel_size=3, p_dropout=0.1))
    decoder = dict(channels=[256, 256], dropout=0.05, attention_head_dim=64, n_blocks=1, num_mid_blocks=2, num_heads=2,
                   act_fn="snakebeta")
    cfm = NS(name="CFM", solver="euler", sigma_min=1e-4)
    m = MatchaTTS(n_vocab=n_vocab, n_spks=n_spks, spk_emb_dim=64, n_feats=80, encoder=encoder, decoder=decoder, cfm=cfm,
                  data_statistics={"mel_mean": float(sd["mel_mean"]), "mel_std": float(sd["mel_std"])}, out_size=None)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


def main():
    install_standins()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    from emojivoice_amd import weights as W
    from matcha.hifigan.config import v1
    from matcha.hifigan.denoiser import Denoiser
    from matcha.hifigan.env import AttrDict
    from matcha.hifigan.models import Generator

    out = {}
    # ------------------------------------------------------------ known answers
    single = build_reference_matcha(178, 1, W.synthetic_matcha_state(178, 1))
    out["ka_params_single_speaker"] = np.int64(sum(p.numel() for p in single.parameters()))  # synthesis.ipynb:127
    del single

    n_vocab, n_spks = 178, 109
    sd = W.synthetic_matcha_state(n_vocab, n_spks)
    ref = build_reference_matcha(n_vocab, n_spks, sd)
    out["ka_params_emoji"] = np.int64(sum(p.numel() for p in ref.parameters()))

    g = torch.Generator().manual_seed(20240901)
    # ------------------------------------------------------------ G1 estimator
    B, Tp = 2, 32
    lengths = torch.tensor([32, 21])
    mask = (torch.arange(Tp)[None, :] < lengths[:, None]).float().unsqueeze(1)
    x = torch.randn(B, 80, Tp, generator=g)
    mu = torch.randn(B, 80, Tp, generator=g)
    spk_ids = torch.tensor([12, 107])
    spk = ref.spk_emb(spk_ids).detach()
    out.update(g1_x=x.numpy(), g1_mu=mu.numpy(), g1_lengths=lengths.numpy(), g1_spk_ids=spk_ids.numpy())
    with torch.inference_mode():
        for i, tv in enumerate([0.0, 0.3, 0.9]):
            v = ref.decoder.estimator(x, mask, mu, torch.tensor(tv), spk)
            out[f"g1_v_t{i}"] = v.numpy()
    out["g1_t"] = np.array([0.0, 0.3, 0.9], dtype=np.float32)

    # ------------------------------------------------------------ G2 CFM Euler
    z = torch.randn(B, 80, Tp, generator=g)
    out["g2_z"] = z.numpy()
    with torch.inference_mode():
        for n in (2, 4, 10):
            t_span = torch.linspace(0, 1, n + 1)
            y = ref.decoder.solve_euler(z * 0.667, t_span=t_span, mu=mu, mask=mask, spks=spk, cond=None)
            out[f"g2_dec_n{n}"] = y.numpy()
    # a second, unpadded single-utterance case (mask-insensitive attention)
    B1, T1 = 1, 24
    mask1 = torch.ones(B1, 1, T1)
    mu1 = torch.randn(B1, 80, T1, generator=g)
    z1 = torch.randn(B1, 80, T1, generator=g)
    spk1 = ref.spk_emb(torch.tensor([58])).detach()
    with torch.inference_mode():
        y1 = ref.decoder.solve_euler(z1 * 0.667, t_span=torch.linspace(0, 1, 11), mu=mu1, mask=mask1, spks=spk1, cond=None)
    out.update(g2b_mu=mu1.numpy(), g2b_z=z1.numpy(), g2b_dec_n10=y1.numpy())

    # ------------------------------------------------------------ G3 synthesise
    Lx = 24
    ids = torch.randint(1, n_vocab, (2, Lx), generator=g)
    ids[:, ::2] = 0  # intersperse(ids, 0) look-alike (utils/utils.py:131-135)
    x_lengths = torch.tensor([24, 17])
    ids[1, 17:] = 0
    spks3 = torch.tensor([79, 0])
    out.update(g3_ids=ids.numpy(), g3_x_lengths=x_lengths.numpy(), g3_spks=spks3.numpy())
    for tag, ls in (("a", 1.0), ("b", 0.8)):
        torch.manual_seed(777)
        with torch.inference_mode():
            r = ref.synthesise(ids, x_lengths, n_timesteps=10, temperature=0.667, spks=spks3, length_scale=ls)
        tp = r["attn"].shape[-1]
        torch.manual_seed(777)
        # the same draw synthesise() made: mu_y is a transposed view (matcha_tts.py:134-135), so
        # randn_like (flow_matching.py:51) keeps those strides and takes torch's scalar
        # (non-contiguous) normal_ path — a different stream than randn(2, 80, tp)
        z3 = torch.randn_like(torch.empty(2, tp, 80).transpose(1, 2))
        out[f"g3{tag}_z"] = np.ascontiguousarray(z3.numpy())
        out[f"g3{tag}_mel"] = r["mel"].numpy()
        out[f"g3{tag}_dec"] = r["decoder_outputs"].numpy()
        out[f"g3{tag}_enc"] = r["encoder_outputs"].numpy()
        out[f"g3{tag}_mel_lengths"] = r["mel_lengths"].numpy()
        out[f"g3{tag}_attn_shape"] = np.array(r["attn"].shape, dtype=np.int64)
        out[f"g3{tag}_attn_sum_text"] = r["attn"].sum(-1).numpy()  # frames per token
    with torch.inference_mode():
        spk_e = ref.spk_emb(spks3)
        mu_x, logw, x_mask = ref.encoder(ids, x_lengths, spk_e)
    out.update(g3_mu_x=mu_x.numpy(), g3_logw=logw.numpy())

    # ------------------------------------------------------------ G4 HiFi-GAN
    h = AttrDict(v1)
    voc_sd = W.synthetic_hifigan_state()
    gen = Generator(h)
    gen.remove_weight_norm()
    gen.load_state_dict(voc_sd, strict=True)
    gen.eval()
    mel = torch.randn(2, 80, 32, generator=g) * 2.0 - 5.0
    out["g4_mel"] = mel.numpy()
    with torch.inference_mode():
        wav = gen(mel)
        # stage taps by re-running the reference module list (models.py:181-197)
        xs = gen.conv_pre(mel)
        out["g4_stage0_head"] = xs[:, :, :16].numpy()
        for i in range(gen.num_upsamples):
            xs = F.leaky_relu(xs, 0.1)
            xs = gen.ups[i](xs)
            if i == 0:
                out["g4_up0_head"] = xs[:, :, :64].numpy()
            acc = None
            for j in range(gen.num_kernels):
                r = gen.resblocks[i * gen.num_kernels + j](xs)
                if i == 0 and j == 1:
                    out["g4_rb01_head"] = r[:, :, :64].numpy()
                acc = r if acc is None else acc + r
            xs = acc / gen.num_kernels
            out[f"g4_stage{i + 1}_head"] = xs[:, :, :64].numpy()
            out[f"g4_stage{i + 1}_tail"] = xs[:, :, -64:].numpy()
    out["g4_wav"] = wav.numpy()
    assert wav.shape == (2, 1, 32 * 256)

    # ------------------------------------------------------------ G5 weight-norm fold
    raw = W.weight_norm_split(voc_sd)
    gen2 = Generator(h)
    gen2.load_state_dict(raw, strict=True)  # weight_g / weight_v parametrisation (models.py:152-179)
    gen2.remove_weight_norm()
    folded = gen2.state_dict()
    for k in ("conv_pre.weight", "ups.1.weight", "resblocks.4.convs1.2.weight", "conv_post.weight"):
        out["g5_" + k.replace(".", "_")] = folded[k].numpy()[:8]

    # ------------------------------------------------------------ G7 denoiser (next row f-1)
    den = Denoiser(gen, mode="zeros")
    with torch.inference_mode():
        audio = wav.clamp(-1, 1)
        d = den(audio.squeeze(), strength=0.00025)
    out["g7_bias_spec"] = den.bias_spec.numpy()
    out["g7_denoised"] = d.numpy()

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    tot = sum(v.nbytes for v in out.values())
    print(f"wrote {len(out)} arrays, {tot / 1e6:.2f} MB raw")
    for k in ("ka_params_single_speaker", "ka_params_emoji"):
        print(k, int(out[k]))


if __name__ == "__main__":
    main()
