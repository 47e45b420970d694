"""Pin the CPU oracle (oracle/matcha_oracle.py) to the vectors the reference's own
modules produced (tests/golden/make_golden.py -> reference_vectors.npz)."""
import numpy as np
import torch

from emojivoice_amd import weights as W
from oracle import matcha_oracle as O

T = torch.from_numpy


def _linf(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


def test_param_counts_known_answers(golden):
    # synthesis.ipynb:127 — the only numeric known-answer the reference publishes
    assert int(golden["ka_params_single_speaker"]) == 18204193
    assert W.count_params(W.matcha_shapes(178, 1)) == 18204193
    assert W.count_params(W.matcha_shapes(178, 109)) == int(golden["ka_params_emoji"]) == 20857569
    assert W.count_params(W.estimator_shapes(109)) == 11139920
    assert W.count_params(W.hifigan_shapes()) == 13926017


def test_estimator_single_call(golden, matcha_sd):
    lengths = T(golden["g1_lengths"])
    mask = O.sequence_mask(lengths, 32).unsqueeze(1).float()
    spk = matcha_sd["spk_emb.weight"][T(golden["g1_spk_ids"])]
    for i, tv in enumerate(golden["g1_t"]):
        v = O.estimator(matcha_sd, T(golden["g1_x"]), mask, T(golden["g1_mu"]), torch.tensor(float(tv)), spk)
        assert _linf(v, golden[f"g1_v_t{i}"]) <= 2e-5, i


def test_cfm_euler(golden, matcha_sd):
    lengths = T(golden["g1_lengths"])
    mask = O.sequence_mask(lengths, 32).unsqueeze(1).float()
    spk = matcha_sd["spk_emb.weight"][T(golden["g1_spk_ids"])]
    for n in (2, 4, 10):
        y = O.cfm_decode(matcha_sd, T(golden["g1_mu"]), mask, n, 0.667, spk, z=T(golden["g2_z"]))
        assert _linf(y, golden[f"g2_dec_n{n}"]) <= 5e-5, n
    spk1 = matcha_sd["spk_emb.weight"][torch.tensor([58])]
    y = O.cfm_decode(matcha_sd, T(golden["g2b_mu"]), torch.ones(1, 1, 24), 10, 0.667, spk1, z=T(golden["g2b_z"]))
    assert _linf(y, golden["g2b_dec_n10"]) <= 5e-5


def test_text_encoder(golden, matcha_sd):
    spk = matcha_sd["spk_emb.weight"][T(golden["g3_spks"])]
    mu_x, logw, _ = O.text_encoder(matcha_sd, T(golden["g3_ids"]), T(golden["g3_x_lengths"]), spk)
    assert _linf(mu_x, golden["g3_mu_x"]) <= 2e-5
    assert _linf(logw, golden["g3_logw"]) <= 2e-5


def test_synthesise_end_to_end(golden, matcha_sd):
    for tag, ls in (("a", 1.0), ("b", 0.8)):
        torch.manual_seed(777)
        r = O.synthesise(matcha_sd, T(golden["g3_ids"]), T(golden["g3_x_lengths"]), 10, 0.667, T(golden["g3_spks"]), ls)
        assert np.array_equal(r["mel_lengths"].numpy(), golden[f"g3{tag}_mel_lengths"])
        assert tuple(r["attn"].shape) == tuple(golden[f"g3{tag}_attn_shape"])  # text axis sliced, Tp kept (matcha_tts.py:148)
        assert np.array_equal(r["attn"].sum(-1).numpy(), golden[f"g3{tag}_attn_sum_text"])
        assert _linf(r["encoder_outputs"], golden[f"g3{tag}_enc"]) <= 2e-5
        assert _linf(r["decoder_outputs"], golden[f"g3{tag}_dec"]) <= 5e-5
        assert _linf(r["mel"], golden[f"g3{tag}_mel"]) <= 1e-4  # north-star mel gate
        # the stored z is the draw the reference made
        r2 = O.synthesise(matcha_sd, T(golden["g3_ids"]), T(golden["g3_x_lengths"]), 10, 0.667, T(golden["g3_spks"]), ls,
                          z=T(golden[f"g3{tag}_z"]))
        assert _linf(r2["mel"], golden[f"g3{tag}_mel"]) <= 1e-4


def test_hifigan(golden, voc_sd):
    wav, stages = O.hifigan_forward(voc_sd, T(golden["g4_mel"]), W.HIFIGAN_V1, return_stages=True)
    assert wav.shape == (2, 1, 32 * 256)
    assert _linf(stages[0][:, :, :16], golden["g4_stage0_head"]) <= 1e-5
    for i in range(1, 5):
        assert _linf(stages[i][:, :, :64], golden[f"g4_stage{i}_head"]) <= 2e-5, i
        assert _linf(stages[i][:, :, -64:], golden[f"g4_stage{i}_tail"]) <= 2e-5, i
    err = wav.numpy() - golden["g4_wav"]
    assert float(np.sqrt(np.mean(err**2))) <= 1e-6
    # the synthetic recipe must make the 1e-3 waveform gate meaningful
    assert float(np.sqrt(np.mean(golden["g4_wav"] ** 2))) > 0.1


def test_weight_norm_fold(golden, voc_sd):
    folded = W.fold_weight_norm(W.weight_norm_split(voc_sd))
    for k in ("conv_pre.weight", "ups.1.weight", "resblocks.4.convs1.2.weight", "conv_post.weight"):
        assert _linf(folded[k][:8], golden["g5_" + k.replace(".", "_")]) <= 1e-6, k
        assert _linf(folded[k], voc_sd[k]) <= 1e-6, k


def test_denoiser(golden, voc_sd):
    bias = O.denoiser_bias_spec(voc_sd, W.HIFIGAN_V1)
    assert _linf(bias, golden["g7_bias_spec"]) <= 1e-4 * max(1.0, float(np.abs(golden["g7_bias_spec"]).max()))
    audio = T(golden["g4_wav"]).clamp(-1, 1).squeeze()
    d = O.denoiser(audio, T(golden["g7_bias_spec"]), strength=0.00025)
    assert d.shape == golden["g7_denoised"].shape
    assert _linf(d, golden["g7_denoised"]) <= 1e-5


def test_emoji_rule_hand_derived():
    # feel_me.py:298-317 lives inside the script's __main__ loop (needs mic/whisper/ollama),
    # so these cases are derived by reading it, not by running it.
    is_e = lambda c: ord(c) >= 0x1F300
    rep = lambda s, r: "".join(r if is_e(c) else c for c in s)
    assert O.parse_emoji_response("Hello world \U0001F60A", is_e, rep) == ("Hello world ", 0)  # 😊 unmapped -> 0
    assert O.parse_emoji_response("Hi \U0001F642 there \U0001F621", is_e, rep) == ("Hi  there ", 12)  # first mapped wins
    assert O.parse_emoji_response("\U0001F60A ok \U0001F923", is_e, rep) == (" ok ", 15)  # unmapped skipped
    assert O.parse_emoji_response("(wow) \U0001F62E", is_e, rep) == ("wow ", 54)
    assert O.parse_emoji_response("\U0001F914", is_e, rep) == ("nice", 17)
