"""CPU-side checks: the C-ABI library loads and exports every symbol include/emojivoice.h declares (no compute
calls without a GPU), host logic (emoji rule, sharding, weight tables, text-encoder host stage vs the oracle)."""
import ctypes
import os
import re

import pytest
import torch

from emojivoice_amd import _lib, emoji
from emojivoice_amd import weights as W
from oracle import matcha_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "emojivoice.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ev_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/emojivoice.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    lib.ev_abi_version.restype = ctypes.c_int
    assert lib.ev_abi_version() == 4


def test_driver_build_hook_accepts_the_library():
    """__graft_entry__.build()'s post-compile check (ABI of the loaded library == the header's) — without the two-minute compile."""
    import __graft_entry__ as entry

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library()
    assert entry._check_abi(_lib.load_library()) == 4


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EvLibraryError):
        _lib.Engine(0)
    from emojivoice_amd.matcha_tts import MatchaTTS

    with pytest.raises(_lib.EvLibraryError):
        MatchaTTS(W.synthetic_matcha_state(), device="cpu")


def test_product_never_imports_oracle():
    import glob

    for f in glob.glob(os.path.join(REPO, "emojivoice_amd", "**", "*.py"), recursive=True):
        txt = open(f).read()
        assert "import oracle" not in txt and "from oracle" not in txt, f


def test_emoji_rule_matches_oracle_restatement():
    cases = ["Hello world \U0001F60A", "Hi \U0001F642 there \U0001F621", "\U0001F60A ok \U0001F923", "(wow) \U0001F62E", "\U0001F914", "plain",
             "\U0001F62D\U0001F60D both"]
    for c in cases:
        assert emoji.parse_response(c) == O.parse_emoji_response(c, emoji.is_emoji, emoji.replace_emoji)
    assert emoji.EMOJI_MAPPING == O.EMOJI_MAPPING and len(emoji.EMOJI_MAPPING) == 11
    assert emoji.parse_response("Hello world \U0001F60A") == ("Hello world ", 0)       # config 1: unmapped emoji -> speaker 0
    assert emoji.first_contained_emoji_spk("sad \U0001F62D then \U0001F60D") == 107      # mapping order, not text order
    assert emoji.first_contained_emoji_spk("no emoji") == 12


# (character, is an emoji for the `emoji` package) — a fixed table, NOT derived from the product's rule: Unicode emoji-data.txt
# `Emoji` property.  The first five are the arrow symbols of the reference's phoneme table (matcha/text/symbols.py:9).
EMOJI_KNOWN = [
    ("↓", False), ("↑", False), ("→", False), ("↗", True), ("↘", True),
    ("←", False), ("↔", True), ("↙", True), ("↩", True), ("↪", True), ("↫", False), ("⇒", False),
    ("☀", True), ("★", False), ("☎", True), ("☐", False), ("☑", True), ("♠", True), ("♡", False),
    ("✁", False), ("✂", True), ("✓", False), ("✔", True), ("❤", True), ("➔", False), ("➡", True),
    ("⬅", True), ("⬈", False), ("⭐", True), ("⭑", False), ("©", True), ("™", True), ("a", False),
    ("1", False), ("#", False), ("(", False), ("ə", False), ("ˈ", False), ("\U0001F60A", True), ("\U0001F923", True),
    ("\U0001F914", True), ("\U0001F644", True), ("\U0001F1E8", False), ("‍", False), ("️", False), ("\U0001F3FB", True),
    ("\U0001F322", False), ("\U0001F54F", False), ("\U0001FAF9", False),
]


def test_emoji_property_table_and_sequences():
    for ch, want in EMOJI_KNOWN:
        assert emoji._is_emoji_fallback(ch) is want, f"U+{ord(ch):04X}"
    assert all(emoji._is_emoji_fallback(e) for e in emoji.EMOJI_MAPPING)
    rep = emoji._replace_emoji_fallback
    assert rep("a↓↑→b") == "a↓↑→b"                    # phoneme-table arrows survive
    assert rep("a↗b↘c") == "abc"                                            # ... the two that are emoji do not
    assert rep("x \U0001F44D\U0001F3FD y") == "x  y"                                  # skin-tone modifier goes with its base
    assert rep("\U0001F468‍\U0001F469‍\U0001F467!") == "!"                  # ZWJ family: one sequence
    assert rep("go \U0001F1E8\U0001F1E6 go") == "go  go"                              # flag = two regional indicators
    assert rep("1️⃣ and 1 #") == " and 1 #"                                 # keycap sequence; bare digit / '#' stay
    assert rep("❤️ ok") == " ok"                                            # VS16 after a text-default emoji
    assert rep("café ★ ✓") == "café ★ ✓"                # non-emoji dingbats stay
    assert emoji.parse_response("(→) ok \U0001F621") == ("→ ok ", 58)


def test_text_encoder_host_stage_matches_oracle(matcha_sd, golden):
    from emojivoice_amd.text_encoder import TextEncoder, generate_path, sequence_mask

    ids, xl = torch.from_numpy(golden["g3_ids"]), torch.from_numpy(golden["g3_x_lengths"])
    spk = matcha_sd["spk_emb.weight"][torch.from_numpy(golden["g3_spks"])]
    mu, logw, mask = TextEncoder(matcha_sd)(ids, xl, spk)
    assert float((mu - torch.from_numpy(golden["g3_mu_x"])).abs().max()) <= 2e-5
    assert float((logw - torch.from_numpy(golden["g3_logw"])).abs().max()) <= 2e-5
    dur = torch.tensor([[2.0, 0.0, 3.0, 1.0]])
    m = (sequence_mask(torch.tensor([4]), 4).unsqueeze(-1) * sequence_mask(torch.tensor([6]), 8).unsqueeze(1)).float()
    assert torch.equal(generate_path(dur, m), O.generate_path(dur, m))


def test_shard_bounds_cover_batch():
    from emojivoice_amd.dist import shard_bounds

    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_estimator_tensor_export(matcha_sd):
    from emojivoice_amd.matcha_tts import estimator_tensors

    t = estimator_tensors(matcha_sd)
    assert "time_mlp.linear_1.weight" in t and "down_blocks.0.1.0.ff.net.0.alpha_exp" in t
    a = matcha_sd["decoder.estimator.mid_blocks.1.1.0.ff.net.0.beta"]
    assert torch.equal(t["mid_blocks.1.1.0.ff.net.0.beta_inv"], 1.0 / (torch.exp(a) + 1e-9))
    assert sum(v.numel() for k, v in t.items() if not k.endswith(("_exp", "_inv"))) == 11139920


class _FakeDictConfig(dict):
    """Stands in for omegaconf.DictConfig inside a Lightning checkpoint's hyper_parameters."""


def test_checkpoint_reader_stubs_foreign_classes(tmp_path, matcha_sd):
    """cli.py:110-118: real .ckpt files pickle omegaconf / functools.partial(optimizer) objects in ``hyper_parameters``;
    the reader must get the tensors out without importing them."""
    import functools

    from emojivoice_amd.matcha_tts import _load_ckpt

    small = {k: v for k, v in list(matcha_sd.items())[:6]}
    ckpt = {"state_dict": small, "hyper_parameters": {"encoder": _FakeDictConfig(n_feats=80), "optimizer": functools.partial(_FakeDictConfig, lr=1e-4)},
            "epoch": 3, "pytorch-lightning_version": "2.1.0"}
    path = tmp_path / "fake.ckpt"
    torch.save(ckpt, path)
    got = _load_ckpt(str(path))
    assert set(got["state_dict"]) == set(small)
    for k in small:
        assert torch.equal(got["state_dict"][k], small[k])


def test_hifigan_state_dict_roundtrip(voc_sd):
    """Generator.load_state_dict accepts the raw weight_g/weight_v form and the folded form, strictly."""
    from emojivoice_amd.hifigan import AttrDict, Generator, v1

    g = Generator(AttrDict(v1))
    g.load_state_dict(W.weight_norm_split(voc_sd))
    g.remove_weight_norm()
    sd = g.state_dict()
    assert set(sd) == set(voc_sd)
    assert max(float((sd[k] - voc_sd[k]).abs().max()) for k in voc_sd) <= 1e-6
    bad = dict(voc_sd)
    bad.pop("conv_post.bias")
    with pytest.raises(RuntimeError):
        Generator(AttrDict(v1)).load_state_dict(bad)


def test_text_encoder_tensor_export():
    """Host loader of ev_load_text_encoder: ``encoder.*`` keys without the prefix + the rotary table of the reference's
    expression (text_encoder.py:115-117) at d = int(k_channels * 0.5), for the multi- and single-speaker widths."""
    import torch

    from emojivoice_amd import weights as W
    from emojivoice_amd.matcha_tts import text_encoder_tensors

    for n_spks, kc in ((109, 128), (1, 96)):
        sd = W.synthetic_matcha_state(178, n_spks)
        t = text_encoder_tensors(sd)
        assert "emb.weight" in t and "proj_w.proj.weight" in t and not any(k.startswith("encoder.emb") for k in t)
        assert t["encoder.attn_layers.0.conv_q.weight"].shape[0] == 2 * kc
        d = kc // 2
        assert t["rope_theta"].shape == (d // 2,)
        assert torch.equal(t["rope_theta"], 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d)))


def test_checkpoint_reader_does_not_execute_pickled_callables(tmp_path):
    """a checkpoint whose pickle REDUCEs ``builtins.eval`` / ``os.system`` must load as inert stubs (or
    fail), never run: the reader allow-lists exact (module, name) pairs, not whole modules."""
    import builtins
    import os as _os

    from emojivoice_amd.matcha_tts import _load_ckpt

    marker = tmp_path / "pwned"

    class EvalPayload:
        def __reduce__(self):
            return (builtins.eval, (f"open({str(marker)!r}, 'w').write('x')",))

    class SystemPayload:
        def __reduce__(self):
            return (_os.system, (f"touch {marker}",))

    class GetattrPayload:
        def __reduce__(self):
            return (builtins.__import__, ("subprocess",))

    for proto in (2, 4):
        for i, payload in enumerate((EvalPayload(), SystemPayload(), GetattrPayload())):
            path = tmp_path / f"evil_{proto}_{i}.ckpt"
            torch.save({"state_dict": {"w": torch.arange(3.0)}, "hyper_parameters": payload}, path, pickle_protocol=proto)
            ck = _load_ckpt(str(path))
            assert torch.equal(ck["state_dict"]["w"], torch.arange(3.0))
            assert type(ck["hyper_parameters"]).__name__ == "_Stub"
            assert not marker.exists()
