"""Drop-in for ``matcha.models.matcha_tts.MatchaTTS`` on the inference path.

Keeps the reference call surface (matcha_tts.py:78, return keys :145-152):

    MatchaTTS.synthesise(x, x_lengths, n_timesteps, temperature=1.0, spks=None, length_scale=1.0) -> dict

``.n_spks``, ``.eval()``, ``.to(device)`` and ``MatchaTTS.load_from_checkpoint(path,
map_location=)`` behave as the callers in cli.py / feel_me.py expect.  The text
encoder + duration predictor (``ev_text_encoder``), the monotonic alignment
(``ev_align``) and the CFM Euler loop over the U-Net estimator (``ev_cfm_decode2``)
run in the HIP library through the C ABI; only the duration rounding and the
integer ``y_lengths`` stay torch ops (data-dependent sizes, matcha_tts.py:121-128).
``encoder_stage = "host"`` selects the plain-torch text encoder + alignment instead
(north_star's "run once on host").  Training (``forward``) is out of scope and raises.
"""
from __future__ import annotations

import datetime as dt
import pickle
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import weights as W
from ._lib import Engine, EvLibraryError
from .text_encoder import TextEncoder, fix_len_compatibility, generate_path, sequence_mask

EST_PREFIX = "decoder.estimator."


def estimator_tensors(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Estimator entries of a Matcha state dict with the prefix stripped, plus the two
    derived SnakeBeta vectors the kernels consume (transformer.py:71-78, evaluated with the
    same torch ops as the reference): ``alpha_exp = exp(alpha)``, ``beta_inv = 1/(exp(beta)+1e-9)``."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if not k.startswith(EST_PREFIX):
            continue
        name = k[len(EST_PREFIX):]
        v = v.detach().to("cpu", torch.float32)
        out[name] = v
        if name.endswith(".ff.net.0.alpha"):
            out[name + "_exp"] = torch.exp(v)
        elif name.endswith(".ff.net.0.beta"):
            out[name + "_inv"] = 1.0 / (torch.exp(v) + 0.000000001)
    return out


def text_encoder_tensors(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Tensors handed to ``ev_load_text_encoder``: ``encoder.*`` without the prefix, plus the rotary frequency table built
    with the reference's own expression (text_encoder.py:115-117; d = int(k_channels * 0.5), :196-199)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.startswith("encoder."):
            out[k[len("encoder."):]] = v.detach().float()
    n_heads = 2
    kc = out["encoder.attn_layers.0.conv_q.weight"].shape[0] // n_heads
    d = int(kc * 0.5)
    out["rope_theta"] = 1.0 / (10000.0 ** (torch.arange(0, d, 2).float() / d))
    return out


class MatchaTTS:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda:0", n_heads_encoder: int = 2, n_layers_encoder: int = 6):
        sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.n_spks = int(sd["spk_emb.weight"].shape[0]) if "spk_emb.weight" in sd else 1
        self.spk_emb_dim = int(sd["spk_emb.weight"].shape[1]) if self.n_spks > 1 else 0
        self.n_vocab = int(sd["encoder.emb.weight"].shape[0])
        self.n_feats = int(sd["encoder.proj_m.weight"].shape[0])
        self.mel_mean = float(sd.get("mel_mean", torch.tensor(0.0)))
        self.mel_std = float(sd.get("mel_std", torch.tensor(1.0)))
        self._cpu_sd = sd
        self._enc_cfg = (n_heads_encoder, n_layers_encoder)
        self.encoder_stage = "device"   # "device": ev_text_encoder (HIP, through the C ABI); "host": the plain-torch TextEncoder
        self.rng = "cpu"   # "cpu": z drawn exactly as the reference CPU run draws it (seed parity); "device": torch.cuda RNG
        self.decode_graphs = None   # enable_decode_graphs(): the CFM decode replayed from HIP graphs (one per shape)
        self.engine: Optional[Engine] = None
        self.device = torch.device("cpu")
        self.to(device)

    # ---- nn.Module-like surface ------------------------------------------------
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise EvLibraryError("emojivoice_amd.MatchaTTS runs on a ROCm GPU only (no CPU fallback); got device " + str(device))
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self._sd = {k: v.to(self.device) for k, v in self._cpu_sd.items() if not k.startswith(EST_PREFIX)}
        self.encoder = TextEncoder(self._sd, *self._enc_cfg)
        if self.engine is not None:
            self.engine.close()
        self.engine = Engine(idx, spk_emb_dim=self.spk_emb_dim)
        self.engine.load_estimator(estimator_tensors(self._cpu_sd))
        self.engine.load_text_encoder(text_encoder_tensors(self._cpu_sd))
        return self

    def encode(self, x, x_lengths, spk):
        """TextEncoder.forward (text_encoder.py:378-410): (mu_x, logw, x_mask).  ``encoder_stage = "host"`` runs the plain
        torch restatement instead (north_star's "text encoder run once on host")."""
        if self.encoder_stage == "host":
            return self.encoder(x, x_lengths, spk)
        mu_x, logw = self.engine.text_encoder(x, x_lengths, spk)
        x_mask = sequence_mask(x_lengths, x.shape[1]).unsqueeze(1).to(mu_x.dtype)
        return mu_x, logw, x_mask

    def state_dict(self):
        return dict(self._cpu_sd)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None):
        """cli.py:110-118.  A Lightning ``.ckpt`` is a pickle with ``state_dict`` and ``hyper_parameters``;
        only tensors are needed here, so unknown classes in ``hyper_parameters`` (omegaconf, functools.partial
        of an optimizer) are stubbed by a restricted unpickler instead of being imported."""
        ckpt = _load_ckpt(checkpoint_path)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        return cls(sd, device=map_location if map_location is not None else "cuda:0")

    def forward(self, *a, **k):
        raise NotImplementedError("training forward (matcha_tts.py:154-246) is out of scope of the inference hot path")

    # ---- the hot call ------------------------------------------------------------
    def _durations(self, x, x_lengths, spks, length_scale):
        """matcha_tts.py:116-125: speaker embedding, text encoder, duration rounding.  Returns
        (spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths) on the device and max(y_lengths) as a host int."""
        dev = self.device
        x, x_lengths = x.to(dev), x_lengths.to(dev)
        if self.n_spks > 1:
            spk = F.embedding(spks.to(dev).long(), self._sd["spk_emb.weight"])   # AttributeError on None, like the reference
        else:
            spk = None
        mu_x, logw, x_mask = self.encode(x, x_lengths, spk)
        w = torch.exp(logw) * x_mask
        w_ceil = torch.ceil(w) * length_scale
        # y_lengths is an INTEGER output (truncation of a float sum): with a fractional length_scale the sum sits on or next to
        # an integer (55 x 0.8 = 44) and the summation order decides the truncation.  The reduction is therefore done by the same
        # torch CPU op the reference CPU run uses, on the (B, Tx) durations copied to the host — the copy replaces the
        # device -> host read of max(y_lengths) that the path needs anyway (matcha_tts.py:125), so it adds no synchronisation.
        y_lengths_host = torch.clamp_min(torch.sum(w_ceil.cpu(), [1, 2]), 1).long()
        return spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths_host.to(dev), int(y_lengths_host.max())

    def _decode_aligned(self, spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths, y_max_length, n_timesteps, temperature, z=None):
        """matcha_tts.py:125-152 for a given (possibly global, see dist.synthesise_sharded) maximum length."""
        y_max_length_ = fix_len_compatibility(y_max_length)
        y_mask = sequence_mask(y_lengths, y_max_length_).unsqueeze(1).to(x_mask.dtype)
        if self.encoder_stage == "host":
            attn_mask = x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)
            attn = generate_path(w_ceil.squeeze(1), attn_mask.squeeze(1)).unsqueeze(1)
            mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
        else:
            mu_y, attn = self.engine.align(w_ceil, mu_x, x_lengths, y_lengths, y_max_length_)
        encoder_outputs = mu_y[:, :, :y_max_length]
        dec, mel = self.decode(mu_y, y_lengths, n_timesteps, temperature, spk, z=z)
        return encoder_outputs, dec[:, :, :y_max_length], mel[:, :, :y_max_length], attn[:, :, :y_max_length]

    @torch.inference_mode()
    def synthesise(self, x, x_lengths, n_timesteps, temperature=1.0, spks=None, length_scale=1.0, *, z=None):
        """Reference matcha_tts.py:77-152.  Extension: keyword-only ``z`` (unit normal, (B, 80, Tp)) replaces the
        internal draw for bit-reproducible parity runs."""
        t0 = dt.datetime.now()
        spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths, y_max_length = self._durations(x, x_lengths, spks, length_scale)
        if self.encoder_stage != "host":
            self.engine.text_encoder_status()        # out-of-range token id -> IndexError, as nn.Embedding raises
        encoder_outputs, dec, mel, attn = self._decode_aligned(spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths, y_max_length,
                                                                n_timesteps, temperature, z=z)
        t = (dt.datetime.now() - t0).total_seconds()
        rtf = t * 22050 / (dec.shape[-1] * 256)
        return {
            "encoder_outputs": encoder_outputs,
            "decoder_outputs": dec,
            "attn": attn,
            "mel": mel,
            "mel_lengths": y_lengths,
            "rtf": rtf,
        }

    @torch.inference_mode()
    def warmup(self, n_tokens: int = 16, n_timesteps: int = 2, max_frames: int = 0, max_tokens: int = 0, batch: int = 1) -> None:
        """One tiny synthesis so that the first real request does not pay for code-object loading and the first workspace
        allocation (150-400 ms on a fresh process).  ``max_frames`` (mel frames of the longest utterance to expect; ``max_tokens``
        likewise for the text, default ``max_frames``) pre-sizes the workspace, scratch and staging through ``ev_reserve``, so
        that no later request up to that length allocates or re-plans on the request path.  No reference counterpart (torch's
        caching allocator amortises the same cost, feel_me.py:181-203)."""
        if max_frames > 0:
            self.engine.reserve(batch, max_tokens if max_tokens > 0 else max_frames, fix_len_compatibility(max_frames), 0)
        ids = torch.ones((1, n_tokens), dtype=torch.long, device=self.device)
        lens = torch.tensor([n_tokens], device=self.device)
        spks = torch.zeros((1,), dtype=torch.long, device=self.device) if self.n_spks > 1 else None
        self.synthesise(ids, lens, n_timesteps, 0.667, spks, 1.0)
        torch.cuda.synchronize(self.device)

    def draw_noise(self, B: int, Tp: int) -> torch.Tensor:
        """The reference draws ``randn_like(mu_y)`` (flow_matching.py:51) where ``mu_y`` is a transposed view
        (matcha_tts.py:134-135): the draw keeps those strides and uses torch's strided ``normal_`` path.  Doing
        exactly that on the CPU reproduces the reference's noise for a given ``torch.manual_seed``."""
        if self.rng == "device":
            return torch.randn(B, Tp, self.n_feats, device=self.device).transpose(1, 2)
        return torch.randn_like(torch.empty(B, Tp, self.n_feats).transpose(1, 2)).to(self.device)

    def decode(self, mu_y, y_lengths, n_timesteps, temperature=1.0, spk=None, z=None):
        """CFM.forward (flow_matching.py:32-53) + denormalize (utils/model.py:71-90) through the C ABI.
        Returns (decoder_outputs, mel), both (B, 80, Tp)."""
        B, _, Tp = mu_y.shape
        if z is None:
            z = self.draw_noise(B, Tp)
        x0 = (z.to(self.device) * temperature).contiguous()
        mu_c = mu_y.contiguous()
        if self.decode_graphs is not None and spk is not None:
            return self.decode_graphs.run(mu_c, y_lengths, spk, x0, n_timesteps)
        return self.engine.cfm_decode2(mu_c, y_lengths, spk, x0, n_timesteps, self.mel_std, self.mel_mean)

    def enable_decode_graphs(self, max_graphs: int = 256) -> "DecodeGraphs":
        """Replay the CFM decode from HIP graphs, one per (batch, padded length, step count) — the streaming loop's ~700 launches per
        utterance (feel_me.py:189-203) become one ``hipGraphLaunch``.  Call after ``warmup(max_frames=...)``: a captured decode needs
        the workspace to be there already.  Results are those of the eager call, bit for bit (``tests/test_gpu_configs.py``)."""
        self.decode_graphs = DecodeGraphs(self, max_graphs)
        return self.decode_graphs


class DecodeGraphs:
    """LRU cache of captured ``ev_cfm_decode2`` calls of one ``MatchaTTS``.  A graph owns static input / output tensors (inputs are
    copied in, outputs cloned out: ~0.4 MB per utterance); the engine keeps the graphs of different shapes valid beside each other
    and beside eager calls (include/emojivoice.h, "Graph capture").  A length that cannot be captured (workspace too small for it:
    not reserved) falls back to the eager call — and says so once."""

    def __init__(self, model: MatchaTTS, max_graphs: int = 256):
        self.model, self.max_graphs = model, int(max_graphs)
        self.stream = torch.cuda.Stream(device=model.device)
        self._cache: "Dict[tuple, dict]" = {}
        self._order: list = []
        self.hits = self.captures = self.fallbacks = 0
        self._warned = False

    def _capture(self, key, mu, lengths, spk, x0, n_timesteps):
        m = self.model
        ent = {"mu": mu.clone(), "len": lengths.to(m.device, torch.int32).clone(), "spk": spk.to(torch.float32).contiguous().clone(), "x0": x0.clone()}
        # one eager call first: it leaves the time-MLP output for this step count on the device (a captured call must find it there,
        # include/emojivoice.h "Graph capture") and is the plain-library error path for anything the shape cannot do
        m.engine.cfm_decode2(ent["mu"], ent["len"], ent["spk"], ent["x0"], n_timesteps, m.mel_std, m.mel_mean)
        cur = torch.cuda.current_stream(m.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                ent["dec"], ent["mel"] = m.engine.cfm_decode2(ent["mu"], ent["len"], ent["spk"], ent["x0"], n_timesteps, m.mel_std, m.mel_mean)
        cur.wait_stream(self.stream)
        ent["graph"] = g
        self.captures += 1
        return ent

    def run(self, mu, lengths, spk, x0, n_timesteps):
        key = (tuple(mu.shape), int(n_timesteps))
        ent = self._cache.get(key)
        if ent is None:
            try:
                ent = self._capture(key, mu, lengths, spk, x0, n_timesteps)
            except Exception as e:  # noqa: BLE001 - the eager call is always available
                self.fallbacks += 1
                if not self._warned:
                    self._warned = True
                    print(f"[emojivoice_amd] decode graph for shape {key} not captured ({type(e).__name__}: {e}); running eagerly")
                m = self.model
                return m.engine.cfm_decode2(mu, lengths, spk, x0, n_timesteps, m.mel_std, m.mel_mean)
            self._cache[key] = ent
            self._order.append(key)
            if len(self._order) > self.max_graphs:
                old = self._order.pop(0)
                self._cache.pop(old, None)
        else:
            self.hits += 1
            self._order.remove(key)
            self._order.append(key)
        ent["mu"].copy_(mu); ent["len"].copy_(lengths); ent["spk"].copy_(spk); ent["x0"].copy_(x0)
        ent["graph"].replay()
        return ent["dec"].clone(), ent["mel"].clone()


# ---------------------------------------------------------------------------
# checkpoint reading without lightning / omegaconf
# ---------------------------------------------------------------------------
class _Stub(dict):
    """Inert stand-in for any class the checkpoint pickles that is not a tensor container (omegaconf nodes,
    functools.partial of an optimizer, lightning enums ...): accepts every way pickle builds or fills an object."""

    def __init__(self, *a, **k):
        super().__init__()

    def __call__(self, *a, **k):
        return self

    def __setstate__(self, s):
        pass

    def __reduce_ex__(self, protocol):
        return (_Stub, ())

    def append(self, *_):
        pass

    def extend(self, *_):
        pass

    def add(self, *_):
        pass

    def __setattr__(self, k, v):
        pass


# Exact (module, name) pairs a tensor checkpoint needs.  Everything else — including the rest of ``builtins`` (eval, exec,
# getattr, __import__ ...), ``torch`` (hub.load ...) and ``numpy`` (load ...) — resolves to the inert ``_Stub``: a pickle
# can name any importable callable in a REDUCE, so allow-listing whole modules is arbitrary code execution.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("_codecs", "encode"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "bytearray"), ("builtins", "complex"), ("builtins", "slice"),
    ("__builtin__", "set"), ("__builtin__", "frozenset"), ("__builtin__", "dict"), ("__builtin__", "list"),
    ("__builtin__", "tuple"), ("__builtin__", "int"), ("__builtin__", "long"), ("__builtin__", "float"),
    ("__builtin__", "bool"), ("__builtin__", "complex"), ("__builtin__", "slice"),
}
_TORCH_STORAGES = {"FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
                   "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage"}
_TORCH_DTYPES = {"float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool", "float", "double",
                 "half", "long", "int", "short"}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED_GLOBALS:
            if module == "__builtin__":
                module, name = "builtins", ("int" if name == "long" else name)
            return super().find_class(module, name)
        if module == "torch" and (name in _TORCH_STORAGES or name in _TORCH_DTYPES):
            return getattr(torch, name)
        return _Stub


class _PickleShim:
    Unpickler = _RestrictedUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def _load_ckpt(path):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception:  # noqa: BLE001 - hyper_parameters hold non-tensor classes
        return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleShim)


def synthetic(n_vocab: int = W.N_VOCAB_DEFAULT, n_spks: int = W.N_SPKS_EMOJI, device="cuda:0") -> MatchaTTS:
    """Random-init model of the emoji architecture (no checkpoint exists offline)."""
    return MatchaTTS(W.synthetic_matcha_state(n_vocab, n_spks), device=device)
