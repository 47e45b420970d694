"""ctypes binding of ``libemojivoice_hip.so`` (C ABI declared in include/emojivoice.h).

There is NO CPU fallback: importing works everywhere (so the package can be
inspected on a CPU box) but any compute call raises ``EvLibraryError`` when the
HIP library is missing or no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EV_LIB_PATH") or os.path.join(_HERE, "lib", "libemojivoice_hip.so")   # EV_LIB_PATH: A/B builds of the same ABI
CSRC = os.path.join(_HERE, "csrc")

EXPORTS = [
    "ev_abi_version", "ev_create", "ev_destroy", "ev_last_error", "ev_load_estimator", "ev_load_vocoder", "ev_load_text_encoder", "ev_text_encoder",
    "ev_text_encoder_status", "ev_stft_magnitude", "ev_denoise", "ev_align", "ev_dbg_conv_bench",
    "ev_workspace_bytes", "ev_cfm_decode", "ev_estimator", "ev_hifigan", "ev_profile_enable", "ev_profile_read", "ev_profile_read_split", "ev_dbg_last_cfg", "ev_set_arithmetic", "ev_get_arithmetic",
    "ev_op_conv1d", "ev_op_groupnorm_mish", "ev_op_layernorm", "ev_op_split_pieces", "ev_op_attention", "ev_op_ln_mlp", "ev_set_mrf_streams_max",
    "ev_cfm_decode2", "ev_reserve", "ev_alloc_count", "ev_dbg_sk_stats", "ev_op_attn_out", "ev_dbg_set_amax", "ev_dbg_set_attn_h16", "ev_dbg_set_chain", "ev_dbg_sk_taken",
]


class EvLibraryError(RuntimeError):
    pass


class ev_tensor_index(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint64), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class ev_model_dims(C.Structure):
    _fields_ = [("n_feats", C.c_int32), ("spk_emb_dim", C.c_int32), ("channels", C.c_int32), ("heads", C.c_int32),
                ("head_dim", C.c_int32)]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into emojivoice_amd/lib/ (hipcc cross-compiles without a GPU)."""
    src = os.path.join(CSRC, "ev_engine.hip")
    deps = [src, os.path.join(CSRC, "ev_kernels.h"), os.path.join(os.path.dirname(_HERE), "include", "emojivoice.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", src, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen the library and declare every prototype.  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the EmojiVoice hot path)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.ev_abi_version.restype = C.c_int
    lib.ev_create.argtypes = [C.POINTER(vp), i32, C.POINTER(ev_model_dims)]
    lib.ev_destroy.argtypes = [vp]
    lib.ev_destroy.restype = None
    lib.ev_last_error.argtypes = [vp]
    lib.ev_last_error.restype = C.c_char_p
    for f in (lib.ev_load_estimator, lib.ev_load_vocoder, lib.ev_load_text_encoder):
        f.argtypes = [vp, vp, C.POINTER(ev_tensor_index), u64]
    lib.ev_set_mrf_streams_max.argtypes = [vp, i32]
    lib.ev_set_mrf_streams_max.restype = i32
    lib.ev_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.ev_workspace_bytes.restype = u64
    lib.ev_cfm_decode.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp]
    lib.ev_cfm_decode2.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, f32, f32, vp, vp]
    lib.ev_reserve.argtypes = [vp, i32, i32, i32, i32, vp]
    lib.ev_alloc_count.argtypes = [vp]
    lib.ev_alloc_count.restype = C.c_int64
    lib.ev_dbg_sk_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.ev_estimator.argtypes = [vp, vp, vp, vp, vp, f32, i32, i32, vp, vp]
    lib.ev_hifigan.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.ev_text_encoder.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.ev_text_encoder_status.argtypes = [vp, vp]
    lib.ev_dbg_conv_bench.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float)]
    lib.ev_stft_magnitude.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.ev_align.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.ev_denoise.argtypes = [vp, vp, i32, i32, vp, f32, vp, vp]
    lib.ev_profile_enable.argtypes = [vp, i32]
    lib.ev_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    lib.ev_profile_read_split.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ev_dbg_last_cfg.argtypes = [vp]
    lib.ev_set_arithmetic.argtypes = [vp, i32]
    lib.ev_get_arithmetic.argtypes = [vp]
    lib.ev_dbg_set_amax.argtypes = [vp, i32]
    lib.ev_dbg_set_attn_h16.argtypes = [vp, i32]
    lib.ev_dbg_set_chain.argtypes = [vp, i32]
    lib.ev_dbg_sk_taken.argtypes = [vp]
    lib.ev_dbg_sk_taken.restype = C.c_int64
    lib.ev_op_conv1d.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp]
    lib.ev_op_groupnorm_mish.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.ev_op_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.ev_op_split_pieces.argtypes = [vp, vp, i32, vp, vp]
    lib.ev_op_attention.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.ev_op_attn_out.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.ev_op_ln_mlp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    for n in EXPORTS:
        getattr(lib, n)  # raises AttributeError if a declared symbol is not exported
    _lib = lib
    return lib


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One ``ev_handle`` on one GPU.  Tensors are torch CUDA tensors (fp32, contiguous)."""

    def __init__(self, device: int = 0, spk_emb_dim: int = 64, heads: int = 2):
        if not torch.cuda.is_available():
            raise EvLibraryError("no ROCm GPU visible: the EmojiVoice hot path has no CPU fallback")
        self.lib = load_library()
        self.device = device
        dims = ev_model_dims(80, spk_emb_dim, 256, heads, 64)
        h = C.c_void_p()
        rc = self.lib.ev_create(C.byref(h), device, C.byref(dims))
        if rc != 0:
            raise EvLibraryError(f"ev_create failed with code {rc}")
        self.h = h
        self.spk_emb_dim = spk_emb_dim
        env = os.environ.get("EV_MRF_STREAMS_MAX", "")   # the handle's default: ev_create reads the same variable the same way (empty = unset)
        if env:                                          # C: `if (fp && *fp) limit = atoi(fp)` — leading integer, 0 when there is none
            import re
            m = re.match(r"\s*([+-]?\d+)", env)
            self.mrf_streams_max = int(m.group(1)) if m else 0
        else:
            self.mrf_streams_max = 16384
        self.pipeline_owner = None                       # {holders, saved limit} while BatchPipelines hold the vocoder's fan-out off (pipeline.py)

    def set_mrf_streams_max(self, max_frames: int) -> None:
        """Largest ``hifigan`` call (B*T mel frames) that runs its three ResBlock1 chains on three streams (0 = never)."""
        if self.lib.ev_set_mrf_streams_max(self.h, int(max_frames)) != 0:
            raise EvLibraryError(self.lib.ev_last_error(self.h).decode())
        self.mrf_streams_max = int(max_frames)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ev_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise EvLibraryError(f"{what} failed: {self.lib.ev_last_error(self.h).decode()}")

    # ---- weights -----------------------------------------------------------
    def _load(self, fn, tensors: Dict[str, torch.Tensor], what: str):
        names = list(tensors.keys())
        arrs = [np.ascontiguousarray(tensors[k].detach().to("cpu", torch.float32).numpy()).reshape(-1) for k in names]
        blob = np.concatenate(arrs) if arrs else np.zeros(1, np.float32)
        idx = (ev_tensor_index * len(names))()
        off = 0
        keep = []
        for i, k in enumerate(names):
            b = k.encode()
            keep.append(b)
            shp = tuple(tensors[k].shape)
            idx[i].name = b
            idx[i].offset = off
            idx[i].ndim = len(shp)
            for d, s in enumerate(shp):
                idx[i].shape[d] = s
            off += arrs[i].size
        self._check(fn(self.h, blob.ctypes.data_as(C.c_void_p), idx, len(names)), what)

    def load_estimator(self, tensors: Dict[str, torch.Tensor]):
        """``tensors``: reference ``decoder.estimator.*`` entries with that prefix stripped, plus the derived
        ``*.ff.net.0.alpha_exp`` / ``*.ff.net.0.beta_inv`` (see matcha_tts.estimator_tensors)."""
        self._load(self.lib.ev_load_estimator, tensors, "ev_load_estimator")

    def load_vocoder(self, tensors: Dict[str, torch.Tensor]):
        self._load(self.lib.ev_load_vocoder, tensors, "ev_load_vocoder")

    def align(self, w_ceil, mu_x, x_lengths, y_lengths, Tp: int, want_attn: bool = True):
        """generate_path + mu_y = attn^T mu_x (utils/model.py:29-41, matcha_tts.py:131-135).  Returns (mu_y (B,80,Tp), attn (B,1,Tx,Tp))."""
        w = self._f32(w_ceil).reshape(w_ceil.shape[0], -1)
        mu_x = self._f32(mu_x)
        B, F, Tx = mu_x.shape
        assert F == 80 and w.shape == (B, Tx)
        xl = x_lengths.to(mu_x.device, torch.int32).contiguous()
        yl = y_lengths.to(mu_x.device, torch.int64).contiguous()
        mu_y = torch.empty((B, F, Tp), dtype=torch.float32, device=mu_x.device)
        attn = torch.empty((B, 1, Tx, Tp), dtype=torch.float32, device=mu_x.device) if want_attn else None
        self._check(self.lib.ev_align(self.h, w.data_ptr(), mu_x.data_ptr(), xl.data_ptr(), yl.data_ptr(), B, Tx, int(Tp), mu_y.data_ptr(),
                                      attn.data_ptr() if attn is not None else None, _stream_ptr()), "ev_align")
        return mu_y, attn

    def stft_magnitude(self, audio):
        """|STFT| (B, 513, L/256 + 1) of (B, L) audio with the denoiser's STFT (denoiser.py:36-56)."""
        audio = self._f32(audio)
        B, L = audio.shape
        mag = torch.empty((B, 513, L // 256 + 1), dtype=torch.float32, device=audio.device)
        self._check(self.lib.ev_stft_magnitude(self.h, audio.data_ptr(), B, L, mag.data_ptr(), _stream_ptr()), "ev_stft_magnitude")
        return mag

    def denoise(self, audio, bias_spec, strength: float):
        """Denoiser.forward (denoiser.py:58-64) on (B, L) audio; bias_spec (513,)."""
        audio = self._f32(audio)
        B, L = audio.shape
        bias = self._f32(bias_spec).reshape(-1)
        assert bias.numel() == 513
        out = torch.empty_like(audio)
        self._check(self.lib.ev_denoise(self.h, audio.data_ptr(), B, L, bias.data_ptr(), float(strength), out.data_ptr(), _stream_ptr()), "ev_denoise")
        return out

    def load_text_encoder(self, tensors: Dict[str, torch.Tensor]):
        self._load(self.lib.ev_load_text_encoder, tensors, "ev_load_text_encoder")

    def text_encoder(self, ids, lengths, spk):
        """(mu_x (B,80,Tx), logw (B,1,Tx)) of TextEncoder.forward (text_encoder.py:378-410), masked by the token lengths."""
        assert ids.is_cuda and ids.dim() == 2
        ids = ids.to(torch.int64).contiguous()
        B, Tx = ids.shape
        lengths = lengths.to(ids.device, torch.int32).contiguous()
        spk_p = None
        if spk is not None:
            spk = self._f32(spk)
            spk_p = spk.data_ptr()
        mu = torch.empty((B, 80, Tx), dtype=torch.float32, device=ids.device)
        logw = torch.empty((B, 1, Tx), dtype=torch.float32, device=ids.device)
        self._check(self.lib.ev_text_encoder(self.h, ids.data_ptr(), lengths.data_ptr(), spk_p, B, Tx, mu.data_ptr(), logw.data_ptr(), _stream_ptr()),
                    "ev_text_encoder")
        return mu, logw

    def text_encoder_status(self):
        """Raises IndexError if a token id outside [0, n_vocab) was seen since the last check (the reference's nn.Embedding
        raises at the lookup, text_encoder.py:395); waits for the current stream."""
        if self.lib.ev_text_encoder_status(self.h, _stream_ptr()) != 0:
            raise IndexError(self.lib.ev_last_error(self.h).decode())

    # ---- hot calls -----------------------------------------------------------
    @staticmethod
    def _f32(t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda, "tensor must live on the GPU"
        return t.contiguous().float()

    def cfm_decode(self, mu, lengths, spk, z, n_steps: int, out_scale: float = 1.0, out_shift: float = 0.0):
        mu, z = self._f32(mu), self._f32(z)
        B, F, Tp = mu.shape
        assert F == 80 and z.shape == mu.shape
        lengths = lengths.to(mu.device, torch.int32).contiguous()
        spk_p = None
        if spk is not None:
            spk = self._f32(spk)
            spk_p = spk.data_ptr()
        out = torch.empty_like(mu)
        self._check(self.lib.ev_cfm_decode(self.h, mu.data_ptr(), lengths.data_ptr(), spk_p, z.data_ptr(), B, Tp, int(n_steps),
                                           float(out_scale), float(out_shift), out.data_ptr(), _stream_ptr()), "ev_cfm_decode")
        return out

    def cfm_decode2(self, mu, lengths, spk, z, n_steps: int, mel_std: float, mel_mean: float):
        """(decoder_outputs, mel) of one decode: both reference outputs from the library, no framework kernel in between."""
        mu, z = self._f32(mu), self._f32(z)
        B, F, Tp = mu.shape
        assert F == 80 and z.shape == mu.shape
        lengths = lengths.to(mu.device, torch.int32).contiguous()
        spk_p = None
        if spk is not None:
            spk = self._f32(spk)
            spk_p = spk.data_ptr()
        dec, mel = torch.empty_like(mu), torch.empty_like(mu)
        self._check(self.lib.ev_cfm_decode2(self.h, mu.data_ptr(), lengths.data_ptr(), spk_p, z.data_ptr(), B, Tp, int(n_steps),
                                            dec.data_ptr(), float(mel_std), float(mel_mean), mel.data_ptr(), _stream_ptr()), "ev_cfm_decode2")
        return dec, mel

    def reserve(self, B: int, Tx_max: int = 0, Tp_max: int = 0, T_voc_max: int = 0) -> None:
        """Pre-size workspace, scratch and staging for batches of B utterances up to these lengths (ev_reserve)."""
        self._check(self.lib.ev_reserve(self.h, int(B), int(Tx_max), int(Tp_max), int(T_voc_max), _stream_ptr()), "ev_reserve")

    def alloc_count(self) -> int:
        return int(self.lib.ev_alloc_count(self.h))

    def sk_taken(self) -> int:
        """Contributor shares taken over by their owners (work stealing of the balanced launches) since the handle was created."""
        return int(self.lib.ev_dbg_sk_taken(self.h))

    def sk_stats(self):
        """(balanced launches so far, arrivals of an unfinished one, hand-off waits that ran out) — diagnostic."""
        out = (C.c_uint32 * 3)()
        self._check(self.lib.ev_dbg_sk_stats(self.h, out), "ev_dbg_sk_stats")
        return tuple(int(v) for v in out)

    def estimator(self, x, mu, lengths, spk, t: float):
        x, mu = self._f32(x), self._f32(mu)
        B, F, Tp = mu.shape
        lengths = lengths.to(mu.device, torch.int32).contiguous()
        spk_p = None
        if spk is not None:
            spk = self._f32(spk)
            spk_p = spk.data_ptr()
        out = torch.empty_like(mu)
        self._check(self.lib.ev_estimator(self.h, x.data_ptr(), mu.data_ptr(), lengths.data_ptr(), spk_p, float(t), B, Tp,
                                          out.data_ptr(), _stream_ptr()), "ev_estimator")
        return out

    def hifigan(self, mel):
        mel = self._f32(mel)
        B, F, T = mel.shape
        assert F == 80
        wav = torch.empty((B, 1, T * 256), dtype=torch.float32, device=mel.device)
        # the kernels address tensors with 32-bit byte offsets (< 4 GiB each): the widest vocoder tensor holds
        # 256 * (T + 8) frames x 128 B per utterance (levels 2-4), so very large batches are processed in row chunks
        per_utt = 256 * (T + 8) * 128
        bmax = max(1, int((2**32 - 2**20) // per_utt))
        for b0 in range(0, B, bmax):
            b1 = min(B, b0 + bmax)
            self._check(self.lib.ev_hifigan(self.h, mel[b0:b1].data_ptr(), b1 - b0, T, wav[b0:b1].data_ptr(), _stream_ptr()), "ev_hifigan")
        return wav

    def workspace_bytes(self, B: int, Tp: int, Tv: int) -> int:
        return int(self.lib.ev_workspace_bytes(self.h, B, Tp, Tv))

    # ---- profiling hooks (bench.py) -------------------------------------------
    def profile_enable(self, on: bool):
        self._check(self.lib.ev_profile_enable(self.h, int(on)), "ev_profile_enable")

    def op_split_pieces(self, x):
        """(3, n) fp32: the three bf16 pieces of every element of x, as the split builds' staging code cuts them."""
        x = self._f32(x).reshape(-1)
        out = torch.empty((3, x.numel()), dtype=torch.float32, device=x.device)
        self._check(self.lib.ev_op_split_pieces(self.h, x.data_ptr(), x.numel(), out.data_ptr(), _stream_ptr()), "ev_op_split_pieces")
        return out

    def set_arithmetic(self, bf16_products: int):
        """Datapath of the deep layers' products (ev_set_arithmetic).  16 (default): two block-scaled fp16 pieces per operand,
        three fp16 products per fp32 product (22-23 significand bits, fp32 accumulation); 6: three bf16 pieces, six exact products;
        0: every product on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 3: opt-in fast bf16 setting, NOT fp32-grade; 9: A/B."""
        self._check(self.lib.ev_set_arithmetic(self.h, int(bf16_products)), "ev_set_arithmetic")

    def set_amax(self, on: bool) -> None:
        """True (default): the fp16 builds take their tile scales from the producers' amax slots; False: every tile pre-scans (ev_dbg_set_amax)."""
        self._check(self.lib.ev_dbg_set_amax(self.h, int(bool(on))), "ev_dbg_set_amax")

    def set_attn_h16(self, on: bool) -> None:
        """True (default): under arithmetic setting 16 the U-Net's self-attention runs on the fp16 pipe (attn_out_h16_kernel); False: on the fp32 MFMA."""
        self._check(self.lib.ev_dbg_set_attn_h16(self.h, int(bool(on))), "ev_dbg_set_attn_h16")

    def set_chain(self, on: bool) -> None:
        """True (default): ResBlock1 chains that qualify (narrow levels, k = 3) run as one launch (resblock_chain_h16_kernel); False: as three fused pairs."""
        self._check(self.lib.ev_dbg_set_chain(self.h, int(bool(on))), "ev_dbg_set_chain")

    def arithmetic(self) -> int:
        return int(self.lib.ev_get_arithmetic(self.h))

    def last_cfg(self) -> int:
        return int(self.lib.ev_dbg_last_cfg(self.h))

    def profile_read_split(self):
        """(ms, flops, launches) of the bf16-split builds among the launches recorded since the last reset."""
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.ev_profile_read_split(self.h, C.byref(ms), C.byref(fl), C.byref(n)), "ev_profile_read_split")
        return ms.value, fl.value, n.value

    def profile_read(self, reset: bool = True):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.ev_profile_read(self.h, C.byref(ms), C.byref(fl), C.byref(n), int(reset)), "ev_profile_read")
        return ms.value, fl.value, n.value

    # ---- operator-level entry points (unit tests) ---------------------------------
    def op_conv1d(self, x, w, bias, dilation=1, transposed=False, stride=1, padding=0, pre_lrelu_slope=-1.0):
        x = self._f32(x)
        B, Cin, T = x.shape
        w = np.ascontiguousarray(w.detach().cpu().float().numpy())
        if transposed:
            Cout, K = w.shape[1], w.shape[2]
            Tout = T * stride
        else:
            Cout, K = w.shape[0], w.shape[2]
            Tout = T // stride
        bp = None
        if bias is not None:
            bnp = np.ascontiguousarray(bias.detach().cpu().float().numpy())
            bp = bnp.ctypes.data_as(C.c_void_p)
        y = torch.empty((B, Cout, Tout), dtype=torch.float32, device=x.device)
        self._check(self.lib.ev_op_conv1d(self.h, x.data_ptr(), w.ctypes.data_as(C.c_void_p), bp, B, Cin, T, Cout, K, dilation,
                                          int(transposed), stride, padding, float(pre_lrelu_slope), y.data_ptr(), _stream_ptr()),
                    "ev_op_conv1d")
        return y

    def op_groupnorm_mish(self, x, gamma, beta, lengths, groups=8):
        x, gamma, beta = self._f32(x), self._f32(gamma), self._f32(beta)
        B, Cc, T = x.shape
        lengths = lengths.to(x.device, torch.int32).contiguous()
        y = torch.empty_like(x)
        self._check(self.lib.ev_op_groupnorm_mish(self.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), lengths.data_ptr(), B, Cc, T,
                                                  groups, y.data_ptr(), _stream_ptr()), "ev_op_groupnorm_mish")
        return y

    def op_layernorm(self, x, gamma, beta):
        x, gamma, beta = self._f32(x), self._f32(gamma), self._f32(beta)
        rows, Cc = x.shape
        y = torch.empty_like(x)
        self._check(self.lib.ev_op_layernorm(self.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, Cc, y.data_ptr(),
                                             _stream_ptr()), "ev_op_layernorm")
        return y

    def op_ln_mlp(self, x, ln_g, ln_b, w1, b1, alpha=None, beta=None, w2=None, b2=None, rowmask=None):
        """ln_mlp_kernel on (rows, 256): with w2 -> x + W2.SnakeBeta(W1.LN(x)+b1)+b2 (* rowmask); without -> W1.LN(x) (+ b1)."""
        x, ln_g, ln_b = self._f32(x), self._f32(ln_g), self._f32(ln_b)
        rows = x.shape[0]
        M1 = w1.shape[0]
        host = lambda t: None if t is None else np.ascontiguousarray(t.detach().cpu().float().numpy())   # noqa: E731
        w1h, b1h, w2h, b2h = host(w1), host(b1), host(w2), host(b2)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)   # noqa: E731
        mode = 0 if w2 is not None else 1
        a_exp = self._f32(torch.exp(alpha.float()).to(x.device)) if alpha is not None else None
        b_inv = self._f32((1.0 / (torch.exp(beta.float()) + 0.000000001)).to(x.device)) if beta is not None else None
        rm = self._f32(rowmask) if rowmask is not None else None
        y = torch.empty((rows, 256 if mode == 0 else M1), dtype=torch.float32, device=x.device)
        dp = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        self._check(self.lib.ev_op_ln_mlp(self.h, x.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), ptr(w1h), ptr(b1h), dp(a_exp), dp(b_inv),
                                          ptr(w2h), ptr(b2h), dp(rm), rows, M1, mode, y.data_ptr(), _stream_ptr()), "ev_op_ln_mlp")
        return y

    def op_attention(self, qkv, lengths, heads=2):
        qkv = self._f32(qkv)
        B, T, _ = qkv.shape
        lengths = lengths.to(qkv.device, torch.int32).contiguous()
        out = torch.empty((B, T, heads * 64), dtype=torch.float32, device=qkv.device)
        self._check(self.lib.ev_op_attention(self.h, qkv.data_ptr(), lengths.data_ptr(), B, T, heads, out.data_ptr(), _stream_ptr()),
                    "ev_op_attention")
        return out

    def op_attn_out(self, qkv, lengths, w_out, b_out, hid):
        """attn_out_kernel: hid + Wout . attention(qkv) + bout; qkv (B, T, 384), hid (B, T, 256), lengths (B,)."""
        qkv, hid = self._f32(qkv), self._f32(hid).clone()
        B, T, _ = qkv.shape
        lengths = lengths.to(qkv.device, torch.int32).contiguous()
        wh = np.ascontiguousarray(w_out.detach().cpu().float().numpy())
        bh = np.ascontiguousarray(b_out.detach().cpu().float().numpy())
        self._check(self.lib.ev_op_attn_out(self.h, qkv.data_ptr(), lengths.data_ptr(), B, T, wh.ctypes.data_as(C.c_void_p), bh.ctypes.data_as(C.c_void_p),
                                            hid.data_ptr(), _stream_ptr()), "ev_op_attn_out")
        return hid
