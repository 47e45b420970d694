"""The TTS leg of the ``feel_me.py`` loop (SURVEY §8 a-9 + BASELINE config 5): one LLM response in, one waveform out.

Mirrors ``feel_me.py:298-317`` (emoji -> speaker, strip emojis / brackets, empty -> 'nice'), ``:130-152``
(``process_text``: cleaner -> ids -> ``intersperse(ids, 0)``), ``:189-203`` (``play_only_synthesis`` with
``SPEAKING_RATE = 0.8``, ``STEPS = 10``, ``TTS_TEMPERATURE = 0.667``, ``:70-77``) and ``:181-187`` (``to_waveform``:
clamp, denoiser at strength 0.00025, copy to the host).  Audio playback (``sounddevice``), ASR and the LLM are out of scope.

The cleaner's phonemiser (espeak-ng / misaki) is absent offline, so the text -> ids step is a callable: by default
``emojivoice_amd.text.text_to_sequence(text, [cleaner of ``language``], phonemize)`` with the caller's ``phonemize``
(en / fr / de / ja as in ``process_text``; "es" raises as the reference's broken ``spanish_cleaners`` does); tests and
benchmarks pass a stand-in front end that maps the stripped text through the symbol table.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from . import text as T
from .emoji import parse_response
from .hifigan import to_waveform

SPEAKING_RATE = 0.8       # feel_me.py:71
STEPS = 10                # feel_me.py:72
TTS_TEMPERATURE = 0.667   # feel_me.py:77


def table_front_end(text: str) -> List[int]:
    """Stand-in front end for boxes without espeak-ng: ``basic_cleaners`` + the symbols the table knows (letters and
    punctuation are table symbols, digits and others are dropped).  NOT the reference's phonemisation — it exists so that the
    streaming loop can be driven end to end from text with ids of realistic length."""
    cleaned = "".join(ch for ch in T.basic_cleaners(text) if ch in T._symbol_to_id)
    return T.cleaned_text_to_sequence(cleaned if cleaned else "a")


class EmojiTTS:
    def __init__(self, model, vocoder, denoiser=None, text_to_ids: Optional[Callable[[str], List[int]]] = None,
                 phonemize: Optional[Callable[[str], str]] = None, language: str = "en"):
        self.model, self.vocoder, self.denoiser = model, vocoder, denoiser
        if language not in T.CLEANER_BY_LANGUAGE:            # feel_me.py:143-145 (the reference prints this and exits)
            raise ValueError("Invalid language. Current supported languages: en (English), fr (French), ja (Japanese), de (German).")
        self.language = language
        if text_to_ids is None:
            def text_to_ids(text, _p=phonemize, _c=T.CLEANER_BY_LANGUAGE[language]):   # process_text (feel_me.py:135-147): cleaner by LANGUAGE
                return T.text_to_sequence(text, [_c], _p)[0]
        self.text_to_ids = text_to_ids

    def process_text(self, text: str) -> Dict[str, torch.Tensor]:
        dev = self.model.device
        x = torch.tensor(T.intersperse(self.text_to_ids(text), 0), dtype=torch.long, device=dev)[None]
        return {"x_orig": text, "x": x, "x_lengths": torch.tensor([x.shape[-1]], dtype=torch.long, device=dev)}

    @torch.inference_mode()
    def synthesise_text(self, text: str, spk: int, z: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """``play_only_synthesis`` (feel_me.py:189-203) up to the waveform on the host."""
        tp = self.process_text(text.strip())
        spks = torch.tensor([spk], device=self.model.device, dtype=torch.long)
        out = self.model.synthesise(tp["x"], tp["x_lengths"], n_timesteps=STEPS, temperature=TTS_TEMPERATURE, spks=spks,
                                    length_scale=SPEAKING_RATE, z=z)
        out["waveform"] = to_waveform(out["mel"], self.vocoder, self.denoiser)
        out["x"] = tp["x"]
        return out

    def respond(self, response: str, z: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """One LLM response -> waveform: the emoji rule, then the synthesis above (feel_me.py:298-317)."""
        text, spk = parse_response(response)
        out = self.synthesise_text(text, spk, z=z)
        out["text"], out["spk"] = text, spk
        return out
