"""MI355X-native EmojiVoice TTS hot path: Matcha-TTS CFM decoder + HiFi-GAN V1 on
hand-written gfx950 kernels behind the reference's Python call surface."""
from . import emoji, weights  # noqa: F401
from .emoji import EMOJI_MAPPING, emoji_to_spk, parse_response  # noqa: F401

__all__ = ["emoji", "weights", "EMOJI_MAPPING", "emoji_to_spk", "parse_response", "MatchaTTS", "Generator", "AttrDict", "v1"]


def __getattr__(name):  # lazy: keeps `import emojivoice_amd` free of ctypes/GPU side effects
    if name == "MatchaTTS":
        from .matcha_tts import MatchaTTS
        return MatchaTTS
    if name in ("Generator", "AttrDict", "v1", "to_waveform"):
        from . import hifigan
        return getattr(hifigan, name)
    raise AttributeError(name)
