"""Emoji -> speaker-id rule of the EmojiVoice demos (reference feel_me.py:84-96, :298-317).

The reference uses the third-party ``emoji`` package for ``is_emoji`` /
``replace_emoji``; it is absent offline, so a code-point-range test covering the
emoji blocks (incl. the 11 mapped ones) stands in when the package is missing.
"""
from __future__ import annotations

from typing import Tuple

EMOJI_MAPPING = {  # feel_me.py:84-96 (female voice set of the Paige checkpoint)
    "\U0001F60D": 107,  # 😍
    "\U0001F621": 58,   # 😡
    "\U0001F60E": 79,   # 😎
    "\U0001F62D": 103,  # 😭
    "\U0001F644": 66,   # 🙄
    "\U0001F601": 18,   # 😁
    "\U0001F642": 12,   # 🙂
    "\U0001F923": 15,   # 🤣
    "\U0001F62E": 54,   # 😮
    "\U0001F605": 22,   # 😅
    "\U0001F914": 17,   # 🤔
}
EMOJI_MAPPING_MALE = {  # feel_me.py:98-111 (commented alternative)
    "\U0001F60D": 4, "\U0001F621": 5, "\U0001F60E": 6, "\U0001F62D": 13, "\U0001F644": 16, "\U0001F601": 26,
    "\U0001F642": 30, "\U0001F923": 38, "\U0001F62E": 60, "\U0001F605": 82, "\U0001F914": 97,
}
DEFAULT_SPK = 0            # feel_me.py:304
FALLBACK_TEXT = "nice"     # feel_me.py:316-317

_RANGES = ((0x1F300, 0x1FAFF), (0x2600, 0x27BF), (0x1F000, 0x1F2FF), (0x2B00, 0x2BFF), (0x2190, 0x21FF), (0xFE00, 0xFE0F),
           (0x200D, 0x200D), (0x20E3, 0x20E3), (0x1F1E6, 0x1F1FF))


def _is_emoji_fallback(ch: str) -> bool:
    o = ord(ch)
    return any(lo <= o <= hi for lo, hi in _RANGES)


try:  # pragma: no cover - package absent in the build image
    import emoji as _emoji

    is_emoji = _emoji.is_emoji

    def replace_emoji(text: str, repl: str = "") -> str:
        return _emoji.replace_emoji(text, repl)
except Exception:  # noqa: BLE001
    is_emoji = _is_emoji_fallback

    def replace_emoji(text: str, repl: str = "") -> str:
        return "".join(repl if _is_emoji_fallback(c) else c for c in text)


def emoji_to_spk(response: str, mapping=None, default: int = DEFAULT_SPK) -> int:
    """First mapped emoji in order of appearance wins (the reference's comment says
    'last' but its loop breaks on the first, feel_me.py:304-308)."""
    mapping = EMOJI_MAPPING if mapping is None else mapping
    for ch in response:
        if is_emoji(ch) and ch in mapping:
            return mapping[ch]
    return default


def parse_response(response: str, mapping=None, default: int = DEFAULT_SPK) -> Tuple[str, int]:
    """(text to speak, speaker id): strip emojis and brackets (feel_me.py:309-312); empty -> 'nice'."""
    spk = emoji_to_spk(response, mapping, default)
    text = replace_emoji(response, "").replace(")", "").replace("(", "")
    return (text if text != "" else FALLBACK_TEXT), spk


def first_contained_emoji_spk(line: str, mapping=None, default: int = 12) -> int:
    """The other rule in the repo (hri-demo/storytelling/demo_story_script.py:177-186):
    first emoji in MAPPING order that the line contains; default speaker 12."""
    mapping = EMOJI_MAPPING if mapping is None else mapping
    for e, spk in mapping.items():
        if e in line:
            return spk
    return default
