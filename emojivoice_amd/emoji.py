"""Emoji -> speaker-id rule of the EmojiVoice demos (reference feel_me.py:84-96, :298-317).

The reference uses the third-party ``emoji`` package for ``is_emoji`` /
``replace_emoji``; it is absent offline, so a restatement over the Unicode ``Emoji``
property (emoji-data.txt) with the sequence rules of UTS #51 stands in when the
package is missing.
"""
from __future__ import annotations

import bisect
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

# Code points with the Unicode ``Emoji`` property that the ``emoji`` package (2.x, ``EMOJI_DATA``) lists as an emoji on their own
# (emoji-data.txt 15.1; ASCII digits / # / * are emoji only inside a keycap sequence, see ``_match_emoji``).  Arrows U+2190-2193
# (three of them are symbols of the phoneme table, matcha/text/symbols.py:9), the non-emoji dingbats and the box arrows are NOT emoji;
# U+2194-2199 and U+21A9-21AA are.
_EMOJI_RANGES = (
    (0xA9, 0xA9), (0xAE, 0xAE), (0x203C, 0x203C), (0x2049, 0x2049), (0x2122, 0x2122), (0x2139, 0x2139), (0x2194, 0x2199),
    (0x21A9, 0x21AA), (0x231A, 0x231B), (0x2328, 0x2328), (0x23CF, 0x23CF), (0x23E9, 0x23F3), (0x23F8, 0x23FA), (0x24C2, 0x24C2),
    (0x25AA, 0x25AB), (0x25B6, 0x25B6), (0x25C0, 0x25C0), (0x25FB, 0x25FE), (0x2600, 0x2604), (0x260E, 0x260E), (0x2611, 0x2611),
    (0x2614, 0x2615), (0x2618, 0x2618), (0x261D, 0x261D), (0x2620, 0x2620), (0x2622, 0x2623), (0x2626, 0x2626), (0x262A, 0x262A),
    (0x262E, 0x262F), (0x2638, 0x263A), (0x2640, 0x2640), (0x2642, 0x2642), (0x2648, 0x2653), (0x265F, 0x2660), (0x2663, 0x2663),
    (0x2665, 0x2666), (0x2668, 0x2668), (0x267B, 0x267B), (0x267E, 0x267F), (0x2692, 0x2697), (0x2699, 0x2699), (0x269B, 0x269C),
    (0x26A0, 0x26A1), (0x26A7, 0x26A7), (0x26AA, 0x26AB), (0x26B0, 0x26B1), (0x26BD, 0x26BE), (0x26C4, 0x26C5), (0x26C8, 0x26C8),
    (0x26CE, 0x26CF), (0x26D1, 0x26D1), (0x26D3, 0x26D4), (0x26E9, 0x26EA), (0x26F0, 0x26F5), (0x26F7, 0x26FA), (0x26FD, 0x26FD),
    (0x2702, 0x2702), (0x2705, 0x2705), (0x2708, 0x270D), (0x270F, 0x270F), (0x2712, 0x2712), (0x2714, 0x2714), (0x2716, 0x2716),
    (0x271D, 0x271D), (0x2721, 0x2721), (0x2728, 0x2728), (0x2733, 0x2734), (0x2744, 0x2744), (0x2747, 0x2747), (0x274C, 0x274C),
    (0x274E, 0x274E), (0x2753, 0x2755), (0x2757, 0x2757), (0x2763, 0x2764), (0x2795, 0x2797), (0x27A1, 0x27A1), (0x27B0, 0x27B0),
    (0x27BF, 0x27BF), (0x2934, 0x2935), (0x2B05, 0x2B07), (0x2B1B, 0x2B1C), (0x2B50, 0x2B50), (0x2B55, 0x2B55), (0x3030, 0x3030),
    (0x303D, 0x303D), (0x3297, 0x3297), (0x3299, 0x3299),
    (0x1F004, 0x1F004), (0x1F0CF, 0x1F0CF), (0x1F170, 0x1F171), (0x1F17E, 0x1F17F), (0x1F18E, 0x1F18E), (0x1F191, 0x1F19A),
    (0x1F201, 0x1F202), (0x1F21A, 0x1F21A), (0x1F22F, 0x1F22F), (0x1F232, 0x1F23A), (0x1F250, 0x1F251), (0x1F300, 0x1F321),
    (0x1F324, 0x1F393), (0x1F396, 0x1F397), (0x1F399, 0x1F39B), (0x1F39E, 0x1F3F0), (0x1F3F3, 0x1F3F5), (0x1F3F7, 0x1F4FD),
    (0x1F4FF, 0x1F53D), (0x1F549, 0x1F54E), (0x1F550, 0x1F567), (0x1F56F, 0x1F570), (0x1F573, 0x1F57A), (0x1F587, 0x1F587),
    (0x1F58A, 0x1F58D), (0x1F590, 0x1F590), (0x1F595, 0x1F596), (0x1F5A4, 0x1F5A5), (0x1F5A8, 0x1F5A8), (0x1F5B1, 0x1F5B2),
    (0x1F5BC, 0x1F5BC), (0x1F5C2, 0x1F5C4), (0x1F5D1, 0x1F5D3), (0x1F5DC, 0x1F5DE), (0x1F5E1, 0x1F5E1), (0x1F5E3, 0x1F5E3),
    (0x1F5E8, 0x1F5E8), (0x1F5EF, 0x1F5EF), (0x1F5F3, 0x1F5F3), (0x1F5FA, 0x1F64F), (0x1F680, 0x1F6C5), (0x1F6CB, 0x1F6D2),
    (0x1F6D5, 0x1F6D7), (0x1F6DC, 0x1F6E5), (0x1F6E9, 0x1F6E9), (0x1F6EB, 0x1F6EC), (0x1F6F0, 0x1F6F0), (0x1F6F3, 0x1F6FC),
    (0x1F7E0, 0x1F7EB), (0x1F7F0, 0x1F7F0), (0x1F90C, 0x1F93A), (0x1F93C, 0x1F945), (0x1F947, 0x1F9FF), (0x1FA70, 0x1FA7C),
    (0x1FA80, 0x1FA88), (0x1FA90, 0x1FABD), (0x1FABF, 0x1FAC5), (0x1FACE, 0x1FADB), (0x1FAE0, 0x1FAE8), (0x1FAF0, 0x1FAF8),
)
_VS16, _ZWJ, _KEYCAP = 0xFE0F, 0x200D, 0x20E3
_KEYCAP_BASE = frozenset("0123456789#*")


_LOWS = [lo for lo, _ in _EMOJI_RANGES]          # (ascending: the table is sorted by code point)


def _is_base(o: int) -> bool:
    if o < 0xA9:                                  # ASCII and Latin-1 below the copyright sign: the common case of an LLM response
        return False
    i = bisect.bisect_right(_LOWS, o) - 1
    return i >= 0 and o <= _EMOJI_RANGES[i][1]


def _is_emoji_fallback(ch: str) -> bool:
    """``emoji.is_emoji`` for what the hot path passes it: ONE character (feel_me.py:305 iterates the response by character)."""
    return len(ch) == 1 and _is_base(ord(ch))


def _match_emoji(text: str, i: int) -> int:
    """Length of the emoji sequence that starts at ``text[i]`` (0 = none): a base, optionally with VS16 / a skin-tone modifier,
    joined to further bases by ZWJ; a flag (two regional indicators); a keycap ([0-9#*] VS16? U+20E3); a tag sequence."""
    n = len(text)
    o = ord(text[i])
    if text[i] in _KEYCAP_BASE:
        j = i + 1
        if j < n and ord(text[j]) == _VS16:
            j += 1
        return j + 1 - i if j < n and ord(text[j]) == _KEYCAP else 0
    if 0x1F1E6 <= o <= 0x1F1FF:                               # regional indicators are emoji only as a pair
        return 2 if i + 1 < n and 0x1F1E6 <= ord(text[i + 1]) <= 0x1F1FF else 0
    if not _is_base(o):
        return 0
    j = i + 1
    while True:
        if j < n and 0x1F3FB <= ord(text[j]) <= 0x1F3FF:      # skin tone
            j += 1
        if j < n and ord(text[j]) == _VS16:
            j += 1
        while j < n and 0xE0020 <= ord(text[j]) <= 0xE007F:    # tag sequence (subdivision flags)
            j += 1
        if j + 1 < n and ord(text[j]) == _ZWJ and _is_base(ord(text[j + 1])):
            j += 2
            continue
        return j - i


def _replace_emoji_fallback(text: str, repl: str = "") -> str:
    out, i = [], 0
    while i < len(text):
        k = _match_emoji(text, i)
        if k:
            out.append(repl)
            i += k
        else:
            out.append(text[i])
            i += 1
    return "".join(out)


try:  # pragma: no cover - package absent in the build image
    import emoji as _emoji

    is_emoji = _emoji.is_emoji

    def replace_emoji(text: str, repl: str = "") -> str:
        return _emoji.replace_emoji(text, repl)
except Exception:  # noqa: BLE001
    is_emoji = _is_emoji_fallback
    replace_emoji = _replace_emoji_fallback


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
