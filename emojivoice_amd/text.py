"""Symbol-table half of the text front end: phoneme string -> ids, the step in front of ``synthesise``.

Follows the reference's ``matcha/text/__init__.py:10-41`` (``text_to_sequence``, ``cleaned_text_to_sequence``,
``sequence_to_text``), ``matcha/text/symbols.py:1-17`` (the 198-entry table) and ``matcha/utils/utils.py:131-135``
(``intersperse``), so that a caller of ``feel_me.py:130-152`` / ``cli.py:33-59`` ``process_text`` can feed this package.

The phonemiser itself (``phonemizer`` + the espeak-ng binary, cleaners.py:26-61) is not available offline, so
``english_cleaners2`` (cleaners.py:248-257) takes it as a callable: everything before and after the phonemiser call is
restated here; without a phonemiser the cleaner raises instead of guessing.  Pre-phonemised (IPA) strings go straight
through ``cleaned_text_to_sequence``.

The table is stored as code points.  Quirks kept (they are part of what a checkpoint's embedding rows mean): the
apostrophe appears five times in the IPA block, so ``len(symbols) == 198`` with 194 distinct symbols, and the
symbol -> id dict keeps the LAST position of a repeated symbol (id 184 for the apostrophe); ``n_vocab`` comes from the
checkpoint, not from this table (configs/model/matcha.yaml:9 says 178).
"""
from __future__ import annotations

import re
from typing import Callable, List, Optional, Sequence, Tuple

_CODEPOINTS = """
5f 3b 3a 2c 2e 21 3f a1 bf 2014 2026 22 ab bb 201c 201d 20 41 42 43 44 45 46 47 48 49 4a 4b 4c 4d 4e 4f 50 51
52 53 54 55 56 57 58 59 5a 61 62 63 64 65 66 67 68 69 6a 6b 6c 6d 6e 6f 70 71 72 73 74 75 76 77 78 79 7a 251
250 252 e6 253 299 3b2 254 255 e7 257 256 f0 2a4 259 258 25a 25b 25c 25d 25e 25f 284 261 260 262 29b 266 267
127 265 29c 268 26a 29d 26d 26c 26b 26e 29f 271 26f 270 14b 273 272 274 f8 275 278 3b8 153 276 298 279 27a 27e
27b 280 281 27d 282 283 288 2a7 289 28a 28b 2c71 28c 263 264 28d 3c7 28e 28f 291 290 292 294 2a1 295 2a2 1c0
1c1 1c2 1c3 2c8 2cc 2d0 2d1 2bc 2b4 2b0 2b1 2b2 2b7 2e0 2e4 2de 2193 2191 2192 2197 2198 27 329 27 1d7b 27 303
27 2d 27 31e 1d5d 2a8 2a6 169 129 2a3 2a5 25 2b 5d 5c 28 29 5b
"""
symbols: List[str] = [chr(int(c, 16)) for c in _CODEPOINTS.split()]      # [pad] + punctuation + letters + IPA (symbols.py:5-14)
SPACE_ID = symbols.index(" ")                                              # symbols.py:17

_symbol_to_id = {s: i for i, s in enumerate(symbols)}                      # later duplicates win, as in text/__init__.py:6
_id_to_symbol = {i: s for i, s in enumerate(symbols)}


def intersperse(lst: Sequence[int], item: int) -> List[int]:
    """utils/utils.py:131-135: ``item`` before, between and after the elements (the blank token of ``add_blank``)."""
    result = [item] * (len(lst) * 2 + 1)
    result[1::2] = lst
    return result


def cleaned_text_to_sequence(cleaned_text: str) -> List[int]:
    """text/__init__.py:27-35.  An unknown symbol is a ``KeyError``, as in the reference."""
    return [_symbol_to_id[symbol] for symbol in cleaned_text]


def sequence_to_text(sequence: Sequence[int]) -> str:
    """text/__init__.py:38-44."""
    return "".join(_id_to_symbol[int(i)] for i in sequence)


# ---- cleaners (cleaners.py), the parts that do not need espeak-ng ----------------------------------------------------
_whitespace_re = re.compile(r"\s+")
_ELLIPSIS = "ELLIPSIS_MARKER"


def _abbr(pairs, raw: bool = False):
    """``\b<abbr>\.`` patterns, case-insensitive (cleaners.py:72-133).  The abbreviation text goes into the pattern UNESCAPED, as in
    the reference: the dots inside "m.", "z.b", "d.h", "u.a", "u.u", "u.v.m" are regex wildcards there, and list order decides
    ("z." fires before "z.b." can)."""
    return [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in pairs]


_ABBREVIATIONS = {
    "en": _abbr((("mrs", "misess"), ("ms", "miss"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"),
                 ("jr", "junior"), ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
                 ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
                 ("col", "colonel"), ("ft", "fort"))),
    "fr": _abbr((("m.", "monsieur"), ("dr", "docteur"), ("st", "saint"))),                                   # cleaners.py:103-110
    "de": _abbr((("hr", "herr"), ("fr", "frau"), ("dr", "doktor"), ("prof", "professor"), ("bsp", "beispiel"),  # cleaners.py:113-132
                 ("usw", "und so weiter"), ("z", "zu"), ("z.b", "zum beispiel"), ("ca", "zirka"), ("bzw", "beziehungsweise"),
                 ("d.h", "das heißt"), ("u.a", "unter anderem"), ("u.u", "unter umständen"), ("u.v.m", "und vieles mehr"),
                 ("vgl", "vergleiche"))),
}


def _currency_rules(dollar: str, euro: str, yen: str, joiner: str, cents: str, yen_cents: str):
    """"5.45$" -> "5 <dollar> <joiner> 45 <cents>" (the fr / de tables put the sign AFTER the amount, cleaners.py:180-182, :198-200)."""
    return [(re.compile(r"(\d+)\.(\d+)\$"), r"\1 %s %s \2 %s" % (dollar, joiner, cents)),
            (re.compile(r"(\d+)\.(\d+)€"), r"\1 %s %s \2 %s" % (euro, joiner, cents)),
            (re.compile(r"(\d+)\.(\d+)¥"), r"\1 %s %s \2 %s" % (yen, joiner, yen_cents))]


def _continental(dot: str, comma: str, euro: str, yen: str, mme: str, mlle: str, equals: str, slash: str, minus: str, currency):
    """The shared shape of the fr and de tables (cleaners.py:176-193, :195-212); order matters."""
    return ([(re.compile(r"\.\.\."), _ELLIPSIS), (re.compile(r"\("), ""), (re.compile(r"\)"), "")] + currency + [
        (re.compile(r"(?<=\D)\.(?=\D)(?!\s)", re.IGNORECASE), dot),
        (re.compile(r"(?<=\d)\,(?=\d)(?!\s)"), comma),
        (re.compile(r"€"), euro), (re.compile(r"¥"), yen),
        (re.compile(r"Mme"), mme), (re.compile(r"Mlle"), mlle),         # (never fire behind lowercase(): kept, as in the reference)
        (re.compile(r"="), equals), (re.compile(r"/"), slash),
        (re.compile(r"-(?=\d)(?!\s)"), minus),
        (re.compile(_ELLIPSIS), "..."),
    ])


_REPLACEMENTS = {
    "en": [                                                  # cleaners.py:161-174, order matters
        (re.compile(r"\.\.\."), _ELLIPSIS),
        (re.compile(r"\$(\d+)\.(\d+)"), r"\1 dollars and \2 cents"),
        (re.compile(r"€(\d+)\.(\d+)"), r"\1 euros and \2 cents"),
        (re.compile(r"¥(\d+)\.(\d+)"), r"\1 yen and \2 cents"),
        (re.compile(r"(?<=\D)\.(?=\D)(?!\s)", re.IGNORECASE), " dot "),
        (re.compile(r"(?<=\d)\.(?=\d)(?!\s)"), " point "),
        (re.compile(r"\$(\d+)"), r"\1 dollars"),
        (re.compile(r"€(\d+)"), r"\1 euros"),
        (re.compile(r"¥(\d+)"), r"\1 yen"),
        (re.compile(_ELLIPSIS), "..."),
    ],
    "ja": [                                                  # cleaners.py:135-147
        (re.compile(r"(?<!\s)\.(?!\s)"), " てん"), (re.compile(r"-(?=\d)"), " えん"), (re.compile(r"%"), " パーセント"),
        (re.compile(r"@"), " アットマーク"), (re.compile(r"\\\\"), " バックスラッシュ"), (re.compile(r"/"), " スラッシュ"),
        (re.compile(r"\$"), " ドル"), (re.compile(r"€"), " ユーロ"), (re.compile(r"¥"), " えん"), (re.compile(r"\+"), " プラス"),
        (re.compile(r"="), " イコール"),
    ],
    "fr": _continental(" point ", " vergule ", " euros", " yen", "madame", "mademoiselle", " égales ", " slash ", "négatif ",
                       _currency_rules("dollars", "euros", "yen", "et", "centimes", "centimes")),
    "de": _continental(" Punkt ", " Komma ", " Euro", " Yen", "Frau", "Fräulein", " gleich ", " Schrägstrich ", "minus ",
                       _currency_rules("Dollar", "Euro", "Yen", "und", "Cent", "Sen")),
}


def lowercase(text: str) -> str:
    return text.lower()


def collapse_whitespace(text: str) -> str:
    return re.sub(_whitespace_re, " ", text)


def expand_abbreviations(text: str, language: str) -> str:
    """cleaners.py:219-228.  A language without a table ("es", "ja") is the reference's ``UnboundLocalError``."""
    if language not in _ABBREVIATIONS:
        raise UnboundLocalError("local variable 'abbv' referenced before assignment (no abbreviation table for %r, cleaners.py:219-228)" % language)
    for regex, replacement in _ABBREVIATIONS[language]:
        text = re.sub(regex, replacement, text)
    return text


def apply_replacements(text: str, language: str) -> str:
    """cleaners.py:205-217."""
    if language not in _REPLACEMENTS:
        raise UnboundLocalError("local variable 'replacements' referenced before assignment (no replacement table for %r, cleaners.py:205-217)" % language)
    for regex, replacement in _REPLACEMENTS[language]:
        text = regex.sub(replacement, text)
    return text


def expand_abbreviations_en(text: str) -> str:
    return expand_abbreviations(text, "en")


def apply_replacements_en(text: str) -> str:
    return apply_replacements(text, "en")


def basic_cleaners(text: str) -> str:
    """cleaners.py:242-246."""
    return collapse_whitespace(lowercase(text))


def _pre_phonemiser(text: str, language: str) -> str:
    """What english_cleaners2 / french_cleaners / german_cleaners do BEFORE the phonemiser (cleaners.py:248-279); japanese_cleaners
    (:281-289) only applies its replacement table (no lowercase, no abbreviations)."""
    text = text.encode("utf-8").decode("utf-8")
    if language == "ja":
        return apply_replacements(text, "ja")
    return apply_replacements(expand_abbreviations(lowercase(text), language), language)


def english_cleaners2_pre(text: str) -> str:
    return _pre_phonemiser(text, "en")


_NO_PHONEMISER = ("%s needs a phonemiser (phonemizer + espeak-ng / misaki, cleaners.py:26-61), which this image does not have: pass "
                  "phonemize=..., or feed pre-phonemised text to cleaned_text_to_sequence")


def _language_cleaner(name: str, language: str, espeak: str):
    def cleaner(text: str, phonemize: Optional[Callable[[str], str]] = None) -> str:
        if phonemize is None:
            raise RuntimeError(_NO_PHONEMISER % name)
        return collapse_whitespace(phonemize(_pre_phonemiser(text, language)))
    cleaner.__name__ = name
    cleaner.__doc__ = ("cleaners.py: ``phonemize(text) -> IPA string`` stands for the reference's %s backend "
                       "(preserve_punctuation, with_stress, language_switch='remove-flags'); there is no fallback." % espeak)
    return cleaner


english_cleaners2 = _language_cleaner("english_cleaners2", "en", "EspeakBackend('en-us')")     # cleaners.py:248-257
french_cleaners = _language_cleaner("french_cleaners", "fr", "EspeakBackend('fr-fr')")         # cleaners.py:259-268
german_cleaners = _language_cleaner("german_cleaners", "de", "EspeakBackend('de')")            # cleaners.py:270-279
japanese_cleaners = _language_cleaner("japanese_cleaners", "ja", "misaki ja.JAG2P()")          # cleaners.py:281-289


def spanish_cleaners(text: str, phonemize: Optional[Callable[[str], str]] = None) -> str:
    """cleaners.py:291-300 is broken in the reference (no "es" abbreviation table): the same error, not a guess."""
    return expand_abbreviations(lowercase(text.encode("utf-8").decode("utf-8")), "es")


_CLEANERS = {"basic_cleaners": basic_cleaners, "english_cleaners2": english_cleaners2, "french_cleaners": french_cleaners,
             "german_cleaners": german_cleaners, "japanese_cleaners": japanese_cleaners, "spanish_cleaners": spanish_cleaners}
_PHONEMISING = frozenset(_CLEANERS) - {"basic_cleaners"}
CLEANER_BY_LANGUAGE = {"en": "english_cleaners2", "fr": "french_cleaners", "ja": "japanese_cleaners", "es": "spanish_cleaners",
                       "de": "german_cleaners"}                         # process_text, feel_me.py:135-141


def text_to_sequence(text: str, cleaner_names: Sequence[str], phonemize: Optional[Callable[[str], str]] = None) -> Tuple[List[int], str]:
    """text/__init__.py:10-24: (ids, cleaned text)."""
    for name in cleaner_names:
        if name not in _CLEANERS:
            raise Exception("Unknown cleaner: %s" % name)
        text = _CLEANERS[name](text, phonemize) if name in _PHONEMISING else _CLEANERS[name](text)
    return cleaned_text_to_sequence(text), text


def process_text_ids(text: str, language: str = "en", phonemize: Optional[Callable[[str], str]] = None) -> List[int]:
    """The id half of ``process_text`` (feel_me.py:135-147, cli.py:33-59): language -> cleaner -> ids -> ``intersperse(ids, 0)``.
    An unknown language is a ``ValueError`` (the reference prints a message and exits the process)."""
    if language not in CLEANER_BY_LANGUAGE:
        raise ValueError("Invalid language. Current supported languages: en (English), fr (French), ja (Japanese), de (German).")
    return intersperse(text_to_sequence(text, [CLEANER_BY_LANGUAGE[language]], phonemize)[0], 0)


def process_phonemes(phonemes: str, add_blank: bool = True) -> List[int]:
    """ids of a pre-phonemised utterance the way ``process_text`` builds them (cli.py:52-56: ``intersperse(ids, 0)``)."""
    ids = cleaned_text_to_sequence(phonemes)
    return intersperse(ids, 0) if add_blank else ids
