#!/usr/bin/env python3
"""Golden vectors for the symbol-table half of the text front end, generated from the REFERENCE's own
``matcha/text/{__init__,symbols,cleaners}.py`` and ``matcha/utils/utils.py`` imported unmodified.

Runs only in the build container (needs /root/reference).  The packages cleaners.py imports at module level and that
are absent offline get inert stand-ins: ``unidecode``, ``misaki`` and ``phonemizer`` — whose ``EspeakBackend.phonemize``
stand-in returns its input unchanged, so ``english_cleaners2`` exercises every line of the reference around the
phonemiser call (lowercase, abbreviations, replacements, whitespace) but no phonemisation.  Output:
tests/golden/text_vectors.json (data only).
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (the inert stand-ins for hydra / lightning ... that matcha.utils pulls in)


def main():
    MG.install_standins()

    class EspeakBackend:
        def __init__(self, *a, **k):
            pass

        def phonemize(self, texts, strip=True, njobs=1):
            return list(texts)

    MG._mod("phonemizer", backend=types.SimpleNamespace(EspeakBackend=EspeakBackend))
    MG._mod("unidecode", unidecode=lambda s: s)
    MG._mod("misaki", ja=types.SimpleNamespace(JAG2P=lambda: (lambda t: (t, None))))
    sys.path.insert(0, MG.REF)
    from matcha import text as T  # reference, unmodified
    from matcha.text import cleaners as C
    from matcha.text.symbols import SPACE_ID, symbols
    from matcha.utils.utils import intersperse

    ipa = [
        "həlˈoʊ wˈɜːld",
        "ðə kwˈɪk bɹˈaʊn fˈɑːks dʒˈʌmps ˌoʊvɚ ðə lˈeɪzi dˈɔɡ.",
        "wˈʌt?! nˈoʊ, ɹˈiəli: jˈɛs; «ˈoʊkˈeɪ» — ɔːlɹˈaɪt…",
        "ˈaɪm sˈoʊ hˈæpi tə sˈiː juː!",
        "ɪts 'kwoʊtᵻd' ænd ɐ-bˈaʊt",
        "nˈaɪs",
        "",
        "a",
    ]
    cases = []
    for s in ipa:
        ids = T.cleaned_text_to_sequence(s)
        cases.append({"phonemes": s, "ids": ids, "ids_blank": intersperse(ids, 0), "roundtrip": T.sequence_to_text(ids)})
    raw = ["Hello   World", "Dr. Smith paid $5.45 to Mr. Jones... OK", "see example.com at 3.14 or €7 and ¥20", "  tabs\tand\nnewlines  ",
           "St. Mary met Capt. Kirk, Esq. and Lt. Dan", "Prices: $100 now.Later"]
    cleaned = []
    for s in raw:
        en2 = C.english_cleaners2(s)                                  # phonemiser stand-in = identity
        seq, ct = T.text_to_sequence(s, ["basic_cleaners"]) if all(ch in T._symbol_to_id for ch in C.basic_cleaners(s)) else (None, None)
        cleaned.append({"text": s, "basic": C.basic_cleaners(s), "english_cleaners2_identity_phonemiser": en2,
                        "expand_abbreviations_en": C.expand_abbreviations(s.lower(), "en"),
                        "apply_replacements_en": C.apply_replacements(s.lower(), "en"), "basic_seq": seq})
    out = {"symbols_codepoints": [ord(c) for c in symbols], "n_symbols": len(symbols), "space_id": SPACE_ID,
           "apostrophe_id": T._symbol_to_id["'"], "cases": cases, "cleaners": cleaned,
           "intersperse": [[[], 0, intersperse([], 0)], [[5], 0, intersperse([5], 0)], [[1, 2, 3], 9, intersperse([1, 2, 3], 9)]]}
    path = os.path.join(HERE, "text_vectors.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    print("wrote", path, len(cases), "phoneme cases,", len(cleaned), "cleaner cases")


if __name__ == "__main__":
    main()
