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
    # fr / de / ja pre-phonemiser pipelines and the broken es one (cleaners.py:103-147, 176-228, 259-300), phonemiser stand-ins = identity
    lang_raw = {
        "fr": ["M.. Dupont (Dr. X) paie 5.45€ et 3,14 = a/b -5 St. Jean... Mme Mlle", "Bonjour   le monde.fr 10.50$ 7.25¥ ¥ €", "M. X et M.Y"],
        "de": ["Hr. Müller, z.B. 5.45$ usw. d.h. 3,5 = x/y -7 (ok) Prof. u.a. z. B.", "Fr. Dr. Bsp. ca. bzw. u.U. u.v.m. vgl. 9.99€ 1.5¥ Mme",
               "Guten   Tag.de ... 2,5"],
        "ja": ["3.14 -5 100% a@b.c 1/2 $5 €6 ¥7 1+1=2 x\\\\y", "こんにちは . 世界.です", "A-B -1"],
    }
    fn = {"fr": C.french_cleaners, "de": C.german_cleaners, "ja": C.japanese_cleaners}
    lang_cases = []
    for lang, texts in lang_raw.items():
        for s in texts:
            rec = {"language": lang, "text": s, "cleaned_identity_phonemiser": fn[lang](s), "apply_replacements": C.apply_replacements(s, lang)}
            if lang != "ja":
                rec["expand_abbreviations"] = C.expand_abbreviations(s.lower(), lang)
            lang_cases.append(rec)
    try:
        C.spanish_cleaners("hola")
        es_error = None
    except Exception as e:  # noqa: BLE001
        es_error = type(e).__name__
    out = {"language_cleaners": lang_cases, "spanish_cleaners_error": es_error, "symbols_codepoints": [ord(c) for c in symbols], "n_symbols": len(symbols), "space_id": SPACE_ID,
           "apostrophe_id": T._symbol_to_id["'"], "cases": cases, "cleaners": cleaned,
           "intersperse": [[[], 0, intersperse([], 0)], [[5], 0, intersperse([5], 0)], [[1, 2, 3], 9, intersperse([1, 2, 3], 9)]]}
    path = os.path.join(HERE, "text_vectors.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    print("wrote", path, len(cases), "phoneme cases,", len(cleaned), "cleaner cases")


if __name__ == "__main__":
    main()
