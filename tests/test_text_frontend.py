"""Symbol-table half of the text front end against vectors generated from the reference's own
matcha/text modules (tests/golden/make_text_golden.py): the 198-entry table, cleaned_text_to_sequence /
sequence_to_text / intersperse, basic_cleaners, and english_cleaners2 around an identity phonemiser."""
import json
import os

import pytest

from emojivoice_amd import text as T

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(REPO, "tests", "golden", "text_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def test_symbol_table_is_the_references(vec):
    assert [ord(c) for c in T.symbols] == vec["symbols_codepoints"]
    assert len(T.symbols) == vec["n_symbols"] == 198 and len(set(T.symbols)) == 194
    assert T.SPACE_ID == vec["space_id"] == 16
    assert T._symbol_to_id["'"] == vec["apostrophe_id"]          # repeated symbol: the dict keeps the last position


def test_phonemes_to_ids(vec):
    for c in vec["cases"]:
        ids = T.cleaned_text_to_sequence(c["phonemes"])
        assert ids == c["ids"], c["phonemes"]
        assert T.intersperse(ids, 0) == c["ids_blank"]
        assert T.process_phonemes(c["phonemes"]) == c["ids_blank"]
        assert T.sequence_to_text(ids) == c["roundtrip"]
    for lst, item, want in vec["intersperse"]:
        assert T.intersperse(lst, item) == want
    with pytest.raises(KeyError):                                 # unknown symbol: KeyError, as in the reference
        T.cleaned_text_to_sequence("héllo 3")


def test_cleaners(vec):
    for c in vec["cleaners"]:
        s = c["text"]
        assert T.basic_cleaners(s) == c["basic"]
        assert T.expand_abbreviations_en(s.lower()) == c["expand_abbreviations_en"]
        assert T.apply_replacements_en(s.lower()) == c["apply_replacements_en"]
        assert T.english_cleaners2(s, phonemize=lambda t: t) == c["english_cleaners2_identity_phonemiser"]
        if c["basic_seq"] is not None:
            assert T.text_to_sequence(s, ["basic_cleaners"]) == (c["basic_seq"], c["basic"])
    with pytest.raises(RuntimeError):                             # no phonemiser, no guess
        T.english_cleaners2("hello")
    with pytest.raises(Exception):
        T.text_to_sequence("x", ["no_such_cleaner"])


def test_language_cleaners_and_switch(vec):
    """fr / de / ja pre-phonemiser pipelines (cleaners.py:103-147, 176-228, 259-289) and the language switch of
    ``process_text`` (feel_me.py:135-141) against vectors generated from the reference's own cleaners (identity phonemiser)."""
    ident = lambda t: t   # noqa: E731
    seen = set()
    for c in vec["language_cleaners"]:
        lang, s = c["language"], c["text"]
        seen.add(lang)
        cleaner = T._CLEANERS[T.CLEANER_BY_LANGUAGE[lang]]
        assert cleaner(s, ident) == c["cleaned_identity_phonemiser"], (lang, s)
        assert T.apply_replacements(s, lang) == c["apply_replacements"]
        if "expand_abbreviations" in c:
            assert T.expand_abbreviations(s.lower(), lang) == c["expand_abbreviations"]
        with pytest.raises(RuntimeError):
            cleaner(s)                                            # no phonemiser, no guess
    assert seen == {"fr", "de", "ja"}
    assert T.CLEANER_BY_LANGUAGE == {"en": "english_cleaners2", "fr": "french_cleaners", "ja": "japanese_cleaners",
                                     "es": "spanish_cleaners", "de": "german_cleaners"}
    assert vec["spanish_cleaners_error"] == "UnboundLocalError"   # broken in the reference (cleaners.py:296): same error here
    with pytest.raises(UnboundLocalError):
        T.text_to_sequence("hola", ["spanish_cleaners"], ident)
    with pytest.raises(ValueError):
        T.process_text_ids("hi", "xx", ident)
    # the switch feeds the right cleaner: a letters-only phonemiser stand-in keeps the ids inside the table
    keep = lambda t: "".join(ch for ch in t if ch in T._symbol_to_id)   # noqa: E731
    assert T.process_text_ids("Dr. X", "en", keep) == T.intersperse(T.cleaned_text_to_sequence("doctor x"), 0)
    assert T.process_text_ids("Dr. X", "fr", keep) == T.intersperse(T.cleaned_text_to_sequence("docteur x"), 0)
    assert T.process_text_ids("Dr. X", "de", keep) == T.intersperse(T.cleaned_text_to_sequence("doktor x"), 0)
