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
