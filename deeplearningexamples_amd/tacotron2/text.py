"""Text front end of the Tacotron2 input pipeline (host side): characters / ARPAbet -> symbol ids.

Restates tacotron2/text/__init__.py:15-76 (text_to_sequence, sequence_to_text), text/symbols.py:10-19 (the 148-symbol table: pad,
'-', punctuation, letters, '@'-prefixed ARPAbet) and text/cleaners.py:46-91 of the reference (which vendors keithito/tacotron).
`basic_cleaners` is complete.  `english_cleaners` / `transliteration_cleaners` lean on two things that are not restated here: the
reference's 48 KB transliteration tables for non-ASCII text (text/unidecoder/) and the third-party `inflect` package that spells
numbers out (absent from this image).  ASCII, digit-free text goes through the same lowercase / abbreviation / whitespace steps
(pinned by tests/golden/tacotron2_frontend.npz); text that would need either raises and says which.
"""
import re

_pad, _special, _punctuation = "_", "-", "!'(),.:;? "
_letters = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_vowels = ["AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW"]
_consonants = ["B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH", "V", "W", "Y",
               "Z", "ZH"]
# CMUdict's 84 phoneme symbols: every vowel bare and with the stress digits 0 / 1 / 2, in alphabetical order with the consonants
valid_symbols = sorted(_consonants + [v + s for v in _vowels for s in ("", "0", "1", "2")])
symbols = [_pad] + list(_special) + list(_punctuation) + list(_letters) + ["@" + s for s in valid_symbols]
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_id_to_symbol = dict(enumerate(symbols))

_curly = re.compile(r"(.*?)\{(.+?)\}(.*)")
_whitespace = re.compile(r"\s+")
_abbreviations = [(re.compile(r"\b%s\." % a, re.IGNORECASE), b) for a, b in [
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"), ("maj", "major"),
    ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"), ("hon", "honorable"), ("sgt", "sergeant"),
    ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort")]]


def _ascii(text):
    if not text.isascii():
        raise ValueError("non-ASCII text: the reference transliterates it with its text/unidecoder tables, not restated here: %r" % text)
    return text


def basic_cleaners(text):
    return _whitespace.sub(" ", text.lower())


def transliteration_cleaners(text):
    return _whitespace.sub(" ", _ascii(text).lower())


def english_cleaners(text):
    text = _ascii(text).lower()
    if re.search(r"[0-9]", text):
        raise ValueError("this text contains numbers to spell out (the reference uses the inflect package, absent here): %r" % text)
    for rx, full in _abbreviations:
        text = rx.sub(full, text)
    return _whitespace.sub(" ", text)


CLEANERS = dict(basic_cleaners=basic_cleaners, transliteration_cleaners=transliteration_cleaners, english_cleaners=english_cleaners)


def _ids(syms):
    return [_symbol_to_id[s] for s in syms if s in _symbol_to_id and s not in ("_", "~")]


def _clean(text, cleaner_names):
    for name in cleaner_names:
        if name not in CLEANERS:
            raise Exception("Unknown cleaner: %s" % name)
        text = CLEANERS[name](text)
    return text


def text_to_sequence(text, cleaner_names):
    """String -> list of symbol ids; `{HH AW1 S}` spans are ARPAbet (text/__init__.py:15-42)."""
    seq = []
    while len(text):
        m = _curly.match(text)
        if not m:
            seq += _ids(_clean(text, cleaner_names))
            break
        seq += _ids(_clean(m.group(1), cleaner_names))
        seq += _ids(["@" + s for s in m.group(2).split()])
        text = m.group(3)
    return seq


def sequence_to_text(sequence):
    out = ""
    for i in sequence:
        s = _id_to_symbol.get(int(i))
        if s is not None:
            out += "{%s}" % s[1:] if len(s) > 1 and s[0] == "@" else s
    return out.replace("}{", " ")
