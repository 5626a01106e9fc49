"""The oracle (oracle/) against the reference's own known-answer tests
(tests/golden/reference_kats.py, transcribed with file:line citations)."""
import ctypes as C

import pytest

from etl_amd import abi
from oracle import oracle
from tests.golden import reference_kats as K


def _check(oid, text, exp):
    got = oracle.parse_text_cell(oid, text)
    if exp == K.ERR:
        assert got.startswith("Err("), (oid, text, got)
    elif isinstance(exp, tuple):
        assert got == "Err(%d)" % exp[1], (oid, text, got)
    else:
        assert got == exp, (oid, text, got)


@pytest.mark.parametrize("name", ["TEXT_RS", "BOOL_RS", "HEX_RS", "TIME_RS", "PG_TIME_RS", "NUMERIC_RS"])
def test_reference_unit_test_vectors(name):
    for oid, text, exp in getattr(K, name):
        _check(oid, text, exp)


def test_fuzz_corpus_seeds():
    for seed, exp in K.FUZZ_SEEDS:
        _check(K.FUZZ_TYPES[seed[0] % len(K.FUZZ_TYPES)], seed[1:].decode(), exp)
    for t, exp in K.FUZZ_NUMERIC:
        _check(K.NUMERIC, t, exp)
    for t, exp in K.FUZZ_BYTEA:
        _check(K.BYTEA, t, exp)


def test_reject_list_from_type_matrix():
    for oid, text in K.REJECT_LIST:
        _check(oid, text, K.ERR)


def test_utc_offset_grammar():
    L = oracle.lib()
    for text, exp in K.UTC_OFFSETS:
        v = C.c_int32()
        ok = L.oracle_parse_utc_offset(text.encode(), len(text.encode()), C.byref(v))
        assert (v.value if ok else None) == exp, text


def test_numeric_max_shape():
    # crates/etl-postgres/src/numeric.rs:656-680
    text = "9" * ((32767 + 1) * 4) + "." + "9" * 16383
    got = oracle.parse_text_cell(K.NUMERIC, text)
    assert got.startswith("Numeric(+,w=32767,s=16383,[9999,")
    digits = got[got.index("[") + 1:got.index("]")].split(",")
    assert len(digits) == 36864 and digits[0] == "9999" and digits[-1] == "9990"


def test_error_table_matches_reference_strings():
    L = oracle.lib()
    for code, (kind, desc) in K.ERR_TABLE.items():
        assert L.oracle_err_kind(code) == kind, code
        assert L.oracle_err_description(code).decode() == desc, code


def test_invalid_utf8_is_rejected_before_type_parse():
    # codec/event.rs:976 — str::from_utf8 precedes the type switch
    for bad in [b"\xff", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"ab\xe2\x82"]:
        assert oracle.parse_text_cell(K.TEXT, bad) == "Err(%d)" % abi.E_UTF8
    assert oracle.parse_text_cell(K.TEXT, "héllo \U0001F914") == 'String("héllo \U0001F914")'
