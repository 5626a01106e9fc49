"""The oracle's table-copy row parser against the reference's own known-answer tests
(crates/etl/src/postgres/codec/table_row.rs:287-633), transcribed case by case."""
import numpy as np
import pytest

from etl_amd import abi
from oracle import oracle

INT4, TEXT, BOOL, FLOAT8 = 23, 25, 16, 701
BASIC = [("id", INT4, False, 1), ("name", TEXT, True, 0), ("active", BOOL, False, 0)]   # create_test_column_schemas :275


def run(cols, row, mode=oracle.MODE_FULL):
    o = oracle.Oracle(mode=mode)
    o.schema_put(42, 0, cols)
    slot = o.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    assert slot >= 0
    buf = np.frombuffer(bytes(row), dtype=np.uint8)
    b = o.copy_decode(slot, buf, np.array([0, len(row)], dtype=np.uint32))
    return b


def row_of(cols, row):
    """Cells of the single decoded row, materialised from the canonical arena (CONTRACT mode)."""
    b = run(cols, row, mode=oracle.MODE_CONTRACT)
    assert b.err_code == 0, (b.err_code, b.err_desc)
    return b.host_batch().materialize()[0]["row"]


def S(t):
    return ("String", t.encode() if isinstance(t, str) else t)


def single(typ):
    return [("value", typ, False, 0)]   # create_single_column_schema :283


import struct
F64 = lambda x: ("F64", struct.unpack("<Q", struct.pack("<d", x))[0])
OK = [
    # (schema, row bytes, expected cells)                                                  table_row.rs line
    (BASIC, b"123\tJohn Doe\tt\n", [("I32", 123), S("John Doe"), ("Bool", True)]),          # :288 simple_row
    (BASIC, b"456\t\\N\tf\n", [("I32", 456), ("Null",), ("Bool", False)]),                   # :301 null values
    (BASIC, b"0\t\tf\n", [("I32", 0), S(""), ("Bool", False)]),                               # :314 empty strings
    (single(INT4), b"42\n", [("I32", 42)]),                                                   # :327 single column
    ([("a", INT4, False, 0), ("b", FLOAT8, False, 0), ("c", TEXT, False, 0), ("d", BOOL, False, 0)],
     b"123\t3.15\tHello World\tt\n", [("I32", 123), F64(3.15), S("Hello World"), ("Bool", True)]),   # :338
    (single(TEXT), b"Text\\\\\n", [S("Text\\")]),                                            # :447 trailing escape
    (single(TEXT), b"\\N\n", [("Null",)]),                                                    # :458 null marker
    (single(TEXT), b"\\\\N\n", [S("\\N")]),                                                  # :463 literal \N
    (single(TEXT), b"\\\\A\n", [S("\\A")]),                                                  # :468
    (BASIC, b"123\t John Doe \tt\n", [("I32", 123), S(" John Doe "), ("Bool", True)]),      # :477 whitespace preserved
    ([("c1", TEXT, False, 0), ("c2", TEXT, False, 0)], b"value\\twith\\ttabs\tnormal\\tvalue\n",
     [S("value\twith\ttabs"), S("normal\tvalue")]),                                         # :524 delimiter escaping
    ([("c1", TEXT, False, 0), ("c2", TEXT, False, 0), ("c3", TEXT, False, 0)], b"\\tstart\tmiddle\\nvalue\tend\\r\n",
     [S("\tstart"), S("middle\nvalue"), S("end\r")]),                                        # :539 escapes at field boundaries
    (single(TEXT), "Hello\\t\U0001F30D\\nWorld\\r\u6d4b\u8bd5\n".encode(),
     [S("Hello\t\U0001F30D\nWorld\r\u6d4b\u8bd5")]),                                       # :556 multibyte with escapes
    (single(TEXT), b"\n", [S("")]),                                                           # :620 empty string
]


@pytest.mark.parametrize("cols,row,want", OK)
def test_copy_row_ok(cols, row, want):
    assert row_of(cols, row) == want


ESCAPES = [  # try_from_postgres_escape_sequences :571-610
    (b"\\b\n", "\x08"), (b"\\f\n", "\x0c"), (b"\\n\n", "\n"), (b"\\r\n", "\r"), (b"\\t\n", "\t"), (b"\\v\n", "\x0b"),
    (b"\\\\\n", "\\"), (b"\\x\n", "x"), (b"\\1\n", "1"), (b"\\!\n", "!"), (b"\\@\n", "@"), (b'\\"\n', '"'),
    (b"value\\Ntail\n", "valueNtail"), (b"Text\\bwith\\bbackspaces\n", "Text\x08with\x08backspaces"),
    (b"Form\\ffeed\\ftest\n", "Form\x0cfeed\x0ctest"), (b"Vertical\\vtab\\vtest\n", "Vertical\x0btab\x0btest"),
    (b"Path\\\\to\\\\file.txt\n", "Path\\to\\file.txt"), (b"\\n\\n\\t\\t\\r\\r\n", "\n\n\t\t\r\r"),
    (b"Line1\\nTab:\\tBackslash:\\\\End\n", "Line1\nTab:\tBackslash:\\End"),
]


@pytest.mark.parametrize("row,want", ESCAPES)
def test_copy_escape_sequences(row, want):
    assert row_of(single(TEXT), row) == [S(want)]


ERR = [
    (single(INT4), b"42", abi.E_COPY_UNTERMINATED),                       # :358 not terminated
    (single(TEXT), b"text\\", abi.E_COPY_UNTERMINATED),                   # :371 trailing backslash, no terminator
    (BASIC, b"123\tJohn\n", abi.E_COPY_FEWER_COLS),                       # :384 column count mismatch
    (BASIC, b"123\tJohn\tt\textra\n", abi.E_COPY_MORE_COLS),              # :405 too many columns
    (single(TEXT), bytes([0xFF, 0xFE, 0xFD, 0x0A]), abi.E_UTF8),          # :427 invalid UTF-8
    (single(INT4), b"not_a_number\n", abi.E_INT),                         # :437 parsing error
    (BASIC, b"\t\t\n", abi.E_INT),                                        # :514 empty values: "" is not an int4
]


@pytest.mark.parametrize("cols,row,code", ERR)
def test_copy_row_errors(cols, row, code):
    b = run(cols, row)
    assert b.err_code == code and b.err_frame == 0 and b.n_events == 0
    assert b.err_kind == abi.ConversionError


def test_copy_large_row():   # :490
    cols = [(f"col{i}", INT4, False, 0) for i in range(50)]
    row = ("\t".join(str(i) for i in range(50)) + "\n").encode()
    assert row_of(cols, row) == [("I32", i) for i in range(50)]


def test_copy_newline_is_a_separator_that_marks_termination():
    """Not a reference test, but the reference's control flow (:112-128): '\\n' ends a field like a tab
    does and only sets the terminated flag, so text after it is parsed as further columns."""
    two = [("a", INT4, False, 0), ("b", INT4, False, 0)]
    assert row_of(two, b"1\n2\n") == [("I32", 1), ("I32", 2)]
    b = run(two, b"1\n2")
    assert b.err_code == abi.E_COPY_FEWER_COLS        # the dangling "2" is dropped (done = true), one column short
