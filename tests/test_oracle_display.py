"""oracle/display.py (the Display impls the sinks write numeric / time / timetz cells with) against the reference's own
expectations: PgNumeric `from_str(..).to_string()` pairs of crates/etl-postgres/src/numeric.rs:694-696, 735-792, 850-943,
PgTimeTz of crates/etl-postgres/src/time.rs:231-258. The texts are parsed by the C++ oracle (itself pinned to the parse KATs of
the same files), so every pair below is parse + Display exactly as the reference's test states it."""
import numpy as np
import pytest

from oracle import display as D
from oracle import oracle
from tests import pgwire as W
from tests import scenarios as SC

# (text, to_string()) — numeric.rs
NUMERIC_DISPLAY = [
    ("1e-2", "0.01"), ("1.23e-2", "0.0123"), ("123e-2", "1.23"),                                 # :694-696 parse_scientific_notation
    ("0", "0"), ("0.0", "0.0"), ("000", "0"), ("000.000", "0.000"),                              # :755 zero_canonicalization_basic
    ("-0", "0"), ("-0.00", "0.00"),                                                               # :770 negative zero
    ("0e-1", "0.0"), ("0e-6", "0.000000"), ("0.00e-1", "0.000"),                                  # :790 zero_display_preserves_scale_from_exponent
    ("0.0012000", "0.0012000"),                                                                   # :864
    ("9999.9999", "9999.9999"), ("10000.0001", "10000.0001"),                                     # :880, :893
    ("0000120.00", "120.00"),                                                                     # :909
    ("1200000", "1200000"),                                                                       # :943
    ("NaN", "NaN"), ("Infinity", "Infinity"), ("-Infinity", "-Infinity"), ("inf", "Infinity"), ("-inf", "-Infinity"),   # Display arms :462-466
]
# roundtrip_stability (:920-936): printing is stable across parse -> print -> parse
NUMERIC_STABLE = ["120.00", "1.2000", "0.0120", "9999.9999", "10000.0001", "-120.00", "1200000"]
# struct literals: display_decimals :736-744, display_zero :747-750, the 120.0000 case :850-858
NUMERIC_STRUCTS = [((0, 0, 0, 2, (1234, 5000)), "1234.50"), ((0, 0, 0, 0, ()), "0")]
# time.rs:231-258
TIMETZ_DISPLAY = [("12:30:00.123+02", "12:30:00.123+02"), ("12:30:00-07:30", "12:30:00-07:30"), ("12:30:00+07:30:15", "12:30:00+07:30:15"),
                  ("00:00:00+15:59:59", "00:00:00+15:59:59")]


def _cells(type_oid, texts):
    o = oracle.Oracle()
    SC.simple_table([("id", SC.INT8, False, 1), ("v", type_oid, True, 0)])(o)
    s = SC.txn([W.insert(42, [str(i), t]) for i, t in enumerate(texts)])
    b = o.decode(np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets)
    assert b.err_code == 0, b.err_desc
    return [e["row"][1] for e in b.host_batch().materialize() if e["kind"] == "I"]


def test_numeric_display_kats():
    cells = _cells(SC.NUMERIC, [t for t, _ in NUMERIC_DISPLAY])
    for (text, want), c in zip(NUMERIC_DISPLAY, cells):
        assert c[0] == "Numeric", (text, c)
        assert D.numeric_string(*c[1:]) == want, text
    for fields, want in NUMERIC_STRUCTS:
        assert D.numeric_string(*fields) == want
    assert D.numeric_string(0, 0, 0, 4, (1200, 0)).endswith("0000")      # :850-858


def test_numeric_display_is_stable():
    first = [D.numeric_string(*c[1:]) for c in _cells(SC.NUMERIC, NUMERIC_STABLE)]
    again = _cells(SC.NUMERIC, first)
    assert [D.numeric_string(*c[1:]) for c in again] == first
    assert again == _cells(SC.NUMERIC, NUMERIC_STABLE)


def test_timetz_display_kats():
    for (text, want), c in zip(TIMETZ_DISPLAY, _cells(SC.TIMETZ, [t for t, _ in TIMETZ_DISPLAY])):
        assert c[0] == "TimeTz", (text, c)
        assert D.timetz_string(*c[1:]) == want, text
    assert D.utc_offset_string(-57_599) == "-15:59:59" and D.utc_offset_string(9_000) == "+02:30" and D.utc_offset_string(0) == "+00"


@pytest.mark.parametrize("secs,nanos,want", [(45045, 0, "12:30:45"), (45045, 500_000_000, "12:30:45.500"), (45045, 123_456_000, "12:30:45.123456"),
                                              (45045, 1, "12:30:45.000000001"), (86399, 1_000_000_000, "23:59:60"), (86399, 1_500_000_000, "23:59:60.500")])
def test_time_display(secs, nanos, want):
    assert D.time_string(secs, nanos) == want
