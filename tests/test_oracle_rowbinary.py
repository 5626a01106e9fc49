"""oracle/rowbinary.py against the byte vectors of the reference's own tests
(crates/etl-destinations/src/clickhouse/encoding.rs:386-470) and the CDC columns of core.rs:96-114."""
import struct
import uuid

import pytest

from oracle import rowbinary as RB


def test_reference_scalar_vectors():
    assert RB.value(("Bool", True)) == bytes([1])                               # :391-393
    assert RB.value(("I32", -1)) == struct.pack("<i", -1)                       # :396
    assert RB.value(("String", b"hi")) == bytes([2, ord("h"), ord("i")])        # :404  varint(2) + bytes
    assert RB.value(("Date", RB.CE_DAYS_1970 + 1)) == struct.pack("<i", 1)      # :408
    assert RB.value(("Date", RB.CE_DAYS_1970 - 1)) == struct.pack("<i", -1)     # :412  1969-12-31
    assert RB.value(("Date", RB.CE_DAYS_1970)) == struct.pack("<i", 0)          # :327-331
    assert RB.value(("Timestamp", RB.CE_DAYS_1970, 0, 0)) == struct.pack("<q", 0)   # :358-364


def test_reference_date_range():                                                # :349-355
    import datetime as dt
    for d in (dt.date(1899, 12, 31), dt.date(2300, 1, 1)):
        with pytest.raises(RB.ConversionError):
            RB.value(("Date", d.toordinal()))
    for d in (dt.date(1900, 1, 1), dt.date(2299, 12, 31)):
        assert RB.value(("Date", d.toordinal())) == struct.pack("<i", (d - dt.date(1970, 1, 1)).days)
    assert dt.date(1970, 1, 1).toordinal() == RB.CE_DAYS_1970      # chrono num_days_from_ce == Python's proleptic ordinal


def test_reference_uuid_wire_format():                                          # :417-430
    u = uuid.UUID("550e8400-e29b-41d4-a716-446655440000")
    hi, lo = struct.unpack(">QQ", u.bytes)
    assert RB.value(("Uuid", u.bytes)) == struct.pack("<Q", hi) + struct.pack("<Q", lo)


def test_reference_nullable_and_varint_and_hex():
    assert RB.nullable(("Null",)) == bytes([1])                                 # :436
    assert RB.nullable(("I32", 42)) == bytes([0]) + struct.pack("<i", 42)       # :440-443
    assert [RB.varint(v) for v in (0, 127, 128, 300)] == [b"\x00", b"\x7f", b"\x80\x01", b"\xac\x02"]   # :449-462
    assert RB.value(("Bytes", bytes([0xde, 0xad, 0xbe, 0xef]))) == RB.string(b"deadbeef")   # :373-383, :466-470
    assert RB.value(("Bytes", b"")) == b"\x00"
    with pytest.raises(RB.ConversionError):                                     # :476-488 NULL in a non-nullable column
        RB.row([("Null",)], [False])
    with pytest.raises(RB.ConversionError):                                     # :492-503 width mismatch
        RB.row([("I32", 1)], [False, False])


def test_cdc_columns():
    # core.rs:104-112: MergeTree = String(operation) + UInt64(commit lsn); ReplacingMergeTree = UInt128 version + UInt8 deleted
    assert RB.cdc_columns("U", 0x1122334455667788, 5, RB.MERGE_TREE) == b"\x06UPDATE" + struct.pack("<Q", 0x1122334455667788)
    v = (0x1122334455667788 << 64) | 5
    assert RB.cdc_columns("D", 0x1122334455667788, 5, RB.REPLACING_MERGE_TREE) == v.to_bytes(16, "little") + b"\x01"
    assert RB.cdc_columns("I", 1, 0, RB.REPLACING_MERGE_TREE)[-1] == 0


def test_time_display():
    assert RB.time_string(45045, 0) == "12:30:45"
    assert RB.time_string(45045, 120_000_000) == "12:30:45.120"
    assert RB.time_string(45045, 123_456_000) == "12:30:45.123456"
    assert RB.time_string(0, 1) == "00:00:00.000000001"


def test_reference_default_cell_and_key_row_expansion():
    """default_cell_string_mapped_values_are_strings (clickhouse/core.rs:1930-1937): money / timetz / interval default to an empty
    String, their array types to an empty array; expand_key_row_rejects_short_key_payload_before_identity_checks (:1719-1726): a key
    image of the wrong width is refused before the identity check. Then the tombstone row itself (:1437-1472): key cells in the
    primary-key columns, NULL where the source column is nullable and not an array, the zero value otherwise."""
    import pytest
    MONEY, TIMETZ, INTERVAL, MONEY_A, TIMETZ_A, INTERVAL_A = 790, 1266, 1186, 791, 1270, 1187
    for oid in (MONEY, TIMETZ, INTERVAL):
        assert RB.default_cell(oid) == ("String", b"")
    for oid in (MONEY_A, TIMETZ_A, INTERVAL_A):
        assert RB.default_cell(oid) == ("EmptyArray",)
    with pytest.raises(RB.HostRow, match="does not match the source primary key"):
        RB.expand_key_row([], [("id", 23, False, 1), ("name", 25, True, 0)], "AlternativeKey")
    with pytest.raises(RB.HostRow, match="primary-key or full replica identity"):
        RB.expand_key_row([("I32", 1)], [("id", 23, False, 1), ("name", 25, True, 0)], "AlternativeKey")
    cols = [("a", 25, True, 0), ("id", 23, False, 1), ("b", 25, False, 0), ("c", 1007, True, 0), ("d", 1184, False, 0),
            ("e", 2950, False, 0), ("f", 1082, False, 0), ("g", 16, False, 0), ("h", 1700, False, 0)]
    row = RB.expand_key_row([("I32", 7)], cols, "PrimaryKey")
    assert row == [("Null",), ("I32", 7), ("String", b""), ("EmptyArray",), ("Timestamp", RB.CE_DAYS_1970, 0, 0), ("Uuid", bytes(16)),
                   ("Date", RB.CE_DAYS_1970), ("Bool", False), ("String", b"")]
    # as bytes: Nullable(String) NULL, Int32 7, String "", Array() , DateTime64 0, UUID nil, Date32 0, Bool false, String ""
    assert RB.row(row, [1, 0, 0, 0, 0, 0, 0, 0, 0]) == b"\x01" + b"\x07\0\0\0" + b"\0" + b"\0" + bytes(8) + bytes(16) + bytes(4) + b"\0" + b"\0"
