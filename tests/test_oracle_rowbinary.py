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
