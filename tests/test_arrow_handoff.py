"""Arena -> Arrow hand-off (etl_amd/arrow.py) against the reference's Cell -> Arrow mapping
(crates/etl-destinations/src/iceberg/encoding.rs:61-360): known answers for every fixed-width class, strings, bytes,
NULLs, and agreement with the per-cell materialiser on synthetic streams. Runs on the oracle's arena (byte-identical
to the HIP path's, which the -m gpu suite proves), so it needs no GPU."""
import datetime as dt
import struct
import uuid

import numpy as np
import pyarrow as pa
import pytest

from etl_amd import abi, synth
from etl_amd.arrow import rows_to_record_batch
from oracle import oracle
from tests import pgwire as W
from tests import scenarios as SC


def _decode(prime, msgs):
    o = oracle.Oracle()
    prime(o)
    s = SC.txn(msgs)
    b = o.decode(np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets)
    assert b.err_code == 0, b.err_desc
    return b.host_batch()


def test_known_answers_for_every_class():
    rows = [SC.alltypes_row(), SC.alltypes_row(id="2", b="f", i2="7", i4="-2147483648", o="4294967295", d="1969-12-31",
                                               t="00:00:00", ts="1969-12-31 23:59:59.5", tstz="2026-01-02 03:04:05+02",
                                               f8="1e300", f4="-0.5", s="", by="\\x"),
            [("3" if c[0] == "id" else W.NULL) for c in SC.ALLTYPES]]
    hb = _decode(SC.simple_table(SC.ALLTYPES), [W.insert(42, r) for r in rows])
    names = [c[0] for c in SC.ALLTYPES]
    with pytest.raises(NotImplementedError):
        rows_to_record_batch(hb, 0, names=names)           # jsonb / array columns are text-form
    keep = list(range(len(names)))
    rb = rows_to_record_batch(hb, 0, names=names, on_text="binary", columns=keep)
    assert rb.num_rows == 3 and rb.schema.names == [names[i] for i in keep]
    col = {n: rb.column(n).to_pylist() for n in rb.schema.names}
    typ = {n: rb.schema.field(n).type for n in rb.schema.names}
    assert typ["id"] == pa.int64() and col["id"] == [1, 2, 3]
    assert typ["b"] == pa.bool_() and col["b"] == [True, False, None]
    assert typ["i2"] == pa.int32() and col["i2"] == [-123, 7, None]                 # cell_to_i32: I16 widens
    assert typ["i4"] == pa.int32() and col["i4"] == [456, -2147483648, None]
    assert typ["o"] == pa.int64() and col["o"] == [42, 4294967295, None]            # cell_to_i64: U32 widens
    assert typ["d"] == pa.date32() and col["d"] == [dt.date(2026, 1, 2), dt.date(1969, 12, 31), None]
    assert typ["t"] == pa.time64("us") and col["t"] == [dt.time(12, 30, 45, 123456), dt.time(0, 0, 0), None]
    assert typ["ts"] == pa.timestamp("us")
    assert col["ts"] == [dt.datetime(2026, 1, 2, 3, 4, 5, 123456), dt.datetime(1969, 12, 31, 23, 59, 59, 500000), None]
    assert typ["tstz"] == pa.timestamp("us", tz="UTC")
    utc = dt.timezone.utc
    assert col["tstz"] == [dt.datetime(2026, 1, 2, 3, 4, 5, 123456, tzinfo=utc), dt.datetime(2026, 1, 2, 1, 4, 5, tzinfo=utc), None]
    assert typ["u"] == pa.binary(16) and col["u"][0] == uuid.UUID("123e4567-e89b-12d3-a456-426614174000").bytes and col["u"][2] is None
    assert typ["f8"] == pa.float64() and col["f8"] == [-7.25, 1e300, None]
    assert typ["f4"] == pa.float32() and col["f4"] == [3.5, -0.5, None]
    assert typ["s"] == pa.string() and col["s"] == ["hello wörld", "", None]
    assert typ["by"] == pa.large_binary() and col["by"] == [b"\x01\x02\xff", b"", None]
    # numeric / timetz: their Display strings (cell_to_string, encoding.rs:349-352)
    assert typ["n"] == pa.string() and col["n"] == ["12345.6789", "12345.6789", None]
    assert typ["tz"] == pa.string() and col["tz"] == ["12:30:45.123456+02", "12:30:45.123456+02", None]
    # text-form classes come back as their heap entries for the host to finish
    assert typ["j"] == pa.large_binary() and col["j"][0] == b'{"kind":"jsonb","nested":{"n":2}}' and col["j"][2] is None
    assert col["arr"][0] == b"{1,NULL,3}"
    assert not rb.schema.field("id").nullable and rb.schema.field("s").nullable


def _cell_py(c):
    k = c[0]
    if k == "Null":
        return None
    if k in ("Bool", "I16", "I32", "U32", "I64"):
        return c[1]
    if k == "String":
        return c[1].decode()
    if k == "Uuid":
        return c[1]
    if k == "TimestampTz":
        days, secs, nanos = c[1:]
        return ((days - 719163) * 86400 + secs) * 1_000_000 + nanos // 1000
    return c


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_agrees_with_the_per_cell_materialiser(mk):
    """Synthetic streams (fixed-width inserts; mixed I/U/D with TEXT / NUMERIC / timestamptz / uuid): every column the
    hand-off supports equals what materialize() yields cell by cell, inserts and full new rows of updates, in event order."""
    w = mk()
    o = oracle.Oracle()
    w.register(o)
    buf, offs = w.fill(256 << 10)
    b = o.decode(buf, offs)
    assert b.err_code == 0
    hb = b.host_batch()
    rb = rows_to_record_batch(hb, 0, kinds=("I", "U"), on_text="binary")
    ev = [e for e in hb.materialize() if e["kind"] == "I" or (e["kind"] == "U" and not e["partial"])]
    assert rb.num_rows == len(ev) > 100
    slot = hb.slots[0]
    for i, colspec in enumerate(slot.cols):
        if colspec.type_class == abi.TC_NUMERIC:
            from oracle import display as D
            assert rb.column(i).to_pylist() == [None if e["row"][i][0] == "Null" else D.numeric_string(*e["row"][i][1:]) for e in ev]
            continue
        want = [_cell_py(e["row"][i]) for e in ev]
        got = rb.column(i)
        if pa.types.is_timestamp(got.type):
            got = got.cast(pa.int64())
        assert got.to_pylist() == want, (i, colspec.type_class)


def test_empty_selection():
    hb = _decode(SC.simple_table(SC.COLS2), [])
    rb = rows_to_record_batch(hb, 0, names=["id", "payload"])
    assert rb.num_rows == 0 and rb.schema.names == ["id", "payload"]
