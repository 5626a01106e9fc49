"""Device-side columnar hand-off (etlg_batch_columns, etl_amd/csrc/columns.hip) against the host hand-off of the oracle's
arena (etl_amd.arrow.rows_to_record_batch — itself pinned to the reference's Cell -> Arrow mapping,
crates/etl-destinations/src/iceberg/encoding.rs:61-360, by tests/test_arrow_handoff.py): same rows, same order, same
values, validity and types, for every class; plus the raw buffers (offsets, bit-packed validity, row_event)."""
import os

import numpy as np
import pyarrow as pa
import pytest

from etl_amd import abi, synth
from etl_amd.arrow import columns_to_record_batch, rows_to_record_batch
from tests import pgwire as W
from tests import scenarios as SC

pytestmark = pytest.mark.gpu


def _both(prime, buf, offs):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    prime(o)
    prime(d)
    rb = o.decode(buf, offs)
    assert rb.err_code == 0, rb.err_desc
    b = d.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
    assert b.rc == 0, b.error
    return rb.host_batch(), b, d


def _same(want, got):
    assert want.num_rows == got.num_rows and want.schema.names == got.schema.names
    for name in want.schema.names:
        a, g = want.column(name), got.column(name)
        if pa.types.is_string(a.type):
            a = a.cast(pa.large_utf8())
        assert a.type == g.type, (name, a.type, g.type)
        assert want.schema.field(name).nullable == got.schema.field(name).nullable, name
        assert a.null_count == g.null_count, name
        assert a.equals(g), name


def _stream(msgs):
    s = SC.txn(msgs)
    return np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets


def test_every_class_known_answers():
    rows = [SC.alltypes_row(), SC.alltypes_row(id="2", b="f", i2="7", i4="-2147483648", o="4294967295", d="1969-12-31",
                                               t="00:00:00", ts="1969-12-31 23:59:59.5", tstz="2026-01-02 03:04:05+02",
                                               f8="1e300", f4="-0.5", s="", by="\\x"),
            [("3" if c[0] == "id" else W.NULL) for c in SC.ALLTYPES]]
    rows += [SC.alltypes_row(id=str(10 + i), s="x" * (i * 7 % 90), i4=str(i * 1001)) for i in range(150)]   # > 2 waves of rows
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    names = [c[0] for c in SC.ALLTYPES]
    cols = b.columns(0)
    assert cols.n_rows == len(rows) and cols.view.n_cols == len(names) and not cols.view.on_device
    # numeric / timetz: Display strings formatted on the device (cell_to_string, iceberg/encoding.rs:349-352)
    assert cols.column(names.index("tz")).arrow_kind == abi.AK_LARGE_UTF8 and cols.column(names.index("n")).arrow_kind == abi.AK_LARGE_UTF8
    with pytest.raises(NotImplementedError):
        columns_to_record_batch(cols, names=names)                       # json / array columns need on_text="binary"
    got = columns_to_record_batch(cols, names=names, on_text="binary")
    want = rows_to_record_batch(hb, 0, names=names, on_text="binary")
    _same(want, got)
    from oracle import display as D
    ev = [e for e in hb.materialize() if e["kind"] == "I"]
    assert got.column("n").to_pylist() == [None if e["row"][names.index("n")][0] == "Null" else D.numeric_string(*e["row"][names.index("n")][1:]) for e in ev]
    assert got.column("tz").to_pylist() == [None if e["row"][names.index("tz")][0] == "Null" else D.timetz_string(*e["row"][names.index("tz")][1:]) for e in ev]
    assert got.column("n").to_pylist()[:3] == ["12345.6789", "12345.6789", None] and got.column("tz").to_pylist()[0] == "12:30:45.123456+02"
    assert got.column("id").to_pylist()[:3] == [1, 2, 3] and got.column("b").to_pylist()[:3] == [True, False, None]
    assert got.column("s").to_pylist()[:3] == ["hello wörld", "", None]
    assert np.array_equal(cols.row_event(), np.flatnonzero(hb.kind == ord("I")).astype(np.uint64))
    cols.close(); b.close(); d.close()


@pytest.mark.parametrize("mk,nbytes", [(synth.cfg2, 1 << 20), (synth.cfg3, 2 << 20)])
@pytest.mark.parametrize("kinds", [("I",), ("I", "U"), ("U",)])
def test_synthetic_streams(mk, nbytes, kinds):
    if os.environ.get("ETLG_SIMT_RUN") == "1":
        nbytes = 192 << 10
    w = mk()
    buf, offs = w.fill(nbytes)
    hb, b, d = _both(w.register, buf, offs)
    cols = b.columns(0, kinds=kinds)
    want = rows_to_record_batch(hb, 0, kinds=kinds, on_text="binary")
    got = columns_to_record_batch(cols, on_text="binary", names=want.schema.names)
    _same(want, got)
    if mk is synth.cfg3 and "U" in kinds:
        assert want.num_rows > 100
    sel = cols.row_event()
    assert np.all(np.diff(sel.astype(np.int64)) > 0)          # event order
    cols.close(); b.close(); d.close()


def test_empty_and_foreign_slots():
    prime = SC.simple_table(SC.COLS2)
    buf, offs = _stream([])
    hb, b, d = _both(prime, buf, offs)
    cols = b.columns(0)
    rb = columns_to_record_batch(cols, names=["id", "payload"])
    assert rb.num_rows == 0 and rb.schema.names == ["id", "payload"]
    with pytest.raises(Exception):
        b.columns(7)                                           # no such slot
    hb2 = b.host()                                             # downloaded: the arena left the device
    with pytest.raises(Exception):
        b.columns(0)
    assert hb2.n_events == hb.n_events
    cols.close(); b.close(); d.close()


def test_deferred_cells_are_flagged():
    """Cells the kernels hand back DEFERRED (here: float8 texts the device rule does not settle): null in the validity of a
    fixed-width column and set in its `deferred` bitmap, exactly where the oracle's arena has state 3."""
    cols3 = [("id", 20, False, True), ("x", 701, True, False), ("y", 700, True, False)]
    vals = ["1.5", "0.1000000000000000055511151231257827021181583404541015625", "NaN", "3.141592653589793238462643383279",
            "1e-320", "2.5", "1.7976931348623157e308", "4.9e-324", "50537618.817359292015891086651596749e82",
            "107896223265412489690691363e88", "28879636596541978310003766487.741e-212", "5693107746173304490483329377e264"]
    msgs = [W.insert(42, [str(i), vals[i % len(vals)], vals[(i + 3) % len(vals)]]) for i in range(200)]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    c = b.columns(0)
    slot = hb.slots[0]
    base = hb.body_off[hb.kind == ord("I")].astype(np.int64)
    total_deferred = 0
    for i in (1, 2):
        st = (hb.fixed[base + i // 4] >> np.uint8(2 * (i % 4))) & 3
        validity, deferred, values, _ = c.host_arrays(i)
        vbits = np.unpackbits(validity, bitorder="little")[:len(base)].astype(bool)
        dbits = np.unpackbits(deferred, bitorder="little")[:len(base)].astype(bool)
        assert np.array_equal(vbits, st == abi.CELL_VALUE) and np.array_equal(dbits, st == abi.CELL_DEFERRED)
        assert c.column(i).deferred_count == int((st == abi.CELL_DEFERRED).sum())
        assert c.column(i).null_count == int((st != abi.CELL_VALUE).sum())
        total_deferred += int(c.column(i).deferred_count)
        w = 8 if i == 1 else 4
        so = base + slot.cols[i].off_full
        raw = hb.fixed[so[:, None] + np.arange(w)[None, :]]
        got = values.reshape(len(base), w)
        assert np.array_equal(got[vbits], raw[vbits])
    assert total_deferred > 0
    c.close(); b.close(); d.close()


def test_buffers_can_stay_on_the_device():
    w = synth.cfg3()
    buf, offs = w.fill(256 << 10)
    hb, b, d = _both(w.register, buf, offs)
    host = b.columns(0, kinds=("I", "U"))
    dev = b.columns(0, kinds=("I", "U"), on_device=True)
    assert dev.view.on_device == 1 and dev.n_rows == host.n_rows > 0
    emu = os.environ.get("ETLG_SIMT_RUN") == "1"

    def read(ptr, nbytes):
        if not nbytes:
            return np.zeros(0, np.uint8)
        if emu:
            import ctypes as C
            return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=np.uint8).copy()
        return abi.device_tensor(ptr, nbytes, 0).cpu().numpy()

    n = host.n_rows
    for i in range(host.view.n_cols):
        hk, dk = host.column(i), dev.column(i)
        assert (hk.arrow_kind, hk.null_count, hk.deferred_count, hk.values_bytes) == (dk.arrow_kind, dk.null_count, dk.deferred_count, dk.values_bytes)
        hv = host.host_arrays(i)
        assert np.array_equal(read(dk.validity, (n + 63) // 64 * 8), hv[0])
        assert np.array_equal(read(dk.values, int(dk.values_bytes)), hv[2])
        if hv[3] is not None:
            assert np.array_equal(read(dk.offsets, (n + 1) * 8).view(np.int64), hv[3])
    host.close(); dev.close(); b.close(); d.close()


ARR_COLS = [("id", SC.INT8, False, 1), ("a4", 1007, True, 0), ("a8", 1016, True, 0), ("a2", 1005, True, 0), ("ab", 1000, True, 0),
            ("ao", 1028, True, 0), ("at", 1009, True, 0)]


def _oracle_list(oid, text):
    """The oracle's parse of one array literal (parse_array_text, oracle_codec.hpp — text.rs:228-312) as a Python list."""
    from oracle import oracle
    r = oracle.parse_text_cell(oid, text)
    assert r.startswith("Array["), r
    body = r[6:-1]
    out = []
    for e in ([] if not body else body.split(",")):
        if e == "NULL":
            out.append(None)
        elif e.startswith("Bool("):
            out.append(e == "Bool(true)")
        else:
            out.append(int(e[e.index("(") + 1:-1]))
    return out


def test_array_literals_are_parsed_on_the_device():
    """bool[] / int2[] / int4[] / int8[] / oid[] columns come back as list columns whose rows equal the oracle's parse of the same
    literal: dimension prefixes, quotes, escapes, NULL (any case, unquoted only), empty arrays, NULL cells; text[] stays text."""
    lits4 = ["{1,NULL,3}", "{}", "[1:2]={1,2}", '{"1",2}', "{+5,-0}", "{-2147483648,2147483647}", '{"\\1",null,NuLl}', "[-1:0]={7,8}", "{0}"]
    lits8 = ["{9223372036854775807,-9223372036854775808}", "{}", "{NULL}", "{1}", '{"12"}']
    lits2 = ["{-32768,32767}", "{1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40}", "{}"]
    litsb = ["{t,f,NULL}", "{}", '{"t"}', "{f}"]
    litso = ["{0,4294967295}", "{NULL,7}", "{}"]
    rows = []
    for i in range(90):
        rows.append([str(i), lits4[i % len(lits4)], lits8[i % len(lits8)], lits2[i % len(lits2)], litsb[i % len(litsb)],
                     litso[i % len(litso)], '{a,"b c",NULL}'])
    rows.append(["900"] + [W.NULL] * 6)
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(ARR_COLS), buf, offs)
    names = [c[0] for c in ARR_COLS]
    cols = b.columns(0, parse_arrays=True)
    assert [cols.column(i).arrow_kind for i in range(1, 7)] == [abi.AK_LIST] * 6
    rb = columns_to_record_batch(cols, names=names, on_text="binary")
    assert rb.schema.field("a4").type == pa.large_list(pa.int32()) and rb.schema.field("a8").type == pa.large_list(pa.int64())
    assert rb.schema.field("a2").type == pa.large_list(pa.int32()) and rb.schema.field("ab").type == pa.large_list(pa.bool_())
    assert rb.schema.field("ao").type == pa.large_list(pa.int64())
    for ci, oid in ((1, 1007), (2, 1016), (3, 1005), (4, 1000), (5, 1028)):
        want = [None if r[ci] is W.NULL else _oracle_list(oid, r[ci]) for r in rows]
        assert rb.column(ci).to_pylist() == want, names[ci]
    assert rb.schema.field("at").type == pa.large_list(pa.large_utf8()) and rb.column(6).to_pylist()[0] == ["a", "b c", None]
    # without the flag the same columns are their source text, as before
    plain = b.columns(0)
    assert plain.column(1).arrow_kind == abi.AK_TEXT_FORM
    plain.close(); cols.close(); b.close(); d.close()


@pytest.mark.parametrize("lit,col", [("{1,{2}}", 1), ("{1,2", 1), ("{a}", 1), ("[1:2={1}", 1), ("[1:1][1:1]={{1}}", 1), ('{"1}', 1), ("{1\\}", 1),
                                     ("{2}", 4), ("{99999}", 3), ("{-1}", 5), ("}", 2), ("{ 1}", 2)])
def test_malformed_array_literals_fail_like_the_reference(lit, col):
    from etl_amd.decoder import EtlError
    from oracle import oracle
    good = ["1", "{1}", "{1}", "{1}", "{t}", "{1}", "{x}"]
    bad = list(good)
    bad[0], bad[col] = "2", lit
    buf, offs = _stream([W.insert(42, good), W.insert(42, bad), W.insert(42, good)])
    hb, b, d = _both(SC.simple_table(ARR_COLS), buf, offs)
    want = oracle.parse_text_cell(ARR_COLS[col][1], lit)
    assert want.startswith("Err("), want
    with pytest.raises(EtlError) as ei:
        b.columns(0, parse_arrays=True)
    assert ei.value.code == int(want[4:-1]) and ei.value.frame_index == 2
    o = oracle.Oracle()
    assert ei.value.description == o.L.oracle_err_description(ei.value.code).decode()
    b.close(); d.close()


def test_full_size_batch_columns_and_rowbinary():
    """BASELINE-size (64 MiB) cfg2 batch: the device-built columns equal the host gather over the SAME arena (whose parity with
    the oracle is test_gpu_parity's job), and the RowBinary buffer equals a vectorised restatement of the reference's row
    (5 x Int32 LE, _etl_version = tx_ordinal | commit_lsn << 64 as UInt128 LE, _etl_deleted UInt8)."""
    if os.environ.get("ETLG_SIMT_RUN") == "1":
        pytest.skip("full size: MI355X only")
    from etl_amd.decoder import Decoder
    w = synth.cfg2()
    d = Decoder(0)
    w.register(d)
    buf, offs = w.fill(64 << 20)
    b = d.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
    assert b.rc == 0
    cols = b.columns(0)
    rb = b.rowbinary(0, [0] * 5 + [0, 0], abi.CH_REPLACING_MERGE_TREE)
    got = columns_to_record_batch(cols)
    rows_ev = cols.row_event().copy()
    rb_bytes, rb_offs, rb_ev = rb.bytes().copy(), rb.row_offsets().copy(), rb.row_event().copy()
    hb = b.host()                                    # downloads the arena: the batch leaves the device here
    want = rows_to_record_batch(hb, 0)
    _same(want, got)
    ins = np.flatnonzero(hb.kind == ord("I"))
    assert len(ins) > 500_000 and np.array_equal(rows_ev, ins.astype(np.uint64)) and np.array_equal(rb_ev, rows_ev)
    n = len(ins)
    exp = np.zeros((n, 37), dtype=np.uint8)
    for c in range(5):
        exp[:, 4 * c:4 * c + 4] = np.ascontiguousarray(want.column(c).to_numpy().astype("<i4")).view(np.uint8).reshape(n, 4)
    exp[:, 20:28] = np.ascontiguousarray(hb.tx_ordinal[ins].astype("<u8")).view(np.uint8).reshape(n, 8)
    exp[:, 28:36] = np.ascontiguousarray(hb.commit_lsn[ins].astype("<u8")).view(np.uint8).reshape(n, 8)
    assert np.array_equal(rb_offs, np.arange(n + 1, dtype=np.int64) * 37)
    assert np.array_equal(rb_bytes.reshape(n, 37), exp)
    cols.close(); rb.close(); b.close(); d.close()


ARR2_COLS = [("id", SC.INT8, False, 1), ("f8", 1022, True, 0), ("f4", 1021, True, 0), ("d", 1182, True, 0), ("t", 1183, True, 0),
             ("ts", 1115, True, 0), ("tz", 1185, True, 0), ("u", 2951, True, 0), ("n", 1231, True, 0)]


def _oracle_elems(oid, text):
    """The oracle's parse of one array literal as Python values comparable with pyarrow's to_pylist()."""
    import datetime as dt
    import struct
    from oracle import oracle
    r = oracle.parse_text_cell(oid, text)
    assert r.startswith("Array["), r
    out = []
    body = r[6:-1]
    for e in ([] if not body else body.split(",")):
        k, _, v = e.partition("(")
        v = v[:-1]
        if e == "NULL":
            out.append(None)
        elif k in ("F64", "F32") and v == "NaN":
            out.append(float("nan"))
        elif k == "F64":
            out.append(struct.unpack("<d", struct.pack("<Q", int(v, 16)))[0])
        elif k == "F32":
            out.append(struct.unpack("<f", struct.pack("<I", int(v, 16)))[0])
        elif k == "Date":
            out.append(dt.date.fromisoformat(v))
        elif k == "Time":
            out.append(dt.time.fromisoformat(v[:15]))            # microseconds
        elif k == "Timestamp":
            out.append(dt.datetime.fromisoformat(v[:26]))
        elif k == "TimestampTz":
            out.append(dt.datetime.fromisoformat(v[:26]).replace(tzinfo=dt.timezone.utc))
        elif k == "Uuid":
            out.append(bytes.fromhex(v))
        else:
            raise AssertionError(e)
    return out


def test_float_temporal_and_uuid_arrays_on_the_device():
    import math
    lits = {
        1: ["{1.5,NULL,-0.25,1e300}", "{}", "{NaN,inf,-Infinity}", '{"3.141592653589793",0}'],
        2: ["{1.5,3.4028235e38}", "{NULL}", "{-0.5}"],
        3: ["{2026-01-02,1969-12-31}", "{}", "{NULL,0001-01-01}"],
        4: ["{12:30:45.123456,00:00:00}", "{23:59:59.5}"],
        5: ['{"2026-01-02 03:04:05.123456"}', '{"1969-12-31 23:59:59.5",NULL}'],
        6: ['{"2026-01-02 03:04:05+02","2026-01-02 03:04:05.123456+00"}', "{}"],
        7: ["{123e4567-e89b-12d3-a456-426614174000,NULL}", "{123E4567E89B12D3A456426614174000}"],
    }
    rows = []
    for i in range(70):
        rows.append([str(i)] + [lits[c][i % len(lits[c])] for c in range(1, 8)] + ["{1.5,NaN}"])
    rows.append(["700"] + [W.NULL] * 8)
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(ARR2_COLS), buf, offs)
    names = [c[0] for c in ARR2_COLS]
    cols = b.columns(0, parse_arrays=True)
    assert [cols.column(i).arrow_kind for i in range(1, 9)] == [abi.AK_LIST] * 8     # (numeric[]: a list of Display strings)
    assert cols.column(8).child_kind == abi.AK_LARGE_UTF8
    rb = columns_to_record_batch(cols, names=names, on_text="binary")
    assert rb.column(8).to_pylist()[0] == ["1.5", "NaN"]
    kinds = [pa.float64(), pa.float32(), pa.date32(), pa.time64("us"), pa.timestamp("us"), pa.timestamp("us", tz="UTC"), pa.binary(16)]
    for ci in range(1, 8):
        assert rb.schema.field(names[ci]).type == pa.large_list(kinds[ci - 1]), names[ci]
        want = [None if r[ci] is W.NULL else _oracle_elems(ARR2_COLS[ci][1], r[ci]) for r in rows]
        got = rb.column(ci).to_pylist()
        assert len(got) == len(want)
        for g, w_ in zip(got, want):
            if w_ is None or g is None:
                assert g is None and w_ is None, names[ci]
                continue
            assert len(g) == len(w_), names[ci]
            for x, y in zip(g, w_):
                if isinstance(y, float) and math.isnan(y):
                    assert math.isnan(x)
                else:
                    assert x == y, (names[ci], x, y)
    cols.close(); b.close(); d.close()


def test_array_elements_outside_the_fast_paths_hand_the_row_back():
    """An element the scalar decoder would hand back DEFERRED (a float text the device rule does not settle) makes its ROW deferred:
    null in the list column, set in `deferred`, the rest of the column intact. Temporal elements of a shape only chrono parses are
    decoded like every other (chrono_fallback, codec.hip.h)."""
    rows = [["1", "{1.5}", "{2026-01-02}"], ["2", "{50537618.817359292015891086651596749e82,2}", "{2023-1-01, 2023-12-5}"], ["3", "{2.5}", "{1999-12-31}"]]
    cols3 = [("id", SC.INT8, False, 1), ("f8", 1022, True, 0), ("d", 1182, True, 0)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    c = b.columns(0, parse_arrays=True)
    rb = columns_to_record_batch(c, names=["id", "f8", "d"])
    assert rb.column(1).to_pylist() == [[1.5], None, [2.5]]
    import datetime as dt
    assert rb.column(2).to_pylist() == [[dt.date(2026, 1, 2)], [dt.date(2023, 1, 1), dt.date(2023, 12, 5)], [dt.date(1999, 12, 31)]]
    deferred = np.unpackbits(c.host_arrays(1)[1], bitorder="little")[:3]
    assert list(deferred) == [0, 1, 0] and c.column(1).deferred_count == 1 and c.column(2).deferred_count == 0
    c.close(); b.close(); d.close()


def test_a_handed_back_row_in_front_of_a_malformed_literal():
    """The first problem of a batch in event order is what the reference reports. A row the device hands back DEFERRED (an element of
    more than 40 characters — here one that is itself malformed) may be that problem, so a malformed literal BEHIND it does not fail
    the call: it is handed back as well, and the host, finishing deferred rows in order, raises the right error. Without a handed-back
    row in front, the malformed literal fails the call as before."""
    from etl_amd.decoder import EtlError
    cols2 = [("id", SC.INT8, False, 1), ("a", 1016, True, 0)]
    long_bad = "{92x337203685477-5807N-9223372036854775808}"      # 42 characters in one element, not an integer: Err at decode time
    rows = [["1", "{1,2}"], ["2", long_bad], ["3", "=NU\"L}"], ["4", "{7}"]]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols2), buf, offs)
    c = b.columns(0, parse_arrays=True)
    rb = columns_to_record_batch(c, names=["id", "a"])
    assert rb.column(1).to_pylist() == [[1, 2], None, None, [7]]
    assert list(np.unpackbits(c.host_arrays(1)[1], bitorder="little")[:4]) == [0, 1, 1, 0] and c.column(1).deferred_count == 2
    c.close(); b.close(); d.close()
    buf, offs = _stream([W.insert(42, r) for r in (rows[0], rows[2], rows[1])])   # the malformed one first: the call fails on it
    hb, b, d = _both(SC.simple_table(cols2), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.columns(0, parse_arrays=True)
    assert ei.value.frame_index == 2 and ei.value.description == "Array input missing braces"
    b.close(); d.close()


def test_hand_back_and_malformed_literal_in_one_row():
    """Inside one row the reference parses the columns in order (convert_tuple_to_row): a cell the device hands back DEFERRED in an EARLIER
    column of a row ranks before a malformed literal in a LATER column of the same row — the call succeeds and both are the host's; the
    other way round (the malformed literal in the earlier column) the call fails with the literal's error."""
    from etl_amd.decoder import EtlError
    cols3 = [("id", SC.INT8, False, 1), ("a", 1016, True, 0), ("b", 1016, True, 0)]
    long_el = "{" + "0" * 41 + "7}"          # one element of 42 characters: handed back (the host accepts it: leading zeros)
    bad = "=NU\"L}"                          # "Array input missing braces"
    buf, offs = _stream([W.insert(42, ["1", "{1}", "{2}"]), W.insert(42, ["2", long_el, bad]), W.insert(42, ["3", "{5}", "{6}"])])
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    c = b.columns(0, parse_arrays=True)
    rb = columns_to_record_batch(c, names=["id", "a", "b"])
    assert rb.column(1).to_pylist() == [[1], None, [5]] and rb.column(2).to_pylist() == [[2], None, [6]]
    assert c.column(1).deferred_count == 1 and c.column(2).deferred_count == 1
    c.close(); b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", "{1}", "{2}"]), W.insert(42, ["2", bad, long_el])])
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.columns(0, parse_arrays=True)
    assert ei.value.frame_index == 2 and ei.value.description == "Array input missing braces"
    b.close(); d.close()


def test_text_arrays_on_the_device():
    """text[] (and every array type without a dedicated element arm: ArrayCell::String) as LargeList<LargeUtf8>: quotes, escapes,
    NULL vs "NULL", braces inside quotes, empty strings, multi-byte text — the known answers of the reference's own tests
    (crates/etl/src/postgres/codec/text.rs:324-415, 824-988) plus the oracle on generated literals."""
    kats = [('{a,"null"}', ["a", "null"]), ("{a,NULL}", ["a", None]), ("{a,nUlL}", ["a", None]), ('{"a b"}', ["a b"]),
            ('{"{","}"}', ["{", "}"]), ('{"{a,b}"}', ["{a,b}"]), ('{"with\\"quotes"}', ['with"quotes']),
            ('{"back\\\\slash"}', ["back\\slash"]), ("{}", []), ('{""}', [""]), ('{"",x,""}', ["", "x", ""]),
            ("{hello,world with spaces}", ["hello", "world with spaces"]), ("{é,中文,😀}", ["é", "中文", "😀"]),
            ("{abcd,abcde,nulls,null}", ["abcd", "abcde", "nulls", None]), ("[0:1]={x,y}", ["x", "y"]),
            ("{" + ",".join("e%d" % k for k in range(300)) + "}", ["e%d" % k for k in range(300)]),
            ("{" + "z" * 5000 + "}", ["z" * 5000])]
    cols2 = [("id", SC.INT8, False, 1), ("t", 1009, True, 0), ("v", 1015, True, 0), ("inet", 1041, True, 0)]
    rows = [[str(i), k[0], k[0], k[0]] for i, k in enumerate(kats)] + [["999", W.NULL, W.NULL, W.NULL]]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols2), buf, offs)
    from oracle import oracle
    for lit, want in kats[:15]:   # the oracle agrees with the known answers (its repr is only parsed where that is unambiguous)
        r = oracle.parse_text_cell(1009, lit)
        assert r.startswith("Array[") and r.count("String(") + r.count("NULL") >= len(want), (lit, r)
    c = b.columns(0, parse_arrays=True)
    assert [c.column(i).arrow_kind for i in (1, 2, 3)] == [abi.AK_LIST] * 3 and c.column(1).child_kind == abi.AK_LARGE_UTF8
    rb = columns_to_record_batch(c, names=["id", "t", "v", "inet"])
    for ci in (1, 2, 3):
        assert rb.schema.field(ci).type == pa.large_list(pa.large_utf8())
        assert rb.column(ci).to_pylist() == [k[1] for k in kats] + [None], ci
    c.close(); b.close(); d.close()


def _oracle_display_list(oid, text):
    """The oracle's parse of a numeric[] / timetz[] / bytea[] literal (its repr), turned into what the sinks hand to Arrow: the
    Display strings (oracle/display.py) / the bytes."""
    import re
    from oracle import display as D
    from oracle import oracle
    r = oracle.parse_text_cell(oid, text)
    assert r.startswith("Array["), r
    out = []
    for m in re.finditer(r"NULL|Numeric\((NaN|\+Inf|-Inf|Infinity|-Infinity)\)|Numeric\(([+-]),w=(-?\d+),s=(\d+),\[([\d,]*)\]\)|TimeTz\((\d+):(\d+):(\d+)\.(\d+),(-?\d+)\)|Bytes\(([0-9a-f]*)\)", r[6:-1]):
        t = m.group(0)
        if t == "NULL":
            out.append(None)
        elif t.startswith("Numeric(") and m.group(1):
            out.append({"NaN": "NaN", "+Inf": "Infinity", "Infinity": "Infinity", "-Inf": "-Infinity", "-Infinity": "-Infinity"}[m.group(1)])
        elif t.startswith("Numeric("):
            digits = tuple(int(x) for x in m.group(5).split(",")) if m.group(5) else ()
            out.append(D.numeric_string(0, 1 if m.group(2) == "-" else 0, int(m.group(3)), int(m.group(4)), digits))
        elif t.startswith("TimeTz("):
            out.append(D.timetz_string(int(m.group(6)) * 3600 + int(m.group(7)) * 60 + int(m.group(8)), int(m.group(9)), int(m.group(10))))
        else:
            out.append(bytes.fromhex(m.group(11)))
    return out


def test_numeric_timetz_and_bytea_arrays_on_the_device():
    """numeric[] / timetz[] as LargeList<LargeUtf8> of the elements' Display strings (ArrayCell::Numeric / TimeTz,
    crates/etl-destinations/src/iceberg/encoding.rs:902-945; the reference's own expectation "12345", "-6789" at :1947-1983),
    bytea[] as LargeList<LargeBinary> of the decoded bytes (:2001-2050) — against the oracle's parse + oracle/display.py."""
    nums = ["{12345,-6789,NULL}", "{}", "{1.50,-0.0012000,NaN,1e3,0.000}", '{"Infinity",-Infinity}', "{123456789012345678901234567890.5}", "[2:3]={0,-0}"]
    tzs = ["{12:30:00+02,NULL}", "{}", '{"12:30:00.123456-07:30","00:00:00+15:59:59"}', "{1:2:3+02}"]
    bys = [r'{"\\x0102ff",NULL,"\\x"}', "{}", r'{"\\x' + "ab" * 300 + '"}', r'{"\\xDEADbeef"}']   # Postgres doubles the backslash inside the quotes
    rows = [[str(i), nums[i % len(nums)], tzs[i % len(tzs)], bys[i % len(bys)]] for i in range(80)] + [["900", W.NULL, W.NULL, W.NULL]]
    cols4 = [("id", SC.INT8, False, 1), ("an", 1231, True, 0), ("atz", 1270, True, 0), ("aby", 1001, True, 0)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols4), buf, offs)
    c = b.columns(0, parse_arrays=True)
    assert [c.column(i).arrow_kind for i in (1, 2, 3)] == [abi.AK_LIST] * 3
    rb = columns_to_record_batch(c, names=["id", "an", "atz", "aby"])
    assert rb.schema.field("an").type == pa.large_list(pa.large_utf8()) and rb.schema.field("aby").type == pa.large_list(pa.large_binary())
    for ci, oid in ((1, 1231), (2, 1270), (3, 1001)):
        want = [None if r[ci] is W.NULL else _oracle_display_list(oid, r[ci]) for r in rows]
        assert rb.column(ci).to_pylist() == want, ci
    assert rb.column(1).to_pylist()[0] == ["12345", "-6789", None]        # the reference's own known answer
    c.close(); b.close(); d.close()
    # a malformed element fails like the reference, at the first such row
    from etl_amd.decoder import EtlError
    for col, lit, code in ((1, "{1,abc}", abi.E_NUMERIC), (3, r'{"\\x0g"}', abi.E_BYTEA), (3, "{0102}", abi.E_BYTEA), (3, r'{"\\x012"}', abi.E_BYTEA), (2, "{12:30:00}", abi.E_DATETIME)):
        good = ["1", "{1}", "{12:30:00+00}", r'{"\\x00"}']
        bad = list(good); bad[0] = "2"; bad[col] = lit
        buf, offs = _stream([W.insert(42, good), W.insert(42, bad)])
        hb, b, d = _both(SC.simple_table(cols4), buf, offs)
        from oracle import oracle
        assert oracle.parse_text_cell(cols4[col][1], lit) == f"Err({code})", (lit, oracle.parse_text_cell(cols4[col][1], lit))
        with pytest.raises(EtlError) as ei:
            b.columns(0, parse_arrays=True)
        assert ei.value.code == code and ei.value.frame_index == 2, lit
        b.close(); d.close()


def test_json_arrays_as_lists_of_display_strings():
    """json[] / jsonb[] as LargeList<LargeUtf8> of `j.to_string()` per element (ArrayCell::Json, crates/etl-destinations/src/iceberg/
    encoding.rs:577-585, 964-971; the element strings are serde_json's Display: oracle/json_display.py through oracle/arrays.py). NULL
    elements, escapes inside the quoted elements, key order, number forms; an element of more than 256 bytes or beyond json_display's
    limits hands its row back; an element that is not JSON is the reference's decode error (codec/text.rs:126-134) at its row."""
    from oracle import arrays as OA

    def q(js):   # a JSON text as a quoted array element (Postgres escapes the quotes and the backslashes)
        return '"' + js.replace("\\", "\\\\").replace('"', '\\"') + '"'
    good = ['{"b":1,"a":[true,null,1.50,"x\\ny"]}', "[1,2,{\"k\":\"v\"}]", '"text"', "1e5", "-0.0", "null", '{"z":{"y":{"x":[]}}}', '{"dup":1,"dup":2}', "  [ 1 , 2 ]  "]
    lits = ["{" + ",".join(q(j) for j in good[:3]) + "}", "{" + q(good[3]) + ",NULL," + q(good[4]) + "}", "{}", "{NULL}", "{" + q(good[5]) + "," + q(good[6]) + "}",
            "[2:3]={" + q(good[7]) + "," + q(good[8]) + "}", "{" + q('{"big":"' + "x" * 300 + '"}') + "}", "{" + q("[" * 20 + "]" * 20) + "}"]
    rows = [[str(i), lits[i % len(lits)], lits[(i + 3) % len(lits)]] for i in range(40)] + [["900", W.NULL, W.NULL]]
    cols3 = [("id", SC.INT8, False, 1), ("j", 199, True, 0), ("jb", 3807, True, 0)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    c = b.columns(0, parse_arrays=True)
    assert [c.column(i).arrow_kind for i in (1, 2)] == [abi.AK_LIST] * 2
    rb = columns_to_record_batch(c, names=["id", "j", "jb"])
    assert rb.schema.field("j").type == pa.large_list(pa.large_utf8())
    n_back = 0
    for ci, oid in ((1, 199), (2, 3807)):
        got = rb.column(ci).to_pylist()
        deferred = np.unpackbits(c.host_arrays(ci)[1], bitorder="little")[:len(rows)]
        for r, g, df in zip(rows, got, deferred):
            if r[ci] is W.NULL:
                assert g is None and not df
                continue
            try:
                want = [None if e is None else e.decode() for e, _ in OA.elements(oid, r[ci].encode())]
            except OA.NeedsHost:
                assert g is None and df, r[ci][:60]
                n_back += 1
                continue
            assert g == want and not df, (r[ci][:80], g, want)
    assert n_back >= 8
    assert rb.column(1).to_pylist()[0] == ['{"a":[true,null,1.50,"x\\ny"],"b":1}', '[1,2,{"k":"v"}]', '"text"']   # (sorted keys; the number keeps its literal: arbitrary_precision)
    c.close(); b.close(); d.close()
    from etl_amd.decoder import EtlError
    for lit in ['{"{\\"a\\":}"}', "{abc}", '{"1 2"}']:
        buf, offs = _stream([W.insert(42, ["1", "{1}", "{2}"]), W.insert(42, ["2", lit, "{3}"])])
        hb, b, d = _both(SC.simple_table(cols3), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.columns(0, parse_arrays=True)
        assert ei.value.code == abi.E_JSON and ei.value.frame_index == 2, lit
        b.close(); d.close()


JSON_GOOD = ['{"key": "value", "number": 42}', '{"value":1e309}', "null", " true ", "false", "0", "-0", "-0.5e+10", "1E-400", "123456789012345678901234567890",
             '""', '"a\\"b\\\\c\\/d\\b\\f\\n\\r\\t"', '"\\u00e9\\uD83D\\uDE00"', "[]", "{}", "[1,[2,[3,{}]],{\"a\":[]}]", '{"a":{"b":{"c":[null,true,false]}}}',
             "\t[ 1 , 2 ]\n", '{"k":"v","k":"dup"}', '"é中😀"', "[" * 127 + "]" * 127, '{"a":' * 126 + "1" + "}" * 126]
JSON_BAD = ["invalid json", "", " ", "{", "}", "[1,]", "[,1]", '{"a":}', '{"a" 1}', "{a:1}", "{'a':1}", '{"a":1,}', "01", "1.", ".5", "1e", "1e+", "+1", "- 1", "0x10", "NaN", "Infinity",
            "tru", "nul", "True", '"abc', '"a\\qb"', '"\\u12"', '"\\u12G4"', '"\\uD83D"', '"\\uD83Dx"', '"\\uDE00"', '"\\uD83D\\u0041"', '"a\tb"', '"line\nbreak"', "1 2", "[1] x",
            "[" * 128 + "]" * 128, '{"a":' * 128 + "1" + "}" * 128, "[1 2]", '{"a":1 "b":2}', '["a" "b"]']


def test_json_cells_are_validated_on_the_device():
    """json / jsonb columns leave etlg_batch_columns as their source text (the sink normalises it: serde_json's Display), but only
    after the device has checked that each cell is one JSON value under serde_json's rules (codec/text.rs:126-134; KATs :794-822):
    a malformed cell fails the call with "JSON deserialization failed" at its event, exactly where the oracle's parse fails."""
    from etl_amd.decoder import EtlError
    from oracle import oracle
    cols2 = [("id", SC.INT8, False, 1), ("j", 114, True, 0), ("jb", 3802, True, 0)]
    for t in JSON_GOOD:
        assert oracle.parse_text_cell(114, t).startswith("Json("), t
    for t in JSON_BAD:
        assert oracle.parse_text_cell(3802, t) == f"Err({abi.E_JSON})", t
    rows = [[str(i), t, JSON_GOOD[-1 - i]] for i, t in enumerate(JSON_GOOD)] + [["999", W.NULL, W.NULL]]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols2), buf, offs)
    c = b.columns(0)
    rb = columns_to_record_batch(c, names=["id", "j", "jb"], on_text="binary")
    assert rb.column("j").to_pylist() == [t.encode() for t in JSON_GOOD] + [None]
    c.close(); b.close(); d.close()
    for k, t in enumerate(JSON_BAD):
        good = ["1", "{}", "[]"]
        bad = ["2", "{}", "[]"]
        bad[1 + k % 2] = t
        buf, offs = _stream([W.insert(42, good), W.insert(42, bad), W.insert(42, good)])
        hb, b, d = _both(SC.simple_table(cols2), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.columns(0)
        assert ei.value.code == abi.E_JSON and ei.value.kind == abi.DeserializationError and ei.value.frame_index == 2, t
        assert ei.value.description == "JSON deserialization failed"
        b.close(); d.close()
