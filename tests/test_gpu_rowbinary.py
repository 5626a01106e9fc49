"""Device-side ClickHouse RowBinary (etlg_batch_rowbinary, etl_amd/csrc/columns.hip) byte for byte against oracle/rowbinary.py
(the restatement of crates/etl-destinations/src/clickhouse/encoding.rs + core.rs:96-114, pinned to the reference's own
vectors by tests/test_oracle_rowbinary.py): every class the device encodes, both engines, nullable and non-nullable
destinations, the reference's two ConversionErrors, and the rows / cells that stay with the host."""
import os

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC

pytestmark = pytest.mark.gpu

RB_COLS = [c for c in SC.ALLTYPES if c[0] not in ("j", "arr")]   # the classes the device encodes (numeric / timetz: Display strings formatted on the device)


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


def _stream(msgs):
    s = SC.txn(msgs)
    return np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets


def _check(hb, b, flags, engine, identity_type="PrimaryKey", schema_cols=None):
    """schema_cols: the table's (name, oid, nullable, pk) columns — with them the oracle builds the tombstone rows of key-only Deletes."""
    from oracle import rowbinary as RB
    slot = hb.slots[0]
    rows, idx, host = RB.encode_events(hb.materialize(), 0, [c.type_class for c in slot.cols], list(flags), engine, identity_type, schema_cols)
    r = b.rowbinary(0, flags, engine)
    assert r.status == abi.RB_OK
    assert r.n_rows == len(rows) and int(r.view.n_host_rows) == host
    assert np.array_equal(r.row_event(), np.array(idx, dtype=np.uint64))
    offs = r.row_offsets()
    assert offs[0] == 0 and np.array_equal(np.diff(offs), np.array([len(x) for x in rows], dtype=np.int64))
    assert r.bytes().tobytes() == b"".join(rows)
    r.close()
    return len(rows)


# numeric / timetz texts whose Display exercises every branch of format_numeric_value (crates/etl-postgres/src/numeric.rs:478-560)
# and write_utc_offset (etl-postgres/src/time.rs:210-225)
NUMERICS = ["0", "0.000", "-0.00", "0e-6", "1", "-1", "9999", "10000", "10000.0001", "9999.9999", "0.0012000", "-0.5", "1e-2", "1.23e-2",
            "123e-2", "1e10", "1.5e10", "120.00", "1200000", "-120.00", "0.00000000000000000000000000000000000001", "1e-40",
            "123456789012345678901234567890.123456789012345678901234567890", "NaN", "Infinity", "-Infinity", "inf", "1e100", "-7e-100",
            "0.1", "0.12", "0.123", "0.1234", "0.12345", "12345678.9", "1_000.5", "  42.50  ", "+17"]
TIMETZS = ["12:30:00.123+02", "12:30:00-07:30", "12:30:00+07:30:15", "00:00:00+15:59:59", "23:59:59.999999-15:59:59", "01:02:03+00",
           "01:02:03.5-00:30", "01:02:03.000001+0530", "01:02:03.123456789+05", "12:00:00-023015"]


def _row(**kw):
    full = dict(zip([c[0] for c in SC.ALLTYPES], SC.alltypes_row(**kw)))
    return [full[c[0]] for c in RB_COLS]


@pytest.mark.parametrize("engine", [abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE])
def test_every_encodable_class(engine):
    names = [c[0] for c in RB_COLS]
    rows = [_row(), _row(id="2", b="f", i2="-7", i4="-2147483648", o="4294967295", d="1900-01-01", t="00:00:00",
                         ts="1969-12-31 23:59:59.5", tstz="2026-01-02 03:04:05+02", f8="1e300", f4="-0.5", s="", by="\\x"),
            _row(id="3", d="2299-12-31", t="23:59:59.12", s="x" * 300, by="\\x" + "ab" * 200),
            [("4" if n == "id" else W.NULL) for n in names]]
    rows += [_row(id=str(10 + i), s="y" * (i * 13 % 200), t=f"01:02:{i % 60:02}.{i:06}", n=NUMERICS[i % len(NUMERICS)], tz=TIMETZS[i % len(TIMETZS)])
             for i in range(130)]
    msgs = [W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
    n = len(names)
    nullable = [0 if nm == "id" else 1 for nm in names]
    assert _check(hb, b, nullable + [0, 0], engine) >= len(rows)
    assert _check(hb, b, nullable + [1, 1], engine) >= len(rows)     # Nullable() CDC columns take their marker byte
    # NULL in a non-nullable destination column: the reference's ConversionError, at the first such row
    from etl_amd.decoder import EtlError
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, [0] * n + [0, 0], engine)
    assert ei.value.kind == abi.ConversionError
    assert ei.value.description == "NULL value for non-nullable ClickHouse column" and ei.value.frame_index == 1 + 3
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, nullable + [0], engine)
    assert ei.value.description == "ClickHouse RowBinary row width mismatch"
    b.close(); d.close()


def test_date_out_of_range_fails_like_the_reference():
    from etl_amd.decoder import EtlError
    cols = [("id", SC.INT8, False, 1), ("d", 1082, True, 0)]
    for bad in ("1899-12-31", "2300-01-01"):
        buf, offs = _stream([W.insert(42, ["1", "2000-01-01"]), W.insert(42, ["2", bad])])
        hb, b, d = _both(SC.simple_table(cols), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.rowbinary(0, [0, 1, 0, 0])
        assert ei.value.description == "Date out of ClickHouse Date32 range" and ei.value.frame_index == 2
        b.close(); d.close()


def test_cells_and_rows_that_stay_with_the_host():
    # every class of the all-types table is written on the device — the json cell as serde_json's Display (tests/test_gpu_json_display.py) ...
    buf, offs = _stream([W.insert(42, SC.alltypes_row())])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    assert _check(hb, b, [1] * len(SC.ALLTYPES) + [0, 0], abi.CH_MERGE_TREE) == 1
    b.close(); d.close()
    # ... but a json cell nested deeper than a lane follows stays with the host, reported with its event and column
    buf, offs = _stream([W.insert(42, SC.alltypes_row()), W.insert(42, SC.alltypes_row(id="2", j="[" * 17 + "]" * 17))])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    r = b.rowbinary(0, [1] * len(SC.ALLTYPES) + [0, 0])
    assert r.status == abi.RB_NEEDS_HOST and r.n_rows == 0 and (int(r.view.host_event), r.view.host_column) == (2, [c[0] for c in SC.ALLTYPES].index("j"))
    r.close(); b.close(); d.close()
    # a DEFERRED float: reported with its event and column
    cols = [("id", SC.INT8, False, 1), ("x", SC.FLOAT8, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "1.5"]), W.insert(42, ["2", "50537618.817359292015891086651596749e82"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.rowbinary(0, [0, 1, 0, 0])
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close(); b.close(); d.close()


VAR_ARRAY_LITS = {
    1009: ['{a,"b c",NULL,"null",nUlL}', '{"x\\"y","a,b","{}"}', '{é,"\\\\"}', "{ a , b }", '{"",x}', "{abcd,abcde,nulls,null}", "[0:1]={x,y}", "{}", "{\\n\\ull,n\\ul}",
           '{"' + "z" * 300 + '",q}'],
    1015: ["{v1,v2}", "{}"], 1014: ['{"ab  ","c"}'], 1002: ["{a,b}"], 1003: ["{pg_catalog,public}"],
    1231: ["{12345,-6789,NULL}", "{}", "{1.50,-0.0012000,NaN,1e3,0.000}", '{"Infinity",-Infinity}', "{123456789012345678901234567890.5}", "[2:3]={0,-0}"],
    1270: ["{12:30:00+02,NULL}", "{}", '{"12:30:00.123456-07:30","00:00:00+15:59:59"}', "{1:2:3+02}"],
    1001: ['{"\\\\x0102ff",NULL,"\\\\x"}', "{}", '{"\\\\x' + "ab" * 300 + '"}', '{"\\\\xDEADbeef"}'],
}


def test_arrays_of_var_len_elements():
    """text-like / numeric / timetz / bytea arrays as Array(Nullable(String)) (array_cell_to_clickhouse_values, encoding.rs:89-111: the
    unescaped text, `n.to_string()`, `t.to_string()`, bytes_to_hex of the decoded bytes): quoting, escapes, the NULL rule (unquoted,
    unescaped-to "null" of any case), elements longer than anything fixed, the dimensions prefix — against oracle/arrays.py; malformed
    literals stay with the host (json[]: tests/test_gpu_json_display.py)."""
    names = sorted(VAR_ARRAY_LITS)
    cols = [("id", SC.INT8, False, 1)] + [(f"a{o}", o, True, 0) for o in names]
    n = max(len(v) for v in VAR_ARRAY_LITS.values())
    rows = [[str(k)] + [VAR_ARRAY_LITS[o][k % len(VAR_ARRAY_LITS[o])] for o in names] for k in range(n)] + [[str(n)] + [W.NULL] * len(names)]
    buf, offs = _stream([W.insert(42, r) for r in rows] + [W.update(42, rows[1])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    for engine in (abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE):
        assert _check(hb, b, [0] + [1] * len(names) + [0, 0], engine) == len(rows) + 1
    b.close(); d.close()
    for oid, lit in ((1009, '{a,"b}'), (1009, "{a,{b}}"), (1001, '{"\\\\x0g"}'), (1001, "{abc}"), (1231, "{1.5,x}"), (1231, "{" + "1" * 41 + "}")):
        buf, offs = _stream([W.insert(42, ["1", "{}"]), W.insert(42, ["2", lit])])
        hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("a", oid, True, 0)]), buf, offs)
        r = b.rowbinary(0, [0, 1, 0, 0])
        assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1), (oid, lit)
        r.close(); b.close(); d.close()


@pytest.mark.parametrize("parts", [1, 2, 4])
def test_rows_split_among_lanes(parts):
    """k_rb_rows writes a row with 1-4 lanes (the host picks by the row count: four for the small batches of this file): the other
    splits, forced, on the 17-column all-classes table and on cfg3 — RowBinary and protobuf rows as the oracle has them."""
    from tests.test_gpu_protobuf import _check as _check_pb
    os.environ["ETLG_RB_PARTS"] = str(parts)
    try:
        rows = [_row(id=str(i), s="y" * (i * 13 % 200), n=NUMERICS[i % len(NUMERICS)], tz=TIMETZS[i % len(TIMETZS)]) for i in range(70)]
        rows[5] = [("5" if c[0] == "id" else W.NULL) for c in RB_COLS]
        buf, offs = _stream([W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])])
        hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
        nullable = [0 if c[0] == "id" else 1 for c in RB_COLS]
        for engine in (abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE):
            assert _check(hb, b, nullable + [1, 0], engine) >= len(rows)
        b.close(); d.close()
        # rows too long for the workgroup's LDS image (64 rows of ~1.5 KB): the lanes write to global memory themselves; a mix of both in one call
        big = [_row(id=str(i), s="z" * (1500 + i if i < 80 else i % 7)) for i in range(200)]
        buf, offs = _stream([W.insert(42, r) for r in big])
        hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
        assert _check(hb, b, nullable + [0, 0], abi.CH_MERGE_TREE) == len(big)
        assert _check_pb(hb, b, RB_COLS) == len(big)
        b.close(); d.close()
        w = synth.cfg3()
        buf, offs = w.fill(96 << 10)
        hb, b, d = _both(w.register, buf, offs)
        flags = [1 if c.nullable else 0 for c in hb.slots[0].cols]
        assert _check(hb, b, flags + [0, 0], abi.CH_MERGE_TREE, schema_cols=w.schema_cols(w.tables[0])) > 50
        assert _check_pb(hb, b, w.schema_cols(w.tables[0])) > 50
        b.close(); d.close()
    finally:
        os.environ.pop("ETLG_RB_PARTS", None)


@pytest.mark.parametrize("mk,nbytes", [(synth.cfg2, 1 << 20), (synth.cfg3, 1 << 20)])   # cfg3: BASELINE's var-len schema (TEXT, NUMERIC, timestamptz, uuid)
@pytest.mark.parametrize("engine", [abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE])
def test_synthetic_stream(mk, nbytes, engine):
    if os.environ.get("ETLG_SIMT_RUN") == "1":
        nbytes = 128 << 10
    w = mk()
    buf, offs = w.fill(nbytes)
    hb, b, d = _both(w.register, buf, offs)
    flags = [1 if c.nullable else 0 for c in hb.slots[0].cols]   # Nullable() where the source column is
    assert _check(hb, b, flags + [0, 0], engine, schema_cols=w.schema_cols(w.tables[0])) > 100   # (cfg3 carries key-only deletes: tombstone rows)
    b.close(); d.close()


def test_updates_deletes_and_host_rows():
    """cfg3-like traffic on a table the device encodes: updates (new row), deletes with a full old row, key-only deletes as
    tombstone rows (expand_key_row); partial updates are counted, not encoded."""
    cols = [("id", SC.INT8, False, 1), ("v", SC.INT4, True, 0), ("s", 25, True, 0)]
    msgs = []
    for i in range(300):
        r = [str(i), str(i * 3), "t" * (i % 50)]
        msgs.append(W.insert(42, r))
        if i % 3 == 0:
            msgs.append(W.update(42, [str(i), W.NULL, "u"]))
        if i % 5 == 0:
            msgs.append(W.update(42, [str(i), "1", W.TOAST]))                   # partial: host
        if i % 7 == 0:
            msgs.append(W.delete(42, key=[str(i), W.NULL, W.NULL]))              # key only: the tombstone row
        if i % 11 == 0:
            msgs.append(W.delete(42, old=r))                                     # full old row
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    for engine in (abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE):
        n = _check(hb, b, [0, 1, 1, 0, 0], engine, schema_cols=cols)
        assert n == 300 + 100 + 43 + 28 and b.rowbinary(0, [0, 1, 1, 0, 0], engine).view.n_host_rows == 60
    b.close(); d.close()


def test_key_only_deletes_become_tombstone_rows():
    """expand_key_row + default_cell (clickhouse/core.rs:1437-1517) on every class: a two-column primary key in the middle of the
    table, nullable and non-nullable columns of every class beside it (NULL where the source column is nullable and not an array,
    the zero value otherwise: typed zeros for date / timestamp / uuid, empty arrays, empty Strings for numeric / time / timetz / bytea /
    text), both engines, Nullable() and plain destination columns. Then the slots whose key-only Deletes stay with the host: another
    replica identity, and a nullable column of a type the value codec has no arm for."""
    from etl_amd.schema import infer_identity_type
    base = [c for c in SC.ALLTYPES if c[0] not in ("j", "arr")]
    cols = []
    for k, c in enumerate(base):                       # every class twice: NOT NULL and nullable
        cols.append((c[0] + "_nn", c[1], False, 0))
        cols.append((c[0] + "_n", c[1], True, 0))
    cols.insert(5, ("k1", SC.INT8, False, 1))
    cols.insert(9, ("k2", 25, False, 1))
    cols.append(("ia_nn", 1007, False, 0))             # int4[]: arrays are never NULL in the tombstone
    cols.append(("ia_n", 1007, True, 0))
    ident = [1 if c[3] else 0 for c in cols]
    assert infer_identity_type(cols, [1] * len(cols), ident) == "PrimaryKey"
    msgs = [W.delete(42, key=[("%d" % i) if c[3] and c[1] == SC.INT8 else ("key-%d" % i) if c[3] else W.NULL for c in cols]) for i in range(70)]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols, ident=ident), buf, offs)
    for engine in (abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE):
        for dest_nullable in (0, 1):
            flags = [dest_nullable if c[2] else 0 for c in cols] + [0, dest_nullable]
            if not dest_nullable:
                # the tombstone's NULLs meet non-nullable destination columns: the reference's encoder error, on the first row
                from etl_amd.decoder import EtlError
                with pytest.raises(EtlError) as ei:
                    b.rowbinary(0, flags, engine)
                assert ei.value.description == "NULL value for non-nullable ClickHouse column" and ei.value.frame_index == 1
                continue
            assert _check(hb, b, flags, engine, schema_cols=cols) == 70
    b.close(); d.close()
    # the same deletes under REPLICA IDENTITY USING INDEX on a non-PK column: expand_key_row refuses them (ensure_clickhouse_key_identity_
    # is_primary_key) -> host rows; and with a nullable `money` column (a String cell; is_array_type is the host's to answer): host rows
    cols2 = [("id", SC.INT8, False, 1), ("k", SC.INT4, False, 0), ("s", 25, True, 0)]
    msgs2 = [W.insert(42, ["1", "2", "x"]), W.delete(42, key=[W.NULL, "2", W.NULL])]
    buf, offs = _stream(msgs2)
    hb, b, d = _both(SC.simple_table(cols2, ident=[0, 1, 0]), buf, offs)
    assert _check(hb, b, [0, 0, 1, 0, 0], abi.CH_MERGE_TREE, "AlternativeKey", schema_cols=cols2) == 1
    assert b.rowbinary(0, [0, 0, 1, 0, 0], abi.CH_MERGE_TREE).view.n_host_rows == 1
    b.close(); d.close()
    cols3 = [("id", SC.INT8, False, 1), ("m", 790, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "$2.00"]), W.delete(42, key=["1", W.NULL])])
    hb, b, d = _both(SC.simple_table(cols3), buf, offs)
    assert _check(hb, b, [0, 1, 0, 0], abi.CH_MERGE_TREE, schema_cols=cols3) == 1
    assert b.rowbinary(0, [0, 1, 0, 0], abi.CH_MERGE_TREE).view.n_host_rows == 1
    b.close(); d.close()


def test_arrays_of_fixed_width_elements():
    """Array(Nullable(T)) (encoding.rs:249-254): varint count + every element with its null marker, for the element classes the
    device parses; text / numeric arrays and literals the device cannot take apart stay with the host."""
    cols = [("id", SC.INT8, False, 1), ("a4", 1007, True, 0), ("ab", 1000, True, 0), ("af", 1022, True, 0), ("ad", 1182, True, 0),
            ("ats", 1185, True, 0), ("au", 2951, True, 0), ("at", 1183, True, 0)]
    lits = [["{1,NULL,3}", "{t,f,NULL}", "{1.5,-0.25}", "{2026-01-02,1969-12-31}", '{"2026-01-02 03:04:05.123456+00"}', "{123e4567-e89b-12d3-a456-426614174000,NULL}", "{12:30:45.5}"],
            ["{}", "{}", "{}", "{}", "{}", "{}", "{}"],
            ["[1:2]={-7,2147483647}", "{t}", "{1e300,NULL}", "{NULL}", "{NULL}", "{NULL}", "{00:00:00,23:59:59.123456}"]]
    rows = [[str(i)] + lits[i % 3] for i in range(70)] + [["99"] + [W.NULL] * 7]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    for engine in (abi.CH_MERGE_TREE, abi.CH_REPLACING_MERGE_TREE):
        assert _check(hb, b, [0] + [1] * 7 + [0, 0], engine) == len(rows)
    b.close(); d.close()
    # a date out of range inside an array fails like a scalar one; a malformed literal and a text[] column are the host's
    from etl_amd.decoder import EtlError
    buf, offs = _stream([W.insert(42, ["1"] + lits[0][:3] + ["{1899-12-31}"] + lits[0][4:])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, [0] + [1] * 7 + [0, 0])
    assert ei.value.description == "Date out of ClickHouse Date32 range"
    b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", "{1,{2}}"] + lits[0][1:])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.rowbinary(0, [0] + [1] * 7 + [0, 0])
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (1, 1)
    r.close(); b.close(); d.close()


def test_replacing_merge_tree_leaves_updates_of_other_identities_to_the_host():
    """REPLICA IDENTITY USING INDEX on a non-PK column (AlternativeKey): clickhouse_update_row refuses Update events under
    ReplacingMergeTree (clickhouse/core.rs:1359-1427, 'ClickHouse requires primary-key or full replica identity'), so the device
    does not encode them there — they are counted as host rows; MergeTree takes them; REPLICA IDENTITY FULL is accepted by both."""
    from etl_amd.schema import infer_identity_type
    cols = [("id", SC.INT8, False, 1), ("k", SC.INT4, False, 0), ("s", 25, True, 0)]
    msgs = []
    for i in range(100):
        r = [str(i), str(i * 2), "t%d" % i]
        msgs.append(W.insert(42, r))
        if i % 3 == 0:
            msgs.append(W.update(42, [str(i), str(i * 2), "u"]))
    buf, offs = _stream(msgs)
    for ident, want in (([0, 1, 0], "AlternativeKey"), ([1, 1, 1], "Full"), ([1, 0, 0], "PrimaryKey")):
        assert infer_identity_type(cols, [1, 1, 1], ident) == want
        hb, b, d = _both(SC.simple_table(cols, ident=ident), buf, offs)
        n_mt = _check(hb, b, [0, 0, 1, 0, 0], abi.CH_MERGE_TREE, want)
        n_rmt = _check(hb, b, [0, 0, 1, 0, 0], abi.CH_REPLACING_MERGE_TREE, want)
        assert n_mt == 134 and n_rmt == (100 if want == "AlternativeKey" else 134)
        b.close(); d.close()


def test_date_range_error_comes_before_a_null_error():
    """cell_to_clickhouse_value runs over every pending row before any row is encoded (clickhouse/core.rs:1193-1203), so a date
    outside Date32's range in a LATER row is reported instead of a NULL in a non-nullable column of an earlier one; inside one
    row a range error behind a NULL cell wins too."""
    from etl_amd.decoder import EtlError
    from oracle import rowbinary as RB
    cols = [("id", SC.INT8, False, 1), ("v", SC.INT4, True, 0), ("d", 1082, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "5", "2000-01-01"]), W.insert(42, ["2", W.NULL, "2000-01-01"]),
                         W.insert(42, ["3", "7", "2000-01-02"]), W.insert(42, ["4", W.NULL, "1899-12-31"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(RB.ConversionError) as oi:
        RB.encode_events(hb.materialize(), 0, [c.type_class for c in hb.slots[0].cols], [0, 0, 1, 0, 0], abi.CH_MERGE_TREE)
    assert str(oi.value) == "Date out of ClickHouse Date32 range"
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, [0, 0, 1, 0, 0])
    assert ei.value.description == "Date out of ClickHouse Date32 range" and ei.value.frame_index == 4
    with pytest.raises(EtlError) as ei:      # without the bad date the NULL error is the first row's that has one
        b.rowbinary(0, [0, 0, 0, 0, 0])
    assert ei.value.description == "Date out of ClickHouse Date32 range"
    b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", W.NULL, "2000-01-01"]), W.insert(42, ["2", W.NULL, "2000-01-01"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, [0, 0, 1, 0, 0])
    assert ei.value.description == "NULL value for non-nullable ClickHouse column" and ei.value.frame_index == 1
    b.close(); d.close()
