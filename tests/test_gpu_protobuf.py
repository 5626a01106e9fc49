"""Device-side BigQuery protobuf rows (etlg_batch_protobuf, etl_amd/csrc/columns.hip) byte for byte against oracle/protobuf.py
(restatement of crates/etl-destinations/src/bigquery/encoding.rs:120-190 on the protobuf wire format): every class the device
encodes incl. negative integers (10-byte varints), NULLs (absent fields), dates / times / timestamps as chrono strings, the
trailing UPSERT / sequence-key fields; updates and deletes counted for the host; host-only classes reported."""
import os

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC
from tests.test_gpu_rowbinary import NUMERICS, RB_COLS, TIMETZS, _both, _row, _stream

pytestmark = pytest.mark.gpu


def _check(hb, b, schema_cols=None, identity_type="PrimaryKey"):
    """schema_cols: the table's (name, oid, nullable, pk) columns — with them the oracle builds the rows of updates and deletes."""
    from oracle import protobuf as PB
    rows, idx, host = PB.event_rows(hb.materialize(), 0, schema_cols, identity_type)
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.n_rows == len(rows) and int(r.view.n_host_rows) == host
    assert np.array_equal(r.row_event(), np.array(idx, dtype=np.uint64))
    offs = r.row_offsets()
    assert np.array_equal(np.diff(offs), np.array([len(x) for x in rows], dtype=np.int64))
    assert r.bytes().tobytes() == b"".join(rows)
    r.close()
    return len(rows)


def test_every_encodable_class():
    names = [c[0] for c in RB_COLS]
    rows = [_row(), _row(id="2", b="f", i2="-7", i4="-2147483648", o="4294967295", d="0001-01-01", t="00:00:00",
                         ts="1969-12-31 23:59:59.5", tstz="1969-12-31 23:59:59.5+00", f8="1e300", f4="-0.5", s="", by="\\x"),
            _row(id="-9223372036854775808", d="9999-12-31", t="23:59:59.12", ts="2026-01-02 03:04:05", s="x" * 300, by="\\x" + "ab" * 200),
            [("4" if n == "id" else W.NULL) for n in names]]
    ok_numerics = [t for t in NUMERICS if t not in ("1e-40", "-7e-100")]      # more than 38 decimal places: BigQuery's validation refuses them (below)
    rows += [_row(id=str(10 + i), s="y" * (i * 13 % 200), t=f"01:02:{i % 60:02}.{i:06}", n=ok_numerics[i % len(ok_numerics)], tz=TIMETZS[i % len(TIMETZS)])
             for i in range(130)]
    msgs = [W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
    assert _check(hb, b, RB_COLS) == len(rows) + 2          # + the update's UPSERT row and the delete's sparse DELETE row
    b.close(); d.close()


def test_host_only_classes_and_deferred_cells():
    from etl_amd.decoder import EtlError
    # the all-classes table: every class is written on the device (json: tests/test_gpu_json_display.py; int4[]: below) ...
    buf, offs = _stream([W.insert(42, SC.alltypes_row(arr="{1,2,3}")), W.insert(42, SC.alltypes_row(id="2", arr="{}"))])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    assert _check(hb, b, SC.ALLTYPES) == 2
    b.close(); d.close()
    # ... and its default row carries {1,NULL,3}: BigQuery takes no NULL inside an array (reject_nulls, validation.rs:127-141)
    buf, offs = _stream([W.insert(42, SC.alltypes_row())])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.NullValuesNotSupportedInArrayInDestination and ei.value.detail == f"Cell at index {[c[0] for c in SC.ALLTYPES].index('arr')} failed validation"
    b.close(); d.close()
    cols = [("id", SC.INT8, False, 1), ("x", SC.FLOAT8, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "1.5"]), W.insert(42, ["2", "50537618.817359292015891086651596749e82"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close(); b.close(); d.close()


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_synthetic_stream(mk):
    w = mk()
    buf, offs = w.fill((128 << 10) if os.environ.get("ETLG_SIMT_RUN") == "1" else (1 << 20))
    hb, b, d = _both(w.register, buf, offs)
    assert _check(hb, b, w.schema_cols(w.tables[0])) > 100   # (cfg3: updates with key / full old rows, key-only deletes)
    b.close(); d.close()


def test_updates_and_deletes():
    """bigquery/core.rs:978-1036 + 1425-1754 on the device: an update is its new row as UPSERT — behind a sparse DELETE row of the OLD
    primary key (sequence ordinal 0, the UPSERT then 1) when the update changed the key, judged by comparing the old image's
    primary-key cells (key image or full old row) with the new row's; a delete is the sparse DELETE row of its old image's primary
    key; two-column keys (int8 + text), NULL in a key column, keys that differ only in the text column's last byte / its length.
    The events the reference refuses stay with the host: partial updates, deletes without an old row."""
    cols = [("a", SC.INT4, True, 0), ("k1", SC.INT8, False, 1), ("s", 25, True, 0), ("k2", 25, False, 1), ("u", SC.UUID, True, 0)]
    ident = [0, 1, 0, 1, 0]
    U1 = "123e4567-e89b-12d3-a456-426614174000"
    msgs = []
    for i in range(120):
        k1, k2 = str(i), "key-%03d" % i
        row = [str(i * 2), k1, "text %d" % i, k2, U1]
        msgs.append(W.insert(42, row))
        m = i % 12
        if m == 0:
            msgs.append(W.update(42, [W.NULL, k1, "changed", k2, W.NULL]))                              # no old image: key unchanged
        elif m == 1:
            msgs.append(W.update(42, ["1", k1, "x", k2, U1], key=[W.NULL, k1, W.NULL, k2, W.NULL]))      # key image, same key
        elif m == 2:
            msgs.append(W.update(42, ["1", str(i + 1000), "x", k2, U1], key=[W.NULL, k1, W.NULL, k2, W.NULL]))   # k1 changed
        elif m == 3:
            msgs.append(W.update(42, ["1", k1, "x", k2[:-1] + "X", U1], key=[W.NULL, k1, W.NULL, k2, W.NULL]))   # k2: last byte
        elif m == 4:
            msgs.append(W.update(42, ["1", k1, "x", k2 + "+", U1], key=[W.NULL, k1, W.NULL, k2, W.NULL]))        # k2: length
        elif m == 5:
            msgs.append(W.update(42, ["1", k1, W.TOAST, k2, U1]))                                        # partial: host
        elif m == 6:
            msgs.append(W.delete(42, key=[W.NULL, k1, W.NULL, k2, W.NULL]))
        elif m == 7:
            msgs.append(W.delete(42, old=row))
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols, ident=ident), buf, offs)
    n = _check(hb, b, cols)
    assert n == 120 + 10 * 5 + 10 * 3 + 10 * 2 and b.protobuf(0).view.n_host_rows == 10
    b.close(); d.close()
    # REPLICA IDENTITY FULL: old rows are full images; the primary key is still (k1, k2)
    msgs = []
    for i in range(40):
        row = [str(i), str(i), "t", "k%d" % i, U1]
        msgs.append(W.insert(42, row))
        if i % 4 == 0:
            msgs.append(W.update(42, [str(i + 1), str(i), "t2", "k%d" % i, U1], old=row))               # other columns changed only
        elif i % 4 == 1:
            msgs.append(W.update(42, [str(i), str(i), "t", "K%d" % i, U1], old=row))                     # key changed
        elif i % 4 == 2:
            msgs.append(W.delete(42, old=row))
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols, ident=[1] * 5), buf, offs)
    assert _check(hb, b, cols, "Full") == 40 + 10 + 20 + 10
    b.close(); d.close()


def test_events_bigquery_refuses_stay_with_the_host():
    """A key image / an update without an old row under a replica identity that is not the primary key
    (ensure_bigquery_key_image_matches_primary_key, ensure_bigquery_update_without_old_row_can_skip_delete: core.rs:1515-1555), a
    delete without an old row (bigquery_delete_old_row :1497-1511); and updates of a table whose primary key has a float column
    (Cell equality is not bit equality there) when they carry an old row."""
    cols = [("id", SC.INT8, False, 1), ("k", SC.INT4, False, 0), ("s", 25, True, 0)]
    msgs = [W.insert(42, ["1", "2", "x"]), W.update(42, ["1", "2", "y"]), W.update(42, ["1", "3", "y"], key=[W.NULL, "2", W.NULL]),
            W.delete(42, key=[W.NULL, "3", W.NULL])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols, ident=[0, 1, 0]), buf, offs)
    assert _check(hb, b, cols, "AlternativeKey") == 1 and b.protobuf(0).view.n_host_rows == 3
    b.close(); d.close()
    cols = [("f", SC.FLOAT8, False, 1), ("s", 25, True, 0)]
    msgs = [W.insert(42, ["1.5", "x"]), W.update(42, ["1.5", "y"]), W.update(42, ["2.5", "y"], key=["1.5", W.NULL]), W.delete(42, key=["2.5", W.NULL])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    assert _check(hb, b, cols) == 3 and b.protobuf(0).view.n_host_rows == 1
    b.close(); d.close()


def test_numeric_with_more_than_38_decimal_places_fails_like_the_reference():
    """validate_numeric_for_bigquery (bigquery/validation.rs:20-35; its own cases :213-229: scale 38 passes, 39 fails) behind
    BigQueryTableRow::try_from_tagged_cells (encoding.rs:37-45): kind UnsupportedValueInDestination, description 'Cell validation
    failed for BigQuery compatibility', detail naming the cell; the first failing row in event order."""
    from etl_amd.decoder import EtlError
    from oracle import protobuf as PB
    cols = [("id", SC.INT8, False, 1), ("v", SC.NUMERIC, True, 0)]
    ok = "0.00000000000000000000000000000000000001"
    bad = "0.000000000000000000000000000000000000001"
    buf, offs = _stream([W.insert(42, ["1", "123.456"]), W.insert(42, ["2", ok])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    assert _check(hb, b) == 2
    b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", ok]), W.insert(42, ["2", bad]), W.insert(42, ["3", bad]), W.insert(42, ["4", "NaN"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(PB.UnsupportedValueInDestination) as oi:
        PB.insert_rows(hb.materialize(), 0)
    assert str(oi.value) == "Cell at index 1 failed validation"
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.description == "Cell validation failed for BigQuery compatibility"
    assert ei.value.detail == "Cell at index 1 failed validation" and ei.value.frame_index == 2
    b.close(); d.close()


def test_reference_pinned_cell_encodings_on_the_device():
    """The vectors the reference's own tests hold (tests/golden/bigquery_kats.py, encoding.rs:451-496) through etlg_batch_protobuf, against
    LITERAL bytes: a timestamptz cell is the int64 varint of its epoch microseconds; a numeric with 38 decimal places is its text, one with
    39 fails the batch with the reference's kind and detail; a json integer literal outside u64 fails the same way; an int4[] with a
    NULL element fails with the reference's kind and detail (arrays: test_arrays_of_fixed_width_elements)."""
    from etl_amd.decoder import EtlError
    from tests.golden import bigquery_kats as K
    cols = [("ts", SC.TIMESTAMPTZ, False, 1), ("v", SC.NUMERIC, True, 0)]
    buf, offs = _stream([W.insert(42, ["2026-01-02 03:04:05+00", K.NUMERIC_AT_SCALE])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.n_rows == 1
    row = r.bytes().tobytes()
    want = K.TSTZ_SCALAR_BYTES + bytes([0x12, len(K.NUMERIC_AT_SCALE)]) + K.NUMERIC_AT_SCALE.encode()     # field 1: int64 varint; field 2: string
    assert row.startswith(want), (row[:len(want)].hex(), want.hex())
    ev = hb.materialize()
    ins = [e for e in ev if e["kind"] == "I"][0]
    assert row == want + bytes([0x1A, 6]) + b"UPSERT" + bytes([0x22, 50]) + f"{ins['commit_lsn']:016x}/{ins['tx_ordinal']:016x}/{0:016x}".encode()
    r.close(); b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["2026-01-02 03:04:05+00", K.NUMERIC_OVER_SCALE])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.detail == "Cell at index 1 failed validation"
    b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", "{" + ",".join("NULL" if v is None else str(v) for v in K.ARRAY_WITH_NULLS) + "}"])])   # encoding.rs:372-383
    hb, b, d = _both(SC.simple_table([("a", SC.INT4_A, True, 1)]), *_stream([W.insert(42, ["{1,NULL,3}"])]))
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.NullValuesNotSupportedInArrayInDestination and ei.value.detail == "Cell at index 0 failed validation"
    b.close(); d.close()
    for text in K.JSON_REFUSED:                        # encoding.rs:353-360
        buf, offs = _stream([W.insert(42, ["1", text])])
        hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("j", SC.JSONB, True, 0)]), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.protobuf(0)
        assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.detail == "Cell at index 1 failed validation"
        b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", K.JSON_ACCEPTED[0]])])      # encoding.rs:344-351: {"value":1e309} is BigQuery's to judge
    hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("j", SC.JSONB, True, 0)]), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.bytes().tobytes().startswith(bytes([0x08, 1, 0x12, 16]) + b'{"value":1e+309}')   # (Display as codec/text.rs:812-815 has it)
    r.close(); b.close(); d.close()


# array type OIDs of the element classes the device encodes (tests/scenarios.py names only int4[])
ARRAYS = {"bool": 1000, "int2": 1005, "int4": 1007, "int8": 1016, "oid": 1028, "float4": 1021, "float8": 1022,
          "date": 1182, "time": 1183, "timestamp": 1115, "timestamptz": 1185, "uuid": 2951}


def test_arrays_of_fixed_width_elements():
    """array_cell_encode_prost (bigquery/encoding.rs:203-290) on the device: bool / int2 / int4 / oid / int8 / float4 / float8 / timestamptz
    arrays PACKED, date / time / timestamp / uuid arrays one string field per element, empty arrays nothing, quoted elements and a
    dimensions prefix; the timestamptz array of the reference's own test against its LITERAL bytes (encoding.rs:451-480); a NULL element
    fails the batch like reject_nulls (validation.rs:127-141; encoding.rs:372-383); text-like / numeric / timetz / bytea elements as one
    string (bytes) field each, the numeric scale rule inside arrays (encoding.rs:385-404)."""
    from etl_amd.decoder import EtlError
    from tests.golden import bigquery_kats as K
    lits = {"bool": ["{t,f,t}", "{}", "{f}"], "int2": ["{1,-2,32767}", "{-32768}", "{}"], "int4": ["{1,2,3}", "{-1,2147483647,-2147483648}", "[0:2]={7,8,9}"],
            "int8": ["{9223372036854775807,-9223372036854775808,0}", "{}", "{1}"], "oid": ["{0,4294967295}", "{42}", "{}"],
            "float4": ["{1.5,-0.25,3e10}", "{}", "{0}"], "float8": ["{1.5,-2.25e-300,1e300}", "{0.1}", "{}"],
            "date": ["{2026-01-02,0001-01-01}", "{}", '{"9999-12-31"}'], "time": ["{12:30:45.123456,00:00:00}", "{23:59:59.5}", "{}"],
            "timestamp": ['{"2026-01-02 03:04:05.123456","1969-12-31 23:59:59.5"}', "{}", '{"2000-01-01 00:00:00"}'],
            "timestamptz": ['{"2026-01-02 03:04:05+00"}', '{"1969-12-31 23:59:59.5+00","2026-01-02 05:04:05.000001+02"}', "{}"],
            "uuid": ["{123e4567-e89b-12d3-a456-426614174000,00000000-0000-0000-0000-000000000000}", "{}", "{FFFFFFFF-FFFF-FFFF-FFFF-FFFFFFFFFFFF}"]}
    names = sorted(ARRAYS)
    cols = [("id", SC.INT8, False, 1)] + [(n, ARRAYS[n], True, 0) for n in names]
    rows = [[str(k)] + [lits[n][k] for n in names] for k in range(3)] + [["3"] + [W.NULL] * len(names)]
    buf, offs = _stream([W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    assert _check(hb, b, cols) == len(rows) + 2
    r = b.protobuf(0)     # row 0: id = 0 (field 1 varint), then the arrays in column order; the timestamptz array is field 1 + its position
    tag = 2 + names.index("timestamptz")
    assert bytes([tag << 3 | 2, len(K.TSTZ_VARINT)]) + K.TSTZ_VARINT in r.bytes().tobytes()[:int(r.row_offsets()[1])]
    r.close(); b.close(); d.close()
    assert K.TSTZ_PACKED_BYTES == bytes([1 << 3 | 2, len(K.TSTZ_VARINT)]) + K.TSTZ_VARINT       # (the same field under the test's tag 1)
    for n in names:                                                                              # a NULL element, in every class
        lit = "{" + ",".join(['"' + e + '"' for e in lits[n][0][1:-1].split(",")][:1] + ["NULL"]) + "}"
        buf, offs = _stream([W.insert(42, ["1", lits[n][0]]), W.insert(42, ["2", lit])])
        hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("a", ARRAYS[n], True, 0)]), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.protobuf(0)
        assert ei.value.kind == abi.NullValuesNotSupportedInArrayInDestination and ei.value.description == "Cell validation failed for BigQuery compatibility", n
        assert ei.value.detail == "Cell at index 1 failed validation" and ei.value.frame_index == 2
        b.close(); d.close()
    # text-like / numeric / timetz / bytea elements: one string (bytes) field per element, against oracle/arrays.py
    from tests.test_gpu_rowbinary import VAR_ARRAY_LITS
    onames = sorted(VAR_ARRAY_LITS)
    vcols = [("id", SC.INT8, False, 1)] + [(f"a{o}", o, True, 0) for o in onames]
    plain = {o: [t for t in VAR_ARRAY_LITS[o] if "null" not in t.lower().replace("\\", "")] for o in onames}     # (no NULL elements)
    nr = max(len(v) for v in plain.values())
    vrows = [[str(k)] + [plain[o][k % len(plain[o])] for o in onames] for k in range(nr)] + [[str(nr)] + [W.NULL] * len(onames)]
    buf, offs = _stream([W.insert(42, r) for r in vrows])
    hb, b, d = _both(SC.simple_table(vcols), buf, offs)
    assert _check(hb, b, vcols) == len(vrows)
    b.close(); d.close()
    for oid, lit, kind in ((1009, "{a,NULL}", abi.NullValuesNotSupportedInArrayInDestination), (1001, "{NULL}", abi.NullValuesNotSupportedInArrayInDestination),
                           (1231, "{1.5,nUlL}", abi.NullValuesNotSupportedInArrayInDestination), (1231, "{123.456,1e-39,789.012}", abi.UnsupportedValueInDestination)):
        buf, offs = _stream([W.insert(42, ["1", "{}"]), W.insert(42, ["2", lit])])          # (the last: encoding.rs:385-404 with the 39 decimal places written short)
        hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("a", oid, True, 0)]), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.protobuf(0)
        assert ei.value.kind == kind and ei.value.detail == "Cell at index 1 failed validation" and ei.value.frame_index == 2, (oid, lit)
        b.close(); d.close()
    # (the reference's own vector writes the element out — 41 characters, one more than the device parses inside an array: handed back, the host raises)
    for oid, lit in ((1009, '{a,"b}'), (1001, "{abc}"), (1231, "{" + ",".join(K.NUMERIC_ARRAY_ROUNDING) + "}")):   # malformed literals: handed back (json[]: tests/test_gpu_json_display.py)
        buf, offs = _stream([W.insert(42, ["1", lit])])
        hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("a", oid, True, 0)]), buf, offs)
        r = b.protobuf(0)
        assert r.status == abi.RB_NEEDS_HOST and r.view.host_column == 1, oid
        r.close(); b.close(); d.close()
    # a malformed literal is the host's to report (the reference fails at decode time): handed back with its event and column
    buf, offs = _stream([W.insert(42, ["1", "{1,2}"]), W.insert(42, ["2", "{1,x}"])])
    hb, b, d = _both(SC.simple_table([("id", SC.INT8, False, 1), ("a", 1007, True, 0)]), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close(); b.close(); d.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_update_delete_streams(seed):
    """Random I / U / D traffic on a table with a three-column primary key (int4, text, uuid) scattered among nullable columns, under
    REPLICA IDENTITY DEFAULT (key images) and FULL (full old rows): every shape of update (no old image, unchanged key, each key
    column changed alone or together, text keys differing in one byte / in length / empty), deletes, partial updates — rows, row
    events and host counts as the oracle has them; the same arena through RowBinary (tombstone rows for the key-only deletes)."""
    import random
    rng = random.Random(seed)
    cols = [("n1", SC.INT4, True, 0), ("ka", SC.INT4, False, 1), ("t", 25, True, 0), ("kb", 25, False, 1), ("d", 1082, True, 0),
            ("kc", SC.UUID, False, 1), ("f", SC.FLOAT8, True, 0)]
    pk = [i for i, c in enumerate(cols) if c[3]]

    def key():
        return [str(rng.randrange(-5, 50)), rng.choice(["", "a", "ab", "abc", "abd", "é", "x" * rng.randrange(1, 40)]),
                "%08x-0000-4000-8000-%012x" % (rng.randrange(3), rng.randrange(3))]

    def row(k):
        r = [rng.choice([W.NULL, str(rng.randrange(100))]), None, rng.choice([W.NULL, "t%d" % rng.randrange(9)]), None,
             rng.choice([W.NULL, "2024-01-%02d" % rng.randrange(1, 28)]), None, rng.choice([W.NULL, "1.5", "-2.25"])]
        for j, i in enumerate(pk):
            r[i] = k[j]
        return r

    def keyimg(k):
        r = [W.NULL] * len(cols)
        for j, i in enumerate(pk):
            r[i] = k[j]
        return r
    for ident, name in (([1 if c[3] else 0 for c in cols], "PrimaryKey"), ([1] * len(cols), "Full")):
        msgs = []
        for _ in range(260):
            k = key()
            old = row(k)
            msgs.append(W.insert(42, old))
            m = rng.randrange(10)
            if m < 5:
                nk = list(k)
                for j in range(3):
                    if rng.random() < 0.3:
                        nk[j] = key()[j]
                new = row(nk)
                if name == "Full":
                    msgs.append(W.update(42, new, old=old))
                elif m == 0:
                    msgs.append(W.update(42, row(k)))                      # no old image: Postgres sends none when the key did not change
                else:
                    msgs.append(W.update(42, new, key=keyimg(k)))
            elif m == 5:
                t = row(k)
                t[2] = W.TOAST
                msgs.append(W.update(42, t))                               # partial (under FULL the old row resolves it: not partial)
            elif m < 8:
                msgs.append(W.delete(42, old=old) if name == "Full" else W.delete(42, key=keyimg(k)))
        buf, offs = _stream(msgs)
        hb, b, d = _both(SC.simple_table(cols, ident=ident), buf, offs)
        assert _check(hb, b, cols, name) > 260
        from tests.test_gpu_rowbinary import _check as rb_check
        flags = [1 if c[2] else 0 for c in cols] + [0, 0]
        rb_check(hb, b, flags, abi.CH_MERGE_TREE, name, cols)
        rb_check(hb, b, flags, abi.CH_REPLACING_MERGE_TREE, name, cols)
        b.close(); d.close()
