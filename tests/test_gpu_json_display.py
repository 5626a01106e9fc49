"""json / jsonb cells as the sinks write them — serde_json's `Value::to_string()` — built on the device (json_display,
etl_amd/csrc/columns.hip) for all three hand-off formats, against oracle/json_display.py (pinned by tests/golden/json_display_kats.py to
the reference's own vectors): etlg_batch_columns with ETLG_ROWS_FORMAT_JSON (LargeUtf8), etlg_batch_rowbinary (String), etlg_batch_protobuf
(string field behind validate_json_for_bigquery). Known answers, seeded random documents (repeated keys, escapes, every number shape,
whitespace), the cells a lane leaves to the host, malformed cells, and the order of the reports."""
import json
import random

import numpy as np
import pytest

from etl_amd import abi
from tests import pgwire as W
from tests import scenarios as SC
from tests.golden import json_display_kats as K
from tests.test_gpu_rowbinary import _both, _stream

pytestmark = pytest.mark.gpu

COLS = [("id", SC.INT8, False, 1), ("j", 114, True, 0), ("jb", SC.JSONB, True, 0)]


def _varint(v):
    from oracle.rowbinary import varint
    return varint(v)


def _three_ways(texts):
    """Every text as the `j` cell of a row of its own (jb: the same list backwards): the Display strings through the three calls."""
    from oracle import json_display as J
    from oracle import protobuf as PB
    from oracle import rowbinary as RB
    rows = [[str(i), t, texts[-1 - i]] for i, t in enumerate(texts)] + [[str(len(texts)), W.NULL, W.NULL]]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(COLS), buf, offs)
    want = [J.display(t) for t in texts]
    # Arrow: LargeUtf8 of the Display strings, nothing deferred
    c = b.columns(0, format_json=True)
    for ci, exp in ((1, want), (2, want[::-1])):
        col = c.column(ci)
        assert col.arrow_kind == abi.AK_LARGE_UTF8 and int(col.deferred_count) == 0 and int(col.null_count) == 1
        _validity, _deferred, values, o = c.host_arrays(ci)
        data = values.tobytes()
        got = [data[o[k]:o[k + 1]] for k in range(len(texts))]
        assert got == exp, [(texts[k], got[k], exp[k]) for k in range(len(texts)) if got[k] != exp[k]][:3]
        assert o[len(texts) + 1] == o[len(texts)]
    c.close()
    # RowBinary and protobuf: the oracle's rows (String(j.to_string()) / a length-delimited field)
    ev = hb.materialize()
    slot = hb.slots[0]
    rrows, idx, host = RB.encode_events(ev, 0, [k.type_class for k in slot.cols], [0, 1, 1, 0, 0], abi.CH_MERGE_TREE, "PrimaryKey", None)
    r = b.rowbinary(0, [0, 1, 1, 0, 0], abi.CH_MERGE_TREE)
    assert r.status == abi.RB_OK and r.n_rows == len(rrows) == len(rows) and host == 0
    assert r.bytes().tobytes() == b"".join(rrows)
    r.close()
    try:
        prow, idx, host = PB.event_rows(ev, 0, COLS, "PrimaryKey")
    except PB.UnsupportedValueInDestination as ex:      # an integer literal BigQuery would round: the batch fails there as well
        from etl_amd.decoder import EtlError
        with pytest.raises(EtlError) as ei:
            b.protobuf(0)
        assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.detail == str(ex)
    else:
        r = b.protobuf(0)
        assert r.status == abi.RB_OK and r.n_rows == len(prow) == len(rows)
        assert r.bytes().tobytes() == b"".join(prow)
        r.close()
    # (and the oracle's rows do hold the strings: the first row is id, then the two cells)
    assert rrows[0].startswith(b"\x00" * 8 + b"\x00" + _varint(len(want[0])) + want[0])
    b.close(); d.close()


def test_known_answers():
    from oracle import json_display as J
    for src, exp in K.PINNED + K.RESTATED:
        assert J.display(src).decode() == exp, src          # (the oracle itself: also in tests/test_oracle_json_display.py)
    _three_ways([src for src, _ in K.PINNED + K.RESTATED])


def _random_doc(rng, depth=0):
    k = rng.random()
    if depth >= 5 or k < 0.35:
        c = rng.randrange(9)
        if c == 0:
            return rng.choice(["null", "true", "false"])
        if c in (1, 2):
            return rng.choice(["0", "-0", "7", "-12", "18446744073709551615", "18446744073709551616", "-9223372036854775808", "-9223372036854775809",
                               "0.5", "-0.0", "1e5", "1E5", "1e+5", "1E-5", "2.50e10", "123456789012345678901234567890.5", "1e309", str(rng.randrange(-10**6, 10**6))])
        return _random_string(rng)
    ws = lambda: rng.choice(["", "", " ", "\n", "\t ", "  \r\n"])   # noqa: E731
    if k < 0.65:
        return "[" + ws() + ("," + ws()).join(_random_doc(rng, depth + 1) + ws() for _ in range(rng.choice([0, 1, 2, 3, 5]))) + "]"
    keys = [_random_string(rng, key=True) for _ in range(rng.choice([0, 1, 2, 3, 4, 6, 9]))]
    if keys and rng.random() < 0.4:
        keys += [rng.choice(keys) for _ in range(rng.randrange(1, 3))]        # repeated keys
        rng.shuffle(keys)
    return "{" + ws() + ("," + ws()).join(k2 + ws() + ":" + ws() + _random_doc(rng, depth + 1) + ws() for k2 in keys) + "}"


def _random_string(rng, key=False):
    parts = []
    for _ in range(rng.choice([0, 1, 1, 2, 3, 8] if key else [0, 1, 3, 10, 40])):
        parts.append(rng.choice(["a", "b", "ab", "z", "A", "0", " ", "é", "中", "😀", "\x7f", "/", "\\/", '\\"', "\\\\", "\\b", "\\f", "\\n", "\\r", "\\t",
                                 "\\u0041", "\\u00e9", "\\u0000", "\\u001f", "\\u0061", "\\uD83D\\uDE00", "\\u4e2d", "$", "k"]))
    return '"' + "".join(parts) + '"'


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_documents(seed):
    from oracle import json_display as J
    rng = random.Random(seed)
    texts = []
    while len(texts) < 120:
        t = rng.choice(["", " ", "\n"]) + _random_doc(rng) + rng.choice(["", " ", "\t\n"])
        json.loads(t)                       # the generator writes JSON
        if J.device_limits_ok(t):
            texts.append(t)
    assert sum(1 for t in texts if J.display(t) != t.encode()) > 60       # most of them do change
    _three_ways(texts)

    def bq_ok(t):
        from oracle import protobuf as PB
        try:
            PB.validate_json_for_bigquery(t)
        except PB.UnsupportedValueInDestination:
            return False
        return True
    keep = [t for t in texts if bq_ok(t)]
    assert 30 < len(keep) < len(texts)
    _three_ways(keep)                                                      # (the protobuf rows of a batch BigQuery's rule lets through)


def test_cells_a_lane_leaves_to_the_host():
    """Nesting deeper than 16, an object of more than 64 members, serde_json's private token as a key: the Arrow column keeps the source
    text with the cell's bit in `deferred`; RowBinary / protobuf report the first such cell as host_event / host_column."""
    from oracle import json_display as J
    deep = "[" * 17 + "]" * 17
    wide = "{" + ",".join(f'"k{i}":{i}' for i in range(65)) + "}"
    token = '{"$serde_json::private::Number":"1"}'
    ok = ["[" * 16 + "]" * 16, "{" + ",".join(f'"k{i:02}":{64 - i}' for i in range(64)) + "}", '{"a":' * 15 + "[1]" + "}" * 15]
    for t in (deep, wide, token):
        assert not J.device_limits_ok(t)
    for t in ok:
        assert J.device_limits_ok(t)
    texts = [ok[0], deep, ok[1], wide, token, ok[2]]
    rows = [[str(i), t, W.NULL] for i, t in enumerate(texts)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(COLS), buf, offs)
    c = b.columns(0, format_json=True)
    col = c.column(1)
    _validity, deferred, values, o = c.host_arrays(1)
    data = values.tobytes()
    got = [data[o[k]:o[k + 1]] for k in range(len(texts))]
    assert got == [J.display(t) if J.device_limits_ok(t) else t.encode() for t in texts]
    bits = np.unpackbits(deferred, bitorder="little")[:len(texts)]
    assert list(bits) == [0, 1, 0, 1, 1, 0] and int(col.deferred_count) == 3
    c.close()
    r = b.rowbinary(0, [0, 1, 1, 0, 0], abi.CH_MERGE_TREE)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close()
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close()
    b.close(); d.close()


def test_malformed_cells_fail_like_the_decode():
    """A json cell that is not one JSON value is the reference's decode error (codec/text.rs:126-134) in the row formats too — at its
    event, and before any report of the sink itself (a date out of range / a NULL in an earlier row, a BigQuery validation failure
    in an earlier row or an earlier column)."""
    from etl_amd.decoder import EtlError
    cols = [("id", SC.INT8, False, 1), ("d", SC.DATE, True, 0), ("j", 114, True, 0), ("n", SC.NUMERIC, True, 0)]
    good = ["1", "2000-01-01", '{"a":1}', "1.5"]
    for bad_text in ("{", '{"a":1,}', '"\\uD83D"', "01", ""):
        msgs = [W.insert(42, good), W.insert(42, ["2", "1800-01-01", "[]", "1e-40"]), W.insert(42, ["3", W.NULL, bad_text, "2"]), W.insert(42, good)]
        buf, offs = _stream(msgs)
        hb, b, d = _both(SC.simple_table(cols), buf, offs)
        for call in (lambda: b.rowbinary(0, [0, 0, 1, 1, 0, 0], abi.CH_MERGE_TREE), lambda: b.protobuf(0)):
            with pytest.raises(EtlError) as ei:
                call()
            assert ei.value.code == abi.E_JSON and ei.value.kind == abi.DeserializationError and ei.value.frame_index == 3, bad_text
            assert ei.value.description == "JSON deserialization failed"
        b.close(); d.close()
    # without the malformed cell the sink's own reports are back: the date out of range (row 2) / the numeric scale (row 2, column 3)
    buf, offs = _stream([W.insert(42, good), W.insert(42, ["2", "1800-01-01", "[]", "1e-40"]), W.insert(42, good)])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(EtlError) as ei:
        b.rowbinary(0, [0, 1, 1, 1, 0, 0], abi.CH_MERGE_TREE)
    assert ei.value.description == "Date out of ClickHouse Date32 range" and ei.value.frame_index == 2
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.detail == "Cell at index 3 failed validation"
    b.close(); d.close()


def test_bigquery_integer_rule_on_the_device():
    """validate_json_for_bigquery (bigquery/validation.rs:47-85; the reference's vectors in tests/golden/bigquery_kats.py): an integer
    literal outside u64 / i64 anywhere in the PARSED value fails the row with the reference's kind and detail; exponents, fractions and
    strings are BigQuery's to judge; what a repeated key lost is not in the parsed value and is not looked at."""
    from etl_amd.decoder import EtlError
    from oracle import protobuf as PB
    from tests.golden import bigquery_kats as BK
    cols = [("id", SC.INT8, False, 1), ("j", SC.JSONB, True, 0)]
    accepted = list(BK.JSON_ACCEPTED) + ['{"a":99999999999999999999,"a":1}', '[18446744073709551615,-9223372036854775808,{"x":1e400}]']
    buf, offs = _stream([W.insert(42, [str(i), t]) for i, t in enumerate(accepted)])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    rows, idx, host = PB.event_rows(hb.materialize(), 0, cols, "PrimaryKey")
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.n_rows == len(accepted) and r.bytes().tobytes() == b"".join(rows)
    r.close(); b.close(); d.close()
    for t in list(BK.JSON_REFUSED) + ['{"a":1,"a":99999999999999999999}', "[0,-9223372036854775809]", "18446744073709551616"]:
        with pytest.raises(PB.UnsupportedValueInDestination):
            PB.validate_json_for_bigquery(t)
        buf, offs = _stream([W.insert(42, ["1", "{}"]), W.insert(42, ["2", t]), W.insert(42, ["3", "[]"])])
        hb, b, d = _both(SC.simple_table(cols), buf, offs)
        with pytest.raises(EtlError) as ei:
            b.protobuf(0)
        assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.description == "Cell validation failed for BigQuery compatibility"
        assert ei.value.detail == "Cell at index 1 failed validation" and ei.value.frame_index == 2
        # (ClickHouse has no such rule: the same batch goes through RowBinary)
        r = b.rowbinary(0, [0, 1, 0, 0], abi.CH_MERGE_TREE)
        assert r.status == abi.RB_OK and r.n_rows == 3
        r.close(); b.close(); d.close()


def test_json_arrays_in_the_row_formats():
    """json[] / jsonb[] cells: String(j.to_string()) per element in RowBinary (clickhouse/encoding.rs:109), one string field per element
    in protobuf (bigquery/encoding.rs:279-284) behind reject_nulls and validate_elements(validate_json_for_bigquery) — quoting and
    escapes of the array literal undone first; an element that is not JSON is the decode error; elements beyond a lane's limits
    (longer than 256 bytes, nested deeper than 16) are handed back."""
    from etl_amd.decoder import EtlError
    from oracle import protobuf as PB
    from oracle import rowbinary as RB
    q = lambda t: '"' + t.replace("\\", "\\\\").replace('"', '\\"') + '"'     # noqa: E731  (an element as Postgres writes it inside an array literal)
    docs = [src for src, _ in K.PINNED + K.RESTATED if "123456789012345678901234567890" not in src]
    lits = ["{" + ",".join(q(t) for t in docs[k:k + 3]) + "}" for k in range(0, len(docs), 3)] + ["{}", "{1,2.50,true,null}", "{NULL," + q('{"b":1,"a":2}') + "}"]
    cols = [("id", SC.INT8, False, 1), ("ja", 199, True, 0), ("jb", 3807, True, 0)]
    rows = [[str(i), t, lits[-1 - i]] for i, t in enumerate(lits)] + [[str(len(lits)), W.NULL, W.NULL]]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    ev = hb.materialize()
    rrows, idx, host = RB.encode_events(ev, 0, [k.type_class for k in hb.slots[0].cols], [0, 1, 1, 0, 0], abi.CH_MERGE_TREE, "PrimaryKey", None)
    r = b.rowbinary(0, [0, 1, 1, 0, 0], abi.CH_MERGE_TREE)
    assert r.status == abi.RB_OK and r.n_rows == len(rows) and r.bytes().tobytes() == b"".join(rrows)
    r.close()
    with pytest.raises(EtlError) as ei:                 # the last literals hold a NULL element (the unquoted null, NULL)
        b.protobuf(0)
    assert ei.value.kind == abi.NullValuesNotSupportedInArrayInDestination
    b.close(); d.close()
    keep = [t for t in lits if "null" not in t.lower().replace('\\"', "")] or lits[:1]
    rows = [[str(i), t, keep[-1 - i]] for i, t in enumerate(keep)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    prow, idx, host = PB.event_rows(hb.materialize(), 0, cols, "PrimaryKey")
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.n_rows == len(rows) and r.bytes().tobytes() == b"".join(prow)
    r.close(); b.close(); d.close()
    for lit, what in (("{" + q("{") + "}", "json"), ("{1," + q("[1,]") + "}", "json"), ("{" + q("[" * 17 + "]" * 17) + "}", "host"), ("{" + q('"' + "x" * 260 + '"') + "}", "host"),
                      ("{" + q('{"n":18446744073709551616}') + "}", "bq")):
        buf, offs = _stream([W.insert(42, ["1", "{}", "{}"]), W.insert(42, ["2", lit, "{}"])])
        hb, b, d = _both(SC.simple_table(cols), buf, offs)
        for call, fmt in ((lambda: b.rowbinary(0, [0, 1, 1, 0, 0], abi.CH_MERGE_TREE), "rb"), (lambda: b.protobuf(0), "pb")):
            if what == "json":
                with pytest.raises(EtlError) as ei:
                    call()
                assert ei.value.code == abi.E_JSON and ei.value.frame_index == 2, (lit, fmt)
            elif what == "bq" and fmt == "pb":
                with pytest.raises(EtlError) as ei:
                    call()
                assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.detail == "Cell at index 1 failed validation"
            else:
                r = call()
                if what == "host":
                    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1), (lit, fmt)
                else:
                    assert r.status == abi.RB_OK
                r.close()
        b.close(); d.close()
