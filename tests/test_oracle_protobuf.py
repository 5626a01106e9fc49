"""oracle/protobuf.py's row decisions against the reference's own tests of bigquery/core.rs (no GPU): which rows an Update / Delete
event becomes, their tags and their sequence keys."""
from oracle import protobuf as PB

USERS = [("id", 23, False, 1), ("name", 25, False, 0), ("age", 23, False, 0)]
TWO = [("id", 23, False, 1), ("name", 25, True, 0)]     # replicated_schema(..) of the reference's tests: id int4 primary key, name text


def _ev(kind, **kw):
    e = {"kind": kind, "schema_slot": 0, "commit_lsn": 1, "tx_ordinal": 0, "partial": False, "old_kind": "None", "old_row": None, "row": None}
    e.update(kw)
    return e


def expected(tagged):
    """expected_row_bytes of the reference's tests: the tagged cells through cell_encode_prost."""
    return b"".join(PB.cell(c, t) for t, c in tagged)


def test_delete_key_row_tags_line_up_with_the_descriptor():            # core.rs:2422-2463 (literal byte windows)
    row = PB.delete_row({"commit_lsn": 0, "tx_ordinal": 0}, PB.pk_tagged("Key", [("I32", 42)], USERS), 3, 0)
    assert bytes([0x08, 0x2A]) in row                                   # 1 => id
    assert bytes([0x22, 0x06]) + b"DELETE" in row                       # 4 => _CHANGE_TYPE
    assert row[row.index(b"DELETE") + 6] == 0x2A                        # 5 => _CHANGE_SEQUENCE_NUMBER
    assert 0x12 not in row and 0x18 not in row                           # the other source columns are omitted (no tag 2 / 3 key byte)


def test_delete_full_row_omits_non_primary_key_columns():               # :2466-2501
    rows, idx, host = PB.event_rows([_ev("D", old_kind="Full", old_row=[("I32", 42), ("String", b"alice"), ("I32", 7)])], 0, USERS, "Full")
    assert host == 0 and rows == [expected([(1, ("I32", 42)), (4, ("String", b"DELETE")),
                                            (5, ("String", b"0000000000000001/0000000000000000/0000000000000000"))])]


def test_update_rows_emit_delete_before_upsert_when_the_primary_key_changes():   # :2504-2534
    rows, idx, host = PB.event_rows([_ev("U", old_kind="Key", old_row=[("I32", 1)], row=[("I32", 2), ("String", b"updated")])], 0, TWO)
    assert idx == [0, 0] and host == 0
    assert rows[0] == expected([(1, ("I32", 1)), (3, ("String", b"DELETE")), (4, ("String", b"0000000000000001/0000000000000000/0000000000000000"))])
    assert rows[1] == expected([(1, ("I32", 2)), (2, ("String", b"updated")), (3, ("String", b"UPSERT")),
                                (4, ("String", b"0000000000000001/0000000000000000/0000000000000001"))])


def test_update_rows_skip_the_delete_when_the_primary_key_is_unchanged():        # :2537-2558
    rows, idx, host = PB.event_rows([_ev("U", old_kind="Key", old_row=[("I32", 1)], row=[("I32", 1), ("String", b"updated")])], 0, TWO)
    assert rows == [expected([(1, ("I32", 1)), (2, ("String", b"updated")), (3, ("String", b"UPSERT")),
                              (4, ("String", b"0000000000000001/0000000000000000/0000000000000000"))])]


def test_update_rows_for_full_identity_primary_key_change():                       # :2561-2600
    rows, idx, host = PB.event_rows([_ev("U", old_kind="Full", old_row=[("I32", 1), ("String", b"before")], row=[("I32", 2), ("String", b"updated")])],
                                    0, TWO, "Full")
    assert len(rows) == 2 and rows[0] == expected([(1, ("I32", 1)), (3, ("String", b"DELETE")),
                                                   (4, ("String", b"0000000000000001/0000000000000000/0000000000000000"))])


def test_events_the_reference_refuses():                                            # :2106-2150
    evs = [_ev("U", partial=True, row=[("I32", 1), ("String", b"x")]),                                      # bigquery_update_new_row_rejects_partial_rows
           _ev("D"),                                                                                         # bigquery_delete_old_row_rejects_missing_old_rows
           _ev("D", old_kind="Key", old_row=[("I32", 1)]),                                                   # ensure_bigquery_key_image_matches_primary_key (AlternativeKey)
           _ev("U", row=[("I32", 1), ("String", b"x")])]                                                     # ensure_bigquery_update_without_old_row (AlternativeKey)
    rows, idx, host = PB.event_rows(evs, 0, TWO, "AlternativeKey")
    assert rows == [] and host == 4


# ---- cell encodings and validation rules the reference's tests pin (tests/golden/bigquery_kats.py) ----------------------------------
from tests.golden import bigquery_kats as K   # noqa: E402


def test_timestamptz_is_an_int64_of_epoch_microseconds_scalar_and_packed():     # encoding.rs:451-480
    import datetime as dt
    assert K.TSTZ_MICROS == int(dt.datetime(2026, 1, 2, 3, 4, 5, tzinfo=dt.timezone.utc).timestamp()) * 1_000_000   # chrono's timestamp_micros()
    assert PB.tstz_micros(K.TSTZ_CELL) == K.TSTZ_MICROS
    assert PB.cell(K.TSTZ_CELL, 1) == K.TSTZ_SCALAR_BYTES
    assert PB.packed_int64(1, [PB.tstz_micros(K.TSTZ_CELL)]) == K.TSTZ_PACKED_BYTES
    assert PB.packed_int64(1, []) == b""


def _numeric_cell(text):
    """("Numeric", kind, sign, weight, scale, digits) of a plain decimal, the way the arena holds it (base-10000 groups)."""
    neg = text.startswith("-")
    t = text.lstrip("+-")
    ip, _, fp = t.partition(".")
    scale = len(fp)
    ip = ip.lstrip("0")
    ipad = "0" * ((-len(ip)) % 4) + ip
    fpad = fp + "0" * ((-len(fp)) % 4)
    groups = [int(ipad[i:i + 4]) for i in range(0, len(ipad), 4)] + [int(fpad[i:i + 4]) for i in range(0, len(fpad), 4)]
    weight = len(ipad) // 4 - 1
    while groups and groups[0] == 0:
        groups.pop(0); weight -= 1
    while groups and groups[-1] == 0:
        groups.pop()
    return ("Numeric", 0, 1 if neg else 0, weight if groups else 0, scale, groups)


def test_numeric_scale_rule_38_passes_39_fails():                                   # encoding.rs:483-496, validation.rs:20-35 / 213-229
    import pytest
    assert PB.cell(_numeric_cell(K.NUMERIC_AT_SCALE), 1) == PB.ld(1, K.NUMERIC_AT_SCALE.encode())
    with pytest.raises(PB.UnsupportedValueInDestination):
        PB.cell(_numeric_cell(K.NUMERIC_OVER_SCALE), 1)
    with pytest.raises(PB.UnsupportedValueInDestination) as e:                      # first failing cell wins (:420-438): tag 1 = cell index 0
        PB.upsert_row({"commit_lsn": 1, "tx_ordinal": 0}, [_numeric_cell(K.NUMERIC_OVER_SCALE), _numeric_cell("0.5")], 0)
    assert "Cell at index 0" in str(e.value)


def test_arrays_null_elements_and_numeric_rounding_inside_an_array():             # encoding.rs:372-418
    import pytest
    with pytest.raises(PB.NullValuesNotSupportedInArrayInDestination) as e:
        PB.validate_array_for_bigquery("I32", K.ARRAY_WITH_NULLS)
    assert "Cell at index 0 failed validation" in str(e.value)
    PB.validate_array_for_bigquery("I32", K.ARRAY_VALID)
    with pytest.raises(PB.UnsupportedValueInDestination) as e:
        PB.validate_array_for_bigquery("Numeric", [_numeric_cell(t)[1:] for t in K.NUMERIC_ARRAY_ROUNDING])
    assert "Cell at index 0" in str(e.value) and "Element at index 1" in str(e.value)


def test_json_integer_precision_rule():                                            # encoding.rs:343-360, validation.rs:44-93
    import pytest
    for t in K.JSON_ACCEPTED:
        PB.validate_json_for_bigquery(t)
    for t in K.JSON_REFUSED:
        with pytest.raises(PB.UnsupportedValueInDestination):
            PB.validate_json_for_bigquery(t)
