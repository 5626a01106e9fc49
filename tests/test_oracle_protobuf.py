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
