"""TEST INFRASTRUCTURE — CPU restatement of the reference's BigQuery row encoder, for the parity test of etlg_batch_protobuf
(etl_amd/csrc/columns.hip). Never imported by the product path.

Follows crates/etl-destinations/src/bigquery/encoding.rs: cell_encode_prost :120-190 (which protobuf field type every Cell
becomes), BigQueryTableRow::try_from :54-66 (tags = position + 1), and bigquery/core.rs:978-1036 + 1404-1754 (which rows an event
becomes: Insert -> one UPSERT row; Update -> the new row as UPSERT, behind a sparse DELETE row of the old primary key when the update
changed it; Delete -> a sparse DELETE row of the old primary key; the sequence key's trailing ordinal), crates/etl/src/event.rs:346-351
(EventSequenceKey Display). The row decisions are pinned by the reference's own tests (core.rs:2422-2600) in tests/test_oracle_protobuf.py.

PINNED by the reference's own tests, transcribed in tests/golden/bigquery_kats.py and checked by tests/test_oracle_protobuf.py: TimestampTz
cells are int64 epoch-microsecond varints and TimestampTz arrays the PACKED form of the same (encoding.rs:451-480); the numeric scale rule
(38 decimal places pass, 39 fail: validation.rs:20-35, encoding.rs:483-496) incl. inside arrays with the element's index (:385-404); the
JSON integer-precision rule (:343-360, validation.rs:44-93); NULL elements of arrays (:372-383); which rows an event becomes (core.rs,
see above). UNPINNED, because the reference compares against prost's own output and prost is an un-vendored dependency: the byte forms of
the remaining classes — bool / int32 / uint32 varints, float / double fixed words, length-delimited strings — which are the protobuf
encoding specification (varint keys, wire types 0 / 1 / 2 / 5, int32 / int64 as sign-extended 64-bit varints). Date / time strings:
chrono's %Y-%m-%d, %H:%M:%S%.f (etl-postgres/src/time.rs:13-21; %.f prints nothing, or 3 / 6 / 9 digits).

Works on the per-cell tuples of etl_amd.view.HostBatch.materialize()."""
import datetime as dt
import struct

from oracle.display import numeric_string, time_string, timetz_string
from oracle.rowbinary import ARRAY_OIDS, NeedsHost


class UnsupportedValueInDestination(Exception):
    """validate_numeric_for_bigquery (bigquery/validation.rs:20-35) behind BigQueryTableRow::try_from_tagged_cells (encoding.rs:37-45)."""


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def key(tag, wire_type):
    return varint((tag << 3) | wire_type)


def ld(tag, b):
    return key(tag, 2) + varint(len(b)) + b


def date_string(days_ce):
    d = dt.date.fromordinal(days_ce)
    return f"{d.year:04}-{d.month:02}-{d.day:02}"


def cell(c, tag):
    k = c[0]
    if k == "Null":
        return b""
    if k == "Bool":
        return key(tag, 0) + varint(1 if c[1] else 0)
    if k in ("I16", "I32", "I64"):
        return key(tag, 0) + varint(c[1])                    # sign-extended to 64 bits
    if k == "U32":
        return key(tag, 0) + varint(c[1])
    if k == "F32":
        return key(tag, 5) + struct.pack("<I", c[1])          # materialize() keeps the bits
    if k == "F64":
        return key(tag, 1) + struct.pack("<Q", c[1])
    if k == "String" or k == "Bytes":
        return ld(tag, c[1])
    if k == "Date":
        return ld(tag, date_string(c[1]).encode())
    if k == "Time":
        return ld(tag, time_string(c[1], c[2]).encode())
    if k == "Timestamp":
        return ld(tag, (date_string(c[1]) + " " + time_string(c[2], c[3])).encode())
    if k == "TimestampTz":
        return key(tag, 0) + varint(tstz_micros(c))
    if k == "TimeTz":                                # t.to_string() (encoding.rs:158-161)
        return ld(tag, timetz_string(*c[1:]).encode())
    if k == "Numeric":                               # validate_cell_for_bigquery, then n.to_string() (encoding.rs:146-149)
        if c[1] == 0 and c[4] > 38:
            raise UnsupportedValueInDestination(f"Cell at index {tag - 1} failed validation")
        return ld(tag, numeric_string(*c[1:]).encode())
    if k == "Uuid":
        h = c[1].hex()
        return ld(tag, f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}".encode())
    if k == "Deferred" and c[1] in ARRAY_OIDS:        # validate_array_cell_for_bigquery, then array_cell_encode_prost (encoding.rs:203-290)
        return array_cell(c[1], c[2], tag)
    if k == "Deferred" and c[1] not in (114, 3802):
        from oracle import arrays
        if c[1] in arrays.JSON_ARRAY_OIDS:           # json[]: reject_nulls, validate_elements(validate_json_for_bigquery), one string field per element
            from oracle import json_display as J
            els = arrays.split_literal(c[1], c[2])
            if any(e is not None and len(e) > arrays.JSON_ELEM_MAX for e in els):   # (the device does not look at such an element: the host's)
                raise NeedsHost("a json element too long for the device")
            if any(e is None for e in els):
                raise NullValuesNotSupportedInArrayInDestination(f"Cell at index {tag - 1} failed validation")
            inside = [e for e in els if J.device_limits_ok(e)]
            for e in inside:
                try:
                    validate_json_for_bigquery(e)
                except UnsupportedValueInDestination:
                    raise UnsupportedValueInDestination(f"Cell at index {tag - 1} failed validation") from None
            if len(inside) != len(els):
                raise NeedsHost("a json element beyond json_display's limits")
            return b"".join(ld(tag, J.display(e)) for e in els)
        if c[1] in arrays.VAR_ARRAY_OIDS or arrays.is_string_array(c[1], c[2]):            # one string / bytes field per element (encoding.rs:215-219, 244-249, 262-267, 285-289)
            items = arrays.elements(c[1], c[2])
            if any(e is None for e, _ in items):
                raise NullValuesNotSupportedInArrayInDestination(f"Cell at index {tag - 1} failed validation")
            if any(sc is not None and sc > 38 for _, sc in items):   # validate_elements(validate_numeric_for_bigquery)
                raise UnsupportedValueInDestination(f"Cell at index {tag - 1} failed validation")
            return b"".join(ld(tag, e) for e, _ in items)
    if k == "Deferred" and c[1] in (114, 3802):      # validate_json_for_bigquery, then j.to_string() (encoding.rs:173-176)
        from oracle import json_display
        if not json_display.device_limits_ok(c[2]):
            raise NeedsHost("json beyond json_display's limits")
        try:
            validate_json_for_bigquery(c[2])
        except UnsupportedValueInDestination:
            raise UnsupportedValueInDestination(f"Cell at index {tag - 1} failed validation") from None
        return ld(tag, json_display.display(c[2]))
    raise NeedsHost(k)


class NullValuesNotSupportedInArrayInDestination(Exception):
    """reject_nulls (bigquery/validation.rs:126-141)."""


def array_cell(type_oid, text, tag):
    """An array cell of a fixed-width element class (the ones the device encodes): reject_nulls (validation.rs:127-141, wrapped by
    try_from_tagged_cells with the cell's index), then bool / int32 / uint32 / int64 / float / double / timestamptz PACKED
    (prost::encoding::*::encode_packed: nothing for an empty array) and date / time / timestamp / uuid as one string field per element."""
    from oracle.rowbinary import array_elements
    elems = array_elements(type_oid, text)             # raises NeedsHost for what the device hands back
    if any(e[0] == "Null" for e in elems):
        raise NullValuesNotSupportedInArrayInDestination(f"Cell at index {tag - 1} failed validation")
    if not elems:
        return b""
    k = elems[0][0]
    if k == "Bool":
        return ld(tag, bytes(1 if e[1] else 0 for e in elems))
    if k in ("I16", "I32", "I64", "U32"):
        return ld(tag, b"".join(varint(e[1]) for e in elems))
    if k == "F32":
        return ld(tag, b"".join(struct.pack("<I", e[1]) for e in elems))
    if k == "F64":
        return ld(tag, b"".join(struct.pack("<Q", e[1]) for e in elems))
    if k == "TimestampTz":
        return packed_int64(tag, [tstz_micros(e) for e in elems])
    return b"".join(cell(e, tag) for e in elems)     # Date / Time / Timestamp / Uuid: the scalar's string field, repeated


def packed_int64(tag, values):
    """prost::encoding::int64::encode_packed (encoding.rs:212-260 use it for the integer / TimestampTz arrays): one length-delimited
    field holding the varints back to back; nothing at all for an empty array."""
    if not values:
        return b""
    return ld(tag, b"".join(varint(v) for v in values))


def tstz_micros(c):
    return ((c[1] - 719163) * 86400 + c[2]) * 1_000_000 + c[3] // 1000


def validate_json_for_bigquery(text):
    """validate_json_for_bigquery / validate_json_number_for_bigquery (bigquery/validation.rs:44-93) on the cell's JSON text (the
    reference parses with serde_json's arbitrary_precision: a number keeps its literal): an integer literal — no '.', 'e', 'E' —
    outside i64 (when negative) / u64 would be stored as FLOAT64 and is refused; everything else is left to BigQuery."""
    from oracle import json_display
    for lit in json_display.numbers(json_display.parse(text)):   # the PARSED value: what a repeated key lost is not looked at
        if not any(ch in lit for ch in ".eE"):
            v = int(lit)
            if (lit.startswith("-") and not (-(1 << 63) <= v < (1 << 63))) or (not lit.startswith("-") and not (0 <= v < (1 << 64))):
                raise UnsupportedValueInDestination("JSON integer would lose precision in BigQuery")


def validate_array_for_bigquery(elem_kind, elems, cell_index=0):
    """validate_array_cell_for_bigquery (validation.rs:119-190) behind try_from_tagged_cells (encoding.rs:37-45: the cell's index goes
    into the detail). elems: Python values or None; numeric elements as (kind, neg, weight, scale, digits) tuples like materialize()'s."""
    nulls = sum(1 for e in elems if e is None)
    if nulls:
        raise NullValuesNotSupportedInArrayInDestination(f"Cell at index {cell_index} failed validation")
    if elem_kind == "Numeric":
        for i, e in enumerate(elems):
            if e[0] == 0 and e[3] > 38:
                raise UnsupportedValueInDestination(f"Cell at index {cell_index} failed validation: Element at index {i} would be rounded by BigQuery")


# classes whose Cell equality is the equality of the arena's words / bytes (bigquery_primary_key_changed compares Cells: float
# NaN != NaN and 0.0 == -0.0, numeric / timetz / json / array cells are compared as parsed values): updates of a table whose
# primary key has a column of another class stay with the host when they carry an old row
PK_COMPARABLE = {"Bool", "I16", "I32", "I64", "U32", "Date", "Time", "Timestamp", "TimestampTz", "Uuid", "String", "Bytes", "Null"}


def seq(e, ordinal):
    return f"{e['commit_lsn']:016x}/{e['tx_ordinal']:016x}/{ordinal:016x}".encode()


def upsert_row(e, cells, ordinal):
    n = len(cells)
    return b"".join(cell(c, t + 1) for t, c in enumerate(cells)) + ld(n + 1, b"UPSERT") + ld(n + 2, seq(e, ordinal))


def delete_row(e, tagged, n_cols, ordinal):
    """bigquery_delete_row (core.rs:1742-1754): the primary-key cells under their column tags, then DELETE and the sequence key."""
    return b"".join(cell(c, t) for t, c in tagged) + ld(n_cols + 1, b"DELETE") + ld(n_cols + 2, seq(e, ordinal))


def pk_tagged(old_kind, old_row, schema_cols):
    """bigquery_primary_key_tagged_cells_from_old_row (:1647-1739) for a key image / a full old row of the right width."""
    pk = [i for i, c in enumerate(schema_cols) if c[3]]
    if old_kind == "Key":
        return [(i + 1, old_row[k]) for k, i in enumerate(pk)]
    return [(i + 1, old_row[i]) for i in pk]


def event_rows(events, slot_index, schema_cols=None, identity_type="PrimaryKey"):
    """(list of row bytes, event index of every row, events of the slot left to the host). Without `schema_cols`: Insert rows only
    (the round-2 contract). With them: what bigquery/core.rs:978-1036 builds for every Insert / Update / Delete event the reference
    accepts; an event it refuses (a partial update, a delete without an old row, a key image under another replica identity, an
    update without an old row under another replica identity) — and an update with an old row whose primary key has a column of a
    class outside PK_COMPARABLE, or a cell of it DEFERRED — stays with the host."""
    rows, idx, host = [], [], 0
    for i, e in enumerate(events):
        if e["kind"] not in "IUD" or e.get("schema_slot") != slot_index:
            continue
        if e["kind"] == "I":
            rows.append(upsert_row(e, e["row"], 0)); idx.append(i)
            continue
        if schema_cols is None:
            host += 1
            continue
        n = len(schema_cols)
        n_pk = sum(1 for c in schema_cols if c[3])
        ok, old = e["old_kind"], e.get("old_row")
        if e["kind"] == "D":
            if ok == "Full" or (ok == "Key" and identity_type == "PrimaryKey" and len(old) == n_pk):
                rows.append(delete_row(e, pk_tagged(ok, old, schema_cols), n, 0)); idx.append(i)
            else:
                host += 1
            continue
        if e["partial"]:
            host += 1
            continue
        new = e["row"]
        if ok == "None":   # ensure_bigquery_update_without_old_row_can_skip_delete (:1515-1536)
            if identity_type != "PrimaryKey":
                host += 1
                continue
            rows.append(upsert_row(e, new, 0)); idx.append(i)
            continue
        if ok == "Key" and (identity_type != "PrimaryKey" or len(old) != n_pk):
            host += 1
            continue
        pairs = [(c, new[t - 1]) for t, c in pk_tagged(ok, old, schema_cols)]
        if any(a[0] not in PK_COMPARABLE or b[0] not in PK_COMPARABLE for a, b in pairs):
            host += 1
            continue
        if any(a != b for a, b in pairs):   # bigquery_primary_key_changed (:1557-1645)
            rows.append(delete_row(e, pk_tagged(ok, old, schema_cols), n, 0)); idx.append(i)
            rows.append(upsert_row(e, new, 1)); idx.append(i)
        else:
            rows.append(upsert_row(e, new, 0)); idx.append(i)
    return rows, idx, host


def insert_rows(events, slot_index):
    """(list of row bytes, event indices, events of the slot left to the host)."""
    return event_rows(events, slot_index)
