"""TEST INFRASTRUCTURE — CPU restatement of the reference's ClickHouse RowBinary encoder, for the parity tests of
etlg_batch_rowbinary (etl_amd/csrc/columns.hip). Never imported by the product path.

Follows crates/etl-destinations/src/clickhouse/encoding.rs:
  cell_to_clickhouse_value :58-83   (which wire type every Cell becomes; Date range check :147-161; bytes_to_hex :176-185)
  rb_varint :188-199 | rb_encode_nullable :202-211 | rb_encode_value :214-255 | encode_to_row_binary :259-283
and crates/etl-destinations/src/clickhouse/core.rs:
  which events become rows :1078-1127 | append_cdc_columns :96-114 (MergeTree: cdc_operation String + cdc_lsn UInt64;
  ReplacingMergeTree: _etl_version UInt128 = commit_lsn << 64 | tx_ordinal, _etl_deleted UInt8).
Pinned by tests/test_oracle_rowbinary.py to the byte vectors of the reference's own tests (encoding.rs:386-470).

`Time`, `TimeTz` and `Numeric` cells are strings in the reference (`to_string()`, encoding.rs:66-71): their Display impls are
restated in oracle/display.py (chrono's NaiveTime Display is a dependency that is not vendored under /root/reference).

Works on the per-cell tuples of etl_amd.view.HostBatch.materialize()."""
import struct

from oracle.display import numeric_string, time_string, timetz_string   # noqa: F401  (time_string is re-exported)

MERGE_TREE, REPLACING_MERGE_TREE = 0, 1
# array types whose element class the device encodes: bool int2 int4 int8 oid float4 float8 date time timestamp timestamptz uuid
ARRAY_OIDS = {1000, 1005, 1007, 1016, 1028, 1021, 1022, 1182, 1183, 1115, 1185, 2951}
CE_DAYS_1970 = 719163
DATE32_MIN, DATE32_MAX = -25567, 120529        # 1900-01-01, 2299-12-31 as days since 1970-01-01


class ConversionError(Exception):
    pass


class NeedsHost(Exception):
    """The cell has no device encoding (json / array text, or a DEFERRED cell)."""


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v == 0:
            out.append(b)
            return bytes(out)
        out.append(b | 0x80)


def string(b):
    return varint(len(b)) + b


def value(cell):
    """rb_encode_value(cell_to_clickhouse_value(cell)) for a non-null cell."""
    k = cell[0]
    if k == "Bool":
        return bytes([1 if cell[1] else 0])
    if k == "I16":
        return struct.pack("<h", cell[1])
    if k == "I32":
        return struct.pack("<i", cell[1])
    if k == "I64":
        return struct.pack("<q", cell[1])
    if k == "U32":
        return struct.pack("<I", cell[1])
    if k == "F32":
        return struct.pack("<I", cell[1])     # materialize() keeps the bits
    if k == "F64":
        return struct.pack("<Q", cell[1])
    if k == "Date":
        days = cell[1] - CE_DAYS_1970
        if days < DATE32_MIN or days > DATE32_MAX:
            raise ConversionError("Date out of ClickHouse Date32 range")
        return struct.pack("<i", days)
    if k == "Time":
        return string(time_string(cell[1], cell[2]).encode())
    if k == "TimeTz":                                  # String(t.to_string()) (encoding.rs:71)
        return string(timetz_string(*cell[1:]).encode())
    if k == "Numeric":                                 # String(n.to_string()) (encoding.rs:66)
        return string(numeric_string(*cell[1:]).encode())
    if k in ("Timestamp", "TimestampTz"):
        days, secs, nanos = cell[1:]
        return struct.pack("<q", ((days - CE_DAYS_1970) * 86400 + secs) * 1_000_000 + nanos // 1000)
    if k == "Uuid":
        hi, lo = struct.unpack(">QQ", cell[1])
        return struct.pack("<QQ", hi, lo)
    if k == "Bytes":
        return string(cell[1].hex().encode())
    if k == "String":
        return string(cell[1])
    if k == "Deferred" and cell[1] in ARRAY_OIDS:   # array columns keep their literal in the arena
        return array(cell[1], cell[2])
    raise NeedsHost(k)


def array_elements(type_oid, text):
    """The elements of an array literal as the cell tuples value() takes, through the C++ oracle's parser
    (parse_array_text, oracle_codec.hpp — codec/text.rs:228-312) and its repr. Only for element classes whose repr is unambiguous
    (no strings)."""
    import datetime as dt
    from oracle import oracle
    r = oracle.parse_text_cell(type_oid, text)
    if not r.startswith("Array["):
        raise NeedsHost(r)
    out = []
    body = r[6:-1]
    for e in ([] if not body else body.split(",")):
        k, _, v = e.partition("(")
        v = v[:-1]
        if e == "NULL":
            out.append(("Null",))
        elif k == "Bool":
            out.append(("Bool", v == "true"))
        elif k in ("I16", "I32", "I64", "U32"):
            out.append((k, int(v)))
        elif k in ("F32", "F64"):
            if v == "NaN":
                raise NeedsHost("NaN bits are not in the repr")
            out.append((k, int(v, 16)))
        elif k == "Date":
            out.append(("Date", dt.date.fromisoformat(v).toordinal()))
        elif k == "Time":
            h, m, rest = v.split(":")
            sec, ns = rest.split(".")
            out.append(("Time", int(h) * 3600 + int(m) * 60 + int(sec), int(ns)))
        elif k in ("Timestamp", "TimestampTz"):
            d, t = v.split(" ")
            h, m, rest = t.split(":")
            sec, ns = rest.split(".")
            out.append((k, dt.date.fromisoformat(d).toordinal(), int(h) * 3600 + int(m) * 60 + int(sec), int(ns)))
        elif k == "Uuid":
            out.append(("Uuid", bytes.fromhex(v)))
        else:
            raise NeedsHost(e)
    return out


def array(type_oid, text):
    """rb_encode_value(ClickHouseValue::Array) (encoding.rs:249-254): varint count, every element through rb_encode_nullable."""
    items = array_elements(type_oid, text)
    return varint(len(items)) + b"".join(nullable(c) for c in items)


def nullable(cell):
    return b"\x01" if cell[0] == "Null" else b"\x00" + value(cell)


def row(cells, nullable_flags):
    if len(cells) != len(nullable_flags):
        raise ConversionError("ClickHouse RowBinary row width mismatch")
    out = bytearray()
    for c, nf in zip(cells, nullable_flags):
        if nf:
            out += nullable(c)
        elif c[0] == "Null":
            raise ConversionError("NULL value for non-nullable ClickHouse column")
        else:
            out += value(c)
    return bytes(out)


def cdc_columns(op, commit_lsn, tx_ordinal, engine):
    if engine == MERGE_TREE:
        return string({"I": b"INSERT", "U": b"UPDATE", "D": b"DELETE"}[op]) + struct.pack("<Q", commit_lsn)
    return struct.pack("<QQ", tx_ordinal, commit_lsn) + bytes([1 if op == "D" else 0])   # u128 LE: low half first


def encode_events(events, slot_index, types_by_col, nullable_flags, engine, identity_type="PrimaryKey"):
    """Rows of the events the device emitter takes (Insert; non-partial Update -> new row; Delete with a full old row).
    `types_by_col`: type classes (only their count is used here: text-form classes raise NeedsHost from value()).
    `identity_type`: ReplicatedTableSchema::identity_type of the slot; under ReplacingMergeTree the reference refuses Update
    events unless it is PrimaryKey or Full (clickhouse_update_row, clickhouse/core.rs:1359-1382) — the emitter leaves those
    Updates to the host, which raises that error.
    Returns (list of row bytes, list of event indices, n events of the slot left to the host)."""
    updates_ok = engine == MERGE_TREE or identity_type in ("PrimaryKey", "Full")
    n_user = len(types_by_col)
    rows, idx, host = [], [], 0
    # the sink converts every cell of every pending row first (cell_to_clickhouse_value, core.rs:1193-1203: Date32 range errors)
    # and encodes afterwards (NULL in a non-nullable column, client.rs): a date out of range anywhere comes first
    for e in events:
        if e["kind"] not in "IUD" or e.get("schema_slot") != slot_index:
            continue
        if e["kind"] == "U" and (e["partial"] or not updates_ok) or e["kind"] == "D" and e["old_kind"] != "Full":
            continue
        for c in (e["old_row"] if e["kind"] == "D" else e["row"]):
            try:
                if c[0] == "Date":
                    value(c)
                elif c[0] == "Deferred" and c[1] in ARRAY_OIDS:
                    for el in array_elements(c[1], c[2]):
                        if el[0] == "Date":
                            value(el)
            except NeedsHost:
                pass
    for i, e in enumerate(events):
        if e["kind"] not in "IUD" or e.get("schema_slot") != slot_index:
            continue
        if e["kind"] == "I":
            cells = e["row"]
        elif e["kind"] == "U":
            if e["partial"] or not updates_ok:
                host += 1
                continue
            cells = e["row"]
        else:
            if e["old_kind"] != "Full":
                host += 1
                continue
            cells = e["old_row"]
        assert len(cells) == n_user
        body = row(cells, nullable_flags[:n_user])
        tail = cdc_columns(e["kind"], e["commit_lsn"], e["tx_ordinal"], engine)
        # the trailing CDC columns are never NULL; a Nullable() destination column still takes its marker byte
        if engine == MERGE_TREE:
            parts = [tail[:len(tail) - 8], tail[-8:]]
        else:
            parts = [tail[:16], tail[16:]]
        for p, nf in zip(parts, nullable_flags[n_user:]):
            body += (b"\x00" if nf else b"") + p
        rows.append(body)
        idx.append(i)
    return rows, idx, host
