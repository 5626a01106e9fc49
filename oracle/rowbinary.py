"""TEST INFRASTRUCTURE — CPU restatement of the reference's ClickHouse RowBinary encoder, for the parity tests of
etlg_batch_rowbinary (etl_amd/csrc/columns.hip). Never imported by the product path.

Follows crates/etl-destinations/src/clickhouse/encoding.rs:
  cell_to_clickhouse_value :58-83   (which wire type every Cell becomes; Date range check :147-161; bytes_to_hex :176-185)
  rb_varint :188-199 | rb_encode_nullable :202-211 | rb_encode_value :214-255 | encode_to_row_binary :259-283
and crates/etl-destinations/src/clickhouse/core.rs:
  which events become rows :1078-1127 | append_cdc_columns :96-114 (MergeTree: cdc_operation String + cdc_lsn UInt64;
  ReplacingMergeTree: _etl_version UInt128 = commit_lsn << 64 | tx_ordinal, _etl_deleted UInt8) |
  expand_key_row :1437-1472 + default_cell :1481-1517 (the tombstone row of a Delete that carries only the key; default_cell pinned
  by the reference's default_cell_string_mapped_values_are_strings :1930-1937 in tests/test_oracle_rowbinary.py).
Pinned by tests/test_oracle_rowbinary.py to the byte vectors of the reference's own tests (encoding.rs:386-470).

`Time`, `TimeTz` and `Numeric` cells are strings in the reference (`to_string()`, encoding.rs:66-71): their Display impls are
restated in oracle/display.py (chrono's NaiveTime Display is a dependency that is not vendored under /root/reference).

Works on the per-cell tuples of etl_amd.view.HostBatch.materialize()."""
import struct

from oracle.display import numeric_string, time_string, timetz_string   # noqa: F401  (time_string is re-exported)

MERGE_TREE, REPLACING_MERGE_TREE = 0, 1
# array types whose element class the device encodes: bool int2 int4 int8 oid float4 float8 date time timestamp timestamptz uuid
ARRAY_OIDS = {1000, 1005, 1007, 1016, 1028, 1021, 1022, 1182, 1183, 1115, 1185, 2951}
JSON_OIDS = {114, 3802}
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
    if k == "Deferred" and cell[1] in JSON_OIDS:    # json cells keep their text in the arena: String(j.to_string()) (encoding.rs:73)
        from oracle import json_display
        if not json_display.device_limits_ok(cell[2]):
            raise NeedsHost("json beyond json_display's limits")
        return string(json_display.display(cell[2]))
    if k == "Deferred" and cell[1] in ARRAY_OIDS:   # array columns keep their literal in the arena
        return array(cell[1], cell[2])
    if k == "Deferred":                              # text-like / numeric / timetz / bytea elements: String(..) each (encoding.rs:89-111; bytea: String(bytes_to_hex))
        from oracle import arrays
        if cell[1] in arrays.VAR_ARRAY_OIDS or (cell[1] not in JSON_OIDS and arrays.is_string_array(cell[1], cell[2])):
            items = arrays.elements(cell[1], cell[2])
            hexed = cell[1] == arrays.BYTEA_A
            return varint(len(items)) + b"".join(b"\x01" if e is None else b"\x00" + string(e.hex().encode() if hexed else e) for e, _ in items)
    if k == "EmptyArray":                            # default_cell of an array column: varint count 0 (encoding.rs:249-254)
        return varint(0)
    raise NeedsHost(k)


def _ce_days(datestr):
    """days from 0001-01-01 = 1 (chrono's num_days_from_ce) of the oracle's `[-]Y-MM-DD` repr, years before 1 and beyond 9999 included
    (datetime.date covers 1..9999 only)."""
    neg = datestr.startswith("-")
    y, m, d = (datestr[1:] if neg or datestr.startswith("+") else datestr).split("-")
    y = -int(y) if neg else int(y)
    m, d = int(m), int(d)
    y -= m <= 2
    era = (y if y >= 0 else y - 399) // 400
    yoe = y - era * 400
    doy = (153 * (m + (-3 if m > 2 else 9)) + 2) // 5 + d - 1
    doe = yoe * 365 + yoe // 4 - yoe // 100 + doy
    return era * 146097 + doe - 719468 + CE_DAYS_1970   # days since 1970-01-01 -> from CE


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
            out.append(("Date", _ce_days(v)))
        elif k == "Time":
            h, m, rest = v.split(":")
            sec, ns = rest.split(".")
            out.append(("Time", int(h) * 3600 + int(m) * 60 + int(sec), int(ns)))
        elif k in ("Timestamp", "TimestampTz"):
            d, t = v.split(" ")
            h, m, rest = t.split(":")
            sec, ns = rest.split(".")
            out.append((k, _ce_days(d), int(h) * 3600 + int(m) * 60 + int(sec), int(ns)))
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


# type OIDs default_cell names (clickhouse/core.rs:1481-1517)
_OID = {"BOOL": 16, "INT2": 21, "INT4": 23, "INT8": 20, "OID": 26, "FLOAT4": 700, "FLOAT8": 701, "DATE": 1082, "TIMESTAMP": 1114,
        "TIMESTAMPTZ": 1184, "UUID": 2950}
_EPOCH = CE_DAYS_1970


TC_STRING, TC_ARRAY = 0, 17                      # etlg_type_class values (include/etlg.h)
TEXT_OIDS = {25, 1043, 1042, 19, 18}             # text varchar bpchar name "char": the scalar types whose cells are Strings
# array types the value codec has no arm for (their cells decode as Strings) but is_array_type still calls arrays; the names the
# reference's own tests use (type_utils.rs:26-60, clickhouse/core.rs:1930-1937): _char _name _bpchar _money _interval ...
EXTRA_ARRAY_OIDS = {1002, 1003, 1014, 791, 1187, 1561, 1563, 1040, 1041, 651, 775}


def class_of(oid):
    from oracle import oracle
    return oracle.lib().oracle_class_of_oid(oid)


def is_array_oid(oid):
    """is_array_type (crates/etl-postgres/src/type_utils.rs:14-18): array kind and an underscore-prefixed name."""
    return class_of(oid) == TC_ARRAY or oid in EXTRA_ARRAY_OIDS


def key_rows_on_device(schema_cols):
    """The device's contract for key-only Deletes (include/etlg.h, etlg_batch_rowbinary): it builds the tombstone unless a nullable
    non-key column has a type outside the value codec's table (a String cell that may or may not be an array type to
    is_array_type: NULL or an empty array) — those slots' key-only Deletes stay with the host."""
    for _name, oid, nullable, pk in schema_cols:
        if not pk and nullable and class_of(oid) == TC_STRING and oid not in TEXT_OIDS:
            return False
    return True


def default_cell(oid):
    """default_cell (clickhouse/core.rs:1481-1517): the zero value of a non-key column in a key-only DELETE tombstone. Array types ->
    an empty array; date / timestamp / uuid -> typed zeros; every other non-primitive type -> an empty String."""
    if oid == _OID["BOOL"]:
        return ("Bool", False)
    if oid == _OID["INT2"]:
        return ("I16", 0)
    if oid == _OID["INT4"]:
        return ("I32", 0)
    if oid == _OID["INT8"]:
        return ("I64", 0)
    if oid == _OID["OID"]:
        return ("U32", 0)
    if oid == _OID["FLOAT4"]:
        return ("F32", 0)
    if oid == _OID["FLOAT8"]:
        return ("F64", 0)
    if oid == _OID["DATE"]:
        return ("Date", _EPOCH)
    if oid in (_OID["TIMESTAMP"], _OID["TIMESTAMPTZ"]):
        return ("Timestamp", _EPOCH, 0, 0)
    if oid == _OID["UUID"]:
        return ("Uuid", bytes(16))
    if is_array_oid(oid):
        return ("EmptyArray",)
    return ("String", b"")


def expand_key_row(key_cells, schema_cols, identity_type):
    """expand_key_row (clickhouse/core.rs:1437-1472). schema_cols: (name, type oid, nullable, primary key) of the slot's replicated
    columns. Returns the full-width cells, or raises HostRow when the reference raises (the host reports it)."""
    n_pk = sum(1 for c in schema_cols if c[3])
    if len(key_cells) != n_pk:
        raise HostRow("ClickHouse key image does not match the source primary key")
    if identity_type not in ("PrimaryKey", "Full"):
        raise HostRow("ClickHouse requires primary-key or full replica identity")
    it = iter(key_cells)
    out = []
    for _name, oid, nullable, pk in schema_cols:
        if pk:
            out.append(next(it, ("Null",)))
        elif nullable and not is_array_oid(oid):
            out.append(("Null",))
        else:
            out.append(default_cell(oid))
    return out


class HostRow(Exception):
    """The reference raises for this row (a replica-identity error): the device leaves it to the host."""


def encode_events(events, slot_index, types_by_col, nullable_flags, engine, identity_type="PrimaryKey", schema_cols=None):
    """Rows of the events the device emitter takes (Insert; non-partial Update -> new row; Delete with a full old row; with
    `schema_cols` also a Delete that carries only the key, as the tombstone expand_key_row builds).
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
            continue   # (a key row holds identity columns only; its dates are range-checked below with the row)
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
            if e["old_kind"] == "Key" and schema_cols is not None and key_rows_on_device(schema_cols):
                try:
                    cells = expand_key_row(e["old_row"], schema_cols, identity_type)
                except HostRow:
                    host += 1
                    continue
            elif e["old_kind"] != "Full":
                host += 1
                continue
            else:
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
