"""TEST INFRASTRUCTURE — CPU restatement of the reference's BigQuery row encoder, for the parity test of etlg_batch_protobuf
(etl_amd/csrc/columns.hip). Never imported by the product path.

Follows crates/etl-destinations/src/bigquery/encoding.rs: cell_encode_prost :120-190 (which protobuf field type every Cell
becomes), BigQueryTableRow::try_from :54-66 (tags = position + 1), and bigquery/core.rs:978-996 + 1404-1406 (the Insert row's two
trailing cells: "UPSERT" and the sequence key), crates/etl/src/event.rs:346-351 (EventSequenceKey Display).

PARITY UNPINNED: the reference's tests compare against prost's own output, never against literal bytes, and prost is an
un-vendored dependency — the wire rules below are the protobuf encoding specification (varint keys, wire types 0 / 1 / 2 / 5,
int32 / int64 as sign-extended 64-bit varints), which prost implements. Date / time strings: chrono's %Y-%m-%d, %H:%M:%S%.f
(etl-postgres/src/time.rs:13-21; %.f prints nothing, or 3 / 6 / 9 digits).

Works on the per-cell tuples of etl_amd.view.HostBatch.materialize()."""
import datetime as dt
import struct

from oracle.display import numeric_string, time_string, timetz_string
from oracle.rowbinary import NeedsHost


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
        return key(tag, 0) + varint(((c[1] - 719163) * 86400 + c[2]) * 1_000_000 + c[3] // 1000)
    if k == "TimeTz":                                # t.to_string() (encoding.rs:158-161)
        return ld(tag, timetz_string(*c[1:]).encode())
    if k == "Numeric":                               # validate_cell_for_bigquery, then n.to_string() (encoding.rs:146-149)
        if c[1] == 0 and c[4] > 38:
            raise UnsupportedValueInDestination(f"Cell at index {tag - 1} failed validation")
        return ld(tag, numeric_string(*c[1:]).encode())
    if k == "Uuid":
        h = c[1].hex()
        return ld(tag, f"{h[:8]}-{h[8:12]}-{h[12:16]}-{h[16:20]}-{h[20:]}".encode())
    raise NeedsHost(k)


def insert_rows(events, slot_index):
    """(list of row bytes, event indices, events of the slot left to the host)."""
    rows, idx, host = [], [], 0
    for i, e in enumerate(events):
        if e["kind"] not in "IUD" or e.get("schema_slot") != slot_index:
            continue
        if e["kind"] != "I":
            host += 1
            continue
        body = b"".join(cell(c, t + 1) for t, c in enumerate(e["row"]))
        n = len(e["row"])
        body += ld(n + 1, b"UPSERT") + ld(n + 2, f"{e['commit_lsn']:016x}/{e['tx_ordinal']:016x}/{0:016x}".encode())
        rows.append(body)
        idx.append(i)
    return rows, idx, host
