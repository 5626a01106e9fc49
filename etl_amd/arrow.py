"""Columnar hand-off of decoded rows to Arrow (first step — host side).

The reference turns `TableRow`s into an Arrow `RecordBatch` one `Cell` at a time
(`rows_to_record_batch`, crates/etl-destinations/src/iceberg/encoding.rs:34-58; per-column builders
`build_array_for_field` :61-84 and the `cell_to_*` converters :150-360). The canonical arena of include/etlg.h is
already columnar in everything but address: every row of one schema slot has the same layout, so a column is a
strided gather over the fixed arena (plus one gather over the heap for var-len cells) — no `Cell` objects, no per-row
Python. This module does that with numpy and hands the buffers to pyarrow, following the reference's type mapping:

    Bool -> Boolean | I16, I32 -> Int32 | I64, U32 -> Int64 (cell_to_i32 / cell_to_i64, :157-171)
    F32 -> Float32 | F64 -> Float64 | String -> Utf8 | Bytes -> LargeBinary
    Date -> Date32 (days since 1970-01-01, :194) | Time -> Time64(us) (:201)
    Timestamp -> Timestamp(us) (:208) | TimestampTz -> Timestamp(us, "UTC") (:215) | Uuid -> FixedSizeBinary(16)

    Numeric, TimeTz -> Utf8 of their Display strings (cell_to_string, :349-352: `n.to_string()`, `t.to_string()`)

Columns whose class the device hands back as text (json, arrays, DEFERRED cells) have no fixed-width Arrow form here:
`on_text="binary"` emits their heap entries as LargeBinary for the host to finish, the default raises. Works on any `HostBatch`
(the HIP path's or the oracle's — they are byte-identical).
"""
import numpy as np

from . import abi

_CE_DAYS_1970 = 719163   # chrono num_days_from_ce of 1970-01-01 (the arena stores dates as days from CE)


def _gather(fx, start, width):
    """fx[start[i] : start[i] + width] for every i, as an (n, width) uint8 array."""
    return fx[start[:, None] + np.arange(width, dtype=np.int64)[None, :]]


def _words(fx, start, nwords):
    return np.ascontiguousarray(_gather(fx, start, 4 * nwords)).view(np.uint32).reshape(len(start), nwords)


def _var(hb, fx, start, valid):
    """(offset, length) slots -> Arrow offsets + contiguous data gathered from the heap."""
    w = _words(fx, start, 2)
    off = w[:, 0].astype(np.int64)
    ln = np.where(valid, w[:, 1], 0).astype(np.int64)
    ends = np.cumsum(ln)
    offsets = np.concatenate([[0], ends]).astype(np.int64)
    total = int(ends[-1]) if len(ends) else 0
    src = np.repeat(off - offsets[:-1], ln) + np.arange(total, dtype=np.int64)
    return offsets, hb.heap[src] if total else np.zeros(0, dtype=np.uint8)


def rows_to_record_batch(hb, slot_index, names=None, kinds=("I",), on_text="raise", columns=None):
    """Full-layout rows of `kinds` events ('I' inserts; 'U' adds the new row of non-partial updates) decoded against
    schema slot `slot_index`, as a pyarrow.RecordBatch in event order. `names`: one field name per replicated column
    (default c<stored_index>); `columns`: positions of the replicated columns to hand off (default all)."""
    import pyarrow as pa
    slot = hb.slots[slot_index]
    fx = hb.fixed
    sel = np.zeros(hb.n_events, dtype=bool)
    base_all = hb.body_off.astype(np.int64)
    if "I" in kinds:
        sel |= hb.kind == ord("I")
    if "U" in kinds:
        upd = (hb.kind == ord("U")) & ((hb.flags & abi.FLAG_PARTIAL) == 0)
        sel |= upd
        ok = hb.flags & 3   # the new row follows the old / key image
        old_sz = np.where(ok == abi.OLD_FULL, slot.row_bytes_full, np.where(ok == abi.OLD_KEY, slot.row_bytes_key, 0))
        base_all = base_all + np.where(upd, old_sz, 0)
    sel &= hb.schema_slot == slot_index
    base = base_all[sel]
    n = len(base)
    arrays, fields = [], []
    for i, col in enumerate(slot.cols):
        if columns is not None and i not in columns:
            continue
        name = names[i] if names else f"c{col.stored_index}"
        st = (fx[base + i // 4] >> np.uint8(2 * (i % 4))) & 3 if n else np.zeros(0, dtype=np.uint8)
        if np.any(st == abi.CELL_MISSING):
            raise ValueError(f"column {name}: unchanged-toast cells have no value to hand off")
        valid = st == abi.CELL_VALUE
        deferred = st == abi.CELL_DEFERRED
        mask = ~valid
        so = base + col.off_full
        tc = col.type_class
        if tc in (abi.TC_TIMETZ, abi.TC_NUMERIC) and not np.any(deferred):
            # Display strings in the reference (cell_to_string :349-352); per-row formatting on this host path
            # (etlg_batch_columns formats them on the device)
            arr = pa.array(_display_strings(hb, fx, so, valid, tc), type=pa.string())
            arrays.append(arr)
            fields.append(pa.field(name, arr.type, nullable=bool(col.nullable)))
            continue
        textual = tc not in _FIXED and tc not in (abi.TC_STRING, abi.TC_BYTEA)   # json, arrays, ...
        if np.any(deferred) or textual:
            if on_text != "binary":
                raise NotImplementedError(f"column {name} (type class {tc}): text-form cells (numeric / json / arrays / deferred) "
                                          "have no fixed-width Arrow form; pass on_text='binary' to receive their heap entries")
            offsets, data = _var(hb, fx, so, valid | deferred)
            arr = pa.LargeBinaryArray.from_buffers(pa.large_binary(), n, [_validity(pa, valid | deferred), pa.py_buffer(offsets), pa.py_buffer(data)])
        elif tc == abi.TC_STRING or tc == abi.TC_BYTEA:
            offsets, data = _var(hb, fx, so, valid)
            if tc == abi.TC_STRING:
                if len(data) >= 1 << 31:   # Utf8 has 32-bit offsets: a bigger hand-off goes out as LargeUtf8
                    arr = pa.Array.from_buffers(pa.large_utf8(), n, [_validity(pa, valid), pa.py_buffer(offsets), pa.py_buffer(data)])
                else:
                    arr = pa.StringArray.from_buffers(n, pa.py_buffer(offsets.astype(np.int32)), pa.py_buffer(data), _validity(pa, valid))
            else:
                arr = pa.LargeBinaryArray.from_buffers(pa.large_binary(), n, [_validity(pa, valid), pa.py_buffer(offsets), pa.py_buffer(data)])
        else:
            arr = _FIXED[tc](pa, fx, so, mask)
        arrays.append(arr)
        fields.append(pa.field(name, arr.type, nullable=bool(col.nullable)))
    return pa.RecordBatch.from_arrays(arrays, schema=pa.schema(fields))


def columns_to_record_batch(cols, names=None, columns=None, on_text="raise"):
    """pyarrow.RecordBatch over the buffers `Batch.columns(slot, ...)` built on the device (host-resident `Columns`): the
    buffers are wrapped, not converted — no per-row or per-column arithmetic happens here. Same field types as
    `rows_to_record_batch`, except that String is LargeUtf8 (64-bit offsets: one hand-off can hold more than 2 GiB of text)."""
    import pyarrow as pa
    n = cols.n_rows
    types = {abi.AK_BOOLEAN: pa.bool_(), abi.AK_INT32: pa.int32(), abi.AK_INT64: pa.int64(), abi.AK_FLOAT32: pa.float32(),
             abi.AK_FLOAT64: pa.float64(), abi.AK_DATE32: pa.date32(), abi.AK_TIME64_US: pa.time64("us"),
             abi.AK_TIMESTAMP_US: pa.timestamp("us"), abi.AK_TIMESTAMP_US_UTC: pa.timestamp("us", tz="UTC"),
             abi.AK_FIXED16: pa.binary(16), abi.AK_LARGE_UTF8: pa.large_utf8(), abi.AK_LARGE_BINARY: pa.large_binary(),
             abi.AK_TEXT_FORM: pa.large_binary()}
    arrays, fields = [], []
    for i in range(cols.view.n_cols):
        if columns is not None and i not in columns:
            continue
        k = cols.column(i)
        name = names[i] if names else f"c{i}"
        if k.arrow_kind == abi.AK_NONE:
            raise NotImplementedError(f"column {name}: not handed off (select other columns with columns=)")
        if (k.arrow_kind == abi.AK_TEXT_FORM or (k.deferred_count and k.arrow_kind != abi.AK_LIST)) and on_text != "binary":
            raise NotImplementedError(f"column {name} (type class {k.type_class}): text-form cells (numeric / json / arrays / deferred) "
                                      "have no fixed-width Arrow form; pass on_text='binary' (deferred cells of a fixed-width "
                                      "column come back null, flagged in the column's `deferred` bitmap)")
        validity, _deferred, values, offsets = cols.host_arrays(i)
        vbuf = None if k.null_count == 0 else pa.py_buffer(validity)
        if k.arrow_kind == abi.AK_LIST:   # array literal parsed on the device: LargeList<child>
            ct = types[k.child_kind]
            cv = None if k.child_null_count == 0 else pa.py_buffer(cols.child_validity(i))
            co = cols.child_offsets(i)
            if k.child_kind == abi.AK_LARGE_UTF8 and co is None:   # no element at all
                co = np.zeros(1, dtype=np.int64)
            cbufs = [cv, pa.py_buffer(values)] if co is None else [cv, pa.py_buffer(co), pa.py_buffer(values)]
            child = pa.Array.from_buffers(ct, int(k.child_count), cbufs, null_count=int(k.child_null_count))
            t = pa.large_list(ct)
            arrays.append(pa.Array.from_buffers(t, n, [vbuf, pa.py_buffer(offsets)], null_count=int(k.null_count), children=[child]))
            fields.append(pa.field(name, t, nullable=bool(k.nullable)))
            continue
        t = types[k.arrow_kind]
        bufs = [vbuf, pa.py_buffer(values)] if offsets is None else [vbuf, pa.py_buffer(offsets), pa.py_buffer(values)]
        arr = pa.Array.from_buffers(t, n, bufs, null_count=int(k.null_count))
        arrays.append(arr)
        fields.append(pa.field(name, t, nullable=bool(k.nullable)))
    return pa.RecordBatch.from_arrays(arrays, schema=pa.schema(fields))


def _display_strings(hb, fx, so, valid, tc):
    import struct
    from .view import numeric_to_string, timetz_to_string
    out = []
    for i in range(len(so)):
        if not valid[i]:
            out.append(None)
        elif tc == abi.TC_TIMETZ:
            out.append(timetz_to_string(*struct.unpack_from("<IIi", fx, int(so[i]))))
        else:
            off, ln = struct.unpack_from("<II", fx, int(so[i]))
            raw = hb.heap[off:off + ln].tobytes()
            kind, sign, weight, scale, _nd = struct.unpack_from("<BBhHH", raw, 0)
            out.append(numeric_to_string(kind, sign, weight, scale, struct.unpack_from(f"<{(ln - 8) // 2}h", raw, 8)))
    return out


def _validity(pa, valid):
    return None if valid.all() else pa.py_buffer(np.packbits(valid, bitorder="little"))


def _prim(np_view, pa_type, nwords=1, conv=None):
    def f(pa, fx, so, mask):
        w = _words(fx, so, nwords)
        v = np.ascontiguousarray(w).view(np_view).reshape(len(so)) if conv is None else conv(w)
        return pa.array(v, type=pa_type(pa), mask=mask)
    return f


def _ts_micros(w):  # (days from CE, second of day, nanos) -> microseconds since the epoch (chrono timestamp_micros)
    days = w[:, 0].view(np.int32).astype(np.int64) - _CE_DAYS_1970
    return (days * 86400 + w[:, 1].astype(np.int64)) * 1_000_000 + w[:, 2].astype(np.int64) // 1000


def _uuid(pa, fx, so, mask):
    raw = np.ascontiguousarray(_gather(fx, so, 16))
    arr = pa.FixedSizeBinaryArray.from_buffers(pa.binary(16), len(so), [None if not mask.any() else pa.py_buffer(np.packbits(~mask, bitorder="little")),
                                                                     pa.py_buffer(raw.reshape(-1))])
    return arr


_FIXED = {
    abi.TC_BOOL: _prim(None, lambda pa: pa.bool_(), conv=lambda w: w[:, 0] != 0),
    abi.TC_I16: _prim(np.int32, lambda pa: pa.int32()),
    abi.TC_I32: _prim(np.int32, lambda pa: pa.int32()),
    abi.TC_U32: _prim(None, lambda pa: pa.int64(), conv=lambda w: w[:, 0].astype(np.int64)),
    abi.TC_I64: _prim(np.int64, lambda pa: pa.int64(), nwords=2),
    abi.TC_F32: _prim(np.float32, lambda pa: pa.float32()),
    abi.TC_F64: _prim(np.float64, lambda pa: pa.float64(), nwords=2),
    abi.TC_DATE: _prim(None, lambda pa: pa.date32(), conv=lambda w: (w[:, 0].view(np.int32) - _CE_DAYS_1970).astype(np.int32)),
    abi.TC_TIME: _prim(None, lambda pa: pa.time64("us"), nwords=2,
                       conv=lambda w: w[:, 0].astype(np.int64) * 1_000_000 + w[:, 1].astype(np.int64) // 1000),
    abi.TC_TIMESTAMP: _prim(None, lambda pa: pa.timestamp("us"), nwords=3, conv=_ts_micros),
    abi.TC_TIMESTAMPTZ: _prim(None, lambda pa: pa.timestamp("us", tz="UTC"), nwords=3, conv=_ts_micros),
    abi.TC_UUID: _uuid,
}
