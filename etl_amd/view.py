"""Host-side view of a decoded batch: numpy copies of the canonical arena of
include/etlg.h plus a materialiser that rebuilds the reference's value model.

`materialize()` is the Python analog of the Rust shim a maintainer would add
(INTEGRATION.md): it walks the arena and yields what
`Destination::write_events` receives — Event{kind, lsns, ordinal, rows of
Cells} (crates/etl/src/event.rs:21-267, crates/etl/src/data/cell.rs:19-57).
"""
import ctypes as C
import struct
from dataclasses import dataclass, field

import numpy as np

from . import abi


def _np_from(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def numeric_to_string(kind, sign, weight, scale, digits):
    """`PgNumeric`'s Display (crates/etl-postgres/src/numeric.rs:460-560) for a materialised ("Numeric", ...) cell: what every
    sink of the reference writes for it (`n.to_string()`). The integer part is groups 0..=weight (base 10000, the first one
    without leading zeros), the fraction the next ceil(scale / 4) groups cut to `scale` digits; missing groups are zeros."""
    if kind != 0:
        return ("NaN", "Infinity", "-Infinity")[kind - 1]
    if not digits:
        return "0" + ("." + "0" * scale if scale else "")
    nd = len(digits)
    whole = "0" if weight < 0 else str(digits[0]) + "".join("%04d" % (digits[d] if d < nd else 0) for d in range(1, weight + 1))
    out = ("-" if sign else "") + whole
    if scale:
        first = ((weight + 1 + 0x8000) & 0xFFFF) - 0x8000   # the reference computes `weight + 1` on an i16
        groups = -(-scale // 4)
        frac = "".join("%04d" % (digits[d] if 0 <= d < nd else 0) for d in range(first, first + groups))
        out += "." + frac[:scale]
    return out


def timetz_to_string(secs, nanos, offset):
    """`PgTimeTz`'s Display (crates/etl-postgres/src/time.rs:113-117, write_utc_offset :210-225): chrono's %H:%M:%S%.f (no
    fraction, or 3 / 6 / 9 digits; a leap second is second 60) + the offset as +HH, +HH:MM or +HH:MM:SS."""
    leap = nanos >= 1_000_000_000
    ns = nanos - 1_000_000_000 if leap else nanos
    t = "%02d:%02d:%02d" % (secs // 3600, secs // 60 % 60, secs % 60 + (1 if leap else 0))
    if ns:
        t += ".%03d" % (ns // 1_000_000) if ns % 1_000_000 == 0 else ".%06d" % (ns // 1000) if ns % 1000 == 0 else ".%09d" % ns
    a = abs(offset)
    t += ("-" if offset < 0 else "+") + "%02d" % (a // 3600)
    if a % 60:
        t += ":%02d:%02d" % (a % 3600 // 60, a % 60)
    elif a % 3600:
        t += ":%02d" % (a % 3600 // 60)
    return t

@dataclass
class SlotCol:
    type_oid: int
    stored_index: int
    type_class: int
    nullable: int
    identity: int
    off_full: int
    off_key: int
    key_index: int


@dataclass
class Slot:
    table_id: int
    n_stored: int
    snapshot_lsn: int
    n_ident: int
    row_bytes_full: int
    row_bytes_key: int
    state_bytes_full: int
    state_bytes_key: int
    cols: list = field(default_factory=list)

    def key_cols(self):
        return [c for c in self.cols if c.identity]


def slots_from_ptr(ptr, n):
    out = []
    for k in range(n):
        d = ptr[k]
        s = Slot(d.table_id, d.n_stored, d.snapshot_lsn, d.n_ident, d.row_bytes_full,
                 d.row_bytes_key, d.state_bytes_full, d.state_bytes_key)
        for i in range(d.n_cols):
            c = d.cols[i]
            s.cols.append(SlotCol(c.type_oid, c.stored_index, c.type_class, c.nullable,
                                  c.identity, c.off_full, c.off_key, c.key_index))
        out.append(s)
    return out


@dataclass
class HostBatch:
    n_events: int
    n_frames: int
    payload_bytes: tuple
    kind: np.ndarray
    flags: np.ndarray
    table_id: np.ndarray
    schema_slot: np.ndarray
    start_lsn: np.ndarray
    commit_lsn: np.ndarray
    tx_ordinal: np.ndarray
    body_off: np.ndarray
    fixed: np.ndarray
    heap: np.ndarray
    slots: list

    ARRAYS = ("kind", "flags", "table_id", "schema_slot", "start_lsn", "commit_lsn",
              "tx_ordinal", "body_off", "fixed", "heap")

    @classmethod
    def from_view(cls, v):
        """v: abi.BatchView with HOST pointers."""
        assert not v.on_device, "copy device views with etl_amd.native first"
        n = v.n_events
        return cls(
            n_events=n, n_frames=v.n_frames, payload_bytes=tuple(v.payload_bytes),
            kind=_np_from(v.ev_kind, n, np.uint8), flags=_np_from(v.ev_flags, n, np.uint8),
            table_id=_np_from(v.ev_table_id, n, np.uint32),
            schema_slot=_np_from(v.ev_schema_slot, n, np.uint32),
            start_lsn=_np_from(v.ev_start_lsn, n, np.uint64),
            commit_lsn=_np_from(v.ev_commit_lsn, n, np.uint64),
            tx_ordinal=_np_from(v.ev_tx_ordinal, n, np.uint64),
            body_off=_np_from(v.ev_body_off, n, np.uint64),
            fixed=_np_from(v.fixed, v.fixed_bytes, np.uint8),
            heap=_np_from(v.heap, v.heap_bytes, np.uint8),
            slots=slots_from_ptr(v.slots, v.n_slots))

    # ---------------------------------------------------------------- parity
    def diff(self, other):
        """Byte-for-byte comparison; returns a list of human-readable differences."""
        out = []
        if self.n_events != other.n_events:
            out.append(f"n_events {self.n_events} != {other.n_events}")
        if self.n_frames != other.n_frames:
            out.append(f"n_frames {self.n_frames} != {other.n_frames}")
        if tuple(self.payload_bytes) != tuple(other.payload_bytes):
            out.append(f"payload_bytes {self.payload_bytes} != {other.payload_bytes}")
        for name in self.ARRAYS:
            a, b = getattr(self, name), getattr(other, name)
            if a.shape != b.shape:
                out.append(f"{name}: shape {a.shape} != {b.shape}")
                continue
            if not np.array_equal(a, b):
                idx = int(np.flatnonzero(a != b)[0])
                out.append(f"{name}: first difference at [{idx}]: {a[idx]} != {b[idx]}")
        if len(self.slots) != len(other.slots):
            out.append(f"slots: {len(self.slots)} != {len(other.slots)}")
        else:
            for k, (a, b) in enumerate(zip(self.slots, other.slots)):
                if a != b:
                    out.append(f"slot {k}: {a} != {b}")
        return out

    # ------------------------------------------------------ shard reassembly
    _HEAP_CLASSES = (abi.TC_STRING, abi.TC_BYTEA, abi.TC_NUMERIC, abi.TC_JSON, abi.TC_ARRAY)

    def _rebase_heap_refs(self, delta):
        """Adds `delta` to every heap reference held in this batch's fixed arena (String / Bytes / Numeric / Json / Array
        cells and every DEFERRED cell keep {u32 heap_off, u32 len} in their slot)."""
        if not delta or self.n_events == 0:
            return
        fx32 = self.fixed.view(np.uint32) if len(self.fixed) % 4 == 0 else None
        assert fx32 is not None
        body = self.body_off.astype(np.int64)
        for si, slot in enumerate(self.slots):
            for kind in (ord("I"), ord("U"), ord("D")):
                base_sel = (self.kind == kind) & (self.schema_slot == si)
                if not base_sel.any():
                    continue
                for ok in (abi.OLD_NONE, abi.OLD_FULL, abi.OLD_KEY):
                    sel = base_sel if kind == ord("I") else base_sel & ((self.flags & 3) == ok)
                    if kind == ord("I") and ok != abi.OLD_NONE:
                        continue
                    ev = np.flatnonzero(sel)
                    if len(ev) == 0:
                        continue
                    images = []   # (row base offsets, key layout?)
                    old_sz = slot.row_bytes_full if ok == abi.OLD_FULL else slot.row_bytes_key if ok == abi.OLD_KEY else 0
                    if kind != ord("I") and ok != abi.OLD_NONE:
                        images.append((body[ev], ok == abi.OLD_KEY))
                    if kind in (ord("I"), ord("U")):
                        images.append((body[ev] + old_sz, False))
                    for bases, key_layout in images:
                        cols = slot.key_cols() if key_layout else slot.cols
                        for i, col in enumerate(cols):
                            st = (self.fixed[bases + i // 4] >> (2 * (i % 4))) & 3
                            is_ref = (st == abi.CELL_DEFERRED) | ((st == abi.CELL_VALUE) & (col.type_class in self._HEAP_CLASSES))
                            at = (bases[is_ref] + (col.off_key if key_layout else col.off_full)) // 4
                            fx32[at] += np.uint32(delta)

    @classmethod
    def concat(cls, parts):
        """Reassembles the batches of consecutive, commit-aligned shards of one stream (rank order == LSN order) into the
        batch a single decode of the whole stream produces: arrays concatenated, body offsets and heap references moved
        by each shard's exclusive prefix (etl_amd/shard.py: global_layout). Every part must come from a context that had
        seen the same schema slots (shard.replay_control gives later shards the earlier shards' Relation / DDL frames)."""
        parts = list(parts)
        fx_off = np.concatenate([[0], np.cumsum([len(p.fixed) for p in parts])])
        hp_off = np.concatenate([[0], np.cumsum([len(p.heap) for p in parts])])
        moved = []
        for k, p in enumerate(parts):
            q = cls(p.n_events, p.n_frames, p.payload_bytes, p.kind, p.flags, p.table_id, p.schema_slot, p.start_lsn, p.commit_lsn,
                    p.tx_ordinal, p.body_off + np.uint64(fx_off[k]), p.fixed.copy(), p.heap, p.slots)
            hold = cls(p.n_events, p.n_frames, p.payload_bytes, p.kind, p.flags, p.table_id, p.schema_slot, p.start_lsn,
                       p.commit_lsn, p.tx_ordinal, p.body_off, q.fixed, p.heap, p.slots)   # shard-local offsets, shared fixed copy
            hold._rebase_heap_refs(int(hp_off[k]))
            moved.append(q)
        cat = lambda name: np.concatenate([getattr(m, name) for m in moved]) if moved else np.zeros(0)
        return cls(
            n_events=sum(p.n_events for p in parts), n_frames=sum(p.n_frames for p in parts),
            payload_bytes=tuple(sum(p.payload_bytes[i] for p in parts) for i in range(3)),
            kind=cat("kind"), flags=cat("flags"), table_id=cat("table_id"), schema_slot=cat("schema_slot"),
            start_lsn=cat("start_lsn"), commit_lsn=cat("commit_lsn"), tx_ordinal=cat("tx_ordinal"), body_off=cat("body_off"),
            fixed=cat("fixed"), heap=cat("heap"), slots=max((p.slots for p in parts), key=len))

    # ----------------------------------------------------------- materialise
    def _heap(self, off, ln):
        return self.heap[off:off + ln].tobytes()

    def _cell(self, col, state, slot_bytes):
        tc = col.type_class
        if state == abi.CELL_NULL:
            return ("Null",)
        if state == abi.CELL_MISSING:
            return ("Missing",)
        if state == abi.CELL_DEFERRED:
            off, ln = struct.unpack_from("<II", slot_bytes, 0)
            return ("Deferred", col.type_oid, self._heap(off, ln))
        if tc == abi.TC_BOOL:
            return ("Bool", bool(struct.unpack_from("<I", slot_bytes)[0]))
        if tc == abi.TC_I16:
            return ("I16", struct.unpack_from("<i", slot_bytes)[0])
        if tc == abi.TC_I32:
            return ("I32", struct.unpack_from("<i", slot_bytes)[0])
        if tc == abi.TC_U32:
            return ("U32", struct.unpack_from("<I", slot_bytes)[0])
        if tc == abi.TC_I64:
            return ("I64", struct.unpack_from("<q", slot_bytes)[0])
        if tc == abi.TC_F32:
            return ("F32", struct.unpack_from("<I", slot_bytes)[0])
        if tc == abi.TC_F64:
            return ("F64", struct.unpack_from("<Q", slot_bytes)[0])
        if tc == abi.TC_DATE:
            return ("Date", struct.unpack_from("<i", slot_bytes)[0])
        if tc == abi.TC_TIME:
            return ("Time",) + struct.unpack_from("<II", slot_bytes)
        if tc == abi.TC_TIMESTAMP:
            return ("Timestamp",) + struct.unpack_from("<iII", slot_bytes)
        if tc == abi.TC_TIMESTAMPTZ:
            return ("TimestampTz",) + struct.unpack_from("<iII", slot_bytes)
        if tc == abi.TC_TIMETZ:
            return ("TimeTz",) + struct.unpack_from("<IIi", slot_bytes)
        if tc == abi.TC_UUID:
            return ("Uuid", bytes(slot_bytes[:16]))
        off, ln = struct.unpack_from("<II", slot_bytes, 0)
        raw = self._heap(off, ln)
        if tc == abi.TC_ARRAY:
            return self._array(raw)
        if tc == abi.TC_NUMERIC:
            kind, sign, weight, scale, nd = struct.unpack_from("<BBhHH", raw, 0)
            digits = struct.unpack_from(f"<{(ln - 8) // 2}h", raw, 8)
            return ("Numeric", kind, sign, weight, scale, digits)
        if tc == abi.TC_BYTEA:
            return ("Bytes", raw)
        return ("String", raw)

    _ELEM_NAMES = {abi.TC_BOOL: "Bool", abi.TC_I16: "I16", abi.TC_I32: "I32", abi.TC_U32: "U32", abi.TC_I64: "I64", abi.TC_F32: "F32", abi.TC_F64: "F64",
                   abi.TC_DATE: "Date", abi.TC_TIME: "Time", abi.TC_TIMESTAMP: "Timestamp", abi.TC_TIMESTAMPTZ: "TimestampTz", abi.TC_TIMETZ: "TimeTz",
                   abi.TC_UUID: "Uuid", abi.TC_NUMERIC: "Numeric", abi.TC_BYTEA: "Bytes", abi.TC_STRING: "String"}

    def _array(self, raw):
        """A typed array entry (etlg_array_hdr, include/etlg.h) -> ("Array", element class name, [element cells | ("Null",)])."""
        n, ecls, ebytes, _ = struct.unpack_from("<IBBH", raw, 0)
        vw = (n + 31) // 32
        valid = struct.unpack_from(f"<{vw}I", raw, 8) if vw else ()
        at = 8 + 4 * vw
        col = type("ElemCol", (), {"type_class": ecls, "type_oid": 0})
        out = []
        if ebytes:
            for k in range(n):
                if not (valid[k >> 5] >> (k & 31)) & 1:
                    out.append(("Null",))
                else:
                    out.append(self._cell(col, abi.CELL_VALUE, raw[at + k * ebytes:at + k * ebytes + 16].ljust(16, b"\0")))
        else:
            ends = struct.unpack_from(f"<{n}I", raw, at) if n else ()
            data = raw[at + 4 * n:]
            prev = 0
            for k in range(n):
                piece = data[prev:ends[k]]
                prev = ends[k]
                if not (valid[k >> 5] >> (k & 31)) & 1:
                    out.append(("Null",))
                elif ecls == abi.TC_NUMERIC:
                    kind, sign, weight, scale, nd = struct.unpack_from("<BBhHH", piece, 0)
                    out.append(("Numeric", kind, sign, weight, scale, struct.unpack_from(f"<{nd}h", piece, 8)))
                else:
                    out.append(("Bytes" if ecls == abi.TC_BYTEA else "String", piece))
        return ("Array", self._ELEM_NAMES.get(ecls, str(ecls)), out)

    def _row(self, slot, base, key_layout):
        cols = slot.key_cols() if key_layout else slot.cols
        fx = self.fixed
        cells = []
        for i, col in enumerate(cols):
            st = (int(fx[base + i // 4]) >> (2 * (i % 4))) & 3
            so = base + (col.off_key if key_layout else col.off_full)
            cells.append(self._cell(col, st, fx[so:so + 16].tobytes()))
        return cells

    def materialize(self):
        """List of event dicts mirroring the reference's Event enum."""
        out = []
        fx = self.fixed
        for i in range(self.n_events):
            k = chr(self.kind[i])
            e = {"kind": k, "start_lsn": int(self.start_lsn[i]), "commit_lsn": int(self.commit_lsn[i]),
                 "tx_ordinal": int(self.tx_ordinal[i])}
            base = int(self.body_off[i])
            fl = int(self.flags[i])
            if k == "B":
                e["xid"] = int(self.table_id[i])
                e["timestamp"] = struct.unpack_from("<q", fx, base)[0]
            elif k == "C":
                e["flags"] = fl - 256 if fl > 127 else fl
                e["end_lsn"], e["timestamp"] = struct.unpack_from("<Qq", fx, base)
            elif k == "R":
                e["table_id"] = int(self.table_id[i])
                e["schema_slot"] = int(self.schema_slot[i])
            elif k == "T":
                e["options"] = fl - 256 if fl > 127 else fl
                n = int(self.table_id[i])
                e["tables"] = [struct.unpack_from("<II", fx, base + 8 * j) for j in range(n)]
            else:
                e["table_id"] = int(self.table_id[i])
                e["schema_slot"] = int(self.schema_slot[i])
                slot = self.slots[e["schema_slot"]]
                if k == "I":
                    e["row"] = self._row(slot, base, False)
                else:
                    ok = fl & 3
                    e["old_kind"] = ("None", "Full", "Key")[ok]
                    old_sz = slot.row_bytes_full if ok == abi.OLD_FULL else slot.row_bytes_key if ok == abi.OLD_KEY else 0
                    if ok:
                        e["old_row"] = self._row(slot, base, ok == abi.OLD_KEY)
                    if k == "U":
                        e["partial"] = bool(fl & abi.FLAG_PARTIAL)
                        e["row"] = self._row(slot, base + old_sz, False)
            out.append(e)
        return out
