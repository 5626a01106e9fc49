"""Test-side encoder for pgoutput (proto v1, text tuples) messages wrapped in
CopyData frames — the Python twin of the reference's own test encoders
(crates/etl/src/postgres/codec/event.rs:1076-1172 `encode_tuple`,
`parse_{insert,update,delete}_body_from_parts`) extended with the XLogData /
Begin / Commit / Relation / Truncate / Message / Type / Origin layouts of the
PostgreSQL protocol documentation.
"""
import struct

NULL = object()   # 'n'
TOAST = object()  # 'u'


class Binary(bytes):
    """A 'b' (binary format) tuple cell."""


def tuple_data(cells):
    out = [struct.pack(">h", len(cells))]
    for c in cells:
        if c is NULL:
            out.append(b"n")
        elif c is TOAST:
            out.append(b"u")
        elif isinstance(c, Binary):
            out.append(b"b" + struct.pack(">i", len(c)) + bytes(c))
        else:
            b = c.encode() if isinstance(c, str) else bytes(c)
            out.append(b"t" + struct.pack(">i", len(b)) + b)
    return b"".join(out)


def cstr(s):
    return (s.encode() if isinstance(s, str) else s) + b"\0"


def begin(final_lsn, ts=0, xid=1):
    return b"B" + struct.pack(">QqI", final_lsn, ts, xid)


def commit(commit_lsn, end_lsn, ts=0, flags=0):
    return b"C" + struct.pack(">bQQq", flags, commit_lsn, end_lsn, ts)


def origin(lsn, name):
    return b"O" + struct.pack(">Q", lsn) + cstr(name)


def type_msg(oid, nsp, name):
    return b"Y" + struct.pack(">I", oid) + cstr(nsp) + cstr(name)


def relation(rel_id, nsp, name, replident, cols):
    """cols: list of (flags, name, type_oid, typmod)."""
    out = b"R" + struct.pack(">I", rel_id) + cstr(nsp) + cstr(name) + replident.encode() + struct.pack(">h", len(cols))
    for flags, cname, oid, typmod in cols:
        out += struct.pack(">b", flags) + cstr(cname) + struct.pack(">Ii", oid, typmod)
    return out


def insert(rel_id, new):
    return b"I" + struct.pack(">I", rel_id) + b"N" + tuple_data(new)


def update(rel_id, new, old=None, key=None):
    out = b"U" + struct.pack(">I", rel_id)
    if old is not None:
        out += b"O" + tuple_data(old)
    elif key is not None:
        out += b"K" + tuple_data(key)
    return out + b"N" + tuple_data(new)


def delete(rel_id, old=None, key=None):
    out = b"D" + struct.pack(">I", rel_id)
    if old is not None:
        return out + b"O" + tuple_data(old)
    return out + b"K" + tuple_data(key)


def truncate(rel_ids, options=0):
    return b"T" + struct.pack(">ib", len(rel_ids), options) + b"".join(struct.pack(">I", r) for r in rel_ids)


def message(prefix, content, lsn=0, transactional=True):
    c = content.encode() if isinstance(content, str) else content
    return b"M" + struct.pack(">bQ", 1 if transactional else 0, lsn) + cstr(prefix) + struct.pack(">i", len(c)) + c


def xlog(wal_start, msg, wal_end=None, ts=0):
    return b"w" + struct.pack(">QQq", wal_start, wal_start if wal_end is None else wal_end, ts) + msg


def keepalive(wal_end, ts=0, reply=0):
    return b"k" + struct.pack(">QqB", wal_end, ts, reply)


def frame(payload):
    return b"d" + struct.pack(">I", len(payload) + 4) + payload


class Stream:
    """Accumulates CopyData frames + the offsets sidecar."""

    def __init__(self, lsn=0x1000):
        self.buf = bytearray()
        self.offsets = [0]
        self.lsn = lsn

    def add_payload(self, payload):
        self.buf += frame(payload)
        self.offsets.append(len(self.buf))
        return self

    def add(self, msg, lsn=None):
        if lsn is None:
            self.lsn += 8
            lsn = self.lsn
        return self.add_payload(xlog(lsn, msg))

    def bytes(self):
        return bytes(self.buf)
