"""Synthetic WAL workloads of SURVEY.md §8(d): ctypes front-end of
libetlg_synth.so (etl_amd/csrc/synth.cpp) plus the per-config table specs.

A workload is (tables, generator config). `register(target)` installs the
stored schemas / table states on anything exposing the etlg control-plane
calls (`schema_put`, `table_state`, `table_ready`) — an etl_amd.Decoder or the
test-only oracle wrapper — so both sides are primed identically.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))

CK_INT4_10D, CK_INT8, CK_INT4, CK_INT2, CK_BOOL, CK_NUMERIC, CK_TEXT, CK_TIMESTAMPTZ, CK_UUID, CK_INT8_SEQ = range(10)


class _Col(C.Structure):
    _fields_ = [("kind", C.c_int32), ("type_oid", C.c_uint32), ("nullable", C.c_uint8), ("pk", C.c_uint8),
                ("null_pct", C.c_uint8), ("utf8_pct", C.c_uint8), ("min_len", C.c_uint32), ("max_len", C.c_uint32),
                ("name", C.c_char * 32)]


class _Table(C.Structure):
    _fields_ = [("rel_id", C.c_uint32), ("ncols", C.c_uint32), ("cols", _Col * 32), ("name", C.c_char * 32)]


class _Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("start_lsn", C.c_uint64), ("ntables", C.c_uint32), ("tables", _Table * 4),
                ("rows_per_txn", C.c_uint32), ("pct_insert", C.c_uint32), ("pct_update", C.c_uint32),
                ("pct_delete", C.c_uint32), ("upd_pct_key", C.c_uint32), ("upd_pct_toast", C.c_uint32),
                ("emit_relations", C.c_uint32), ("emit_origin", C.c_uint32), ("ddl_every_txns", C.c_uint32),
                ("type_msg_pct", C.c_uint32), ("keepalive_every", C.c_uint32)]


_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libetlg_synth.so")
    src = os.path.join(_HERE, "csrc", "synth.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, src])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.synth_cfg_size.restype = C.c_size_t
        L.synth_state_size.restype = C.c_size_t
        assert L.synth_cfg_size() == C.sizeof(_Cfg), (L.synth_cfg_size(), C.sizeof(_Cfg))
        L.synth_init.argtypes = [C.POINTER(_Cfg), C.c_void_p]
        L.synth_fill.argtypes = [C.POINTER(_Cfg), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.c_uint64, C.POINTER(C.c_size_t)]
        L.synth_fill.restype = C.c_size_t
        _LIB = L
    return _LIB


# Postgres OIDs
INT8, INT2, INT4, TEXT, BOOL, NUMERIC, TIMESTAMPTZ, UUID = 20, 21, 23, 25, 16, 1700, 1184, 2950


def col(name, kind, oid, nullable=False, pk=False, null_pct=0, utf8_pct=0, min_len=0, max_len=0):
    return dict(name=name, kind=kind, oid=oid, nullable=nullable, pk=pk, null_pct=null_pct, utf8_pct=utf8_pct,
                min_len=min_len, max_len=max_len)


def table_fixed(rel_id=16384, name="bench_fixed"):
    """cfg 1/2: bench_fixed(c0..c4 int4 not null), PK c0; every value exactly 10 digits."""
    return dict(rel_id=rel_id, name=name,
                cols=[col(f"c{i}", CK_INT4_10D, INT4, pk=(i == 0)) for i in range(5)])


def table_mixed(rel_id=16385, name="bench_mixed"):
    """cfg 3: 12 columns with TEXT / NUMERIC / timestamptz / uuid."""
    return dict(rel_id=rel_id, name=name, cols=[
        col("id", CK_INT8_SEQ, INT8, pk=True), col("a", CK_INT4, INT4), col("b", CK_INT4, INT4),
        col("c", CK_INT2, INT2), col("f", CK_BOOL, BOOL), col("n1", CK_NUMERIC, NUMERIC), col("n2", CK_NUMERIC, NUMERIC),
        col("t1", CK_TEXT, TEXT, min_len=8, max_len=64, utf8_pct=10),
        col("t2", CK_TEXT, TEXT, min_len=0, max_len=256, utf8_pct=10),
        col("t3", CK_TEXT, TEXT, nullable=True, null_pct=30, min_len=1, max_len=128, utf8_pct=10),
        col("ts", CK_TIMESTAMPTZ, TIMESTAMPTZ), col("u", CK_UUID, UUID)])


def table_w8(rel_id=16386, name="bench_w8"):
    return dict(rel_id=rel_id, name=name, cols=[
        col("id", CK_INT8_SEQ, INT8, pk=True), col("a", CK_INT4, INT4), col("f", CK_BOOL, BOOL),
        col("n", CK_NUMERIC, NUMERIC), col("t", CK_TEXT, TEXT, min_len=4, max_len=48, utf8_pct=5),
        col("tn", CK_TEXT, TEXT, nullable=True, null_pct=25, min_len=1, max_len=32),
        col("ts", CK_TIMESTAMPTZ, TIMESTAMPTZ), col("u", CK_UUID, UUID)])


class Workload:
    def __init__(self, tables, seed, rows_per_txn=1000, mix=(100, 0, 0), upd_key=0, upd_toast=0, emit_relations=0,
                 emit_origin=0, ddl_every=0, type_msg_pct=0, keepalive_every=0, start_lsn=0x1000000, name="custom"):
        self.name = name
        self.tables = tables
        c = _Cfg()
        c.seed, c.start_lsn, c.ntables = seed, start_lsn, len(tables)
        for ti, t in enumerate(tables):
            ct = c.tables[ti]
            ct.rel_id, ct.ncols, ct.name = t["rel_id"], len(t["cols"]), t["name"].encode()
            for ci, cc in enumerate(t["cols"]):
                k = ct.cols[ci]
                k.kind, k.type_oid, k.nullable, k.pk = cc["kind"], cc["oid"], int(cc["nullable"]), int(cc["pk"])
                k.null_pct, k.utf8_pct, k.min_len, k.max_len = cc["null_pct"], cc["utf8_pct"], cc["min_len"], cc["max_len"]
                k.name = cc["name"].encode()
        c.rows_per_txn = rows_per_txn
        c.pct_insert, c.pct_update, c.pct_delete = mix
        c.upd_pct_key, c.upd_pct_toast = upd_key, upd_toast
        c.emit_relations, c.emit_origin, c.ddl_every_txns = emit_relations, emit_origin, ddl_every
        c.type_msg_pct, c.keepalive_every = type_msg_pct, keepalive_every
        self.cfg = c
        self.state = C.create_string_buffer(_lib().synth_state_size())
        _lib().synth_init(C.byref(c), self.state)

    def reset(self):
        _lib().synth_init(C.byref(self.cfg), self.state)

    def fill(self, cap_bytes, max_txns=1 << 62, max_frames=None):
        """Next batch: whole transactions, at most cap_bytes. Returns (np.uint8 buf, np.uint32 offsets)."""
        if max_frames is None:
            max_frames = cap_bytes // 24 + 64
        buf = np.empty(cap_bytes, dtype=np.uint8)
        offs = np.empty(max_frames + 1, dtype=np.uint32)
        nfr = C.c_size_t()
        n = _lib().synth_fill(C.byref(self.cfg), self.state, buf.ctypes.data, cap_bytes, offs.ctypes.data,
                              max_frames, max_txns, C.byref(nfr))
        return buf[:n], offs[:nfr.value + 1].copy()

    def schema_cols(self, t):
        return [(c["name"], c["oid"], c["nullable"], c["pk"]) for c in t["cols"]]

    def register(self, target, ready=True, snapshot_lsn=0):
        """Install stored schemas + Ready table state (+ Ready cache entry unless the
        stream carries its own Relation messages)."""
        for t in self.tables:
            target.schema_put(t["rel_id"], snapshot_lsn, self.schema_cols(t), name=t["name"])
            target.table_state(t["rel_id"], abi.TS_READY)
            if ready:
                n = len(t["cols"])
                target.table_ready(t["rel_id"], snapshot_lsn, [1] * n, [1 if c["pk"] else 0 for c in t["cols"]])


def cfg1(seed=0xE710001):
    """1M-row INSERT-only stream, 1000 txns x 1000 rows (CPU-runnable reference case)."""
    return Workload([table_fixed()], seed, rows_per_txn=1000, emit_relations=1, name="cfg1_insert_only_1M")


def cfg2(seed=0xE710002):
    """Fixed-width 5 x int4 INSERT tuples, 113-byte frames (coalesced baseline)."""
    return Workload([table_fixed()], seed, rows_per_txn=1000, name="cfg2_fixed_5xint4")


def cfg3(seed=0xE710003):
    """Mixed I/U/D 60/30/10 on the 12-column TEXT/NUMERIC schema."""
    return Workload([table_mixed()], seed, rows_per_txn=500, mix=(60, 30, 10), upd_key=10, upd_toast=5,
                    name="cfg3_mixed_12col")


def cfg5(seed=0xE710005):
    """3 tables round-robin, DDL message -> Relation every 200 txns, Type/Origin noise."""
    return Workload([table_fixed(), table_w8(), table_mixed()], seed, rows_per_txn=50, mix=(60, 30, 10), upd_key=10,
                    upd_toast=5, emit_relations=1, emit_origin=1, ddl_every=200, type_msg_pct=10, keepalive_every=997,
                    name="cfg5_ddl_3tables")


# ---- table-copy rows (COPY ... TO STDOUT, text format: what etlg_copy_decode takes) -----------------------------------
BYTEA, FLOAT8 = 17, 701
COPY_COLS = [("id", INT8, False, 1), ("a", INT4, False, 0), ("b", BOOL, False, 0), ("n", NUMERIC, False, 0),
             ("t", TEXT, False, 0), ("tn", TEXT, True, 0), ("ts", TIMESTAMPTZ, False, 0), ("u", UUID, False, 0),
             ("f", FLOAT8, False, 0), ("by", BYTEA, False, 0)]


def copy_rows(n, seed, clean=False):
    """n COPY text rows for COPY_COLS (int8, int4, bool, numeric, text, text NULLable, timestamptz, uuid, float8, bytea),
    every backslash escape of the format included; clean=True: text without any character COPY escapes."""
    import random
    rng = random.Random(seed)
    alphabet = "abcdefghij XYZ\t\n\\\r\x08\x0c\x0b\u00e9\u4e2d\U0001F600,;{}\"'"
    if clean:
        alphabet = "abcdefghijklmnopqrstuvwxyz ABCDEFGHIJ0123456789,;.-\u00e9"
    esc_map = {"\t": "\\t", "\n": "\\n", "\\": "\\\\", "\r": "\\r", "\x08": "\\b", "\x0c": "\\f", "\x0b": "\\v"}

    def esc(s):
        return "".join(esc_map.get(ch, ch) for ch in s)

    rows = []
    for i in range(n):
        txt = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 80)))
        f = [str(i), str(rng.randint(-2**31, 2**31 - 1)), rng.choice("tf"),
             rng.choice(["0", "-12.5", "123456789.000100", "NaN", "1e5", "0.000012"]),
             esc(txt), "\\N" if rng.random() < 0.3 else esc(txt[:10]),
             "2024-0%d-1%d 0%d:30:15.%06d+0%d" % (rng.randint(1, 9), rng.randint(0, 9), rng.randint(0, 9), rng.randint(0, 999999), rng.randint(0, 9)),
             "%08x-1111-2222-3333-%012x" % (rng.getrandbits(32), rng.getrandbits(48)),
             rng.choice(["1.5", "-0.25", "1e300", "3.141592653589793", "12345678901234567890123", "nan"]),
             "\\\\x" + "".join("%02x" % rng.getrandbits(8) for _ in range(rng.randint(0, 20)))]
        rows.append(("\t".join(f) + "\n").encode())
    return rows


# ---- the reference's type-matrix table (crates/etl/tests/replication_stream.rs:184-268, the row of :303-400): one column of every type
#      the reference's parser has an arm for, every array form, and the types it hands on as text — 68 replicated columns, i.e. a table
#      wider than any kernel's column masks. The cell texts are PostgreSQL's text output of the values that test inserts.
MATRIX_UUID = "a0eebc99-9c0b-4ef8-bb6d-6bb9bd380a11"
TYPE_MATRIX = [   # (name, type oid, nullable, text of the test's value; None = NULL)
    ("id", 20, False, "1"), ("bool_col", 16, False, "t"), ("char_col", 18, False, "x"), ("bpchar_col", 1042, False, "ab "),
    ("varchar_col", 1043, False, "varchar"), ("name_col", 19, False, "pg_name"), ("text_col", 25, False, "hello world"),
    ("text_null_col", 25, True, None), ("text_null_marker_literal_col", 25, False, "\\N"), ("text_embedded_null_marker_col", 25, False, "value\\Ntail"),
    ("money_col", 790, False, "$12.34"), ("int2_col", 21, False, "-123"), ("int4_col", 23, False, "456"), ("int8_col", 20, False, "7890123456"),
    ("oid_col", 26, False, "42"), ("float4_col", 700, False, "3.5"), ("float8_col", 701, False, "-7.25"), ("numeric_col", 1700, False, "12345.6789"),
    ("bytea_col", 17, False, "\\x0102ff"), ("date_col", 1082, False, "2026-01-02"), ("time_col", 1083, False, "12:30:45.123456"),
    ("timetz_col", 1266, False, "12:30:45.123456+02"), ("timestamp_col", 1114, False, "2026-01-02 03:04:05.123456"),
    ("timestamptz_col", 1184, False, "2026-01-02 03:04:05.123456+00"), ("uuid_col", 2950, False, MATRIX_UUID),
    ("json_col", 114, False, '{"kind":"json","n":1}'), ("jsonb_col", 3802, False, '{"kind": "jsonb", "nested": {"n": 2}}'),
    ("bool_arr", 1000, False, "{t,f,NULL}"), ("char_arr", 1002, False, "{a,NULL,b}"), ("bpchar_arr", 1014, False, '{"ab ",NULL,"cd "}'),
    ("varchar_arr", 1015, False, "{left,NULL,right}"), ("name_arr", 1003, False, "{alpha,NULL,beta}"), ("text_arr", 1009, False, "{hello,NULL,world}"),
    ("money_arr", 791, False, "{$12.34,NULL,-$0.01}"), ("int2_arr", 1005, False, "{-123,NULL,321}"), ("int4_arr", 1007, False, "{456,NULL,-654}"),
    ("int8_arr", 1016, False, "{7890123456,NULL,-9876543210}"), ("oid_arr", 1028, False, "{42,NULL,43}"), ("float4_arr", 1021, False, "{3.5,NULL,-1.25}"),
    ("float8_arr", 1022, False, "{-7.25,NULL,8.5}"), ("numeric_arr", 1231, False, "{12345.6789,NULL,-0.5}"),
    ("bytea_arr", 1001, False, '{"\\\\x00",NULL,"\\\\x0102"}'), ("date_arr", 1182, False, "{2026-01-02,NULL,2026-01-03}"),
    ("time_arr", 1183, False, "{12:30:45.123456,NULL,23:59:59}"), ("timetz_arr", 1270, False, "{12:30:45.123456+02,NULL,23:59:59-07:30}"),
    ("timestamp_arr", 1115, False, '{"2026-01-02 03:04:05.123456",NULL,"2026-01-03 04:05:06"}'),
    ("timestamptz_arr", 1185, False, '{"2026-01-02 03:04:05.123456+00",NULL,"2026-01-03 04:05:06+00"}'),
    ("uuid_arr", 2951, False, "{" + MATRIX_UUID + ",NULL,00000000-0000-0000-0000-000000000000}"),
    ("json_arr", 199, False, '{"{\\"a\\":1}",NULL,"{\\"b\\":2}"}'), ("jsonb_arr", 3807, False, '{"{\\"a\\": 1}",NULL,"{\\"b\\": 2}"}'),
    ("interval_col", 1186, False, "1 day 02:03:04"), ("interval_arr", 1187, False, '{"1 day",NULL,02:00:00}'),
    ("inet_col", 869, False, "192.0.2.1"), ("inet_arr", 1041, False, "{192.0.2.1,NULL,2001:db8::1}"),
    ("cidr_col", 650, False, "192.0.2.0/24"), ("cidr_arr", 651, False, "{192.0.2.0/24,NULL,2001:db8::/32}"),
    ("macaddr_col", 829, False, "aa:bb:cc:dd:ee:ff"), ("macaddr_arr", 1040, False, "{aa:bb:cc:dd:ee:ff,NULL,00:11:22:33:44:55}"),
    ("macaddr8_col", 774, False, "08:00:2b:01:02:03:04:05"), ("macaddr8_arr", 775, False, "{08:00:2b:01:02:03:04:05,NULL,02:03:04:05:06:07:08:09}"),
    ("xml_col", 142, False, '<root a="1"/>'), ("xml_arr", 143, False, "{<left/>,NULL,<right/>}"),
    ("int4_range_col", 3904, False, "[1,5)"), ("int4_range_arr", 3905, False, '{"[1,5)",NULL,"[10,20)"}'),
    ("num_multirange_col", 4532, False, "{[1.0,2.0)}"), ("num_multirange_arr", 6151, False, '{"{[1.0,2.0)}",NULL,"{[3.0,4.0)}"}'),
    ("int2_vector_col", 22, False, "1 2 3"), ("oid_vector_col", 30, False, "10 20"),
]
TYPE_MATRIX_COLS = [(n, oid, nullable, 1 if n == "id" else 0) for n, oid, nullable, _ in TYPE_MATRIX]
TYPE_MATRIX_REL = 16500


def type_matrix_stream(nrows, rows_per_txn=200, start_lsn=0x2000000, mix=False):
    """CopyData-framed pgoutput of `nrows` changes of the type-matrix table (ids 1, 2, ...): Begin / rows / Commit per transaction.
    mix: every fifth row an Update with a key image, every eleventh a Delete by key. Returns (np.uint8 bytes, np.uint32 offsets)."""
    import struct

    def tup(cells):
        out = [struct.pack(">h", len(cells))]
        for c in cells:
            if c is None:
                out.append(b"n")
            else:
                b = c.encode()
                out.append(b"t" + struct.pack(">i", len(b)) + b)
        return b"".join(out)

    tail = tup([t for _, _, _, t in TYPE_MATRIX[1:]])[2:]       # the cells behind the id (the count travels with the first)
    ncol = struct.pack(">h", len(TYPE_MATRIX))
    buf = bytearray()
    offs = [0]
    lsn = start_lsn

    def put(msg):
        nonlocal lsn
        lsn += 8
        payload = b"w" + struct.pack(">QQq", lsn, lsn, 0) + msg
        buf.extend(b"d" + struct.pack(">I", len(payload) + 4) + payload)
        offs.append(len(buf))

    i = 0
    while i < nrows:
        n = min(rows_per_txn, nrows - i)
        final = lsn + 8 * (n + 2)
        put(b"B" + struct.pack(">QqI", final, 0, 700 + i // rows_per_txn))
        for k in range(i, i + n):
            idt = str(k + 1).encode()
            cell0 = b"t" + struct.pack(">i", len(idt)) + idt
            row = ncol + cell0 + tail
            if mix and k % 11 == 10:
                put(b"D" + struct.pack(">I", TYPE_MATRIX_REL) + b"K" + struct.pack(">h", 1) + cell0)
            elif mix and k % 5 == 4:
                put(b"U" + struct.pack(">I", TYPE_MATRIX_REL) + b"K" + struct.pack(">h", 1) + cell0 + b"N" + row)
            else:
                put(b"I" + struct.pack(">I", TYPE_MATRIX_REL) + b"N" + row)
        put(b"C" + struct.pack(">bQQq", 0, final, final + 8, 0))
        i += n
    return np.frombuffer(bytes(buf), dtype=np.uint8), np.array(offs, dtype=np.uint32)


def type_matrix_register(target):
    target.schema_put(TYPE_MATRIX_REL, 0, TYPE_MATRIX_COLS, name="type_matrix")
    target.table_state(TYPE_MATRIX_REL, abi.TS_READY)
    n = len(TYPE_MATRIX_COLS)
    target.table_ready(TYPE_MATRIX_REL, 0, [1] * n, [1] + [0] * (n - 1))
