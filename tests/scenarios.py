"""Shared decode scenarios: each yields (prime(target), [batches]) where a batch is
(bytes, offsets_or_None). The same scenario is replayed on the oracle (CPU) and
on the HIP path (GPU); results must match byte for byte, errors included.

Hand-written cases follow the shapes the reference tests exercise
(crates/etl/src/postgres/codec/event.rs:1202-1695,
crates/etl/tests/replication.rs:2250-2579, crates/etl/tests/replication_stream.rs:31-409).
"""
import json

import numpy as np

from etl_amd import abi, synth
from tests import pgwire as W

N, U = W.NULL, W.TOAST
INT8, INT4, INT2, TEXT, BOOL, NUMERIC, BYTEA = 20, 23, 21, 25, 16, 1700, 17
DATE, TIME, TIMETZ, TIMESTAMP, TIMESTAMPTZ, UUID, JSONB, FLOAT8, FLOAT4, OID, INT4_A = 1082, 1083, 1266, 1114, 1184, 2950, 3802, 701, 700, 26, 1007


class Scenario:
    def __init__(self, name, prime, batches, worker=None):
        self.name, self.prime, self.batches, self.worker = name, prime, batches, worker

    def __repr__(self):
        return self.name


def simple_table(cols, table_id=42, ident=None, repl=None, state=abi.TS_READY, state_lsn=0, ready=True):
    def prime(t):
        t.schema_put(table_id, 0, cols)
        if state is not None:
            t.table_state(table_id, state, state_lsn)
        if ready:
            n = len(cols)
            r = repl if repl is not None else [1] * n
            i = ident if ident is not None else [1 if c[3] else 0 for c in cols]
            assert t.table_ready(table_id, 0, r, i) >= 0
    return prime


def txn(msgs, final=0x2000, lsn0=0x1000, commit=True):
    s = W.Stream(lsn=lsn0)
    s.add(W.begin(final, ts=1234, xid=77))
    for m in msgs:
        if isinstance(m, tuple) and m[0] == "raw":
            s.add_payload(m[1])
        else:
            s.add(m)
    if commit:
        s.add(W.commit(final, final + 8, ts=5678, flags=0), lsn=final)
    return s


def one(name, prime, msgs, **kw):
    s = txn(msgs, **kw)
    return Scenario(name, prime, [(s.bytes(), s.offsets)])


COLS2 = [("id", INT8, False, 1), ("payload", TEXT, True, 0)]
COMPOSITE = [("id", INT8, False, 1), ("name", TEXT, False, 0), ("surname", TEXT, False, 1), ("city", TEXT, False, 0),
             ("large_text", TEXT, False, 0)]
USERS = [("id", INT8, False, 1), ("name", TEXT, False, 0), ("surname", TEXT, False, 1), ("city", TEXT, False, 0)]
ALLTYPES = [("id", INT8, False, 1), ("b", BOOL, True, 0), ("i2", INT2, True, 0), ("i4", INT4, True, 0), ("o", OID, True, 0),
            ("n", NUMERIC, True, 0), ("by", BYTEA, True, 0), ("d", DATE, True, 0), ("t", TIME, True, 0),
            ("tz", TIMETZ, True, 0), ("ts", TIMESTAMP, True, 0), ("tstz", TIMESTAMPTZ, True, 0), ("u", UUID, True, 0),
            ("j", JSONB, True, 0), ("f8", FLOAT8, True, 0), ("f4", FLOAT4, True, 0), ("arr", INT4_A, True, 0),
            ("s", TEXT, True, 0)]


def alltypes_row(**over):
    base = dict(id="1", b="t", i2="-123", i4="456", o="42", n="12345.6789", by="\\x0102ff", d="2026-01-02",
                t="12:30:45.123456", tz="12:30:45.123456+02", ts="2026-01-02 03:04:05.123456",
                tstz="2026-01-02 03:04:05.123456+00", u="123e4567-e89b-12d3-a456-426614174000",
                j='{"kind":"jsonb","nested":{"n":2}}', f8="-7.25", f4="3.5", arr="{1,NULL,3}", s="hello wörld")
    base.update(over)
    return [base[c[0]] for c in ALLTYPES]


def hand_written():
    S = []
    P2 = simple_table(COLS2)
    S.append(one("insert_basic", P2, [W.insert(42, ["1", "héllo"]), W.insert(42, ["-9223372036854775808", N])]))
    S.append(one("payload_sizes", simple_table([("a", TEXT, True, 1), ("b", TEXT, True, 0)]),
                 [W.insert(42, ["é", N]), W.update(42, ["é", U], key=["old", N]), W.delete(42, old=["gone", N])]))
    S.append(one("toast_partial", P2, [W.update(42, ["1", U])]))
    S.append(one("toast_from_full_old", simple_table(COLS2, ident=[1, 1]), [W.update(42, ["1", U], old=["1", "toast"])]))
    S.append(one("toast_from_key", simple_table([("id", INT8, False, 1), ("payload", TEXT, False, 1)]),
                 [W.update(42, ["2", U], key=["1", "toast"])]))
    S.append(one("key_full_width_and_dense", simple_table(USERS, table_id=1, ident=[1, 0, 1, 0]),
                 [W.delete(1, key=["1", "alice", "smith", "toast"]), W.delete(1, key=["1", "smith"])]))
    S.append(one("update_shapes", simple_table(COMPOSITE), [
        W.update(42, ["1", "alice", "smith", "vienna", "toast"]),
        W.update(42, ["1", "alice", "smith", "vienna", U]),
        W.update(42, ["1", "alice", "smithers", "rome", "toast"], key=["1", N, "smith", N, N])]))
    S.append(one("alt_identity", simple_table(USERS, table_id=43, ident=[0, 1, 1, 0]),
                 [W.update(43, ["1", "alice", "smith", "vienna"], key=[N, "alice", "smith", N]),
                  W.delete(43, key=[N, "alice", "smith", N])]))
    S.append(one("full_identity", simple_table(USERS, table_id=44, ident=[1, 1, 1, 1]),
                 [W.update(44, ["1", "alice", "smith", "vienna"], old=["1", "alice", "smith", "rome"]),
                  W.delete(44, old=["1", "alice", "smith", "rome"])]))
    S.append(one("column_filtered_publication", simple_table(USERS, repl=[1, 0, 1, 1], ident=[1, 0, 1, 0]),
                 [W.insert(42, ["5", "smith", "paris"]), W.delete(42, key=["5", "smith"])]))
    # all value classes in one row + deferred shapes + nulls
    PA = simple_table(ALLTYPES)
    S.append(one("all_types_row", PA, [W.insert(42, alltypes_row()),
                                       W.insert(42, alltypes_row(id="2", b=N, i2=N, i4=N, o=N, n=N, by=N, d=N, t=N, tz=N, ts=N, tstz=N, u=N, j=N, f8=N, f4=N, arr=N, s=N))]))
    S.append(one("deferred_temporal_shapes", PA, [W.insert(42, alltypes_row(
        d="2023-1-01", t="23:59:60", tz="12:30:00.1234567890+02", ts="2023-12-25 12:30:45.1234567890",
        tstz="2023-12-25 23:59:60+00"))]))
    S.append(one("uuid_forms", PA, [W.insert(42, alltypes_row(u="123E4567E89B12D3A456426614174000")),
                                    W.insert(42, alltypes_row(u="{123e4567-e89b-12d3-a456-426614174000}")),
                                    W.insert(42, alltypes_row(u="urn:uuid:123e4567-e89b-12d3-a456-426614174000"))]))
    nums = ["0", "-0.00", "0e-6", "1e131071", "1e-16383", "0.0012000", "9999.9999", "10000.0001", "0000120.00", "1200000",
            "NaN", "  NaN   ", "Infinity", "-inf", "+Infinity   ", "1_000.5", "123e-2", ".5", "5.", " 1.5 ",
            "9" * 200 + "." + "9" * 150, "-1.23e2"]
    S.append(one("numeric_shapes", PA, [W.insert(42, alltypes_row(id=str(i), n=x)) for i, x in enumerate(nums)]))
    tz = ["2026-01-01 12:30:00+02", "2026-01-01 12:30:00+0230", "2026-01-01 12:30:00+023015", "2026-01-01 12:30:00+02:30:15",
          "2026-01-01 12:30:00.123456-07:30", "2026-01-01 12:30:00+15:59:59", "2026-01-01 12:30:00-15:59:59",
          "0001-01-01 00:00:00+15:00", "9999-12-31 23:59:59.999999-15:59:59", "2024-02-29 23:59:59+00"]
    S.append(one("timestamptz_offsets", PA, [W.insert(42, alltypes_row(id=str(i), tstz=x)) for i, x in enumerate(tz)]))
    ints = [("i2", "-32768"), ("i2", "+32767"), ("i4", "-2147483648"), ("i4", "+0002147483647"), ("o", "4294967295"), ("o", "+7"),
            ("id", "9223372036854775807")]
    S.append(one("int_boundaries", PA, [W.insert(42, alltypes_row(**{k: v})) for k, v in ints]))
    S.append(one("bytea_shapes", PA, [W.insert(42, alltypes_row(by=x)) for x in ["\\x", "\\xaBcD", "\\x00" * 1, "\\x" + "ff" * 33]]))
    S.append(one("long_text", PA, [W.insert(42, alltypes_row(s="x" * 5000 + "é" * 777))]))

    # ---- value errors: (name, column override, expected handled by the oracle)
    bad = [("bad_bool", dict(b="true")), ("bad_int_overflow", dict(i4="2147483648")), ("bad_int_char", dict(i2="12a")),
           ("bad_int_empty", dict(i4="")), ("bad_int_lone_sign", dict(i4="-")), ("bad_oid_negative", dict(o="-1")),
           ("bad_numeric", dict(n="1.2.3")), ("bad_numeric_range", dict(n="1e131072")), ("bad_numeric_signed_nan", dict(n="-NaN")),
           ("bad_bytea_prefix", dict(by="0x41")), ("bad_bytea_odd", dict(by="\\x414")), ("bad_bytea_digit", dict(by="\\x4g")),
           ("bad_tstz_no_offset", dict(tstz="2026-01-01 12:30:00")), ("bad_tstz_offset", dict(tstz="2026-01-01 12:30:00+16")),
           ("bad_timetz_no_offset", dict(tz="12:30:00")), ("bad_timetz_offset", dict(tz="12:30:00+15:60")),
           ("bad_uuid", dict(u="invalid-uuid")), ("bad_uuid_len36", dict(u="123e4567-e89b-12d3-a456_426614174000")),
           ("bad_utf8_text", dict(s=b"ab\xff")), ("bad_utf8_in_int", dict(i4=b"1\xc3")), ("bad_utf8_deferred_json", dict(j=b"\xed\xa0\x80"))]
    for name, over in bad:
        S.append(one(name, PA, [W.insert(42, alltypes_row(id="1")), W.insert(42, alltypes_row(**over)), W.insert(42, alltypes_row(id="3"))]))
    # ---- tuple-level errors
    S.append(one("err_required_null", simple_table([("id", INT8, False, 1), ("d", DATE, False, 0)]), [W.insert(42, ["1", N])]))
    S.append(one("err_width", P2, [W.insert(42, ["1"])]))
    S.append(one("err_width_update_new", P2, [W.update(42, ["1", "a", "b"])]))
    S.append(one("err_full_row_toast", P2, [W.insert(42, ["1", U])]))
    S.append(one("err_binary_cell", P2, [W.insert(42, ["1", W.Binary(b"\x01\x02")])]))
    S.append(one("err_key_shape", simple_table(USERS, ident=[1, 0, 1, 0]), [W.delete(42, key=["1", "a", "b"])]))
    S.append(one("err_key_no_identity", simple_table(USERS, ident=[0, 0, 0, 0]), [W.delete(42, key=["1"])]))
    S.append(one("err_key_toast", simple_table(USERS, ident=[1, 0, 1, 0]), [W.delete(42, key=["1", U])]))
    S.append(one("err_old_before_new", simple_table(COLS2, ident=[1, 1]), [W.update(42, ["x", "new"], old=["y", "old"])]))
    # ---- wire-level errors
    s = txn([W.insert(42, ["1", "a"])])
    raw = bytearray(s.bytes())
    raw[s.offsets[1] + 40] ^= 0xFF  # corrupt the column count area of the insert
    S.append(Scenario("err_wire_corrupt_tuple", P2, [(bytes(raw), s.offsets)]))
    S.append(one("err_wire_unknown_tag", P2, [("raw", W.xlog(0x1100, b"Zjunk"))]))
    S.append(one("err_wire_bad_tuple_marker", P2, [("raw", W.xlog(0x1100, b"I" + b"\x00\x00\x00\x2a" + b"X" + W.tuple_data(["1", "a"])))]))
    S.append(one("err_wire_negative_len", P2, [("raw", W.xlog(0x1100, b"I\x00\x00\x00\x2aN\x00\x01t\xff\xff\xff\xff"))]))
    S.append(one("err_wire_truncated_begin", P2, [("raw", W.xlog(0x1100, b"B\x00\x00"))]))
    S.append(one("err_wire_delete_without_tuple", P2, [("raw", W.xlog(0x1100, b"D\x00\x00\x00\x2aN" + W.tuple_data(["1", "a"])))]))
    S.append(one("err_wire_outer_tag", P2, [("raw", b"x123456789012345678901234567890")]))
    s2 = txn([W.insert(42, ["1", "a"])])
    bad_off = list(s2.offsets); bad_off[2] -= 1; bad_off[2:] = bad_off[2:]
    S.append(Scenario("err_sidecar_disagrees", P2, [(s2.bytes(), bad_off)]))
    # ---- transaction state
    s3 = W.Stream(); s3.add(W.insert(42, ["1", "a"]))
    S.append(Scenario("err_insert_outside_txn", P2, [(s3.bytes(), s3.offsets)]))
    s4 = W.Stream(); s4.add(W.commit(0x2000, 0x2008))
    S.append(Scenario("err_commit_without_begin", P2, [(s4.bytes(), s4.offsets)]))
    s5 = W.Stream(); s5.add(W.begin(0x2000)); s5.add(W.insert(42, ["1", "a"])); s5.add(W.commit(0x2001, 0x2008)); s5.add(W.begin(0x3000))
    S.append(Scenario("err_commit_lsn_mismatch", P2, [(s5.bytes(), s5.offsets)]))
    s6 = W.Stream(); s6.add(W.begin(0x2000)); s6.add(W.commit(0x2000, 0x2008)); s6.add(W.truncate([42]))
    S.append(Scenario("err_truncate_outside_txn", P2, [(s6.bytes(), s6.offsets)]))
    # ordinals: O/Y/M(unknown prefix)/k do not consume; filtered rows do
    s7 = W.Stream()
    s7.add(W.begin(0x2000)); s7.add(W.origin(5, "o")); s7.add(W.insert(42, ["1", "a"])); s7.add(W.type_msg(9, "p", "t"))
    s7.add(W.insert(99, ["zz"])); s7.add_payload(W.keepalive(0x1234)); s7.add(W.message("other", "{}")); s7.add(W.insert(42, ["2", "b"]))
    s7.add(W.commit(0x2000, 0x2008)); s7.add(W.message("other", "x", transactional=False)); s7.add(W.begin(0x3000)); s7.add(W.insert(42, ["3", N]))
    s7.add(W.commit(0x3000, 0x3008))
    S.append(Scenario("ordinals_and_noise", P2, [(s7.bytes(), s7.offsets)]))
    # ---- ownership
    S.append(one("own_sync_done_before", simple_table(COLS2, state=abi.TS_SYNC_DONE, state_lsn=0x2001), [W.insert(42, ["1", "a"])]))
    S.append(one("own_sync_done_at", simple_table(COLS2, state=abi.TS_SYNC_DONE, state_lsn=0x2000), [W.insert(42, ["1", "a"])]))
    S.append(one("own_other_state", simple_table(COLS2, state=abi.TS_OTHER), [W.insert(42, ["1", "a"]), W.truncate([42])]))
    S.append(one("own_unknown_table", P2, [W.insert(4242, ["1", "a"]), W.insert(42, ["1", "a"])]))
    S.append(one("err_missing_shared_state", simple_table(COLS2, ready=False), [W.insert(42, ["1", "a"])]))
    S.append(Scenario("table_sync_worker", P2, [(txn([W.insert(42, ["1", "a"]), W.insert(43, ["1", "a"]), W.truncate([43, 42])]).bytes(), None)],
                      worker=(abi.WORKER_TABLE_SYNC, 42, 0)))

    def prime_two(t):
        simple_table(COLS2)(t)
        simple_table(USERS, table_id=43)(t)
    S.append(one("truncate_two_tables", prime_two, [W.truncate([42, 43], options=3), W.truncate([4242]), W.truncate([], options=1)]))
    # ---- relation + DDL flows (crates/etl/tests/replication.rs:2250-2328)
    REL_COLS = [(1, "id", INT8, -1), (0, "payload", TEXT, -1)]
    PR = simple_table(COLS2, ready=False)
    S.append(one("relation_then_rows", PR, [W.relation(42, "public", "t", "d", REL_COLS), W.insert(42, ["1", "a"]),
                                            W.delete(42, key=["1"])]))
    S.append(one("relation_full_identity", PR, [W.relation(42, "public", "t", "f", [(0, "id", INT8, -1), (0, "payload", TEXT, -1)]),
                                                W.delete(42, old=["1", "x"])]))
    S.append(one("relation_subset_of_columns", PR, [W.relation(42, "public", "t", "d", [(1, "id", INT8, -1)]), W.insert(42, ["7"])]))
    S.append(one("err_relation_unknown_column", PR, [W.relation(42, "public", "t", "d", [(1, "id", INT8, -1), (0, "nope", TEXT, -1)]),
                                                      W.insert(42, ["1"])]))
    S.append(one("err_relation_no_schema", lambda t: t.table_state(42, abi.TS_READY), [W.relation(42, "public", "t", "d", REL_COLS)]))
    S.append(one("relation_not_owned", simple_table(COLS2, state=abi.TS_OTHER, ready=False),
                 [W.relation(42, "public", "t", "d", REL_COLS), W.insert(42, ["1", "a"])]))
    s8 = W.Stream(); s8.add(W.relation(42, "public", "t", "d", REL_COLS))
    S.append(Scenario("err_relation_outside_txn", PR, [(s8.bytes(), s8.offsets)]))

    def ddl(cols, oid=42, pk=(1,)):
        return json.dumps({"command_tag": "ALTER TABLE", "nspname": "public", "relname": "t", "oid": oid,
                           "identity": {"primary_key_attnums": list(pk), "relreplident": "d", "replica_identity_index_attnums": []},
                           "columns": [{"attname": n, "atttypid": o, "atttypmod": -1, "attnum": i + 1, "attnotnull": nn,
                                        "default_expression": None} for i, (n, o, nn) in enumerate(cols)], "extra": {"x": [1, 2.5e3, "é\\n"]}})
    NEW = [("id", INT8, True), ("payload", TEXT, False), ("extra", INT4, False)]
    S.append(one("ddl_then_relation_then_rows", P2, [
        W.insert(42, ["1", "old"]), W.message("supabase_etl_ddl", ddl(NEW)),
        W.relation(42, "public", "t", "d", REL_COLS + [(0, "extra", INT4, -1)]), W.insert(42, ["2", "new", "77"])]))
    S.append(one("err_row_while_waiting_for_relation", P2, [W.message("supabase_etl_ddl", ddl(NEW)), W.insert(42, ["2", "new", "77"])]))
    S.append(one("err_ddl_bad_json", P2, [W.message("supabase_etl_ddl", '{"command_tag": 1}'), W.insert(42, ["1", "a"])]))
    S.append(one("err_ddl_duplicate_field", P2, [W.message("supabase_etl_ddl", ddl(NEW)[:-1] + ',"oid":43}')]))
    s9 = W.Stream(); s9.add(W.message("supabase_etl_ddl", ddl(NEW)))
    S.append(Scenario("err_ddl_outside_txn", P2, [(s9.bytes(), s9.offsets)]))
    S.append(one("ddl_not_owned_is_skipped", simple_table(COLS2, state=abi.TS_OTHER), [W.message("supabase_etl_ddl", ddl(NEW)), W.insert(42, ["1", "a"])]))
    S.append(one("err_after_ddl_rolls_back_later_control", P2, [
        W.insert(42, ["x", "bad"]), W.message("supabase_etl_ddl", ddl(NEW)), W.relation(42, "public", "t", "d", REL_COLS + [(0, "extra", INT4, -1)])]))
    # ---- multi-batch carry: a transaction spanning three decode calls, then an error
    big = txn([W.insert(42, [str(i), "v%d" % i]) for i in range(700)], final=0x9000)
    cut1, cut2 = 250, 600
    o = big.offsets
    b1 = (big.bytes()[:o[cut1]], o[:cut1 + 1])
    b2 = (big.bytes()[o[cut1]:o[cut2]], [x - o[cut1] for x in o[cut1:cut2 + 1]])
    b3 = (big.bytes()[o[cut2]:], [x - o[cut2] for x in o[cut2:]])
    S.append(Scenario("txn_spanning_batches", P2, [b1, b2, b3, (b"", [0])]))
    S.append(Scenario("empty_batch", P2, [(b"", [0])]))
    S.append(Scenario("no_sidecar_host_scan", P2, [(big.bytes(), None)]))
    trail = txn([W.insert(42, ["1", "a"])])
    S.append(Scenario("no_sidecar_trailing_garbage", P2, [(trail.bytes() + b"d\x00\x00", None)]))
    return S


def synth_scenarios(nbytes=192 << 10):
    out = []
    for mk in (synth.cfg1, synth.cfg2, synth.cfg3, synth.cfg5):
        w = mk()
        batches = []
        for _ in range(3):
            buf, offs = w.fill(nbytes)
            batches.append((buf, offs))

        def prime(t, w=w):
            w.register(t, ready=not w.cfg.emit_relations)
        out.append(Scenario("synth_" + w.name, prime, batches))
    return out


def all_scenarios():
    return hand_written() + synth_scenarios()


def replay(target, sc):
    """Replays a scenario on `target` (oracle wrapper or etl_amd.Decoder).
    Returns [(err_code, err_kind, err_desc, err_frame, HostBatch)] per batch."""
    if sc.worker:
        target.set_worker(*sc.worker)
    sc.prime(target)
    out = []
    for buf, offs in sc.batches:
        a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        b = target.decode(a, offs)
        if hasattr(b, "err_code"):  # oracle
            out.append((b.err_code, b.err_kind, b.err_desc, b.err_frame, b.host_batch()))
        else:
            e = b.error
            out.append((e.code if e else 0, e.kind if e else 0, e.description if e else "", e.frame_index if e else -1, b.host()))
    return out
