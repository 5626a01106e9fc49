"""Event-level known answers: raw pgoutput bytes -> Event, transcribed from
crates/etl/src/postgres/codec/event.rs:1202-1695 and run through the oracle's
wire parser + apply-loop state machine (both FULL and CONTRACT mode)."""
import pytest

from etl_amd import abi
from oracle import oracle
from tests import pgwire as W
from tests.golden import reference_kats as K

INT8, TEXT, DATE = 20, 25, 1082
N, U = W.NULL, W.TOAST


def make(cols, table_id=42, repl=None, ident=None, mode=oracle.MODE_CONTRACT):
    """cols: (name, oid, nullable, pk). Registers the table Ready with the given masks."""
    o = oracle.Oracle(mode=mode)
    o.schema_put(table_id, 0, cols)
    o.table_state(table_id, abi.TS_READY)
    n = len(cols)
    repl = repl if repl is not None else [1] * n
    ident = ident if ident is not None else [1 if c[3] else 0 for c in cols]
    assert o.table_ready(table_id, 0, repl, ident) >= 0
    return o


def run(o, msgs, table_id=42):
    s = W.Stream()
    s.add(W.begin(0x2000), lsn=10)
    for m in msgs:
        s.add(m, lsn=10)
    b = o.decode(s.bytes(), s.offsets)
    return b


# event_schema / composite_primary_key_schema / alternative / full (event.rs:1016-1074)
COMPOSITE = [("id", INT8, False, 1), ("name", TEXT, False, 0), ("surname", TEXT, False, 1),
             ("city", TEXT, False, 0), ("large_text", TEXT, False, 0)]
USERS = [("id", INT8, False, 1), ("name", TEXT, False, 0), ("surname", TEXT, False, 1), ("city", TEXT, False, 0)]


@pytest.mark.parametrize("mode", [oracle.MODE_FULL, oracle.MODE_CONTRACT])
class TestEventKats:
    def test_payload_sizes(self, mode):
        # event.rs:1202-1230
        o = make([("a", TEXT, True, 1), ("b", TEXT, True, 0)], mode=mode)
        b = run(o, [W.insert(42, ["é", N])])
        assert b.err_code == 0
        assert b.host_batch().payload_bytes == (2, 0, 0) if mode == oracle.MODE_CONTRACT else True
        o2 = make([("a", TEXT, True, 1), ("b", TEXT, True, 0)])
        b2 = run(o2, [W.insert(42, ["é", N]), W.update(42, ["é", U], key=["old", N]), W.delete(42, old=["gone", N])])
        assert b2.host_batch().payload_bytes == (2, 5, 4)

    def test_null_in_non_nullable_full_row(self, mode):
        # event.rs:1356-1368
        o = make([("id", INT8, False, 1), ("d", DATE, False, 0)], mode=mode)
        b = run(o, [W.insert(42, ["1", N])])
        assert (b.err_code, b.err_kind, b.err_desc) == (abi.E_REQUIRED_NULL, abi.InvalidData, "Required column missing from tuple")
        assert b.err_frame == 1 and b.n_events == 1  # the Begin before it stays valid

    def test_toast_partial_without_old(self, mode):
        # event.rs:1370-1390
        o = make([("id", INT8, False, 1), ("payload", TEXT, False, 0)], mode=mode)
        b = run(o, [W.update(42, ["1", U])])
        assert b.err_code == 0
        assert b.event_repr(1).endswith('old=None new=Partial[I64(1), Missing]')

    def test_toast_reuses_full_old(self, mode):
        # event.rs:1392-1417
        o = make([("id", INT8, False, 1), ("payload", TEXT, False, 0)], ident=[1, 1], mode=mode)
        b = run(o, [W.update(42, ["1", U], old=["1", "toast"])])
        assert b.err_code == 0
        assert b.event_repr(1).endswith('old=Full[I64(1), String("toast")] new=Full[I64(1), String("toast")]')

    def test_toast_reuses_key_value(self, mode):
        # event.rs:1419-1443
        o = make([("id", INT8, False, 1), ("payload", TEXT, False, 1)], mode=mode)
        b = run(o, [W.update(42, ["2", U], key=["1", "toast"])])
        assert b.err_code == 0
        assert b.event_repr(1).endswith('old=Key[I64(1), String("toast")] new=Full[I64(2), String("toast")]')

    def test_key_tuple_full_width_and_dense(self, mode):
        # event.rs:1445-1499 (identity mask [1,0,1,0])
        cols = [("id", INT8, False, 1), ("name", TEXT, False, 0), ("surname", TEXT, False, 1), ("payload", TEXT, False, 0)]
        o = make(cols, table_id=1, mode=mode)
        o.table_state(1, abi.TS_READY)
        b = run(o, [W.delete(1, key=["1", "alice", "smith", "toast"]), W.delete(1, key=["1", "smith"])])
        assert b.err_code == 0
        assert b.event_repr(1).endswith('old=Key[I64(1), String("smith")]')
        assert b.event_repr(2).endswith('old=Key[I64(1), String("smith")]')

    def test_update_shapes(self, mode):
        # event.rs:1501-1597 composite primary key
        o = make(COMPOSITE, mode=mode)
        b = run(o, [
            W.update(42, ["1", "alice", "smith", "vienna", "toast"]),
            W.update(42, ["1", "alice", "smith", "vienna", U]),
            W.update(42, ["1", "alice", "smithers", "rome", "toast"], key=["1", N, "smith", N, N]),
        ])
        assert b.err_code == 0
        assert b.event_repr(1).endswith(
            'old=None new=Full[I64(1), String("alice"), String("smith"), String("vienna"), String("toast")]')
        assert b.event_repr(2).endswith(
            'old=None new=Partial[I64(1), String("alice"), String("smith"), String("vienna"), Missing]')
        assert 'old=Key[I64(1), String("smith")] new=Full[' in b.event_repr(3)

    def test_alternative_identity(self, mode):
        # event.rs:1599-1637, 1672-1695 (identity mask [0,1,1,0])
        o = make(USERS, table_id=43, ident=[0, 1, 1, 0], mode=mode)
        b = run(o, [W.update(43, ["1", "alice", "smith", "vienna"], key=[N, "alice", "smith", N]),
                    W.delete(43, key=[N, "alice", "smith", N])])
        assert b.err_code == 0
        assert b.event_repr(1).endswith(
            'old=Key[String("alice"), String("smith")] new=Full[I64(1), String("alice"), String("smith"), String("vienna")]')
        assert b.event_repr(2).endswith('old=Key[String("alice"), String("smith")]')

    def test_full_identity(self, mode):
        # event.rs:1639-1670 (identity mask [1,1,1,1])
        o = make(USERS, table_id=44, ident=[1, 1, 1, 1], mode=mode)
        b = run(o, [W.update(44, ["1", "alice", "smith", "vienna"], old=["1", "alice", "smith", "rome"])])
        assert b.err_code == 0
        assert 'old=Full[I64(1), String("alice"), String("smith"), String("rome")]' in b.event_repr(1)


def test_type_matrix_row_scalars():
    """One row of the reference's type matrix (crates/etl/tests/replication_stream.rs:303-345
    inputs, :613-700 assertions); text forms are PostgreSQL's ISO/UTC renderings."""
    cols = [("id", 20, True, 1), ("bool_col", 16, False), ("char_col", 18, False), ("bpchar_col", 1042, False),
            ("varchar_col", 1043, False), ("name_col", 19, False), ("text_col", 25, False), ("text_null_col", 25, True),
            ("lit", 25, False), ("emb", 25, False), ("money_col", 790, False), ("int2_col", 21, False),
            ("int4_col", 23, False), ("int8_col", 20, False), ("oid_col", 26, False), ("float4_col", 700, False),
            ("float8_col", 701, False), ("numeric_col", 1700, False), ("bytea_col", 17, False), ("date_col", 1082, False),
            ("time_col", 1083, False), ("timetz_col", 1266, False), ("timestamp_col", 1114, False),
            ("timestamptz_col", 1184, False), ("uuid_col", 2950, False)]
    cols = [c if len(c) == 4 else c + (0,) for c in cols]
    o = make(cols, mode=oracle.MODE_FULL)
    row = ["7", "t", "x", "ab ", "varchar", "pg_name", "hello world", N, "\\N", "value\\Ntail", "$12.34", "-123", "456",
           "7890123456", "42", "3.5", "-7.25", "12345.6789", "\\x0102ff", "2026-01-02", "12:30:45.123456",
           "12:30:45.123456+02", "2026-01-02 03:04:05.123456", "2026-01-02 03:04:05.123456+00",
           "123e4567-e89b-12d3-a456-426614174000"]
    b = run(o, [W.insert(42, row)])
    assert b.err_code == 0
    exp = ['I64(7)', 'Bool(true)', 'String("x")', 'String("ab ")', 'String("varchar")', 'String("pg_name")',
           'String("hello world")', 'Null', 'String("\\N")', 'String("value\\Ntail")', 'String("$12.34")', 'I16(-123)',
           'I32(456)', 'I64(7890123456)', 'U32(42)', K.f32(3.5), K.f64(-7.25), 'Numeric(+,w=1,s=4,[1,2345,6789])',
           'Bytes(0102ff)', 'Date(2026-01-02)', 'Time(12:30:45.123456000)', 'TimeTz(12:30:45.123456000,7200)',
           'Timestamp(2026-01-02 03:04:05.123456000)', 'TimestampTz(2026-01-02 03:04:05.123456000)',
           'Uuid(123e4567e89b12d3a456426614174000)']
    assert b.event_repr(1).endswith("new=Full[" + ", ".join(exp) + "]")
