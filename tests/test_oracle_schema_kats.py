"""The reference's known-answer tests for the schema side input — replication / identity masks
(crates/etl/src/schema.rs:802-988, crates/etl/src/postgres/codec/event.rs:1232-1354) and the shared table cache
(crates/etl/src/replication/table_cache.rs:185-302), transcribed in tests/golden/reference_kats.py — replayed on

  * the host-side mirror of the mask rules (etl_amd/schema.py),
  * the ORACLE, through what a stream can do to it: a Relation message names columns (the masks are then visible in the schema
    slot of the RelationEvent), a DDL message puts a table into WaitingForRelation, table_ready / table_forget are the copy path's
    note_ready / remove_table,
  * the product (GPU tier; the CPU suite runs it on the emulator build): same drivers, through the C ABI."""
import json

import numpy as np
import pytest

from etl_amd import abi, schema
from tests import pgwire as W
from tests.golden import reference_kats as K


# ---------------------------------------------------------------------------------------------- the mirror (etl_amd/schema.py)
@pytest.mark.parametrize("name,names,want", K.SCHEMA_RS_REPLICATION_MASK, ids=[k[0] for k in K.SCHEMA_RS_REPLICATION_MASK])
def test_mirror_replication_mask_try_build(name, names, want):
    if isinstance(want, tuple):
        with pytest.raises(schema.UnknownReplicatedColumns) as e:
            schema.replication_mask_try_build(K.TEST_TABLE, names)
        assert sorted(e.value.columns) == want[1]
    else:
        assert schema.replication_mask_try_build(K.TEST_TABLE, names) == want


def test_mirror_build_or_all_and_all():
    for name, names, want in K.SCHEMA_RS_BUILD_OR_ALL:
        assert schema.replication_mask_build_or_all(K.TEST_TABLE, names) == want, name
    assert schema.replication_mask_all(K.TEST_TABLE) == [1, 1, 1]   # schema.rs:894


def test_mirror_identity_type_and_primary_key_coverage():
    for name, rm, im, want in K.SCHEMA_RS_IDENTITY_TYPE:
        im = schema.identity_mask_default(K.TEST_TABLE, rm) if im is None else im
        assert schema.infer_identity_type(K.TEST_TABLE, rm, im) == want, name
    for name, cols, rm, im, omitted in K.SCHEMA_RS_PK_REPLICATED:
        assert schema.unreplicated_primary_key_columns(cols, rm) == omitted, name


@pytest.mark.parametrize("case", K.EVENT_RS_BUILD_IDENTITY, ids=[k[0] for k in K.EVENT_RS_BUILD_IDENTITY])
def test_mirror_build_identity_mask(case):
    name, cols, rm, pk, mode, idx, want_mask, want_type = case
    im = schema.identity_mask_from_metadata(cols, rm, pk, mode, idx)
    if want_mask is not None:
        assert im == want_mask
    if want_type is not None:
        assert schema.infer_identity_type(cols, rm, im) == want_type


# ------------------------------------------------------------------------------------------------- oracle / product drivers
def _fresh(kind):
    if kind == "oracle":
        from oracle import oracle
        return oracle.Oracle()
    from etl_amd.decoder import Decoder
    return Decoder(0)


def _decode(t, s):
    return t.decode(np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets)


def _err(b):
    if hasattr(b, "err_code"):
        return b.err_code
    return b.error.code if b.error is not None else 0


def _host(b):
    return b.host_batch() if hasattr(b, "host_batch") else b.host()


def _relation_masks(kind, cols, names, ident_flags, replident="d"):
    """Installs `cols` as the stored schema of table 123 and sends a Relation message that carries `names`
    (flag bit 0 = identity for those in ident_flags). Returns (error code, replication mask, identity mask)."""
    t = _fresh(kind)
    t.schema_put(123, 0, cols, name="test_table")
    t.table_state(123, abi.TS_READY)
    s = W.Stream()
    s.add(W.begin(0x2000))
    s.add(W.relation(123, "public", "test_table", replident, [(1 if n in ident_flags else 0, n, 23, -1) for n in names]))
    s.add(W.commit(0x2000, 0x2008))
    b = _decode(t, s)
    code = _err(b)
    if code:
        t.close()
        return code, None, None
    hb = _host(b)
    rel = [i for i in range(hb.n_events) if hb.kind[i] == ord("R")]
    assert len(rel) == 1
    slot = hb.slots[int(hb.schema_slot[rel[0]])]
    rm, im = [0] * len(cols), [0] * len(cols)
    for c in slot.cols:
        rm[c.stored_index] = 1
        im[c.stored_index] = 1 if c.identity else 0
    t.close()
    return 0, rm, im


def _run_mask_kats(kind):
    for name, names, want in K.SCHEMA_RS_REPLICATION_MASK:
        code, rm, im = _relation_masks(kind, K.TEST_TABLE, names, ["id"])
        if isinstance(want, tuple):
            assert code == abi.E_UNKNOWN_COLUMNS, name     # SchemaError::UnknownReplicatedColumns -> CorruptedTableSchema (error.rs:1092-1104)
        else:
            assert code == 0 and rm == want, (name, code, rm)
    # identity from the Relation message: flagged columns, or every replicated column under REPLICA IDENTITY FULL
    # (codec/event.rs:352-396), classified as schema.rs:686-721 does
    for name, rm_want, im_want, type_want in K.SCHEMA_RS_IDENTITY_TYPE:
        im_want = schema.identity_mask_default(K.TEST_TABLE, rm_want) if im_want is None else im_want
        flagged = [c[0] for c, i in zip(K.TEST_TABLE, im_want) if i]
        code, rm, im = _relation_masks(kind, K.TEST_TABLE, [c[0] for c in K.TEST_TABLE], flagged)
        assert code == 0 and rm == rm_want and im == im_want, (name, rm, im)
        assert schema.infer_identity_type(K.TEST_TABLE, rm, im) == type_want, name
    code, rm, im = _relation_masks(kind, K.TEST_TABLE, ["id", "age"], [], replident="f")
    assert code == 0 and rm == [1, 0, 1] and im == [1, 0, 1]
    # build_identity_mask KATs with the replication mask as the relation's column list and the identity columns flagged
    for name, cols, rm_in, pk, mode, idx, want_mask, want_type in K.EVENT_RS_BUILD_IDENTITY:
        expect = schema.identity_mask_from_metadata(cols, rm_in, pk, mode, idx)
        names = [c[0] for c, r in zip(cols, rm_in) if r]
        flagged = [c[0] for c, i in zip(cols, expect) if i]
        code, rm, im = _relation_masks(kind, cols, names, flagged, replident="f" if mode == "f" else "d")
        assert code == 0 and rm == rm_in and im == expect, (name, rm, im)


def _ddl_json(table_id, cols):
    return json.dumps({"command_tag": "ALTER TABLE", "nspname": "public", "relname": "test_table", "oid": table_id,
                       "identity": {"primary_key_attnums": [1], "relreplident": "d", "replica_identity_index_attnums": []},
                       "columns": [{"attname": n, "atttypid": o, "atttypmod": -1, "attnum": i + 1, "attnotnull": not nl,
                                    "default_expression": None} for i, (n, o, nl, _) in enumerate(cols)]})


def _run_cache_kats(kind):
    names = {1: "WaitingForRelation", 2: "Ready"}
    for name, steps, want in K.TABLE_CACHE_RS:
        t = _fresh(kind)
        for tid in (123, 456):
            t.schema_put(tid, 10, K.TEST_TABLE, name="test_table")   # TableSchema::with_snapshot_id(.., SnapshotId 10)
            t.table_state(tid, abi.TS_READY)
        lsn = 0x100
        for st in steps:
            if st[0] == "ready":      # note_ready(create_test_schema()): replication mask [1, 0, 1], identity mask [1, 0, 1]
                assert t.table_ready(st[1], 10, [1, 0, 1], [1, 0, 1]) >= 0
            elif st[0] == "forget":
                t.table_forget(st[1])
            else:                     # note_waiting_for_relation(table, snapshot): what a DDL message at wal_start = snapshot does
                s = W.Stream()
                lsn += 0x100
                s.add(W.begin(lsn), lsn=1)
                s.add(W.message("supabase_etl_ddl", _ddl_json(st[1], K.TEST_TABLE)), lsn=st[2])
                s.add(W.commit(lsn, lsn + 8), lsn=st[2] + 1)
                assert _err(_decode(t, s)) == 0, name
        for tid, exp in want.items():
            got = t.cache_state(tid)
            if exp is None:
                assert got is None, (name, got)
            else:
                assert got is not None and (names[got[0]], got[1]) == exp, (name, tid, got)
                assert (got[2] >= 0) == (exp[0] == "Ready")     # a waiting entry exposes its snapshot without a schema (:256)
        if hasattr(t, "cache_tables"):
            assert t.cache_tables() == sorted(k for k, v in want.items() if v is not None), name
        # a row of a table in WaitingForRelation / without an entry cannot be decoded (apply.rs:3709-3732)
        for tid, exp in want.items():
            s = W.Stream()
            s.add(W.begin(0x9000), lsn=0x8000)
            s.add(W.insert(tid, ["1", "7"] if exp and exp[0] == "Ready" else ["1", "x", "7"]), lsn=0x8008)
            s.add(W.commit(0x9000, 0x9008), lsn=0x8010)
            code = _err(_decode(t, s))
            assert code == (0 if exp and exp[0] == "Ready" else abi.E_WAITING_RELATION if exp else abi.E_MISSING_SHARED_STATE), (name, tid, code)
            t.reset_stream_state()
        t.close()


def test_oracle_masks_from_relation_messages():
    _run_mask_kats("oracle")


def test_oracle_shared_table_cache():
    _run_cache_kats("oracle")


@pytest.mark.gpu
def test_device_masks_from_relation_messages():
    _run_mask_kats("device")


@pytest.mark.gpu
def test_device_shared_table_cache():
    _run_cache_kats("device")
