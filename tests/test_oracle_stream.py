"""Stream-level behaviour of the oracle: transaction state machine, ordinals,
ownership filter, Relation / DDL control flow, fail-fast errors
(crates/etl/src/replication/apply.rs:2026-2617)."""
import pytest

from etl_amd import abi
from oracle import oracle
from tests import scenarios as SC

ALL = {s.name: s for s in SC.all_scenarios()}


def run(name):
    return SC.replay(oracle.Oracle(), ALL[name])


@pytest.mark.parametrize("name", sorted(ALL))
def test_scenario_replays(name):
    res = run(name)
    for code, kind, desc, frame, hb in res:
        if code:
            assert frame >= 0 and desc
            assert hb.n_frames == frame
        else:
            assert frame == -1


EXPECT_ERR = {
    # name: (code, frame index)
    "bad_bool": (abi.E_BOOL, 2), "bad_int_overflow": (abi.E_INT, 2), "bad_numeric": (abi.E_NUMERIC, 2),
    "bad_bytea_odd": (abi.E_BYTEA, 2), "bad_tstz_no_offset": (abi.E_DATETIME, 2), "bad_uuid": (abi.E_UUID, 2),
    "bad_utf8_text": (abi.E_UTF8, 2), "bad_utf8_deferred_json": (abi.E_UTF8, 2),
    "err_required_null": (abi.E_REQUIRED_NULL, 1), "err_width": (abi.E_TUPLE_WIDTH, 1),
    "err_full_row_toast": (abi.E_FULL_ROW_MISSING, 1), "err_binary_cell": (abi.E_BINARY_FORMAT, 1),
    "err_key_shape": (abi.E_KEY_SHAPE, 1), "err_key_no_identity": (abi.E_KEY_MISSING_COLS, 1),
    "err_key_toast": (abi.E_KEY_MISSING_VALUE, 1), "err_old_before_new": (abi.E_INT, 1),
    "err_wire_corrupt_tuple": (abi.E_WIRE, 1), "err_wire_unknown_tag": (abi.E_WIRE, 1),
    "err_wire_negative_len": (abi.E_WIRE, 1), "err_sidecar_disagrees": (abi.E_WIRE, 1),
    "err_insert_outside_txn": (abi.E_TXN_STATE, 0), "err_commit_without_begin": (abi.E_TXN_STATE, 0),
    "err_commit_lsn_mismatch": (abi.E_COMMIT_LSN, 2), "err_truncate_outside_txn": (abi.E_TXN_STATE, 2),
    "err_missing_shared_state": (abi.E_MISSING_SHARED_STATE, 1),
    "err_relation_unknown_column": (abi.E_UNKNOWN_COLUMNS, 1), "err_relation_no_schema": (abi.E_SCHEMA_NOT_FOUND, 1),
    "err_relation_outside_txn": (abi.E_TXN_STATE, 0), "err_row_while_waiting_for_relation": (abi.E_WAITING_RELATION, 2),
    "err_ddl_bad_json": (abi.E_DDL_PARSE, 1), "err_ddl_duplicate_field": (abi.E_DDL_PARSE, 1),
    "err_ddl_outside_txn": (abi.E_TXN_STATE, 0), "err_after_ddl_rolls_back_later_control": (abi.E_INT, 1),
    "no_sidecar_trailing_garbage": (abi.E_WIRE, 3),
}


@pytest.mark.parametrize("name", sorted(EXPECT_ERR))
def test_expected_errors(name):
    code, frame = EXPECT_ERR[name]
    res = run(name)
    got = res[-1] if name != "txn_spanning_batches" else res[0]
    assert (got[0], got[3]) == (code, frame)


def test_ordinals_and_noise():
    (code, _, _, _, hb), = run("ordinals_and_noise")
    assert code == 0
    ev = hb.materialize()
    assert [e["kind"] for e in ev] == ["B", "I", "I", "C", "B", "I", "C"]
    # the filtered insert on table 99 consumed ordinal 2; O / Y / M / k consumed none
    assert [e["tx_ordinal"] for e in ev] == [0, 1, 3, 4, 0, 1, 2]
    assert ev[1]["commit_lsn"] == 0x2000 and ev[5]["commit_lsn"] == 0x3000
    assert ev[5]["row"] == [("I64", 3), ("Null",)]


def test_ownership_sync_done_boundary():
    (_, _, _, _, hb), = run("own_sync_done_before")
    assert [chr(k) for k in hb.kind] == ["B", "C"]
    (_, _, _, _, hb), = run("own_sync_done_at")
    assert [chr(k) for k in hb.kind] == ["B", "I", "C"]


def test_table_sync_worker_owns_only_its_table():
    (code, _, _, _, hb), = run("table_sync_worker")
    assert code == 0
    ev = hb.materialize()
    assert [e["kind"] for e in ev] == ["B", "I", "T", "C"]
    assert ev[2]["tables"] == [(42, 0)] and ev[2]["tx_ordinal"] == 3


def test_relation_builds_masks_by_name():
    (code, _, _, _, hb), = run("relation_subset_of_columns")
    assert code == 0
    ev = hb.materialize()
    assert [e["kind"] for e in ev] == ["B", "R", "I", "C"]
    slot = hb.slots[ev[1]["schema_slot"]]
    assert [(c.stored_index, c.identity) for c in slot.cols] == [(0, 1)]
    assert ev[2]["row"] == [("I64", 7)]


def test_ddl_switches_schema_after_relation():
    (code, _, _, _, hb), = run("ddl_then_relation_then_rows")
    assert code == 0
    ev = hb.materialize()
    assert [e["kind"] for e in ev] == ["B", "I", "R", "I", "C"]
    assert ev[1]["row"] == [("I64", 1), ("String", b"old")]
    assert ev[3]["row"] == [("I64", 2), ("String", b"new"), ("I32", 77)]
    assert hb.slots[ev[3]["schema_slot"]].snapshot_lsn == ev[2]["start_lsn"] - 8  # snapshot id = the DDL message's wal_start


def test_txn_spanning_batches_carries_state():
    res = run("txn_spanning_batches")
    assert all(r[0] == 0 for r in res)
    ords = [int(x) for r in res for x in r[4].tx_ordinal]
    assert ords == list(range(702))
    assert {int(x) for r in res for x in r[4].commit_lsn} == {0x9000}


def test_synth_cfg5_has_ddl_epochs():
    res = run("synth_cfg5_ddl_3tables")
    assert all(r[0] == 0 for r in res)
    kinds = "".join(chr(k) for r in res for k in r[4].kind)
    assert kinds.count("R") >= 3 and "U" in kinds and "D" in kinds
