"""Device-side BigQuery protobuf rows (etlg_batch_protobuf, etl_amd/csrc/columns.hip) byte for byte against oracle/protobuf.py
(restatement of crates/etl-destinations/src/bigquery/encoding.rs:120-190 on the protobuf wire format): every class the device
encodes incl. negative integers (10-byte varints), NULLs (absent fields), dates / times / timestamps as chrono strings, the
trailing UPSERT / sequence-key fields; updates and deletes counted for the host; host-only classes reported."""
import os

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC
from tests.test_gpu_rowbinary import RB_COLS, _both, _row, _stream

pytestmark = pytest.mark.gpu


def _check(hb, b):
    from oracle import protobuf as PB
    rows, idx, host = PB.insert_rows(hb.materialize(), 0)
    r = b.protobuf(0)
    assert r.status == abi.RB_OK and r.n_rows == len(rows) and int(r.view.n_host_rows) == host
    assert np.array_equal(r.row_event(), np.array(idx, dtype=np.uint64))
    offs = r.row_offsets()
    assert np.array_equal(np.diff(offs), np.array([len(x) for x in rows], dtype=np.int64))
    assert r.bytes().tobytes() == b"".join(rows)
    r.close()
    return len(rows)


def test_every_encodable_class():
    names = [c[0] for c in RB_COLS]
    rows = [_row(), _row(id="2", b="f", i2="-7", i4="-2147483648", o="4294967295", d="0001-01-01", t="00:00:00",
                         ts="1969-12-31 23:59:59.5", tstz="1969-12-31 23:59:59.5+00", f8="1e300", f4="-0.5", s="", by="\\x"),
            _row(id="-9223372036854775808", d="9999-12-31", t="23:59:59.12", ts="2026-01-02 03:04:05", s="x" * 300, by="\\x" + "ab" * 200),
            [("4" if n == "id" else W.NULL) for n in names]]
    rows += [_row(id=str(10 + i), s="y" * (i * 13 % 200), t=f"01:02:{i % 60:02}.{i:06}") for i in range(130)]
    msgs = [W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
    assert _check(hb, b) == len(rows)
    b.close(); d.close()


def test_host_only_classes_and_deferred_cells():
    buf, offs = _stream([W.insert(42, SC.alltypes_row())])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and r.view.host_column == [c[0] for c in SC.ALLTYPES].index("n")
    r.close(); b.close(); d.close()
    cols = [("id", SC.INT8, False, 1), ("x", SC.FLOAT8, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "1.5"]), W.insert(42, ["2", "50537618.817359292015891086651596749e82"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close(); b.close(); d.close()


def test_synthetic_stream():
    w = synth.cfg2()
    buf, offs = w.fill((128 << 10) if os.environ.get("ETLG_SIMT_RUN") == "1" else (1 << 20))
    hb, b, d = _both(w.register, buf, offs)
    assert _check(hb, b) > 100
    b.close(); d.close()
