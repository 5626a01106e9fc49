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
from tests.test_gpu_rowbinary import NUMERICS, RB_COLS, TIMETZS, _both, _row, _stream

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
    ok_numerics = [t for t in NUMERICS if t not in ("1e-40", "-7e-100")]      # more than 38 decimal places: BigQuery's validation refuses them (below)
    rows += [_row(id=str(10 + i), s="y" * (i * 13 % 200), t=f"01:02:{i % 60:02}.{i:06}", n=ok_numerics[i % len(ok_numerics)], tz=TIMETZS[i % len(TIMETZS)])
             for i in range(130)]
    msgs = [W.insert(42, r) for r in rows] + [W.update(42, rows[1]), W.delete(42, old=rows[0])]
    buf, offs = _stream(msgs)
    hb, b, d = _both(SC.simple_table(RB_COLS), buf, offs)
    assert _check(hb, b) == len(rows)
    b.close(); d.close()


def test_host_only_classes_and_deferred_cells():
    buf, offs = _stream([W.insert(42, SC.alltypes_row())])
    hb, b, d = _both(SC.simple_table(SC.ALLTYPES), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and r.view.host_column == [c[0] for c in SC.ALLTYPES].index("j")
    r.close(); b.close(); d.close()
    cols = [("id", SC.INT8, False, 1), ("x", SC.FLOAT8, True, 0)]
    buf, offs = _stream([W.insert(42, ["1", "1.5"]), W.insert(42, ["2", "50537618.817359292015891086651596749e82"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    r = b.protobuf(0)
    assert r.status == abi.RB_NEEDS_HOST and (int(r.view.host_event), r.view.host_column) == (2, 1)
    r.close(); b.close(); d.close()


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_synthetic_stream(mk):
    w = mk()
    buf, offs = w.fill((128 << 10) if os.environ.get("ETLG_SIMT_RUN") == "1" else (1 << 20))
    hb, b, d = _both(w.register, buf, offs)
    assert _check(hb, b) > 100
    b.close(); d.close()


def test_numeric_with_more_than_38_decimal_places_fails_like_the_reference():
    """validate_numeric_for_bigquery (bigquery/validation.rs:20-35; its own cases :213-229: scale 38 passes, 39 fails) behind
    BigQueryTableRow::try_from_tagged_cells (encoding.rs:37-45): kind UnsupportedValueInDestination, description 'Cell validation
    failed for BigQuery compatibility', detail naming the cell; the first failing row in event order."""
    from etl_amd.decoder import EtlError
    from oracle import protobuf as PB
    cols = [("id", SC.INT8, False, 1), ("v", SC.NUMERIC, True, 0)]
    ok = "0.00000000000000000000000000000000000001"
    bad = "0.000000000000000000000000000000000000001"
    buf, offs = _stream([W.insert(42, ["1", "123.456"]), W.insert(42, ["2", ok])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    assert _check(hb, b) == 2
    b.close(); d.close()
    buf, offs = _stream([W.insert(42, ["1", ok]), W.insert(42, ["2", bad]), W.insert(42, ["3", bad]), W.insert(42, ["4", "NaN"])])
    hb, b, d = _both(SC.simple_table(cols), buf, offs)
    with pytest.raises(PB.UnsupportedValueInDestination) as oi:
        PB.insert_rows(hb.materialize(), 0)
    assert str(oi.value) == "Cell at index 1 failed validation"
    with pytest.raises(EtlError) as ei:
        b.protobuf(0)
    assert ei.value.kind == abi.UnsupportedValueInDestination and ei.value.description == "Cell validation failed for BigQuery compatibility"
    assert ei.value.detail == "Cell at index 1 failed validation" and ei.value.frame_index == 2
    b.close(); d.close()
