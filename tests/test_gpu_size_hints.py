"""etlg_batch_size_hints (etl_amd/csrc/columns.hip) against oracle/size_hint.py, the restatement of Event::size_hint
(crates/etl/src/event.rs:295-320) and estimate_table_row_allocated_bytes (crates/etl/src/data/table_row.rs:248-299):
every event kind, full / key / partial rows, every cell class, and the events whose estimate the host must finish."""
import os

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC

pytestmark = pytest.mark.gpu


def _model():
    from oracle import size_hint as SH
    m = abi.SizeModel()
    for k, v in SH.MODEL.items():
        setattr(m, k, v)
    return m, SH


def _check(prime, buf, offs):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    prime(o)
    prime(d)
    rb = o.decode(buf, offs)
    assert rb.err_code == 0, rb.err_desc
    b = d.decode(buf, offs, flags=abi.F_OUTPUT_ON_DEVICE)
    assert b.rc == 0, b.error
    hb = rb.host_batch()
    m, SH = _model()
    want = np.array([SH.event_hint(e, hb.slots, SH.MODEL) for e in hb.materialize()], dtype=np.uint64)
    got = b.size_hints(m)
    bad = np.flatnonzero(want != got)
    assert not len(bad), [(int(i), chr(hb.kind[i]), int(want[i]), int(got[i])) for i in bad[:5]]
    b.close(); d.close()
    return want


def _stream(msgs):
    s = SC.txn(msgs)
    return np.frombuffer(s.bytes(), dtype=np.uint8), s.offsets


def test_every_event_kind_and_row_shape():
    N, U = W.NULL, W.TOAST
    cols = [("id", SC.INT8, False, 1), ("a", SC.TEXT, True, 0), ("k", SC.TEXT, False, 1), ("n", SC.NUMERIC, True, 0), ("by", SC.BYTEA, True, 0)]
    r = ["1", "alice", "key", "12345.678900", "\\x0102ff"]
    msgs = [W.insert(42, r), W.insert(42, ["2", N, "", "NaN", N]), W.insert(42, ["3", "x" * 500, "k" * 33, "0", "\\x"]),
            W.update(42, ["1", "bob", "key", "1e10", N]),
            W.update(42, ["1", "bob", "key2", "7", N], key=["1", N, "key", N, N]),
            W.update(42, ["1", "carol", "key2", "-0.5", "\\xff"], old=r),
            W.update(42, ["1", U, "key2", "1", N]),                       # unchanged toast, no old image: partial
            W.delete(42, key=["1", N, "key2", N, N]), W.delete(42, old=r),
            W.truncate([42], 0)]
    buf, offs = _stream(msgs)
    want = _check(SC.simple_table(cols), buf, offs)
    assert int((want >> np.uint64(63)).sum()) == 1                         # only the partial update


def test_every_class_and_the_cells_the_host_sizes():
    rows = [SC.alltypes_row(), SC.alltypes_row(id="2", j="[1,2]", arr="{}", n="NaN"),
            [("3" if c[0] == "id" else W.NULL) for c in SC.ALLTYPES],
            SC.alltypes_row(id="4", j=W.NULL, arr=W.NULL)]
    buf, offs = _stream([W.insert(42, r) for r in rows])
    want = _check(SC.simple_table(SC.ALLTYPES), buf, offs)
    inc = (want >> np.uint64(63)).astype(bool)
    assert list(inc[1:5]) == [True, True, False, False]                    # json / array cells need the parsed value


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_synthetic_streams(mk):
    w = mk()
    buf, offs = w.fill((128 << 10) if os.environ.get("ETLG_SIMT_RUN") == "1" else (1 << 20))
    want = _check(w.register, buf, offs)
    assert len(want) > 300
    if mk is synth.cfg2:
        assert not (want >> np.uint64(63)).any()
