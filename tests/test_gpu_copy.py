"""Table-copy rows on the device (copy.hip + the decode kernels in copy mode) against the oracle's
restatement of table_row.rs, byte for byte on the arena, through etlg_copy_decode."""
import random

import numpy as np
import pytest

from etl_amd import abi
from tests import scenarios as SC
from tests import test_oracle_copy as K
from tests.test_gpu_parity import path, PATHS  # noqa: F401  (the kernel-path fixture)

pytestmark = pytest.mark.gpu


def both(cols, rows):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    for t in (o, d):
        t.schema_put(42, 0, cols)
    so = o.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    sd = d.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    assert so == sd >= 0
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    rb = o.copy_decode(so, buf, offs)
    gb = d.copy_decode(sd, buf, offs)
    return o, d, rb, gb


def assert_same(rb, gb):
    e = gb.error
    got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
    assert (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame) == got
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]


@pytest.mark.parametrize("cols,row,want", K.OK)
def test_copy_kat_rows(cols, row, want, path):
    o, d, rb, gb = both(cols, [bytes(row)])
    assert_same(rb, gb)
    assert gb.host().materialize()[0]["row"] == want
    d.close()


@pytest.mark.parametrize("row,want", K.ESCAPES)
def test_copy_kat_escapes(row, want):
    o, d, rb, gb = both(K.single(K.TEXT), [bytes(row)])
    assert_same(rb, gb)
    d.close()


@pytest.mark.parametrize("cols,row,code", K.ERR)
def test_copy_kat_errors(cols, row, code, path):
    # a good row first, then the bad one: the batch keeps the good row and fails at row 1
    good = {len(K.BASIC): b"1\tx\tt\n", 1: (b"7\n" if cols[0][1] == K.INT4 else b"ok\n")}[len(cols)]
    o, d, rb, gb = both(cols, [good, bytes(row)])
    assert rb.err_code == code and rb.err_frame == 1
    assert_same(rb, gb)
    assert gb.view().n_events == 1
    d.close()


from etl_amd.synth import copy_rows as _gen_rows, COPY_COLS as GEN_COLS   # the generator bench.py's copy leg uses


def test_copy_generated_rows(path):
    rows = _gen_rows(5000, 11)
    o, d, rb, gb = both(GEN_COLS, rows)
    assert rb.err_code == 0
    assert_same(rb, gb)
    v = gb.view()
    assert v.n_events == len(rows) and v.payload_bytes[0] == sum(len(r) for r in rows)
    assert d.debug_paths()["redone"] == 0
    # stream state of the context is untouched by a copy batch: a normal transaction still decodes
    from tests import pgwire as W
    s = SC.txn([W.insert(42, [r.decode() for r in [b"1", b"2", b"t", b"1", b"x", b"y", b"2024-01-01 00:00:00+00",
                                                   b"123e4567-e89b-12d3-a456-426614174000", b"1.5", b"\\x00"]])])
    for t in (o, d):
        t.table_state(42, abi.TS_READY)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    r2, g2 = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert r2.err_code == 0 and g2.rc == 0 and not r2.host_batch().diff(g2.host())
    d.close()


def test_copy_error_in_the_middle(path):
    rows = _gen_rows(600, 5)
    rows[417] = rows[417].replace(b"\t", b"\t\t", 1)       # one field too many (and an empty int4) in row 417
    o, d, rb, gb = both(GEN_COLS, rows)
    assert rb.err_frame == 417 and rb.err_code != 0
    assert_same(rb, gb)
    assert gb.view().n_events == 417
    d.close()


def test_copy_device_resident_rows():
    import torch
    rows = _gen_rows(2000, 3)
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    for t in (o, d):
        t.schema_put(42, 0, GEN_COLS)
    so = o.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
    sd = d.table_ready(42, 0, [1] * 10, [1] + [0] * 9)
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    tb = torch.from_numpy(buf.copy()).cuda()
    to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
    torch.cuda.synchronize()
    gb = d.copy_decode_device(sd, tb.data_ptr(), tb.numel(), to.data_ptr(), len(rows))
    assert gb.rc == 0 and gb.view().on_device == 1
    assert not o.copy_decode(so, buf, offs).host_batch().diff(gb.host())
    d.close()
