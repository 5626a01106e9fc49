"""The fixed-width plan of the fused kernel (variant ETLG_FIXED_TILE, etl_amd/csrc/fixed_tile.hip.h) against the
oracle, through the C ABI: tiles that conform (Begin / Commit / Insert into a Ready fixed-width table) take
schema-constant sizing, every other tile of the same launch takes the generic body, and the arena must not show
the seam. On a library built without the variant the same cases run through the generic body alone (still a
parity test); ETLG_EXPECT_FIXED_TILE=1 (set by the run that loads a variant build) additionally demands that the
plan was really taken."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from etl_amd import synth

pytestmark = pytest.mark.gpu

EXPECT = os.environ.get("ETLG_EXPECT_FIXED_TILE") == "1"
_KNOBS = ("ETLG_FUSED_KERNEL", "ETLG_FORCE_MULTIPASS", "ETLG_FUSED_BLK", "ETLG_FUSED_DBG")


@pytest.fixture(params=["fused256", "fused64"])
def fused(request):
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "0" if request.param == "fused256" else "1"
    os.environ["ETLG_FUSED_DBG"] = "64"   # k_fused only: count the tiles that took the plan (DevResult.dbg_t[11])
    yield 256 if request.param == "fused256" else 64
    for k in _KNOBS:
        os.environ.pop(k, None)
        if saved[k] is not None:
            os.environ[k] = saved[k]


def _pair(w):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    d, o = Decoder(0), oracle.Oracle()
    w.register(d, ready=True)
    w.register(o, ready=True)
    return d, o


def _plan_tiles(d):
    out = (C.c_ulonglong * 12)()
    d.L.etlg_ctx_debug_times(d.h, out)
    return int(out[11])


def _agree(d, o, buf, offs):
    gb, rb = d.decode(buf, offs), o.decode(buf, offs)
    e = gb.error
    got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
    assert (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame) == got, got
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:4]
    return rb.err_code


def test_conforming_stream_takes_the_plan_and_matches(fused):
    """cfg2 (5 x int4 INSERT, 1000 rows per transaction): every tile conforms; transactions span tiles and
    batches, so the carried transaction state crosses plan tiles in both directions."""
    w = synth.cfg2()
    d, o = _pair(w)
    for nbytes in (300 << 10, 64 << 10, 1 << 20):
        buf, offs = w.fill(nbytes)
        assert _agree(d, o, buf, offs) == 0
        ntiles = (len(offs) - 1 + fused - 1) // fused
        if EXPECT:
            assert _plan_tiles(d) == ntiles
        assert d.debug_paths()["redone"] == 0
    d.close()


def _concat(parts):
    bufs, offs, base = [], [np.zeros(1, dtype=np.uint32)], 0
    for b, o in parts:
        bufs.append(b)
        offs.append(o[1:].astype(np.uint32) + np.uint32(base))
        base += len(b)
    return np.concatenate(bufs), np.concatenate(offs)


def test_plan_and_generic_tiles_mix_in_one_launch(fused):
    """Runs of whole transactions on a fixed-width table alternate with runs on a table with TEXT / NUMERIC columns
    (inserts, updates, deletes): tiles inside the first kind of run conform, the others do not, and both kinds
    publish to the same look-back descriptors."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    wa = synth.Workload([synth.table_fixed()], 0xF1D0001, rows_per_txn=700, name="fixed_runs")
    wb = synth.Workload([synth.table_mixed()], 0xF1D0002, rows_per_txn=150, mix=(80, 15, 5), upd_key=10, start_lsn=0x9000000,
                        name="mixed_runs")
    d, o = Decoder(0), oracle.Oracle()
    for w in (wa, wb):
        w.register(d, ready=True)
        w.register(o, ready=True)
    seen = 0
    for sizes in ((200, 60, 150, 40, 100), (90, 30, 250)):
        buf, offs = _concat([(wa if i % 2 == 0 else wb).fill(kb << 10) for i, kb in enumerate(sizes)])
        assert _agree(d, o, buf, offs) == 0
        ntiles = (len(offs) - 1 + fused - 1) // fused
        seen += _plan_tiles(d)
        assert _plan_tiles(d) < ntiles          # the var-len table's tiles are generic
    if EXPECT:
        assert seen > 0
    assert d.debug_paths()["redone"] == 0
    d.close()


def _edit(buf, offs, f, fn):
    b = bytearray(buf.tobytes())
    fn(b, int(offs[f]), int(offs[f + 1]))
    return np.frombuffer(bytes(b), dtype=np.uint8)


@pytest.mark.parametrize("what", ["bad_digit", "null_required", "tuple_width", "update_tag", "commit_lsn", "keepalive_tag",
                                  "insert_outside_txn", "int_overflow", "unchanged_toast", "binary_cell"])
def test_errors_and_odd_frames_inside_conforming_tiles(what, fused):
    """One frame of an otherwise conforming cfg2 batch is damaged: errors that surface while decoding are recorded
    by the plan itself, shapes it does not cover send the tile to the generic body — either way the oracle's
    error (code, kind, text, frame) and every arena byte before it."""
    w = synth.cfg2()
    d, o = _pair(w)
    buf, offs = w.fill(200 << 10)
    tags = [buf[int(offs[i]) + 30] for i in range(len(offs) - 1)]
    ins = [i for i, t in enumerate(tags) if t == ord("I")]
    f = ins[len(ins) // 2 + 37]

    def cell0(b, lo):  # first cell of an Insert: 'I' rel(4) 'N' ncols(2) then 't' len(4) text
        return lo + 38

    if what == "bad_digit":
        buf = _edit(buf, offs, f, lambda b, lo, hi: b.__setitem__(hi - 1, ord("x")))
    elif what == "null_required":
        def fn(b, lo, hi):   # first cell becomes NULL: shrink is not possible in place, so turn the LAST cell's tag instead
            c = cell0(b, lo)
            for _ in range(4):
                c += 5 + struct.unpack_from(">I", b, c + 1)[0]
            n = struct.unpack_from(">I", b, c + 1)[0]
            # 'n' followed by n + 4 bytes of trailing garbage (the wire parser ignores bytes after the tuple)
            b[c] = ord("n")
        buf = _edit(buf, offs, f, fn)
    elif what == "tuple_width":
        buf = _edit(buf, offs, f, lambda b, lo, hi: struct.pack_into(">h", b, lo + 36, 4))
    elif what == "update_tag":
        buf = _edit(buf, offs, f, lambda b, lo, hi: b.__setitem__(lo + 30, ord("U")))
    elif what == "keepalive_tag":
        buf = _edit(buf, offs, f, lambda b, lo, hi: b.__setitem__(lo + 5, ord("k")))
    elif what == "commit_lsn":
        c = [i for i, t in enumerate(tags) if t == ord("C")][1]
        buf = _edit(buf, offs, c, lambda b, lo, hi: b.__setitem__(lo + 31 + 8, b[lo + 31 + 8] ^ 1))
    elif what == "insert_outside_txn":
        bframes = [i for i, t in enumerate(tags) if t == ord("B")]
        bf = bframes[1]   # this Begin becomes an Origin message: the rows after it arrive outside a transaction
        buf = _edit(buf, offs, bf, lambda b, lo, hi: b.__setitem__(lo + 30, ord("O")))
    elif what == "int_overflow":
        def fn(b, lo, hi):
            c = cell0(b, lo)
            n = struct.unpack_from(">I", b, c + 1)[0]
            b[c + 5:c + 5 + n] = b"9" * n if n >= 10 else b[c + 5:c + 5 + n]
        buf = _edit(buf, offs, f, fn)
    elif what == "unchanged_toast":
        buf = _edit(buf, offs, f, lambda b, lo, hi: b.__setitem__(cell0(b, lo), ord("u")))
    elif what == "binary_cell":
        buf = _edit(buf, offs, f, lambda b, lo, hi: b.__setitem__(cell0(b, lo), ord("b")))
    err = _agree(d, o, buf, offs)
    print("fixed-plan damage", what, "-> oracle error code", err)
    if what not in ("keepalive_tag", "update_tag", "int_overflow"):   # legal shapes (keepalive; Update without an old tuple); a short value cannot overflow
        assert err != 0
    d.close()


def test_many_registered_tables(fused):
    """~6 KiB of side tables (30 registered tables of 12 columns): more than four dwords per lane, so the kernel head's
    side-table copy needs its remainder loop on top of the in-register part (ETLG_EARLY_SPAN variant), and the generic
    head its full loops; the stream itself only touches two of the tables."""
    from etl_amd import abi
    from etl_amd.decoder import Decoder
    from oracle import oracle
    wa = synth.Workload([synth.table_fixed()], 0xF1D0003, rows_per_txn=300, name="fixed_runs")
    wb = synth.Workload([synth.table_mixed()], 0xF1D0004, rows_per_txn=100, mix=(70, 20, 10), upd_key=10, start_lsn=0x9000000,
                        name="mixed_runs")
    d, o = Decoder(0), oracle.Oracle()
    extra = synth.table_mixed()
    for t in (d, o):
        for k in range(30):   # ids on both sides of the two real tables
            rel = 100 + 7 * k if k % 2 else 900000 + 11 * k
            t.schema_put(rel, 0, wb.schema_cols(extra), name="extra_%d" % k)
            t.table_state(rel, abi.TS_READY)
            n = len(extra["cols"])
            t.table_ready(rel, 0, [1] * n, [1 if c["pk"] else 0 for c in extra["cols"]])
    for w in (wa, wb):
        w.register(d, ready=True)
        w.register(o, ready=True)
    buf, offs = _concat([wa.fill(120 << 10), wb.fill(50 << 10), wa.fill(90 << 10)])
    assert _agree(d, o, buf, offs) == 0
    assert d.debug_paths()["redone"] == 0
    d.close()
