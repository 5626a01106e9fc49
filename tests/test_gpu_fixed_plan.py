"""Fixed-width decode plans against the oracle, through the C ABI.

k_plan (etl_amd/csrc/plan.hip): whole batches of Begin / Commit / Insert frames into Ready tables of bool / integer
columns, one wave per tile; anything else makes the kernel give the batch up and the generic kernel decodes it. A tile's
prefix comes from the sidecar pre-pass (k_plan_pre: frames priced by their length) when every planned table has the same row
size, from the kernel's own look-back otherwise (ETLG_PLAN_PRE=0 forces that).
The plan of k_fused (etl_amd/csrc/fixed_tile.hip.h): tiles that conform take schema-constant sizing, every other tile
of the same launch takes the generic body, and the arena must not show the seam.
Every case runs on both (`fused` fixture: k_fused with 256 / 64 frames per tile forced, k_plan forced, k_plan reading the
input in place instead of its LDS window), with the demand that the plan was really taken where a stream conforms."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from etl_amd import synth

pytestmark = pytest.mark.gpu

EXPECT = True
_KNOBS = ("ETLG_FUSED_KERNEL", "ETLG_FORCE_MULTIPASS", "ETLG_FUSED_DBG", "ETLG_PLAN", "ETLG_PLAN_DBG", "ETLG_PLAN_PRE")


@pytest.fixture(params=["fused256", "fused64", "plan", "plan_one", "plan_inplace", "plan_lookback", "plan_pre_two", "plan_pre_inplace"])
def fused(request):
    """Yields the frames per tile of the forced k_fused instance, or 0 when k_plan is forced."""
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    if request.param.startswith("plan"):
        os.environ["ETLG_FUSED_KERNEL"] = "3"
        # plan: the default (sidecar pre-pass, one tile per wave). The look-back kernels: plan_lookback (two tiles per wave, k_plan2),
        # plan_one (one tile per wave), plan_inplace (tiles read in place instead of through the LDS window)
        if request.param in ("plan_one", "plan_inplace", "plan_lookback"):
            os.environ["ETLG_PLAN_PRE"] = "0"
        if request.param in ("plan_inplace", "plan_pre_inplace"):
            os.environ["ETLG_PLAN_DBG"] = "1"
        if request.param == "plan_one":
            os.environ["ETLG_PLAN_DBG"] = "512"   # one tile per wave (k_plan) instead of two (k_plan2)
        if request.param == "plan_pre_two":
            os.environ["ETLG_PLAN_PRE"] = "2"     # the pre-pass in front of the two-tiles-per-wave kernel
    else:
        os.environ["ETLG_FUSED_KERNEL"] = "0" if request.param == "fused256" else "1"
        os.environ["ETLG_FUSED_DBG"] = "64"   # k_fused only: count the tiles that took the plan (DevResult.dbg_t[11])
    yield {"fused256": 256, "fused64": 64}.get(request.param, 0)
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
    for k, nbytes in enumerate((300 << 10, 64 << 10, 1 << 20)):
        buf, offs = w.fill(nbytes)
        assert _agree(d, o, buf, offs) == 0
        if fused:
            assert _plan_tiles(d) == (len(offs) - 1 + fused - 1) // fused
        else:
            assert d.debug_paths()["plan"] == k + 1, d.debug_paths()
        assert d.debug_paths()["redone"] == 0 and d.debug_paths()["plan_redone"] == 0
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
        if fused:
            ntiles = (len(offs) - 1 + fused - 1) // fused
            seen += _plan_tiles(d)
            assert _plan_tiles(d) < ntiles          # the var-len table's tiles are generic
    if fused:
        assert seen > 0
    else:
        assert d.debug_paths()["plan"] == 0         # a table with TEXT columns is owned: k_plan is not even tried
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


# ---- k_plan's own cases -------------------------------------------------------------------------------------------------
from tests import pgwire as W   # noqa: E402
from tests import scenarios as SC   # noqa: E402

WIDE = [("id", SC.INT8, False, 1)] + [("c%d" % i, (SC.INT4, SC.INT8, SC.INT2, SC.BOOL, SC.OID)[i % 5], i % 3 == 0, 0) for i in range(1, 14)]
NARROW = [("k", SC.INT4, False, 1), ("v", SC.INT8, True, 0)]


def _two_tables(t):
    for rel, cols in ((42, WIDE), (43, NARROW)):
        t.schema_put(rel, 0, cols, name="t%d" % rel)
        t.table_state(rel, 1)
        t.table_ready(rel, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])


def _wide_row(rng, i):
    vals = [str(i)]
    for k in range(1, 14):
        oid = WIDE[k][1]
        if WIDE[k][2] and rng.random() < 0.3:
            vals.append(W.NULL)
        elif oid == SC.BOOL:
            vals.append(rng.choice("tf"))
        elif oid == SC.INT2:
            vals.append(str(rng.choice([-32768, 32767, 0, -1, 7, 12345])))
        elif oid == SC.INT8:
            vals.append(rng.choice(["-9223372036854775808", "9223372036854775807", "0", "+17", "-0", str(rng.randint(-10**18, 10**18))]))
        elif oid == SC.OID:
            vals.append(rng.choice(["0", "4294967295", "+12", str(rng.randint(0, 2**32 - 1))]))
        else:
            vals.append(rng.choice(["-2147483648", "2147483647", "+5", "-0", "0007", str(rng.randint(-2**31, 2**31 - 1))]))
    return vals


def _plan_stream(seed, ntx, rows, odd=None):
    import random
    rng = random.Random(seed)
    s = W.Stream()
    lsn = 0x5000
    n = 0
    for t in range(ntx):
        lsn += 0x100
        s.add(W.begin(lsn, ts=1000 + t, xid=70 + t))
        for r in range(rows):
            if rng.random() < 0.5:
                s.add(W.insert(42, _wide_row(rng, n)))
            else:
                s.add(W.insert(43, [str(n), W.NULL if rng.random() < 0.2 else str(rng.randint(-10**12, 10**12))]))
            if odd and n == odd[0]:
                s.add(odd[1])
            n += 1
        s.add(W.commit(lsn, lsn + 8, ts=2000 + t, flags=0))
    return s


@pytest.mark.parametrize("inplace", [False, True])
def test_k_plan_two_tables_wide_rows_nulls_and_boundary_values(inplace):
    """A 14-column table (60-byte rows: they wait for the look-back in the separate LDS region, not in the frame's own
    head) interleaved frame by frame with a 2-column one (waves hold both tables: one pass per table), NULLs in nullable
    columns, every integer width at its limits, explicit signs, leading zeros; transactions of 37 rows span tiles."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    if inplace:
        os.environ["ETLG_PLAN_DBG"] = "1"
    try:
        d, o = Decoder(0), oracle.Oracle()
        _two_tables(d); _two_tables(o)
        for seed in (1, 2):
            s = _plan_stream(seed, 40, 37)
            buf = np.frombuffer(s.bytes(), dtype=np.uint8)
            assert _agree(d, o, buf, s.offsets) == 0
        p = d.debug_paths()
        assert p["plan"] == 2 and p["plan_redone"] == 0 and p["redone"] == 0, p
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("what", ["long_leading_zeros", "null_in_required", "oid_minus_zero", "int2_overflow", "update_with_key", "update_toast", "delete_full",
                                  "delete_key_shape", "delete_key_toast", "truncate", "keepalive", "unknown_table", "lsn_top_bits"])
def test_k_plan_gives_up_and_the_generic_kernel_answers(what):
    """Shapes k_plan does not take. Legal ones (a value with twenty leading zeros, an Update, a keepalive ...) must come out
    exactly as the oracle has them, produced by the generic kernel ('plan_redone'); illegal ones must report the oracle's error."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    try:
        d, o = Decoder(0), oracle.Oracle()
        _two_tables(d); _two_tables(o)
        odd = {
            "long_leading_zeros": W.insert(43, ["00000000000000000000042", "1"]),
            "null_in_required": W.insert(43, [W.NULL, "1"]),
            "oid_minus_zero": W.insert(42, ["1"] + ["0", "0", "0", "t", "-0"] + ["0", "0", "0", "t", "0", "0", "0", "0"]),
            "int2_overflow": W.insert(42, ["1"] + ["0", "0", "32768", "t", "0"] + ["0", "0", "0", "t", "0", "0", "0", "0"]),
            "update_with_key": W.update(43, ["5", "6"], key=["4"]),   # (an Update WITHOUT an old image is the plan's own since round 6: test_k_plan_takes_updates_without_an_old_image)
            "update_toast": W.update(43, ["5", W.TOAST]),
            "delete_full": W.delete(43, old=["5", "6"]),          # REPLICA IDENTITY FULL: 'O' (a Delete BY KEY is the plan's own since round 6: test_k_plan_takes_deletes_by_key)
            "delete_key_shape": W.delete(43, key=["5", "6", "7"]),   # neither the identity columns nor the full width
            "delete_key_toast": W.delete(43, key=[W.TOAST]),
            "truncate": W.truncate([43], 1),
            "keepalive": None,
            "unknown_table": W.insert(4242, ["1"]),
            "lsn_top_bits": None,
        }[what]
        s = _plan_stream(7, 6, 90, odd=(251, odd) if odd is not None else None)
        if what == "keepalive":
            s2 = W.Stream()
            for i in range(len(s.offsets) - 1):
                s2.add_payload(bytes(s.buf[s.offsets[i] + 5:s.offsets[i + 1]]))
                if i == 200:
                    s2.add_payload(W.keepalive(0x9999))
            s = s2
        buf = np.frombuffer(s.bytes(), dtype=np.uint8).copy()
        if what == "lsn_top_bits":   # a Begin whose final_lsn does not fit the 62 bits of k_plan's LSN word (and its Commit)
            offs = s.offsets
            tags = [buf[int(offs[i]) + 30] for i in range(len(offs) - 1)]
            bi = [i for i, t in enumerate(tags) if t == ord("B")][2]
            ci = [i for i, t in enumerate(tags) if t == ord("C") and i > bi][0]
            buf[int(offs[bi]) + 31] |= 0xC0
            buf[int(offs[ci]) + 32] |= 0xC0
        err = _agree(d, o, buf, s.offsets)
        p = d.debug_paths()
        assert p["plan"] == 0 and p["plan_redone"] == 1, p
        legal = what in ("long_leading_zeros", "update_with_key", "update_toast", "delete_full", "truncate", "keepalive", "lsn_top_bits", "unknown_table")   # a table without a state is not owned: its rows are skipped
        assert (err == 0) == legal, (what, err)
        if not legal:
            assert p["redone"] == 1, p
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("pre", ["1", "0"])
def test_k_plan_takes_updates_without_an_old_image(pre):
    """pgoutput sends an Update of a table under its default replica identity WITHOUT an old image whenever the key did not change: rel |
    'N' | tuple, the Insert's layout. Since round 6 the fixed-width plan decodes those itself (kind 'U', no old row, the update payload
    counter) instead of handing the batch to the generic kernel — one such row in a 64 MiB batch used to cost the whole batch a second
    attempt and the chain behind it its overlap. Behind the sidecar pre-pass and with the plan's own look-back."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    os.environ["ETLG_PLAN_PRE"] = pre
    try:
        d, o = Decoder(0), oracle.Oracle()
        _two_tables(d); _two_tables(o)
        s = _plan_stream(7, 6, 90, odd=(251, W.update(43, ["5", "6"])))
        s2 = _plan_stream(8, 5, 70, odd=(13, W.update(43, ["-2147483648", W.NULL])))
        for st in (s, s2):
            buf = np.frombuffer(st.bytes(), dtype=np.uint8).copy()
            assert _agree(d, o, buf, st.offsets) == 0
        p = d.debug_paths()
        assert p["plan"] == 2 and p["plan_redone"] == 0 and p["redone"] == 0, p
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("tables", ["narrow_prepass", "narrow_lookback", "two_tables"])
def test_k_plan_takes_deletes_by_key(tables):
    """A Delete BY KEY (rel | 'K' | tuple: what pgoutput sends under the default replica identity) decodes to the table's key-layout row
    (normalize_key_tuple_to_row, codec/event.rs:795-923). Since round 6's last session the fixed-width plan takes it: the dense form (the
    identity columns only), the full-width form ('n' — or a value that is skipped unread — at the other positions), keys at the integer
    limits. Behind the sidecar pre-pass a Delete is priced by its length — shorter than any row frame can be, or, in the range where rows
    and Deletes overlap, by its tag byte — and a Delete the pre-pass priced as a row (a key with leading zeros beyond that range) gives
    the batch up like every broken assumption. Every event against the oracle; the plan must have been the kernel."""
    import random
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    if tables == "narrow_lookback":
        os.environ["ETLG_PLAN_PRE"] = "0"
    try:
        if tables == "two_tables":
            d, o = Decoder(0), oracle.Oracle()
            _two_tables(d); _two_tables(o)
        else:
            d, o = _pre_pair()
        rng = random.Random(11)
        forms43 = [lambda k: W.delete(43, key=[k]), lambda k: W.delete(43, key=[k, W.NULL]), lambda k: W.delete(43, key=[k[:3], "77"])]   # (a value at a position the key row skips makes the frame longer than the pre-pass looks for when the key is long too: short keys here)
        keys = ["5", "0", "-12", "2147483647", "-2147483648", "+7", "007", "123456"]
        for rnd in range(2):
            s = W.Stream()
            lsn, n, nd = 0x7000 + rnd * 0x10000, 0, 0
            for t in range(9):
                lsn += 0x100
                s.add(W.begin(lsn, ts=1000 + t, xid=70 + t))
                for r in range(41):
                    if tables == "two_tables" and rng.random() < 0.4:
                        s.add(W.insert(42, _wide_row(rng, n)))
                    else:
                        s.add(W.insert(43, [str(n), W.NULL if rng.random() < 0.3 else str(rng.randint(-10**12, 10**12))]))
                    if n % 5 == 3:
                        if tables == "two_tables" and nd % 3 == 2:
                            s.add(W.delete(42, key=[str(n)]) if nd % 2 else W.delete(42, key=[str(-n)] + [W.NULL] * 13))
                        else:
                            s.add(forms43[nd % 3](keys[nd % len(keys)]))
                        nd += 1
                    n += 1
                s.add(W.commit(lsn, lsn + 8, ts=2000 + t, flags=0))
            buf = np.frombuffer(s.bytes(), dtype=np.uint8).copy()
            assert _agree(d, o, buf, s.offsets) == 0
            assert nd > 60
        p = d.debug_paths()
        assert p["plan"] == 2 and p["plan_redone"] == 0 and p["redone"] == 0, p
        if tables == "narrow_prepass":   # a key with more leading zeros than the pre-pass looks for: priced as a row, found out by the decode kernel, decoded by the generic one
            s = _plan_stream(3, 4, 30, odd=(17, W.delete(43, key=["0" * 30 + "9"])))
            buf = np.frombuffer(s.bytes(), dtype=np.uint8).copy()
            # (_plan_stream writes table 42 as well, which this pair does not know: rows of a table without a state are skipped — by the generic kernel)
            assert _agree(d, o, buf, s.offsets) == 0
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


# ---- the sidecar pre-pass (k_plan_pre): frames priced by their length ------------------------------------------------------------
def _one_narrow_table(t):
    t.schema_put(43, 0, NARROW, name="t43")
    t.table_state(43, 1)
    t.table_ready(43, 0, [1] * len(NARROW), [1 if c[3] else 0 for c in NARROW])


def _pre_pair():
    from etl_amd.decoder import Decoder
    from oracle import oracle
    d, o = Decoder(0), oracle.Oracle()
    _one_narrow_table(d); _one_narrow_table(o)
    return d, o


@pytest.mark.parametrize("pre", ["1", "2", "0"])
def test_prepass_rows_as_long_as_a_begin_or_a_commit(pre):
    """pgoutput's Begin is 51 bytes on the wire, its Commit 56 — and so is an Insert into a two-column table whose values have three
    (eight) characters together. The pre-pass reads the tag of every frame of those two lengths, so such rows are priced as rows: the
    plan is taken, nothing is decoded again, and the arena is the oracle's. (pre = 0: the same stream through the look-back kernel.)"""
    import random
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    os.environ["ETLG_PLAN_PRE"] = pre
    try:
        d, o = _pre_pair()
        rng = random.Random(5156)
        for rep in range(3):
            s = W.Stream()
            lsn = 0x7000 + 0x10000 * rep
            n = 0
            for t in range(60):
                lsn += 0x100
                s.add(W.begin(lsn, ts=10 + t, xid=900 + t))
                for r in range(rng.randrange(0, 90)):
                    kind = rng.randrange(4)
                    if kind == 0:
                        row = [str(rng.randrange(10, 100)), str(rng.randrange(10))]               # 38 + (5 + 2) + (5 + 1) = 51 bytes
                    elif kind == 1:
                        row = [str(rng.randrange(1000, 10000)), str(rng.randrange(1000, 10000))]   # 38 + 9 + 9 = 56 bytes
                    elif kind == 2:
                        row = [str(rng.randrange(10)), W.NULL]
                    else:
                        row = [str(rng.randrange(-2**31, 2**31)), str(rng.randrange(-10**15, 10**15))]
                    s.add(W.insert(43, row))
                    n += 1
                s.add(W.commit(lsn, lsn + 8, ts=20 + t, flags=0))
            buf = np.frombuffer(s.bytes(), dtype=np.uint8)
            lens = np.diff(np.asarray(s.offsets))
            tags = np.asarray([buf[int(x) + 30] for x in s.offsets[:-1]])
            assert ((lens == 51) & (tags == ord("I"))).any() and ((lens == 56) & (tags == ord("I"))).any()
            assert _agree(d, o, buf, s.offsets) == 0
        p = d.debug_paths()
        assert p["plan"] == 3 and p["plan_redone"] == 0 and p["redone"] == 0, p
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


@pytest.mark.parametrize("what", ["begin_with_a_trailing_byte", "commit_with_trailing_bytes", "keepalive_as_long_as_a_begin"])
def test_prepass_assumption_broken_by_the_stream(what):
    """The pre-pass takes a Begin for 51 bytes and a Commit for 56. A stream that pads them (the wire parser ignores bytes behind a
    message), or carries another kind of frame of such a length, breaks what the pre-pass assumed about a frame it never read: the
    decode kernel notices (every frame is checked against the assumption), the batch is given up, and the generic kernel answers."""
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ["ETLG_FUSED_KERNEL"] = "3"
    try:
        d, o = _pre_pair()
        s = W.Stream()
        lsn = 0x8000
        for t in range(12):
            lsn += 0x100
            pad_b = b"\x00" if (what == "begin_with_a_trailing_byte" and t == 7) else b""
            pad_c = b"\x00\x00\x00" if (what == "commit_with_trailing_bytes" and t == 5) else b""
            s.add(W.begin(lsn, ts=10 + t, xid=900 + t) + pad_b)
            for r in range(40):
                s.add(W.insert(43, [str(t * 100 + r), str(r)]))
                if what == "keepalive_as_long_as_a_begin" and t == 6 and r == 20:
                    s.add_payload(W.keepalive(0x9999) + b"\x00" * (51 - 5 - len(W.keepalive(0x9999))))
            s.add(W.commit(lsn, lsn + 8, ts=20 + t, flags=0) + pad_c)
        buf = np.frombuffer(s.bytes(), dtype=np.uint8)
        _agree(d, o, buf, s.offsets)   # (whatever the oracle makes of the padding, the device makes the same)
        p = d.debug_paths()
        assert p["plan"] == 0 and p["plan_redone"] == 1, p
        d.close()
    finally:
        for k in _KNOBS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]


def test_prepass_buffers_through_their_rotation():
    """Fourteen batches of very different sizes through one context: the pre-pass's buffers rotate (four), its status tag cycles
    (three), and a word a larger batch left in a buffer must never be taken for a smaller one's (the ticket sits at a fixed place,
    group words carry the launch's tag, prefixes none)."""
    w = synth.cfg2()
    d, o = _pair(w)
    for k, kb in enumerate((4096, 120, 2300, 115, 6000, 130, 130, 2500, 120, 4500, 125, 250, 120, 2100)):   # 1 .. 4 groups of 256 tiles
        buf, offs = w.fill(kb << 10)
        assert len(offs) > 1
        assert _agree(d, o, buf, offs) == 0, (k, kb)
    p = d.debug_paths()
    assert p["plan"] == 14 and p["plan_redone"] == 0 and p["redone"] == 0, p
    d.close()


@pytest.mark.skipif(os.environ.get("ETLG_SIMT_RUN") == "1", reason="1.1 M frames: minutes on the emulator")
def test_prepass_more_than_64_groups():
    """130 MiB of cfg2 in one batch: 18 k tiles are 71 groups of 256, so the last group to arrive scans the group aggregates in two
    trips with the fold carried over; size-independent checks on the whole arena, the oracle on its head and tail."""
    from etl_amd.decoder import Decoder
    w = synth.cfg2()
    d = Decoder(0)
    w.register(d, ready=True)
    buf, offs = w.fill(130 << 20)
    nf = len(offs) - 1
    assert (nf + 63) // 64 > 64 * 256
    b = d.decode(buf, offs)
    assert b.error is None
    h = b.host()
    assert h.n_events == nf and d.debug_paths()["plan"] == 1 and d.debug_paths()["plan_redone"] == 0
    tags = buf[offs[:-1].astype(np.int64) + 30]
    fixed_dw = np.where(tags == ord("B"), 2, np.where(tags == ord("C"), 4, 6)).astype(np.int64)
    want_body = np.concatenate([[0], np.cumsum(fixed_dw)[:-1]]) * 4
    assert np.array_equal(np.asarray(h.body_off, dtype=np.int64), want_body)          # every tile's prefix, all 71 groups
    assert len(h.fixed) == int(fixed_dw.sum()) * 4
    assert np.array_equal(np.asarray(h.kind), tags)
    # commit_lsn of every row = final_lsn of the Begin that opened its transaction (the LSN the pre-pass folds across tiles and groups)
    is_b = tags == ord("B")
    b_lsn = np.zeros(nf, dtype=np.uint64)
    pos = offs[:-1].astype(np.int64)[is_b] + 31
    raw = np.stack([buf[pos + i] for i in range(8)], axis=1).astype(np.uint64)
    b_lsn[is_b] = sum(raw[:, i] << np.uint64(8 * (7 - i)) for i in range(8))
    last_b = np.maximum.accumulate(np.where(is_b, np.arange(nf), -1))
    rows = tags == ord("I")
    assert (last_b[rows] >= 0).all()
    assert np.array_equal(np.asarray(h.commit_lsn, dtype=np.uint64)[rows], b_lsn[last_b[rows]])
    d.close()

