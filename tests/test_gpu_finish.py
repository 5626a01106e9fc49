"""The finish pass (etlg_batch_finish_cells / ETLG_F_FINISH_CELLS, include/etlg.h): array cells typed and float cells settled ON THE DEVICE,
in the arena, so that a consumer of the decoded batch no longer re-parses them on the host — the reference produces Cell::Array and
correctly rounded floats at decode time (crates/etl/src/postgres/codec/text.rs:52-59, 126-134, 163-312).

Pinned two ways: (1) straight against the reference's own array vectors (tests/golden/reference_kats.py, text.rs:324-1003): literal in,
typed entry out, materialised and compared with the value the reference's test asserts; (2) arena against arena with the oracle's
restatement of the pass (oracle_batch_finish: its FULL-mode array parser — itself pinned by those vectors in tests/test_oracle_kats.py —
and glibc strtod) on the type-matrix table, fuzzed literals, toast-aliased cells, key images and the float matrix."""
import os
import random
import struct

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC
from tests.golden import reference_kats as K

pytestmark = pytest.mark.gpu

FIN = abi.F_FINISH_CELLS


def _pair():
    from etl_amd.decoder import Decoder
    from oracle import oracle
    return oracle.Oracle(), Decoder(0)


def _elem_repr(e):
    """A materialised element in the notation of tests/golden/reference_kats.py (the classes whose notation needs no calendar)."""
    k = e[0]
    if k == "Null":
        return "NULL"
    if k == "Bool":
        return "Bool(true)" if e[1] else "Bool(false)"
    if k in ("I16", "I32", "I64", "U32"):
        return f"{k}({e[1]})"
    if k == "F32":
        return "F32(NaN)" if (e[1] & 0x7FFFFFFF) > 0x7F800000 else "F32(0x%08x)" % e[1]
    if k == "F64":
        return "F64(NaN)" if (e[1] & 0x7FFFFFFFFFFFFFFF) > 0x7FF0000000000000 else "F64(0x%016x)" % e[1]
    if k == "String":
        return 'String("%s")' % e[1].decode()
    if k == "Bytes":
        return "Bytes(%s)" % e[1].hex()
    if k == "Uuid":
        return "Uuid(%s)" % e[1].hex()
    if k == "Numeric":
        _, kind, sign, weight, scale, digits = e
        if kind != abi.NUM_VALUE:
            return {abi.NUM_NAN: "Numeric(NaN)", abi.NUM_PINF: "Numeric(+Inf)", abi.NUM_NINF: "Numeric(-Inf)"}[kind]
        return "Numeric(%s,w=%d,s=%d,[%s])" % ("-" if sign else "+", weight, scale, ",".join(str(d) for d in digits))
    return None


def _array_kats():
    out = []
    for oid, text, want in K.ALL_TEXT_KATS:
        if abi_class(oid) == abi.TC_ARRAY:
            out.append((oid, text, want))
    return out


def abi_class(oid):
    from etl_amd import native
    return native.lib().etlg_type_class_of_oid(oid)


def test_reference_array_vectors_through_the_finish_pass():
    """Every array vector the reference's tests hold: an accepted literal comes back as a typed entry with the asserted elements (compared
    in the golden file's notation where that needs no calendar; the temporal ones arena-to-arena below), a rejected one stays DEFERRED."""
    from etl_amd import native
    kats = _array_kats()
    assert len(kats) >= 40
    o, d = _pair()
    checked = typed = 0
    for n, (oid, text, want) in enumerate(kats):
        table = 1000 + n
        cols = [("id", SC.INT8, False, 1), ("a", oid, True, 0)]
        prime = SC.simple_table(cols, table_id=table)
        prime(o); prime(d)
        s = SC.txn([W.insert(table, ["1", text])])
        buf = np.frombuffer(s.bytes(), dtype=np.uint8)
        rb = o.decode(buf, s.offsets)
        gb = d.decode(buf, s.offsets, flags=FIN)
        assert rb.err_code == 0 and gb.rc == 0, (text, gb.error)
        rb.finish()
        hb, gh = rb.host_batch(), gb.host()
        diff = hb.diff(gh)
        assert not diff, (oid, text, diff[:4])
        cell = [e for e in gh.materialize() if e["kind"] == "I"][0]["row"][1]
        elem = native.lib().etlg_array_elem_class(oid)
        if want == K.ERR or isinstance(want, tuple) or elem == abi.TC_JSON:
            assert cell[0] == "Deferred" and cell[2] == text.encode(), (oid, text, cell)
        else:
            assert cell[0] == "Array", (oid, text, cell)
            typed += 1
            parts = [_elem_repr(e) for e in cell[2]]
            if all(p is not None for p in parts):
                assert "Array[" + ",".join(parts) + "]" == want, (oid, text)
                checked += 1
        o.reset_stream_state(); d.reset_stream_state()
    d.close()
    assert typed >= 20 and checked >= 12, (typed, checked)


@pytest.mark.parametrize("mix", [False, True])
@pytest.mark.parametrize("how", ["flag", "flag_async", "call"])
def test_type_matrix_table_finished(mix, how):
    """The reference's type-matrix table (crates/etl/tests/replication_stream.rs:184-268): 31 array columns + 2 json per row. After the
    pass only the json / json[] cells are DEFERRED; the arena equals the oracle's; through the decode flag (synchronous, and ASYNC with
    device output) and through the call on a finished batch."""
    o, d = _pair()
    buf, offs = synth.type_matrix_stream(300, mix=mix)
    synth.type_matrix_register(o)
    synth.type_matrix_register(d)
    rb = o.decode(buf, offs)
    assert rb.err_code == 0
    settled = rb.finish()
    if how == "flag":
        gb = d.decode(buf, offs, flags=abi.F_NO_CONTROL | FIN)
        assert gb.rc == 0, gb.error
    elif how == "flag_async":
        from tests.test_gpu_async import DevBufs
        dev = DevBufs([(buf, offs)])
        p, n, po, nf = dev.items[0]
        gb = d.decode_device(p, n, po, nf, abi.F_INPUT_ON_DEVICE | abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC | FIN)
        assert gb.sync() == 0, gb.error
    else:
        gb = d.decode(buf, offs, flags=abi.F_NO_CONTROL | abi.F_OUTPUT_ON_DEVICE)
        assert gb.rc == 0, gb.error
        st = gb.finish_cells()
        assert st.arrays_typed + st.floats_settled == settled and st.left_deferred == st.deferred_seen - settled, (st.arrays_typed, st.left_deferred, settled)
        st2 = gb.finish_cells()   # idempotent
        assert st2.arrays_typed == 0 and st2.heap_bytes_added == 0
    hb, gh = rb.host_batch(), gb.host()
    diff = hb.diff(gh)
    assert not diff, diff[:6]
    row = [e for e in gh.materialize() if e["kind"] == "I"][0]["row"]
    names = [c[0] for c in synth.TYPE_MATRIX_COLS]
    kinds = {names[i]: row[i][0] for i in range(len(names))}
    assert [k for k in names if k.endswith("_arr") and kinds[k] != "Array"] == ["json_arr", "jsonb_arr"]
    assert sorted(k for k in names if kinds[k] == "Deferred") == sorted(["json_col", "jsonb_col", "json_arr", "jsonb_arr"]) or \
        sorted(k for k in names if kinds[k] == "Deferred") == sorted(k for k in names if k.startswith("json"))
    vals = dict(zip(names, row))
    assert vals["int4_arr"][2] == [("I32", 456), ("Null",), ("I32", -654)]
    assert vals["text_arr"][2] == [("String", b"hello"), ("Null",), ("String", b"world")]
    assert vals["bytea_arr"][2] == [("Bytes", b"\x00"), ("Null",), ("Bytes", b"\x01\x02")]
    assert vals["float8_arr"][2] == [("F64", struct.unpack("<Q", struct.pack("<d", -7.25))[0]), ("Null",), ("F64", struct.unpack("<Q", struct.pack("<d", 8.5))[0])]
    d.close()


def _fuzz_literal(rng, elem_kind):
    """An array literal, mostly well formed: elements of the class (sometimes not), quoting, escapes, NULLs in every case, the dimensions
    prefix, and now and then damage (an unbalanced quote, a stray brace, a dangling backslash, missing braces)."""
    def elem():
        r = rng.random()
        if r < 0.12:
            return rng.choice(["NULL", "null", "Null", "nUlL"])
        if r < 0.16:
            return rng.choice(['"NULL"', '"null"', "nul\\l", '""', "nulll", "nul"])
        if elem_kind == "int":
            v = str(rng.randint(-2 ** 40, 2 ** 40) if rng.random() < 0.2 else rng.randint(-99999, 99999))
            return rng.choice([v, v, f'"{v}"', "0" * rng.randint(0, 45) + v.lstrip("-"), v + rng.choice(["", "", "x", " "])])
        if elem_kind == "float":
            return rng.choice(["1.5", "-0", "1e10", "nan", "Infinity", "-inf", "3.4028235e38", "2.2250738585072011e-308", "9007199254740993",
                               "1.00000000000000011102230246251565404236316680908203125", "0.30000000000000004", "1e-400", "1e400", "4.9e-324", "x1",
                               "8.5070591730234615865843651857942052864e37", str(rng.random() * 10 ** rng.randint(-30, 30)),
                               "%d.%020d" % (rng.randint(0, 9), rng.randint(0, 10 ** 20 - 1))])
        if elem_kind == "numeric":
            return rng.choice(["12345.6789", "-0.5", "NaN", "Infinity", "-Infinity", "1e5", "0", "0.000", "1" + "0" * rng.randint(0, 50), "1.2.3",
                               "%d.%d" % (rng.randint(0, 10 ** 12), rng.randint(0, 10 ** 9))])
        if elem_kind == "bytea":
            h = "".join(rng.choice("0123456789abcdefABCDEF") for _ in range(2 * rng.randint(0, 12)))
            return rng.choice(['"\\\\x%s"' % h, '"\\\\x%s"' % h, "\\\\x" + h, '"\\\\x%s0"' % h, '"%s"' % h])
        if elem_kind == "bool":
            return rng.choice(["t", "f", "true", "false", "T", "1", "0", "yes"])
        if elem_kind == "temporal":
            return rng.choice(['"2023-12-25 14:30:45.123"', '"2023-01-01 12:00:00+00"', "2024-02-29", '"14:30:45.123456"', '"0001-01-01 00:00:00"', "junk"])
        if elem_kind == "uuid":
            return rng.choice(["550e8400-e29b-41d4-a716-446655440000", "550E8400E29B41D4A716446655440000", "{550e8400-e29b-41d4-a716-446655440000}", "nope"])
        # text
        body = "".join(rng.choice(["a", "b", " ", "é", "日", "x,y", "{", "}", '\\"', "\\\\", "\\n", "q"]) for _ in range(rng.randint(0, 12)))
        if any(c in body for c in ", {}") or rng.random() < 0.5 or body == "":
            return '"' + body + '"'
        return body
    n = rng.choice([0, 1, 1, 2, 3, 5, 8, 40, 70]) if rng.random() < 0.9 else rng.randint(0, 200)
    lit = "{" + ",".join(elem() for _ in range(n)) + "}"
    r = rng.random()
    if r < 0.08:
        lit = "[%d:%d]=" % (rng.randint(-3, 3), rng.randint(-3, 9)) + lit
    elif r < 0.10:
        lit = rng.choice(["[1:2][1:2]=", "[1:", "[a:b]=", "[1:2]"]) + lit
    elif r < 0.13:
        lit = rng.choice([lit[:-1], lit[1:], lit + "}", "{" + lit, lit[:-1] + '"}', lit[:-1] + "\\}", "", "{", "{{1},{2}}"])
    return lit


FUZZ_COLS = [("id", SC.INT8, False, 1), ("i4", K.INT4_A, True, 0), ("i8", K.INT8_A, True, 1), ("f8", K.FLOAT8_A, True, 0), ("f4", K.FLOAT4_A, True, 0),
             ("num", K.NUMERIC_A, True, 0), ("by", K.BYTEA_A, True, 0), ("b", K.BOOL_A, True, 0), ("ts", K.TIMESTAMP_A, True, 0), ("u", K.UUID_A, True, 0),
             ("t", K.TEXT_A, True, 1), ("m", K.MONEY_A, True, 0), ("x8", SC.FLOAT8, True, 0), ("x4", SC.FLOAT4, True, 1), ("j", K.JSONB_A, True, 0)]
FUZZ_KINDS = [None, "int", "int", "float", "float", "numeric", "bytea", "bool", "temporal", "uuid", "text", "text", None, None, "text"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzzed_literals_updates_and_key_images(seed):
    """Random literals of every element class in Insert / Update (full and key old images, unchanged-toast cells aliasing the old value) /
    Delete frames of one table whose identity holds two array columns and a float: after the pass the arena is the oracle's, settled and
    unsettled cells alike."""
    rng = random.Random(seed)
    o, d = _pair()
    ident_full = rng.random() < 0.5
    prime = SC.simple_table(FUZZ_COLS, ident=[1] * len(FUZZ_COLS) if ident_full else None)
    prime(o); prime(d)
    hard_floats = ["9007199254740993", "2.2250738585072011e-308", "1.00000000000000011102230246251565404236316680908203125", "8.5070591730234615865843651857942052864e37",
                   "0.1", "1e23", "3.4028235677973366e38", "7.0064923216240854e-46", "1.1754942106924411e-38", "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791.9999999999999999999999999999999999999999999999999999999999999999999999"]

    def row(i):
        r = [str(i)]
        for kind in FUZZ_KINDS[1:]:
            if kind is None:
                r.append(rng.choice(hard_floats) if rng.random() < 0.7 else str(rng.random()))
            elif rng.random() < 0.1:
                r.append(W.NULL)
            else:
                r.append(_fuzz_literal(rng, kind))
        return r

    def key_of(r):
        return [c for c, col in zip(r, FUZZ_COLS) if col[3]] if not ident_full else r
    msgs, live = [], []
    for i in range(400):
        q = rng.random()
        if q < 0.55 or not live:
            r = row(i); live.append(r); msgs.append(W.insert(42, r))
        elif q < 0.85:
            old = rng.choice(live); new = row(i)
            if ident_full:
                new2 = [W.TOAST if (rng.random() < 0.2 and c is not W.NULL and k > 0) else c for k, c in enumerate(new)]
                msgs.append(W.update(42, new2, old=old))
            else:
                msgs.append(W.update(42, new, key=[c if c is not W.NULL else "0" for c in key_of(old)]) if rng.random() < 0.7 else W.update(42, new))
        else:
            old = rng.choice(live)
            msgs.append(W.delete(42, old=old) if ident_full else W.delete(42, key=[c if c is not W.NULL else "0" for c in key_of(old)]))
    s = SC.txn(msgs)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb = o.decode(buf, s.offsets)
    gb = d.decode(buf, s.offsets, flags=FIN)
    assert (rb.err_code != 0) == (gb.rc != 0), (rb.err_code, gb.error)
    if rb.err_code:
        assert (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame)
    settled = rb.finish()
    assert settled > 100
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    d.close()


def test_float_matrix_leaves_no_float_deferred():
    """tests/test_gpu_parity.py's float matrix (4 000 texts around every boundary of the fast rule) with the flag: no float cell is
    DEFERRED any more, every value is glibc's correctly rounded one — the bits Rust's str::parse returns (text.rs:52-59)."""
    from tests.test_gpu_parity import _float_texts
    cols = [("id", SC.INT8, False, 1), ("f8", SC.FLOAT8, False, 0), ("f4", SC.FLOAT4, False, 0)]
    texts = _float_texts() + ["9007199254740993", "2.2250738585072011e-308", "1.00000000000000011102230246251565404236316680908203125"]
    s = SC.txn([W.insert(42, [str(i), t, t]) for i, t in enumerate(texts)])
    prime = SC.simple_table(cols)
    o, d = _pair()
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb = o.decode(buf, s.offsets)
    plain = d.decode(buf, s.offsets).host()
    n_def = sum(1 for i in range(len(texts)) if (plain.fixed[int(plain.body_off[i + 1])] >> 2) & 3 == abi.CELL_DEFERRED)
    assert n_def > 0
    d.reset_stream_state()
    gb = d.decode(buf, s.offsets, flags=FIN)
    assert rb.err_code == 0 and gb.rc == 0
    assert rb.finish() >= n_def
    hb, gh = rb.host_batch(), gb.host()
    assert not hb.diff(gh)
    for i, t in enumerate(texts):
        st = gh.fixed[int(gh.body_off[i + 1])] & 0x3F
        assert st == 0, t
        base = int(gh.body_off[i + 1])
        f8, f4 = struct.unpack_from("<Q", gh.fixed, base + 4 + 8)[0], struct.unpack_from("<I", gh.fixed, base + 4 + 16)[0]
        assert f8 == struct.unpack("<Q", struct.pack("<d", float(t)))[0] or t.lower().lstrip("+-") in ("nan",), t
        if abs(float(t)) < 1e38 and len(t) < 18:   # (short texts: the double is exact enough for numpy's second rounding to be the right one)
            assert f4 == struct.unpack("<I", struct.pack("<f", float(np.float32(t))))[0], t
    d.close()


def test_heap_grows_for_wide_arrays_and_sharded_header_follows():
    """int8[] literals of one-digit elements: 2 characters become 8 bytes per element — the typed entries outgrow the heap the decode
    allocated; the pass moves the heap and every reference stays an offset. The device header a shard publishes carries the new size."""
    cols = [("id", SC.INT8, False, 1), ("a", K.INT8_A, True, 0)]
    lit = "{" + ",".join(str(i % 10) for i in range(3000)) + "}"
    s = SC.txn([W.insert(42, [str(i), lit]) for i in range(200)])
    prime = SC.simple_table(cols)
    o, d = _pair()
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb = o.decode(buf, s.offsets)
    gb = d.decode(buf, s.offsets, flags=abi.F_OUTPUT_ON_DEVICE)
    assert gb.rc == 0
    before = gb.view().heap_bytes
    st = gb.finish_cells()
    assert st.arrays_typed == 200 and st.heap_bytes_added > 3 * before
    rb.finish()
    hb, gh = rb.host_batch(), gb.host()
    assert not hb.diff(gh)
    assert gh.materialize()[1]["row"][1][2][2999] == ("I64", 9)
    d.close()


def test_finish_on_table_copy_rows_and_on_a_batch_that_ends_in_an_error():
    """(1) etlg_copy_decode honours ETLG_F_FINISH_CELLS: COPY text rows of a table with array and float columns (the row's escapes are
    undone before the literal is seen: a backslash in the COPY form is two in the array literal) come back with typed arrays, arena =
    the oracle's copy_decode + finish. (2) A WAL batch whose 40th row has a malformed int: the events before it are finished like any
    other (the reference's consumer gets a valid prefix, fail-fast), the error and its frame are the oracle's."""
    o, d = _pair()
    cols = [("id", SC.INT8, False, 1), ("a", K.INT4_A, True, 0), ("t", K.TEXT_A, True, 0), ("f", SC.FLOAT8, True, 0), ("n", K.NUMERIC_A, True, 0)]
    for t in (o, d):
        t.schema_put(42, 0, cols)
    so, sd = o.table_ready(42, 0, [1] * 5, [1, 0, 0, 0, 0]), d.table_ready(42, 0, [1] * 5, [1, 0, 0, 0, 0])
    rng = random.Random(5)
    rows = []
    for i in range(300):
        arr = _fuzz_literal(rng, "int")
        txt = _fuzz_literal(rng, "text").replace("\\", "\\\\").replace("\t", "\\t").replace("\n", "\\n")   # the COPY form of the literal
        flt = rng.choice(["9007199254740993", "0.1", "1.00000000000000011102230246251565404236316680908203125", "\\N"])
        num = _fuzz_literal(rng, "numeric")
        rows.append(("\t".join([str(i), arr if rng.random() > 0.1 else "\\N", txt, flt, num]) + "\n").encode())
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    rb = o.copy_decode(so, buf, offs)
    gb = d.copy_decode(sd, buf, offs, flags=FIN)
    assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, gb.error)
    assert rb.finish() > 300
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    d.close()
    # (2)
    o, d = _pair()
    prime = SC.simple_table([("id", SC.INT8, False, 1), ("a", K.INT8_A, True, 0), ("v", SC.INT4, False, 0), ("f", SC.FLOAT4, False, 0)])
    prime(o); prime(d)
    msgs = [W.insert(42, [str(i), "{%d,NULL,%d}" % (i, -i), "7" if i != 40 else "7x", "16777217.5"]) for i in range(80)]
    s = SC.txn(msgs)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb = o.decode(buf, s.offsets)
    gb = d.decode(buf, s.offsets, flags=FIN)
    assert rb.err_code != 0 and gb.rc != 0 and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame)
    assert rb.finish() == 40
    hb, gh = rb.host_batch(), gb.host()
    assert not hb.diff(gh)
    assert [e for e in gh.materialize() if e["kind"] == "I"][39]["row"][1] == ("Array", "I64", [("I64", 39), ("Null",), ("I64", -39)])
    d.close()
