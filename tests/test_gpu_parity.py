"""Parity tests proper: the HIP path (through the C ABI of libetl_gfx950.so)
against the oracle on the same inputs — byte for byte on the canonical arena,
and on (error code, kind, description, frame) for failing batches."""
import numpy as np
import pytest

from etl_amd import abi, synth
from tests import scenarios as SC

pytestmark = pytest.mark.gpu

ALL = {s.name: s for s in SC.all_scenarios()}

# Every device path must give the oracle's bytes: the default choice (by frame size) and each
# kernel forced through the environment knobs read by etlg_ctx_create.
PATHS = {
    "default": {},
    "fused256": {"ETLG_FUSED_KERNEL": "0"},   # k_fused, 256 frames per tile
    "fused64": {"ETLG_FUSED_KERNEL": "1"},    # k_fused, 64 frames per tile
    "cells": {"ETLG_FUSED_KERNEL": "2"},      # k_cells (column-parallel)
    "rows": {"ETLG_FUSED_KERNEL": "4"},       # k_rows (row-synchronous walk, rows.hip) wherever a batch is eligible
    "norows": {"ETLG_ROWS": "0"},             # the default choice without k_rows (k_cells / k_fused as before round 6)
    "plan": {"ETLG_FUSED_KERNEL": "3"},       # the fixed-width plan whenever the batch is eligible, no back-off (tile prefixes from the sidecar pre-pass where the tables allow it)
    "plan_lookback": {"ETLG_FUSED_KERNEL": "3", "ETLG_PLAN_PRE": "0"},   # ... with the kernel's own look-back (k_plan2: two tiles per wave)
    "plan_one": {"ETLG_FUSED_KERNEL": "3", "ETLG_PLAN_PRE": "0", "ETLG_PLAN_DBG": "512"},     # ... one tile per wave (k_plan, the kernel wide rows take)
    "plan_inplace": {"ETLG_FUSED_KERNEL": "3", "ETLG_PLAN_DBG": "1"},   # ... reading the input in place instead of the LDS window
    "noplan": {"ETLG_PLAN": "0"},             # the default choice without the plan
    "multipass": {"ETLG_FORCE_MULTIPASS": "1"},
}
_KNOBS = ("ETLG_FUSED_KERNEL", "ETLG_FORCE_MULTIPASS", "ETLG_FUSED_DBG", "ETLG_PLAN", "ETLG_PLAN_DBG", "ETLG_PLAN_PRE", "ETLG_ROWS")


@pytest.fixture(params=sorted(PATHS))
def path(request):
    import os
    saved = {k: os.environ.pop(k, None) for k in _KNOBS}
    os.environ.update(PATHS[request.param])
    yield request.param
    for k in _KNOBS:
        os.environ.pop(k, None)
        if saved[k] is not None:
            os.environ[k] = saved[k]


def _both(sc):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    ref = SC.replay(oracle.Oracle(), sc)
    dec = Decoder(0)
    got = SC.replay(dec, sc)
    dec.close()
    return ref, got


@pytest.mark.parametrize("name", sorted(ALL))
def test_scenario_parity(name, path):
    ref, got = _both(ALL[name])
    assert len(ref) == len(got)
    for i, (r, g) in enumerate(zip(ref, got)):
        assert r[:4] == g[:4], f"batch {i}: error {g[:4]} != oracle {r[:4]}"
        d = r[4].diff(g[4])
        assert not d, f"batch {i}: " + "; ".join(d[:6])


def test_native_library_is_the_one_in_tree():
    import os
    from etl_amd import native
    native.lib()
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(native.LIB_PATH) in maps


@pytest.mark.parametrize("mk,nbytes", [(synth.cfg2, 8 << 20), (synth.cfg3, 8 << 20), (synth.cfg5, 4 << 20)])
def test_large_batch_parity(mk, nbytes, path):
    """MiB-scale batches, several in a row on one context (state carried across batches).
    These streams decode without error, so the forced kernel must have produced the result
    itself (no silent redo by the multi-pass kernels)."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = mk()
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o, ready=not w.cfg.emit_relations)
    w.register(d, ready=not w.cfg.emit_relations)
    for _ in range(2):
        buf, offs = w.fill(nbytes)
        rb = o.decode(buf, offs)
        gb = d.decode(buf, offs)
        assert rb.err_code == 0 and gb.rc == 0
        diff = rb.host_batch().diff(gb.host())
        assert not diff, diff[:6]
    n = d.debug_paths()
    n.update(d.debug_rows())
    d.close()
    assert n["redone"] == 0 and n["rows_redone"] == 0, n
    want = {"fused256": "fused", "fused64": "fused", "cells": "cells", "multipass": "multipass", "rows": "rows"}.get(path)
    if path in ("plan", "plan_lookback", "plan_one", "plan_inplace", "default") and mk is synth.cfg2:
        want = "plan"   # cfg2 is what the fixed-width plan is for
    if want:
        assert n[want] == 2, n
    assert n["plan_redone"] == 0 and n["chain_rerun"] == 0, n


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_full_size_parity(mk):
    """BASELINE batch size (64 MiB) against the oracle, byte for byte, on the default path."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = mk()
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    buf, offs = w.fill(64 << 20)
    rb = o.decode(buf, offs)
    gb = d.decode(buf, offs, flags=abi.F_NO_CONTROL)
    assert rb.err_code == 0 and gb.rc == 0
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    n = d.debug_paths()
    d.close()
    assert n["redone"] == 0 and n["multipass"] == 0, n


def test_type_matrix_table_without_k_rows():
    """... and the same table with the row-synchronous kernel switched off: k_fused / 64, byte for byte, without a redo."""
    import os
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = os.environ.get("ETLG_ROWS")
    os.environ["ETLG_ROWS"] = "0"
    try:
        buf, offs = synth.type_matrix_stream(600, mix=True)
        o, d = oracle.Oracle(), Decoder(0)
        synth.type_matrix_register(o)
        synth.type_matrix_register(d)
        rb, gb = o.decode(buf, offs), d.decode(buf, offs, flags=abi.F_NO_CONTROL)
        assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, gb.rc, gb.error)
        diff = rb.host_batch().diff(gb.host())
        assert not diff, diff[:6]
        n = d.debug_paths()
        n.update(d.debug_rows())
        d.close()
        assert n["fused"] == 1 and n["rows"] == 0 and n["redone"] == 0 and n["multipass"] == 0, n
    finally:
        if saved is None:
            os.environ.pop("ETLG_ROWS", None)
        else:
            os.environ["ETLG_ROWS"] = saved


@pytest.mark.parametrize("mix", [False, True])
def test_type_matrix_table_of_68_columns(mix):
    """The reference's own type-matrix table (crates/etl/tests/replication_stream.rs:184-268, the row of :303-400): 68 replicated
    columns — one of every type the parser has an arm for, every array form, the types it hands on as text. Wider than k_cells' 32-column
    masks: the batch must be decoded by k_rows (and, with ETLG_ROWS=0, by k_fused / 64: no limit on the column count), byte for byte like the oracle, without a redo;
    inserts only, and with key-image updates and key deletes mixed in. Arrays and json stay DEFERRED source text in the arena (§4)
    and the columnar hand-off parses every array class of the row on the device."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    buf, offs = synth.type_matrix_stream(1500, mix=mix)
    o, d = oracle.Oracle(), Decoder(0)
    synth.type_matrix_register(o)
    synth.type_matrix_register(d)
    rb = o.decode(buf, offs)
    gb = d.decode(buf, offs, flags=abi.F_NO_CONTROL)
    assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, gb.rc, gb.error)
    hb = rb.host_batch()
    diff = hb.diff(gb.host())
    assert not diff, diff[:6]
    ev = hb.materialize()
    row = [e for e in ev if e["kind"] == "I"][0]["row"]
    assert len(row) == 68 and row[7] == ("Null",) and row[8] == ("String", b"\\N") and row[17][0] == "Numeric" and row[27][0] == "Deferred"
    n = d.debug_paths()
    n.update(d.debug_rows())
    # (since round 6 the row-synchronous kernel takes tables of up to 128 columns; k_fused / 64 remains the path behind it: ETLG_ROWS=0 below)
    assert n["rows"] == 1 and n["fused"] == 0 and n["cells"] == 0 and n["redone"] == 0 and n["multipass"] == 0 and n["rows_redone"] == 0, n
    if not mix:
        # the hand-off of the same batch: 29 of the 31 array columns come back as list columns parsed on the device (json[] / jsonb[]
        # stay text in the Arrow form), with the values the reference's test asserts (replication_stream.rs:613-856)
        from etl_amd.arrow import columns_to_record_batch
        gd = d.decode(buf, offs, flags=abi.F_NO_CONTROL | abi.F_OUTPUT_ON_DEVICE)
        names = [c[0] for c in synth.TYPE_MATRIX_COLS]
        cols = gd.columns(0, parse_arrays=True)
        kinds = {names[i]: cols.column(i).arrow_kind for i in range(len(names))}
        assert [k for k in names if k.endswith("_arr") and kinds[k] != abi.AK_LIST] == []   # (round 6: json[] / jsonb[] as lists of `j.to_string()` too)
        rec = columns_to_record_batch(cols, names=names, on_text="binary")
        assert rec.num_rows == 1500
        want = {"bool_arr": [True, False, None], "int4_arr": [456, None, -654], "int8_arr": [7890123456, None, -9876543210], "text_arr": ["hello", None, "world"],
                "bpchar_arr": ["ab ", None, "cd "], "numeric_arr": ["12345.6789", None, "-0.5"], "timetz_arr": ["12:30:45.123456+02", None, "23:59:59-07:30"],
                "bytea_arr": [b"\x00", None, b"\x01\x02"], "money_arr": ["$12.34", None, "-$0.01"], "float8_arr": [-7.25, None, 8.5],
                "num_multirange_arr": ["{[1.0,2.0)}", None, "{[3.0,4.0)}"], "inet_arr": ["192.0.2.1", None, "2001:db8::1"],
                "json_arr": ['{"a":1}', None, '{"b":2}'], "jsonb_arr": ['{"a":1}', None, '{"b":2}']}
        for k, v in want.items():
            assert rec.column(k)[0].as_py() == v and rec.column(k)[1499].as_py() == v, k
        cols.close()
        # ... and the whole table as ClickHouse rows: every one of the 68 columns — the 31 array columns (Array(Nullable(T)); text-like,
        # numeric, timetz, bytea and json elements as strings), json as serde_json's Display — written on the device, byte for byte
        from oracle import protobuf as PB
        from oracle import rowbinary as RB
        flags = [1 if c.nullable else 0 for c in hb.slots[0].cols] + [0, 0]
        rrows, idx, host = RB.encode_events(ev, 0, [c.type_class for c in hb.slots[0].cols], flags, abi.CH_MERGE_TREE, "PrimaryKey", None)
        r = gd.rowbinary(0, flags, abi.CH_MERGE_TREE)
        assert r.status == abi.RB_OK and r.n_rows == len(rrows) == 1500 and host == 0
        assert r.bytes().tobytes() == b"".join(rrows)
        r.close()
        # BigQuery takes no NULL inside an array (the row's arrays hold one): the reference's error for the first such cell of the first row
        from etl_amd.decoder import EtlError
        with pytest.raises(PB.NullValuesNotSupportedInArrayInDestination) as oi:
            PB.event_rows(ev, 0, synth.TYPE_MATRIX_COLS, "PrimaryKey")
        with pytest.raises(EtlError) as ei:
            gd.protobuf(0)
        assert ei.value.kind == abi.NullValuesNotSupportedInArrayInDestination and ei.value.detail == str(oi.value) and ei.value.frame_index == 1
        gd.close()
    d.close()


def test_full_size_parity_cfg5_control_path():
    """BASELINE configs[4] at the BASELINE batch size: 64 MiB of the cfg5 stream (Relation / DDL messages, Type / Origin noise,
    keepalives, 3 tables) with DEFAULT flags — the optimistic kernel, then the control path — byte for byte, two batches in a
    row (the second starts from the schemas the first one's DDL messages left)."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg5()
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o, ready=False)
    w.register(d, ready=False)
    for _ in range(2):
        buf, offs = w.fill(64 << 20)
        rb = o.decode(buf, offs)
        gb = d.decode(buf, offs)
        assert rb.err_code == 0 and gb.rc == 0, gb.error
        diff = rb.host_batch().diff(gb.host())
        assert not diff, diff[:6]
    n = d.debug_paths()
    d.close()
    assert n["redone"] == 0 and n["control"] == 2, n


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3])
def test_full_size_parity_without_sidecar(mk):
    """64 MiB with frame_offsets = NULL: the device finds the record boundaries itself (scan.hip) before it decodes."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = mk()
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    buf, offs = w.fill(64 << 20)
    rb = o.decode(buf, offs)
    gb = d.decode(buf, None, flags=abi.F_NO_CONTROL)
    assert rb.err_code == 0 and gb.rc == 0
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    d.close()


def test_device_resident_io_and_no_control_flag():
    """Input already in HBM, output left in HBM, control-plane round trip skipped."""
    import torch
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg2()
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o)
    w.register(d)
    buf, offs = w.fill(4 << 20)
    tb = torch.from_numpy(buf.copy()).cuda()
    to = torch.from_numpy(offs.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    flags = abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL | abi.F_ASYNC
    b = d.decode_device(tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, flags)
    assert b.sync() == 0
    assert b.view().on_device == 1
    diff = o.decode(buf, offs).host_batch().diff(b.host())
    assert not diff, diff[:6]
    # the NO_CONTROL assertion is verified on the device
    w5 = synth.cfg1()
    d2 = Decoder(0)
    w5.register(d2, ready=False)
    buf5, offs5 = w5.fill(1 << 20)
    b5 = d2.decode(buf5, offs5, flags=abi.F_NO_CONTROL)
    assert b5.error is not None and b5.error.code == abi.E_CTRL_HINT and b5.error.frame_index == 1
    d.close(); d2.close()


def test_full_size_properties_cfg2():
    """BASELINE config size (64 MiB batch): size-independent properties instead of the oracle —
    one event per frame, ordinals restart at every Begin, LSNs monotone, values = the decimal text."""
    from etl_amd.decoder import Decoder
    w = synth.cfg2()
    d = Decoder(0)
    w.register(d)
    buf, offs = w.fill(64 << 20)
    b = d.decode(buf, offs, flags=abi.F_NO_CONTROL)
    assert b.rc == 0
    hb = b.host()
    nfr = len(offs) - 1
    assert hb.n_events == nfr == hb.n_frames
    kinds = hb.kind
    is_b = kinds == ord("B")
    assert np.all(hb.tx_ordinal[is_b] == 0)
    starts = np.flatnonzero(is_b)
    ends = np.append(starts[1:], nfr)
    # ordinal = position inside its transaction
    pos = np.arange(nfr) - np.repeat(starts, ends - starts)
    assert np.array_equal(hb.tx_ordinal, pos.astype(np.uint64))
    assert np.all(np.diff(hb.start_lsn.astype(np.int64)) > 0)
    ins = np.flatnonzero(kinds == ord("I"))
    assert hb.payload_bytes == (50 * len(ins), 0, 0)
    # spot-check 4096 random rows against the decimal text in the input
    rng = np.random.default_rng(7)
    for i in rng.choice(ins, 4096, replace=False):
        o0 = int(offs[i])
        fr = buf[o0:int(offs[i + 1])].tobytes()
        vals = [int(fr[43 + 15 * k:53 + 15 * k]) for k in range(5)]
        base = int(hb.body_off[i])
        got = hb.fixed[base + 4:base + 24].view(np.int32).tolist()
        assert got == vals
    d.close()


# ---- UTF-8 validation (core::str::from_utf8, call site codec/event.rs:976): the single-pass
# kernels validate String cells a dword at a time across lanes, so every sequence is tried at
# every alignment, at the start and at the very end of the text.
_VALID = [b"", b"a"] + [chr(cp).encode() for cp in (0x80, 0x7FF, 0x800, 0xD7FF, 0xE000, 0xFFFF, 0x10000, 0x10FFFF)] + \
         [(chr(0xE9) + chr(0x4E2D) + chr(0x1F600)).encode()]
_INVALID = [bytes.fromhex(h) for h in (
    "80", "bf", "c080", "c1bf", "e08080", "e09fbf", "f0808080", "f08fbfbf", "eda080", "edbfbf", "f4908080", "f5808080",
    "ff", "fe", "c2", "e0a0", "f09080", "e180", "f180", "f1", "c241", "e18041", "f1808041", "c28080", "e14180", "f1804180")]


def _utf8_texts(seqs):
    out = []
    for s in seqs:
        for p in range(9):
            for q in (0, 1, 2, 3, 5):
                out.append(b"a" * p + s + b"b" * q)
    return out


def _text_batch(texts, wide):
    from tests import pgwire as W
    cols = [("id", SC.INT8, False, 1), ("t", SC.TEXT, False, 0)] + ([("pad", SC.TEXT, False, 0)] if wide else [])
    rows = [[str(i), t] + (["x" * 300] if wide else []) for i, t in enumerate(texts)]
    s = SC.txn([W.insert(42, r) for r in rows])
    return SC.simple_table(cols), s


@pytest.mark.parametrize("wide", [False, True])
def test_utf8_matrix(path, wide):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    # all valid texts in one batch: no error, identical arenas, and no redo by the multi-pass kernels
    prime, s = _text_batch(_utf8_texts(_VALID), wide)
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert rb.err_code == 0 and gb.rc == 0
    assert not rb.host_batch().diff(gb.host())
    assert d.debug_paths()["redone"] == 0
    d.close()
    # invalid texts one per batch: same error, same frame
    texts = _utf8_texts(_INVALID)
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    for i in range(0, len(texts), 7):
        _, s = _text_batch([b"ok", texts[i], b"ok"], wide)
        buf = np.frombuffer(s.bytes(), dtype=np.uint8)
        rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
        assert rb.err_code == abi.E_UTF8 and rb.err_frame == 2
        assert gb.error is not None and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame), texts[i]
        o.reset_stream_state(); d.reset_stream_state()
    d.close()


# ---- float4 / float8 (Rust dec2flt, codec/text.rs:52-59; KATs :510-578): the device decodes the
# exactly-representable fast path and defers the rest (include/etlg.h). Bits and the value/deferred
# decision must match the oracle, whose values come from glibc strtod / strtof.
def _float_texts():
    import random
    rng = random.Random(20260921)
    t = ["0", "-0", "0.0", "-0.0", "+0", "0e0", "0e999999999", "-0.000e-5", "1", "-1", "1.5", "-7.25", "3.5", "0.1", "0.2", "0.3",
         "1e22", "1e23", "1e-22", "1e-23", "123456789012345678", "9007199254740992", "9007199254740993", "9007199254740991",
         "12345678901234567890", "0.000001", "1.7976931348623157e308", "2.2250738585072014e-308", "5e-324", "4.9e-324",
         "3.4028235e38", "3.4028236e38", "1.17549435e-38", "1e-45", "1.401298464324817e-45", "16777216", "16777217", "16777218",
         "33554434", "8388608.5", "8388609.5", "0.5", "1.0000000596046448", "1.00000005960464477539", "1.00000011920928955",
         "inf", "-inf", "Infinity", "-INFINITY", "+inf", "nan", "NaN", "-nan", "1.", ".5", "-.5e1", "+1.25E+2", "1E5", "1e+05",
         "100000000000000000000000", "1000000000000000000000", "0.00000000000000000000001", "123.456e-2", "00012.500",
         # wide exponents, subnormals, overflow, and long mantissas whose truncation is inconclusive (still DEFERRED)
         "1e308", "1.7976931348623158e308", "1e309", "2.2250738585072011e-308", "4.9406564584124654e-324", "2.4703282292062327e-324",
         "1e-400", "3.4028235677973366e38", "1.4012984643248171e-45", "7.0064923216240854e-46", "1e-46",
         "50537618.817359292015891086651596749e82", "107896223265412489690691363e88", "28879636596541978310003766487.741e-212",
         "679604465747276.5742380775679666e23", "7137255.607280446341269933e14", "5693107746173304490483329377e264"]
    for _ in range(3000):
        nd = rng.randint(1, 21)
        digs = "".join(rng.choice("0123456789") for _ in range(nd))
        if rng.random() < 0.6:
            k = rng.randint(0, nd)
            digs = digs[:k] + "." + digs[k:]
            if digs == ".":
                digs = "0."
        s = rng.choice(["", "", "-", "+"]) + digs
        if rng.random() < 0.5:
            s += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randint(0, 40) if rng.random() < 0.8 else rng.randint(0, 400))
        t.append(s)
    return t


def test_float_matrix(path):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    from tests import pgwire as W
    cols = [("id", SC.INT8, False, 1), ("f8", SC.FLOAT8, False, 0), ("f4", SC.FLOAT4, False, 0)]
    texts = _float_texts()
    s = SC.txn([W.insert(42, [str(i), t, t]) for i, t in enumerate(texts)])
    prime = SC.simple_table(cols)
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert rb.err_code == 0 and gb.rc == 0
    hb = rb.host_batch()
    diff = hb.diff(gb.host())
    assert not diff, diff[:6]
    assert d.debug_paths()["redone"] == 0
    # the matrix exercises both outcomes for both widths
    st = [hb.fixed[int(hb.body_off[i + 1])] & 0x3F for i in range(len(texts))]   # state bits of (id, f8, f4)
    f8 = [(x >> 2) & 3 for x in st]
    f4 = [(x >> 4) & 3 for x in st]
    assert abi.CELL_DEFERRED in f8   # the inconclusive long mantissas above
    assert sum(1 for x in f8 if x == abi.CELL_VALUE) > len(texts) * 0.98 and sum(1 for x in f4 if x == abi.CELL_VALUE) > len(texts) * 0.98
    d.close()
    # malformed texts: the reference's "Float parsing failed" at the right frame
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    for bad in ["", "+", "-", ".", "e5", "1e", "1e+", "1.2.3", "1 ", " 1", "0x10", "1_0", "infinit", "nane", "--1", "1e5.0", "1f", "١"]:
        for col in (1, 2):
            row = ["7", "1.0", "1.0"]
            row[col] = bad
            s = SC.txn([W.insert(42, ["1", "2.5", "2.5"]), W.insert(42, row)])
            buf = np.frombuffer(s.bytes(), dtype=np.uint8)
            rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
            assert rb.err_code == abi.E_FLOAT and rb.err_frame == 2, bad
            assert gb.error is not None and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame), bad
            o.reset_stream_state(); d.reset_stream_state()
    d.close()


def _numeric_texts(seed=20260922, n=5000):
    """Decimal texts around the device's loop-free path (sign, digits, one '.', at most 24 characters) and just outside it:
    leading / trailing zeros, every position of the '.', lengths up to 30, exponents, '_', blanks, special values."""
    import random
    rng = random.Random(seed)
    t = ["0", "-0", "+0", "0.0", "-0.00", ".5", "-.5", "+.5", "5.", "-5.", "00.00", "0.0001", "0.00010", "100", "1000", "10000", "100000000",
         "9999", "9999.9999", "10000.0001", "0000120.00", "1200000", "000000000000000000000000", "0000000000000000000000001",
         "999999999999999999999999", "9999999999999999999999999", "1" + "0" * 23, "1" + "0" * 24, "." + "0" * 22 + "1", "." + "0" * 23 + "1",
         "12345678901.2345678901234", "123456789012.345678901234", "NaN", "Infinity", "-Infinity", "1e5", "1.5e-3", "1_000", " 1.5", "1.5 "]
    for _ in range(n):
        nd = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17, 20, 22, 23, 24, 25, 26, 30])
        kind = rng.random()
        if kind < 0.25:
            digs = "".join(rng.choice("0000123456789") for _ in range(nd))
        elif kind < 0.5:
            z1, z2 = rng.randint(0, nd), rng.randint(0, nd)
            digs = ("0" * z1 + "".join(rng.choice("0123456789") for _ in range(nd)) + "0" * z2)[:nd]
        else:
            digs = "".join(rng.choice("0123456789") for _ in range(nd))
        if rng.random() < 0.7:
            k = rng.randint(0, len(digs))
            digs = digs[:k] + "." + digs[k:]
            if digs == ".":
                digs = "0."
        s = rng.choice(["", "", "-", "+"]) + digs
        r = rng.random()
        if r < 0.05:
            s += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.randint(0, 30))
        elif r < 0.08 and len(s) > 3 and s[-2].isdigit() and s[-1].isdigit():
            s = s[:-1] + "_" + s[-1]
        t.append(s)
    return t


def test_numeric_matrix(path):
    """PgNumeric::from_str on the device, with and without the loop-free path for plain decimals (k_cells / k_fused staged
    tiles take it, the multi-pass kernels do not): header, weight, scale and base-10000 digits byte for byte against the oracle."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    from tests import pgwire as W
    cols = [("id", SC.INT8, False, 1), ("n", SC.NUMERIC, False, 0), ("s", SC.TEXT, False, 0), ("m", SC.NUMERIC, True, 0)]
    texts = _numeric_texts()
    s = SC.txn([W.insert(42, [str(i), t, "x" * (i % 7), texts[-1 - i] if i % 3 else SC.N]) for i, t in enumerate(texts)])
    prime = SC.simple_table(cols)
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, rb.err_frame, gb.rc)
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    assert d.debug_paths()["redone"] == 0
    d.close()
    # malformed texts: "invalid numeric" at the right frame, whatever path looked at them first
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    for bad in ["", "+", "-", ".", "-.", "..", "1..2", "1.2.3", ".1.", "--1", "+-1", "1-", "12a", "1e", "1e+", "e5", "1__0", "_1", "1_", "1._5", "0x10", "١", "1.5é"]:
        sb = SC.txn([W.insert(42, ["1", "2.5", "a", "2.5"]), W.insert(42, ["7", bad, "b", SC.N])])
        buf = np.frombuffer(sb.bytes(), dtype=np.uint8)
        rb, gb = o.decode(buf, sb.offsets), d.decode(buf, sb.offsets)
        assert rb.err_code != 0 and rb.err_frame == 2, bad
        assert gb.error is not None and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame), bad
        o.reset_stream_state(); d.reset_stream_state()
    d.close()


def _temporal_texts():
    """(type oid, text): the reference's own temporal KATs (codec/time.rs:169-345, etl-postgres/src/time.rs:231-258 — fast-path and
    chrono-fallback shapes, leap seconds, > 9 fraction digits, every rejected shape) plus shapes around chrono's grammar: whitespace
    in front of numeric items, one-digit fields, signed and > 4-digit years, the year range's ends, offsets that move the date."""
    from tests.golden import reference_kats as K
    t = [(oid, text) for oid, text, _e in K.TIME_RS + K.PG_TIME_RS]
    t += [(SC.DATE, x) for x in ["2023-1-1", " 2023-01-01", "2023- 1- 1", "2023-01-01 ", "+2023-01-01", "-0001-01-01", "+12023-01-01", "-262143-01-01",
                                 "+262142-12-31", "+262143-01-01", "-262144-12-31", "0-1-1", "00000-01-01", "+0000000001-01-01", "2023-001-01", "2023-01-001",
                                 "2023\u00a0-01-01", "\u20032023-01-01", "2023-02-29", "2024-02-29", "1900-02-29", "2000-02-29", "+99999999999999999999-01-01"]]
    t += [(SC.TIME, x) for x in ["1:2:3", " 01:02:03", "01: 02: 03", "01:02:03 ", "01:02:03.", "01:02:03.1234567891234", "23:59:60.5", "23:60:00", "00:00:61",
                                 "1:02:03.5", "001:02:03", "12:30", "12:30:45:10", "\u00a012:30:45"]]
    t += [(SC.TIMESTAMP, x) for x in ["2023-1-01 1:2:3", "2023-01-01  12:30:45", "2023-01-0112:30:45", "2023-01-01 12:30:45.1234567890", "2023-12-31 23:59:60",
                                      "+12023-01-01 00:00:00", "2023-01-01\t12:30:45", "2023-01-01 12:30:45 ", "2023-01-01", "-0044-03-15 12:00:00.25"]]
    t += [(SC.TIMESTAMPTZ, x) for x in ["2023-1-01 1:2:3+02", "2023-01-01 12:30:45 +02", "2023-01-01 12:30:45\u00a0+02", "2023-12-31 23:59:60+00", "2023-12-31 23:59:60-05:30",
                                        "+262142-12-31 23:59:59-01", "-262143-01-01 00:00:00+01", "+262142-12-31 23:59:59+01", "2023-01-01 12:30:45.1234567890-15:59:59",
                                        "2023-01-01 12:30:45+", "2023-01-01 12:30:45", "0000-01-01 00:00:00+15:59:59", "2023-1-01 00:00:00+00:00:01"]]
    # around the register-only fast path (iso_timestamp_swar / tz_hours_swar, codec.hip.h): every fraction length, a non-digit in every field,
    # the range limits of every field, whole-hour offsets up to the 16-hour bound, the separators one by one
    swar = ["2023-06-15 12:30:45"] + ["2023-06-15 12:30:45." + "123456789"[:k] for k in range(1, 10)] + ["2023-06-15 12:30:45." + "000000001"[:k] for k in (1, 5, 9)]
    swar += ["2023-06-15 12:30:45.12345678x", "2023-06-15 12:30:45.x", "2023-06-15 12:30:45.1234x678", "2023-06-15 12:30:45.", "2023-06-15 12:30:45.1234567890"]
    swar += ["2023-06-15 24:00:00", "2023-06-15 23:60:00", "2023-06-15 23:59:60", "2023-06-15 23:59:59.999999999", "2023-13-01 00:00:00", "2023-00-10 00:00:00",
             "2023-02-29 00:00:00", "2024-02-29 00:00:00", "2023-04-31 00:00:00", "0000-01-01 00:00:00", "9999-12-31 23:59:59.999999"]
    swar += ["2023-06-15 12:30:45"[:i] + ch + "2023-06-15 12:30:45"[i + 1:] for i in range(19) for ch in ("x", "/", "\u00e9")[:2]]
    t += [(SC.TIMESTAMP, x) for x in swar]
    t += [(SC.TIMESTAMPTZ, x + z) for x in swar[:24] for z in ("+00", "-00", "+15", "-15", "+16", "-16", "+1x", "+05:30", "-0530", "+9")]
    t += [(SC.TIMETZ, x) for x in ["1:2:3+02", "12:30:00 +02", "12:30:00\u2003+02", " 12:30:00+02", "23:59:60+00", "12:30:00.1234567890+02", "12:30:00.+02", "12:30+02", "+02",
                                   "12:30:00+02 ", "12-30-00"]]
    return t


def test_temporal_matrix(path):
    """date / time / timestamp / timestamptz / timetz texts of every shape — the reference's fixed-layout fast paths AND what it
    hands to chrono (codec/time.rs:21-71) — are decoded on the device: same value bytes as the oracle's full semantics, no cell
    DEFERRED, and for rejected texts "Datetime parsing failed" (or the UTF-8 error) at the same frame, on every kernel path."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    from tests import pgwire as W
    order = [SC.DATE, SC.TIME, SC.TIMESTAMP, SC.TIMESTAMPTZ, SC.TIMETZ]
    cols = [("id", SC.INT8, False, 1)] + [(f"c{k}", oid, True, 0) for k, oid in enumerate(order)] + [("s", SC.TEXT, True, 0)]
    prime = SC.simple_table(cols)
    texts = _temporal_texts()
    good, bad = [], []
    for oid, text in texts:
        (good if not oracle.parse_text_cell(oid, text).startswith("Err(") else bad).append((oid, text))
    assert len(good) > 60 and len(bad) > 60

    def row(i, oid, text):
        r = [str(i)] + [SC.N] * len(order) + ["x" * (i % 5)]
        r[1 + order.index(oid)] = text.encode("utf-8", "surrogatepass")
        return r
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    s = SC.txn([W.insert(42, row(i, oid, text)) for i, (oid, text) in enumerate(good)])
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, rb.err_frame, gb.rc, gb.error)
    hb = rb.host_batch()
    diff = hb.diff(gb.host())
    assert not diff, diff[:6]
    assert all(c[0] != "Deferred" for e in hb.materialize() if e["kind"] == "I" for c in e["row"])
    o.reset_stream_state(); d.reset_stream_state()
    for oid, text in bad:
        sb = SC.txn([W.insert(42, row(1, SC.DATE, "2024-02-29")), W.insert(42, row(2, oid, text))])
        buf = np.frombuffer(sb.bytes(), dtype=np.uint8)
        rb, gb = o.decode(buf, sb.offsets), d.decode(buf, sb.offsets)
        assert rb.err_code != 0 and rb.err_frame == 2, (oid, text)
        assert gb.error is not None and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame), (oid, text, gb.error)
        o.reset_stream_state(); d.reset_stream_state()
    d.close()


def test_uuid_matrix(path):
    """Uuid::parse_str forms (simple, hyphenated, braced, urn) in both cases at every alignment the preceding text column
    produces, and every malformed neighbour of them (wrong hyphen, non-hex character, one character short / long)."""
    import random
    from etl_amd.decoder import Decoder
    from oracle import oracle
    from tests import pgwire as W
    rng = random.Random(7)
    cols = [("id", SC.INT8, False, 1), ("s", SC.TEXT, False, 0), ("u", SC.UUID, False, 0)]
    prime = SC.simple_table(cols)
    good = []
    for i in range(600):
        h = "%032x" % rng.getrandbits(128)
        if rng.random() < 0.5:
            h = h.upper() if rng.random() < 0.5 else "".join(c.upper() if rng.random() < 0.5 else c for c in h)
        hy = "-".join([h[:8], h[8:12], h[12:16], h[16:20], h[20:]])
        good.append(rng.choice([h, hy, hy, "{" + hy + "}", "urn:uuid:" + hy]))
    s = SC.txn([W.insert(42, [str(i), "p" * (i % 9), u]) for i, u in enumerate(good)])
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    rb, gb = o.decode(buf, s.offsets), d.decode(buf, s.offsets)
    assert rb.err_code == 0 and gb.rc == 0
    diff = rb.host_batch().diff(gb.host())
    assert not diff, diff[:6]
    d.close()
    hy = "123e4567-e89b-12d3-a456-426614174000"
    bad = [hy[:-1], hy + "0", hy.replace("-", "", 1), hy[:8] + "_" + hy[9:], hy[:13] + "0" + hy[14:], "g" + hy[1:], hy[:35] + "G", hy[:20] + "é" + hy[22:],
           "{" + hy, hy + "}", "{" + hy + ")", "urn:uuid:" + hy[:-1], "urn-uuid:" + hy, "URN:UUID:" + hy, hy.replace("-", "")[:-1], hy.replace("-", "") + "0",
           hy.replace("-", "")[:31] + "x", "{" + hy.replace("-", "") + "}", hy[:8] + hy[9:13] + "-" + hy[13:], "@" * 36, "`" * 32, "/" * 36, ":" * 36]
    o, d = oracle.Oracle(), Decoder(0)
    prime(o); prime(d)
    for k, b in enumerate(bad):
        sb = SC.txn([W.insert(42, ["1", "q" * (k % 5), hy]), W.insert(42, ["2", "r" * (k % 4), b])])
        buf = np.frombuffer(sb.bytes(), dtype=np.uint8)
        rb, gb = o.decode(buf, sb.offsets), d.decode(buf, sb.offsets)
        assert rb.err_code != 0 and rb.err_frame == 2, b
        assert gb.error is not None and (gb.error.code, gb.error.frame_index) == (rb.err_code, rb.err_frame), b
        o.reset_stream_state(); d.reset_stream_state()
    d.close()


@pytest.mark.parametrize("max_len", [40, 700, 1500, 2600, 5000])
def test_cells_tile_sizes(max_len):
    """k_cells across its tile regimes: small frames (the LDS allocation holds the three-dword table of a tile read in place),
    windows of tens of KB (one-dword cell table, four workgroups per CU), windows near the 120 KB the one-dword entries
    can address, and frames so long that every tile is read in place — all byte for byte against the oracle, with updates,
    key images, NULLs and multi-byte text in the mix."""
    import os
    from etl_amd.decoder import Decoder
    from oracle import oracle
    t = dict(rel_id=16500, name="wide_text", cols=[
        synth.col("id", synth.CK_INT8_SEQ, synth.INT8, pk=True), synth.col("n", synth.CK_NUMERIC, synth.NUMERIC),
        synth.col("t", synth.CK_TEXT, synth.TEXT, min_len=max_len // 3, max_len=max_len, utf8_pct=10),
        synth.col("u", synth.CK_UUID, synth.UUID), synth.col("tn", synth.CK_TEXT, synth.TEXT, nullable=True, null_pct=30, min_len=0, max_len=max_len // 2),
        synth.col("ts", synth.CK_TIMESTAMPTZ, synth.TIMESTAMPTZ)])
    w = synth.Workload([t], 0xC0FFEE + max_len, rows_per_txn=37, mix=(60, 30, 10), upd_key=20, upd_toast=10, name="wide_text")
    saved = os.environ.get("ETLG_FUSED_KERNEL")
    os.environ["ETLG_FUSED_KERNEL"] = "2"
    try:
        o, d = oracle.Oracle(), Decoder(0)
        w.register(o); w.register(d)
        for _ in range(2):
            buf, offs = w.fill(3 << 20)
            rb, gb = o.decode(buf, offs), d.decode(buf, offs)
            assert rb.err_code == 0 and gb.rc == 0
            diff = rb.host_batch().diff(gb.host())
            assert not diff, diff[:6]
        n = d.debug_paths()
        d.close()
        assert n["cells"] == 2 and n["redone"] == 0, n
    finally:
        if saved is None:
            os.environ.pop("ETLG_FUSED_KERNEL", None)
        else:
            os.environ["ETLG_FUSED_KERNEL"] = saved


def _wide_table(ncols, rel_id):
    """A table of `ncols` columns cycling through every value class, with text / numeric / nullable columns beyond position 16."""
    kinds = [(synth.CK_INT4, synth.INT4, {}), (synth.CK_TEXT, synth.TEXT, dict(min_len=0, max_len=40, utf8_pct=10)), (synth.CK_NUMERIC, synth.NUMERIC, {}),
             (synth.CK_BOOL, synth.BOOL, {}), (synth.CK_TIMESTAMPTZ, synth.TIMESTAMPTZ, {}), (synth.CK_UUID, synth.UUID, {}), (synth.CK_INT2, synth.INT2, {}),
             (synth.CK_TEXT, synth.TEXT, dict(nullable=True, null_pct=30, min_len=1, max_len=24))]
    cols = [synth.col("id", synth.CK_INT8_SEQ, synth.INT8, pk=True)]
    for i in range(1, ncols):
        ck, oid, kw = kinds[i % len(kinds)]
        cols.append(synth.col(f"c{i}", ck, oid, **kw))
    return dict(rel_id=rel_id, name=f"wide{ncols}", cols=cols)


@pytest.mark.parametrize("ncols", [17, 24, 32])
def test_tables_wider_than_16_columns(ncols, path):
    """The reference has no column limit (crates/etl/src/schema.rs:380-441). k_cells decodes tables of up to 32 replicated columns
    (its WIDE instantiation: two state words per row image, 64-bit column masks); wider tables take k_fused/64 or the multi-pass
    kernels (the synthetic generator stops at 32 columns; tests/scenarios.py has the hand-written wider ones). Inserts, updates with key and full old images, unchanged-toast cells (also in columns >= 16), deletes, NULLs — byte for
    byte against the oracle on every kernel path; when k_cells is forced and the table fits, it must be the kernel that produced
    the batch."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.Workload([_wide_table(ncols, 16600 + ncols)], 0xA11CE + ncols, rows_per_txn=23, mix=(50, 35, 15), upd_key=30, upd_toast=20, name=f"wide{ncols}")
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o); w.register(d)
    for _ in range(2):
        buf, offs = w.fill(1 << 20)
        rb, gb = o.decode(buf, offs), d.decode(buf, offs)
        assert rb.err_code == 0 and gb.rc == 0, (rb.err_code, gb.error)
        diff = rb.host_batch().diff(gb.host())
        assert not diff, diff[:6]
    n = d.debug_paths()
    d.close()
    assert n["redone"] == 0, n
    if path == "cells":
        assert n["cells"] == 2, n


@pytest.mark.parametrize("cap", [1, 150, 400, 4096])
def test_control_frames_that_do_not_fit_the_staging_buffer(cap):
    """The control pre-pass gathers the Relation / DDL frames' bytes into one staging buffer for a single device-to-host
    copy; a frame that does not fit is fetched from the input by itself. With the buffer forced down to a few hundred
    bytes both routes serve the same batch, device-resident input or not, and the decode stays byte-identical."""
    import os
    from etl_amd.decoder import Decoder
    from oracle import oracle
    saved = os.environ.get("ETLG_CTRL_STAGE_CAP")
    os.environ["ETLG_CTRL_STAGE_CAP"] = str(cap)
    try:
        w = synth.cfg5()
        o, d = oracle.Oracle(), Decoder(0)
        w.register(o, ready=False); w.register(d, ready=False)
        import torch
        on_gpu = torch.cuda.is_available()   # the SIMT emulator build (CPU suite) takes host pointers as device pointers
        for k in range(4):
            buf, offs = w.fill(600 << 10)
            rb = o.decode(buf, offs)
            if k % 2 == 0:
                gb = d.decode(buf, offs)
                got = gb.host()
            else:
                o32 = np.ascontiguousarray(offs, dtype=np.uint32)
                if on_gpu:
                    tb, to = torch.from_numpy(buf.copy()).cuda(), torch.from_numpy(o32.view(np.int32).copy()).cuda()
                    bp, op = tb.data_ptr(), to.data_ptr()
                else:
                    keep = (np.ascontiguousarray(buf), o32)
                    bp, op = keep[0].ctypes.data, keep[1].ctypes.data
                gb = d.decode_device(bp, len(buf), op, len(offs) - 1, 0)
                got = gb.host()
            assert rb.err_code == 0 and gb.rc == 0
            diff = rb.host_batch().diff(got)
            assert not diff, diff[:6]
        n = d.debug_paths()
        d.close()
        assert n["control"] >= 3, n   # the first batch of a context starts on the optimistic path
    finally:
        if saved is None:
            os.environ.pop("ETLG_CTRL_STAGE_CAP", None)
        else:
            os.environ["ETLG_CTRL_STAGE_CAP"] = saved
