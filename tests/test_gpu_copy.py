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


def _fuzz_rows(seed, n):
    """Two text columns whose rows stress the splitter's step boundaries: field lengths around multiples of 64, runs of backslashes
    of every parity in front of separators, NULL markers at every alignment, escaped multi-byte characters, rows of several KB
    (longer than the 4 KB block), one-byte and empty fields."""
    rng = random.Random(seed)
    pieces = ["a", "xy", "\\\\", "\\t", "\\n", "\\N", "\\\\\\\\", "é", "\\é", "中", "😀", "\\😀", "\\q", " ", "\\b\\f\\r\\v"]

    def field(target):
        s = ""
        while len(s.encode()) < target:
            s += rng.choice(pieces)
        return s
    rows = []
    for i in range(n):
        k = rng.random()
        if k < 0.1:
            a, b = "\\N", field(rng.randrange(0, 140))
        elif k < 0.2:
            a, b = field(rng.choice([61, 62, 63, 64, 65, 127, 128, 129])), "\\N"
        elif k < 0.25:
            a, b = field(rng.randrange(3000, 9000)), field(rng.randrange(0, 70))
        elif k < 0.3:
            a, b = "", ""
        else:
            a, b = field(rng.randrange(0, 200)), field(rng.randrange(0, 200))
        rows.append((a + "\t" + b + "\n").encode())
    return rows


@pytest.fixture(params=["direct", "lane_per_row", "lane_per_byte"])
def splitter(request):
    """The three COPY front ends: rows -> arena in one kernel (k_copy_cells, the default; a batch with a malformed row is decoded
    again through the frames), and the two row -> frame splitters of copy.hip on their own (ETLG_COPY_DIRECT=0: one lane per row, and
    with ETLG_COPY_KERNEL=1 the data-parallel one). The knobs are read when the context is created."""
    import os
    saved = {k: os.environ.pop(k, None) for k in ("ETLG_COPY_KERNEL", "ETLG_COPY_DIRECT")}
    if request.param != "direct":
        os.environ["ETLG_COPY_DIRECT"] = "0"
    if request.param == "lane_per_byte":
        os.environ["ETLG_COPY_KERNEL"] = "1"
    yield request.param
    for k, v in saved.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_copy_splitter_step_boundaries(seed, splitter):
    cols = [("a", K.TEXT, True, 0), ("b", K.TEXT, True, 0)]
    rows = _fuzz_rows(seed, 700)
    o, d, rb, gb = both(cols, rows)
    assert rb.err_code == 0, (rb.err_desc, rb.err_frame)
    assert_same(rb, gb)
    # (rows of several KB: a tile of 64 of them can outgrow the LDS window — such a batch takes the frames even by default)
    dc = d.debug_copy()
    assert dc["direct"] + dc["frames"] == 1 and (splitter == "direct" or dc["frames"] == 1)
    d.close()
    if splitter == "direct":   # the same kinds of rows without the multi-KB ones stay in the one-kernel path
        short = [r for r in rows if len(r) < 600]
        o, d, rb, gb = both(cols, short)
        assert rb.err_code == 0
        assert_same(rb, gb)
        assert d.debug_copy() == {"direct": 1, "frames": 0}
        d.close()


@pytest.mark.parametrize("kind", ["utf8_cut_at_row_end", "utf8_overlong", "cont_at_row_start", "unterminated", "more", "fewer", "trailing_backslash",
                                  "empty_row", "dangling_after_newline", "escaped_tab_before_sep"])
def test_copy_splitter_row_errors(kind, splitter):
    """One bad (or odd) row between good ones, at several positions of the 64-byte step: the first error and everything before it
    as the oracle has them (rows are independent: a sequence cut off by a row's end is that row's error, not its neighbour's)."""
    cols = [("a", K.TEXT, True, 0), ("b", K.TEXT, True, 0)]
    bad = {"utf8_cut_at_row_end": b"abc\txy\xe4\xb8\n"[:-1] + b"",           # a 3-byte sequence cut off (and no terminator)
           "utf8_overlong": b"a\xc0\xafb\tc\n", "cont_at_row_start": b"\xa0bc\td\n",
           "unterminated": b"abc\tdef", "more": b"a\tb\tc\n", "fewer": b"abc\n", "trailing_backslash": b"a\tb\\",
           "empty_row": b"", "dangling_after_newline": b"a\tb\nrest", "escaped_tab_before_sep": b"a\\\t\tb\n"}[kind]
    for pad in (0, 5, 57, 60, 62, 63, 64, 120):
        good1 = ("g" * pad + "\tq\n").encode()
        rows = [good1, bad, b"\xb8tail\tz\n" if kind == "utf8_cut_at_row_end" else b"x\ty\n", b"k\tl\n"]
        o, d, rb, gb = both(cols, rows)
        assert_same(rb, gb)
        if kind not in ("escaped_tab_before_sep",):   # a malformed row: whatever ran first, the frames decided
            assert d.debug_copy() == {"direct": 0, "frames": 1}, (kind, pad)
        d.close()


def test_copy_generated_rows_lane_per_byte(splitter):
    """The generated rows of bench.py's copy leg (every escape of the format, NULLs, all ten classes) and the reference's known-answer
    rows through both splitters."""
    rows = _gen_rows(3000, 17)
    o, d, rb, gb = both(GEN_COLS, rows)
    assert rb.err_code == 0
    assert_same(rb, gb)
    assert d.debug_copy() == ({"direct": 1, "frames": 0} if splitter == "direct" else {"direct": 0, "frames": 1})
    assert d.debug_paths()["redone"] == 0
    d.close()
    for cols, row, _want in K.OK:
        o, d, rb, gb = both(cols, [bytes(row)])
        assert_same(rb, gb)
        d.close()
    for cols, row, code in K.ERR:
        good = {len(K.BASIC): b"1\tx\tt\n", 1: (b"7\n" if cols[0][1] == K.INT4 else b"ok\n")}[len(cols)]
        o, d, rb, gb = both(cols, [good, bytes(row)])
        assert rb.err_code == code
        assert_same(rb, gb)
        d.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_copy_field_tasks_fuzz(seed):
    """The rows -> arena kernel's splitter (cells.hip): four text columns of pieces chosen to confuse a splitter that works from
    bitmaps — escaped tabs and newlines as raw bytes behind backslash runs of every parity, `\\N` as a whole field / as a prefix / behind
    an escaped backslash, fields that end in an escaped backslash right before the separator, empty fields, multi-byte characters
    behind a backslash, rows from 5 bytes to a few hundred so that rows and fields start at every bit of the bitmap words."""
    rng = random.Random(seed)
    pieces = ["a", "bc", "\\\\", "\\\t", "\\\n", "\\N", "\\\\\\\\", "\\\\\\\t", "é", "\\é", "😀", "\\😀", "\\q", "N", "\\b", " ", "0123456789" * 3]
    cols = [(c, K.TEXT, True, 0) for c in "abcd"]

    def field():
        k = rng.random()
        if k < 0.12:
            return "\\N"
        if k < 0.2:
            return ""
        if k < 0.25:
            return "\\\\N"          # an escaped backslash, then N: the text `\N`, not NULL
        if k < 0.3:
            return "\\N" + rng.choice(pieces)
        return "".join(rng.choice(pieces) for _ in range(rng.randrange(1, rng.choice([3, 8, 25]))))
    rows = [("\t".join(field() for _ in range(4)) + "\n").encode() for _ in range(900)]
    o, d, rb, gb = both(cols, rows)
    assert rb.err_code == 0, (rb.err_desc, rb.err_frame)
    assert_same(rb, gb)
    assert d.debug_copy() == {"direct": 1, "frames": 0}
    d.close()
    # the same rows with one malformed row in the middle of a tile: the batch is decoded again through the frames, up to that row
    bad = rows[:500] + [b"a\tb\tc\n"] + rows[500:]
    o, d, rb, gb = both(cols, bad)
    assert rb.err_frame == 500 and rb.err_code != 0
    assert_same(rb, gb)
    assert d.debug_copy() == {"direct": 0, "frames": 1}
    d.close()


def test_copy_rows_of_several_kb():
    """Rows of several KB beside short ones in one tile: the row walk takes the bitmap words eight at a time (256 bytes per batch of
    loads), so long rows take several batches while their neighbours are done after one."""
    rng = random.Random(5)
    cols = [("a", K.TEXT, True, 0), ("b", K.TEXT, True, 0), ("c", K.INT4, True, 0)]
    pieces = ["abc", "\\\\", "\\t", "\\\t", "é", "\\é", "x" * 40, "\\N", " "]
    rows = []
    for i in range(24):
        if i % 5 == 2:
            a = "".join(rng.choice(pieces) for _ in range(400))          # ~5 KB
            while len(a.encode()) < 4200:
                a += rng.choice(pieces)
        else:
            a = "".join(rng.choice(pieces) for _ in range(rng.randrange(0, 6)))
        b = "\\N" if i % 3 == 0 else "".join(rng.choice(pieces) for _ in range(rng.randrange(0, 4)))
        rows.append((a + "\t" + b + "\t" + ("\\N" if i % 4 == 1 else str(i - 7)) + "\n").encode())
    o, d, rb, gb = both(cols, rows)
    assert rb.err_code == 0, (rb.err_desc, rb.err_frame)
    assert_same(rb, gb)
    assert d.debug_copy() == {"direct": 1, "frames": 0}
    d.close()


def test_copy_utf8_in_the_direct_kernel():
    """Row validity as UTF-8 (table_row.rs:51) is checked on the tile's bytes sixteen at a time while the bitmaps are built: multi-byte
    characters across every chunk and tile boundary pass; one broken byte anywhere — the first / last row of a tile, the first /
    last byte of a row, inside a sequence that straddles a 16-byte chunk — fails the batch at that row, as the oracle has it."""
    rng = random.Random(3)
    cols = [("a", K.TEXT, True, 0), ("b", K.TEXT, True, 0)]
    chars = ["a", "é", "中", "😀", "z", "ß", "ࠀ", "\U00010000"]
    rows = [("".join(rng.choice(chars) for _ in range(rng.randrange(0, 30))) + "\t" +
             "".join(rng.choice(chars) for _ in range(rng.randrange(0, 12))) + "\n").encode() for _ in range(150)]
    o, d, rb, gb = both(cols, rows)
    assert rb.err_code == 0
    assert_same(rb, gb)
    assert d.debug_copy() == {"direct": 1, "frames": 0}
    d.close()
    for r in (0, 1, 63, 64, 65, 127, 128, 149):
        row = rows[r]
        multi = [i for i, c in enumerate(row) if c >= 0x80]
        if not multi:
            continue
        for pos, repl in ((multi[0], b"A"), (multi[-1], b"\xff"), (multi[len(multi) // 2], b"\xc0"), (multi[0], b"")):
            broken = row[:pos] + repl + row[pos + 1:]
            try:
                broken.decode()
                continue   # (dropping that byte happened to leave valid text)
            except UnicodeDecodeError:
                pass
            rs = rows[:r] + [broken] + rows[r + 1:]
            o, d, rb, gb = both(cols, rs)
            assert rb.err_frame == r and rb.err_code != 0
            assert_same(rb, gb)
            d.close()


def test_copy_backslash_run_across_a_malformed_row_boundary():
    """A row without its newline that ends in backslashes, in front of a row that starts with backslashes (found by tools/copy_fuzz.py):
    inside the tile the two runs are one, so the second row's separators are judged by the wrong parity — the batch fails on the first
    row anyway, but the kernel must stay inside its fields while it works on the second (it once did not)."""
    cols = [("a", K.TEXT, True, 0), ("b", K.TEXT, True, 0), ("c", K.TEXT, True, 0)]
    for tail in (b"\\", b"\\\\", b"\\\\\\"):
        for head in range(1, 12):
            rows = [b"good\tx\ty\n", b"bbbbbbbb\t         \t" + tail, b"\\" * head + b"\tNNN\t\n", b"k\tl\tm\n"]
            o, d, rb, gb = both(cols, rows)
            assert rb.err_code != 0 and rb.err_frame == 1
            assert_same(rb, gb)
            d.close()


@pytest.mark.parametrize("ncols", [17, 24, 32])
def test_copy_wide_tables(ncols):
    """Tables of more than 16 columns take the WIDE instantiation of the rows -> arena kernel (two state words per row, 64-bit column
    masks, the splitter's column mask of the fields with backslashes up to bit 31): every class, NULLs and escapes in every column
    position, rows that span several bitmap batches."""
    rng = random.Random(ncols)
    kinds = [(K.INT4, lambda: str(rng.randrange(-10**9, 10**9))), (K.TEXT, lambda: "".join(rng.choice(["a", "é", "\\\\", "\\t", "x y", "\\N", ""]) for _ in range(rng.randrange(0, 9)))),
             (20, lambda: str(rng.randrange(-2**62, 2**62))), (16, lambda: rng.choice("tf")), (1700, lambda: rng.choice(["0", "-12.5", "1e5", "NaN", "123456.789"])),
             (701, lambda: rng.choice(["1.5", "-0.25", "1e300", "nan", "3.141592653589793"])), (17, lambda: "\\\\x" + "".join("%02x" % rng.randrange(256) for _ in range(rng.randrange(0, 6)))),
             (2950, lambda: "%08x-1111-2222-3333-%012x" % (rng.getrandbits(32), rng.getrandbits(48))), (1184, lambda: "2024-0%d-1%d 0%d:30:15.%06d+0%d" % (rng.randrange(1, 10), rng.randrange(10), rng.randrange(10), rng.randrange(10**6), rng.randrange(10)))]
    cols, gens = [], []
    for i in range(ncols):
        oid, g = kinds[(i * 5 + 3) % len(kinds)]
        cols.append(("c%d" % i, oid, True, 1 if i == 0 else 0))
        gens.append(g)
    rows = [("\t".join(("\\N" if rng.random() < 0.15 else g()) for g in gens) + "\n").encode() for _ in range(300)]
    o, d, rb, gb = both(cols, rows)
    assert rb.err_code == 0, (rb.err_desc, rb.err_frame)
    assert_same(rb, gb)
    assert d.debug_copy() == {"direct": 1, "frames": 0}
    d.close()


# ---- the ASYNC form (etlg_copy_decode with ETLG_F_ASYNC | ETLG_F_OUTPUT_ON_DEVICE): the reference's caller streams rows continuously
#      (crates/etl/src/postgres/stream/table_copy.rs:78-99); batches are enqueued back to back and finished by etlg_batch_sync
def _copy_ctx(cols=GEN_COLS):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    o, d = oracle.Oracle(), Decoder(0)
    for t in (o, d):
        t.schema_put(42, 0, cols)
    so = o.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    sd = d.table_ready(42, 0, [1] * len(cols), [1 if c[3] else 0 for c in cols])
    assert so == sd >= 0
    return o, d, so, sd


def _rows_np(rows):
    return np.frombuffer(b"".join(rows), dtype=np.uint8), np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)


ASYNC = abi.F_ASYNC | abi.F_OUTPUT_ON_DEVICE


def test_copy_async_batches_from_host_buffers():
    """Six batches of different sizes enqueued without a sync in between (host rows: each is uploaded into a block of its own beside the
    decode of the batch before it), the third with a malformed row, synced in issue order: every batch as the oracle has it — the bad
    one ends at its row with the reference's error and does not disturb the batches behind it —, payload metadata included."""
    o, d, so, sd = _copy_ctx()
    batches = []
    for k, n in enumerate([700, 64, 1500, 1, 333, 900]):
        rows = _gen_rows(n, 100 + k)
        if k == 2:
            rows[1234] = rows[1234].replace(b"\t", b"\t\t", 1)
        batches.append(_rows_np(rows))
    staged0 = d.debug_staged()
    inflight = [d.copy_decode(sd, buf, offs, flags=ASYNC) for buf, offs in batches]
    assert all(g.rc == 0 for g in inflight)                      # enqueued only
    assert d.debug_staged() - staged0 == len(batches)
    for k, ((buf, offs), g) in enumerate(zip(batches, inflight)):
        rb = o.copy_decode(so, buf, offs)
        g.sync()
        assert (g.rc != 0) == (k == 2)
        assert_same(rb, g)
        assert g.view().payload_bytes[0] == (int(offs[1234]) if k == 2 else len(buf))
    assert d.debug_copy()["direct"] == len(batches) - 1 and d.debug_paths()["chain_rerun"] == 0
    d.close()


def test_copy_async_leaves_the_stream_state_alone():
    """WAL batches and ASYNC table-copy batches on one context: a transaction left open by a WAL batch is still open — same commit LSN,
    next ordinal — after copy batches were enqueued and finished in between, and a WAL batch issued while copy batches are in flight
    finishes them first instead of chaining to their virtual transaction."""
    from tests import pgwire as W
    o, d, so, sd = _copy_ctx()
    for t in (o, d):
        t.table_state(42, abi.TS_READY)
    vals = [r.decode() for r in [b"1", b"2", b"t", b"1", b"x", b"y", b"2024-01-01 00:00:00+00", b"123e4567-e89b-12d3-a456-426614174000", b"1.5", b"\\x00"]]
    s = SC.txn([W.insert(42, vals), W.insert(42, vals)])
    whole = np.frombuffer(s.bytes(), dtype=np.uint8)
    offs = np.asarray(s.offsets, dtype=np.uint32)
    cut = 2                                                       # Begin + first Insert | second Insert + Commit
    a, ao = whole[:offs[cut]], offs[:cut + 1]
    b, bo = whole[offs[cut]:], offs[cut:] - offs[cut]
    r1, g1 = o.decode(a, ao), d.decode(a, ao)
    assert r1.err_code == 0 and g1.rc == 0 and not r1.host_batch().diff(g1.host())
    cb = [_rows_np(_gen_rows(n, 7 + n)) for n in (400, 90)]
    inflight = [d.copy_decode(sd, x, xo, flags=ASYNC) for x, xo in cb]
    g2 = d.decode(b, bo)                                          # finishes the copy batches first
    r2 = o.decode(b, bo)
    assert r2.err_code == 0 and g2.rc == 0 and not r2.host_batch().diff(g2.host())
    for (x, xo), g in zip(cb, inflight):
        assert g.sync() == 0
        assert_same(o.copy_decode(so, x, xo), g)
    d.close()


def test_copy_async_rows_the_direct_kernel_leaves_to_the_frame_path():
    """An ASYNC batch whose rows the rows -> arena kernel hands back (a row of several KB beside short ones: the tile's window) is redone
    through the frame rewrite when it is synced — the context's shared frame buffers — while the batches behind it stay in flight."""
    cols = [("a", 25, -1, False), ("b", 25, -1, False)]
    o, d, so, sd = _copy_ctx(cols)
    sets = [_fuzz_rows(31, 300), _fuzz_rows(32, 50), _fuzz_rows(33, 700)]
    sets[1][7] = sets[1][7][:-1] + b"\\\n"                         # a trailing backslash: a row-level error of the frame path
    batches = [_rows_np(r) for r in sets]
    inflight = [d.copy_decode(sd, buf, offs, flags=ASYNC) for buf, offs in batches]
    for (buf, offs), g in zip(batches, inflight):
        rb = o.copy_decode(so, buf, offs)
        g.sync()
        assert_same(rb, g)
    assert d.debug_copy()["frames"] >= 1
    d.close()


def test_copy_async_frame_path_second_attempt_behind_a_larger_batch(monkeypatch):
    """ASYNC table-copy batches that take the frame path at enqueue (a kernel is forced) share the context's frame buffers, and
    a later, larger batch re-allocates them. The earlier batch — a small one with a malformed row, so that it needs a second attempt (the
    multi-pass kernels, for the exact error cut) when it is synced — must decode through the buffers of NOW, not through the pointers of
    its first attempt (a device use-after-free before the fix): its error, its rows before the error and the batches behind it are the
    oracle's."""
    monkeypatch.setenv("ETLG_FUSED_KERNEL", "1")
    o, d, so, sd = _copy_ctx()
    small = _gen_rows(40, 901)
    small[17] = small[17].replace(b"\t", b"\t\t", 1)               # one column too many: a row-level error
    big = _gen_rows(9000, 902)                                       # ~200 x the small batch: the shared buffers grow
    mid = _gen_rows(700, 903)
    batches = [_rows_np(small), _rows_np(big), _rows_np(mid)]
    frames0 = d.debug_copy()["frames"]
    inflight = [d.copy_decode(sd, buf, offs, flags=ASYNC) for buf, offs in batches]
    for k, ((buf, offs), g) in enumerate(zip(batches, inflight)):
        rb = o.copy_decode(so, buf, offs)
        g.sync()
        assert (g.rc != 0) == (k == 0)
        assert_same(rb, g)
    assert d.debug_copy()["frames"] - frames0 == 3
    d.close()


def test_copy_async_device_resident_rows():
    import torch
    o, d, so, sd = _copy_ctx()
    keep, inflight, want = [], [], []
    for k, n in enumerate([3000, 10, 1200, 2500]):
        buf, offs = _rows_np(_gen_rows(n, 40 + k))
        tb = torch.from_numpy(buf.copy()).cuda()
        to = torch.from_numpy(offs.view(np.int32).copy()).cuda()
        keep.append((tb, to))
        want.append((buf, offs))
    torch.cuda.synchronize()
    for (tb, to), (buf, offs) in zip(keep, want):
        inflight.append(d.copy_decode_device(sd, tb.data_ptr(), tb.numel(), to.data_ptr(), len(offs) - 1, flags=ASYNC))
    for (buf, offs), g in zip(want, inflight):
        assert g.sync() == 0 and g.view().payload_bytes[0] == len(buf)
        assert not o.copy_decode(so, buf, offs).host_batch().diff(g.host())
    assert d.debug_copy()["direct"] == 4
    d.close()


def test_copy_async_second_attempt_behind_the_ring_lap_leaves_no_residue():
    """The result ring (32 blocks) is re-initialised when the batch that takes block 0 is issued. A table-copy batch of the old lap that
    is still in flight then, and is decoded again when it is synced (its rows go through the frame rewrite), writes its block AFTER that
    re-initialisation; the block's next user, 32 batches later, must not inherit what the second attempt left (payload shards are
    added to, error / give-up words are min-ed / or-ed into). Found by tools/copy_async_fuzz.py: a transaction's payload_bytes 489
    instead of 71. Sequence: 30 batches; A (block 30, redone at its sync), B (31), C (block 0: the lap's re-initialisation goes out with
    A in flight); 29 more batches; then a WAL transaction lands on block 30."""
    from tests import pgwire as W
    cols = [("a", 25, -1, False), ("b", 25, -1, False)]
    o, d, so, sd = _copy_ctx(cols)
    for t in (o, d):
        t.table_state(42, abi.TS_READY)
    small = _rows_np(_fuzz_rows(5, 20))

    def plain(n):
        for _ in range(n):
            g = d.copy_decode(sd, *small)
            assert g.rc == 0
            g.close()
    plain(30)                                                        # blocks 0 .. 29
    heavy = _rows_np([("x" * 4000 + "\t" + "y" * (k % 17) + "\n").encode() for k in range(80)] + _fuzz_rows(31, 40))   # 64 rows of 4 KB do not fit a tile's window: the direct kernel hands the batch back
    frames0 = d.debug_copy()["frames"]
    a = d.copy_decode(sd, *heavy, flags=ASYNC)                       # block 30
    b = d.copy_decode(sd, *small, flags=ASYNC)                       # block 31
    c = d.copy_decode(sd, *small, flags=ASYNC)                       # block 0: re-initialises 0 .. 30 with A still in flight
    for g, (buf, offs) in ((a, heavy), (b, small), (c, small)):
        rb = o.copy_decode(so, buf, offs)
        g.sync()
        assert_same(rb, g)
        g.close()
    assert d.debug_copy()["frames"] == frames0 + 1                   # A was decoded again, through the frame rewrite
    assert d.debug_ring_recleared() == 1                             # ... and its block cleared behind it
    plain(29)                                                        # blocks 1 .. 29
    s = SC.txn([W.insert(42, ["x" * 30, "y" * 41])])
    wb = np.frombuffer(s.bytes(), dtype=np.uint8)
    r2, g2 = o.decode(wb, s.offsets), d.decode(wb, s.offsets)        # block 30
    assert r2.err_code == 0 and g2.rc == 0
    assert not r2.host_batch().diff(g2.host())
    assert g2.view().payload_bytes[0] == 71
    d.close()
