"""The Rust shim's call sequence, executed. crates/etl-gfx950 cannot be compiled here (no rustc); tests/native/shim_twin.cpp
is its twin in C++ — StagingBatcher over a ring of pinned buffers (etlg_host_alloc), decode_async / finish over ETLG_F_ASYNC from HOST
buffers, the oversize-message detour, FlushTracker, InFlight's drop order — statement for statement through the C ABI. Here it runs over
recorded streams with random dispatch points, ring sizes and buffer budgets:

  * the events it delivers, batch after batch, concatenated == the oracle's decode of the whole stream (every field);
  * its LSN bookkeeping == a model of what the reference does per message (apply.rs:2039-2051 update_last_received_lsn on receipt,
    :2055-2057 keepalives, :1918-1923 update_last_commit_end_lsn per pushed event), row by row of the twin's trace;
  * a decode error ends the run like `status?` in gpu_collect: the events before the failing frame were delivered, the error is the
    oracle's, the batches queued behind it are synced and freed before their pinned buffers (InFlight's Drop)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from etl_amd import abi, native
from etl_amd.view import HostBatch
from tests import pgwire as W
from tests import scenarios as SC

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = [("id", SC.INT8, False, 1), ("t", SC.TEXT, True, 0), ("n", SC.NUMERIC, True, 0), ("ts", SC.TIMESTAMPTZ, True, 0)]
START_LSN = 0x1000


class Trace(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("last_received_lsn", C.c_uint64), ("last_commit_end_lsn", C.c_uint64),
                ("effective_flush_lsn", C.c_uint64), ("has_commit_end", C.c_uint32), ("undelivered", C.c_uint32), ("in_transaction", C.c_uint32),
                ("in_flight", C.c_uint32)]


@pytest.fixture(scope="module")
def twin():
    out = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libshim_twin.so")
    src = os.path.join(ROOT, "tests", "native", "shim_twin.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "etlg.h"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-Wall", "-o", so, src, "-ldl"])
    L = C.CDLL(so)
    L.twin_run.restype = C.c_int32
    return L


def _stream(rng, ntxn, oversize=0, bad_at=None, keepalives=True):
    """CopyData-framed messages of `ntxn` transactions (+ keepalives between them); returns (bytes, offsets, per message: is it staged)."""
    s = W.Stream(lsn=START_LSN)
    rid = 0
    for t in range(ntxn):
        n = rng.randint(1, 40)
        final = s.lsn + 8 * (n + 2)
        s.add(W.begin(final, xid=100 + t))
        for k in range(n):
            rid += 1
            text = "x" * rng.choice([0, 3, 17, 64, 200, 900]) if rng.random() < 0.9 else W.NULL
            if oversize and t == ntxn // 2 and k == n // 2:
                text = "y" * oversize
            num = rng.choice(["0", "-12.5", "123456789.000100", "1e5", "NaN", "0.000012"])
            ts = "2026-01-02 03:04:05.%06d+00" % rng.randint(0, 999999)
            idt = str(rid)
            if bad_at is not None and rid == bad_at:
                idt = "12x4"                                   # a malformed integer: the batch that holds it fails there
            r = rng.random()
            if r < 0.7:
                s.add(W.insert(42, [idt, text, num, ts]))
            elif r < 0.9:
                s.add(W.update(42, [idt, text, num, ts], key=[idt]))
            else:
                s.add(W.delete(42, key=[idt]))
        s.add(W.commit(final, final + 8))
        if keepalives and rng.random() < 0.3:
            s.add_payload(W.keepalive(s.lsn + 4))
    return np.frombuffer(s.bytes(), dtype=np.uint8).copy(), np.array(s.offsets, dtype=np.uint32)


def _run(twin, buf, offs, ring, cap, cuts):
    from etl_amd.decoder import Decoder
    d = Decoder(0)
    SC.simple_table(COLS)(d)
    parts, status = [], []

    @C.CFUNCTYPE(None, C.c_void_p, C.POINTER(abi.BatchView), C.c_uint64, C.c_int32, C.c_int32, C.c_int64)
    def on_batch(_user, view, nframes, rc, code, frame):
        hb = HostBatch.from_view(view.contents)
        parts.append(hb)
        status.append((int(nframes), int(rc), int(code), int(frame)))

    trace = (Trace * (2 * len(offs) + 64))()
    nt = C.c_uint32()
    cut_arr = np.ascontiguousarray(cuts, dtype=np.uint8)
    rc = twin.twin_run(native.LIB_PATH.encode(), C.c_void_p(d.h.value if hasattr(d.h, "value") else d.h), C.c_void_p(buf.ctypes.data), C.c_void_p(offs.ctypes.data),
                       C.c_uint32(len(offs) - 1), C.c_uint32(ring), C.c_uint32(cap), C.c_void_p(cut_arr.ctypes.data), on_batch, None, trace, C.c_uint32(len(trace)),
                       C.byref(nt), C.c_uint64(START_LSN))
    rows = [trace[i] for i in range(min(nt.value, len(trace)))]
    d.close()
    return rc, parts, status, rows


def _model(buf, offs):
    """What the reference's loop holds after each message (receipt) and after each pushed event (delivery)."""
    recv, out, staged = START_LSN, [], []          # staged: (global message index, commit end lsn or None, tag)
    for i in range(len(offs) - 1):
        p = bytes(buf[offs[i] + 5:offs[i + 1]])
        if p[0:1] == b"k":
            recv = max(recv, int.from_bytes(p[1:9], "big"))                                   # apply.rs:2055-2057
        else:
            recv = max(recv, int.from_bytes(p[1:9], "big"), int.from_bytes(p[9:17], "big"))   # apply.rs:2039-2043
            tag = p[25:26]
            staged.append((i, int.from_bytes(p[26 + 9:26 + 17], "big") if tag == b"C" else None, tag))   # Commit body: flags, commit_lsn, end_lsn
        out.append(recv)
    return out, staged


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_twin_delivers_the_oracles_events_and_the_reference_lsn_bookkeeping(twin, seed):
    from oracle import oracle
    rng = random.Random(seed)
    cap = rng.choice([64 << 10, 256 << 10, 1 << 20])
    ring = rng.choice([2, 2, 3, 4])
    buf, offs = _stream(rng, rng.randint(30, 120), oversize=(cap + 5000) if seed % 2 == 0 else 0)
    nmsg = len(offs) - 1
    cuts = np.array([1 if rng.random() < 0.02 else 0 for _ in range(nmsg)], dtype=np.uint8)
    o = oracle.Oracle()
    SC.simple_table(COLS)(o)
    rb = o.decode(buf, offs)
    assert rb.err_code == 0, rb.err_desc
    rc, parts, status, rows = _run(twin, buf, offs, ring, cap, cuts)
    assert rc == 0, rc
    assert all(st[1] == 0 for st in status), status
    recv, staged = _model(buf, offs)
    assert sum(st[0] for st in status) == len(staged)                        # every staged frame was delivered exactly once
    assert len(status) >= 2 and (seed % 2 or any(st[0] == 1 for st in status))   # several batches; the oversize message travelled alone
    got = HostBatch.concat(parts)
    ref = rb.host_batch()
    got.n_frames, got.payload_bytes = ref.n_frames, ref.payload_bytes       # (keepalives are not staged; payload counters are per batch)
    diff = ref.diff(got)
    assert not diff, diff[:6]
    # ---- LSN bookkeeping, row by row
    msg_rows = [r for r in rows if r.kind == 0]
    assert [r.index for r in msg_rows] == list(range(nmsg))
    for r in msg_rows:
        assert r.last_received_lsn == recv[r.index], (r.index, hex(r.last_received_lsn), hex(recv[r.index]))
        assert r.effective_flush_lsn in (START_LSN, r.last_received_lsn)
        if r.undelivered or r.has_commit_end or r.in_transaction:
            assert r.effective_flush_lsn == START_LSN                      # nothing durable yet: the flush position cannot follow receipt
    for r in [r for r in rows if r.kind == 1]:
        ends = [e for _i, e, _t in staged[:r.index] if e is not None]
        assert bool(r.has_commit_end) == bool(ends) and (not ends or r.last_commit_end_lsn == max(ends)), (r.index, r.last_commit_end_lsn)
        last_bc = [t for _i, _e, t in staged[:r.index] if t in (b"B", b"C")]
        assert bool(r.in_transaction) == bool(last_bc and last_bc[-1] == b"B")
    end = rows[-1]
    assert end.kind == 2 and end.undelivered == 0 and end.in_flight == 0 and end.index == len(staged)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_twin_stops_at_a_decode_error_like_the_loop(twin, seed):
    from oracle import oracle
    rng = random.Random(seed)
    buf, offs = _stream(rng, 60, bad_at=rng.randint(300, 700), keepalives=True)
    o = oracle.Oracle()
    SC.simple_table(COLS)(o)
    rb = o.decode(buf, offs)
    assert rb.err_code != 0
    cuts = np.array([1 if rng.random() < 0.03 else 0 for _ in range(len(offs) - 1)], dtype=np.uint8)
    rc, parts, status, rows = _run(twin, buf, offs, 3, 128 << 10, cuts)
    assert rc == 0
    bad = [k for k, st in enumerate(status) if st[1] != 0]
    assert len(bad) == 1 and bad[0] == len(status) - 1, status          # the failing batch is the last one delivered
    _recv, staged = _model(buf, offs)
    before = sum(st[0] for st in status[:-1])
    assert staged[before + status[-1][3]][0] == rb.err_frame and status[-1][2] == rb.err_code    # same frame of the stream, same code
    got = HostBatch.concat(parts)
    ref = rb.host_batch()
    assert got.n_events == ref.n_events
    got.n_frames, got.payload_bytes = ref.n_frames, ref.payload_bytes
    diff = ref.diff(got)
    assert not diff, diff[:6]
    assert rows[-1].kind == 2 and rows[-1].in_flight == 0


# ---- copy.rs: CopyStaging ring, copy_decode_async / copy_finish, CopyInFlight's drop order (twin_copy_run)
def _copy_run(twin, rows, ring, cap):
    from etl_amd.decoder import Decoder
    from etl_amd.synth import COPY_COLS
    d = Decoder(0)
    d.schema_put(42, 0, COPY_COLS)
    slot = d.table_ready(42, 0, [1] * len(COPY_COLS), [1 if c[3] else 0 for c in COPY_COLS])
    buf = np.frombuffer(b"".join(rows), dtype=np.uint8).copy()
    offs = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint32)
    parts, status = [], []

    @C.CFUNCTYPE(None, C.c_void_p, C.POINTER(abi.BatchView), C.c_uint64, C.c_int32, C.c_int32, C.c_int64)
    def on_batch(_user, view, nrows, rc, code, frame):
        parts.append(HostBatch.from_view(view.contents))
        status.append((int(nrows), int(rc), int(code), int(frame), int(view.contents.payload_bytes[0])))

    delivered = C.c_uint64()
    twin.twin_copy_run.restype = C.c_int32
    rc = twin.twin_copy_run(native.LIB_PATH.encode(), C.c_void_p(d.h.value if hasattr(d.h, "value") else d.h), C.c_int32(slot), C.c_void_p(buf.ctypes.data),
                            C.c_void_p(offs.ctypes.data), C.c_uint32(len(rows)), C.c_uint32(ring), C.c_uint32(cap), on_batch, None, C.byref(delivered))
    paths = d.debug_copy()
    d.close()
    return rc, parts, status, int(delivered.value), slot, buf, offs, paths


@pytest.mark.parametrize("seed,ring,cap", [(1, 2, 64 << 10), (2, 3, 200 << 10), (3, 4, 32 << 10)])
def test_copy_twin_streams_rows_through_a_ring_of_pinned_stagings(twin, seed, ring, cap):
    """The rows of a table copy go through `ring` pinned staging buffers, `ring - 1` batches in flight at most, each collected in issue
    order: the rows delivered, batch after batch, are the oracle's rows of the whole stream, and every batch's payload metadata is the
    byte length of its rows (TableCopyPayloadMetadata, table_copy.rs:84)."""
    from etl_amd.synth import COPY_COLS, copy_rows
    from oracle import oracle
    rows = copy_rows(6000, 70 + seed)
    rc, parts, status, delivered, slot, buf, offs, paths = _copy_run(twin, rows, ring, cap)
    assert rc == 0 and delivered == len(rows) and all(st[1] == 0 for st in status), (rc, delivered, status[:4])
    assert len(status) >= 6 and sum(st[0] for st in status) == len(rows)
    at = 0
    for nrows, _rc, _code, _frame, payload in status:          # per batch: its rows' bytes
        assert payload == int(offs[at + nrows]) - int(offs[at])
        at += nrows
    o = oracle.Oracle()
    o.schema_put(42, 0, COPY_COLS)
    so = o.table_ready(42, 0, [1] * len(COPY_COLS), [1 if c[3] else 0 for c in COPY_COLS])
    ref = o.copy_decode(so, buf, offs).host_batch()
    got = HostBatch.concat(parts)
    got.tx_ordinal = ref.tx_ordinal                          # (the ordinal of a copied row is its index in ITS batch)
    got.n_frames, got.payload_bytes = ref.n_frames, ref.payload_bytes
    diff = ref.diff(got)
    assert not diff, diff[:6]
    assert paths["frames"] == 0


def test_copy_twin_stops_at_a_bad_row_and_drops_what_is_in_flight(twin):
    from etl_amd.synth import copy_rows
    rows = copy_rows(5000, 91)
    bad = 3333
    rows[bad] = rows[bad].replace(b"\t", b"\t\t", 1)          # one field too many
    rc, parts, status, delivered, slot, buf, offs, _paths = _copy_run(twin, rows, 3, 48 << 10)
    assert rc == 0
    failing = [k for k, st in enumerate(status) if st[1] != 0]
    assert len(failing) == 1 and failing[0] == len(status) - 1, status      # fail-fast: nothing is delivered behind the failing batch
    before = sum(st[0] for st in status[:-1])
    assert before + status[-1][3] == bad and delivered == bad               # the rows in front of the bad one, no more
