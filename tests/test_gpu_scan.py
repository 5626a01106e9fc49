"""Record-boundary scan on the device (etl_amd/csrc/scan.hip) against the sequential rule the
oracle and the reference's socket framing follow: 'd' | Int32-BE length chains from offset 0, a
malformed header turns the rest of the buffer into one last frame."""
import struct

import numpy as np
import pytest

from etl_amd import abi, synth
from tests import pgwire as W
from tests import scenarios as SC

pytestmark = pytest.mark.gpu


def ref_scan(buf):
    b = bytes(buf)
    n = len(b)
    offs = [0]
    p = 0
    while p < n:
        nxt = n
        if n - p >= 5:
            (L,) = struct.unpack(">I", b[p + 1:p + 5])
            if b[p] == 0x64 and L >= 4 and p + 1 + L <= n:
                nxt = p + 1 + L
        offs.append(nxt)
        p = nxt
    return np.array(offs, dtype=np.uint32)


def _check(dec, buf):
    got = dec.scan_boundaries(buf)
    want = ref_scan(buf)
    assert len(got) == len(want), (len(got), len(want))
    assert np.array_equal(got, want)


@pytest.fixture()
def dec():
    from etl_amd.decoder import Decoder
    d = Decoder(0)
    yield d
    d.close()


@pytest.mark.parametrize("mk,nbytes", [(synth.cfg2, 16 << 20), (synth.cfg3, 16 << 20), (synth.cfg5, 4 << 20), (synth.cfg1, 1 << 20)])
def test_scan_matches_sidecar(dec, mk, nbytes):
    w = mk()
    buf, offs = w.fill(nbytes)
    got = dec.scan_boundaries(buf)
    assert np.array_equal(got, offs)
    assert dec.debug_scan() == (0, 0)   # well-formed streams never need a rerun


def test_scan_small_and_empty(dec):
    _check(dec, np.zeros(0, dtype=np.uint8))
    s = SC.txn([W.insert(42, ["1", "x"])])
    _check(dec, np.frombuffer(s.bytes(), dtype=np.uint8))
    _check(dec, np.frombuffer(s.bytes()[:7], dtype=np.uint8))      # cut inside the first header
    _check(dec, np.frombuffer(b"zzz", dtype=np.uint8))              # no frame at all


def test_scan_malformed_tail_and_middle(dec):
    w = synth.cfg2()
    buf, offs = w.fill(1 << 20)
    _check(dec, buf[:len(buf) - 17])                                 # last frame cut short
    bad = buf.copy()
    bad[int(offs[len(offs) // 2])] = ord("x")                        # a header in the middle is not 'd'
    _check(dec, bad)
    bad = buf.copy()
    o = int(offs[1000])
    bad[o + 1:o + 5] = np.frombuffer(struct.pack(">I", 3), dtype=np.uint8)   # length below the minimum
    _check(dec, bad)


def test_scan_frames_longer_than_many_tiles(dec):
    big = "x" * 300_000
    rows = [W.insert(42, [str(i), big if i % 3 == 0 else "small"]) for i in range(12)]
    s = SC.txn(rows)
    _check(dec, np.frombuffer(s.bytes(), dtype=np.uint8))
    assert dec.debug_scan() == (0, 0)


def test_scan_payload_that_mimics_frames(dec):
    """Values made of byte-exact fake keepalive frames: tiles that start inside such a value guess a
    fake entry, fail their check against the real chain, and the hinted rerun repairs them."""
    fake = b"d" + struct.pack(">I", 22) + b"k" + bytes(17)          # a perfectly plausible 23-byte frame
    val = fake * 3000                                               # 69 000 bytes: covers several 8 KiB tiles
    rows = [W.insert(42, ["1", val]), W.insert(42, ["2", "plain"]), W.insert(42, ["3", val]), W.insert(42, ["4", "tail"])]
    s = SC.txn(rows)
    buf = np.frombuffer(s.bytes(), dtype=np.uint8)
    _check(dec, buf)
    reruns, seq = dec.debug_scan()
    assert reruns >= 1 and seq == 0


def _byte_soup(seed):
    """One buffer of tools/simt_fuzz.py's boundary-scan worker: frames whose bodies are random bytes, header look-alikes or whole fake
    frame chains, some of them truncated."""
    import random
    rng = random.Random(seed)

    def payload(n):
        k = rng.random()
        if k < 0.3:
            return bytes(rng.getrandbits(8) for _ in range(n))
        if k < 0.6:
            return bytes(rng.choice(b"d\x00\x00\x01\x10w") for _ in range(n))
        return (b"d" + struct.pack(">I", rng.choice([4, 5, 17, 60, 200, 4000])) + b"w") * (n // 6 + 1)
    parts, size, total = [], 0, rng.choice([40000, 150000])
    while size < total:
        n = rng.choice([108, 500, 3000, 9000, 20000, 70000])
        body = payload(n)[:n]
        fr = b"d" + struct.pack(">I", len(body) + 4) + body
        if rng.random() < 0.05:
            fr = fr[:rng.randrange(1, len(fr) + 1)]
        parts.append(fr)
        size += len(fr)
    return np.frombuffer(b"".join(parts), dtype=np.uint8)


def test_scan_after_a_one_lane_fallback(dec):
    """A scan that ends in the one-lane fallback (four runs of wrong guesses) must leave the look-back buffers clean for the scan
    after it. It did not (round 4, found by running the fuzzer's boundary-scan worker against the real library): k_bounds_seq has no
    clearing loop, the host marked the other buffer clean all the same, and on the MI355X — where a tile may look at a descriptor
    before its predecessor has published — the NEXT scan took stale words for published ones: wrong boundaries for ~30 % of the
    scans behind a fallback. (The emulator runs workgroups in order and never sees it.)"""
    soup = _byte_soup(7)
    w = synth.cfg3()
    good, offs = w.fill(1 << 20)
    fake = b"d" + struct.pack(">I", 22) + b"k" + bytes(17)
    s = SC.txn([W.insert(42, ["1", fake * 3000]), W.insert(42, ["2", "plain"]), W.insert(42, ["3", fake * 2500]), W.insert(42, ["4", "tail"])])
    mimic = np.frombuffer(s.bytes(), dtype=np.uint8)
    seq0 = dec.debug_scan()[1]
    for rep in range(24):
        _check(dec, soup)                                   # ends in the fallback ...
        assert dec.debug_scan()[1] == seq0 + rep + 1
        got = dec.scan_boundaries(good if rep % 2 else mimic)   # ... and the scan behind it is still right
        want = offs if rep % 2 else ref_scan(mimic)
        assert np.array_equal(got, np.asarray(want, dtype=np.uint32)), rep


def test_decode_without_sidecar_device_input(dec):
    """etlg_decode(frame_offsets = NULL) on HBM-resident input: boundaries come from the device scan."""
    import torch
    from oracle import oracle
    w = synth.cfg3()
    o = oracle.Oracle()
    w.register(o)
    w.register(dec)
    buf, offs = w.fill(8 << 20)
    tb = torch.from_numpy(buf.copy()).cuda()
    torch.cuda.synchronize()
    b = dec.decode_device(tb.data_ptr(), tb.numel(), None, 0, abi.F_OUTPUT_ON_DEVICE | abi.F_NO_CONTROL)
    assert b.rc == 0
    diff = o.decode(buf, offs).host_batch().diff(b.host())
    assert not diff, diff[:6]
    # host input, no sidecar
    o.reset_stream_state(); dec.reset_stream_state()
    buf2, offs2 = w.fill(4 << 20)
    g = dec.decode(buf2, None)
    assert g.rc == 0
    assert not o.decode(buf2, offs2).host_batch().diff(g.host())


@pytest.mark.parametrize("sidecar", [False, True])
def test_error_after_in_batch_relation_keeps_the_schema_slots(sidecar):
    """A batch that carries its own Relation messages and fails later: the events before the failing frame
    still name schema slots created by those Relation messages, so the view must list them — also when the
    batch came without a sidecar (the host replays the control frames from the copies it kept of their bytes;
    found by tools/simt_fuzz.py: the replay used to need host-visible offsets)."""
    from etl_amd.decoder import Decoder
    from oracle import oracle
    w = synth.cfg5()
    buf, offs = w.fill(48 << 10)
    b = bytearray(buf.tobytes())
    tags = [b[int(o) + 30] if int(offs[i + 1]) - int(o) > 31 else 0 for i, o in enumerate(offs[:-1])]
    rels = [i for i, t in enumerate(tags) if t == ord("R")]
    assert len(rels) >= 2
    # a row between two Relation messages: the failure rolls the second one back, the first one must survive
    victim = max(i for i, t in enumerate(tags) if t == ord("I") and rels[0] < i < rels[-1])
    b[int(offs[victim + 1]) - 1] = ord("x")      # the last character of the row's last value: a decode error
    mb = np.frombuffer(bytes(b), dtype=np.uint8)
    o, d = oracle.Oracle(), Decoder(0)
    w.register(o, ready=False)
    w.register(d, ready=False)
    rb, gb = o.decode(mb, offs if sidecar else None), d.decode(mb, offs if sidecar else None)
    e = gb.error
    assert rb.err_code != 0 and e is not None and (rb.err_code, rb.err_frame) == (e.code, e.frame_index)
    assert rb.err_frame == victim
    hb = rb.host_batch()
    assert len(hb.slots) > 0
    diff = hb.diff(gb.host())
    d.close()
    assert not diff, diff[:4]
