"""Mutation fuzzing of the HIP path against the oracle: seeded byte flips, truncations, length and tag
edits on small synthetic streams. Whatever the mutation does, both sides must agree on the error
(code, kind, description, frame) and on every byte of the arena before it — on every kernel path."""
import random
import struct

import numpy as np
import pytest

from etl_amd import synth
from tests.test_gpu_parity import path, PATHS  # noqa: F401

pytestmark = pytest.mark.gpu


def _mutate(rng, buf, offs):
    b = bytearray(buf.tobytes())
    o = [int(x) for x in offs]
    kind = rng.choice(["flip", "flip", "flip", "digit", "len", "tag", "cut", "celltag", "ncols"])
    f = rng.randrange(len(o) - 1)
    lo, hi = o[f], o[f + 1]
    if kind == "flip":
        b[rng.randrange(lo, hi)] ^= 1 << rng.randrange(8)
    elif kind == "digit":                      # turn a character of the payload into something else
        b[rng.randrange(min(lo + 38, hi - 1), hi)] = rng.choice(b"x-+. \x00\xff9")
    elif kind == "len":                        # CopyData length no longer matches the sidecar
        struct.pack_into(">I", b, lo + 1, max(0, (hi - lo - 1) + rng.choice([-3, -1, 1, 7])))
    elif kind == "tag" and hi - lo > 31:
        b[lo + 30] = rng.choice(b"BCIUDTRMOYZq")
    elif kind == "cut":                        # drop the tail of the buffer: last frame truncated
        cut = rng.randrange(lo + 1, hi)
        b = b[:cut]
        o = o[:f + 1] + [cut]
    elif kind == "celltag" and hi - lo > 45:
        b[rng.randrange(lo + 36, hi)] = rng.choice(b"ntub")
    elif kind == "ncols" and hi - lo > 40:
        struct.pack_into(">h", b, lo + 36, rng.choice([-1, 0, 1, 4, 6, 13, 300]))
    return np.frombuffer(bytes(b), dtype=np.uint8), np.array(o, dtype=np.uint32)


@pytest.mark.parametrize("mk", [synth.cfg2, synth.cfg3, synth.cfg5])
def test_mutations_agree_with_oracle(mk, path):
    from etl_amd.decoder import Decoder
    from oracle import oracle
    import zlib
    rng = random.Random(zlib.crc32(f"{mk.__name__}/{path}".encode()))   # fixed per (workload, path)
    w = mk()
    buf, offs = w.fill(96 << 10)
    n_err = 0
    for it in range(60):
        mb, mo = _mutate(rng, buf, offs)
        o, d = oracle.Oracle(), Decoder(0)
        w.register(o, ready=not w.cfg.emit_relations)
        w.register(d, ready=not w.cfg.emit_relations)
        rb, gb = o.decode(mb, mo), d.decode(mb, mo)
        e = gb.error
        got = (e.code, e.kind, e.description, e.frame_index) if e else (0, 0, "", -1)
        assert (rb.err_code, rb.err_kind, rb.err_desc, rb.err_frame) == got, (it, got)
        diff = rb.host_batch().diff(gb.host())
        assert not diff, (it, diff[:4])
        n_err += rb.err_code != 0
        d.close()
    assert n_err > 5   # the mutations do hit the error paths
