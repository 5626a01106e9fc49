"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. The product package (etl_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from etl_amd import abi
from etl_amd.view import HostBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_FULL, MODE_CONTRACT = 0, 1
DEFAULT_DEFER_MASK = 0   # classes deferred wholesale on top of json / arrays (floats follow the rule in include/etlg.h)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("etl_oracle.cpp", "oracle_codec.hpp")] + \
           [os.path.join(_HERE, "..", "include", "etlg.h"), os.path.join(_HERE, "..", "etl_amd", "csrc", "float_fast.h"),
            os.path.join(_HERE, "..", "etl_amd", "csrc", "pow5_table.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_ctx_create.restype = C.c_void_p
        L.oracle_ctx_destroy.argtypes = [C.c_void_p]
        L.oracle_ctx_set_mode.argtypes = [C.c_void_p, C.c_int32, C.c_uint32]
        L.oracle_ctx_set_worker.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64]
        L.oracle_ctx_reset_stream_state.argtypes = [C.c_void_p]
        L.oracle_schema_put.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_char_p,
                                        C.c_uint32, C.POINTER(abi.Col)]
        L.oracle_table_state.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint64]
        L.oracle_table_ready.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_void_p)]
        L.oracle_copy_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_void_p)]
        L.oracle_decode_timed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        L.oracle_decode_timed.restype = C.c_double
        L.oracle_last_error.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
        L.oracle_batch_view.argtypes = [C.c_void_p, C.POINTER(abi.BatchView)]
        L.oracle_batch_n_events.argtypes = [C.c_void_p]
        L.oracle_batch_n_events.restype = C.c_uint64
        L.oracle_batch_free.argtypes = [C.c_void_p]
        L.oracle_event_repr.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_event_repr.restype = C.c_char_p
        L.oracle_parse_text_cell.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t, C.c_int32, C.c_char_p, C.c_size_t]
        L.oracle_err_description.argtypes = [C.c_int32]
        L.oracle_err_description.restype = C.c_char_p
        L.oracle_err_kind.argtypes = [C.c_int32]
        L.oracle_class_of_oid.argtypes = [C.c_uint32]
        L.oracle_array_elem_class.argtypes = [C.c_uint32]
        L.oracle_slot_bytes.argtypes = [C.c_int32]
        L.oracle_slot_bytes.restype = C.c_uint32
        L.oracle_parse_utc_offset.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32)]
        _LIB = L
    return _LIB


def parse_text_cell(type_oid, text, check_utf8=True):
    """Full reference semantics for one text value -> repr string ('I32(5)' | 'Err(16)')."""
    b = text if isinstance(text, (bytes, bytearray)) else text.encode("utf-8", "surrogatepass")
    cap = 256 + 16 * max(len(b), 1)
    out = C.create_string_buffer(cap)
    lib().oracle_parse_text_cell(type_oid, bytes(b), len(b), 1 if check_utf8 else 0, out, cap)
    return out.value.decode("utf-8", "replace")


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Oracle:
    """Mirror of the apply-loop state an etlg_ctx holds."""

    def __init__(self, mode=MODE_CONTRACT, defer_mask=DEFAULT_DEFER_MASK):
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_ctx_create())
        self.L.oracle_ctx_set_mode(self.h, mode, defer_mask)
        self.mode = mode

    def close(self):
        if self.h:
            self.L.oracle_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_worker(self, worker=abi.WORKER_APPLY, table_id=0, bootstrap_lsn=0):
        self.L.oracle_ctx_set_worker(self.h, worker, table_id, bootstrap_lsn)

    def reset_stream_state(self):
        self.L.oracle_ctx_reset_stream_state(self.h)

    def schema_put(self, table_id, snapshot_lsn, cols, schema="public", name="t"):
        arr = abi.make_cols(cols)
        return self.L.oracle_schema_put(self.h, table_id, snapshot_lsn, schema.encode(), name.encode(), len(cols), arr)

    def table_state(self, table_id, kind, lsn=0):
        return self.L.oracle_table_state(self.h, table_id, kind, lsn)

    def table_ready(self, table_id, snapshot_lsn, repl_mask, ident_mask):
        r = np.asarray(repl_mask, dtype=np.uint8)
        i = np.asarray(ident_mask, dtype=np.uint8)
        return self.L.oracle_table_ready(self.h, table_id, snapshot_lsn, _ptr(r), _ptr(i), len(r))

    def cache_state(self, table_id):
        """SharedTableCache::get (table_cache.rs:99-102): None, or (kind 1 WaitingForRelation | 2 Ready, snapshot id, schema slot)."""
        k, sn, sl = C.c_int32(), C.c_uint64(), C.c_int32()
        self.L.oracle_cache_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
        if not self.L.oracle_cache_state(self.h, table_id, C.byref(k), C.byref(sn), C.byref(sl)):
            return None
        return k.value, sn.value, sl.value

    def table_forget(self, table_id):
        self.L.oracle_table_forget.argtypes = [C.c_void_p, C.c_uint32]
        self.L.oracle_table_forget.restype = None
        self.L.oracle_table_forget(self.h, table_id)

    def cache_tables(self):
        out = (C.c_uint32 * 256)()
        self.L.oracle_cache_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        self.L.oracle_cache_tables.restype = C.c_uint32
        n = self.L.oracle_cache_tables(self.h, out, 256)
        return [int(out[i]) for i in range(n)]

    @staticmethod
    def _prep(buf, offsets):
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint32)
        nfr = 0 if off is None else len(off) - 1
        return a, off, nfr

    def decode(self, buf, offsets=None):
        """buf: bytes/np.uint8. Returns OracleBatch (always; .err_code != 0 on failure)."""
        a, off, nfr = self._prep(buf, offsets)
        out = C.c_void_p()
        code = self.L.oracle_decode(self.h, _ptr(a), a.size, _ptr(off), nfr, C.byref(out))
        kind, desc, frame = C.c_int32(), C.c_char_p(), C.c_int64()
        self.L.oracle_last_error(self.h, C.byref(kind), C.byref(desc), C.byref(frame))
        return OracleBatch(self, out, code, kind.value, (desc.value or b"").decode(), frame.value, keep=(a, off, buf))

    def copy_decode(self, slot, buf, row_offsets):
        """Table-copy rows (COPY text format, one row per offsets interval) against schema slot `slot`."""
        a, off, nrows = self._prep(buf, row_offsets)
        out = C.c_void_p()
        code = self.L.oracle_copy_decode(self.h, slot, _ptr(a), a.size, _ptr(off), nrows, C.byref(out))
        kind, desc, frame = C.c_int32(), C.c_char_p(), C.c_int64()
        self.L.oracle_last_error(self.h, C.byref(kind), C.byref(desc), C.byref(frame))
        return OracleBatch(self, out, code, kind.value, (desc.value or b"").decode(), frame.value, keep=(a, off, buf))

    def decode_timed(self, buf, offsets=None):
        a, off, nfr = self._prep(buf, offsets)
        ne, nf, ec = C.c_uint64(), C.c_uint64(), C.c_int32()
        secs = self.L.oracle_decode_timed(self.h, _ptr(a), a.size, _ptr(off), nfr, C.byref(ne), C.byref(nf), C.byref(ec))
        return secs, ne.value, nf.value, ec.value


class OracleBatch:
    def __init__(self, orc, handle, code, kind, desc, frame, keep):
        self.orc, self.h = orc, handle
        self.err_code, self.err_kind, self.err_desc, self.err_frame = code, kind, desc, frame
        self._keep = keep  # Deferred cells point into the input buffer

    @property
    def n_events(self):
        return self.orc.L.oracle_batch_n_events(self.h)

    def host_batch(self):
        v = abi.BatchView()
        rc = self.orc.L.oracle_batch_view(self.h, C.byref(v))
        if rc != 0:
            raise RuntimeError("oracle arena is only defined for CONTRACT-mode batches")
        return HostBatch.from_view(v)

    def finish(self, what=3):
        """The finish pass of include/etlg.h on the canonical arena (typed arrays, exact floats): oracle_batch_finish. Returns the cells settled."""
        self.orc.L.oracle_batch_finish.restype = C.c_int64
        self.orc.L.oracle_batch_finish.argtypes = [C.c_void_p, C.c_uint32]
        n = self.orc.L.oracle_batch_finish(self.h, what)
        if n < 0:
            raise RuntimeError("oracle arena is only defined for CONTRACT-mode batches")
        return n

    def event_repr(self, i):
        return self.orc.L.oracle_event_repr(self.h, i).decode("utf-8", "replace")

    def close(self):
        if self.h:
            self.orc.L.oracle_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
